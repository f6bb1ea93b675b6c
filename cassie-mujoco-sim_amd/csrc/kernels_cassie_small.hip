/* plain cassie.xml, small batches (a single cassie_sim_t: SMALL_BATCH envs at most, step_launch.h): the full instantiation with two
 * wavefronts per env AND 512 registers a lane -- a batch that cannot fill the chip has no use for the second workgroup per SIMD
 * pair that the 256-register form makes room for, and at 512 the kernel keeps its row of A in registers (no scratch) */
#include "step_launch.h"
namespace ck {
bool launch_full_cassie_small(dim3 grid, hipStream_t s, PhysIO io) {
    hipLaunchKernelGGL((cassie_step_kernel<32, TopoCassie32, 0, CM_MAXEFC, 2, false, 1>), grid, dim3(2 * WV_WAVE), 0, s, io);
    return hipGetLastError() == hipSuccess;
}
}  // namespace ck
