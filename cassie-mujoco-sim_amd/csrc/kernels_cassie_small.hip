/* plain cassie.xml, ALONE -- forward / read-out passes, a single cassie_sim_t, small batches (step_launch.h) -- for models whose caps
 * are 63 rows: the 63-row instantiation with two wavefronts per env AND 512 registers a lane.  A batch that cannot fill the chip has
 * no use for the second workgroup per SIMD pair that the 256-register form makes room for, and at 512 the kernel keeps its row of A in
 * registers (no scratch). */
#include "step_launch.h"
namespace ck {
bool launch_alone63_cassie(dim3 grid, hipStream_t s, PhysIO io) {
    hipLaunchKernelGGL((cassie_step_kernel<32, TopoCassie32, 0, MID_ROWS, 2, false, 1>), grid, dim3(2 * WV_WAVE), 0, s, io);
    return hipGetLastError() == hipSuccess;
}
}  // namespace ck
