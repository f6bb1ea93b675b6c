/* cassie_hfield.xml, small batches: the full instantiation with two wavefronts per env and 512 registers a lane (kernels_cassie_small.hip) */
#include "step_launch.h"
namespace ck {
bool launch_full_cassie_hfield_small(dim3 grid, hipStream_t s, PhysIO io) {
    hipLaunchKernelGGL((cassie_step_kernel<32, TopoCassie32, FEAT_HFIELD, CM_MAXEFC, 2, false, 1>), grid, dim3(2 * WV_WAVE), 0, s, io);
    return hipGetLastError() == hipSuccess;
}
}  // namespace ck
