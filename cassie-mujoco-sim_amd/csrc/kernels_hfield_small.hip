/* cassie_hfield.xml, ALONE (see kernels_cassie_small.hip): the 63-row instantiation, two wavefronts per env, 512 registers a lane */
#include "step_launch.h"
namespace ck {
bool launch_alone63_cassie_hfield(dim3 grid, hipStream_t s, PhysIO io) {
    hipLaunchKernelGGL((cassie_step_kernel<32, TopoCassie32, FEAT_HFIELD, MID_ROWS, 2, false, 1>), grid, dim3(2 * WV_WAVE), 0, s, io);
    return hipGetLastError() == hipSuccess;
}
}  // namespace ck
