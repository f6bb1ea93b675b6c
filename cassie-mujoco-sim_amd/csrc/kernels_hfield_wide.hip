/* cassie_hfield.xml, the 127-row instantiation (kernels_cassie_wide.hip): what CM_FLAG_HFPRISM's one contact per penetrated grid
 * triangle needs on rough terrain */
#include "step_launch.h"
namespace ck {
bool launch_wide_cassie_hfield(dim3 grid, hipStream_t s, PhysIO io) {
    if (io.handover_list) hipLaunchKernelGGL((cassie_step_kernel<32, TopoCassie32, FEAT_HFIELD, WIDE_ROWS, 2, true, 1>), grid, dim3(2 * WV_WAVE), 0, s, io);
    else hipLaunchKernelGGL((cassie_step_kernel<32, TopoCassie32, FEAT_HFIELD, WIDE_ROWS, 2, false, 1>), grid, dim3(2 * WV_WAVE), 0, s, io);
    return hipGetLastError() == hipSuccess;
}
}  // namespace ck
