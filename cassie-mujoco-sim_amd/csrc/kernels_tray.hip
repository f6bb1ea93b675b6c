/* the step kernel for cassie_tray_box.xml (BASELINE config 5): 40-dof instantiation (38 used), block-dense factor rows,
 * plane-box / box-box pairs handled by the whole wave; with or without height-field pairs */
#include "step_launch.h"
namespace ck {
bool launch_step_tray(dim3 grid, hipStream_t s, PhysIO io, bool hfield, int waves) {
    io.progress = nullptr; io.resume = 0; io.handover_list = nullptr;
    if (waves == 2 && !hfield) return launch_step_tray_2w(grid, s, io);
    if (!hfield) hipLaunchKernelGGL((cassie_step_kernel<40, TopoCassieTray38, FEAT_WAVEPAIRS>), grid, dim3(WV_WAVE), 0, s, io);
    else hipLaunchKernelGGL((cassie_step_kernel<40, TopoCassieTray38, FEAT_ALL>), grid, dim3(WV_WAVE), 0, s, io);
    return hipGetLastError() == hipSuccess;
}
}  // namespace ck
