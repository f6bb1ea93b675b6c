/* cassie_tray_box.xml, the full instantiation (63 rows) in the two-wave form: alone, or as the list-walking pass behind the fast one */
#include "step_launch.h"
namespace ck {
bool launch_full_tray_2w(dim3 grid, hipStream_t s, PhysIO io) {
    if (io.handover_list) hipLaunchKernelGGL((cassie_step_kernel<40, TopoCassieTray38, FEAT_WAVEPAIRS, MID_ROWS, 2, true>), grid, dim3(2 * WV_WAVE), 0, s, io);
    else hipLaunchKernelGGL((cassie_step_kernel<40, TopoCassieTray38, FEAT_WAVEPAIRS, MID_ROWS, 2>), grid, dim3(2 * WV_WAVE), 0, s, io);
    return hipGetLastError() == hipSuccess;
}
}  // namespace ck
