/* cassie_tray_box.xml (BASELINE config 5), the row-capped instantiation (47 rows) with ONE wavefront per env and 512 registers:
 * the Gram matrix of the staged rows on the matrix core, through the staged tile's own LDS (physics_kernel.h, gram_in_place);
 * a substep with more rows hands the env over to the full instantiation behind it, which walks the hand-over list */
#include "step_launch.h"
namespace ck {
bool launch_fast_tray(dim3 grid, hipStream_t s, PhysIO io) {
    hipLaunchKernelGGL((cassie_step_kernel<40, TopoCassieTray38, FEAT_WAVEPAIRS, FAST_ROWS_TRAY>), grid, dim3(WV_WAVE), 0, s, io);
    return hipGetLastError() == hipSuccess;
}
/* the pass behind it: the full instantiation walking the hand-over list, one wavefront per env as well -- a workgroup of it fits
 * wherever a workgroup of the fast kernel has retired (the two-wave form of the pass wants two SIMDs with 256 free registers each
 * on one CU, and waited 4 ms for them behind the other env range's one-wave workgroups) */
bool launch_full_tray_walk(dim3 grid, hipStream_t s, PhysIO io) {
    hipLaunchKernelGGL((cassie_step_kernel<40, TopoCassieTray38, FEAT_WAVEPAIRS, MID_ROWS, 1, true>), grid, dim3(WV_WAVE), 0, s, io);
    return hipGetLastError() == hipSuccess;
}
}  // namespace ck
