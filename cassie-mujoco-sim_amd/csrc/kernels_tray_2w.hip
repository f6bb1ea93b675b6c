/* cassie_tray_box.xml (BASELINE config 5), the fast instantiation (47 rows) in the two-wave form: two wavefronts per env, wave 1
 * running the mass-matrix stage group, the drive-level pass, the factorisations, the bias / passive stage and the stages behind
 * the solve beside wave 0's collision (box-box by the whole wave), velocity, constraint-row and solve stages (physics_kernel.h) */
#include "step_launch.h"
namespace ck {
bool launch_fast_tray_2w(dim3 grid, hipStream_t s, PhysIO io) {
    hipLaunchKernelGGL((cassie_step_kernel<40, TopoCassieTray38, FEAT_WAVEPAIRS, FAST_ROWS_TRAY, 2>), grid, dim3(2 * WV_WAVE), 0, s, io);
    return hipGetLastError() == hipSuccess;
}
}  // namespace ck
