/* cassie_tray_box.xml (BASELINE config 5) in the two-wave form: two wavefronts per env, wave 1 running the mass-matrix stage group,
 * the drive-level pass, the factorisations and the bias / passive stage beside wave 0's collision (box-box by the whole wave),
 * velocity and constraint-row stages (physics_kernel.h, env_step) */
#include "step_launch.h"
namespace ck {
bool launch_step_tray_2w(dim3 grid, hipStream_t s, PhysIO io) {
    hipLaunchKernelGGL((cassie_step_kernel<40, TopoCassieTray38, FEAT_WAVEPAIRS, CM_MAXEFC, 2>), grid, dim3(2 * WV_WAVE), 0, s, io);
    return hipGetLastError() == hipSuccess;
}
}  // namespace ck
