/* plain cassie.xml, the row-capped fast instantiation in its two-wave form: two wavefronts per env (128-thread workgroups, two
 * waves per SIMD at 256 registers each), wave 1 running the mass-matrix stage group beside wave 0's collision, velocity and
 * constraint-row stages (physics_kernel.h, env_step) */
#include "step_launch.h"
namespace ck {
bool launch_fast_cassie_2w(dim3 grid, hipStream_t s, PhysIO io) {
    hipLaunchKernelGGL((cassie_step_kernel<32, TopoCassie32, 0, FAST_ROWS, 2>), grid, dim3(2 * WV_WAVE), 0, s, io);
    return hipGetLastError() == hipSuccess;
}
}  // namespace ck
