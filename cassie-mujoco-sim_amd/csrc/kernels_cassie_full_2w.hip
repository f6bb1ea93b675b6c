/* plain cassie.xml, the 63-row instantiation in its two-wave form as the pass behind the two-wave fast kernel (step_launch.h): it
 * walks the list of envs the fast kernel handed over and hands on what needs more than 63 rows / 16 contacts */
#include "step_launch.h"
namespace ck {
bool launch_mid_cassie_2w(dim3 grid, hipStream_t s, PhysIO io) {
    hipLaunchKernelGGL((cassie_step_kernel<32, TopoCassie32, 0, MID_ROWS, 2, true>), grid, dim3(2 * WV_WAVE), 0, s, io);
    return hipGetLastError() == hipSuccess;
}
}  // namespace ck
