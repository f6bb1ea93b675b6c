/* plain cassie.xml, the row-capped fast instantiation in its two-wave form WITH the 63-row code behind it in the same kernel
 * (cassie_step_kernel's INROWS, round 6): a substep that needs more than 31 rows is finished in place by the 63-row instantiation's
 * code and the env returns to the fast code -- no hand-over list, no pass behind the kernel for models whose caps are 63 rows */
#include "step_launch.h"
namespace ck {
bool launch_fast_cassie_2w_inplace(dim3 grid, hipStream_t s, PhysIO io) {
    hipLaunchKernelGGL((cassie_step_kernel<32, TopoCassie32, 0, FAST_ROWS, 2, false, 2, MID_ROWS>), grid, dim3(2 * WV_WAVE), 0, s, io);
    return hipGetLastError() == hipSuccess;
}
}  // namespace ck
