/* plain cassie.xml, the full instantiation (63 rows) in its two-wave form: the pass behind the two-wave fast kernel (step_launch.h) */
#include "step_launch.h"
namespace ck {
bool launch_full_cassie_2w(dim3 grid, hipStream_t s, PhysIO io) {
    if (io.handover_list) hipLaunchKernelGGL((cassie_step_kernel<32, TopoCassie32, 0, CM_MAXEFC, 2, true>), grid, dim3(2 * WV_WAVE), 0, s, io);
    else hipLaunchKernelGGL((cassie_step_kernel<32, TopoCassie32, 0, CM_MAXEFC, 2>), grid, dim3(2 * WV_WAVE), 0, s, io);
    return hipGetLastError() == hipSuccess;
}
}  // namespace ck
