/* plain cassie.xml, the 127-row instantiation: two wavefronts per env with 512 registers each, the solve of a substep with more than 64
 * rows spread over both (physics_kernel.h, wide_solve).  Alone -- forward / read-out passes, a cassie_sim_t, small batches -- or as the
 * pass that walks the list of envs the 63-row pass handed on (io.handover_list set). */
#include "step_launch.h"
namespace ck {
bool launch_wide_cassie(dim3 grid, hipStream_t s, PhysIO io) {
    if (io.handover_list) hipLaunchKernelGGL((cassie_step_kernel<32, TopoCassie32, 0, WIDE_ROWS, 2, true, 1>), grid, dim3(2 * WV_WAVE), 0, s, io);
    else hipLaunchKernelGGL((cassie_step_kernel<32, TopoCassie32, 0, WIDE_ROWS, 2, false, 1>), grid, dim3(2 * WV_WAVE), 0, s, io);
    return hipGetLastError() == hipSuccess;
}
}  // namespace ck
