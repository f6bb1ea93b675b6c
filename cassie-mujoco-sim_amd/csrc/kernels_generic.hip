/* the step kernel for any other model of the supported MJCF subset (dof tree read from the model at run time), and the
 * all-features instantiation of the Cassie topology (a Cassie model with both height-field and box pairs) */
#include "step_launch.h"
namespace ck {
bool launch_step_cassie_all(dim3 grid, hipStream_t s, PhysIO io) {
    no_tiers(io);
    hipLaunchKernelGGL((cassie_step_kernel<32, TopoCassie32, FEAT_ALL>), grid, dim3(WV_WAVE), 0, s, io);
    return hipGetLastError() == hipSuccess;
}
bool launch_step_generic(dim3 grid, hipStream_t s, PhysIO io, bool wide) {
    no_tiers(io);
    if (!wide) hipLaunchKernelGGL((cassie_step_kernel<32, TopoRuntime, FEAT_ALL>), grid, dim3(WV_WAVE), 0, s, io);
    else hipLaunchKernelGGL((cassie_step_kernel<40, TopoRuntime, FEAT_ALL>), grid, dim3(WV_WAVE), 0, s, io);
    return hipGetLastError() == hipSuccess;
}
}  // namespace ck
