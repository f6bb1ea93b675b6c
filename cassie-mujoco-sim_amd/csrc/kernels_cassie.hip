/* the step kernel for plain cassie.xml (BASELINE configs 1-3): 32 dofs, compile-time topology, no height-field / box code */
#include "step_launch.h"
namespace ck {
bool launch_step_cassie(dim3 grid, dim3 pass_grid, hipStream_t s, PhysIO io, bool fast, hipEvent_t after_first, int waves) {
    return launch_fast_then_full<32, TopoCassie32, 0>(grid, pass_grid, s, io, fast, after_first, waves == 2 ? launch_fast_cassie_2w : nullptr, waves == 2 ? launch_full_cassie_2w : nullptr, waves == 2 ? launch_full_cassie_small : nullptr);
}
}  // namespace ck
