/* the step kernel for cassie_hfield.xml (BASELINE config 4): 32 dofs, compile-time topology, height-field pairs */
#include "step_launch.h"
namespace ck {
bool launch_step_cassie_hfield(dim3 grid, dim3 pass_grid, hipStream_t s, PhysIO io, bool fast, hipEvent_t after_first, int waves) {
    return launch_fast_then_full<32, TopoCassie32, FEAT_HFIELD>(grid, pass_grid, s, io, fast, after_first, waves == 2 ? launch_fast_cassie_hfield_2w : nullptr, waves == 2 ? launch_full_cassie_hfield_2w : nullptr, waves == 2 ? launch_full_cassie_hfield_small : nullptr);
}
}  // namespace ck
