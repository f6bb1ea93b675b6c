/* the step kernel for cassie_hfield.xml (BASELINE config 4): 32 dofs, compile-time topology, height-field pairs */
#include "step_launch.h"
namespace ck {
bool launch_step_cassie_hfield(dim3 grid, const TierGrids &tg, hipStream_t s, PhysIO io, const HandoverLists &hl, bool fast, bool wide_caps, hipEvent_t after_first, int waves, bool inplace) {
    return launch_three_tiers<32, TopoCassie32, FEAT_HFIELD>(grid, tg, s, io, hl, fast, wide_caps, after_first, waves == 2 ? launch_fast_cassie_hfield_2w : nullptr, waves == 2 ? launch_mid_cassie_hfield_2w : nullptr, launch_wide_cassie_hfield, launch_alone63_cassie_hfield, inplace ? launch_fast_cassie_hfield_2w_inplace : nullptr);
}
}  // namespace ck
