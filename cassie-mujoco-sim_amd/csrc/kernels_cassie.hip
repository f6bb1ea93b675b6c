/* the step kernel for plain cassie.xml (BASELINE configs 1-3): 32 dofs, compile-time topology, no height-field / box code */
#include "step_launch.h"
namespace ck {
bool launch_step_cassie(dim3 grid, const TierGrids &tg, hipStream_t s, PhysIO io, const HandoverLists &hl, bool fast, bool wide_caps, hipEvent_t after_first, int waves, bool inplace) {
    return launch_three_tiers<32, TopoCassie32, 0>(grid, tg, s, io, hl, fast, wide_caps, after_first, waves == 2 ? launch_fast_cassie_2w : nullptr, waves == 2 ? launch_mid_cassie_2w : nullptr, launch_wide_cassie, launch_alone63_cassie, inplace ? launch_fast_cassie_2w_inplace : nullptr);
}
}  // namespace ck
