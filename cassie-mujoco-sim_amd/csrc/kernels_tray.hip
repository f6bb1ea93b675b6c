/* the step kernel for cassie_tray_box.xml (BASELINE config 5): 40-dof instantiation (38 used), block-dense factor rows,
 * plane-box / box-box pairs handled by the whole wave; with or without height-field pairs */
#include "step_launch.h"
namespace ck {
bool launch_step_tray(dim3 grid, dim3 pass_grid, hipStream_t s, PhysIO io, const HandoverLists &hl, bool hfield, bool fast, hipEvent_t after_first, int waves) {
    if ((waves == 2 || fast) && !hfield) {
        /* the fast instantiation first (when asked for: one wave per env with its Gram matrix on the matrix core, or two waves per
         * env), the full one in its two-wave form behind it, walking the hand-over list -- or alone (two waves per env) */
        if (fast && hl.list1) {
            /* (the model's caps are 63 rows / 16 contacts, cm_model_t::maxefc: two tiers) */
            io.resume = 0; io.has_next = 1;
            io.handover_list = nullptr; io.handover_count = nullptr; io.handover_seen = nullptr;
            io.handover_out_list = hl.list1; io.handover_out_count = hl.count1;
            const dim3 fast_grid = chunked_grid(grid, io);
            if (!(waves == 2 ? launch_fast_tray_2w(fast_grid, s, io) : launch_fast_tray(fast_grid, s, io))) return false;
            if (after_first) { (void)hipEventRecord(after_first, s); after_first = nullptr; }
            io.resume = 1; io.nchunk = 1; io.has_next = 0;
            io.handover_out_list = nullptr; io.handover_out_count = nullptr;
            io.handover_list = hl.list1; io.handover_count = hl.count1; io.handover_seen = hl.seen1;
            /* the pass in the form of the kernel it follows (its workgroups must fit where that kernel's retire) */
            static const bool pass_2w = measurement_switch("CASSIE_DEBUG_TRAY_PASS_TWO_WAVES"); /* (measurement aid) */
            if (!(waves == 2 || pass_2w ? launch_full_tray_2w(pass_grid, s, io) : launch_full_tray_walk(pass_grid, s, io))) return false;
        } else {
            no_tiers(io);
            if (!launch_full_tray_2w(grid, s, io)) return false;
        }
        if (after_first) (void)hipEventRecord(after_first, s);
        return true;
    }
    no_tiers(io);
    if (!hfield) hipLaunchKernelGGL((cassie_step_kernel<40, TopoCassieTray38, FEAT_WAVEPAIRS>), grid, dim3(WV_WAVE), 0, s, io);
    else hipLaunchKernelGGL((cassie_step_kernel<40, TopoCassieTray38, FEAT_ALL>), grid, dim3(WV_WAVE), 0, s, io);
    if (after_first) (void)hipEventRecord(after_first, s);
    return hipGetLastError() == hipSuccess;
}
}  // namespace ck
