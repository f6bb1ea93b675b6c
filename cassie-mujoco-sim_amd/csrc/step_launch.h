/*
 * step_launch.h -- the step kernel's instantiations live in translation units of their own (kernels_*.hip), one per model
 * family, so that they compile side by side (each takes about a minute of hipcc time); phys_batch.hip picks one per launch
 * through these functions.  Every function launches the row-capped fast instantiation first where `fast` is set and one
 * exists (see ck::cassie_step_kernel), then the passes that finish what it handed over; returns false if a launch failed.
 */
#ifndef CASSIE_STEP_LAUNCH_H
#define CASSIE_STEP_LAUNCH_H

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#include "physics_kernel.h"

namespace ck {
/* The tiers of a stepping launch (round 5): the row-capped FAST instantiation (31 rows; 47 for the 40-dof model) steps every env
 * until a substep needs more rows or contacts than it holds; the MID instantiation (63 rows, 16 contacts) walks the list of envs the
 * fast one handed over; the WIDE instantiation (127 rows, 32 contacts: models on the 32-dof Cassie dof tree, cm_model_t::maxefc)
 * walks the list of envs the mid one handed on.  A launch that hands nothing over pays a few workgroup placements per pass. */
struct HandoverLists {          /* per env range: the two lists, their [count, ticket] pairs, the host words the passes report into */
    int *list1, *count1; volatile int *seen1;
    int *list2, *count2; volatile int *seen2;
};
struct TierGrids { dim3 mid, wide; };   /* grids of the passes behind the fast kernel (sized by what the range's last launch handed over) */

/* io.progress must be set when fast is; io.resume / has_next / the hand-over lists are managed here */
/* after_first (may be null): recorded behind the first kernel of the launch -- the one that does the work -- for per-kernel timing */
/* waves: 2 = the row-capped fast instantiation in its two-wave form (two wavefronts per env, see env_step), 1 = one wave per env */
/* inplace: the fast kernel in the form that finishes the substeps it cannot hold inside its own workgroups (kernels_*_2w_inplace.hip) */
bool launch_step_cassie(dim3 grid, const TierGrids &tg, hipStream_t s, PhysIO io, const HandoverLists &hl, bool fast, bool wide_caps, hipEvent_t after_first, int waves, bool inplace);            /* <32, TopoCassie32, 0>: plain cassie.xml */
bool launch_step_cassie_hfield(dim3 grid, const TierGrids &tg, hipStream_t s, PhysIO io, const HandoverLists &hl, bool fast, bool wide_caps, hipEvent_t after_first, int waves, bool inplace);     /* <32, TopoCassie32, FEAT_HFIELD> */
/* the two-wave forms of the fast instantiations, in translation units of their own (kernels_*_2w.hip) */
bool launch_fast_cassie_2w(dim3 grid, hipStream_t s, PhysIO io);
bool launch_fast_cassie_hfield_2w(dim3 grid, hipStream_t s, PhysIO io);
/* ... with the 63-row code behind them in the same kernel (kernels_*_2w_inplace.hip: cassie_step_kernel's INROWS) */
bool launch_fast_cassie_2w_inplace(dim3 grid, hipStream_t s, PhysIO io);
bool launch_fast_cassie_hfield_2w_inplace(dim3 grid, hipStream_t s, PhysIO io);
/* ... and of the 63-row instantiations in their role as the pass behind the fast kernel: there a workgroup must be placeable
 * wherever a fast kernel's is -- two waves of 256 registers, 40 KB of LDS -- or it waits for a SIMD to empty while the other env
 * range's kernel keeps every SIMD half full */
bool launch_mid_cassie_2w(dim3 grid, hipStream_t s, PhysIO io);
bool launch_mid_cassie_hfield_2w(dim3 grid, hipStream_t s, PhysIO io);
/* ... alone, for models whose caps are 63 rows (forward / read-out passes, a cassie_sim_t, small batches): two waves of 512 registers
 * (kernels_*_small.hip) -- a batch that cannot fill the chip has no use for the second workgroup per SIMD pair */
bool launch_alone63_cassie(dim3 grid, hipStream_t s, PhysIO io);
bool launch_alone63_cassie_hfield(dim3 grid, hipStream_t s, PhysIO io);
/* ... the 127-row instantiations: two waves of 512 registers, 84 KB of LDS (kernels_*_wide.hip) -- alone, or walking the second list */
bool launch_wide_cassie(dim3 grid, hipStream_t s, PhysIO io);
bool launch_wide_cassie_hfield(dim3 grid, hipStream_t s, PhysIO io);
bool launch_step_cassie_all(dim3 grid, hipStream_t s, PhysIO io);                   /* <32, TopoCassie32, FEAT_ALL> */
bool launch_step_tray(dim3 grid, dim3 pass_grid, hipStream_t s, PhysIO io, const HandoverLists &hl, bool hfield, bool fast, hipEvent_t after_first, int waves); /* <40, TopoCassieTray38, FEAT_WAVEPAIRS | FEAT_ALL> */
/* the 40-dof model's two-wave forms: the fast instantiation (FAST_ROWS_TRAY rows) and the 63-row one (alone, or as the list-walking pass) */
bool launch_fast_tray_2w(dim3 grid, hipStream_t s, PhysIO io);
bool launch_fast_tray(dim3 grid, hipStream_t s, PhysIO io);   /* one wave per env, 47 rows, Gram matrix on the matrix core (kernels_tray_fast.hip) */
bool launch_full_tray_walk(dim3 grid, hipStream_t s, PhysIO io); /* the one-wave 63-row instantiation walking the hand-over list behind it (io.handover_list set) */
bool launch_full_tray_2w(dim3 grid, hipStream_t s, PhysIO io);
bool launch_step_generic(dim3 grid, hipStream_t s, PhysIO io, bool wide);           /* <32 | 40, TopoRuntime, FEAT_ALL> */

/* measurement switches read from the environment (A/B runs on one GPU box): a switch that is set says so on stderr, once per
 * process -- CASSIE_DEBUG_SKIP_RESUME_PASS leaves handed-over envs unfinished, which must not happen silently */
inline bool measurement_switch(const char *name) {
    const bool on = getenv(name) != nullptr;
    if (on) fprintf(stderr, "cassie_phys: measurement switch %s is set -- results are not the product's%s\n", name,
                    name[13] == 'S' ? " (envs handed over by the fast kernel stay UNFINISHED)" : "");
    return on;
}

constexpr int SMALL_BATCH_NSUB = 4;  /* substeps per launch up to which a small batch skips the fast kernel + passes (three launches) for the 127-row kernel alone */
constexpr unsigned SMALL_BATCH = 512; /* envs up to which that holds (half the chip's workgroup slots) */

/* the grid of a fast kernel whose launch goes in chunks (PhysIO::nchunk): workgroups [k nenv, (k + 1) nenv) are chunk k */
inline dim3 chunked_grid(dim3 grid, const PhysIO &io) { return dim3(grid.x * (unsigned)(io.nchunk > 1 ? io.nchunk : 1)); }

inline void no_tiers(PhysIO &io) {
    io.progress = nullptr; io.resume = 0; io.has_next = 0; io.nchunk = 1;
    io.handover_list = nullptr; io.handover_count = nullptr; io.handover_seen = nullptr; io.handover_out_list = nullptr; io.handover_out_count = nullptr;
}

/* a model on the 32-dof Cassie dof tree: fast -> mid (-> wide where the model's caps are 127 rows: hl.list2 set), or one instantiation alone */
template <int NVP, class TOPO, int FEAT>
inline bool launch_three_tiers(dim3 grid, const TierGrids &tg, hipStream_t s, PhysIO io, const HandoverLists &hl, bool fast, bool wide_caps, hipEvent_t after_first,
                               bool (*fast_2w)(dim3, hipStream_t, PhysIO), bool (*mid_2w)(dim3, hipStream_t, PhysIO), bool (*wide)(dim3, hipStream_t, PhysIO),
                               bool (*alone63)(dim3, hipStream_t, PhysIO), bool (*fast_2w_inplace)(dim3, hipStream_t, PhysIO) = nullptr) {
    /* a small batch stepping a few substeps per launch (somebody's control loop around a handful of envs): one launch instead of two
     * or three -- a launch costs what four substeps' difference between the kernels saves */
    if (fast && grid.x <= SMALL_BATCH && io.nsub <= SMALL_BATCH_NSUB) fast = false;
    /* (measurement aid: CASSIE_DEBUG_SKIP_RESUME_PASS -- what the passes behind the fast kernel cost; handed-over envs are then
     * left unfinished, so only for workloads that hand nothing over) */
    static const bool skip_passes = measurement_switch("CASSIE_DEBUG_SKIP_RESUME_PASS");
    if (!fast || !hl.list1 || (wide_caps && !hl.list2)) {
        /* alone: forward / read-out passes, batches with the read-out enabled such as a cassie_sim_t, the fast kernel switched off */
        no_tiers(io);
        /* (a LARGE grid alone -- phys_batch_derive / forward passes of a whole batch, the fast kernel switched off -- and 63-row caps:
         * the one-wave form, whose 421 registers leave room for four envs per CU; the two-wave 512-register form halves that and only
         * pays where the chip is not full anyway.  CASSIE_ALONE_512: the A/B switch) */
        static const bool alone512 = measurement_switch("CASSIE_ALONE_512");
        if (!wide_caps && grid.x > SMALL_BATCH && !alone512) hipLaunchKernelGGL((cassie_step_kernel<NVP, TOPO, FEAT>), grid, dim3(WV_WAVE), 0, s, io);
        else if (!(wide_caps ? wide : alone63)(grid, s, io)) return false;
        if (after_first) (void)hipEventRecord(after_first, s);
        return hipGetLastError() == hipSuccess;
    }
    /* Round 6: the fast kernel that finishes the substeps it cannot hold IN PLACE (the 63-row code inside the same workgroup): no
     * list, no 63-row pass; with the wide caps the inner 63-row call hands on to the second list, which the 127-row pass walks.
     * The caller (phys_batch.hip) picks this form per env range and launch: see phys_batch_set_inplace. */
    if (fast_2w && fast_2w_inplace) {
        io.resume = 0; io.has_next = 1;
        io.inplace_count = hl.count1;       /* (the first list's count word is free in this form: it counts the env-launches that needed the wider code) */
        io.handover_list = nullptr; io.handover_count = nullptr; io.handover_seen = nullptr;
        io.handover_out_list = nullptr; io.handover_out_count = nullptr;
        /* once in the 63-row code an env stays there until a substep needs at most FAST_ROWS - 4 rows again (the margin keeps an env
         * that hovers about the fast code's capacity from changing codes every substep); CASSIE_INPLACE_STAY_ROWS: the A/B switch, 0 =
         * back to the fast code after every substep */
        static const int stay_rows = getenv("CASSIE_INPLACE_STAY_ROWS") ? atoi(getenv("CASSIE_INPLACE_STAY_ROWS")) : FAST_ROWS - 4;
        io.inplace_stay_rows = stay_rows < 0 ? 0 : (stay_rows > FAST_ROWS ? FAST_ROWS : stay_rows);
        io.inplace_has_next = wide_caps ? 1 : 0;
        io.inplace_out_list = wide_caps ? hl.list2 : nullptr; io.inplace_out_count = wide_caps ? hl.count2 : nullptr;
        if (!fast_2w_inplace(chunked_grid(grid, io), s, io)) return false;
        if (after_first) (void)hipEventRecord(after_first, s);
        if (!wide_caps || skip_passes) return hipGetLastError() == hipSuccess;
        io.resume = 1; io.nchunk = 1; io.has_next = 0; io.handover_out_list = nullptr; io.handover_out_count = nullptr;
        io.handover_list = hl.list2; io.handover_count = hl.count2; io.handover_seen = hl.seen2;
        return wide(tg.wide, s, io);
    }
    /* the fast kernel: every env of the launch (in chunks, perhaps) */
    const bool walk1 = mid_2w != nullptr;   /* (the one-wave form's 63-row pass looks every env's record up instead) */
    io.resume = 0; io.has_next = 1;
    io.handover_list = nullptr; io.handover_count = nullptr; io.handover_seen = nullptr;
    io.handover_out_list = walk1 ? hl.list1 : nullptr; io.handover_out_count = walk1 ? hl.count1 : nullptr;
    const dim3 fast_grid = chunked_grid(grid, io);
    if (fast_2w) { if (!fast_2w(fast_grid, s, io)) return false; }
    else hipLaunchKernelGGL((cassie_step_kernel<NVP, TOPO, FEAT, FAST_ROWS>), fast_grid, dim3(WV_WAVE), 0, s, io);
    if (hipGetLastError() != hipSuccess) return false;
    if (after_first) (void)hipEventRecord(after_first, s);
    if (skip_passes) return true;
    /* the 63-row pass: walks the first list (two-wave form), or one workgroup per env that looks its env's record up (one-wave form);
     * with the wide caps it hands on to the second list, otherwise 63 rows are the model's cap and it is the last */
    io.resume = 1; io.nchunk = 1; io.has_next = wide_caps ? 1 : 0;
    io.handover_out_list = wide_caps ? hl.list2 : nullptr; io.handover_out_count = wide_caps ? hl.count2 : nullptr;
    if (walk1) {
        io.handover_list = hl.list1; io.handover_count = hl.count1; io.handover_seen = hl.seen1;
        if (!mid_2w(tg.mid, s, io)) return false;
    } else hipLaunchKernelGGL((cassie_step_kernel<NVP, TOPO, FEAT>), grid, dim3(WV_WAVE), 0, s, io);
    if (hipGetLastError() != hipSuccess) return false;
    if (!wide_caps) return true;
    /* the 127-row pass: walks the second list.  (Its workgroups need two empty SIMDs and 84 KB of LDS: on a busy chip even an empty pass
     * waits for the other env range's kernel to drain -- which is why it is launched only for models that can need it.) */
    io.has_next = 0; io.handover_out_list = nullptr; io.handover_out_count = nullptr;
    io.handover_list = hl.list2; io.handover_count = hl.count2; io.handover_seen = hl.seen2;
    return wide(tg.wide, s, io);
}
}  // namespace ck
#endif
