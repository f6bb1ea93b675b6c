/*
 * step_launch.h -- the step kernel's instantiations live in translation units of their own (kernels_*.hip), one per model
 * family, so that they compile side by side (each takes about a minute of hipcc time); phys_batch.hip picks one per launch
 * through these functions.  Every function launches the row-capped fast instantiation first where `fast` is set and one
 * exists (see ck::cassie_step_kernel), then the full instantiation; returns false if a launch failed.
 */
#ifndef CASSIE_STEP_LAUNCH_H
#define CASSIE_STEP_LAUNCH_H

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#include "physics_kernel.h"

namespace ck {
/* io.progress must be set when fast is; io.resume is managed here */
/* after_first (may be null): recorded behind the first kernel of the launch -- the one that does the work -- for per-kernel timing */
/* waves: 2 = the row-capped fast instantiation in its two-wave form (two wavefronts per env, see env_step), 1 = one wave per env */
/* pass_grid: the grid of the pass behind the fast kernel (= grid, or fewer workgroups when it walks the hand-over list) */
bool launch_step_cassie(dim3 grid, dim3 pass_grid, hipStream_t s, PhysIO io, bool fast, hipEvent_t after_first, int waves);            /* <32, TopoCassie32, 0>: plain cassie.xml */
bool launch_step_cassie_hfield(dim3 grid, dim3 pass_grid, hipStream_t s, PhysIO io, bool fast, hipEvent_t after_first, int waves);     /* <32, TopoCassie32, FEAT_HFIELD> */
/* the two-wave forms of the fast instantiations, in translation units of their own (kernels_*_2w.hip) */
bool launch_fast_cassie_2w(dim3 grid, hipStream_t s, PhysIO io);
bool launch_fast_cassie_hfield_2w(dim3 grid, hipStream_t s, PhysIO io);
/* ... and of the full instantiations in their role as the pass behind the fast kernel (io.resume = 1): there a workgroup must be
 * placeable wherever a fast kernel's is -- two waves of 256 registers -- or it waits for a SIMD to empty while the other
 * env range's kernel keeps every SIMD half full (the one-wave full kernel holds 421 registers) */
bool launch_full_cassie_2w(dim3 grid, hipStream_t s, PhysIO io);
bool launch_full_cassie_hfield_2w(dim3 grid, hipStream_t s, PhysIO io);
/* ... and for batches of at most SMALL_BATCH envs alone: two waves of 512 registers (kernels_*_small.hip) */
bool launch_full_cassie_small(dim3 grid, hipStream_t s, PhysIO io);
bool launch_full_cassie_hfield_small(dim3 grid, hipStream_t s, PhysIO io);
bool launch_step_cassie_all(dim3 grid, hipStream_t s, PhysIO io);                   /* <32, TopoCassie32, FEAT_ALL> */
bool launch_step_tray(dim3 grid, dim3 pass_grid, hipStream_t s, PhysIO io, bool hfield, bool fast, hipEvent_t after_first, int waves); /* <40, TopoCassieTray38, FEAT_WAVEPAIRS | FEAT_ALL> */
/* the 40-dof model's two-wave forms: the fast instantiation (FAST_ROWS_TRAY rows) and the full one (alone, or as the list-walking pass) */
bool launch_fast_tray_2w(dim3 grid, hipStream_t s, PhysIO io);
bool launch_fast_tray(dim3 grid, hipStream_t s, PhysIO io);   /* one wave per env, 47 rows, Gram matrix on the matrix core (kernels_tray_fast.hip) */
bool launch_full_tray_walk(dim3 grid, hipStream_t s, PhysIO io); /* the one-wave full instantiation walking the hand-over list behind it (io.handover_list set) */
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

constexpr int SMALL_BATCH_NSUB = 4;  /* ... and substeps per launch up to which such a batch skips the row-capped fast kernel + pass pair (two launches) */
constexpr unsigned SMALL_BATCH = 512; /* envs up to which the full kernel alone runs in its two-wave form (half the chip's workgroup slots) */

/* the grid of a fast kernel whose launch goes in chunks (PhysIO::nchunk): workgroups [k nenv, (k + 1) nenv) are chunk k */
inline dim3 chunked_grid(dim3 grid, const PhysIO &io) { return dim3(grid.x * (unsigned)(io.nchunk > 1 ? io.nchunk : 1)); }

template <int NVP, class TOPO, int FEAT>
inline bool launch_fast_then_full(dim3 grid, dim3 pass_grid, hipStream_t s, PhysIO io, bool fast, hipEvent_t after_first, bool (*fast_2w)(dim3, hipStream_t, PhysIO),
                                  bool (*full_2w)(dim3, hipStream_t, PhysIO), bool (*full_small)(dim3, hipStream_t, PhysIO)) {
    /* a small batch stepping a few substeps per launch (somebody's control loop around a handful of envs): one launch of the full
     * kernel in its small-batch form instead of two -- a launch costs what four substeps' difference between the kernels saves */
    if (fast && full_small && grid.x <= SMALL_BATCH && io.nsub <= SMALL_BATCH_NSUB) fast = false;
    if (!fast) io.nchunk = 1;
    /* (measurement aids: CASSIE_DEBUG_SKIP_RESUME_PASS -- what the pass behind the fast kernel costs; handed-over envs are then
     * left unfinished, so only for workloads that hand nothing over; CASSIE_DEBUG_RESUME_ONE_WAVE -- the pass behind a two-wave
     * fast kernel as one-wave workgroups) */
    static const bool skip_resume = measurement_switch("CASSIE_DEBUG_SKIP_RESUME_PASS");
    static const bool resume_one_wave = measurement_switch("CASSIE_DEBUG_RESUME_ONE_WAVE");
    /* the hand-over list is kept only when the pass behind the fast kernel walks it (and clears its count): a fast kernel that
     * appended to a list nobody clears would run past the list's end after a few launches */
    const bool walk = fast && full_2w && !resume_one_wave && !skip_resume;
    if (!walk) { io.handover_list = nullptr; pass_grid = grid; }
    if (fast) {
        io.resume = 0;
        const dim3 fast_grid = chunked_grid(grid, io);   /* (a launch in chunks: one workgroup per env and chunk) */
        if (fast_2w) { if (!fast_2w(fast_grid, s, io)) return false; }
        else hipLaunchKernelGGL((cassie_step_kernel<NVP, TOPO, FEAT, FAST_ROWS>), fast_grid, dim3(WV_WAVE), 0, s, io);
        if (hipGetLastError() != hipSuccess) return false;
        if (after_first) { (void)hipEventRecord(after_first, s); after_first = nullptr; }
        io.resume = 1; io.nchunk = 1;
    } else {
        io.progress = nullptr; io.resume = 0; io.handover_list = nullptr;
        pass_grid = grid;
    }
    if (fast && skip_resume) {}
    else if (walk) { if (!full_2w(pass_grid, s, io)) return false; }
    else if (!fast && full_2w && grid.x <= SMALL_BATCH) {
        /* the full kernel alone (forward / read-out passes, batches with the read-out enabled such as a cassie_sim_t) on a
         * batch too small to fill the chip: latency is what counts, and two wavefronts per env cut it by a fifth */
        io.handover_list = nullptr;
        if (!(full_small ? full_small : full_2w)(grid, s, io)) return false;
    }
    else { /* the one-wave full kernel: one workgroup per env of the launch (no list walk) */
        io.handover_list = nullptr;
        hipLaunchKernelGGL((cassie_step_kernel<NVP, TOPO, FEAT>), grid, dim3(WV_WAVE), 0, s, io);
    }
    if (after_first) (void)hipEventRecord(after_first, s);
    return hipGetLastError() == hipSuccess;
}
}  // namespace ck
#endif
