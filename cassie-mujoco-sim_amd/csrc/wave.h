/*
 * wave.h -- the handful of wavefront primitives the physics kernel is written
 * against (gfx950, wave64, ONE wavefront per workgroup = one Cassie env).
 *
 * Everything cross-lane in physics_kernel.h goes through these wrappers so the
 * kernel's control flow can be read (and unit-tested, see tests/emu/) without
 * chasing builtins.
 */
#ifndef CASSIE_WAVE_H
#define CASSIE_WAVE_H

#include <hip/hip_runtime.h>

#define WV_DEVICE __device__ __forceinline__
#define WV_GLOBAL __global__
#define WV_SHARED __shared__
#define WV_WAVE 64
/* at least n of a kernel's waves resident per SIMD (512 / n registers per lane); n = 1 leaves the compiler's default */
#define WV_WAVES_PER_SIMD(n) __attribute__((amdgpu_waves_per_eu(n)))
/* address-space qualifier of the model pointer: nothing in the kernel writes the model, and the constant address
 * space lets the wave-uniform reads (sizes, options, solver parameters, pair-loop bounds) issue as scalar loads
 * even after the kernel has stored to global memory -- with a generic pointer every one of them is a 64-lane
 * vector load of a single address.  (Measured: 0.474 -> 0.440 ms per 4096-env step.) */
#define WV_CONST_AS __attribute__((address_space(4)))

namespace wv {

WV_DEVICE int lane() { return (int)(threadIdx.x & 63u); }
/* Two-wave workgroups (the step kernel's NW = 2 form: one env = two wavefronts that work on different stage groups of the
 * same substep): which wave this is (wave-uniform, in a scalar register), and the barrier between the workgroup's waves.
 * The barrier drains this wave's LDS traffic (so the other wave sees what was written) but NOT its vector-memory loads:
 * model constants requested a stage ahead stay in flight across it. */
WV_DEVICE int wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }
/* One-directional hand-over between the two waves of a workgroup through a word in LDS, where a barrier would make the producer
 * wait for the consumer: publish() after the data (this wave's LDS writes are complete before the word changes), wait_for() before
 * reading it (polls, sleeping a few clocks between reads; both waves are resident, so the producer always gets to run). */
WV_DEVICE void publish(int *flag, int value) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    if (threadIdx.x % 64u == 0) *(volatile int *)flag = value;
}
WV_DEVICE void wait_for(const int *flag, int value) {
    while (__builtin_amdgcn_readfirstlane(*(const volatile int *)flag) != value) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
/* the same without the s_sleep between two looks: for a wave that has its SIMD to itself (the 127-row instantiation) */
WV_DEVICE void wait_for_spin(const int *flag, int value) {
    while (__builtin_amdgcn_readfirstlane(*(const volatile int *)flag) != value) {}
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
/* every vector memory operation of this wave has completed (loads returned, stores acknowledged) */
WV_DEVICE void drain_vmem() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
WV_DEVICE void block_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
/* Hand-over between WORKGROUPS of one kernel through a word in device memory (PhysIO::chunk_flag): publish_global() once every
 * store of the workgroup is on its way (each wave waits for its own, the waves meet, one lane writes the word); wait_global()
 * polls the word -- the producer is a workgroup with a lower number, dispatched earlier, so it runs or has run -- and then drops
 * this CU's cached copies of what the producer wrote before anybody loads.
 * The word is 64 tag + 8 XCC + done: the producer states WHICH XCD's L2 its stores went to (they are released at workgroup scope:
 * they have arrived in that L2, nothing writes it back), and the consumer, who runs on the same XCD when workgroup w is placed on
 * XCD w % 8 (phys_batch.hip probes that, per stream), CHECKS it: wait_global returns false when the producer's XCD is not this
 * workgroup's -- the caller then flags the env and tells the launcher, which stops chunking (a CU-masked stream or another
 * partition mode broke the placement rule). */
WV_DEVICE int xcc_id() { unsigned x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x)); return (int)(x & 7u); }
template <int NW> WV_DEVICE void publish_global(int *flag, int tag, int done) {
#ifdef CK_CHUNK_RELEASE_AGENT
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#else
    /* (the consumer runs on the same XCD and so shares this workgroup's L2: the stores only have to have arrived there, which the
     * counter wait of a workgroup-scope release says; an agent-scope release would write the XCD's whole L2 back, 25 k clocks a time) */
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    /* (EVERY wave's stores, before the waves meet: at workgroup scope the compiler waits for the publishing wave's only) */
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    if (NW > 1) __builtin_amdgcn_s_barrier();
    if (threadIdx.x == 0) __hip_atomic_store(flag, 64 * tag + 8 * xcc_id() + done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int NW> WV_DEVICE bool wait_global(const int *flag, int tag, int done) {
    int word = 0;
    if (threadIdx.x < 64u) {
        for (;;) {
            word = __builtin_amdgcn_readfirstlane(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            if ((word >> 6) == tag && (word & 7) == done) break;
            __builtin_amdgcn_s_sleep(20);
        }
    }
    if (NW > 1) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __builtin_amdgcn_s_dcache_inv(); /* (the scalar cache too, should the compiler ever read a wave-uniform word of the env's state through it) */
#ifdef CK_CHUNK_RELEASE_AGENT
    return true;
#else
    return threadIdx.x >= 64u || ((word >> 3) & 7) == xcc_id(); /* (wave 0 holds the verdict) */
#endif
}
/* the lane index recomputed from nothing (two VALU ops) through an asm the optimiser cannot merge or hoist: values
 * derived from it (LDS addresses, lane predicates) then live only inside the stage that asked, instead of being
 * computed once at the top of the kernel and carried -- i.e. spilled -- across everything in between */
WV_DEVICE int fresh_lane() {
    int x;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(x));
    return x;
}
/* the wave's priority in its SIMD's instruction arbitration (0 .. 3): a SIMD hosts the wave 0 of one env and the wave 1 of another,
 * and a wave in a stretch the other wave of ITS env does not wait for can yield its issue slots to whoever shares the SIMD.
 * Measured (profiles/round5/wave_priority_ab.txt; against no s_setprio at all): config 2 +0.6 %, config 4 +1.6 % (+3.0 % with wave
 * 1 low from F to J there), config 5 +1.1 %; wave 0 at the top up to the barrier F: another +1.1 % on config 2 (none on config 5, -1 % on
 * config 4: off there); wave 0 higher in its PGS sweeps, or wave 1 low from F to J on cassie.xml: -0.5 .. -1 %. */
#ifndef CK_PRIO
#define CK_PRIO 1
#endif
#ifndef CK_PRIO_W0          /* wave 0, all of its substep */
#define CK_PRIO_W0 1
#endif
#ifndef CK_PRIO_W0_KIN      /* ... the stretch up to the barrier F (guard, drive I/O in, kinematics), which wave 1 waits for */
#define CK_PRIO_W0_KIN 3
#endif
#ifndef CK_PRIO_W0_PGS      /* ... its PGS sweeps */
#define CK_PRIO_W0_PGS CK_PRIO_W0
#endif
#ifndef CK_PRIO_W1_FJ       /* wave 1 between the barriers F and J */
#define CK_PRIO_W1_FJ 1
#endif
#ifndef CK_PRIO_W1_FJ_HFIELD /* ... in the height-field instantiations, whose wave 0 is 13 k clocks longer on that stretch */
#define CK_PRIO_W1_FJ_HFIELD 0
#endif
#ifndef CK_PRIO_W1_JP       /* ... J and P */
#define CK_PRIO_W1_JP 0
#endif
#ifndef CK_PRIO_W1_PE       /* ... P and E: the tail wave 0 waits for */
#define CK_PRIO_W1_PE 3
#endif
#ifndef CK_PRIO_W1_EF       /* ... E and F: the outputs, then the wait for wave 0's kinematics */
#define CK_PRIO_W1_EF 0
#endif
template <int P> WV_DEVICE void set_priority() { if (CK_PRIO) __builtin_amdgcn_s_setprio(P); }
WV_DEVICE int env_id() { return (int)blockIdx.x; }
WV_DEVICE int grid_size() { return (int)gridDim.x; }
/* device-scope atomic add on an int in global memory, returns the old value */
WV_DEVICE int atomic_add(int *p, int v) { return atomicAdd(p, v); }
WV_DEVICE int atomic_or(int *p, int v) { return atomicOr(p, v); }

/* Orders LDS traffic between the lanes of the wave.  The workgroup IS one wave, whose LDS instructions are issued and
 * executed in program order, so a later read already sees an earlier write of any lane: all that is needed is that
 * the compiler keeps the order (a wavefront-scope fence, no instructions).  __syncthreads() would add
 * `s_waitcnt vmcnt(0) lgkmcnt(0)` -- a full drain of LDS and vector memory -- at each of the ~40 stage boundaries. */
WV_DEVICE void sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

WV_DEVICE double shfl(double v, int src_lane) { return __shfl(v, src_lane, WV_WAVE); }
WV_DEVICE double shfl_xor(double v, int mask) { return __shfl_xor(v, mask, WV_WAVE); }
WV_DEVICE int shfl_i(int v, int src_lane) { return __shfl(v, src_lane, WV_WAVE); }

/* v_mfma_f64_16x16x4_f64: D = A B + C on the matrix core, A 16 x 4, B 4 x 16, C / D 16 x 16 spread over the wave -- lane l
 * supplies A[l & 15][l >> 4] and B[l >> 4][l & 15] and holds C / D[(l >> 4) + 4 v][l & 15] in c[v], v = 0 .. 3.  Every
 * element is the plain FMA chain over k = 0 .. 3 on top of C, bit for bit (tools/mfma_f64_probe.hip), at 64 clocks
 * per instruction whether or not the next one depends on it.  All 64 lanes must be active.  Callers keep three or four
 * independent accumulations in flight (the _x3 / _x4 forms) so that a dependent instruction never follows directly;
 * mfma_f64_drain is a no-op on the device (the compiler inserts the wait states in front of the first read of a result). */
struct mfma_acc { double c[4]; };
WV_DEVICE void mfma_f64_16x16x4(double a, double b, mfma_acc &c) {
    typedef double d4 __attribute__((ext_vector_type(4)));
    d4 x = {c.c[0], c.c[1], c.c[2], c.c[3]};
    x = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, x, 0, 0, 0);
    c.c[0] = x[0]; c.c[1] = x[1]; c.c[2] = x[2]; c.c[3] = x[3];
}
WV_DEVICE void mfma_f64_16x16x4_x3(double a0, double b0, mfma_acc &c0, double a1, double b1, mfma_acc &c1, double a2, double b2, mfma_acc &c2) {
    mfma_f64_16x16x4(a0, b0, c0); mfma_f64_16x16x4(a1, b1, c1); mfma_f64_16x16x4(a2, b2, c2);
}
WV_DEVICE void mfma_f64_16x16x4_x4(double a0, double b0, mfma_acc &c0, double a1, double b1, mfma_acc &c1, double a2, double b2, mfma_acc &c2,
                                   double a3, double b3, mfma_acc &c3) {
    mfma_f64_16x16x4(a0, b0, c0); mfma_f64_16x16x4(a1, b1, c1); mfma_f64_16x16x4(a2, b2, c2); mfma_f64_16x16x4(a3, b3, c3);
}
WV_DEVICE void mfma_f64_drain(mfma_acc &, mfma_acc &, mfma_acc &) {}
WV_DEVICE void mfma_f64_drain4(mfma_acc &, mfma_acc &, mfma_acc &, mfma_acc &) {}

/* for lanes 0..31: the value lane + 32 holds (the upper lanes get their own value back): one v_permlane32_swap_b32 per
 * dword, no LDS crossbar.  Used where the two halves of the wave each do half of a lane's work (dense loops whose work
 * items number at most 32) and the lower half collects the results. */
WV_DEVICE double from_upper_half(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const auto rl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto rh = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double((int)rh[1], (int)rl[1]);
}

/* broadcast lane `src` (wave-uniform index) of a per-lane double: two v_readlane_b32 */
WV_DEVICE double readlane(double v, int src) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, src);
    hi = __builtin_amdgcn_readlane(hi, src);
    return __hiloint2double(hi, lo);
}

/* write the wave-uniform value s into lane DST of v: two v_writelane_b32 (no builtin in this compiler) */
template <int DST> WV_DEVICE double writelane(double v, double s) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    const int slo = __builtin_amdgcn_readfirstlane(__double2loint(s)), shi = __builtin_amdgcn_readfirstlane(__double2hiint(s));
    asm("v_writelane_b32 %0, %1, %2" : "+v"(lo) : "s"(slo), "n"(DST));
    asm("v_writelane_b32 %0, %1, %2" : "+v"(hi) : "s"(shi), "n"(DST));
    return __hiloint2double(hi, lo);
}

/* single-precision sum over the 64 lanes (same tree as wave_sum; the DPP moves fold into the adds): for estimates */
template <int CTRL, int ROW_MASK> WV_DEVICE float dpp_take_f32(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
WV_DEVICE float wave_sum_f32(float v) {
    v += dpp_take_f32<0xB1, 0xf>(v);
    v += dpp_take_f32<0x4E, 0xf>(v);
    v += dpp_take_f32<0x141, 0xf>(v);
    v += dpp_take_f32<0x140, 0xf>(v);
    v += dpp_take_f32<0x142, 0xa>(v);
    v += dpp_take_f32<0x143, 0xc>(v);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

WV_DEVICE unsigned long long ballot(bool p) { return __ballot(p); }

/* v taken from another lane through a DPP control (quad_perm / row_mirror / row_bcast), 0.0 where the control or the
 * row mask selects nothing */
template <int CTRL, int ROW_MASK> WV_DEVICE double dpp_take(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}

/* sum over the 64 lanes, returned wave-uniform: butterfly inside each row of 16 (quad permutes and mirrors), then the
 * row totals chained through lane 15 -> next row and lane 31 -> upper half; no LDS crossbar traffic */
WV_DEVICE double wave_sum(double v) {
    v += dpp_take<0xB1, 0xf>(v);  /* quad_perm [1,0,3,2] */
    v += dpp_take<0x4E, 0xf>(v);  /* quad_perm [2,3,0,1] */
    v += dpp_take<0x141, 0xf>(v); /* row_half_mirror */
    v += dpp_take<0x140, 0xf>(v); /* row_mirror: every lane of a row now holds the row total */
    v += dpp_take<0x142, 0xa>(v); /* row_bcast15 into rows 1 and 3 */
    v += dpp_take<0x143, 0xc>(v); /* row_bcast31 into rows 2 and 3 */
    return readlane(v, 63);
}

/* individually rounded IEEE double operations the compiler may not contract into FMAs or reassociate: the encoder /
 * motor models must reproduce the reference's host arithmetic bit for bit (reference src/cassiemujoco.c:558-664) */
WV_DEVICE double mul_rn(double a, double b) { return __dmul_rn(a, b); }
WV_DEVICE double add_rn(double a, double b) { return __dadd_rn(a, b); }
WV_DEVICE double sub_rn(double a, double b) { return __dsub_rn(a, b); }
WV_DEVICE double div_rn(double a, double b) { return __ddiv_rn(a, b); }
WV_DEVICE double sqrt_rn(double a) { return __dsqrt_rn(a); }

/* hardware reciprocal estimate (v_rcp_f64) */
WV_DEVICE double rcp_estimate(double x) { return __builtin_amdgcn_rcp(x); }
/* hardware reciprocal-square-root estimate (v_rsq_f64) */
WV_DEVICE double rsq_estimate(double x) { return __builtin_amdgcn_rsq(x); }

/* value-preserving move the optimiser cannot see through (wave-uniform ints only) */
WV_DEVICE int opaque(int x) { x = __builtin_amdgcn_readfirstlane(x); asm volatile("" : "+s"(x)); return x; }

/* same for a wave-uniform pointer: loads through the result cannot be hoisted above this point */
template <class P> WV_DEVICE P opaque_ptr(P p) {
    const unsigned long long v = (unsigned long long)p;
    unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    asm volatile("" : "+s"(lo), "+s"(hi));
    return (P)(((unsigned long long)hi << 32) | lo);
}

/* the value stays what it is, but the compiler must have it in a register HERE and may not have derived anything from it
 * earlier: used where a constant is requested from memory well ahead of its use -- a compare or a select folded into the
 * load would pull the wait for the value forward to the request */
WV_DEVICE void keep(int &x) { asm volatile("" : "+v"(x)); }

/* the value must still be in a register here (no instruction) */
WV_DEVICE void touch(double x) { asm volatile("" : : "v"(x)); }

/* instruction-scheduling fence: nothing is moved across it.  Used between hand-staged load / compute groups so the
 * scheduler's appetite for early loads cannot push the register allocator into scratch. */
WV_DEVICE void sched_fence() { __builtin_amdgcn_sched_barrier(0); }

/* test hooks of the CPU emulator (tests/emu/wave.h gives them bodies); on the device they are constants the compiler folds
 * away: take the guarded PGS sweep every time / skip the once-per-launch initialisation of the centre-of-mass rows (the
 * bug the LDS-poison checks were written for) / fill the env's LDS block with NaN patterns at the start of a launch */
WV_DEVICE constexpr bool debug_force_guarded() { return false; }
WV_DEVICE constexpr bool test_skip_com_init() { return false; }
template <class S> WV_DEVICE void test_launch_hook(S *, unsigned long) {}

WV_DEVICE long long clock() { return (long long)__builtin_readcyclecounter(); }
WV_DEVICE long long wall_clock() { return (long long)__builtin_readsteadycounter(); } /* constant 100 MHz (s_memrealtime) */
/* where the hardware placed this wave (profiling aid): HW_ID (wave slot [3:0], SIMD [5:4], CU [11:8], SH [12], SE [15:13], workgroup slot
 * [19:16]) in the low word, XCC_ID in the high word */
WV_DEVICE long long hw_id() {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    return (long long)(((unsigned long long)xcc << 32) | hw);
}

WV_DEVICE int popc64(unsigned long long x) { return __popcll(x); }
/* max of two doubles that are known not to be signalling NaNs, as the single instruction */
WV_DEVICE double max_raw(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

}  // namespace wv
#endif
