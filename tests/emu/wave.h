/*
 * tests/emu/wave.h -- TEST-ONLY stand-in for cassie-mujoco-sim_amd/csrc/wave.h.
 *
 * Lets the CPU-only test suite execute the *same* physics_kernel.h source the
 * GPU runs, lane by lane, so kernel logic errors are caught before a GPU box is
 * used.  64 lanes run as 64 coroutines; every cross-lane primitive is a
 * rendezvous of all 64.  Nothing here is linked into the product library (the
 * product builds with -Icsrc only and therefore sees the real wave.h).
 */
#ifndef CASSIE_WAVE_H
#define CASSIE_WAVE_H

#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>

#define CK_EMULATED 1
#define WV_DEVICE inline
#define WV_GLOBAL static /* internal linkage: the product library exports host stubs of the same kernel names */
#define WV_SHARED static
#define WV_WAVE 64
#define WV_CONST_AS
#define __launch_bounds__(x)
#define WV_WAVES_PER_SIMD(n)

#include <cstdio>
namespace wv {
int lane();
int wave_id();
int env_id();
int grid_size();
inline int atomic_add(int *p, int v) { const int old = *p; *p = old + v; return old; } /* (lanes run one at a time) */
inline int atomic_or(int *p, int v) { const int old = *p; *p = old | v; return old; }
void sync();
void block_barrier();
inline void drain_vmem() {}
void spin_yield();
inline void publish(int *flag, int value) { sync(); if (lane() == 0) *flag = value; sync(); }
int shfl_i(int v, int src_lane);
inline void wait_for(const int *flag, int value) { while (shfl_i(*flag, 0) != value) spin_yield(); } /* (lane 0's reading decides for the wave) */
inline void wait_for_spin(const int *flag, int value) { wait_for(flag, value); }
double shfl(double v, int src_lane);
double shfl_xor(double v, int mask);
int shfl_i(int v, int src_lane);
struct mfma_acc { double c[4]; };
void mfma_f64_16x16x4(double a, double b, double (&c)[4]);
inline void mfma_f64_16x16x4_x3(double a0, double b0, mfma_acc &c0, double a1, double b1, mfma_acc &c1, double a2, double b2, mfma_acc &c2) {
    mfma_f64_16x16x4(a0, b0, c0.c); mfma_f64_16x16x4(a1, b1, c1.c); mfma_f64_16x16x4(a2, b2, c2.c);
}
inline void mfma_f64_drain(mfma_acc &, mfma_acc &, mfma_acc &) {}
inline void mfma_f64_16x16x4_x4(double a0, double b0, mfma_acc &c0, double a1, double b1, mfma_acc &c1, double a2, double b2, mfma_acc &c2,
                                double a3, double b3, mfma_acc &c3) {
    mfma_f64_16x16x4(a0, b0, c0.c); mfma_f64_16x16x4(a1, b1, c1.c); mfma_f64_16x16x4(a2, b2, c2.c); mfma_f64_16x16x4(a3, b3, c3.c);
}
inline void mfma_f64_drain4(mfma_acc &, mfma_acc &, mfma_acc &, mfma_acc &) {}
inline double from_upper_half(double v) { const double o = shfl_xor(v, 32); return lane() < 32 ? o : v; }
double readlane(double v, int src);
int lane();
template <int DST> inline double writelane(double v, double s) { return lane() == DST ? s : v; } /* s is the same in every lane */
unsigned long long ballot(bool p);
double wave_sum(double v);
float wave_sum_f32(float v);
/* workgroups run one after the other here, in launch order: the word a chunk waits for must be there already */
/* (the word is 64 tag + 8 XCC + done, wave.h of the product; the emulator has one "XCD", number emu_xcc(): a test hook that lets
 * a test place a producer elsewhere and see the consumer's check fire) */
int emu_xcc();
template <int NW> inline void publish_global(int *flag, int tag, int done) { if (NW > 1) block_barrier(); if (lane() == 0 && (NW == 1 || wave_id() == 0)) *flag = 64 * tag + 8 * emu_xcc() + done; }
void emu_fail(const char *what);
template <int NW> inline bool wait_global(const int *flag, int tag, int done) {
    const int word = *flag;
    if ((word >> 6) != tag || (word & 7) != done) { fprintf(stderr, "emu: chunk flag %d, expected tag %d done %d\n", word, tag, done); emu_fail("a chunk of a launch started before the chunk in front of it had finished"); }
    if (NW > 1) block_barrier(); else sync(); /* (as on the device: nobody moves on -- and publishes -- before every lane of every wave has looked) */
    return ((word >> 3) & 7) == 0; /* (consumers run on "XCD" 0) */
}
inline long long clock() { return 0; }
inline long long wall_clock() { return 0; }
inline long long hw_id() { return 0; }
inline double max_raw(double a, double b) { return a > b ? a : b; }
inline int opaque(int x) { return x; }
template <int P> inline void set_priority() {}
#define CK_PRIO_W0 0
#define CK_PRIO_W0_PGS 0
#define CK_PRIO_W0_KIN 0
#define CK_PRIO_W1_FJ 0
#define CK_PRIO_W1_FJ_HFIELD 0
#define CK_PRIO_W1_JP 0
#define CK_PRIO_W1_PE 0
#define CK_PRIO_W1_EF 0
inline void sched_fence() {}
inline void keep(int &) {}
inline void touch(double) {}
extern int g_force_guarded;
extern int g_poison_lds;
extern unsigned long g_poison_lo, g_poison_hi;
extern int g_skip_com_init;
inline bool debug_force_guarded() { return g_force_guarded != 0; }
inline bool test_skip_com_init() { return g_skip_com_init != 0; }
/* LDS is not initialised on the device: fill the env's block with NaN patterns so that a read-before-write shows */
inline void test_launch_hook(void *shared, unsigned long size) {
    if (g_poison_lds) {
        if (lane() == 0 && wave_id() == 0) {
            const unsigned long lo = g_poison_lo < size ? g_poison_lo : size, hi = g_poison_hi < size ? g_poison_hi : size;
            if (hi > lo) memset((char *)shared + lo, 0xff, hi - lo);
        }
        block_barrier();
    }
}
inline int fresh_lane() { return lane(); }
template <class P> inline P opaque_ptr(P p) { return p; }
inline double rcp_estimate(double x) { return (double)(1.0f / (float)x); } /* deliberately low precision, like the hardware estimate */
inline double rsq_estimate(double x) { return (double)(1.0f / sqrtf((float)x)); }
/* individually rounded double operations (the emulator is built without FMA contraction: baseline x86-64) */
inline double mul_rn(double a, double b) { volatile double r = a * b; return r; }
inline double add_rn(double a, double b) { volatile double r = a + b; return r; }
inline double sub_rn(double a, double b) { volatile double r = a - b; return r; }
inline double div_rn(double a, double b) { volatile double r = a / b; return r; }
inline double sqrt_rn(double a) { volatile double r = sqrt(a); return r; }
inline int popc64(unsigned long long x) { return __builtin_popcountll(x); }
}  // namespace wv
#endif
