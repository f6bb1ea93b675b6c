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

namespace wv {

WV_DEVICE int lane() { return (int)threadIdx.x; }
WV_DEVICE int env_id() { return (int)blockIdx.x; }

/* workgroup == one wave: this is an LDS fence + s_barrier that the backend
 * reduces to a wave barrier; it orders LDS traffic between lanes */
WV_DEVICE void sync() { __syncthreads(); }

WV_DEVICE double shfl(double v, int src_lane) { return __shfl(v, src_lane, WV_WAVE); }
WV_DEVICE double shfl_xor(double v, int mask) { return __shfl_xor(v, mask, WV_WAVE); }
WV_DEVICE int shfl_i(int v, int src_lane) { return __shfl(v, src_lane, WV_WAVE); }

/* broadcast lane `src` (wave-uniform index) of a per-lane double: two v_readlane_b32 */
WV_DEVICE double readlane(double v, int src) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, src);
    hi = __builtin_amdgcn_readlane(hi, src);
    return __hiloint2double(hi, lo);
}

WV_DEVICE unsigned long long ballot(bool p) { return __ballot(p); }

WV_DEVICE double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += shfl_xor(v, o);
    return v;
}

/* value-preserving move the optimiser cannot see through (wave-uniform ints only) */
WV_DEVICE int opaque(int x) { asm volatile("" : "+s"(x)); return x; }

WV_DEVICE long long clock() { return (long long)__builtin_readcyclecounter(); }

WV_DEVICE int popc64(unsigned long long x) { return __popcll(x); }

}  // namespace wv
#endif
