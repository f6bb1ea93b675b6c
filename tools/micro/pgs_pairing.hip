// Micro-benchmark for the question "two envs per wavefront in the lane-sparse stages?" (round 5's verdict, item 4), on the stage that
// is both the largest and the most lane-sparse: the PGS sweep of the fast kernel (<= 31 rows in lanes 0 .. 30 of a 64-lane wave).
//   form A  today's: one env per wave; a row's step is v_max_f64, two v_readlane_b32, v_fma_f64 (+ the two v_cndmask_b32 that keep the
//           row's starting residual)                                                  -- 6 vector instructions per env-row
//   form B  two envs per wave, env 0 in lanes 0 .. 31 and env 1 in lanes 32 .. 63: a row's step needs the broadcast of lane I WITHIN
//           each half.  gfx950 has no such DPP control (row_newbcast spans 16 lanes); what it has is ds_swizzle_b32 in bit-mask mode
//           (and_mask 0, or_mask I: every lane of a group of 32 reads lane I of its group) -- two of them per 64-bit value, through
//           the LDS crossbar                                                          -- 6 vector/LDS instructions per TWO env-rows
//   form C  two envs per wave with the broadcast done by lane reads: four v_readlane_b32 and the merge of the two scalars into one
//           vector (v_mov_b32 x2 + v_cndmask_b32 x2 against a half mask)              -- 11 per two env-rows
// Every form runs SWEEPS sweeps over ROWS rows from synthetic data; the figure is shader clocks per ENV-row, for one wave per SIMD
// (grid = 1 workgroup per SIMD... a single wave here) and for the occupancy the step kernel runs at (two waves per SIMD, all SIMDs busy).
// Build + run on a GPU box:  hipcc -O3 --offload-arch=gfx950 tools/micro/pgs_pairing.hip -o /tmp/pgs_pairing && /tmp/pgs_pairing
#include <hip/hip_runtime.h>
#include <cstdio>
#define ROWS 24
#define SWEEPS 64

__device__ __forceinline__ double rl(double v, int l) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, l); hi = __builtin_amdgcn_readlane(hi, l);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double vmax(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
template <int I> __device__ __forceinline__ double half_bcast(double v) {
    // ds_swizzle bit-mask mode: offset = and_mask | or_mask << 5 | xor_mask << 10 (bit 15 clear); and_mask 0 -> lane id within the group of 32 = or_mask
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_ds_swizzle(lo, (I << 5));
    hi = __builtin_amdgcn_ds_swizzle(hi, (I << 5));
    return __hiloint2double(hi, lo);
}

template <int I> __device__ __forceinline__ void rowA(const double (&b)[ROWS], int lane, double lo, double &s, double &mys) {
    const double d = vmax(s, lo);
    if (lane == I) mys = s;
    s = fma(b[I], rl(d, I), s);
}
template <int I> __device__ __forceinline__ void rowB(const double (&b)[ROWS], int lane32, double lo, double &s, double &mys) {
    const double d = vmax(s, lo);
    if (lane32 == I) mys = s;
    s = fma(b[I], half_bcast<I>(d), s);
}
template <int I> __device__ __forceinline__ void rowC(const double (&b)[ROWS], int lane32, bool upper, double lo, double &s, double &mys) {
    const double d = vmax(s, lo);
    if (lane32 == I) mys = s;
    const double d0 = rl(d, I), d1 = rl(d, 32 + I);
    s = fma(b[I], upper ? d1 : d0, s);
}
template <int I, int FORM> __device__ __forceinline__ void rows(const double (&b)[ROWS], int lane, double lo, double &s, double &mys) {
    if constexpr (I < ROWS) {
        if constexpr (FORM == 0) rowA<I>(b, lane, lo, s, mys);
        else if constexpr (FORM == 1) rowB<I>(b, lane & 31, lo, s, mys);
        else rowC<I>(b, lane & 31, lane >= 32, lo, s, mys);
        rows<I + 1, FORM>(b, lane, lo, s, mys);
    }
}

template <int FORM> __global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) k(double *out, long long *t, const double *in) {
    const int lane = threadIdx.x;
    double b[ROWS];
#pragma unroll
    for (int i = 0; i < ROWS; ++i) b[i] = in[(lane & 31) * ROWS + i] * ((lane & 31) == i ? 0.0 : -0.02);
    double s = in[lane] - 0.5, f = 0, mys = 0;
    const long long t0 = __builtin_readcyclecounter();
    for (int sw = 0; sw < SWEEPS; ++sw) {
        const double lo = 0.0 - f;
        rows<0, FORM>(b, lane, lo, s, mys);
        const double d = vmax(mys, lo);
        f += d; s = fma(-0.1, d, s);
        asm volatile("" : "+v"(s), "+v"(f));
    }
    const long long t1 = __builtin_readcyclecounter();
    out[(size_t)blockIdx.x * 64 + lane] = s + f;
    if (lane == 0) t[blockIdx.x] = t1 - t0;
}

int main() {
    double *out, *in; long long *t;
    const int maxb = 4096;
    hipMalloc(&out, (size_t)maxb * 64 * 8); hipMalloc(&t, maxb * 8); hipMalloc(&in, 64 * ROWS * 8);
    double h[64 * ROWS];
    for (int i = 0; i < 64 * ROWS; ++i) h[i] = (double)((i * 2654435761u) % 1000) / 1000.0;
    hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
    long long ht[maxb];
    const char *names[3] = {"A one env per wave (readlane)", "B two envs per wave (ds_swizzle)", "C two envs per wave (4 readlanes + merge)"};
    for (int blocks : {1, 2048}) {          // 2048 waves = 256 CUs x 4 SIMDs x 2 waves: the step kernel's occupancy
        for (int form = 0; form < 3; ++form) {
            for (int rep = 0; rep < 2; ++rep) {
                if (form == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(64), 0, 0, out, t, in);
                if (form == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(64), 0, 0, out, t, in);
                if (form == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(64), 0, 0, out, t, in);
                hipDeviceSynchronize();
            }
            hipMemcpy(ht, t, blocks * 8, hipMemcpyDeviceToHost);
            double mean = 0; for (int i = 0; i < blocks; ++i) mean += ht[i]; mean /= blocks;
            const int envs = form == 0 ? 1 : 2;
            printf("%5d waves  %-42s %8.1f clocks per wave-row = %6.1f per env-row\n", blocks, names[form], mean / (SWEEPS * ROWS), mean / (SWEEPS * ROWS) / envs);
        }
    }
    return 0;
}
