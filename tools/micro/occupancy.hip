// What a second (third, fourth) wave per SIMD would buy on the step kernel's two characteristic inner loops, measured in
// isolation: the real kernel cannot run more than one wave per SIMD -- its 40 KB of LDS per env fill the
// CU with four envs -- so this is the upper bound that the LDS wall keeps out of reach.
//   loop A  PGS row chain: max -> 2 x v_readlane -> fma, coefficient from a 32-entry register array
//   loop B  LDS broadcast row + 32 FMAs (the A = Y Y^T / half-solve pattern)
// Occupancy is set by the dynamic LDS size of the launch (160 KB per CU, 4 SIMDs): 40 KB -> 1 wave per SIMD, 20 KB -> 2,
// 10 KB -> 4.  Every wave does the same work; the figure is wave-iterations per microsecond per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#define ROWS 24
#define SWEEPS 400
__device__ __forceinline__ double rl(double v, int l) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
template <int I> __device__ __forceinline__ void row(const double (&b)[32], double lo, double &s) {
    const double d = fmax(s, lo);
    s = fma(b[I], rl(d, I), s);
}
template <int I> __device__ __forceinline__ void rows(const double (&b)[32], double lo, double &s) {
    if constexpr (I < ROWS) { row<I>(b, lo, s); rows<I + 1>(b, lo, s); }
}
__global__ void __launch_bounds__(64) loopA(double *out, double a) {
    extern __shared__ double lds[];
    double b[32];
    for (int i = 0; i < 32; ++i) b[i] = -1e-3 * (1 + ((threadIdx.x + i) & 7)) * a;
    double s = 0.5 + 1e-3 * threadIdx.x;
    for (int it = 0; it < SWEEPS; ++it) rows<0>(b, -1.0, s);
    if (s == 12345.0) lds[threadIdx.x] = s;
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
// loop C: TWO envs in one wave (lanes 0-31 / 32-63), their PGS chains advanced together: one max, a readlane per env, a
// per-half select, one FMA -- two env-rows per step.
template <int I> __device__ __forceinline__ void row2(const double (&b)[32], double lo, double &s, bool upper) {
    const double d = fmax(s, lo);
    const double da = rl(d, I), db = rl(d, 32 + I);
    s = fma(b[I], upper ? db : da, s);
}
template <int I> __device__ __forceinline__ void rows2(const double (&b)[32], double lo, double &s, bool upper) {
    if constexpr (I < ROWS) { row2<I>(b, lo, s, upper); rows2<I + 1>(b, lo, s, upper); }
}
__global__ void __launch_bounds__(64) loopC(double *out, double a) {
    extern __shared__ double lds[];
    double b[32];
    for (int i = 0; i < 32; ++i) b[i] = -1e-3 * (1 + ((threadIdx.x + i) & 7)) * a;
    double s = 0.5 + 1e-3 * threadIdx.x;
    const bool upper = threadIdx.x >= 32;
    for (int it = 0; it < SWEEPS; ++it) rows2<0>(b, -1.0, s, upper);
    if (s == 12345.0) lds[threadIdx.x] = s;
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
__global__ void __launch_bounds__(64) loopB(double *out, double a) {
    extern __shared__ double lds[];
    for (int i = threadIdx.x; i < 32 * 34; i += 64) lds[i] = 1e-3 * (i & 15) * a;
    __syncthreads();
    double y[32], acc = 0;
    for (int i = 0; i < 32; ++i) y[i] = 1.0 + 1e-3 * ((threadIdx.x + i) & 3);
    for (int it = 0; it < SWEEPS; ++it) {
#pragma unroll 4
        for (int r = 0; r < ROWS; ++r) {
            double p = 0;
#pragma unroll
            for (int k = 0; k < 32; ++k) p = fma(lds[r * 34 + k], y[k], p);
            acc += p;
            y[0] = fma(acc, 1e-9, y[0]);                // keeps the rows from being hoisted out of the sweep loop
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}
template <class K> static void run(const char *name, K kern, double *out) {
    int ncu = 0;
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int occ : {1, 2, 4}) {
        const size_t lds = (size_t)40 * 1024 / occ;
        const int blocks = ncu * 4 * occ;            // exactly one resident generation of waves
        (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 40 * 1024);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), lds, 0, out, 1.0);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), lds, 0, out, 1.0);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double iters = (double)blocks * SWEEPS * ROWS;
        static double base = 0;
        const double rate = iters / (ms * 1e3) / ncu;
        if (occ == 1) base = rate;
        printf("%s  %d wave(s)/SIMD: %8.3f ms  %10.1f wave-row-iterations/us/CU  (x%.2f of one wave per SIMD)\n", name, occ, ms, rate, rate / base);
    }
}
// two envs per wave: the footprint per wave doubles (80 KB -> two waves per CU, two SIMDs idle) unless it is halved per env
static void run_pairs(double *out) {
    int ncu = 0;
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipFuncSetAttribute((const void *)loopC, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    for (int waves_per_cu : {2, 4}) {
        const size_t lds = (size_t)160 * 1024 / waves_per_cu;
        const int blocks = ncu * waves_per_cu;
        hipLaunchKernelGGL(loopC, dim3(blocks), dim3(64), lds, 0, out, 1.0);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(loopC, dim3(blocks), dim3(64), lds, 0, out, 1.0);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double envrows = (double)blocks * SWEEPS * ROWS * 2;
        printf("PGS row chain, TWO envs per wave, %d wave(s)/CU (%3zu KB LDS each): %8.3f ms  %10.1f env-row-iterations/us/CU\n",
               waves_per_cu, lds / 1024, ms, envrows / (ms * 1e3) / ncu);
    }
}
int main() {
    double *out;
    (void)hipMalloc(&out, (size_t)1 << 24);
    run("PGS row chain (max, readlane, fma)   ", loopA, out);
    run_pairs(out);
    run("LDS broadcast row x 32 FMAs          ", loopB, out);
    return 0;
}
