// PGS row patterns in isolation (profiling aid).
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 512
__device__ __forceinline__ double qb(double v) {
    int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x55, 0xf, 0xf, true), hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x55, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double rl(double v, int l) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
__global__ void __launch_bounds__(64) k(double *out, long long *t, double a, double b, int sel) {
    double x = threadIdx.x * 1e-3 + a, y = b + threadIdx.x, u = 0, cap = 0;
    const bool m1 = (threadIdx.x & 7) == sel, m2 = (threadIdx.x >> 2) == sel;   // masks computed once, ahead of the loops
    long long t0 = __builtin_readcyclecounter();
#pragma unroll 16
    for (int i = 0; i < N; ++i) { double d = fmax(x, b); cap = m1 ? x : cap; x = fma(a, rl(d, 5), x); }  // 0: current kernel row: max, 2 cndmask, 2 readlane, fmac
    long long t1 = __builtin_readcyclecounter();
#pragma unroll 16
    for (int i = 0; i < N; ++i) { double d = fmax(x, b); double s = rl(d, 5); cap = m2 ? d : cap; x = fma(a, qb(d), x); u = fma(y, s, u); } // 1: quad row incl. slow path fma
    long long t2 = __builtin_readcyclecounter();
#pragma unroll 16
    for (int i = 0; i < N; ++i) { double d = fmax(x, b); x = fma(a, qb(d), x); cap = m2 ? d : cap; }       // 2: quad row without readlane
    long long t3 = __builtin_readcyclecounter();
#pragma unroll 16
    for (int i = 0; i < N; ++i) { double d = fmax(x, b); double s = rl(d, 5); x = fma(a, qb(d), x); u = fma(y, s, u); } // 3: quad row without capture
    long long t4 = __builtin_readcyclecounter();
#pragma unroll 16
    for (int i = 0; i < N; ++i) { cap = m1 ? x : cap; x = fma(x, a, b); }                                   // 4: fma + precomputed-mask capture
    long long t5 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + threadIdx.x] = x + y + u + cap;
    if (threadIdx.x == 0 && blockIdx.x == 0) { t[0] = t1 - t0; t[1] = t2 - t1; t[2] = t3 - t2; t[3] = t4 - t3; t[4] = t5 - t4; }
}
int main() {
    double *out; long long *t;
    (void)hipMalloc(&out, 4096 * 64 * 8); (void)hipMalloc(&t, 64);
    hipLaunchKernelGGL(k, dim3(1024), dim3(64), 0, 0, out, t, 0.999, 1e-3, 3);
    (void)hipDeviceSynchronize();
    long long h[8]; (void)hipMemcpy(h, t, 64, hipMemcpyDeviceToHost);
    printf("row now %.1f | quad row %.1f | quad row no readlane %.1f | quad row no capture %.1f | fma+premask capture %.1f\n",
           h[0] / (double)N, h[1] / (double)N, h[2] / (double)N, h[3] / (double)N, h[4] / (double)N);
    return 0;
}
