// More dependent-issue measurements: readlane throughput, DPP quad broadcast chains, cndmask, exec masking (profiling aid).
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 512
__device__ __forceinline__ double qb(double v) {
    int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x55, 0xf, 0xf, true), hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x55, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__global__ void __launch_bounds__(64) k(double *out, long long *t, double a, double b) {
    double x = threadIdx.x * 1e-3 + a, y = b + threadIdx.x;
    int acc = 0;
    long long t0 = __builtin_readcyclecounter();
#pragma unroll 16
    for (int i = 0; i < N; ++i) { acc += __builtin_amdgcn_readlane(__double2loint(y), i & 63); }      // 0: independent readlanes, SALU accumulate
    long long t1 = __builtin_readcyclecounter();
#pragma unroll 16
    for (int i = 0; i < N; ++i) { x = fma(x, a, b); acc += __builtin_amdgcn_readlane(__double2loint(y), i & 63); }  // 1: fma chain + independent readlane
    long long t2 = __builtin_readcyclecounter();
#pragma unroll 16
    for (int i = 0; i < N; ++i) { double d = fmax(x, b); x = fma(a, qb(d), x); }                        // 2: max -> dpp x2 -> fma
    long long t3 = __builtin_readcyclecounter();
#pragma unroll 16
    for (int i = 0; i < N; ++i) { x = fma(x, a, b); y = (threadIdx.x == (i & 63)) ? x : y; }            // 3: fma + per-lane capture (cmp + 2 cndmask)
    long long t4 = __builtin_readcyclecounter();
#pragma unroll 16
    for (int i = 0; i < N; ++i) { x = fma(x, a, b); if ((threadIdx.x >> 2) == (i & 15)) y = fma(y, a, x); }  // 4: fma + exec-masked fma
    long long t5 = __builtin_readcyclecounter();
#pragma unroll 16
    for (int i = 0; i < N; ++i) { x = fma(x, a, b); y = fma(y, a, b); x = fmax(x, b); y = fmax(y, a); }  // 5: 4 VALU, two chains
    long long t6 = __builtin_readcyclecounter();
#pragma unroll 16
    for (int i = 0; i < N; ++i) { double d = fmax(x, b); int lo = __builtin_amdgcn_readlane(__double2loint(d), 5), hi = __builtin_amdgcn_readlane(__double2hiint(d), 5); x = fma(a, __hiloint2double(hi, lo), x); } // 6: max->readlane(const lane)->fma
    long long t7 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + threadIdx.x] = x + y + acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) { t[0] = t1 - t0; t[1] = t2 - t1; t[2] = t3 - t2; t[3] = t4 - t3; t[4] = t5 - t4; t[5] = t6 - t5; t[6] = t7 - t6; }
}
int main() {
    double *out; long long *t;
    (void)hipMalloc(&out, 4096 * 64 * 8); (void)hipMalloc(&t, 64);
    for (int blocks : {1, 1024}) {
        hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, out, t, 0.999, 1e-3);
        (void)hipDeviceSynchronize();
        long long h[8]; (void)hipMemcpy(h, t, 64, hipMemcpyDeviceToHost);
        printf("blocks %4d: readlane alone %.1f | fma+indep readlane %.1f | max-dpp2-fma %.1f | fma+capture %.1f | fma+masked fma %.1f | 4 valu 2 chains %.1f | max-readlane(const)-fma %.1f\n",
               blocks, h[0] / (double)N, h[1] / (double)N, h[2] / (double)N, h[3] / (double)N, h[4] / (double)N, h[5] / (double)N, h[6] / (double)N);
    }
    return 0;
}
