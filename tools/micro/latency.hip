// Micro-benchmarks of dependent-issue latencies that bound the one-wave-per-SIMD step kernel (profiling aid).
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ double rl(double v, int l) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, l); hi = __builtin_amdgcn_readlane(hi, l);
    return __hiloint2double(hi, lo);
}
#define N 512
__global__ void __launch_bounds__(64) k(double *out, long long *t, double a, double b, const double *g, int stride) {
    __shared__ double lds[4096];
    double x = threadIdx.x * 1e-3 + a, y = b;
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = i * 1e-9;
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
#pragma unroll 16
    for (int i = 0; i < N; ++i) x = fma(x, a, b);                      // 0: dependent fma
    long long t1 = __builtin_readcyclecounter();
#pragma unroll 16
    for (int i = 0; i < N; ++i) { x = fma(x, a, b); y = fma(y, a, b); } // 1: two independent chains
    long long t2 = __builtin_readcyclecounter();
#pragma unroll 16
    for (int i = 0; i < N; ++i) { double d = fmax(x, b); x = fma(a, rl(d, i & 63), x); } // 2: max -> readlane -> fma
    long long t3 = __builtin_readcyclecounter();
#pragma unroll 16
    for (int i = 0; i < N; ++i) { x = fma(a, rl(x, i & 63), x); }       // 3: readlane -> fma
    long long t4 = __builtin_readcyclecounter();
    int idx = threadIdx.x;
#pragma unroll 16
    for (int i = 0; i < N; ++i) { double v = lds[idx & 4095]; idx = (int)(v * 1e9) + 1 + threadIdx.x; x += v; } // 4: dependent LDS read
    long long t5 = __builtin_readcyclecounter();
    int gi = threadIdx.x;
#pragma unroll 4
    for (int i = 0; i < 64; ++i) { double v = g[(size_t)gi * stride]; gi = (int)v + threadIdx.x; x += v; }     // 5: dependent global read (L2-resident)
    long long t6 = __builtin_readcyclecounter();
#pragma unroll 16
    for (int i = 0; i < N; ++i) { x = x * a; x = fmax(x, b); }          // 6: mul -> max
    long long t7 = __builtin_readcyclecounter();
    float xf = (float)x;
#pragma unroll 16
    for (int i = 0; i < N; ++i) xf = fmaf(xf, (float)a, (float)b);       // 7: dependent fp32 fma
    long long t8 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + threadIdx.x] = x + y + xf;
    if (threadIdx.x == 0 && blockIdx.x == 0) { t[0] = t1 - t0; t[1] = t2 - t1; t[2] = t3 - t2; t[3] = t4 - t3; t[4] = t5 - t4; t[5] = t6 - t5; t[6] = t7 - t6; t[7] = t8 - t7; }
}
int main() {
    double *out, *g; long long *t;
    hipMalloc(&out, 4096 * 64 * 8); hipMalloc(&t, 64); hipMalloc(&g, 1 << 20); hipMemset(g, 0, 1 << 20);
    for (int blocks : {1, 1024, 2048}) {
        hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, out, t, 0.999, 1e-3, g, 16);
        hipDeviceSynchronize();
        long long h[8]; hipMemcpy(h, t, 64, hipMemcpyDeviceToHost);
        printf("blocks %4d: fma %.1f | 2 chains %.1f/iter | max-readlane-fma %.1f | readlane-fma %.1f | lds dep %.1f | global dep %.1f | mul-max %.1f | fp32 fma %.1f  (cycles per iteration)\n",
               blocks, h[0] / (double)N, h[1] / (double)N, h[2] / (double)N, h[3] / (double)N, h[4] / (double)N, h[5] / 64.0, h[6] / (double)N, h[7] / (double)N);
    }
    return 0;
}
