// v_mfma_f64_16x16x4_f64 on gfx950: register layout, rounding order and issue rate (used by the A = Y Y^T stage of the
// step kernel).  Assumed layout, checked here against a host product:
//   A operand: lane l holds A[l % 16][l / 16]      B operand: lane l holds B[l / 16][l % 16]
//   C / D    : lane l, element v holds D[4 * (l / 16) + v][l % 16]   or   D[4 * v + l / 16][l % 16] -- both are tried
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(64) one(const double *A, const double *B, const double *C, double *D, int alt) {
    const int l = threadIdx.x;
    v4d c;
    for (int v = 0; v < 4; ++v) c[v] = C[(alt ? 4 * v + l / 16 : 4 * (l / 16) + v) * 16 + l % 16];
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(A[(l % 16) * 4 + l / 16], B[(l / 16) * 16 + l % 16], c, 0, 0, 0);
    for (int v = 0; v < 4; ++v) D[(alt ? 4 * v + l / 16 : 4 * (l / 16) + v) * 16 + l % 16] = c[v];
}
template <int TILES> __global__ void __launch_bounds__(64) rate(double *out, double a, int reps) {
    v4d c[TILES];
    for (int t = 0; t < TILES; ++t) for (int v = 0; v < 4; ++v) c[t][v] = 0.0;
    double x = a + threadIdx.x * 1e-3, y = a - threadIdx.x * 1e-3;
    for (int r = 0; r < reps; ++r)
#pragma unroll
        for (int t = 0; t < TILES; ++t) c[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c[t], 0, 0, 0);
    double s = 0;
    for (int t = 0; t < TILES; ++t) for (int v = 0; v < 4; ++v) s += c[t][v];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
int main() {
    double hA[64], hB[64], hC[256], hD[256], *dA, *dB, *dC, *dD;
    srand(1);
    for (double &x : hA) x = rand() / (double)RAND_MAX - 0.5;
    for (double &x : hB) x = rand() / (double)RAND_MAX - 0.5;
    for (double &x : hC) x = rand() / (double)RAND_MAX - 0.5;
    (void)hipMalloc(&dA, sizeof hA); (void)hipMalloc(&dB, sizeof hB); (void)hipMalloc(&dC, sizeof hC); (void)hipMalloc(&dD, 1 << 22);
    (void)hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    (void)hipMemcpy(dC, hC, sizeof hC, hipMemcpyHostToDevice);
    for (int alt = 0; alt < 2; ++alt) {
        hipLaunchKernelGGL(one, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD, alt);
        (void)hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
        double worst = 0; int exact_fwd = 0, exact_rev = 0;
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
            double f = hC[i * 16 + j], r = hC[i * 16 + j];
            for (int k = 0; k < 4; ++k) f = fma(hA[i * 4 + k], hB[k * 16 + j], f);
            for (int k = 3; k >= 0; --k) r = fma(hA[i * 4 + k], hB[k * 16 + j], r);
            worst = fmax(worst, fabs(hD[i * 16 + j] - f));
            exact_fwd += hD[i * 16 + j] == f; exact_rev += hD[i * 16 + j] == r;
        }
        printf("D rows %s: max |D - (C + A B)| = %.3g; bit-equal to fma chain k = 0..3: %d / 256, k = 3..0: %d / 256\n",
               alt ? "4 v + l / 16" : "4 (l / 16) + v", worst, exact_fwd, exact_rev);
    }
    int ncu = 0; (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int reps = 4000;
    auto time = [&](auto kern, int tiles, const char *name) {
        hipLaunchKernelGGL(kern, dim3(ncu * 4), dim3(64), 0, 0, dD, 0.5, reps); (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(ncu * 4), dim3(64), 0, 0, dD, 0.5, reps); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%s: %.1f ns per MFMA per SIMD (one wave per SIMD)\n", name, ms * 1e6 / ((double)reps * tiles));
    };
    time(rate<1>, 1, "1 accumulator (dependent chain)     ");
    time(rate<4>, 4, "4 accumulators (independent, interleaved)");
    return 0;
}
