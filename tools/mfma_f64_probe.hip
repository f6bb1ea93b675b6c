// Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_f64_probe.hip -o /tmp/mfma_f64_probe (output: profiles/round3/v27_mfma_f64_probe.txt)
// Probe of v_mfma_f64_16x16x4_f64 on gfx950: (1) is a result element the plain FMA chain over k = 0..3 on top of C
// (bit for bit)?  (2) issue interval of independent / dependent instructions in one wave per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void one(const double *a, const double *b, const double *c, double *d) {
    const int l = threadIdx.x;
    d4 cc = {c[l * 4], c[l * 4 + 1], c[l * 4 + 2], c[l * 4 + 3]};
    d4 r = __builtin_amdgcn_mfma_f64_16x16x4f64(a[l], b[l], cc, 0, 0, 0);
    r = __builtin_amdgcn_mfma_f64_16x16x4f64(a[64 + l], b[64 + l], r, 0, 0, 0);
    for (int v = 0; v < 4; ++v) d[l * 4 + v] = r[v];
}
template <int CHAINS>
__global__ void rate(double *out, int iters, long long *clk) {
    const int l = threadIdx.x;
    d4 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) acc[c] = {0.0, 0.0, 0.0, 0.0};
    double a = 1.0 + l * 1e-3, b = 1.0 - l * 1e-3;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
    }
    long long t1 = __builtin_readcyclecounter();
    double s = 0;
    for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * 64 + l] = s;
    if (l == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}
int main() {
    std::vector<double> a(128), b(128), c(256), d(256);
    srand(1);
    auto rnd = [] { return (rand() / (double)RAND_MAX - 0.5) * ldexp(1.0, rand() % 40 - 20); };
    for (auto &x : a) x = rnd(); for (auto &x : b) x = rnd(); for (auto &x : c) x = rnd();
    double *da, *db, *dc, *dd; long long *dk;
    hipMalloc(&da, 128 * 8); hipMalloc(&db, 128 * 8); hipMalloc(&dc, 256 * 8); hipMalloc(&dd, 1 << 20); hipMalloc(&dk, 8);
    hipMemcpy(da, a.data(), 128 * 8, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), 128 * 8, hipMemcpyHostToDevice); hipMemcpy(dc, c.data(), 256 * 8, hipMemcpyHostToDevice);
    one<<<1, 64>>>(da, db, dc, dd);
    hipMemcpy(d.data(), dd, 256 * 8, hipMemcpyDeviceToHost);
    int bad_fwd = 0, bad_rev = 0;
    for (int l = 0; l < 64; ++l) for (int v = 0; v < 4; ++v) {
        const int i = (l >> 4) + 4 * v, j = l & 15;
        double f = c[l * 4 + v], r = c[l * 4 + v];
        for (int blk = 0; blk < 2; ++blk) {
            for (int k = 0; k < 4; ++k) f = fma(a[blk * 64 + i + 16 * k], b[blk * 64 + j + 16 * k], f);
            for (int k = 3; k >= 0; --k) r = fma(a[blk * 64 + i + 16 * k], b[blk * 64 + j + 16 * k], r);
        }
        bad_fwd += memcmp(&f, &d[l * 4 + v], 8) != 0; bad_rev += memcmp(&r, &d[l * 4 + v], 8) != 0;
    }
    printf("elements that differ from the k = 0..3 fma chain: %d of 256; from the k = 3..0 chain: %d\n", bad_fwd, bad_rev);
    long long k;
    const int it = 2000;
    rate<1><<<1, 64>>>(dd, it, dk); hipMemcpy(&k, dk, 8, hipMemcpyDeviceToHost); printf("dependent chain: %.1f clocks per mfma\n", (double)k / it);
    rate<3><<<1, 64>>>(dd, it, dk); hipMemcpy(&k, dk, 8, hipMemcpyDeviceToHost); printf("3 interleaved chains: %.1f clocks per mfma\n", (double)k / it / 3);
    rate<6><<<1, 64>>>(dd, it, dk); hipMemcpy(&k, dk, 8, hipMemcpyDeviceToHost); printf("6 interleaved chains: %.1f clocks per mfma\n", (double)k / it / 6);
    return 0;
}
