// Micro-benchmark: f64 matrix-core and f64 VALU issue rates on gfx950 (MI355X).
// hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_peak.hip -o tools/mfma_peak && ./tools/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double f64x4 __attribute__((ext_vector_type(4)));

// NACC 16x16x4 MFMAs + NV independent VALU FMA chains (x RV repeats) per iteration.
template <int NACC, int NV, int RV, int SMALL>
__global__ __launch_bounds__(256) void peak(double* out, unsigned long long* cyc, int iters, double seed) {
    f64x4 acc[NACC > 0 ? NACC : 1];
    double acc1[NACC > 0 ? NACC : 1];
    for (int i = 0; i < (NACC > 0 ? NACC : 1); ++i) { acc[i] = (f64x4){seed, 0, 0, 0}; acc1[i] = seed; }
    double a = seed + threadIdx.x * 1e-9, b = 1.0 - seed;
    double v[NV > 0 ? NV : 1];
    for (int i = 0; i < (NV > 0 ? NV : 1); ++i) v[i] = seed * i;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            if (SMALL) acc1[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc1[i], 0, 0, 0);
            else acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < RV; ++r)
#pragma unroll
            for (int i = 0; i < NV; ++i) v[i] = fma(v[i], a, b);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + acc1[i];
    for (int i = 0; i < NV; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NACC, int NV, int RV, int SMALL>
void run(const char* name, int blocks_per_cu, int iters = 4000) {
    double* out;
    unsigned long long* cyc;
    int nblk = 256 * blocks_per_cu;
    (void)hipMalloc(&out, sizeof(double) * nblk * 256);
    (void)hipMalloc(&cyc, 8);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((peak<NACC, NV, RV, SMALL>), dim3(nblk), dim3(256), 0, 0, out, cyc, 100, 0.5);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((peak<NACC, NV, RV, SMALL>), dim3(nblk), dim3(256), 0, 0, out, cyc, iters, 0.5);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c;
    (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    double waves = (double)nblk * 4;
    double mf = waves * iters * NACC * (SMALL ? 512.0 : 2048.0);
    double vf = waves * iters * (double)RV * NV * 64 * 2.0;
    printf("%-34s blk/CU %d %8.3f ms  mfma %6.1f TF  valu %6.1f TF  sum %6.1f | %7.1f memtime ticks/iter (100MHz? => %.2f us)\n", name,
           blocks_per_cu, ms, mf / ms / 1e9, vf / ms / 1e9, (mf + vf) / ms / 1e9, (double)c / iters, (double)c / 100.0);
    (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
    run<4, 0, 0, 0>("mfma16x16x4 x4", 1);
    run<4, 0, 0, 0>("mfma16x16x4 x4", 2);
    run<8, 0, 0, 0>("mfma16x16x4 x8", 1);
    run<16, 0, 0, 0>("mfma16x16x4 x16", 1);
    run<16, 0, 0, 0>("mfma16x16x4 x16", 2);
    run<16, 0, 0, 1>("mfma4x4x4_4b x16", 1);
    run<16, 0, 0, 1>("mfma4x4x4_4b x16", 2);
    run<4, 0, 0, 1>("mfma4x4x4_4b x4 (dep dist 4)", 1);
    run<8, 0, 0, 1>("mfma4x4x4_4b x8 (dep dist 8)", 1);
    run<2, 0, 0, 1>("mfma4x4x4_4b x2 (dep dist 2)", 1);
    run<1, 0, 0, 1>("mfma4x4x4_4b x1 (dep dist 1)", 1);
    run<0, 16, 8, 0>("valu fma x128", 1);
    run<0, 16, 8, 0>("valu fma x128", 2);
    run<0, 16, 8, 0>("valu fma x128", 4);
    run<16, 16, 2, 0>("mfma x16 + valu x32", 1);
    run<16, 16, 4, 0>("mfma x16 + valu x64", 1);
    run<16, 16, 8, 0>("mfma x16 + valu x128", 1);
    run<16, 16, 16, 0>("mfma x16 + valu x256", 1);
    run<16, 16, 24, 0>("mfma x16 + valu x384", 1);
    run<16, 16, 8, 0>("mfma x16 + valu x128", 2);
    run<16, 16, 16, 0>("mfma x16 + valu x256", 2);
    run<16, 16, 24, 0>("mfma x16 + valu x384", 2);
    run<16, 0, 0, 1>("mfma4x4x4_4b x16 SUSTAINED", 1, 1000000);
    return 0;
}
