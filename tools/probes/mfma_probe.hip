// Probe: lane layouts of v_mfma_f64_4x4x4_4b_f64 (A, B, D operands) and the
// CBSZ/ABID block-broadcast controls on gfx950; plus 4x4x4 + VALU mixing.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int CBSZ, int ABID>
__global__ void probe(int* table) {
    const int lane = threadIdx.x;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            double a = (lane == la) ? 1.0 : 0.0;
            double b = (lane == lb) ? 1.0 : 0.0;
            double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, CBSZ, ABID, 0);
            unsigned long long m = __ballot(d != 0.0);
            if (lane == 0) { table[(la * 64 + lb) * 2] = (int)(m & 0xffffffffu); table[(la * 64 + lb) * 2 + 1] = (int)(m >> 32); }
        }
}

template <int NV, int RV>
__global__ __launch_bounds__(256) void mix(double* out, unsigned long long* cyc, int iters, double seed) {
    double acc[64];
    for (int i = 0; i < 64; ++i) acc[i] = seed;
    double a = seed + threadIdx.x * 1e-9, b = 1.0 - seed;
    double v[NV > 0 ? NV : 1];
    for (int i = 0; i < (NV > 0 ? NV : 1); ++i) v[i] = seed * i;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 64; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < RV; ++r)
#pragma unroll
            for (int i = 0; i < NV; ++i) v[i] = fma(v[i], a, b);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int i = 0; i < 64; ++i) s += acc[i];
    for (int i = 0; i < NV; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NV, int RV>
void runmix(const char* name, int bpc) {
    double* out; unsigned long long* cyc;
    int nblk = 256 * bpc, iters = 2000;
    (void)hipMalloc(&out, sizeof(double) * nblk * 256); (void)hipMalloc(&cyc, 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((mix<NV, RV>), dim3(nblk), dim3(256), 0, 0, out, cyc, 50, 0.5);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((mix<NV, RV>), dim3(nblk), dim3(256), 0, 0, out, cyc, iters, 0.5);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    double waves = nblk * 4.0;
    double mf = waves * iters * 64 * 512.0, vf = waves * iters * (double)RV * NV * 128.0;
    printf("%-30s blk/CU %d %8.3f ms mfma %6.1f TF valu %6.1f TF sum %6.1f | %7.1f ticks/iter\n", name, bpc, ms,
           mf / ms / 1e9, vf / ms / 1e9, (mf + vf) / ms / 1e9, (double)c / iters);
}

template <int CBSZ, int ABID>
void runprobe(const char* name) {
    int* t; (void)hipMalloc(&t, 64 * 64 * 2 * sizeof(int));
    hipLaunchKernelGGL((probe<CBSZ, ABID>), dim3(1), dim3(64), 0, 0, t);
    int* h = (int*)malloc(64 * 64 * 2 * sizeof(int));
    (void)hipMemcpy(h, t, 64 * 64 * 2 * sizeof(int), hipMemcpyDeviceToHost);
    printf("PROBE %s: for each A-lane la: list of (lb -> D lanes)\n", name);
    for (int la = 0; la < 64; ++la) {
        printf(" la=%2d:", la);
        for (int lb = 0; lb < 64; ++lb) {
            unsigned long long m = ((unsigned long long)(unsigned)h[(la * 64 + lb) * 2 + 1] << 32) | (unsigned)h[(la * 64 + lb) * 2];
            if (!m) continue;
            printf(" %d->", lb);
            for (int l = 0; l < 64; ++l) if (m >> l & 1) printf("%d,", l);
        }
        printf("\n");
    }
}

int main() {
    runprobe<0, 0>("cbsz0");
    runprobe<2, 0>("cbsz2 abid0");
    runprobe<2, 1>("cbsz2 abid1");
    runprobe<2, 3>("cbsz2 abid3");
    runprobe<1, 0>("cbsz1 abid0");
    runmix<0, 0>("4x4x4 x64", 1);
    runmix<16, 4>("4x4x4 x64 + valu x64", 1);
    runmix<16, 8>("4x4x4 x64 + valu x128", 1);
    runmix<16, 16>("4x4x4 x64 + valu x256", 1);
    runmix<16, 8>("4x4x4 x64 + valu x128", 2);
    runmix<16, 16>("4x4x4 x64 + valu x256", 2);
    return 0;
}
