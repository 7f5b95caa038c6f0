// Micro-benchmark: TWO wavefronts per SIMD on gfx950.  Even waves run a pure
// v_mfma_f64_4x4x4_4b_f64 stream (64 independent accumulators would not fit two waves with
// the real kernel's 128, hence 32 here); odd waves run the traffic the sweep kernel
// interleaves into its MFMA stream today (ds_read_b128, ds_write_b128, buffer-style global
// loads, fp64 VALU).  Question: does the companion's traffic slow the MFMA wave, i.e. could a
// two-wave-per-SIMD sweep hide the ~15 % issue overhead of the one-wave design?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double f64x2 __attribute__((ext_vector_type(2)));

// MODE (companion): 0 idle, 1 ds_read_b128, 2 ds_write_b128, 3 global_load_dwordx4, 4 v_fma_f64,
// 5 the sweep's per-pair mix (4 ds_read + 1 load + 1 ds_write per 32 MFMA-times), 6 = second MFMA wave
template <int MODE, int PRIO>
__global__ __launch_bounds__(512) void k(double* out, const double* gsrc, unsigned long long* cyc, int iters, double seed) {
    __shared__ __attribute__((aligned(16))) double lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 512) lds[i] = seed * i;
    __syncthreads();
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool mf = w < 4 || MODE == 6;            // waves w and w + 4 share SIMD w & 3
    double s = 0.0;
    if (mf) {
        double acc[32];
        for (int i = 0; i < 32; ++i) acc[i] = seed;
        double a = seed + lane * 1e-9, b = 1.0 - seed;
        unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 32; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
        }
        unsigned long long t1 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < 32; ++i) s += acc[i];
        if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    } else {
        if (PRIO) __builtin_amdgcn_s_setprio(3);
        f64x2 l[8];
        for (int i = 0; i < 8; ++i) l[i] = (f64x2){seed, seed};
        double v[8];
        for (int i = 0; i < 8; ++i) v[i] = seed * i;
        f64x2* lp = (f64x2*)lds + lane + (w & 3) * 512;
        const f64x2* gp = (const f64x2*)gsrc + threadIdx.x + (size_t)blockIdx.x * 4096;
        // roughly as long as the MFMA waves run: iters * 64 MFMA * 16 cycles
        const int reps = iters / 8;                // short enough to finish while the MFMA waves still run
        int iv = lane;
        const unsigned long long c0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < reps; ++it) {
            if (MODE == 1) {
#pragma unroll
                for (int i = 0; i < 8; ++i) l[i] = lp[i * 64];
                for (int i = 0; i < 8; ++i) iv += __double2loint(l[i][0]);
            } else if (MODE == 2) {
#pragma unroll
                for (int i = 0; i < 4; ++i) { l[i][0] += 1.0; lp[i * 64] = l[i]; }
            } else if (MODE == 3) {
#pragma unroll
                for (int i = 0; i < 4; ++i) l[i] = gp[i * 512 + (it & 7) * 64];
                for (int i = 0; i < 4; ++i) iv += __double2loint(l[i][0]);
            } else if (MODE == 4) {
#pragma unroll
                for (int i = 0; i < 64; ++i) v[i & 7] = fma(v[i & 7], seed, 1e-3);
            } else if (MODE == 7) {
#pragma unroll
                for (int i = 0; i < 64; ++i) iv = iv * 3 + 1;          // 32-bit VALU stream
            } else if (MODE == 5) {
                // two pairs' worth per 64 MFMAs: 8 ds_read_b128, 2 global loads, 2 ds_write_b128
#pragma unroll
                for (int i = 0; i < 8; ++i) l[i] = lp[i * 64];
                f64x2 g0 = gp[(it & 7) * 64], g1 = gp[512 + (it & 7) * 64];
                for (int i = 0; i < 8; ++i) s += l[i][0] * 1e-300;
                lp[2048] = g0; lp[2048 + 64] = g1;
            }
        }
        const unsigned long long c1 = __builtin_amdgcn_s_memtime();
        if (threadIdx.x == 256 && blockIdx.x == 0) cyc[1] = c1 - c0;
        s += iv;
        for (int i = 0; i < 8; ++i) s += v[i] + l[i][1];
    }
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MODE, int PRIO = 0>
void run(const char* name) {
    double *out, *g; unsigned long long* cyc;
    int nblk = 256, iters = 2000;
    (void)hipMalloc(&out, sizeof(double) * nblk * 512); (void)hipMalloc(&cyc, 16); (void)hipMemset(cyc, 0, 16);
    (void)hipMalloc(&g, sizeof(double) * 2 * (4096 * 256 + 8192)); (void)hipMemset(g, 0, sizeof(double) * 2 * (4096 * 256 + 8192));
    hipLaunchKernelGGL((k<MODE, PRIO>), dim3(nblk), dim3(512), 0, 0, out, g, cyc, 50, 0.5);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, PRIO>), dim3(nblk), dim3(512), 0, 0, out, g, cyc, iters, 0.5);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long cc[2]; (void)hipMemcpy(cc, cyc, 16, hipMemcpyDeviceToHost);
    unsigned long long c = cc[0];
    const double nw = MODE == 6 ? 8.0 : 4.0;
    printf("%-58s %7.2f cycles per MFMA of the MFMA wave | kernel %.3f ms = %.1f TFLOP/s\n", name, (double)c / iters / 64,
           ms, nblk * nw * iters * 64.0 * 512.0 / (ms * 1e-3) * 1e-12);
    printf("%-58s companion: %.0f cycles per loop iteration (while the MFMA waves run)\n", "", (double)cc[1] / (iters / 8));
}

int main() {
    run<0>("companion idle");
    run<1>("companion: ds_read_b128 stream");
    run<2>("companion: ds_write_b128 stream");
    run<3>("companion: global_load_dwordx4 stream");
    run<4>("companion: v_fma_f64 stream");
    run<5>("companion: the sweep's staging + A-read mix");
    run<7>("companion: 32-bit VALU stream (64 ops per iteration)");
    run<3, 1>("companion PRIO 3: global_load_dwordx4 stream");
    run<2, 1>("companion PRIO 3: ds_write_b128 stream");
    run<7, 1>("companion PRIO 3: 32-bit VALU stream");
    run<5, 1>("companion PRIO 3: staging + A-read mix");
    run<6>("both waves MFMA (2 MFMA waves per SIMD)");
    return 0;
}
