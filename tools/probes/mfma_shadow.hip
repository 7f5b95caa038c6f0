// Micro-benchmark: what rides in the shadow of v_mfma_f64_4x4x4_4b_f64 on gfx950
// with ONE wavefront per SIMD?  Each variant issues 64 MFMAs per iteration with a
// given filler after every MFMA (or every 2nd/4th).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double f64x2 __attribute__((ext_vector_type(2)));

template <int FILL, int EVERY>
__global__ __launch_bounds__(256) void k(double* out, unsigned long long* cyc, int iters, double seed) {
    __shared__ __attribute__((aligned(16))) double lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = seed * i;
    __syncthreads();
    double acc[64];
    for (int i = 0; i < 64; ++i) acc[i] = seed;
    double a = seed + threadIdx.x * 1e-9, b = 1.0 - seed;
    f64x2 l[8];
    for (int i = 0; i < 8; ++i) l[i] = (f64x2){seed, seed};
    double v[8];
    for (int i = 0; i < 8; ++i) v[i] = seed * i;
    int iv[8];
    for (int i = 0; i < 8; ++i) iv[i] = threadIdx.x + i;
    const f64x2* lp = (const f64x2*)lds + (threadIdx.x & 63);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
            if (i % EVERY == 0) {
                if (FILL == 1) l[i & 7] = lp[(i & 15) * 64];                    // ds_read_b128
                if (FILL == 2) v[i & 7] = fma(v[i & 7], a, b);                  // v_fma_f64
                if (FILL == 3) iv[i & 7] = iv[i & 7] * 3 + 1;                   // 32-bit VALU
                if (FILL == 4) asm volatile("s_nop 0");
                if (FILL == 5) { l[i & 7] = lp[(i & 15) * 64]; iv[i & 7] = iv[i & 7] * 3 + 1; }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (FILL == 1 || FILL == 5) { for (int i = 0; i < 8; ++i) a += l[i][0] * 1e-300; }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int i = 0; i < 64; ++i) s += acc[i];
    for (int i = 0; i < 8; ++i) s += v[i] + iv[i] + l[i][1];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int FILL, int EVERY>
void run(const char* name) {
    double* out; unsigned long long* cyc;
    int nblk = 256, iters = 2000;
    (void)hipMalloc(&out, sizeof(double) * nblk * 256); (void)hipMalloc(&cyc, 8);
    hipLaunchKernelGGL((k<FILL, EVERY>), dim3(nblk), dim3(256), 0, 0, out, cyc, 50, 0.5);
    (void)hipDeviceSynchronize();
    hipLaunchKernelGGL((k<FILL, EVERY>), dim3(nblk), dim3(256), 0, 0, out, cyc, iters, 0.5);
    (void)hipDeviceSynchronize();
    unsigned long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-40s %7.2f cycles per MFMA\n", name, (double)c / iters / 64);
}

int main() {
    run<0, 1>("mfma only");
    run<1, 1>("mfma + ds_read_b128 every 1");
    run<1, 2>("mfma + ds_read_b128 every 2");
    run<1, 4>("mfma + ds_read_b128 every 4");
    run<2, 1>("mfma + v_fma_f64 every 1");
    run<2, 4>("mfma + v_fma_f64 every 4");
    run<3, 1>("mfma + 2x 32-bit valu every 1");
    run<3, 4>("mfma + 2x 32-bit valu every 4");
    run<4, 1>("mfma + s_nop every 1");
    run<5, 1>("mfma + ds_read + int every 1");
    run<5, 2>("mfma + ds_read + int every 2");
    return 0;
}
