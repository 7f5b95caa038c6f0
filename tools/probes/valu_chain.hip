// fp64 VALU dependent-chain latency vs number of interleaved chains (gfx950, 1 wave/SIMD)
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int NCH>
__global__ __launch_bounds__(256) void k(double* out, unsigned long long* cyc, int iters, double seed) {
    double v[NCH];
    for (int i = 0; i < NCH; ++i) v[i] = seed * (i + 1);
    double a = seed + threadIdx.x * 1e-9, b = 1.0 - seed;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int i = 0; i < NCH; ++i) v[i] = fma(v[i], a, b);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int i = 0; i < NCH; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NCH> void run() {
    double* out; unsigned long long* cyc; int iters = 2000;
    (void)hipMalloc(&out, 8 * 256 * 256); (void)hipMalloc(&cyc, 8);
    hipLaunchKernelGGL((k<NCH>), dim3(256), dim3(256), 0, 0, out, cyc, 10, 0.5);
    hipLaunchKernelGGL((k<NCH>), dim3(256), dim3(256), 0, 0, out, cyc, iters, 0.5);
    (void)hipDeviceSynchronize();
    unsigned long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%d chains: %.2f cycles per fma (%.2f per round of %d)\n", NCH, (double)c / iters / 16 / NCH, (double)c / iters / 16, NCH);
}
int main() { run<1>(); run<2>(); run<3>(); run<4>(); run<6>(); run<8>(); run<16>(); return 0; }
