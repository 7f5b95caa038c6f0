// Probe: do two MFMA wavefronts on one SIMD hide each other's LDS-read issue time, and what
// does a workgroup barrier per tile cost them?   
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/mfma_duo tools/probes/mfma_duo.hip ; run: tools/mfma_duo
// Each wavefront runs TILES x [ NG groups of ( 4 x ds_read_b128 , 32 x v_mfma_f64_4x4x4 ) ], the
// instruction mix of the sweep's matrix role; variants: waves per SIMD (1 or 2), barrier per tile.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x2 __attribute__((ext_vector_type(2)));

template <int WPS, bool BAR, bool LDSR>
__global__ __launch_bounds__(256 * WPS, 1) void duo(double* out, unsigned long long* cyc, int tiles) {
    __shared__ __attribute__((aligned(16))) double lds[8192];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = 1e-3 * (i & 7);
    __syncthreads();
    constexpr int NG = 8 / WPS;              // groups per tile per wavefront (same work per SIMD)
    double acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][r] = 0.0;
    double b[4] = {1.0 + lane, 2.0, 3.0, 4.0};
    f64x2 av[2][4];
    const f64x2* A2 = (const f64x2*)lds + lane;
#pragma unroll
    for (int q = 0; q < 4; ++q) av[0][q] = A2[q * 64];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < tiles; ++t) {
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            // first k-step, then request the next group's operands, then the other three k-steps
            auto kstep = [&](int kk) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    acc[2 * (g & 3)][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[g & 1][kk >> 1][kk & 1], b[r], acc[2 * (g & 3)][r], 0, 0, 0);
                    acc[2 * (g & 3) + 1][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[g & 1][2 + (kk >> 1)][kk & 1], b[r], acc[2 * (g & 3) + 1][r], 0, 0, 0);
                }
            };
            kstep(0);
            __builtin_amdgcn_sched_barrier(0);
            if (BAR && g == NG / 2) __syncthreads();
            if (LDSR) {
#pragma unroll
                for (int q = 0; q < 4; ++q) av[(g + 1) & 1][q] = A2[((g * 4 + q) & 63) * 64];
            }
            kstep(1); kstep(2); kstep(3);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int WPS, bool BAR, bool LDSR>
static void run(const char* name, double* out, unsigned long long* cyc) {
    const int tiles = 20000;
    hipLaunchKernelGGL((duo<WPS, BAR, LDSR>), dim3(256), dim3(256 * WPS), 0, 0, out, cyc, 200);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((duo<WPS, BAR, LDSR>), dim3(256), dim3(256 * WPS), 0, 0, out, cyc, tiles);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double mfma_per_simd = (double)tiles * 8 * 32;      // per SIMD, both variants
    const double tf = 256.0 * 4 * mfma_per_simd * 64 * 4 * 2 / (ms * 1e-3) / 1e12 / 4;   // 4x4x4 x 4 blocks = 256 MAC
    printf("%-44s %8.2f ms  %6.2f memtime-cycles per MFMA per SIMD (100 MHz ticks x24)  %6.1f TF\n", name, ms,
           (double)h * 24.0 / mfma_per_simd, 256.0 * 4 * mfma_per_simd * 512 / (ms * 1e-3) / 1e12);
    (void)tf;
}

int main() {
    double* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 8); hipMalloc(&cyc, 64);
    run<1, false, false>("1 wave/SIMD, MFMA only", out, cyc);
    run<1, false, true>("1 wave/SIMD, + LDS reads", out, cyc);
    run<1, true, true>("1 wave/SIMD, + LDS reads + barrier/tile", out, cyc);
    run<2, false, false>("2 waves/SIMD, MFMA only", out, cyc);
    run<2, false, true>("2 waves/SIMD, + LDS reads", out, cyc);
    run<2, true, true>("2 waves/SIMD, + LDS reads + barrier/tile", out, cyc);
    return 0;
}
