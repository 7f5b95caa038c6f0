// Probe: what does a feeder wavefront streaming tiles into LDS by LDS-DMA cost the MFMA
// wavefront on its SIMD, whose own instruction mix (4 ds_read_b128 per 32 MFMAs, one barrier per
// 256) is within 1 % of the MFMA-only rate in isolation (tools/probes/mfma_duo.hip)?   
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/mfma_feed tools/probes/mfma_feed.hip ; run: tools/mfma_feed
// Workgroup = 4 matrix wavefronts + 4 feeder wavefronts; per tile each feeder issues NREQ
// buffer_load ... lds requests (1 KiB each) from a 64 MiB stream, then everybody meets at a barrier.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void;

template <int NREQ, bool READS>
__global__ __launch_bounds__(512, 1) void feed(const double* src, unsigned src_bytes, double* out, int tiles) {
    extern __shared__ __attribute__((aligned(16))) double lds[];   // 3 x 32 KiB ring
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    for (int i = t; i < 3 * 4096; i += 512) lds[i] = 1e-3 * (i & 7);
    __syncthreads();
    if (w < 4) {
        double acc[16][4];
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][r] = 0.0;
        double b[4] = {1.0 + lane, 2.0, 3.0, 4.0};
        f64x2 av[2][4];
        int slot = 0;
        const f64x2* A2 = (const f64x2*)lds + lane;
#pragma unroll
        for (int q = 0; q < 4; ++q) av[0][q] = A2[q * 64];
        for (int tl = 0; tl < tiles; ++tl) {
            const f64x2* At = A2 + slot * 2048;
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                auto kstep = [&](int kk) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        acc[2 * g][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[g & 1][kk >> 1][kk & 1], b[r], acc[2 * g][r], 0, 0, 0);
                        acc[2 * g + 1][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[g & 1][2 + (kk >> 1)][kk & 1], b[r], acc[2 * g + 1][r], 0, 0, 0);
                    }
                };
                kstep(0);
                __builtin_amdgcn_sched_barrier(0);
                if (g == 4) __syncthreads();
                if (READS) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) av[(g + 1) & 1][q] = At[(((g + 1) & 7) * 4 + q) * 64];
                }
                kstep(1); kstep(2); kstep(3);
                __builtin_amdgcn_sched_barrier(0);
            }
            slot = slot == 2 ? 0 : slot + 1;
        }
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) s += acc[i][r];
        out[blockIdx.x * 256 + t] = s;
        return;
    }
    const int hw = __builtin_amdgcn_readfirstlane(w - 4);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)src_bytes, 0x00020000);
    unsigned off = (blockIdx.x * 131072u) % (src_bytes - (1u << 20));
    int slot = 1;
    for (int tl = 0; tl < tiles; ++tl) {
        double* dst = lds + slot * 4096 + hw * 128;
#pragma unroll
        for (int q = 0; q < NREQ; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(dst + q * 512), 16, (unsigned)lane * 16u,
                                                     off + (unsigned)(q * 4096) + (unsigned)hw * 1024u, 0, 0);
        off += 32768u;
        if (off > src_bytes - (1u << 20)) off = 0;
        slot = slot == 2 ? 0 : slot + 1;
        __syncthreads();
    }
}

template <int NREQ, bool READS>
static void run(const char* name, const double* src, unsigned bytes, double* out) {
    const int tiles = 20000;
    hipFuncSetAttribute((const void*)feed<NREQ, READS>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 32768);
    hipLaunchKernelGGL((feed<NREQ, READS>), dim3(256), dim3(512), 3 * 32768, 0, src, bytes, out, 200);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((feed<NREQ, READS>), dim3(256), dim3(512), 3 * 32768, 0, src, bytes, out, tiles);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-52s %8.2f ms  %6.1f TF\n", name, ms, 256.0 * 4 * tiles * 256.0 * 512 / (ms * 1e-3) / 1e12);
}

int main() {
    double *src, *out; const unsigned bytes = 64u << 20;
    hipMalloc(&src, bytes); hipMemset(src, 0, bytes); hipMalloc(&out, 256 * 256 * 8);
    run<0, false>("MFMA + barrier, idle feeder", src, bytes, out);
    run<0, true>("+ A reads (4 per 32 MFMAs)", src, bytes, out);
    run<8, true>("+ feeder: 8 LDS-DMA requests per tile (32 KiB/WG)", src, bytes, out);
    run<8, false>("feeder DMA, matrix without A reads", src, bytes, out);
    return 0;
}
