// Probe (one wavefront): issue cost / latency of the building blocks of the Cholesky panel kernel:
// uniform-address (broadcast) and per-lane ds_read_b128 / b64, v_readlane_b32, dependent and
// independent v_fma_f64, v_rsq_f64.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x2 __attribute__((ext_vector_type(2)));
__global__ void k(double* out, unsigned long long* cyc, int mode) {
    __shared__ __attribute__((aligned(16))) double buf[4096];
    const int lane = threadIdx.x;
    for (int i = lane; i < 4096; i += 64) buf[i] = 1.0 + 1e-9 * i;
    __syncthreads();
    double acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
    double x = 1.0 + lane * 1e-6;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (mode == 0) {          // broadcast b128
        for (int it = 0; it < 64; ++it) {
            f64x2 v[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) v[q] = *(const f64x2*)(&buf[(it * 32 + q * 2) & 4095]);
#pragma unroll
            for (int q = 0; q < 16; ++q) { acc0 += v[q].x; acc1 += v[q].y; }
        }
    } else if (mode == 1) {   // per-lane b128 (conflict-free: consecutive 16 B)
        for (int it = 0; it < 64; ++it) {
            f64x2 v[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) v[q] = *(const f64x2*)(&buf[(it * 32 + q * 128 + lane * 2) & 4095]);
#pragma unroll
            for (int q = 0; q < 16; ++q) { acc0 += v[q].x; acc1 += v[q].y; }
        }
    } else if (mode == 2) {   // broadcast b64
        for (int it = 0; it < 64; ++it) {
            double v[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) v[q] = buf[(it * 16 + q) & 4095];
#pragma unroll
            for (int q = 0; q < 16; ++q) acc0 += v[q];
        }
    } else if (mode == 3) {   // dependent fma chain
        for (int it = 0; it < 64; ++it) {
#pragma unroll
            for (int q = 0; q < 16; ++q) x = fma(x, 0.999999, 1e-7);
        }
    } else if (mode == 4) {   // 4 independent fma chains
        double y = x + 1, z = x + 2, w = x + 3;
        for (int it = 0; it < 64; ++it) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { x = fma(x, 0.999999, 1e-7); y = fma(y, 0.999999, 1e-7); z = fma(z, 0.999999, 1e-7); w = fma(w, 0.999999, 1e-7); }
        }
        x += y + z + w;
    } else if (mode == 5) {   // dependent rsq chain
        for (int it = 0; it < 64; ++it) {
#pragma unroll
            for (int q = 0; q < 16; ++q) x = __builtin_amdgcn_rsq(x);
        }
    } else if (mode == 6) {   // readlane
        for (int it = 0; it < 64; ++it) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int lo = __builtin_amdgcn_readlane(__double2loint(x), q * 3);
                acc0 += __hiloint2double(0x3ff00000, lo);
            }
        }
    } else if (mode == 7) {   // broadcast b128 + 8 independent fma per 4 reads (the trailing-update mix)
        double r0 = x, r1 = x + 1, r2 = x + 2, r3 = x + 3;
        for (int it = 0; it < 64; ++it) {
            f64x2 v[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) v[q] = *(const f64x2*)(&buf[(it * 32 + q * 2) & 4095]);
#pragma unroll
            for (int q = 0; q < 16; q += 4) {
                r0 = fma(x, v[q].x, r0); r1 = fma(x, v[q + 1].x, r1); r2 = fma(x, v[q + 2].x, r2); r3 = fma(x, v[q + 3].x, r3);
                r0 = fma(x, v[q].y, r0); r1 = fma(x, v[q + 1].y, r1); r2 = fma(x, v[q + 2].y, r2); r3 = fma(x, v[q + 3].y, r3);
            }
        }
        x = r0 + r1 + r2 + r3;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[lane + 64 * mode] = x + acc0 + acc1 + acc2 + acc3;
    if (lane == 0) cyc[mode] = t1 - t0;
}
int main() {
    double* out; unsigned long long* cyc;
    (void)hipMalloc(&out, 8 * 64 * 8); (void)hipMalloc(&cyc, 8 * 8);
    const char* names[8] = {"broadcast ds_read_b128", "per-lane ds_read_b128", "broadcast ds_read_b64", "dependent v_fma_f64", "4 independent v_fma_f64 chains", "dependent v_rsq_f64", "v_readlane_b32 + use", "16 broadcast b128 + 32 fma"};
    for (int rep = 0; rep < 2; ++rep)
        for (int m = 0; m < 8; ++m) {
            k<<<1, 64>>>(out, cyc, m);
            (void)hipDeviceSynchronize();
            unsigned long long c; (void)hipMemcpy(&c, cyc + m, 8, hipMemcpyDeviceToHost);
            if (rep) printf("%-34s %8.2f cycles per operation (1024 operations)\n", names[m], (double)c / 1024.0);
        }
    return 0;
}
