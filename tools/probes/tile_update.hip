// Round 5 probe: the rank-128 tile update of a WIDE launch-per-step Cholesky step (potrf_step_kernel's "other tiles"
// branch: cin - v(col j-1) - v(col j), two 64 x 64 x 64 products accumulated from zero) in isolation, eight ways:
//   V0  the shipped product: operands staged through LDS in 16-column chunks, one chunk in flight, a barrier per chunk
//       (apgp_gemm64_tile, mma16.h);
//   V1  fragments straight from global memory into registers in MFMA layout (8-byte loads, B rotations by DPP), no LDS,
//       no barrier; loads of chunk c + 1 issued before the MFMAs of chunk c;
//   V2  as V1 with ALL 64 fragment loads of a product issued up front;
//   V3  V0's LDS pipeline with all four chunks' global loads issued up front;
//   V4  both operands staged whole, one barrier before and one after the 16 k-steps;
//   V5  V0's products alone (no tile loads, no stores) | V6 no tile loads, real stores | V7 tile loads, no stores.
// MI355X, n = 4096 (1,953 tiles, 2.05 GF): V0 64.9 us (0.40 of the FP64 peak) | V1 133 | V2 137 | V3 62.8 | V4 68.7 |
// V5 50.2 (0.52) | V6 52.2 | V7 57.4 (profiles/r05h_tile_update_probe.txt): the products alone reach half the peak
// however their operands arrive; the tile's own 8-byte strided loads and stores cost another 15 us.
// Same k order per accumulator in V0 .. V4: outputs must agree bit for bit (checked).
// Build: hipcc --offload-arch=gfx950 -O3 -I approxposterior_amd/csrc -I include tools/probes/tile_update.hip -o tools/probes/bin/tile_update
// Run (GPU box): tools/probes/bin/tile_update [n = 4096]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "mma16.h"

#define PB 64

struct Args {
    const double* A;      // n x n, lda = n: block columns 0 and 1 are the operand panels
    double* out;          // updated tiles (same layout)
    long long n, lda;
    long long tb;         // trailing matrix: blocks 2 .. 2 + tb - 1
};

__device__ __forceinline__ void tile_of(long long tix, long long& bi, long long& bk) {
    long long b = (long long)((sqrt(8.0 * (double)tix + 1.0) - 1.0) * 0.5);
    while ((b + 1) * (b + 2) / 2 <= tix) ++b;
    while (b * (b + 1) / 2 > tix) --b;
    bi = b;
    bk = tix - b * (b + 1) / 2;
}

// ---- V1 / V2: one 64 x 64 x 64 product, fragments from global memory ----
template <bool UPFRONT>
__device__ __forceinline__ void product_direct(const double* __restrict__ Ap, long long lda, long long a_rows,
                                               const double* __restrict__ Bp, long long ldb, long long b_rows,
                                               double (&acc)[2][2][4]) {
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int wr = (w >> 1) * 32, wc = (w & 1) * 32;
    const int kq = lane >> 4, l15 = lane & 15;
    const double* pa[2];
    const double* pb[2];
    bool oka[2], okb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const long long ra = wr + 16 * i + l15, rb = wc + 16 * i + l15;
        oka[i] = ra < a_rows; okb[i] = rb < b_rows;
        pa[i] = Ap + (oka[i] ? ra : 0) * lda + kq;
        pb[i] = Bp + (okb[i] ? rb : 0) * ldb + kq;
    }
    if (UPFRONT) {
        double fa[2][16], fb[2][16];
#pragma unroll
        for (int g = 0; g < 16; ++g)
#pragma unroll
            for (int i = 0; i < 2; ++i) { fa[i][g] = pa[i][4 * g]; fb[i][g] = pb[i][4 * g]; }
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            double bf[2][4];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const double b0 = okb[j] ? fb[j][g] : 0.0;
                bf[j][0] = b0; bf[j][1] = apgp_row_ror4<1>(b0); bf[j][2] = apgp_row_ror4<2>(b0); bf[j][3] = apgp_row_ror4<3>(b0);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        acc[i][j][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(oka[i] ? fa[i][g] : 0.0, bf[j][r], acc[i][j][r], 0, 0, 0);
        }
    } else {
        double fa[2][2][4], fb[2][2][4];            // [buffer][block][k-step of the chunk]
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i) { fa[0][i][ks] = pa[i][4 * ks]; fb[0][i][ks] = pb[i][4 * ks]; }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int cur = c & 1, nxt = cur ^ 1;
            if (c < 3) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        fa[nxt][i][ks] = pa[i][16 * (c + 1) + 4 * ks];
                        fb[nxt][i][ks] = pb[i][16 * (c + 1) + 4 * ks];
                    }
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                double bf[2][4];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const double b0 = okb[j] ? fb[cur][j][ks] : 0.0;
                    bf[j][0] = b0; bf[j][1] = apgp_row_ror4<1>(b0); bf[j][2] = apgp_row_ror4<2>(b0); bf[j][3] = apgp_row_ror4<3>(b0);
                }
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            acc[i][j][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(oka[i] ? fa[cur][i][ks] : 0.0, bf[j][r], acc[i][j][r], 0, 0, 0);
            }
        }
    }
}

// ---- V3: the shipped LDS pipeline with all four chunks' global loads issued up front ----
__device__ __forceinline__ void product_k64(const double* __restrict__ Ap, long long lda, long long a_rows,
                                            const double* __restrict__ Bp, long long ldb, long long b_rows,
                                            double* lds, double (&acc)[2][2][4]) {
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int wr = (w >> 1) * 32, wc = (w & 1) * 32;
    const int rr = t >> 2, rq = (t & 3) * 4;
    const bool a_ok = rr < a_rows, b_ok = rr < b_rows;
    int bcol[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bcol[r] = ((lane & 15) - 4 * r) & 15;
    f64x2 ra[4][2], rb[4][2];
    const double* pa = Ap + (long long)(a_ok ? rr : 0) * lda + rq;
    const double* pb = Bp + (long long)(b_ok ? rr : 0) * ldb + rq;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        ra[c][0] = *(const f64x2_g*)(pa + 16 * c); ra[c][1] = *(const f64x2_g*)(pa + 16 * c + 2);
        rb[c][0] = *(const f64x2_g*)(pb + 16 * c); rb[c][1] = *(const f64x2_g*)(pb + 16 * c + 2);
    }
    const f64x2 zero = {0.0, 0.0};
    auto sstore = [&](int c) {
        double* As = lds + (c & 1) * 2560;
        double* Bs = As + 1280;
        double* da = As + rr * 18 + rq;
        double* db = Bs + rr * 18 + rq;
        *(f64x2*)da = a_ok ? ra[c][0] : zero; *(f64x2*)(da + 2) = a_ok ? ra[c][1] : zero;
        *(f64x2*)db = b_ok ? rb[c][0] : zero; *(f64x2*)(db + 2) = b_ok ? rb[c][1] : zero;
    };
    sstore(0);
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const double* As = lds + (c & 1) * 2560;
        const double* Bs = As + 1280;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            double af[2], bf[2][4];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = As[(wr + 16 * i + (lane & 15)) * 18 + ks * 4 + (lane >> 4)];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int j = 0; j < 2; ++j) bf[j][r] = Bs[(wc + 16 * j + bcol[r]) * 18 + ks * 4 + (lane >> 4)];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        acc[i][j][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(af[i], bf[j][r], acc[i][j][r], 0, 0, 0);
        }
        if (c < 3) sstore(c + 1);
        __syncthreads();
    }
}

// ---- V4: both operands of a product staged WHOLE (4 chunks each), one barrier before and one after the 16 k-steps ----
#define V4_LDS (2 * 4 * 64 * 18)
__device__ __forceinline__ void product_whole(const double* __restrict__ Ap, long long lda, long long a_rows,
                                              const double* __restrict__ Bp, long long ldb, long long b_rows,
                                              double* lds, double (&acc)[2][2][4]) {
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int wr = (w >> 1) * 32, wc = (w & 1) * 32;
    int bcol[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bcol[r] = ((lane & 15) - 4 * r) & 15;
    // thread -> row e >> 5, columns 2 (e & 31), four rounds (coalesced 512 B per row)
    f64x2 ra[4], rb[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int e = it * 256 + t, rw = e >> 4, col = 4 * (e & 15);
        (void)col;
    }
    // 64 rows x 64 columns = 2048 f64x2 per operand; 256 threads x 8 each
    f64x2 va[8], vb[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int e = it * 256 + t, rw = e >> 5, col = 2 * (e & 31);
        va[it] = *(const f64x2_g*)(Ap + (long long)(rw < a_rows ? rw : 0) * lda + col);
        vb[it] = *(const f64x2_g*)(Bp + (long long)(rw < b_rows ? rw : 0) * ldb + col);
    }
    (void)ra; (void)rb;
    const f64x2 zero = {0.0, 0.0};
    double* As = lds;
    double* Bs = lds + 4 * 64 * 18;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int e = it * 256 + t, rw = e >> 5, col = 2 * (e & 31);
        *(f64x2*)(As + (col >> 4) * (64 * 18) + rw * 18 + (col & 15)) = rw < a_rows ? va[it] : zero;
        *(f64x2*)(Bs + (col >> 4) * (64 * 18) + rw * 18 + (col & 15)) = rw < b_rows ? vb[it] : zero;
    }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < 16; ++g) {
        const double* Ach = As + (g >> 2) * (64 * 18);
        const double* Bch = Bs + (g >> 2) * (64 * 18);
        const int ks = g & 3;
        double af[2], bf[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i) af[i] = Ach[(wr + 16 * i + (lane & 15)) * 18 + ks * 4 + (lane >> 4)];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j][r] = Bch[(wc + 16 * j + bcol[r]) * 18 + ks * 4 + (lane >> 4)];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    acc[i][j][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(af[i], bf[j][r], acc[i][j][r], 0, 0, 0);
    }
    __syncthreads();
}

template <int V>
__global__ __launch_bounds__(256, 2) void update_kernel(Args a) {
    __shared__ __attribute__((aligned(16))) double lds[V == 0 || V == 3 || V >= 5 ? GEMM64_LDS_DOUBLES : (V == 4 ? V4_LDS : 2)];
    long long bi, bk;
    tile_of(blockIdx.x, bi, bk);
    const long long base = 2 * PB;
    const long long ri = base + bi * PB, rk = base + bk * PB;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int wr = (w >> 1) * 32, wc = (w & 1) * 32;
    const long long jc = PB;
    double cin[2][2][4], v[2][2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long long gr = ri + wr + 16 * i + apgp_mma16_row(lane);
                const long long gc = rk + wc + 16 * j + apgp_mma16_col(lane, r);
                cin[i][j][r] = (V != 5 && V != 6 && gr < a.n && gc < a.n && gc <= gr) ? a.A[gr * a.lda + gc] : 0.0;
                v[i][j][r] = 0.0;
            }
    if (V == 0 || V >= 5) apgp_gemm64_tile<false, false>(a.A + ri * a.lda + jc - PB, a.lda, a.n - ri, a.A + rk * a.lda + jc - PB, a.lda, a.n - rk, 0, PB, lds, v);
    else if (V == 3) product_k64(a.A + ri * a.lda + jc - PB, a.lda, a.n - ri, a.A + rk * a.lda + jc - PB, a.lda, a.n - rk, lds, v);
    else if (V == 4) product_whole(a.A + ri * a.lda + jc - PB, a.lda, a.n - ri, a.A + rk * a.lda + jc - PB, a.lda, a.n - rk, lds, v);
    else product_direct<V == 2>(a.A + ri * a.lda + jc - PB, a.lda, a.n - ri, a.A + rk * a.lda + jc - PB, a.lda, a.n - rk, v);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                cin[i][j][r] = cin[i][j][r] - v[i][j][r];
                v[i][j][r] = 0.0;
            }
    if (V == 0 || V >= 5) apgp_gemm64_tile<false, false>(a.A + ri * a.lda + jc, a.lda, a.n - ri, a.A + rk * a.lda + jc, a.lda, a.n - rk, 0, PB, lds, v);
    else if (V == 3) product_k64(a.A + ri * a.lda + jc, a.lda, a.n - ri, a.A + rk * a.lda + jc, a.lda, a.n - rk, lds, v);
    else if (V == 4) product_whole(a.A + ri * a.lda + jc, a.lda, a.n - ri, a.A + rk * a.lda + jc, a.lda, a.n - rk, lds, v);
    else product_direct<V == 2>(a.A + ri * a.lda + jc, a.lda, a.n - ri, a.A + rk * a.lda + jc, a.lda, a.n - rk, v);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long long gr = ri + wr + 16 * i + apgp_mma16_row(lane);
                const long long gc = rk + wc + 16 * j + apgp_mma16_col(lane, r);
                const double res = cin[i][j][r] - v[i][j][r];
                if (gr < a.n && gc < a.n && gc <= gr && ((V != 5 && V != 7) || res != res)) a.out[gr * a.lda + gc] = res;
            }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
    const long long n = argc > 1 ? atoll(argv[1]) : 4096;
    const long long nb = (n + PB - 1) / PB, tb = nb - 2;
    const long long tiles = tb * (tb + 1) / 2;
    std::vector<double> h((size_t)n * n);
    unsigned long long sd = 88172645463325252ull;
    for (auto& x : h) { sd ^= sd << 13; sd ^= sd >> 7; sd ^= sd << 17; x = ((double)(sd >> 11) / 9007199254740992.0 - 0.5) * 0.25; }
    double *A, *o[8];
    CK(hipMalloc(&A, sizeof(double) * n * n));
    CK(hipMemcpy(A, h.data(), sizeof(double) * n * n, hipMemcpyHostToDevice));
    for (int k = 0; k < 8; ++k) { CK(hipMalloc(&o[k], sizeof(double) * n * n)); CK(hipMemset(o[k], 0, sizeof(double) * n * n)); }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("n = %lld: %lld tiles (rank-128 update each), %.2f GF per launch\n", n, tiles, tiles * 2 * 2.0 * 64 * 64 * 64 / 1e9);
    for (int k = 0; k < 8; ++k) {
        Args a{A, o[k], n, n, tb};
        float best = 1e30f, sum = 0.f;
        const int reps = 20;
        for (int rep = 0; rep < reps + 3; ++rep) {
            CK(hipEventRecord(e0, 0));
            if (k == 0) hipLaunchKernelGGL(update_kernel<0>, dim3((unsigned)tiles), dim3(256), 0, 0, a);
            if (k == 1) hipLaunchKernelGGL(update_kernel<1>, dim3((unsigned)tiles), dim3(256), 0, 0, a);
            if (k == 2) hipLaunchKernelGGL(update_kernel<2>, dim3((unsigned)tiles), dim3(256), 0, 0, a);
            if (k == 3) hipLaunchKernelGGL(update_kernel<3>, dim3((unsigned)tiles), dim3(256), 0, 0, a);
            if (k == 4) hipLaunchKernelGGL(update_kernel<4>, dim3((unsigned)tiles), dim3(256), 0, 0, a);
            if (k == 5) hipLaunchKernelGGL(update_kernel<5>, dim3((unsigned)tiles), dim3(256), 0, 0, a);
            if (k == 6) hipLaunchKernelGGL(update_kernel<6>, dim3((unsigned)tiles), dim3(256), 0, 0, a);
            if (k == 7) hipLaunchKernelGGL(update_kernel<7>, dim3((unsigned)tiles), dim3(256), 0, 0, a);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep >= 3) { sum += ms; if (ms < best) best = ms; }
        }
        const double gf = tiles * 2 * 2.0 * 64 * 64 * 64 / 1e9;
        printf("V%d: %.1f us avg, %.1f us best  (%.1f TF, %.2f of 78.6)\n", k, sum / reps * 1e3, best * 1e3, gf / (best * 1e-3) / 1e3,
               gf / (best * 1e-3) / 1e3 / 78.6);
    }
    std::vector<double> r0((size_t)n * n), r1((size_t)n * n);
    CK(hipMemcpy(r0.data(), o[0], sizeof(double) * n * n, hipMemcpyDeviceToHost));
    for (int k = 1; k < 5; ++k) {
        CK(hipMemcpy(r1.data(), o[k], sizeof(double) * n * n, hipMemcpyDeviceToHost));
        long long bad = 0;
        for (size_t i = 0; i < r0.size(); ++i) bad += memcmp(&r0[i], &r1[i], 8) != 0;
        printf("V%d vs V0: %lld differing elements\n", k, bad);
    }
    return 0;
}
