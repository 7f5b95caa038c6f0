// Shared by the GEMM-shaped fit kernels (potrf.hip, linalg.hip, grad.hip).
#pragma once
#include "apgp_common.h"

// ---------------------------------------------------------------------------
// 16 x 16 (+)= (16 x 4) * (4 x 16) on v_mfma_f64_4x4x4_4b_f64, the fp64 matrix instruction that
// runs at the FP64 peak on gfx950 (v_mfma_f64_16x16x4_f64 runs at less than half of it; DESIGN.md
// section 2).  Operand fragments are those of the 16x16x4 instruction (A: lane = row + 16 k,
// B: lane = col + 16 k).  The four-block instruction multiplies row group b (rows 4b .. 4b+3)
// with column group b only, so the product takes four of them with the B fragment rotated by
// 4 r lanes inside each 16-lane row (DPP row_ror: lane l reads lane l - 4 r, tools/probes/mma16_probe.hip):
// rotation r pairs row group b with column group (b - r) & 3.  acc[r] of lane l then holds element
//     row = 4 * ((l >> 2) & 3) + (l >> 4),   col = 4 * ((((l >> 2) & 3) - r) & 3) + (l & 3).
// ---------------------------------------------------------------------------
template <int R>
__device__ __forceinline__ double apgp_row_ror4(double v) {
    if (R == 0) return v;
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x120 + 4 * R, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x120 + 4 * R, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
struct ApgpBRot { double r[4]; };
__device__ __forceinline__ ApgpBRot apgp_brot(double bf) {
    ApgpBRot b;
    b.r[0] = bf; b.r[1] = apgp_row_ror4<1>(bf); b.r[2] = apgp_row_ror4<2>(bf); b.r[3] = apgp_row_ror4<3>(bf);
    return b;
}
__device__ __forceinline__ void apgp_mma16(double af, const ApgpBRot& b, double (&acc)[4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = __builtin_amdgcn_mfma_f64_4x4x4f64(af, b.r[r], acc[r], 0, 0, 0);
}
__device__ __forceinline__ int apgp_mma16_row(int lane) { return 4 * ((lane >> 2) & 3) + (lane >> 4); }
__device__ __forceinline__ int apgp_mma16_col(int lane, int r) { return 4 * ((((lane >> 2) & 3) - r) & 3) + (lane & 3); }

// ---------------------------------------------------------------------------
// 64 x 64 output tile of C = A B over k in [k0, k1) (multiples of 16), 256 threads = 4 wavefronts
// (2 x 2), each 32 x 32 = 2 x 2 blocks of apgp_mma16.  K is staged through LDS in chunks of 16,
// double-buffered: the global loads of chunk c + 1 are in flight while chunk c is multiplied, one
// barrier per chunk (40 KiB of LDS: four workgroups per CU).  Either operand is k-major when its
// flag is set (A[k][row] / B[k][col]) and row-major otherwise (A[row][k] / B[col][k]; rows >=
// a_rows / b_rows of a row-major operand read as zero).
// acc[i][j][r]: block (i, j) of the wavefront's 32 x 32 in the layout of apgp_mma16.
// ---------------------------------------------------------------------------
#define GEMM64_LDS_DOUBLES (2 * (16 * 80 + 16 * 80))
typedef double f64x2_g __attribute__((ext_vector_type(2), aligned(8)));   // global operand rows are only 8-byte aligned (ld = n)
// With BB (row-major B only) the product B B^T of the B rows rides along in acc2 (same tile shape:
// its A fragments are read from the B chunk), at the cost of its MFMAs alone.
// `t` = index of the thread among the 256 that share the tile (threadIdx.x for a 256-thread workgroup; a
// 512-thread workgroup runs two tiles side by side, t = threadIdx.x & 255, each half with its own `lds`; the
// barriers are the workgroup's, so both halves make the call together -- a half without a tile passes
// a_rows = b_rows = 0 and multiplies zeros).
template <bool AK, bool BK, bool BB>
__device__ __forceinline__ void apgp_gemm64_tile2_t(const int t, const double* __restrict__ Ap, long long lda, long long a_rows,
                                                    const double* __restrict__ Bp, long long ldb, long long b_rows,
                                                    long long k0, long long k1, double* lds, double (&acc)[2][2][4],
                                                    double (&acc2)[2][2][4]) {
    static_assert(!BB || !BK, "B B^T needs the row-major B chunk");
    // LDS: [buffer][A 1280 | B 1280]; a k-major chunk is [16][80], a row-major one [64][18]
    const int lane = t & 63, w = t >> 6;
    const int wr = (w >> 1) * 32, wc = (w & 1) * 32;
    const int kr = t >> 4, cq = (t & 15) * 4;       // k-major chunk: thread -> (k, 4 consecutive columns)
    const int rr = t >> 2, rq = (t & 3) * 4;        // row-major chunk: thread -> (row, 4 consecutive k)
    const bool a_ok = AK || rr < a_rows, b_ok = BK || rr < b_rows;
    int bcol[4];                                     // column of the B fragment lane under rotation r (apgp_brot)
#pragma unroll
    for (int r = 0; r < 4; ++r) bcol[r] = ((lane & 15) - 4 * r) & 15;
    f64x2 ra[2], rb[2];
    auto gload = [&](long long kk) {
        const double* pa = AK ? Ap + (kk + kr) * lda + cq : Ap + (long long)rr * lda + kk + rq;
        const double* pb = BK ? Bp + (kk + kr) * ldb + cq : Bp + (long long)rr * ldb + kk + rq;
        if (a_ok) { ra[0] = *(const f64x2_g*)pa; ra[1] = *(const f64x2_g*)(pa + 2); }
        else { ra[0] = (f64x2){0.0, 0.0}; ra[1] = ra[0]; }
        if (b_ok) { rb[0] = *(const f64x2_g*)pb; rb[1] = *(const f64x2_g*)(pb + 2); }
        else { rb[0] = (f64x2){0.0, 0.0}; rb[1] = rb[0]; }
    };
    auto sstore = [&](int buf) {
        double* As = lds + buf * 2560;
        double* Bs = As + 1280;
        double* da = AK ? As + kr * 80 + cq : As + rr * 18 + rq;
        double* db = BK ? Bs + kr * 80 + cq : Bs + rr * 18 + rq;
        *(f64x2*)da = ra[0]; *(f64x2*)(da + 2) = ra[1];
        *(f64x2*)db = rb[0]; *(f64x2*)(db + 2) = rb[1];
    };
    if (k0 >= k1) return;
    gload(k0);
    sstore(0);
    __syncthreads();
    int buf = 0;
    for (long long kk = k0; kk < k1; kk += 16) {
        const bool more = kk + 16 < k1;
        if (more) gload(kk + 16);
        const double* As = lds + buf * 2560;
        const double* Bs = As + 1280;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            // the four rotations of a B fragment are four reads at rotated addresses (LDS is nearly
            // idle here; six DPP moves per fragment cost more issue slots than the MFMAs leave)
            double af[2], bf[2][4];
#pragma unroll
            for (int i = 0; i < 2; ++i)
                af[i] = AK ? As[(ks * 4 + (lane >> 4)) * 80 + wr + 16 * i + (lane & 15)]
                           : As[(wr + 16 * i + (lane & 15)) * 18 + ks * 4 + (lane >> 4)];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    bf[j][r] = BK ? Bs[(ks * 4 + (lane >> 4)) * 80 + wc + 16 * j + bcol[r]]
                                  : Bs[(wc + 16 * j + bcol[r]) * 18 + ks * 4 + (lane >> 4)];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        acc[i][j][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(af[i], bf[j][r], acc[i][j][r], 0, 0, 0);
            if constexpr (BB) {
                double bfa[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) bfa[i] = Bs[(wr + 16 * i + (lane & 15)) * 18 + ks * 4 + (lane >> 4)];
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            acc2[i][j][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(bfa[i], bf[j][r], acc2[i][j][r], 0, 0, 0);
            }
        }
        if (more) sstore(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
}

template <bool AK, bool BK, bool BB>
__device__ __forceinline__ void apgp_gemm64_tile2(const double* __restrict__ Ap, long long lda, long long a_rows,
                                                  const double* __restrict__ Bp, long long ldb, long long b_rows,
                                                  long long k0, long long k1, double* lds, double (&acc)[2][2][4],
                                                  double (&acc2)[2][2][4]) {
    apgp_gemm64_tile2_t<AK, BK, BB>((int)threadIdx.x, Ap, lda, a_rows, Bp, ldb, b_rows, k0, k1, lds, acc, acc2);
}

template <bool AK, bool BK>
__device__ __forceinline__ void apgp_gemm64_tile(const double* __restrict__ Ap, long long lda, long long a_rows,
                                                 const double* __restrict__ Bp, long long ldb, long long b_rows,
                                                 long long k0, long long k1, double* lds, double (&acc)[2][2][4]) {
    double none[2][2][4];
    apgp_gemm64_tile2<AK, BK, false>(Ap, lda, a_rows, Bp, ldb, b_rows, k0, k1, lds, acc, none);
}
