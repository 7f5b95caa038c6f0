// Blocked right-looking Cholesky factorisation (lower, row-major, in place) for
// the GP fit: replaces scipy.linalg.cholesky inside george BasicSolver.compute
// (every gpUtils._nll evaluation, gpUtils.py:74-78, and GP.compute,
// gpUtils.py:178; approx.py:717).  rocSOLVER's dpotrf reaches ~2 TFLOP/s at
// N = 4096 on MI355X (11.8 ms, dozens of tiny kernels); this version is two
// launches per 64-column block step:
//   panel : every workgroup re-factorises the 64x64 diagonal block in LDS
//           (redundantly -- it is the critical path either way, and it saves a
//           launch + a grid-wide dependency), then solves its 256 rows of the
//           panel against it (row-parallel forward substitution);
//   update: trailing lower-triangle tiles A_ik -= L_ij L_kj^T (64x64x64 MFMA
//           f64 tiles; HBM-bound: each tile is read-modify-written once).
// info follows LAPACK: 0 = OK, k > 0 = leading minor of order k not positive
// definite (first failing pivot).
#include "apgp_common.h"

#define PB 64

struct PotrfArgs {
    double* A;
    double* rhs;         // optional right-hand side carried through the factorisation
    long long n, lda;
    long long j0;        // first column of the current block
    double shift;        // rhs is used as (rhs - shift); applied at block step 0 .. as read
    int* info;
};

// factorise the 64x64 block held in LDS (S[64][65]); 256 threads
__device__ __forceinline__ void potf2_lds(double (*S)[PB + 1], double* invd, int bs, long long j0, int* info,
                                          bool reporter) {
    const int t = threadIdx.x;
    for (int k = 0; k < bs; ++k) {
        const double piv = S[k][k];
        // every thread reads the same pivot: uniform decision, no extra barrier
        const bool bad = !(piv > 0.0) || !(piv < INFINITY);
        const double d = bad ? 1.0 : sqrt(piv);
        const double inv = 1.0 / d;
        if (bad && reporter && t == 0) atomicMin((unsigned int*)info, (unsigned int)(j0 + k + 1));
        __syncthreads();
        if (t == 0) { S[k][k] = d; invd[k] = inv; }
        if (t > k && t < bs) S[t][k] *= inv;
        __syncthreads();
        // trailing update of the lower triangle: (i, j) with k < j <= i < bs;
        // 16 x 16 thread grid striding both indices (no integer division)
        for (int i = k + 1 + (t >> 4); i < bs; i += 16) {
            const double lik = S[i][k];
            for (int j = k + 1 + (t & 15); j <= i; j += 16) S[i][j] = fma(-lik, S[j][k], S[i][j]);
        }
        __syncthreads();
    }
}

// panel kernel: blockIdx.x = 0 also writes the diagonal factor back.
// Workgroup b >= 0 owns panel rows [j0 + 64 + 256 b, +256).
__global__ __launch_bounds__(256) void potrf_panel_kernel(PotrfArgs a) {
    __shared__ double S[PB][PB + 1];
    __shared__ double invd[PB];
    __shared__ double zblk[PB];
    const int t = threadIdx.x;
    if (t < PB) invd[t] = 1.0;
    const long long j0 = a.j0;
    const int bs = (int)((a.n - j0) < PB ? (a.n - j0) : PB);
    for (int e = t; e < PB * PB; e += 256) {
        const int i = e >> 6, k = e & 63;
        double v = (i == k) ? 1.0 : 0.0;
        if (i < bs && k <= i) v = a.A[(j0 + i) * a.lda + j0 + k];
        S[i][k] = v;
    }
    __syncthreads();
    potf2_lds(S, invd, bs, j0, a.info, blockIdx.x == 0);
    if (blockIdx.x == 0) {
        for (int e = t; e < PB * PB; e += 256) {
            const int i = e >> 6, k = e & 63;
            if (i < bs && k <= i) a.A[(j0 + i) * a.lda + j0 + k] = S[i][k];
        }
    }
    // fused forward solve z = L^-1 (rhs): every workgroup solves the 64-block of
    // the right-hand side against the fresh diagonal factor (wavefront 0, shuffle
    // forward substitution), then each panel row subtracts its share below.
    if (a.rhs) {
        if (t < 64) {
            double ri = (t < bs) ? a.rhs[j0 + t] : 0.0;
            for (int k = 0; k < bs; ++k) {
                const double zk = __shfl(ri, k) * invd[k];
                if (t == k) ri = zk;
                else if (t > k) ri = fma(-S[t][k], zk, ri);
            }
            zblk[t] = ri;
            if (blockIdx.x == 0 && t < bs) a.rhs[j0 + t] = ri;
        }
        __syncthreads();
    }
    // rows of the panel below the diagonal block: x L_jj^T = a  (row-wise forward substitution)
    const long long row = j0 + PB + (long long)blockIdx.x * 256 + t;
    if (row < a.n) {
        double* ap = a.A + row * a.lda + j0;
        double x[PB];
#pragma unroll
        for (int k = 0; k < PB; ++k) x[k] = ap[k];
#pragma unroll
        for (int k = 0; k < PB; ++k) {
            // four partial sums: the dependent-FMA chain is the cost here
            double s0 = x[k], s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
            for (int m = 0; m + 3 < k; m += 4) {
                s0 = fma(-x[m], S[k][m], s0);
                s1 = fma(-x[m + 1], S[k][m + 1], s1);
                s2 = fma(-x[m + 2], S[k][m + 2], s2);
                s3 = fma(-x[m + 3], S[k][m + 3], s3);
            }
#pragma unroll
            for (int m = k & ~3; m < k; ++m) s0 = fma(-x[m], S[k][m], s0);
            x[k] = ((s0 + s1) + (s2 + s3)) * invd[k];
        }
#pragma unroll
        for (int k = 0; k < PB; ++k) ap[k] = x[k];
        if (a.rhs) {
            double d0 = 0.0, d1 = 0.0;
#pragma unroll
            for (int k = 0; k < PB; k += 2) {
                d0 = fma(x[k], zblk[k], d0);
                d1 = fma(x[k + 1], zblk[k + 1], d1);
            }
            a.rhs[row] -= d0 + d1;
        }
    }
}

// trailing update: tile (bi, bk), bi >= bk, of the blocks below/right of column block j:
// A[ri.., rk..] -= L[ri.., j0..j0+64) * L[rk.., j0..j0+64)^T
__global__ __launch_bounds__(256) void potrf_update_kernel(PotrfArgs a) {
    __shared__ double Ls[PB][PB + 1];   // L[ri + r][j0 + k]
    __shared__ double Rs[PB][PB + 1];   // L[rk + c][j0 + k]
    // linear tile index -> (bi, bk) in the lower triangle
    const long long tix = blockIdx.x;
    long long bi = (long long)((sqrt(8.0 * (double)tix + 1.0) - 1.0) * 0.5);
    while ((bi + 1) * (bi + 2) / 2 <= tix) ++bi;
    while (bi * (bi + 1) / 2 > tix) --bi;
    const long long bk = tix - bi * (bi + 1) / 2;
    const long long base = a.j0 + PB;
    const long long ri = base + bi * PB, rk = base + bk * PB;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    for (int e = t; e < PB * PB; e += 256) {
        const int r = e >> 6, k = e & 63;
        Ls[r][k] = (ri + r < a.n) ? a.A[(ri + r) * a.lda + a.j0 + k] : 0.0;
        Rs[r][k] = (rk + r < a.n) ? a.A[(rk + r) * a.lda + a.j0 + k] : 0.0;
    }
    __syncthreads();
    const int wr = (w >> 1) * 32, wc = (w & 1) * 32;
    f64x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f64x4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < PB / 4; ++ks) {
        double af[2], bf[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) af[i] = Ls[wr + 16 * i + (lane & 15)][ks * 4 + (lane >> 4)];
#pragma unroll
        for (int j = 0; j < 2; ++j) bf[j] = Rs[wc + 16 * j + (lane & 15)][ks * 4 + (lane >> 4)];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const long long gr = ri + wr + 16 * i + (lane >> 4) + 4 * rg;
                const long long gc = rk + wc + 16 * j + (lane & 15);
                if (gr < a.n && gc < a.n && gc <= gr) a.A[gr * a.lda + gc] -= acc[i][j][rg];
            }
}

__global__ __launch_bounds__(256) void potrf_rhs_init_kernel(const double* y, double shift, double* rhs, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) rhs[i] = y[i] - shift;
}

__global__ void potrf_finish_kernel(int* info) {
    if (*(unsigned int*)info == 0xffffffffu) *info = 0;
}

extern "C" int apgp_potrf(double* A, int64_t n, int64_t lda, const double* y, double shift, double* z,
                          int32_t* info_dev, void* stream) {
    APGP_CHECK_ARG(A && info_dev, "null pointer");
    APGP_CHECK_ARG((y == NULL) == (z == NULL), "y and z must be given together");
    APGP_CHECK_ARG(n >= 1 && lda >= n, "n >= 1 and lda >= n required");
    hipStream_t s = (hipStream_t)stream;
    // info = UINT_MAX means "no failure yet"; normalised to 0 by the caller-visible finish kernel
    if (hipMemsetAsync(info_dev, 0xff, sizeof(int32_t), s) != hipSuccess) {
        apgp_set_error("apgp_potrf: memset failed");
        return -2;
    }
    PotrfArgs a;
    a.A = A; a.rhs = z; a.n = n; a.lda = lda; a.shift = shift; a.info = info_dev;
    if (z) hipLaunchKernelGGL(potrf_rhs_init_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, y, shift, z, (long long)n);
    const long long nb = (n + PB - 1) / PB;
    for (long long jb = 0; jb < nb; ++jb) {
        a.j0 = jb * PB;
        const long long below = n - (a.j0 + PB);
        const unsigned pg = below > 0 ? (unsigned)((below + 255) / 256) : 1u;
        hipLaunchKernelGGL(potrf_panel_kernel, dim3(pg), dim3(256), 0, s, a);
        if (below > 0) {
            const long long tb = (below + PB - 1) / PB;
            hipLaunchKernelGGL(potrf_update_kernel, dim3((unsigned)(tb * (tb + 1) / 2)), dim3(256), 0, s, a);
        }
    }
    hipLaunchKernelGGL(potrf_finish_kernel, dim3(1), dim3(1), 0, s, info_dev);
    APGP_CHECK_LAUNCH();
    return 0;
}
