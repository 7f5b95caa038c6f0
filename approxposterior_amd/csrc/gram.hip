// K1 "gram": lower triangle of the N x N Gram matrix  K_ij = amp * exp(-|xs_i - xs_j|^2)
// [+ lin_coef * sum_d (x_id x_jd)^P], LDS-tiled, coalesced row writes.  HBM-write-bound
// (4 N^2 bytes out -- only the 64 x 64 tiles on and below the diagonal, which is all the
// Cholesky reads --, 8 N D bytes in).
// Reference semantics: george ExpSquaredKernel.get_value + the diagonal update
// of GP.compute (called from gpUtils.py:178,244,254; approx.py:717).
#include "apgp_common.h"
#include <stdarg.h>
#include <string.h>

static thread_local char g_err[512] = "";

void apgp_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* apgp_last_error(void) { return g_err; }
extern "C" int apgp_abi_version(void) { return APGP_ABI_VERSION; }

struct GramArgs {
    const double* X;
    double* K;
    long long n, ldk;
    KernConst kc;
    // optional (apgp_nll_eval): the Cholesky's right-hand side z = y - shift and its info word are
    // initialised by this launch instead of a memset and a kernel of their own
    const double* y;
    double* z;
    int* info;
    double shift;
};

// 64 x 64 output tile per workgroup, 256 threads: thread = (column c, row
// group g); each thread produces 16 rows of its column so that a wavefront
// writes 64 consecutive doubles (512 B) of one row at a time.
template <int DPAD>
__device__ __forceinline__ void gram_body(const GramArgs& a) {
    __shared__ double xi[64][DPAD + 1];
    __shared__ double xj[64][DPAD + 1];
    __shared__ double etab[APGP_EXP_TAB_N];
    apgp_exp_tab_load(etab);
    const int t = threadIdx.x;
    // lower triangle only: linear tile index -> (row block bi, column block bj <= bi)
    const long long tix = blockIdx.x;
    long long bi = (long long)((sqrt(8.0 * (double)tix + 1.0) - 1.0) * 0.5);
    while ((bi + 1) * (bi + 2) / 2 <= tix) ++bi;
    while (bi * (bi + 1) / 2 > tix) --bi;
    const long long i0 = bi * 64, j0 = (tix - bi * (bi + 1) / 2) * 64;
    if (a.z && j0 == 0 && t < 64 && i0 + t < a.n) a.z[i0 + t] = a.y[i0 + t] - a.shift;    // (tile column 0 covers all rows)
    if (a.info && tix == 0 && t == 0) *(unsigned int*)a.info = 0xffffffffu;               // "no failure yet" (potrf.hip)
    for (int e = t; e < 64 * DPAD; e += 256) {
        int r = e / DPAD, d = e % DPAD;
        long long gi = i0 + r, gj = j0 + r;
        double vi = 0.0, vj = 0.0;
        if (d < a.kc.ndim) {
            if (gi < a.n) vi = a.X[gi * a.kc.ndim + d] * a.kc.sc[d];
            if (gj < a.n) vj = a.X[gj * a.kc.ndim + d] * a.kc.sc[d];
        }
        xi[r][d] = vi;
        xj[r][d] = vj;
    }
    __syncthreads();
    const int c = t & 63, g = t >> 6;
    double xc[DPAD];
#pragma unroll
    for (int d = 0; d < DPAD; ++d) xc[d] = xj[c][d];
    const long long gj = j0 + c;
#pragma unroll 4
    for (int q = 0; q < 16; ++q) {
        const int r = g + 4 * q;
        const long long gi = i0 + r;
        // (same association as the sweep / mean kernels: two interleaved sums -- apgp_gram_value)
        const double k = apgp_gram_value<DPAD>(xi[r], xc, a.kc, gi == gj, etab);
        if (gi < a.n && gj <= gi) a.K[gi * a.ldk + gj] = k;      // (lower triangle only: LAPACK-style, as apgp_potrf)
    }
}

template <int DPAD>
__global__ __launch_bounds__(256) void gram_kernel(GramArgs a) { gram_body<DPAD>(a); }

// `batch` Gram matrices of the SAME training set at different hyper-parameters in ONE launch (gridDim.y = batch): the
// argument records of all matrices travel in the kernel-argument segment, a workgroup picks its own by blockIdx.y and runs
// the single launch's body -- the same values (apgp_nll_eval_batch for a Powell look-ahead, potrf.hip).
#define APGP_GRAM_BATCH_MAX 6
struct GramBatchArgs { GramArgs m[APGP_GRAM_BATCH_MAX]; };
static_assert(sizeof(GramBatchArgs) <= 4096, "kernel-argument segment");
template <int DPAD>
__global__ __launch_bounds__(256) void gram_batch_kernel(GramBatchArgs b) { gram_body<DPAD>(b.m[blockIdx.y]); }

// Gram matrix; with z != NULL also z = y - shift and *info_dev = "no failure yet" for the
// factorisation that follows (apgp_nll_eval: two launches fewer per evaluation)
int apgp_gram_with_rhs(const double* X, int64_t n, const apgp_kernel_t* kern, double* K, int64_t ldk,
                       const double* y, double shift, double* z, int32_t* info_dev, void* stream) {
    APGP_CHECK_ARG(X && K && kern, "null pointer");
    APGP_CHECK_ARG(n >= 1 && n <= APGP_MAX_N && ldk >= n, "n >= 1 and ldk >= n required");
    GramArgs a;
    APGP_CHECK_ARG(apgp_make_kernconst(kern, &a.kc) == 0, "kernel parameters");
    a.X = X; a.K = K; a.n = n; a.ldk = ldk;
    a.y = y; a.z = z; a.info = info_dev; a.shift = shift;
    const long long nb = (n + 63) / 64;
    dim3 grid((unsigned)(nb * (nb + 1) / 2)), block(256);
    hipStream_t s = (hipStream_t)stream;
    switch (a.kc.dpad) {
        case 2: hipLaunchKernelGGL(gram_kernel<2>, grid, block, 0, s, a); break;
        case 4: hipLaunchKernelGGL(gram_kernel<4>, grid, block, 0, s, a); break;
        case 8: hipLaunchKernelGGL(gram_kernel<8>, grid, block, 0, s, a); break;
        case 16: hipLaunchKernelGGL(gram_kernel<16>, grid, block, 0, s, a); break;
        default: hipLaunchKernelGGL(gram_kernel<32>, grid, block, 0, s, a); break;
    }
    APGP_CHECK_LAUNCH();
    return 0;
}

// the batched form: matrix b = K + b n^2 (ld n), right-hand side z + b n = y - shifts[b], info word info_dev + b
int apgp_gram_with_rhs_batch(const double* X, int64_t n, int64_t batch, const apgp_kernel_t* kerns, double* K,
                             const double* y, const double* shifts, double* z, int32_t* info_dev, void* stream) {
    APGP_CHECK_ARG(X && K && kerns, "null pointer");
    APGP_CHECK_ARG(n >= 1 && n <= APGP_MAX_N && batch >= 1 && batch <= APGP_GRAM_BATCH_MAX, "n >= 1 and 1 <= batch <= 6 required");
    GramBatchArgs g;
    for (int64_t b = 0; b < batch; ++b) {
        GramArgs& a = g.m[b];
        APGP_CHECK_ARG(apgp_make_kernconst(kerns + b, &a.kc) == 0, "kernel parameters");
        APGP_CHECK_ARG(a.kc.dpad == g.m[0].kc.dpad && a.kc.ndim == g.m[0].kc.ndim, "kernels of one batch share their dimension");
        a.X = X; a.K = K + b * n * n; a.n = n; a.ldk = n;
        a.y = y; a.z = z ? z + b * n : nullptr; a.info = info_dev ? info_dev + b : nullptr; a.shift = shifts ? shifts[b] : 0.0;
    }
    for (int64_t b = batch; b < APGP_GRAM_BATCH_MAX; ++b) g.m[b] = g.m[0];
    const long long nb = (n + 63) / 64;
    dim3 grid((unsigned)(nb * (nb + 1) / 2), (unsigned)batch), block(256);
    hipStream_t s = (hipStream_t)stream;
    switch (g.m[0].kc.dpad) {
        case 2: hipLaunchKernelGGL(gram_batch_kernel<2>, grid, block, 0, s, g); break;
        case 4: hipLaunchKernelGGL(gram_batch_kernel<4>, grid, block, 0, s, g); break;
        case 8: hipLaunchKernelGGL(gram_batch_kernel<8>, grid, block, 0, s, g); break;
        case 16: hipLaunchKernelGGL(gram_batch_kernel<16>, grid, block, 0, s, g); break;
        default: hipLaunchKernelGGL(gram_batch_kernel<32>, grid, block, 0, s, g); break;
    }
    APGP_CHECK_LAUNCH();
    return 0;
}

extern "C" int apgp_gram(const double* X, int64_t n, const apgp_kernel_t* kern, double* K,
                         int64_t ldk, void* stream) {
    return apgp_gram_with_rhs(X, n, kern, K, ldk, NULL, 0.0, NULL, NULL, stream);
}


// ---------------------------------------------------------------------------
// Cross kernel matrix  C_ij = amp * exp(-|xs1_i - xs2_j|^2)  (no diagonal term):
// george ``kernel.get_value(x1, x2)``.  Used by the incremental factor update
// (new training rows against the old ones).  One thread per output element.
// ---------------------------------------------------------------------------
struct CrossArgs {
    const double* X1;
    const double* X2;
    double* C;
    long long m, n, ldc;
    KernConst kc;
};

__global__ __launch_bounds__(256) void kernel_cross_kernel(CrossArgs a) {
    __shared__ double etab[APGP_EXP_TAB_N];
    apgp_exp_tab_load(etab);
    __syncthreads();
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= a.m * a.n) return;
    const long long i = e / a.n, j = e % a.n;
    double s = 0.0, s3 = 0.0;
    for (int d = 0; d < a.kc.dpad; d += 2) {
        double df0 = 0.0, df1 = 0.0;
        if (d < a.kc.ndim) df0 = a.X1[i * a.kc.ndim + d] * a.kc.sc[d] - a.X2[j * a.kc.ndim + d] * a.kc.sc[d];
        if (d + 1 < a.kc.ndim)
            df1 = a.X1[i * a.kc.ndim + d + 1] * a.kc.sc[d + 1] - a.X2[j * a.kc.ndim + d + 1] * a.kc.sc[d + 1];
        s = fma(df0, df0, s);
        s3 = fma(df1, df1, s3);
    }
    double k = a.kc.amp * apgp_exp(-(s + s3), etab);
    if (a.kc.lin_coef != 0.0) {
        double ls;
        APGP_LIN_SUM(ls, a.kc.ndim, a.kc.ndim, a.kc.lin_order,
                     a.X1[i * a.kc.ndim + d_] * a.X2[j * a.kc.ndim + d_]);
        k = fma(a.kc.lin_coef, ls, k);
    }
    a.C[i * a.ldc + j] = k;
}

extern "C" int apgp_kernel_cross(const double* X1, int64_t m, const double* X2, int64_t n,
                                 const apgp_kernel_t* kern, double* C, int64_t ldc, void* stream) {
    APGP_CHECK_ARG(X1 && X2 && C && kern, "null pointer");
    APGP_CHECK_ARG(m >= 1 && m <= APGP_MAX_M && n >= 1 && n <= APGP_MAX_N && ldc >= n, "m >= 1, n >= 1 and ldc >= n required");
    CrossArgs a;
    APGP_CHECK_ARG(apgp_make_kernconst(kern, &a.kc) == 0, "kernel parameters");
    a.X1 = X1; a.X2 = X2; a.C = C; a.m = m; a.n = n; a.ldc = ldc;
    hipLaunchKernelGGL(kernel_cross_kernel, dim3((unsigned)((m * n + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, a);
    APGP_CHECK_LAUNCH();
    return 0;
}
