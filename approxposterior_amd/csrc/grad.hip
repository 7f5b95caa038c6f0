// K4 "grad": gradient of the GP log-likelihood with respect to the kernel
// hyper-parameters, george GP.grad_log_likelihood semantics
// (gpUtils._grad_nll, gpUtils.py:83-111):
//   g_mean = sum(alpha)
//   g_k    = 0.5 * sum_ij (alpha_i alpha_j - Kinv_ij) dK_ij/dtheta_k
//   dK/dlog_constant = amp * k_se           (white noise excluded)
//   dK/dlog_M_d      = K * 0.5 * dx_d^2 / M_d
//   g_white_noise    = 0.5 * exp(white_noise) * trace(alpha alpha^T - Kinv)   (fit_white_noise)
// K^-1 = W^T W (W = L^-1) is formed with an MFMA-f64 GEMM (four-block instruction, apgp_mma16); K and dK are
// regenerated in registers from X (no N x N x P tensor in HBM).
#include "apgp_common.h"
#include "mma16.h"

// Kinv[i][j] = sum_{k >= max(i,j)} W[k][i] W[k][j]   (W lower triangular); tiles with bi >= bj only
struct SyrkArgs {
    const double* W;
    double* Kinv;
    long long ldw, np, n;
};

__global__ __launch_bounds__(256) void syrk_wtw_kernel(SyrkArgs a) {
    __shared__ __attribute__((aligned(16))) double lds[GEMM64_LDS_DOUBLES];
    // lower-triangle tiles only (Kinv is symmetric and grad_tile_kernel reads the lower tiles),
    // row block by row block: the k range of tile (bi, bj) is [64 bi, np), so the longest come first
    const long long tix = blockIdx.x;
    long long bi = (long long)((sqrt(8.0 * (double)tix + 1.0) - 1.0) * 0.5);
    while ((bi + 1) * (bi + 2) / 2 <= tix) ++bi;
    while (bi * (bi + 1) / 2 > tix) --bi;
    const long long bj = tix - bi * (bi + 1) / 2;
    const long long i0 = bi * 64, j0 = bj * 64;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int wr = (w >> 1) * 32, wc = (w & 1) * 32;
    double acc[2][2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.0;
    apgp_gemm64_tile<true, true>(a.W + i0, a.ldw, 0, a.W + j0, a.ldw, 0, i0, a.np, lds, acc);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                long long gr = i0 + wr + 16 * i + apgp_mma16_row(lane);
                long long gc = j0 + wc + 16 * j + apgp_mma16_col(lane, r);
                if (gr < a.n && gc < a.n) a.Kinv[gr * a.n + gc] = acc[i][j][r];
            }
}

struct GradArgs {
    const double* X;
    const double* alpha;
    const double* Kinv;
    double* partial;     // nblocks x (2 + DPAD): amp | metric d ... | linear term
    long long n;
    KernConst kc;
};

template <int DPAD>
__global__ __launch_bounds__(256) void grad_tile_kernel(GradArgs a) {
    __shared__ double xi[64][DPAD + 1];
    __shared__ double xj[64][DPAD + 1];
    __shared__ double ai[64], aj[64];
    __shared__ double red[4][2 + DPAD];
    __shared__ double etab[APGP_EXP_TAB_N];
    apgp_exp_tab_load(etab);
    const int t = threadIdx.x;
    // lower-triangle tiles (bi >= bj); the summand is symmetric, so an off-diagonal tile counts twice
    const long long tix = blockIdx.x;
    long long bi = (long long)((sqrt(8.0 * (double)tix + 1.0) - 1.0) * 0.5);
    while ((bi + 1) * (bi + 2) / 2 <= tix) ++bi;
    while (bi * (bi + 1) / 2 > tix) --bi;
    const long long bj = tix - bi * (bi + 1) / 2;
    const long long i0 = bi * 64, j0 = bj * 64;
    const double tile_weight = bi == bj ? 1.0 : 2.0;
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
    if (t < 64) {
        ai[t] = (i0 + t < a.n) ? a.alpha[i0 + t] : 0.0;
        aj[t] = (j0 + t < a.n) ? a.alpha[j0 + t] : 0.0;
    }
    __syncthreads();
    const int c = t & 63, g = t >> 6;
    double xc[DPAD];
#pragma unroll
    for (int d = 0; d < DPAD; ++d) xc[d] = xj[c][d];
    const long long gj = j0 + c;
    double gsum[2 + DPAD];
#pragma unroll
    for (int p = 0; p < 2 + DPAD; ++p) gsum[p] = 0.0;
    for (int q = 0; q < 16; ++q) {
        const int r = g + 4 * q;
        const long long gi = i0 + r;
        if (gi < a.n && gj < a.n) {
            double df2[DPAD];
            double s = 0.0, s3 = 0.0;
#pragma unroll
            for (int d = 0; d < DPAD; d += 2) {
                double df0 = xi[r][d] - xc[d];
                double df1 = xi[r][d + 1] - xc[d + 1];
                df2[d] = df0 * df0;
                df2[d + 1] = df1 * df1;
                s = fma(df0, df0, s);
                s3 = fma(df1, df1, s3);
            }
            const double k = a.kc.amp * apgp_exp(-(s + s3), etab);
            const double A = ai[r] * aj[c] - a.Kinv[gi * a.n + gj];
            const double Ak = A * k;
            gsum[0] += Ak;
#pragma unroll
            for (int d = 0; d < DPAD; ++d) gsum[1 + d] = fma(Ak, df2[d], gsum[1 + d]);
            if (a.kc.lin_coef != 0.0) {
                // d/d log_constant2 = k_lin, d/d log_gamma2 = -k_lin (the host applies the sign)
                double ls;
                APGP_LIN_SUM(ls, DPAD, a.kc.ndim, a.kc.lin_order, xi[r][d_] * xc[d_] * a.kc.lw[d_]);
                gsum[1 + DPAD] = fma(A * a.kc.lin_coef, ls, gsum[1 + DPAD]);
            }
        }
    }
#pragma unroll
    for (int p = 0; p < 2 + DPAD; ++p) {
        double v = gsum[p];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if ((t & 63) == 0) red[g][p] = v;
    }
    __syncthreads();
    if (t < 2 + DPAD) {
        double v = red[0][t] + red[1][t] + red[2][t] + red[3][t];
        a.partial[tix * (2 + DPAD) + t] = tile_weight * v;
    }
}

// out[0] = sum alpha ; out[1] = 0.5 * sum A K ; out[2+d] = 0.5 * sum A K dx_d^2 w_d / 2 ;
// out[2 + APGP_MAX_DIM] = 0.5 * trace(A) ; out[3 + APGP_MAX_DIM] = 0.5 * sum A K_lin
__global__ __launch_bounds__(1024) void grad_final_kernel(const double* partial, long long nblk, int pw,
                                                          const double* alpha, const double* Kinv,
                                                          long long n, int ndim, double* out) {
    __shared__ double red[16];
    const int t = threadIdx.x;
    for (int p = 0; p <= pw + 1; ++p) {   // p == pw: sum(alpha); p == pw + 1: trace(A)
        double v = 0.0;
        if (p < pw) for (long long b = t; b < nblk; b += 1024) v += partial[b * pw + p];
        else if (p == pw) for (long long i = t; i < n; i += 1024) v += alpha[i];
        else for (long long i = t; i < n; i += 1024) v += alpha[i] * alpha[i] - Kinv[i * n + i];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if ((t & 63) == 0) red[t >> 6] = v;
        __syncthreads();
        if (t == 0) {
            double s = 0.0;
            for (int i = 0; i < 16; ++i) s += red[i];
            if (p == pw) out[0] = s;
            else if (p == pw + 1) out[2 + APGP_MAX_DIM] = 0.5 * s;
            else if (p == 0) out[1] = 0.5 * s;
            else if (p == pw - 1) out[3 + APGP_MAX_DIM] = 0.5 * s;
            else if (p - 1 < ndim) out[2 + (p - 1)] = 0.5 * s;
        }
        __syncthreads();
    }
}

extern "C" int64_t apgp_grad_work_len(int64_t n) {
    int64_t nb = (n + 63) / 64;
    return n * n + nb * nb * (2 + APGP_MAX_DIM);
}

extern "C" int apgp_grad_loglik(const double* X, const double* alpha, const double* winv, int64_t ldw,
                                int64_t n, const apgp_kernel_t* kern, double* work, double* out,
                                void* stream) {
    APGP_CHECK_ARG(X && alpha && winv && kern && work && out, "null pointer");
    APGP_CHECK_ARG(n >= 1, "n >= 1 required");
    const long long np = apgp_round_up(n, 64);
    APGP_CHECK_ARG(ldw >= np, "winv must be the padded dense inverse left in apgp_trtri_pack's work buffer");
    GradArgs g;
    APGP_CHECK_ARG(apgp_make_kernconst(kern, &g.kc) == 0, "kernel parameters");
    hipStream_t s = (hipStream_t)stream;
    const unsigned nb = (unsigned)(np / 64);
    SyrkArgs sa;
    sa.W = winv; sa.Kinv = work; sa.ldw = ldw; sa.np = np; sa.n = n;
    const unsigned nlow = nb * (nb + 1) / 2;
    hipLaunchKernelGGL(syrk_wtw_kernel, dim3(nlow), dim3(256), 0, s, sa);
    g.X = X; g.alpha = alpha; g.Kinv = work; g.partial = work + n * n; g.n = n;
    const int pw = 2 + g.kc.dpad;
    switch (g.kc.dpad) {
        case 2: hipLaunchKernelGGL(grad_tile_kernel<2>, dim3(nlow), dim3(256), 0, s, g); break;
        case 4: hipLaunchKernelGGL(grad_tile_kernel<4>, dim3(nlow), dim3(256), 0, s, g); break;
        case 8: hipLaunchKernelGGL(grad_tile_kernel<8>, dim3(nlow), dim3(256), 0, s, g); break;
        default: hipLaunchKernelGGL(grad_tile_kernel<16>, dim3(nlow), dim3(256), 0, s, g); break;
    }
    hipLaunchKernelGGL(grad_final_kernel, dim3(1), dim3(1024), 0, s, (const double*)g.partial,
                       (long long)nlow, pw, alpha, (const double*)work, (long long)n, g.kc.ndim, out);
    APGP_CHECK_LAUNCH();
    return 0;
}
