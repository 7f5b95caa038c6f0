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


// ---------------------------------------------------------------------------
// K^-1 by the SOLVE route, for factors the explicit inverse must not be trusted on (condition estimate above the
// gate of gp.py, COND_SOLVE): what george's grad_log_likelihood does -- K^-1 = cho_solve(L, I) (gpUtils.py:110 ->
// george GP.grad_log_likelihood -> solver.apply_inverse(identity)) -- two triangular solves against the identity,
// never a product of inverses:
//   pass 0  X = L^-1 I   blocked forward substitution, block row j = 0 .. nb-1  (X is lower triangular)
//   pass 1  Y = L^-T X   blocked back substitution,    block row j = nb-1 .. 0  (lower-triangle tiles of K^-1)
// One launch per block row and pass, one workgroup per 64 x 64 tile (j, c), c <= j: the tile's right-hand side minus
// the already solved block rows (plain fp64 FMA on LDS-staged 64 x 16 chunks, every index bounds-checked -- this
// path is rare and not where the time goes), then the 64 x 64 triangular solve by substitution, one column per thread.
// ---------------------------------------------------------------------------
struct KinvSolveArgs {
    const double* L;
    double* X;        // np x np, ld np (pass 0 output)
    double* Y;        // n x n, ld n (pass 1 output: K^-1, lower tiles)
    long long n, ldl, np;
    int j;
};

template <int PASS>
__global__ __launch_bounds__(256) void kinv_solve_step_kernel(KinvSolveArgs a) {
    __shared__ double As[64][17], Bs[16][65], Rt[64][65], Lj[64][65];
    const int t = threadIdx.x;
    const long long j = a.j, c = blockIdx.x, n = a.n, np = a.np;
    const long long r0 = j * 64, c0 = c * 64;
    const int ty = t >> 4, tx = t & 15;                   // thread -> outputs rows 4 ty .. +3, columns 4 tx .. +3
    double acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[i][q] = 0.0;
    // sum over the solved block rows: pass 0: k in [c0, r0) of L[r0 + i][k] X[k][c0 + q];
    //                                pass 1: k in [r0 + 64, np) of L[k][r0 + i] Y[k][c0 + q]   (k < n only)
    const long long kb = PASS == 0 ? c0 : r0 + 64, ke = PASS == 0 ? r0 : np;
    for (long long kk = kb; kk < ke; kk += 16) {
        for (int e = t; e < 64 * 16; e += 256) {
            const int i = e >> 4, k = e & 15;
            const long long gk = kk + k, gi = r0 + i;
            double v = 0.0;
            if (PASS == 0) { if (gi < n && gk < n) v = a.L[gi * a.ldl + gk]; }
            else { if (gk < n && gi < n) v = a.L[gk * a.ldl + gi]; }
            As[i][k] = v;
        }
        for (int e = t; e < 16 * 64; e += 256) {
            const int k = e >> 6, q = e & 63;
            const long long gk = kk + k, gq = c0 + q;
            double v = 0.0;
            if (PASS == 0) v = a.X[gk * np + gq];
            else if (gk < n && gq < n) v = a.Y[gk * n + gq];
            Bs[k][q] = v;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[i][q] = fma(As[4 * ty + i][k], Bs[k][4 * tx + q], acc[i][q]);
        __syncthreads();
    }
    // right-hand side tile minus the sum; the diagonal block of L (identity rows past n)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int li = 4 * ty + i, lq = 4 * tx + q;
            double b;
            if (PASS == 0) b = (c == j && li == lq) ? 1.0 : 0.0;
            else b = a.X[(r0 + li) * np + c0 + lq];
            Rt[li][lq] = b - acc[i][q];
        }
    for (int e = t; e < 64 * 64; e += 256) {
        const int i = e >> 6, k = e & 63;
        const long long gi = r0 + i, gk = r0 + k;
        Lj[i][k] = (gi < n && gk < n && k <= i) ? a.L[gi * a.ldl + gk] : ((i == k) ? 1.0 : 0.0);
    }
    __syncthreads();
    if (t < 64) {
        double x[64];
#pragma unroll
        for (int i = 0; i < 64; ++i) x[i] = Rt[i][t];
        if (PASS == 0) {
#pragma unroll
            for (int i = 0; i < 64; ++i) {
                double v = x[i];
#pragma unroll
                for (int m = 0; m < i; ++m) v = fma(-Lj[i][m], x[m], v);
                x[i] = v / Lj[i][i];
            }
        } else {
#pragma unroll
            for (int i = 63; i >= 0; --i) {
                double v = x[i];
#pragma unroll
                for (int m = i + 1; m < 64; ++m) v = fma(-Lj[m][i], x[m], v);
                x[i] = v / Lj[i][i];
            }
        }
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            const long long gi = r0 + i, gq = c0 + t;
            if (PASS == 0) a.X[gi * np + gq] = (gi < n && gq < n) ? x[i] : 0.0;
            else if (gi < n && gq < n) a.Y[gi * n + gq] = x[i];
        }
    }
}

extern "C" int64_t apgp_kinv_solve_work_len(int64_t n) {
    if (n > APGP_MAX_N) return -1;
    const int64_t np = apgp_round_up(n < 1 ? 1 : n, 64);
    return np * np;
}

extern "C" int apgp_kinv_solve(const double* L, int64_t n, int64_t ldl, double* xwork, double* kinv, void* stream) {
    APGP_CHECK_ARG(L && xwork && kinv, "null pointer");
    APGP_CHECK_ARG(n >= 1 && n <= APGP_MAX_N && ldl >= n, "n >= 1 and ldl >= n required");
    hipStream_t s = (hipStream_t)stream;
    KinvSolveArgs a;
    a.L = L; a.X = xwork; a.Y = kinv; a.n = n; a.ldl = ldl; a.np = apgp_round_up(n, 64);
    const int nb = (int)(a.np / 64);
    if (hipMemsetAsync(xwork, 0, sizeof(double) * (size_t)a.np * (size_t)a.np, s) != hipSuccess) {
        apgp_set_error("apgp_kinv_solve: memset failed");
        return -2;
    }
    for (int j = 0; j < nb; ++j) {
        a.j = j;
        hipLaunchKernelGGL(kinv_solve_step_kernel<0>, dim3((unsigned)(j + 1)), dim3(256), 0, s, a);
    }
    for (int j = nb - 1; j >= 0; --j) {
        a.j = j;
        hipLaunchKernelGGL(kinv_solve_step_kernel<1>, dim3((unsigned)(j + 1)), dim3(256), 0, s, a);
    }
    APGP_CHECK_LAUNCH();
    return 0;
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
    if (n < 0 || n > APGP_MAX_N) return -1;
    int64_t nb = (n + 63) / 64;
    return n * n + nb * nb * (2 + APGP_MAX_DIM);
}

extern "C" int apgp_grad_loglik(const double* X, const double* alpha, const double* winv, int64_t ldw,
                                int64_t n, const apgp_kernel_t* kern, double* work, double* out,
                                void* stream) {
    APGP_CHECK_ARG(X && alpha && kern && work && out, "null pointer");
    APGP_CHECK_ARG(n >= 1 && n <= APGP_MAX_N, "n >= 1 required");
    const long long np = apgp_round_up(n, 64);
    // winv == NULL: `work` already holds K^-1 (its lower tiles), formed by apgp_kinv_solve (the solve route)
    APGP_CHECK_ARG(!winv || ldw >= np, "winv must be the padded dense inverse left in apgp_trtri_pack's work buffer");
    GradArgs g;
    APGP_CHECK_ARG(apgp_make_kernconst(kern, &g.kc) == 0, "kernel parameters");
    hipStream_t s = (hipStream_t)stream;
    const unsigned nb = (unsigned)(np / 64);
    SyrkArgs sa;
    sa.W = winv; sa.Kinv = work; sa.ldw = ldw; sa.np = np; sa.n = n;
    const unsigned nlow = nb * (nb + 1) / 2;
    if (winv) hipLaunchKernelGGL(syrk_wtw_kernel, dim3(nlow), dim3(256), 0, s, sa);
    g.X = X; g.alpha = alpha; g.Kinv = work; g.partial = work + n * n; g.n = n;
    const int pw = 2 + g.kc.dpad;
    switch (g.kc.dpad) {
        case 2: hipLaunchKernelGGL(grad_tile_kernel<2>, dim3(nlow), dim3(256), 0, s, g); break;
        case 4: hipLaunchKernelGGL(grad_tile_kernel<4>, dim3(nlow), dim3(256), 0, s, g); break;
        case 8: hipLaunchKernelGGL(grad_tile_kernel<8>, dim3(nlow), dim3(256), 0, s, g); break;
        case 16: hipLaunchKernelGGL(grad_tile_kernel<16>, dim3(nlow), dim3(256), 0, s, g); break;
        default: hipLaunchKernelGGL(grad_tile_kernel<32>, dim3(nlow), dim3(256), 0, s, g); break;
    }
    hipLaunchKernelGGL(grad_final_kernel, dim3(1), dim3(1024), 0, s, (const double*)g.partial,
                       (long long)nlow, pw, alpha, (const double*)work, (long long)n, g.kc.ndim, out);
    APGP_CHECK_LAUNCH();
    return 0;
}
