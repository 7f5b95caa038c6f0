// Shared helpers for the libapgp.so HIP sources (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/apgp.h"

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

void apgp_set_error(const char* fmt, ...);

#define APGP_CHECK_ARG(cond, msg)                                   \
    do {                                                            \
        if (!(cond)) {                                              \
            apgp_set_error("%s: bad argument: %s", __func__, msg);  \
            return -1;                                              \
        }                                                           \
    } while (0)

// Sizes the entry points accept: beyond them nothing fits a device anyway, and the size arithmetic (n * n, tile counts,
// grid dimensions) stays inside int64 / unsigned -- found by the sanitizer build (tools/run_asan.sh: apgp_grad_work_len(2^40)
// overflowed).  The *_len functions return -1 above the limits.
#define APGP_MAX_N (1ll << 24)         /* training points */
#define APGP_MAX_M (1ll << 40)         /* candidates per call */

#define APGP_CHECK_LAUNCH()                                                       \
    do {                                                                          \
        hipError_t e__ = hipGetLastError();                                       \
        if (e__ != hipSuccess) {                                                  \
            apgp_set_error("%s: HIP error: %s", __func__, hipGetErrorString(e__)); \
            return -2;                                                            \
        }                                                                         \
    } while (0)

// Feature dimension padded to a template-friendly width (zero scale => zero
// contribution of the padded coordinates).
static inline int apgp_dpad(int d) { return d <= 2 ? 2 : d <= 4 ? 4 : d <= 8 ? 8 : d <= 16 ? 16 : 32; }
// doubles per row of the packed training stream: scaled x (Dpad) | alpha | 0
static inline int apgp_xs_stride(int d) { return apgp_dpad(d) + 2; }

static inline int64_t apgp_round_up(int64_t n, int64_t b) { return (n + b - 1) / b * b; }

// Scaled-coordinate / amplitude constants shared by every kernel that
// evaluates the squared-exponential kernel, so that K (Gram), K* (sweep) and
// dK (gradient) use bit-identical expressions:
//   k(x,x') = amp * exp(-sum_d (xs_d - xs'_d)^2),  xs = x * sqrt(inv_metric/2)
// and, when lin_coef != 0, the linear-regression term in the same scaled coordinates:
//   k_lin(x,x') = lin_coef * sum_d (xs_d xs'_d lw_d)^P,  lw_d = 1 / sc_d^2
struct KernConst {
    double sc[APGP_MAX_DIM];
    double lw[APGP_MAX_DIM];
    double log_amp;
    double amp;
    double diag_add;
    double lin_coef;
    int lin_order;
    int ndim;
    int dpad;
};

// sum_d (p_d)^P over the real dimensions, p_d = xs_d xs'_d lw_d (padded dimensions carry
// xs = 0: their product is 0 for P >= 1; P = 0 counts ndim ones as george does)
#define APGP_LIN_SUM(acc, DP, ndim, order, prod_d)                         \
    do {                                                                   \
        acc = 0.0;                                                         \
        if ((order) == 0) acc = (double)(ndim);                            \
        else                                                               \
            for (int d_ = 0; d_ < (DP); ++d_) {                            \
                const double p_ = (prod_d);                                \
                double q_ = p_;                                            \
                for (int e_ = 1; e_ < (order); ++e_) q_ *= p_;             \
                acc += q_;                                                 \
            }                                                              \
    } while (0)

static inline int apgp_make_kernconst(const apgp_kernel_t* k, KernConst* c) {
    if (!k || k->ndim < 1 || k->ndim > APGP_MAX_DIM) return -1;
    if (!(k->amp > 0.0)) return -1;
    c->ndim = k->ndim;
    c->dpad = apgp_dpad(k->ndim);
    c->amp = k->amp;
    c->log_amp = log(k->amp);
    c->diag_add = k->diag_add;
    if (!(k->lin_coef >= 0.0) || k->lin_order < 0 || k->lin_order > 16) return -1;
    c->lin_coef = k->lin_coef;
    c->lin_order = k->lin_order;
    for (int d = 0; d < APGP_MAX_DIM; ++d) {
        c->sc[d] = d < k->ndim ? sqrt(0.5 * k->inv_metric[d]) : 0.0;
        c->lw[d] = d < k->ndim ? 2.0 / k->inv_metric[d] : 0.0;
    }
    return 0;
}

// ---------------------------------------------------------------------------
// exp() for the squared-exponential kernel values.  Every kernel that evaluates
// k(x,x') (Gram, sweep, mean-only predict, gradient) uses THIS routine, so the
// training Gram matrix and the candidate rows are produced by bit-identical
// arithmetic.  Table-driven (32 entries of 2^(i/32) in LDS, degree-6 polynomial
// on |r| <= ln2/64), ~1 ulp; 12 full-rate fp64 ops and no quarter-rate
// convert/ldexp/compare instructions, which matters because on MI355X fp64 VALU
// work shares the DP pipes with the fp64 MFMA.
// Domain: the argument is clamped to [-700, 700]; below -700 the true value is
// < 1e-304 and is returned as ~1e-304 (never a denormal), far below any
// tolerance of the path.
// ---------------------------------------------------------------------------
#define APGP_EXP_TAB_N 32
__device__ static const double apgp_exp_tab_init[APGP_EXP_TAB_N] = {
    0x1.0000000000000p+0,
    0x1.059b0d3158574p+0,
    0x1.0b5586cf9890fp+0,
    0x1.11301d0125b51p+0,
    0x1.172b83c7d517bp+0,
    0x1.1d4873168b9aap+0,
    0x1.2387a6e756238p+0,
    0x1.29e9df51fdee1p+0,
    0x1.306fe0a31b715p+0,
    0x1.371a7373aa9cbp+0,
    0x1.3dea64c123422p+0,
    0x1.44e086061892dp+0,
    0x1.4bfdad5362a27p+0,
    0x1.5342b569d4f82p+0,
    0x1.5ab07dd485429p+0,
    0x1.6247eb03a5585p+0,
    0x1.6a09e667f3bcdp+0,
    0x1.71f75e8ec5f74p+0,
    0x1.7a11473eb0187p+0,
    0x1.82589994cce13p+0,
    0x1.8ace5422aa0dbp+0,
    0x1.93737b0cdc5e5p+0,
    0x1.9c49182a3f090p+0,
    0x1.a5503b23e255dp+0,
    0x1.ae89f995ad3adp+0,
    0x1.b7f76f2fb5e47p+0,
    0x1.c199bdd85529cp+0,
    0x1.cb720dcef9069p+0,
    0x1.d5818dcfba487p+0,
    0x1.dfc97337b9b5fp+0,
    0x1.ea4afa2a490dap+0,
    0x1.f50765b6e4540p+0};

// Copy the table into LDS (call from every thread of the workgroup, then sync).
__device__ __forceinline__ void apgp_exp_tab_load(double* tab_lds) {
    if (threadIdx.x < APGP_EXP_TAB_N) tab_lds[threadIdx.x] = apgp_exp_tab_init[threadIdx.x];
}

// (Every caller passes x <= 0, so the upper clamp is dead arithmetic -- and it stays: round 6 measured the sweep without it,
// same registers, no spill, 313 against 252 ms per C3 call on one box (profiles/r06i_exp_variants_ab.txt): hipcc schedules
// the four interleaved chains of apgp_exp4 differently once the v_min is gone and the feeders' fp64 operations land
// between the matrix wavefronts' MFMAs.  A 64-entry table with a degree-5 polynomial: 251 ms, i.e. nothing.)
__device__ __forceinline__ double apgp_exp(double x, const double* tab_lds) {
    x = fmin(fmax(x, -700.0), 700.0);
    const double magic = 6755399441055744.0;                       // 1.5 * 2^52
    const double t = fma(x, 0x1.71547652b82fep+5 /* 32/ln2 */, magic);
    const double kf = t - magic;                                   // round(x * 32/ln2)
    const int j = __double2loint(t);
    double r = fma(kf, -0x1.62e42fefa39efp-6 /* ln2/32 hi */, x);
    r = fma(kf, -0x1.abc9e3b39803fp-61 /* ln2/32 lo */, r);
    double p = fma(r, 1.0 / 720.0, 1.0 / 120.0);
    p = fma(r, p, 1.0 / 24.0);
    p = fma(r, p, 1.0 / 6.0);
    p = fma(r, p, 0.5);
    p = fma(r * r, p, r);                                          // expm1(r)
    const double T = tab_lds[j & (APGP_EXP_TAB_N - 1)];
    const double res = fma(T, p, T);                               // 2^(j&31)/32 * e^r in [1,2)
    const int hi = __double2hiint(res) + ((j >> 5) << 20);         // * 2^(j>>5)
    return __hiloint2double(hi, __double2loint(res));
}

// Four independent exponentials, written operation-by-operation across the four
// values so that the four dependent fp64 chains interleave (one chain alone is
// latency-bound: ~11 cycles per dependent op against a 4.35-cycle issue slot).
__device__ __forceinline__ void apgp_exp4(const double (&xin)[4], double (&out)[4],
                                          const double* tab_lds) {
    const double magic = 6755399441055744.0;
    double x[4], t[4], kf[4], r[4], p[4], T[4], res[4];
    int j[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = fmin(fmax(xin[i], -700.0), 700.0);
#pragma unroll
    for (int i = 0; i < 4; ++i) t[i] = fma(x[i], 0x1.71547652b82fep+5, magic);
#pragma unroll
    for (int i = 0; i < 4; ++i) { kf[i] = t[i] - magic; j[i] = __double2loint(t[i]); }
#pragma unroll
    for (int i = 0; i < 4; ++i) T[i] = tab_lds[j[i] & (APGP_EXP_TAB_N - 1)];
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = fma(kf[i], -0x1.62e42fefa39efp-6, x[i]);
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = fma(kf[i], -0x1.abc9e3b39803fp-61, r[i]);
#pragma unroll
    for (int i = 0; i < 4; ++i) p[i] = fma(r[i], 1.0 / 720.0, 1.0 / 120.0);
#pragma unroll
    for (int i = 0; i < 4; ++i) p[i] = fma(r[i], p[i], 1.0 / 24.0);
#pragma unroll
    for (int i = 0; i < 4; ++i) p[i] = fma(r[i], p[i], 1.0 / 6.0);
#pragma unroll
    for (int i = 0; i < 4; ++i) p[i] = fma(r[i], p[i], 0.5);
#pragma unroll
    for (int i = 0; i < 4; ++i) p[i] = fma(r[i] * r[i], p[i], r[i]);
#pragma unroll
    for (int i = 0; i < 4; ++i) res[i] = fma(T[i], p[i], T[i]);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int hi = __double2hiint(res[i]) + ((j[i] >> 5) << 20);
        out[i] = __hiloint2double(hi, __double2loint(res[i]));
    }
}

// ---------------------------------------------------------------------------
// One Gram-matrix entry K(row, col) from the scaled coordinates of the two points (xi = row point,
// xc = column point; xi may live in LDS).  Shared by gram_kernel and the fused small-N _nll kernel
// (potrf.hip) so that both produce the same bits: explicit fma()s, contraction of the remaining
// operations switched off (with -ffp-contract=fast hipcc would otherwise be free to fuse
// amp * e + diag_add in one kernel and not in the other).
// ---------------------------------------------------------------------------
template <int DPAD, class XI>
__device__ __forceinline__ double apgp_gram_value(const XI& xi, const double (&xc)[DPAD], const KernConst& kc,
                                                  bool diagonal, const double* etab) {
#pragma clang fp contract(off)
    double s = 0.0, s3 = 0.0;
#pragma unroll
    for (int d = 0; d < DPAD; d += 2) {
        const double df0 = xi[d] - xc[d];
        const double df1 = xi[d + 1] - xc[d + 1];
        s = fma(df0, df0, s);
        s3 = fma(df1, df1, s3);
    }
    double k = kc.amp * apgp_exp(-(s + s3), etab);
    if (kc.lin_coef != 0.0) {
        double ls;
        APGP_LIN_SUM(ls, DPAD, kc.ndim, kc.lin_order, (xi[d_] * xc[d_]) * kc.lw[d_]);
        k = fma(kc.lin_coef, ls, k);
    }
    if (diagonal) k = k + kc.diag_add;
    return k;
}
