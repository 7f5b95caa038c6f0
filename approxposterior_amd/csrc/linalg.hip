// K2/K3 + factor-side helpers: log-determinant, triangular solves, blocked
// triangular inversion (MFMA-f64 GEMM merges) and the packed layouts the sweep
// kernel streams.  Reference semantics: george BasicSolver.compute /
// apply_inverse (scipy cholesky + cho_solve) as used by GP.log_likelihood,
// GP._compute_alpha and GP.predict (gpUtils.py:78; utility.py:131,178,224).
#include "apgp_common.h"
#include "mma16.h"
#include "scratch.h"
#include <atomic>
#include <chrono>
#include <mutex>
#include <type_traits>
#include <utility>
template <int... Is, class F>
__device__ __forceinline__ void trtri_static_for_impl(std::integer_sequence<int, Is...>, F&& f) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void trtri_static_for(F&& f) {
    trtri_static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

// ---------------------------------------------------------------------------
// sizes
// ---------------------------------------------------------------------------
extern "C" int64_t apgp_npad(int64_t n) { return (n < 0 || n > APGP_MAX_N) ? -1 : apgp_round_up(n, APGP_ROW_BLOCK); }

extern "C" int64_t apgp_packed_linv_len(int64_t n) {
    if (n < 0 || n > APGP_MAX_N) return -1;
    int64_t nrb = apgp_npad(n) / APGP_ROW_BLOCK;
    // row block ib holds (ib+1)*CPB tiles of ROW_BLOCK x K_CHUNK doubles
    const int64_t cpb = APGP_ROW_BLOCK / APGP_K_CHUNK;
    return cpb * nrb * (nrb + 1) / 2 * (int64_t)(APGP_ROW_BLOCK * APGP_K_CHUNK);
}

extern "C" int64_t apgp_packed_train_len(int64_t n, int32_t ndim) {
    if (n < 0 || n > APGP_MAX_N || ndim < 1 || ndim > APGP_MAX_DIM) return -1;
    return apgp_npad(n) * apgp_xs_stride(ndim);
}

extern "C" int64_t apgp_trtri_work_len(int64_t n) {
    if (n < 0 || n > APGP_MAX_N) return -1;
    int64_t np = apgp_round_up(n, 64);
    return 2 * np * np;
}

extern "C" int apgp_release_scratch(void* stream) {
    return apgp_stream_scratch_release((hipStream_t)stream);
}

// ---------------------------------------------------------------------------
// K2: logdet + diagonal range.  One workgroup; the diagonal is N doubles.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void logdet_kernel(const double* L, long long n, long long ldl,
                                                      const double* z, const int* info, double* out) {
    __shared__ double ssum[16], smin[16], smax[16], szz[16];
    double s = 0.0, mn = INFINITY, mx = -INFINITY, zz = 0.0;
    for (long long i = threadIdx.x; i < n; i += 1024) {
        double d = L[i * ldl + i];
        s += log(d);
        mn = fmin(mn, d);
        mx = fmax(mx, d);
        if (z) zz = fma(z[i], z[i], zz);
    }
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_xor(s, o);
        zz += __shfl_xor(zz, o);
        mn = fmin(mn, __shfl_xor(mn, o));
        mx = fmax(mx, __shfl_xor(mx, o));
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { ssum[w] = s; smin[w] = mn; smax[w] = mx; szz[w] = zz; }
    __syncthreads();
    if (threadIdx.x == 0) {
        s = 0.0; zz = 0.0; mn = INFINITY; mx = -INFINITY;
        for (int i = 0; i < 16; ++i) { s += ssum[i]; zz += szz[i]; mn = fmin(mn, smin[i]); mx = fmax(mx, smax[i]); }
        out[0] = 2.0 * s;
        out[1] = mn;
        out[2] = mx;
        out[3] = zz;
        out[4] = info ? (double)(*info) : 0.0;
    }
}

extern "C" int apgp_logdet(const double* L, int64_t n, int64_t ldl, double* out3, void* stream) {
    APGP_CHECK_ARG(L && out3, "null pointer");
    APGP_CHECK_ARG(n >= 1 && n <= APGP_MAX_N && ldl >= n, "n >= 1 and ldl >= n required");
    // legacy 3-value form: the kernel writes 5 doubles, callers of this entry pass >= 5
    hipLaunchKernelGGL(logdet_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, L, (long long)n,
                       (long long)ldl, (const double*)NULL, (const int*)NULL, out3);
    APGP_CHECK_LAUNCH();
    return 0;
}

extern "C" int apgp_fit_summary(const double* L, int64_t n, int64_t ldl, const double* z,
                                const int32_t* info_dev, double* out5, void* stream) {
    APGP_CHECK_ARG(L && out5, "null pointer");
    APGP_CHECK_ARG(n >= 1 && n <= APGP_MAX_N && ldl >= n, "n >= 1 and ldl >= n required");
    hipLaunchKernelGGL(logdet_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, L, (long long)n,
                       (long long)ldl, z, (const int*)info_dev, out5);
    APGP_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------
// K3: triangular solve with the lower Cholesky factor, one workgroup.
// The right-hand side lives in LDS for the whole solve; the factor's lower
// triangle is streamed once (8 * N^2/2 bytes).
// ---------------------------------------------------------------------------
#define TRSV_B 64
struct TrsvArgs {
    const double* L;
    const double* b;
    double* x;
    double* sumsq;
    long long n, ldl;
    double shift;
    int trans;
};

__global__ __launch_bounds__(1024) void trsv_kernel(TrsvArgs a) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* r = smem;                                   // n (rounded up to 64)
    const long long n = a.n;
    const long long nr = (n + 63) / 64 * 64;
    double (*Lb)[TRSV_B + 1] = (double (*)[TRSV_B + 1])(smem + nr);   // 64 x 65
    double* red = smem + nr + TRSV_B * (TRSV_B + 1);    // 16 x 64 partials / 16 sums
    const int t = threadIdx.x;
    const int lane = t & 63;
    for (long long i = t; i < nr; i += 1024) r[i] = i < n ? a.b[i] - a.shift : 0.0;
    __syncthreads();
    const long long nb = (n + TRSV_B - 1) / TRSV_B;

    if (!a.trans) {
        // forward, left-looking: r_blk -= L[blk, 0:j0] z[0:j0]; then 64x64 solve
        for (long long jb = 0; jb < nb; ++jb) {
            const long long j0 = jb * TRSV_B;
            const int row = t >> 4, part = t & 15;       // 64 rows x 16 threads
            const long long gi = j0 + row;
            double acc = 0.0;
            if (gi < n) {
                const double* lrow = a.L + gi * a.ldl;
                for (long long k = part * 2; k < j0; k += 32) {
                    acc = fma(lrow[k], r[k], acc);
                    acc = fma(lrow[k + 1], r[k + 1], acc);
                }
            }
            acc += __shfl_xor(acc, 8);
            acc += __shfl_xor(acc, 4);
            acc += __shfl_xor(acc, 2);
            acc += __shfl_xor(acc, 1);
            // stage the diagonal block
            for (int e = t; e < TRSV_B * TRSV_B; e += 1024) {
                int i = e >> 6, k = e & 63;
                long long gr = j0 + i, gc = j0 + k;
                double v = (i == k) ? 1.0 : 0.0;
                if (gr < n && gc < n && k <= i) v = a.L[gr * a.ldl + gc];
                Lb[i][k] = v;
            }
            __syncthreads();   // all reads of r[0:j0] done, Lb staged
            if (part == 0 && gi < n) r[gi] -= acc;
            __syncthreads();
            if (t < 64) {
                double ri = r[j0 + lane];
                for (int k = 0; k < TRSV_B; ++k) {
                    double zk = __shfl(ri, k) / Lb[k][k];
                    if (lane == k) ri = zk;
                    else if (lane > k) ri = fma(-Lb[lane][k], zk, ri);
                }
                r[j0 + lane] = ri;
            }
            __syncthreads();
        }
    } else {
        // backward (L^T x = r), right-looking over row blocks from the bottom
        for (long long jb = nb - 1; jb >= 0; --jb) {
            const long long j0 = jb * TRSV_B;
            for (int e = t; e < TRSV_B * TRSV_B; e += 1024) {
                int i = e >> 6, k = e & 63;
                long long gr = j0 + i, gc = j0 + k;
                double v = (i == k) ? 1.0 : 0.0;
                if (gr < n && gc < n && k <= i) v = a.L[gr * a.ldl + gc];
                Lb[i][k] = v;
            }
            __syncthreads();
            if (t < 64) {
                double ri = r[j0 + lane];
                for (int k = TRSV_B - 1; k >= 0; --k) {
                    double xk = __shfl(ri, k) / Lb[k][k];
                    if (lane == k) ri = xk;
                    else if (lane < k) ri = fma(-Lb[k][lane], xk, ri);   // (L^T)[lane][k] = L[k][lane]
                }
                r[j0 + lane] = ri;
            }
            __syncthreads();
            // r[c] -= sum_k L[j0+k, c] x_k  for c < j0  (coalesced along c)
            const long long kmax = (n - j0) < TRSV_B ? (n - j0) : TRSV_B;
            for (long long c = t; c < j0; c += 1024) {
                double acc = 0.0;
                const double* lcol = a.L + j0 * a.ldl + c;
#pragma unroll 8
                for (long long k = 0; k < kmax; ++k) acc = fma(lcol[k * a.ldl], r[j0 + k], acc);
                r[c] -= acc;
            }
            __syncthreads();
        }
    }
    double ss = 0.0;
    for (long long i = t; i < n; i += 1024) {
        double v = r[i];
        a.x[i] = v;
        ss = fma(v, v, ss);
    }
    if (a.sumsq) {
        for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
        if (lane == 0) red[t >> 6] = ss;
        __syncthreads();
        if (t == 0) {
            double s = 0.0;
            for (int i = 0; i < 16; ++i) s += red[i];
            *a.sumsq = s;
        }
    }
}

// ---------------------------------------------------------------------------
// K3, blocked: the same solves as a sequence of small launches, four 64-row blocks per launch, for N >= 256.
// Launch jb: every workgroup (four wavefronts) solves the 64 x 64 diagonal block against the current
// right-hand side of block jb itself (redundantly: 1 us, and no workgroup waits for another) and
// applies the result to ITS block of the remaining rows (forward: block rows below, a wavefront-wide sum
// per row; transposed: block columns to the left, coalesced as they lie).  Workgroup 0 keeps
// the solved block.  The working right-hand side lives in stream-ordered scratch (scratch.h): the
// solution goes to x, so no workgroup overwrites what a sibling still has to read.
// 64 launches of ~4 us at N = 4096 instead of one single-workgroup kernel of 1.6 ms.
// ---------------------------------------------------------------------------
struct TrsvStepArgs {
    const double* L;
    double* r;          // working right-hand side (scratch)
    double* x;          // solution
    double* sumsq;      // x.x accumulated block by block (or NULL)
    long long n, ldl, j0;
};

__global__ __launch_bounds__(256) void trsv_init_kernel(const double* b, double shift, double* r, double* sumsq, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) r[i] = b[i] - shift;
    if (sumsq && i == 0) *sumsq = 0.0;
}

__device__ __forceinline__ double trsv_bcast(double v, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

template <int TRANS>
__global__ __launch_bounds__(256) void trsv_step_kernel(TrsvStepArgs a) {
    // four wavefronts: all of them request their 16 rows of the diagonal block AND of the workgroup's
    // own off-diagonal block at once (one memory latency for both), wavefront 0 solves the diagonal
    // block while the others wait, then each applies the solved block to its 16 rows
    __shared__ double Lb[64][65];
    __shared__ double zb[64];
    __shared__ double part[4][64];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const long long j0 = a.j0, n = a.n;
    const int bs = (int)((n - j0) < 64 ? (n - j0) : 64);
    const bool off = blockIdx.x != 0;
    const long long i0 = TRANS ? j0 - 64 * (long long)blockIdx.x : j0 + 64 * (long long)blockIdx.x;
    double dg[16], ob[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int r = 16 * w + q;
        dg[q] = (r < bs && lane <= r) ? a.L[(j0 + r) * a.ldl + j0 + lane] : ((r == lane) ? 1.0 : 0.0);
        // own block, the 16 k of this wavefront: forward element (row i0 + lane, column j0 + 16 w + q)
        // -- one 128-byte line per lane --, transposed element (row j0 + 16 w + q, column i0 + lane)
        if (!TRANS) ob[q] = (off && i0 + lane < n) ? a.L[(i0 + lane) * a.ldl + j0 + r] : 0.0;
        else ob[q] = (off && r < bs) ? a.L[(j0 + r) * a.ldl + i0 + lane] : 0.0;
    }
    const double rj = (w == 0 && lane < bs) ? a.r[j0 + lane] : 0.0;
    const double rown = (w == 0 && off && i0 + lane < n) ? a.r[i0 + lane] : 0.0;   // (the entries this workgroup updates: requested now)
#pragma unroll
    for (int q = 0; q < 16; ++q) Lb[16 * w + q][lane] = dg[q];                 // (coalesced: lane = column)
    __syncthreads();
    if (w == 0) {
        double ri = rj;
        const double inv = 1.0 / Lb[lane][lane];
        double v[64];
#pragma unroll
        for (int k = 0; k < 64; ++k) v[k] = TRANS ? Lb[k][lane] : Lb[lane][k];   // column / row `lane` of the block
        if (!TRANS) {
            trtri_static_for<64>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                const double zk = trsv_bcast(ri, k) * trsv_bcast(inv, k);
                ri = lane == k ? zk : (lane > k ? fma(-v[k], zk, ri) : ri);
            });
        } else {
            trtri_static_for<64>([&](auto kc) {
                constexpr int k = 63 - decltype(kc)::value;
                const double xk = trsv_bcast(ri, k) * trsv_bcast(inv, k);
                ri = lane == k ? xk : (lane < k ? fma(-v[k], xk, ri) : ri);   // (L^T)[lane][k] = L[k][lane]
            });
        }
        zb[lane] = ri;
        if (!off) {
            if (lane < bs) a.x[j0 + lane] = ri;
            if (a.sumsq) {
                double ss = lane < bs ? ri * ri : 0.0;
                for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
                if (lane == 0) *a.sumsq += ss;              // (one writer per launch, launches in stream order)
            }
        }
    }
    __syncthreads();
    if (!off) return;
    // forward: r_i -= L[i, j0 .. j0+63] . z (lane = row i); transposed: r_c -= sum_k L[j0 + k, c] x_k
    // (lane = column c).  Either way 16 of the 64 k per wavefront, the four partial sums through LDS.
    double acc = 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc = fma(ob[q], zb[16 * w + q], acc);
    part[w][lane] = acc;
    __syncthreads();
    if (w == 0 && i0 + lane < n) a.r[i0 + lane] = rown - ((part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]));
}

// Round 3: FOUR 64-row blocks (256 rows) per launch.  The launch-to-launch dependency (~7 us) is what a step of the
// chain above costs whatever it computes, so a step now takes a 256-row diagonal "super-block": every workgroup
// solves it itself -- four 64 x 64 diagonal solves with the six 64 x 64 products between them, redundantly as
// before -- and applies the 256 solved values to ITS 64 rows (forward) / columns (transposed) of the remaining
// right-hand side.  Same arithmetic in the same order per element as the one-block step (a tile's product is the
// four wavefronts' 16-term partial sums added pairwise, tiles are subtracted in block order), so the solution has
// the same bits; a quarter of the launches.
#define TRSV_SB 4
template <int TRANS>
__global__ __launch_bounds__(256) void trsv_step4_kernel(TrsvStepArgs a) {
    __shared__ double Lb[64][65];
    __shared__ double zb[TRSV_SB * 64];
    __shared__ double part[4][64];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const long long J0 = a.j0, n = a.n;
    const int nsb = (int)(((n - J0) < 64 * TRSV_SB ? (n - J0) : 64 * TRSV_SB) + 63) / 64;    // real sub-blocks
    const bool off = blockIdx.x != 0;
    // this workgroup's block of the remaining right-hand side (forward: rows after the super-block; transposed:
    // columns before it)
    const long long i0 = TRANS ? J0 - 64 * (long long)blockIdx.x : J0 + 64 * TRSV_SB + 64 * ((long long)blockIdx.x - 1);
    // its tiles against the super-block's solution: requested now, used last
    double own[TRSV_SB][16];
#pragma unroll
    for (int q = 0; q < TRSV_SB; ++q)
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const long long k = J0 + 64 * q + 16 * w + kk;      // element of the super-block's solution
            if (!TRANS) own[q][kk] = (off && q < nsb && i0 + lane < n && k < n) ? a.L[(i0 + lane) * a.ldl + k] : 0.0;
            else own[q][kk] = (off && q < nsb && k < n) ? a.L[k * a.ldl + i0 + lane] : 0.0;
        }
    const double rown = (w == 0 && off && (TRANS || i0 + lane < n)) ? a.r[i0 + lane] : 0.0;
    // sub-blocks in dependency order: forward 0 .. nsb-1, transposed nsb-1 .. 0
    for (int step = 0; step < nsb; ++step) {
        const int sb = TRANS ? nsb - 1 - step : step;
        const long long j0 = J0 + 64 * sb;
        const int bs = (int)((n - j0) < 64 ? (n - j0) : 64);
        double dg[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int r = 16 * w + q;
            dg[q] = (r < bs && lane <= r) ? a.L[(j0 + r) * a.ldl + j0 + lane] : ((r == lane) ? 1.0 : 0.0);
        }
        double rj = (w == 0 && lane < bs) ? a.r[j0 + lane] : 0.0;
        // minus the already solved sub-blocks of this super-block, one tile at a time, in their order
        for (int st2 = 0; st2 < step; ++st2) {
            const int qb = TRANS ? nsb - 1 - st2 : st2;
            const long long q0 = J0 + 64 * qb;
            double ob[16];
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                const long long k = q0 + 16 * w + kk;
                if (!TRANS) ob[kk] = (lane < bs) ? a.L[(j0 + lane) * a.ldl + k] : 0.0;           // (k < j0 <= n)
                else ob[kk] = (k < n) ? a.L[k * a.ldl + j0 + lane] : 0.0;
            }
            double acc = 0.0;
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) acc = fma(ob[kk], zb[64 * qb + 16 * w + kk], acc);
            __syncthreads();                                 // (part is free: the previous tile's sum was consumed)
            part[w][lane] = acc;
            __syncthreads();
            if (w == 0) rj = rj - ((part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]));
        }
        __syncthreads();                                     // (Lb / part free)
#pragma unroll
        for (int q = 0; q < 16; ++q) Lb[16 * w + q][lane] = dg[q];
        __syncthreads();
        if (w == 0) {
            double ri = rj;
            const double inv = 1.0 / Lb[lane][lane];
            double v[64];
#pragma unroll
            for (int k = 0; k < 64; ++k) v[k] = TRANS ? Lb[k][lane] : Lb[lane][k];
            if (!TRANS) {
                trtri_static_for<64>([&](auto kc) {
                    constexpr int k = decltype(kc)::value;
                    const double zk = trsv_bcast(ri, k) * trsv_bcast(inv, k);
                    ri = lane == k ? zk : (lane > k ? fma(-v[k], zk, ri) : ri);
                });
            } else {
                trtri_static_for<64>([&](auto kc) {
                    constexpr int k = 63 - decltype(kc)::value;
                    const double xk = trsv_bcast(ri, k) * trsv_bcast(inv, k);
                    ri = lane == k ? xk : (lane < k ? fma(-v[k], xk, ri) : ri);
                });
            }
            zb[64 * sb + lane] = lane < bs ? ri : 0.0;
            if (!off) {
                if (lane < bs) a.x[j0 + lane] = ri;
                if (a.sumsq) {
                    double ss = lane < bs ? ri * ri : 0.0;
                    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
                    if (lane == 0) *a.sumsq += ss;          // (one writer, sub-blocks in order, launches in stream order)
                }
            }
        }
        __syncthreads();
    }
    if (!off) return;
    // this workgroup's block minus the super-block's contribution, tile by tile in the solving order
    double racc = rown;
    for (int step = 0; step < nsb; ++step) {
        const int qb = TRANS ? nsb - 1 - step : step;
        double acc = 0.0;
#pragma unroll
        for (int q = 0; q < TRSV_SB; ++q)
            if (q == qb) {
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) acc = fma(own[q][kk], zb[64 * q + 16 * w + kk], acc);
            }
        __syncthreads();
        part[w][lane] = acc;
        __syncthreads();
        if (w == 0) racc = racc - ((part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]));
    }
    if (w == 0 && (TRANS || i0 + lane < n)) a.r[i0 + lane] = racc;
}

// The 64 x 64 diagonal block's substitution in GROUPS of four pivots: the group's four right-hand-side entries and its
// 4 x 4 block are broadcast (v_readlane) and solved by every lane alike, then every lane applies the four solved values
// to its own row -- per lane the same FMAs in the same order as one pivot at a time (so the same bits), but the
// cross-lane dependency is met once per four pivots instead of once per pivot: a pivot-by-pivot step costs ~100 cycles
// (readlane -> SGPR -> multiply -> FMA), 2.7 us per block; a group ~140.
// v[k]: forward L[lane][k], transposed L[k][lane]; inv = 1 / L[lane][lane].
template <int TRANS>
__device__ __forceinline__ double trsv_diag_solve(double ri, const double inv, const double (&v)[64], const int lane) {
    trtri_static_for<16>([&](auto gc) {
        constexpr int g = TRANS ? 15 - decltype(gc)::value : decltype(gc)::value;
        constexpr int c0 = 4 * g;
        double r[4], iv[4], z[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { r[q] = trsv_bcast(ri, c0 + q); iv[q] = trsv_bcast(inv, c0 + q); }
        if (!TRANS) {
            // rows c0 .. c0+3 ascending: L[c0+q][c0+m] (m < q) = lane (c0+q)'s v[c0+m]
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int m = 0; m < q; ++m) r[q] = fma(-trsv_bcast(v[c0 + m], c0 + q), z[m], r[q]);
                z[q] = r[q] * iv[q];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) ri = lane == c0 + q ? z[q] : (lane > c0 + q ? fma(-v[c0 + q], z[q], ri) : ri);
        } else {
            // rows c0+3 .. c0 descending: (L^T)[c0+q][c0+m] (m > q) = L[c0+m][c0+q] = lane (c0+q)'s v[c0+m]
#pragma unroll
            for (int q = 3; q >= 0; --q) {
#pragma unroll
                for (int m = 3; m > q; --m) r[q] = fma(-trsv_bcast(v[c0 + m], c0 + q), z[m], r[q]);
                z[q] = r[q] * iv[q];
            }
#pragma unroll
            for (int q = 3; q >= 0; --q) ri = lane == c0 + q ? z[q] : (lane < c0 + q ? fma(-v[c0 + q], z[q], ri) : ri);
        }
    });
    return ri;
}

// ---------------------------------------------------------------------------
// K3, persistent (round 4): the blocked solve as ONE launch.  The launch-per-256-rows chain above costs its dependency
// (~7 us store -> first load of the next launch) 16 times at N = 4096 (0.59 / 0.43 ms); here workgroup j owns block row
// j (transposed: block column j), keeps its right-hand side block in a register per row, and is handed the solved
// blocks it depends on as data-tagged granules ({value half, tag} in one 8-byte word, 16 bytes per value, 1 KB per
// block: no flag, no fence; tools/probes/handoff_probe.hip: ~1 us idle) in dependency order; its tiles are requested
// one ahead.  A block costs its producer the 64-step diagonal solve + one hand-off instead of a launch.
// Roles come from a ticket (one atomic add per workgroup), not from blockIdx: a workgroup only ever waits for blocks
// whose tickets were drawn before its own, i.e. for workgroups that are already running -- no assumption about
// dispatch order or residency, so the spins cannot deadlock (they are bounded anyway: on timeout the block is written
// as NaN and *err set).
// Arithmetic per element = trsv_step4_kernel's (a tile's product is the four wavefronts' 16-term FMA chains added
// pairwise, tiles subtracted in block order, the diagonal block by the same substitution; x.x accumulated block by
// block in solving order), so the solution carries the same bits.
// ---------------------------------------------------------------------------
static std::atomic<int> g_trsv_multi_launch{0};
static thread_local int tl_trsv_mode = -1;      // >= 0: this thread's current call overrides the switch (apgp_trsv_ex)
// test / profiling switch: 1 = the launch-per-256-rows path (not read from the environment); returns the previous value
extern "C" int apgp_trsv_mode(int multi_launch) {
    if (multi_launch < 0) return g_trsv_multi_launch.load();
    return g_trsv_multi_launch.exchange(multi_launch ? 1 : 0);
}

struct TrsvPArgs {
    const double* L;
    const double* b;
    double* x;
    double* sumsq;
    long long n, ldl;
    double shift;
    unsigned long long* ticket;          // monotonic across calls; this call's tickets start at ticket_base
    unsigned long long ticket_base;
    unsigned long long* gran;            // [256 blocks][65 values][2 words]
    unsigned long long timeout;          // 100 MHz ticks
    int* err;
    unsigned tag;
};

template <int TRANS>
__global__ __launch_bounds__(256) void trsv_persist_kernel(TrsvPArgs a) {
    __shared__ double Lb[64][65];
    __shared__ double zb[64];
    __shared__ double part[4][64];
    __shared__ int ord_s, bad_s;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const long long n = a.n;
    const int nb = (int)((n + 63) / 64);
    if (t == 0) {
        ord_s = (int)(__hip_atomic_fetch_add(a.ticket, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - a.ticket_base);
        bad_s = 0;
    }
    __syncthreads();
    const int ord = ord_s;                                     // position in solving order
    if (ord < 0 || ord >= nb) return;
    const int j = TRANS ? nb - 1 - ord : ord;
    const long long j0 = (long long)j * 64;
    const int bs = (int)((n - j0) < 64 ? (n - j0) : 64);
    double dg[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int r = 16 * w + q;
        dg[q] = (r < bs && lane <= r) ? a.L[(j0 + r) * a.ldl + j0 + lane] : ((r == lane) ? 1.0 : 0.0);
    }
    double rj = (w == 0 && lane < bs) ? a.b[j0 + lane] - a.shift : 0.0;
    double srun = 0.0;                                         // x.x of the blocks solved before this one (wavefront 0)
    // this workgroup's tile against solved block p (in solving order): forward element (row j0 + lane, column k0 + 16 w + q)
    // -- one 128-byte line per lane --, transposed element (row k0 + 16 w + q, column j0 + lane)
    auto tile = [&](int p, double (&ob)[16]) {
        const int k = TRANS ? nb - 1 - p : p;
        const long long k0 = (long long)k * 64;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const long long kk = k0 + 16 * w + q;
            if (!TRANS) ob[q] = (lane < bs) ? a.L[(j0 + lane) * a.ldl + kk] : 0.0;             // (kk < j0 <= n)
            else ob[q] = (kk < n && lane < bs) ? a.L[kk * a.ldl + j0 + lane] : 0.0;
        }
    };
    // the diagonal block's row / column of this lane and its reciprocal pivot are in registers BEFORE the solved blocks
    // arrive (wavefront 0): after the last one only its tile's product and the 64-step substitution remain
#pragma unroll
    for (int q = 0; q < 16; ++q) Lb[16 * w + q][lane] = dg[q];                 // (coalesced: lane = column)
    __syncthreads();
    double v[64], inv = 1.0;
    if (w == 0) {
        inv = 1.0 / Lb[lane][lane];
#pragma unroll
        for (int k = 0; k < 64; ++k) v[k] = TRANS ? Lb[k][lane] : Lb[lane][k];   // column / row `lane` of the block
    }
    double obn[16];
    if (ord > 0) tile(0, obn);
    for (int p = 0; p < ord; ++p) {
        double ob[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) ob[q] = obn[q];
        if (p + 1 < ord) tile(p + 1, obn);                      // (the next tile's loads fly while this block is awaited)
        if (w == 0) {
            const int k = TRANS ? nb - 1 - p : p;
            // ONE 16-byte write-through-coherent load per lane and look (lane 0: a second one for the running sum), no sleep
            // between looks: this wavefront has nothing else to do, and the block it waits for is the critical path
            typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
            const __amdgpu_buffer_rsrc_t rsg = __builtin_amdgcn_make_buffer_rsrc((void*)a.gran, 0, 256 * 65 * 16, 0x00020000);
            const bool want_s = p == ord - 1;                   // (the running x.x travels with the immediate predecessor)
            u32x4_t g, gs = {0u, a.tag, 0u, a.tag};
            unsigned long long t0 = 0;
            unsigned it = 0;
            bool gave_up = false;
            // (two looks in flight instead of one measured slower: 0.273 vs 0.259 ms at n = 4096)
            for (;;) {
                g = __builtin_amdgcn_raw_buffer_load_b128(rsg, (unsigned)(lane * 16), (unsigned)(k * 65 * 16), 16);
                if (want_s) gs = __builtin_amdgcn_raw_buffer_load_b128(rsg, (unsigned)(64 * 16), (unsigned)(k * 65 * 16), 16);
                const bool ok = g.y == a.tag && g.w == a.tag && gs.y == a.tag && gs.w == a.tag;
                if (__all(ok)) break;
                // (only the immediate successor of the awaited block polls flat out: a workgroup d blocks further on has d
                // block times of slack, and its looks at the same kilobyte compete with the one that matters -- worth 1 %)
                for (int dd = ord - 1 - p < 8 ? ord - 1 - p : 8; dd > 0; --dd) __builtin_amdgcn_s_sleep(24);
                if ((++it & 63u) == 0) {
                    const unsigned long long now = __builtin_amdgcn_s_memrealtime();
                    if (t0 == 0) t0 = now;
                    else if (now - t0 > a.timeout) { gave_up = true; break; }
                }
            }
            const unsigned long long g0 = g.x, g1 = g.z, s0 = gs.x, s1 = gs.z;
            if (gave_up) {
                if (lane == 0) { bad_s = 1; atomicExch(a.err, 1); }
                zb[lane] = __builtin_nan("");
            } else {
                zb[lane] = __hiloint2double((int)(unsigned)g1, (int)(unsigned)g0);
                if (want_s) srun = __hiloint2double((int)(unsigned)s1, (int)(unsigned)s0);
            }
        }
        __syncthreads();                                         // (zb in place; part free: the previous tile's sum was consumed)
        double acc = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc = fma(ob[q], zb[16 * w + q], acc);
        part[w][lane] = acc;
        __syncthreads();
        if (w == 0) rj = rj - ((part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]));
    }
    if (w != 0) return;
    const double ri = trsv_diag_solve<TRANS>(rj, inv, v, lane);
    const double xo = lane < bs ? ri : 0.0;
    double ss = lane < bs ? ri * ri : 0.0;
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    const double snew = srun + ss;                               // (*sumsq += ss, block after block in solving order)
    if (ord + 1 < nb) {
        // hand the block on: 16 bytes per value, one write-through store per lane; the running sum as the 65th value
        typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.gran, 0, 256 * 65 * 16, 0x00020000);
        const u32x4_t g = {(unsigned)__double2loint(xo), a.tag, (unsigned)__double2hiint(xo), a.tag};
        // (the data registers stay live for two more issue slots: a 16-byte MUBUF store with an SGPR soffset reads its
        // data late on this chip although the ISA manual and LLVM's hazard recogniser exempt it -- see PP_STORE16 in
        // potrf_persist.h, where a recycled register put wrong low words under valid tags)
        __builtin_amdgcn_raw_buffer_store_b128(g, rs, (unsigned)(lane * 16), (unsigned)(j * 65 * 16), 16);
        asm volatile("s_nop 1" : : "v"(g));
        if (lane == 0) {
            const u32x4_t gs = {(unsigned)__double2loint(snew), a.tag, (unsigned)__double2hiint(snew), a.tag};
            __builtin_amdgcn_raw_buffer_store_b128(gs, rs, (unsigned)(64 * 16), (unsigned)(j * 65 * 16), 16);
            asm volatile("s_nop 1" : : "v"(gs));
        }
    }
    if (lane < bs) a.x[j0 + lane] = bad_s ? __builtin_nan("") : ri;
    if (a.sumsq && ord + 1 == nb && lane == 0) *a.sumsq = bad_s ? __builtin_nan("") : snew;
}

// words of stream scratch (slot 3) the persistent solve keeps: [0] ticket counter, [8 ...] granules
#define TRSV_P_WORDS (8 + 256 * 65 * 2)
#define TRSV_P_MAX_NB 256

extern "C" int apgp_trsv(const double* L, int64_t n, int64_t ldl, const double* b, double shift,
                         int trans, double* x, double* sumsq, void* stream) {
    APGP_CHECK_ARG(L && b && x, "null pointer");
    APGP_CHECK_ARG(n >= 1 && n <= APGP_MAX_N && ldl >= n, "n >= 1 and ldl >= n required");
    hipStream_t st = (hipStream_t)stream;
    const int multi_launch = tl_trsv_mode >= 0 ? tl_trsv_mode : g_trsv_multi_launch.load();
    if (n >= 256 && (n + 63) / 64 <= TRSV_P_MAX_NB && !multi_launch) {
        // ONE persistent launch (trsv_persist_kernel): flags-free granule hand-offs, call-unique tags and tickets (the
        // scratch is zeroed when it is allocated, never between calls); b may alias x (a block's owner alone touches it)
        std::lock_guard<std::mutex> enqueue_lock(apgp_stream_lock(st));
        bool fresh = false;
        unsigned long long* calls = nullptr;
        unsigned long long* pw = (unsigned long long*)apgp_stream_scratch_ex(3, st, (size_t)TRSV_P_WORDS + 8, &fresh, &calls);
        if (!pw) {
            apgp_set_error("apgp_trsv: scratch allocation failed");
            return -2;
        }
        // calls[0] doubles as the ticket base (tickets handed out so far); the tag is a second counter kept in word 1..
        // (both live with the scratch entry: a fresh buffer starts them over)
        if (fresh) {
            if (hipMemsetAsync(pw, 0, ((size_t)TRSV_P_WORDS + 8) * 8, st) != hipSuccess) {
                apgp_set_error("apgp_trsv: memset failed");
                return -2;
            }
            *calls = 0;
        }
        const long long nb = (n + 63) / 64;
        TrsvPArgs pa;
        pa.L = L; pa.b = b; pa.x = x; pa.sumsq = sumsq; pa.n = n; pa.ldl = ldl; pa.shift = shift;
        pa.ticket = pw; pa.ticket_base = *calls & 0xffffffffffull;          // low 40 bits: tickets drawn by earlier calls
        pa.gran = pw + 8;
        pa.timeout = 5000000ull;                                           // 50 ms of the 100 MHz clock
        pa.err = (int*)(pw + 4);
        unsigned long long ncall = (*calls >> 40) + 1;                     // high bits: calls so far -> the granule tag
        if ((unsigned)ncall == 0u || ncall >= (1ull << 23)) {
            // (the tag space of this buffer is used up: start over on a zeroed buffer)
            if (hipMemsetAsync(pw, 0, ((size_t)TRSV_P_WORDS + 8) * 8, st) != hipSuccess) {
                apgp_set_error("apgp_trsv: memset failed");
                return -2;
            }
            *calls = 0; ncall = 1; pa.ticket_base = 0;
        }
        pa.tag = (unsigned)ncall;
        if (!trans) hipLaunchKernelGGL(trsv_persist_kernel<0>, dim3((unsigned)nb), dim3(256), 0, st, pa);
        else hipLaunchKernelGGL(trsv_persist_kernel<1>, dim3((unsigned)nb), dim3(256), 0, st, pa);
        APGP_CHECK_LAUNCH();
        // (only a launch that was accepted draws its tickets on the device: the host-side base moves with it, not before --
        // a failed launch would otherwise leave every later call on this stream with tickets nobody hands out)
        *calls = (ncall << 40) | ((pa.ticket_base + (unsigned long long)nb) & 0xffffffffffull);
        return 0;
    }
    if (n >= 256) {
        // blocked form (trsv_step4_kernel): one small launch per 256 rows (round 2, trsv_step_kernel: per 64-row block (~10 us each, bound by the
        // launch-to-launch dependency; the single-workgroup form below costs ~10 us per block at N = 512
        // and 26 us per block at N = 4096).  Scratch and launches
        // of one solve are taken as a unit: host threads sharing a stream share the scratch.
        std::lock_guard<std::mutex> enqueue_lock(apgp_stream_lock(st));   // (per device and stream)
        double* r = apgp_stream_scratch(1, st, (size_t)n);
        if (!r) {
            apgp_set_error("apgp_trsv: scratch allocation failed");
            return -2;
        }
        hipLaunchKernelGGL(trsv_init_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, b, shift, r, sumsq, (long long)n);
        TrsvStepArgs sa;
        sa.L = L; sa.r = r; sa.x = x; sa.sumsq = sumsq; sa.n = n; sa.ldl = ldl;
        const long long nb = (n + 63) / 64;
        const long long nsup = (nb + TRSV_SB - 1) / TRSV_SB;          // 256-row super-blocks
        if (!trans) {
            for (long long sj = 0; sj < nsup; ++sj) {
                sa.j0 = sj * 64 * TRSV_SB;
                const long long after = nb - (sj + 1) * TRSV_SB;     // 64-row blocks below the super-block
                hipLaunchKernelGGL(trsv_step4_kernel<0>, dim3((unsigned)(1 + (after > 0 ? after : 0))), dim3(256), 0, st, sa);
            }
        } else {
            for (long long sj = nsup - 1; sj >= 0; --sj) {
                sa.j0 = sj * 64 * TRSV_SB;
                hipLaunchKernelGGL(trsv_step4_kernel<1>, dim3((unsigned)(1 + sj * TRSV_SB)), dim3(256), 0, st, sa);
            }
        }
        APGP_CHECK_LAUNCH();
        return 0;
    }
    APGP_CHECK_ARG(n <= 15360, "n <= 15360 (right-hand side is kept in LDS)");
    TrsvArgs a;
    a.L = L; a.b = b; a.x = x; a.sumsq = sumsq; a.n = n; a.ldl = ldl; a.shift = shift;
    a.trans = trans;
    size_t nr = (size_t)apgp_round_up(n, 64);
    size_t lds = (nr + TRSV_B * (TRSV_B + 1) + 16 * 64) * sizeof(double);
    {
        // per device, checked (the single-workgroup form keeps the right-hand side in LDS)
        static std::mutex attr_mu;
        static bool attr_set[64] = {false};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
            apgp_set_error("apgp_trsv: hipGetDevice failed");
            return -2;
        }
        std::lock_guard<std::mutex> lock(attr_mu);
        if (!attr_set[dev]) {
            if (hipFuncSetAttribute((const void*)trsv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
                apgp_set_error("apgp_trsv: hipFuncSetAttribute failed");
                return -2;
            }
            attr_set[dev] = true;
        }
    }
    hipLaunchKernelGGL(trsv_kernel, dim3(1), dim3(1024), lds, (hipStream_t)stream, a);
    APGP_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------
// Diagonal entry of an appended factor row (incremental fit, approx.py:693-717): after
// l = L^-1 k(x_new, X_old) (apgp_kernel_cross + apgp_trsv, *ss = l.l) the new pivot is
// d = sqrt(kdiag - l.l); a non-positive pivot is reported LAPACK-style through *info_dev
// (0 = none; the first failing leading-minor order is kept) and d = 1 keeps later rows finite.
// No host round trip per appended row.
// ---------------------------------------------------------------------------
__global__ void append_diag_kernel(double* ljj, const double* ss, double kdiag, int* info, int order) {
    const double d2 = kdiag - *ss;
    if (d2 > 0.0 && d2 < INFINITY) *ljj = sqrt(d2);
    else { *ljj = 1.0; atomicCAS(info, 0, order); }
}

// apgp_trsv with the path chosen PER CALL (mode 0: the persistent launch where it applies, 1: a launch per 256 rows, < 0: the
// process-wide switch apgp_trsv_mode) -- what a caller uses to re-run one solve whose persistent launch gave up (NaN), without
// changing the path of other threads' or streams' calls.
extern "C" int apgp_trsv_ex(const double* L, int64_t n, int64_t ldl, const double* b, double shift,
                            int trans, double* x, double* sumsq, int mode, void* stream) {
    const int saved = tl_trsv_mode;
    tl_trsv_mode = mode < 0 ? -1 : (mode ? 1 : 0);
    const int rc = apgp_trsv(L, n, ldl, b, shift, trans, x, sumsq, stream);
    tl_trsv_mode = saved;
    return rc;
}

extern "C" int apgp_append_diag(double* ljj, const double* ss, double kdiag, int32_t* info_dev, int64_t order,
                                void* stream) {
    APGP_CHECK_ARG(ljj && ss && info_dev, "null pointer");
    APGP_CHECK_ARG(order >= 1 && order < (1ll << 31), "order out of range");
    hipLaunchKernelGGL(append_diag_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, ljj, ss, kdiag, (int*)info_dev, (int)order);
    APGP_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------
// K3 with the explicit inverse: z = W (b - shift) or x = W^T b for the dense lower-triangular
// W = L^-1 that apgp_trtri_pack leaves resident for every sweep set-up.  Two HBM-rate GEMVs
// (the 67 MB triangle at N = 4096 in ~20 us each) replace the two single-workgroup triangular
// solves (1.6 ms each) of GP._compute_alpha (george; utility.py:131 via GP.predict).  Same
// accuracy class as the sweep's own use of W: the caller keeps the solve form above cond 1e10.
// Both kernels reduce in a fixed order (bit-reproducible).
// ---------------------------------------------------------------------------
#define WA_RCH 128     // rows per partial-sum chunk of the transposed product
__global__ __launch_bounds__(256) void winv_gemv_kernel(const double* W, long long ldw, long long n,
                                                        const double* b, double shift, double* x) {
    // one wavefront per row i: x_i = sum_{k <= i} W_ik (b_k - shift)
    const int lane = threadIdx.x & 63;
    const long long i = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const double* wr = W + i * ldw;
    double s0 = 0.0, s1 = 0.0;
    for (long long k = 2 * lane; k <= i; k += 128) {
        const f64x2 w2 = *(const f64x2*)(wr + k);          // W is zero above the diagonal
        s0 = fma(w2.x, b[k] - shift, s0);
        if (k + 1 <= i) s1 = fma(w2.y, b[k + 1] - shift, s1);
    }
    double s = s0 + s1;
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) x[i] = s;
}

__global__ __launch_bounds__(256) void winv_gemvt_partial_kernel(const double* W, long long ldw, long long n,
                                                                 const double* z, double* part) {
    // thread per column k, rows [r0, r0 + WA_RCH): part[chunk][k] = sum_i W_ik z_i (i >= k)
    const long long k = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long r0 = (long long)blockIdx.y * WA_RCH;
    if (r0 + WA_RCH <= (long long)blockIdx.x * 256) {          // chunk entirely above the diagonal
        if (k < n) part[(long long)blockIdx.y * n + k] = 0.0;
        return;
    }
    __shared__ double zs[WA_RCH];
    if (threadIdx.x < WA_RCH) zs[threadIdx.x] = (r0 + threadIdx.x < n) ? z[r0 + threadIdx.x] : 0.0;
    __syncthreads();
    if (k >= n) return;
    const long long r1 = r0 + WA_RCH < n ? r0 + WA_RCH : n;
    double s0 = 0.0, s1 = 0.0;
    long long i = r0 > k ? r0 : k;
    for (; i + 1 < r1; i += 2) {
        s0 = fma(W[i * ldw + k], zs[i - r0], s0);
        s1 = fma(W[(i + 1) * ldw + k], zs[i + 1 - r0], s1);
    }
    if (i < r1) s0 = fma(W[i * ldw + k], zs[i - r0], s0);
    part[(long long)blockIdx.y * n + k] = s0 + s1;
}

__global__ __launch_bounds__(256) void winv_gemvt_reduce_kernel(const double* part, long long n, int nchunk, double* x) {
    const long long k = (long long)blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    double s = 0.0;
    for (int c = 0; c < nchunk; ++c) s += part[(long long)c * n + k];
    x[k] = s;
}

__global__ __launch_bounds__(1024) void sumsq_kernel(const double* x, long long n, double* out) {
    __shared__ double red[16];
    double s = 0.0;
    for (long long i = threadIdx.x; i < n; i += 1024) s = fma(x[i], x[i], s);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < 16; ++i) t += red[i];
        *out = t;
    }
}

extern "C" int64_t apgp_winv_apply_work_len(int64_t n) {
    return n < 1 ? 0 : (n > APGP_MAX_N ? -1 : ((n + WA_RCH - 1) / WA_RCH) * n);
}

extern "C" int apgp_winv_apply(const double* winv, int64_t ldw, int64_t n, const double* b, double shift,
                               int trans, double* x, double* sumsq, double* work, void* stream) {
    APGP_CHECK_ARG(winv && b && x, "null pointer");
    APGP_CHECK_ARG(n >= 1 && n <= APGP_MAX_N && ldw >= n, "n >= 1 and ldw >= n required");
    APGP_CHECK_ARG(!trans || work, "work (apgp_winv_apply_work_len doubles) required for the transposed product");
    APGP_CHECK_ARG(!trans || shift == 0.0, "shift applies to the forward product only");
    APGP_CHECK_ARG(x != b, "x must not alias b");
    hipStream_t s = (hipStream_t)stream;
    if (!trans) {
        hipLaunchKernelGGL(winv_gemv_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, winv, (long long)ldw,
                           (long long)n, b, shift, x);
    } else {
        const int nchunk = (int)((n + WA_RCH - 1) / WA_RCH);
        hipLaunchKernelGGL(winv_gemvt_partial_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)nchunk), dim3(256), 0, s,
                           winv, (long long)ldw, (long long)n, b, work);
        hipLaunchKernelGGL(winv_gemvt_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s,
                           (const double*)work, (long long)n, nchunk, x);
    }
    if (sumsq) hipLaunchKernelGGL(sumsq_kernel, dim3(1), dim3(1024), 0, s, (const double*)x, (long long)n, sumsq);
    APGP_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------
// ONE candidate: the reference-faithful scalar path.  utility.AGPUtility / BAPEUtility / JonesUtility call
// gp.predict(y, theta[1 x D], return_var=True) once per Nelder-Mead step (utility.py:131,178,224 from
// minimizeObjective, utility.py:336-372: 372-429 calls per restart), and ApproxPosterior.findNextPoint uses that
// search by default.  Through the fused sweep a single candidate costs a split launch, its finish and arg-min
// kernels, five allocations and three synchronising copies (75 us at N = 50, 660 us at N = 4096; substitution
// form 4.4 ms).  Here: k* (one thread per training point, mu partial per workgroup) -> v = L^-1 k* as ONE
// matrix-vector product with the resident dense L^-1 (or one triangular solve against L when the inverse is
// not trusted) -> a single-workgroup epilogue (sum v^2, mu, sigma^2) that writes (mu, sigma^2) into the stream's
// pinned mailbox; the candidate travels in the kernel arguments.  No allocation, no copy, no synchronisation.
// ---------------------------------------------------------------------------
struct Pred1Args {
    const double* xs;
    double* kstar;
    double* mu_part;
    const double* v;
    const double* q_in;      // sum v^2 already reduced (triangular-solve path) or NULL
    double* out2;
    double* mail;
    long long seq;
    long long n;
    int nparts;
    double mean, ktt, amp, lin_coef;
    int ndim, lin_order, has_nan;
    double tt[APGP_MAX_DIM], lw[APGP_MAX_DIM];
};

template <int DPAD>
__global__ __launch_bounds__(256) void pred1_kstar_kernel(Pred1Args a) {
    constexpr int XS = DPAD + 2;
    __shared__ double etab[APGP_EXP_TAB_N];
    __shared__ double red[4];
    apgp_exp_tab_load(etab);
    __syncthreads();
    const long long k = (long long)blockIdx.x * 256 + threadIdx.x;
    double contrib = 0.0;
    if (k < a.n) {
        const double* xr = a.xs + k * XS;
        double s = 0.0, s3 = 0.0;
#pragma unroll
        for (int d = 0; d < DPAD; d += 2) {
            const double df0 = a.tt[d] - xr[d], df1 = a.tt[d + 1] - xr[d + 1];
            s = fma(df0, df0, s);
            s3 = fma(df1, df1, s3);
        }
        double kv = a.amp * apgp_exp(-(s + s3), etab);
        if (a.lin_coef != 0.0) {
            double ls;
            APGP_LIN_SUM(ls, DPAD, a.ndim, a.lin_order, a.tt[d_] * xr[d_] * a.lw[d_]);
            kv = fma(a.lin_coef, ls, kv);
        }
        a.kstar[k] = kv;
        contrib = kv * xr[DPAD];                       // k* alpha
    }
    for (int o = 32; o > 0; o >>= 1) contrib += __shfl_xor(contrib, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = contrib;
    __syncthreads();
    if (threadIdx.x == 0) a.mu_part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(1024) void pred1_final_kernel(Pred1Args a) {
    __shared__ double red[16];
    double q = 0.0;
    if (a.q_in) {
        q = *a.q_in;
    } else {
        double s = 0.0;
        for (long long i = threadIdx.x; i < a.n; i += 1024) s = fma(a.v[i], a.v[i], s);
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0)
            for (int i = 0; i < 16; ++i) q += red[i];
    }
    if (threadIdx.x != 0) return;
    double mu = 0.0;
    for (int i = 0; i < a.nparts; ++i) mu += a.mu_part[i];
    mu += a.mean;
    double var = a.ktt - q;                             // k(t,t): no white noise (george predict)
    if (a.has_nan) { mu = NAN; var = NAN; }
    a.out2[0] = mu;
    a.out2[1] = var;
    if (a.mail) {
        a.mail[0] = mu;
        a.mail[1] = var;
        __hip_atomic_store((long long*)(a.mail + 5), a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// The same three steps for n <= 256 through the dense inverse as ONE single-workgroup launch (round 5: the README
// configuration's point search is 18,600 of these calls at N = 50 .. 90): k* and v stay in LDS; every sum is formed by the
// thread, in the order, of the kernel it replaces -- pred1_kstar_kernel's workgroup reduction (nparts = 1), winv_gemv_kernel's
// wavefront per row (four rows at a time), pred1_final_kernel's 1,024 "virtual threads" of which the first 256 are these
// (the other twelve wavefront partials are zeros) -- so (mu, sigma^2) carry the same bits.
template <int DPAD>
__global__ __launch_bounds__(256) void pred1_small_kernel(Pred1Args a, const double* W, long long ldw) {
    constexpr int XS = DPAD + 2;
    __shared__ double etab[APGP_EXP_TAB_N];
    __shared__ double red[4], red2[4];
    __shared__ __attribute__((aligned(16))) double ks[256];
    __shared__ double vs[256];
    apgp_exp_tab_load(etab);
    __syncthreads();
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const long long n = a.n;
    double contrib = 0.0, kv = 0.0;
    if (t < n) {
        const double* xr = a.xs + (long long)t * XS;
        double s = 0.0, s3 = 0.0;
#pragma unroll
        for (int d = 0; d < DPAD; d += 2) {
            const double df0 = a.tt[d] - xr[d], df1 = a.tt[d + 1] - xr[d + 1];
            s = fma(df0, df0, s);
            s3 = fma(df1, df1, s3);
        }
        kv = a.amp * apgp_exp(-(s + s3), etab);
        if (a.lin_coef != 0.0) {
            double ls;
            APGP_LIN_SUM(ls, DPAD, a.ndim, a.lin_order, a.tt[d_] * xr[d_] * a.lw[d_]);
            kv = fma(a.lin_coef, ls, kv);
        }
        contrib = kv * xr[DPAD];                       // k* alpha
    }
    ks[t] = kv;
    for (int o = 32; o > 0; o >>= 1) contrib += __shfl_xor(contrib, o);
    if (lane == 0) red[w] = contrib;
    __syncthreads();
    // v = W k*: one wavefront per row, as winv_gemv_kernel (shift 0)
    for (long long i = w; i < n; i += 4) {
        const double* wr = W + i * ldw;
        double s0 = 0.0, s1 = 0.0;
        for (long long k = 2 * lane; k <= i; k += 128) {
            const f64x2 w2 = *(const f64x2*)(wr + k);
            s0 = fma(w2.x, ks[k] - 0.0, s0);
            if (k + 1 <= i) s1 = fma(w2.y, ks[k + 1] - 0.0, s1);
        }
        double s = s0 + s1;
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) vs[i] = s;
    }
    __syncthreads();
    // sum v^2: virtual thread t of pred1_final_kernel's 1,024
    double sq = 0.0;
    if (t < n) sq = fma(vs[t], vs[t], sq);
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    if (lane == 0) red2[w] = sq;
    __syncthreads();
    if (t != 0) return;
    double q = 0.0;
    for (int i = 0; i < 4; ++i) q += red2[i];              // (+ twelve zero partials)
    double mu = 0.0;
    mu += (red[0] + red[1]) + (red[2] + red[3]);           // mu_part[0]
    mu += a.mean;
    double var = a.ktt - q;
    if (a.has_nan) { mu = NAN; var = NAN; }
    a.out2[0] = mu;
    a.out2[1] = var;
    if (a.mail) {
        a.mail[0] = mu;
        a.mail[1] = var;
        __hip_atomic_store((long long*)(a.mail + 5), a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

extern "C" int apgp_potrf_mode(int mode);      // (mode 1 = separate launches everywhere: the bit-identity tests' reference)
extern "C" int64_t apgp_predict1_work_len(int64_t n) { return n < 1 ? 0 : (n > APGP_MAX_N ? -1 : 2 * apgp_npad(n) + (n + 255) / 256 + 8); }

extern "C" int apgp_predict1_host(const double* t_host, const double* xs, int64_t n, const apgp_kernel_t* kern, double mean,
                                  const double* winv, int64_t ldw, const double* L, int64_t ldl, double* work,
                                  double* out2_host, void* stream) {
    APGP_CHECK_ARG(t_host && xs && kern && work && out2_host, "null pointer");
    APGP_CHECK_ARG(n >= 1 && n <= APGP_MAX_N, "n >= 1 required");
    APGP_CHECK_ARG((winv && ldw >= n) || (L && ldl >= n), "the dense inverse (ldw >= n) or the factor (ldl >= n) is required");
    KernConst kc;
    APGP_CHECK_ARG(apgp_make_kernconst(kern, &kc) == 0, "kernel parameters");
    hipStream_t s = (hipStream_t)stream;
    const long long npad = apgp_npad(n);
    Pred1Args a;
    a.xs = xs; a.kstar = work; a.v = work + npad; a.mu_part = work + 2 * npad;
    a.nparts = (int)((n + 255) / 256);
    double* qbuf = work + 2 * npad + a.nparts;
    a.out2 = qbuf + 2;
    a.q_in = nullptr; a.mail = nullptr; a.seq = 0;
    a.n = n; a.mean = mean; a.amp = kc.amp; a.lin_coef = kc.lin_coef; a.ndim = kc.ndim; a.lin_order = kc.lin_order;
    a.has_nan = 0;
    double ktl = kc.lin_order == 0 ? (double)kc.ndim : 0.0;
    for (int d = 0; d < APGP_MAX_DIM; ++d) {
        const double v = d < kc.ndim ? t_host[d] : 0.0;
        if (v != v) a.has_nan = 1;
        a.tt[d] = v * kc.sc[d];
        a.lw[d] = kc.lw[d];
        if (d < kc.ndim && kc.lin_coef != 0.0 && kc.lin_order > 0) {
            double p = v * v, qq = p;
            for (int e = 1; e < kc.lin_order; ++e) qq *= p;
            ktl += qq;
        }
    }
    a.ktt = kc.lin_coef != 0.0 ? fma(kc.lin_coef, ktl, kc.amp) : kc.amp;
    const dim3 grid((unsigned)a.nparts), block(256);
    const bool fused = winv && n <= 256 && (apgp_potrf_mode(-1) & 15) != 1;
    if (!fused)
    switch (kc.dpad) {
        case 2: hipLaunchKernelGGL(pred1_kstar_kernel<2>, grid, block, 0, s, a); break;
        case 4: hipLaunchKernelGGL(pred1_kstar_kernel<4>, grid, block, 0, s, a); break;
        case 8: hipLaunchKernelGGL(pred1_kstar_kernel<8>, grid, block, 0, s, a); break;
        case 16: hipLaunchKernelGGL(pred1_kstar_kernel<16>, grid, block, 0, s, a); break;
        default: hipLaunchKernelGGL(pred1_kstar_kernel<32>, grid, block, 0, s, a); break;
    }
    if (fused) {
        // (launched below, with the mailbox)
    } else if (winv) {
        hipLaunchKernelGGL(winv_gemv_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, winv, (long long)ldw,
                           (long long)n, (const double*)a.kstar, 0.0, (double*)a.v);
    } else {
        // (takes the stream's enqueue lock itself for n >= 256: not held here)
        const int rc = apgp_trsv(L, n, ldl, a.kstar, 0.0, 0, (double*)a.v, qbuf, stream);
        if (rc != 0) return rc;
        a.q_in = qbuf;
    }
    ApgpMailbox* mb = apgp_stream_mailbox(s);
    const bool mail = mb && mb->host;
    std::lock_guard<std::mutex> lock(apgp_stream_lock(s));
    if (mail) { a.mail = mb->dev; a.seq = ++mb->seq; }
    if (fused) {
        switch (kc.dpad) {
            case 2: hipLaunchKernelGGL(pred1_small_kernel<2>, dim3(1), dim3(256), 0, s, a, winv, (long long)ldw); break;
            case 4: hipLaunchKernelGGL(pred1_small_kernel<4>, dim3(1), dim3(256), 0, s, a, winv, (long long)ldw); break;
            case 8: hipLaunchKernelGGL(pred1_small_kernel<8>, dim3(1), dim3(256), 0, s, a, winv, (long long)ldw); break;
            case 16: hipLaunchKernelGGL(pred1_small_kernel<16>, dim3(1), dim3(256), 0, s, a, winv, (long long)ldw); break;
            default: hipLaunchKernelGGL(pred1_small_kernel<32>, dim3(1), dim3(256), 0, s, a, winv, (long long)ldw); break;
        }
    } else {
        hipLaunchKernelGGL(pred1_final_kernel, dim3(1), dim3(1024), 0, s, a);
    }
    APGP_CHECK_LAUNCH();
    if (mail) {
        volatile long long* flag = (volatile long long*)(mb->host + 5);
        const auto t0 = std::chrono::steady_clock::now();
        unsigned spins = 0;
        while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != a.seq) {
            if ((++spins & 255u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(400)) {
                if (hipStreamSynchronize(s) != hipSuccess || __atomic_load_n(flag, __ATOMIC_ACQUIRE) != a.seq) {
                    apgp_set_error("apgp_predict1_host: result record not written");
                    return -2;
                }
                break;
            }
        }
        out2_host[0] = mb->host[0];
        out2_host[1] = mb->host[1];
        return 0;
    }
    if (hipMemcpyAsync(out2_host, a.out2, 2 * sizeof(double), hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess) {
        apgp_set_error("apgp_predict1_host: D2H copy failed");
        return -2;
    }
    return 0;
}

// ---------------------------------------------------------------------------
// Triangular inversion W = L^-1.
//   (1) invert every 64x64 diagonal block (one wavefront per block, LDS);
//   (2) log2 merge levels: for adjacent blocks A (first) and B (second) with
//       C = L[second, first]:  W[second, first] = -B^-1 (C A^-1)
//       as two MFMA-f64 GEMM launches per level that skip the structurally
//       zero parts of the triangular operands;
//   (3) pack into the sweep's tile layout.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void trtri_diag_kernel(const double* L, long long n, long long ldl,
                                                        double* W, long long ldw) {
    // lane c owns column c of X = L_jj^-1 in registers; row i of L_jj is the same for every lane
    // and comes back as broadcast LDS reads (the bound: ~20 cycles per ds_read_b128 and wavefront,
    // tools/probes/lat_probe.hip).  Fully unrolled, x[k] = 0 above the diagonal, so the sum runs over all
    // k < i in order: the same operations on the same values as a per-column loop from k = c.
    __shared__ __attribute__((aligned(16))) double Lb[64][66];
    const long long j0 = (long long)blockIdx.x * 64;
    const int c = threadIdx.x;
    __shared__ __attribute__((aligned(16))) double rinv[64];
    {
        double v[64];                                   // all 64 rows requested together: one latency
#pragma unroll
        for (int r = 0; r < 64; ++r) {
            const long long gr = j0 + r, gc = j0 + c;
            v[r] = (gr < n && gc < n && c <= r) ? L[gr * ldl + gc] : ((r == c) ? 1.0 : 0.0);
        }
#pragma unroll
        for (int r = 0; r < 64; ++r) {
            Lb[r][c] = v[r];                            // (coalesced: lane = column)
            if (r == c) rinv[c] = 1.0 / v[r];           // reciprocal pivots, as LAPACK's dtrti2 scales by -1/a_jj
        }
    }
    __syncthreads();
    double x[64];
    trtri_static_for<64>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        double srow = 0.0;
        f64x2 lr[(i + 1) / 2 > 0 ? (i + 1) / 2 : 1];
#pragma unroll
        for (int k = 0; k + 1 < i + 1 && k < i; k += 2) lr[k / 2] = *(const f64x2*)(&Lb[i][k]);
        const double ri = rinv[i];
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < i; ++k) srow = fma((k & 1) ? lr[k / 2].y : lr[k / 2].x, x[k], srow);
        x[i] = i < c ? 0.0 : (i == c ? ri : -srow * ri);
        asm volatile("" : "+v"(x[i]));
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    });
#pragma unroll
    for (int i = 0; i < 64; ++i) W[(j0 + i) * ldw + j0 + c] = x[i];
}

struct MergeArgs {
    const double* L;
    double* W;
    double* T;
    long long ldl, ldw, n;
    int nb;      // number of 64-blocks
    int s;       // level: blocks per half
    int phase;   // 1: T = C * Ainv ; 2: W[second,first] = -Binv * T
    int pair;    // two complementary tiles per workgroup (launches of >= 512 tiles)
};

// 64 x 64 output tile, 4 wavefronts (2 x 2), each 32 x 32 = 2 x 2 blocks of 16 x 16 on the
// four-block f64 MFMA (apgp_mma16: twice the rate of v_mfma_f64_16x16x4 on gfx950); K staged
// through LDS in chunks of 16.
__global__ __launch_bounds__(256) void trtri_merge_kernel(MergeArgs a) {
    __shared__ __attribute__((aligned(16))) double lds[GEMM64_LDS_DOUBLES];
    const int q = blockIdx.z;
    const long long first0 = (long long)2 * q * a.s;          // in 64-blocks
    const long long second0 = first0 + a.s;
    if (second0 >= a.nb) return;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int wr = (w >> 1) * 32, wc = (w & 1) * 32;
    double* Cp = (a.phase == 1) ? a.T : a.W;
    const double sgn = (a.phase == 1) ? 1.0 : -1.0;
    // The k range of a tile grows with its column block in phase 1 (k >= c) and with its row block
    // in phase 2 (k <= r): a workgroup takes the two tiles whose ranges add up to the same length
    // for every workgroup (block u and block s - 1 - u), so the launch is balanced by construction
    // whatever the dispatch order.  (Small launches: one tile per workgroup.)
    const int ntile = a.pair ? 2 : 1;
    for (int h = 0; h < ntile; ++h) {
        long long bx = blockIdx.x, by = blockIdx.y;
        if (a.phase == 1) { if (h) bx = a.s - 1 - bx; }
        else { if (h) by = a.s - 1 - by; }
        const long long rb = second0 + by;                   // output row block
        if (rb >= a.nb) continue;
        const long long cb = first0 + bx;                    // output col block
        const long long r0 = rb * 64, c0 = cb * 64;
        double acc[2][2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.0;
        if (a.phase == 1)     // T[r, c] = sum_k L[r, k] W[k, c], k in first, W lower => k >= c ; L has only n rows
            apgp_gemm64_tile<false, true>(a.L + r0 * a.ldl, a.ldl, a.n - r0, a.W + c0, a.ldw, 0, c0, second0 * 64, lds, acc);
        else                  // W[r, c] = -sum_k W[r, k] T[k, c], k in second, W lower => k <= r
            apgp_gemm64_tile<false, true>(a.W + r0 * a.ldw, a.ldw, 64, a.T + c0, a.ldw, 0, second0 * 64, r0 + 64, lds, acc);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // four-block layout (mma16.h): rotation r of lane l -> (row, col) of the 16 x 16 block
                    long long gr = r0 + wr + 16 * i + apgp_mma16_row(lane);
                    long long gc = c0 + wc + 16 * j + apgp_mma16_col(lane, r);
                    Cp[gr * a.ldw + gc] = sgn * acc[i][j][r];
                }
    }
}

// Packed tile (ib, kc): ROW_BLOCK rows x K_CHUNK(=16) k stored as
//   [s 0..RS-1][kp 0..1][p 0..63][q 0..1]   (doubles)
// holding W[ib*ROW_BLOCK + 16 s + (p & 15)][kc*16 + 4*(2 kp + q) + (p >> 4)]:
// p is the MFMA A-operand lane (row-in-16 + 16 * k-in-4) and (kp, q) the k-step
// kk = 2 kp + q.  One ds_read_b128 at ((s*2+kp)*64 + p)*16 B therefore returns
// the A values of k-steps 2kp and 2kp+1 for lane position p; consecutive lanes
// read consecutive 16-byte slots (conflict-free, full LDS rate from one
// wavefront per SIMD).  Tiles are stored row block by row block, k-chunks
// 0 .. (ib+1)*CPB-1 each.
__global__ __launch_bounds__(256) void pack_linv_kernel(const double* W, long long ldw, long long n,
                                                        double* packed) {
    constexpr int CPB = APGP_ROW_BLOCK / APGP_K_CHUNK;
    constexpr int RS = APGP_ROW_BLOCK / 16;
    static_assert(APGP_K_CHUNK == 16 && RS * 256 == APGP_ROW_BLOCK * APGP_K_CHUNK, "tile layout");
    const long long tile = blockIdx.x;
    // invert tile = CPB*ib*(ib+1)/2 + kc
    long long ib = (long long)((sqrt(8.0 * (double)tile / CPB + 1.0) - 1.0) * 0.5);
    while (CPB * (ib + 1) * (ib + 2) / 2 <= tile) ++ib;
    while (CPB * ib * (ib + 1) / 2 > tile) --ib;
    const long long kc = tile - CPB * ib * (ib + 1) / 2;
    double* out = packed + tile * (long long)(APGP_ROW_BLOCK * APGP_K_CHUNK);
    for (int e = threadIdx.x; e < APGP_ROW_BLOCK * APGP_K_CHUNK; e += 256) {
        const int q = e & 1, lane = (e >> 1) & 63, kp = (e >> 7) & 1, s = e >> 8;
        const int kk = 2 * kp + q;
        long long row = ib * APGP_ROW_BLOCK + 16 * s + (lane & 15);
        long long col = kc * APGP_K_CHUNK + 4 * kk + (lane >> 4);
        double v = 0.0;
        if (row < n && col <= row) v = W[row * ldw + col];
        out[e] = v;
    }
}

extern "C" int apgp_trtri_pack(const double* L, int64_t n, int64_t ldl, double* work,
                               double* packed, double* winv_dense, void* stream) {
    APGP_CHECK_ARG(L && work, "null pointer");
    APGP_CHECK_ARG(n >= 1 && n <= APGP_MAX_N && ldl >= n, "n >= 1 and ldl >= n required");
    hipStream_t s = (hipStream_t)stream;
    const long long np = apgp_round_up(n, 64);
    const int nb = (int)(np / 64);
    double* W = work;
    double* T = work + np * np;
    if (hipMemsetAsync(W, 0, sizeof(double) * np * np, s) != hipSuccess) {
        apgp_set_error("apgp_trtri_pack: memset failed");
        return -2;
    }
    hipLaunchKernelGGL(trtri_diag_kernel, dim3(nb), dim3(64), 0, s, L, (long long)n, (long long)ldl, W, np);
    for (int lev = 1; lev < nb; lev *= 2) {
        MergeArgs a;
        a.L = L; a.W = W; a.T = T; a.ldl = ldl; a.ldw = np; a.n = n; a.nb = nb; a.s = lev;
        int pairs = (nb + 2 * lev - 1) / (2 * lev);
        // two complementary tiles per workgroup once the launch fills the chip (trtri_merge_kernel);
        // below that the second tile would only lengthen the launch
        a.pair = (lev > 1 && (long long)lev * lev * pairs >= 512) ? 1 : 0;
        const int half = a.pair ? lev / 2 : lev;
        a.phase = 1;
        hipLaunchKernelGGL(trtri_merge_kernel, dim3(half, lev, pairs), dim3(256), 0, s, a);
        a.phase = 2;
        hipLaunchKernelGGL(trtri_merge_kernel, dim3(lev, half, pairs), dim3(256), 0, s, a);
    }
    if (packed) {
        long long nrb = apgp_npad(n) / APGP_ROW_BLOCK;
        long long ntiles = (APGP_ROW_BLOCK / APGP_K_CHUNK) * nrb * (nrb + 1) / 2;
        hipLaunchKernelGGL(pack_linv_kernel, dim3((unsigned)ntiles), dim3(256), 0, s, W, np,
                           (long long)n, packed);
    }
    if (winv_dense) {
        if (hipMemcpy2DAsync(winv_dense, n * sizeof(double), W, np * sizeof(double),
                             n * sizeof(double), n, hipMemcpyDeviceToDevice, s) != hipSuccess) {
            apgp_set_error("apgp_trtri_pack: copy of dense inverse failed");
            return -2;
        }
    }
    APGP_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------
// packed training stream: rows [ x_k * sc (Dpad) | alpha_k | 0 ], zero past N
// ---------------------------------------------------------------------------
struct PackTrainArgs {
    const double* X;
    const double* alpha;
    double* xs;
    long long n, npad;
    KernConst kc;
};

__global__ __launch_bounds__(256) void pack_train_kernel(PackTrainArgs a) {
    const int stride = a.kc.dpad + 2;
    long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= a.npad * stride) return;
    long long k = e / stride;
    int d = (int)(e % stride);
    double v = 0.0;
    if (k < a.n) {
        if (d < a.kc.ndim) v = a.X[k * a.kc.ndim + d] * a.kc.sc[d];
        else if (d == a.kc.dpad) v = a.alpha[k];
    }
    a.xs[e] = v;
}

extern "C" int apgp_pack_train(const double* X, const double* alpha, int64_t n,
                               const apgp_kernel_t* kern, double* xs, void* stream) {
    APGP_CHECK_ARG(X && alpha && xs && kern, "null pointer");
    APGP_CHECK_ARG(n >= 1 && n <= APGP_MAX_N, "n >= 1 required");
    PackTrainArgs a;
    APGP_CHECK_ARG(apgp_make_kernconst(kern, &a.kc) == 0, "kernel parameters");
    a.X = X; a.alpha = alpha; a.xs = xs; a.n = n; a.npad = apgp_npad(n);
    long long total = a.npad * (a.kc.dpad + 2);
    hipLaunchKernelGGL(pack_train_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, a);
    APGP_CHECK_LAUNCH();
    return 0;
}
