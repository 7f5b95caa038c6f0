// Blocked right-looking Cholesky factorisation (lower, row-major, in place) for
// the GP fit: replaces scipy.linalg.cholesky inside george BasicSolver.compute
// (every gpUtils._nll evaluation, gpUtils.py:74-78, and GP.compute,
// gpUtils.py:178; approx.py:717).  rocSOLVER's dpotrf reaches ~2 TFLOP/s at
// N = 4096 on MI355X (11.8 ms, dozens of tiny kernels); this version is two
// launches per 64-column block step:
//   panel : every workgroup (two wavefronts) re-factorises the 64x64 diagonal
//           block in registers (redundantly -- it is the critical path either
//           way, and it saves a launch + a grid-wide dependency) while its second
//           wavefront solves 64 rows of the panel against it (row-per-lane substitution);
//   update: trailing lower-triangle tiles A_ik -= L_ij L_kj^T (64x64x64 MFMA
//           f64 tiles; HBM-bound: each tile is read-modify-written once).
// info follows LAPACK: 0 = OK, k > 0 = leading minor of order k not positive
// definite (first failing pivot).
#include "apgp_common.h"
#include <type_traits>
#include <utility>

// compile-time loop: the body sees its index as a constant expression, so the register
// arrays below are indexed statically whatever hipcc's unroll heuristics decide
template <int... Is, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, F&& f) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

#define PB 64

struct PotrfArgs {
    double* A;
    double* rhs;         // optional right-hand side carried through the factorisation
    long long n, lda;
    long long j0;        // first column of the current block
    double shift;        // rhs is used as (rhs - shift); applied at block step 0 .. as read
    int* info;
    // batched factorisation (apgp_nll_eval_batch): blockIdx.y selects the matrix
    long long batch_A, batch_rhs;   // element strides between consecutive matrices / right-hand sides
};

// matrix of this workgroup in a batched launch (gridDim.y = batch size; strides 0 otherwise)
__device__ __forceinline__ void potrf_select(PotrfArgs& a) {
    a.A += (long long)blockIdx.y * a.batch_A;
    if (a.rhs) a.rhs += (long long)blockIdx.y * a.batch_rhs;
    a.info += blockIdx.y;
}

// Panel step of block column j, two wavefronts per workgroup; workgroup b owns the 64 panel rows
// [j0 + 64 + 64 b, +64) and re-factorises the 64x64 diagonal block itself (it is the critical path
// either way; redundancy saves a launch and a grid-wide dependency).
//
// Wavefront 0 factorises the diagonal block in REGISTERS: lane i holds row i (64 doubles),
// right-looking, fully unrolled.  Per pivot k: the pivot is a v_readlane, its reciprocal square
// root a v_rsq_f64 + two Newton steps (no IEEE sqrt + divide on the critical path), the scaled
// column goes through a 512-byte LDS line and comes back as broadcast ds_read_b128 for the
// rank-1 update.  One wavefront, in-order LDS: no barrier anywhere.  It publishes the factor as it
// goes -- after pivot k column k of L_jj (Ls[.][k]), 1/L_kk (invd[k]) and the progress counter.
// Wavefront 1 solves the workgroup's 64 panel rows, x L_jj^T = a, one row per lane against
// broadcast reads of L_jj (four partial sums per dot product, the L values requested one stage
// ahead), CONCURRENTLY and two pivots behind: row k of the solve needs row k of L_jj, final after
// pivot k-1, and the prefetch of the next stage needs invd[k+1].  The panel solve (21.5 k cycles)
// thus hides behind the factorisation (46 k) instead of following it (round 1: one wavefront did
// both in turn, 89 k cycles per step; now ~62 k: 4.72 -> 4.05 ms at N = 4096, 0.98 -> 0.78 ms at
// N = 1152).  Same arithmetic in the same order per element as the one-wavefront form.
//
// Scheduling fences for the straight-line code: the asm memory clobber stops the SelectionDAG
// from hoisting the (address-independent) LDS reads of later steps, the sched_barrier stops the
// machine scheduler -- without both, hundreds of reads are in flight at once and the 64-double
// register rows spill.
#ifdef APGP_PANEL_TIMING
__device__ unsigned long long apgp_panel_stamps[8];
#define PANEL_STAMP(i) do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && a.j0 == 2048) apgp_panel_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PANEL_STAMP(i) do { } while (0)
#endif
// broadcast of lane `src` (compile-time constant): two v_readlane_b32 (a few cycles) --
// __shfl with a constant lane still goes through ds_bpermute (an LDS round trip)
__device__ __forceinline__ double bcast_lane(double v, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}
// staging of the panel solve: 16-column groups of row k before its last (k & 3) columns
__host__ __device__ constexpr int trsm_ng(int k) { return ((k & ~3) + 15) / 16; }
__host__ __device__ constexpr int trsm_gcount(int k, int g) {
    int c = g;
    for (int q = 0; q < k; ++q) c += trsm_ng(q);
    return c;
}
#define PANEL_FENCE() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
// (a C++ volatile store through a generic pointer becomes a FLAT system-scope store plus
// s_waitcnt vmcnt(0) -- hundreds of cycles per pivot on the critical path; this is the LDS store)
__device__ __forceinline__ void lds_store_volatile(int* p, int v) {
    asm volatile("ds_write_b32 %0, %1" : : "v"((unsigned)(size_t)(__attribute__((address_space(3))) int*)p), "v"(v) : "memory");
}
__device__ __forceinline__ int lds_load_volatile(const int* p) {
    int v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned)(size_t)(__attribute__((address_space(3))) const int*)p) : "memory");
    return v;
}
__global__ __launch_bounds__(128) void potrf_panel_kernel(PotrfArgs a) {
    potrf_select(a);
    __shared__ __attribute__((aligned(16))) double col[2][PB];
    __shared__ __attribute__((aligned(16))) double Ls[PB][PB + 2];
    __shared__ __attribute__((aligned(16))) double invd[PB];
    __shared__ double zblk[PB];
    __shared__ int prog;                 // pivots published so far; PB + 1 once zblk is published too
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long j0 = a.j0;
    const int bs = (int)((a.n - j0) < PB ? (a.n - j0) : PB);
    const long long row = j0 + PB + (long long)blockIdx.x * PB + lane;
    const bool has_row = row < a.n;
    if (threadIdx.x == 0) prog = 0;
    __syncthreads();

    if (wv == 0) {
        // ---------------- wavefront 0: the diagonal block ----------------
        PANEL_STAMP(0);
        double ar[PB];
        if (bs == PB) {
            // coalesced: lane = column (one 512-byte row of the block per load), transposed to
            // lane = row through Ls (free until the first pivot is published).  A row-per-lane load
            // touches 64 cache lines per instruction: ~10 k cycles at the head of every step, ~6 k so.
            const double* src = a.A + j0 * a.lda + j0 + lane;
#pragma unroll
            for (int r0 = 0; r0 < PB; r0 += 16) {
                double t[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) t[r] = src[(long long)(r0 + r) * a.lda];
#pragma unroll
                for (int r = 0; r < 16; ++r) Ls[r0 + r][lane] = t[r];
            }
            PANEL_FENCE();
#pragma unroll
            for (int k = 0; k < PB; k += 2) {
                const f64x2 v = *(const f64x2*)(&Ls[lane][k]);
                ar[k] = k <= lane ? v.x : 0.0;
                ar[k + 1] = k + 1 <= lane ? v.y : 0.0;
            }
        } else {
            const double* src = a.A + (j0 + (lane < bs ? lane : 0)) * a.lda + j0;
#pragma unroll
            for (int k = 0; k < PB; ++k) {
                double v = 0.0;
                if (lane < bs && k <= lane) v = src[k];
                ar[k] = (lane < bs && k <= lane) ? v : (k == lane ? 1.0 : 0.0);
            }
        }
        PANEL_FENCE();
        PANEL_STAMP(1);
        static_for<PB>([&](auto kc_) {
            constexpr int k = decltype(kc_)::value;
            const double piv = bcast_lane(ar[k], k);
            const bool bad = !(piv > 0.0) || !(piv < INFINITY);
            if (bad && blockIdx.x == 0 && lane == 0) atomicMin((unsigned int*)a.info, (unsigned int)(j0 + k + 1));
            const double pv = bad ? 1.0 : piv;
            double r = __builtin_amdgcn_rsq(pv);
            r = r * fma(-0.5 * pv * r, r, 1.5);
            r = r * fma(-0.5 * pv * r, r, 1.5);
            double d = pv * r;
            d = fma(0.5 * r, fma(-d, d, pv), d);
            const double inv = r;
            invd[k] = inv;                           // (uniform value, every lane stores it: no divergent branch)
            ar[k] = lane == k ? d : ar[k] * inv;
            double* cb = col[k & 1];
            cb[lane] = ar[k];
            Ls[lane][k] = k <= lane ? ar[k] : 0.0;      // column k of L_jj for the panel solve
            lds_store_volatile(&prog, k + 1);           // (every lane, same word; same wavefront: LDS stores stay in order)
            const double lik = ar[k];
            if (((k + 1) & 1) && k + 1 < PB) {
                const double ljk = cb[k + 1];
                ar[k + 1] = fma(-lik, ljk, ar[k + 1]);
                asm volatile("" : "+v"(ar[k + 1]));
            }
            // (one group per pivot: this wavefront has a SIMD's whole register file behind it -- the
            // workgroup's two wavefronts sit on different SIMDs --, so the column's broadcast values
            // are all requested at once, up to 62 registers; the fences still keep hipcc from
            // requesting the columns of LATER pivots up here)
#pragma unroll
            for (int j0g = (k + 2) & ~1; j0g + 1 < PB; j0g += 64) {
                f64x2 l2[32];
#pragma unroll
                for (int g = 0; g < 32; ++g)
                    if (j0g + 2 * g + 1 < PB) l2[g] = *(const f64x2*)(cb + j0g + 2 * g);
                PANEL_FENCE();
#pragma unroll
                for (int g = 0; g < 32; ++g)
                    if (j0g + 2 * g + 1 < PB) {
                        ar[j0g + 2 * g] = fma(-lik, l2[g].x, ar[j0g + 2 * g]);
                        ar[j0g + 2 * g + 1] = fma(-lik, l2[g].y, ar[j0g + 2 * g + 1]);
                        asm volatile("" : "+v"(ar[j0g + 2 * g]), "+v"(ar[j0g + 2 * g + 1]));
                    }
                PANEL_FENCE();
            }
        });
        PANEL_STAMP(2);
        if (blockIdx.x == 0 && lane < bs) {
            double* dst = a.A + (j0 + lane) * a.lda + j0;
#pragma unroll
            for (int k = 0; k < PB; ++k)
                if (k <= lane) dst[k] = ar[k];
        }
        if (a.rhs) {
            double ri = (lane < bs) ? a.rhs[j0 + lane] : 0.0;
            static_for<PB>([&](auto kc_) {
                constexpr int k = decltype(kc_)::value;
                const double zk = bcast_lane(ri, k) * invd[k];
                ri = lane == k ? zk : (lane > k ? fma(-ar[k], zk, ri) : ri);
            });
            zblk[lane] = ri;
            if (blockIdx.x == 0 && lane < bs) a.rhs[j0 + lane] = ri;
        }
        lds_store_volatile(&prog, PB + 1);
        PANEL_STAMP(3);
        return;
    }

    // ---------------- wavefront 1: the 64 panel rows of this workgroup ----------------
    double x[PB];
    {
        const double* src = a.A + (has_row ? row : j0) * a.lda + j0;
#pragma unroll
        for (int k = 0; k < PB; ++k) x[k] = (bs == PB && has_row) ? src[k] : 0.0;
    }
    PANEL_FENCE();
    auto wait_prog = [&](int need) {
        while (lds_load_volatile(&prog) < need) __builtin_amdgcn_s_sleep(1);
        PANEL_FENCE();
    };
    f64x2 lq[2][8], tq[2][2];
    double ivq[2];
    auto load_wide = [&](auto kt, auto gt) {
        constexpr int k = decltype(kt)::value, g = decltype(gt)::value;
        constexpr int buf = trsm_gcount(k, g) & 1;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (16 * g + 2 * i + 1 < (k & ~3)) lq[buf][i] = *(const f64x2*)(&Ls[k][16 * g + 2 * i]);
    };
    auto load_tail = [&](auto kt) {
        constexpr int k = decltype(kt)::value;
        tq[k & 1][0] = *(const f64x2*)(&Ls[k][k & ~3]);
        tq[k & 1][1] = *(const f64x2*)(&Ls[k][(k & ~3) + 2]);
        ivq[k & 1] = invd[k];
    };
    auto load_first_of = [&](auto kt) {
        constexpr int k = decltype(kt)::value;
        if constexpr (k < PB) {
            if constexpr (trsm_ng(k) > 0) load_wide(kt, std::integral_constant<int, 0>{});
            else load_tail(kt);
        }
    };
    wait_prog(1);                          // row 0 of L_jj and invd[0]
    load_first_of(std::integral_constant<int, 0>{});
    static_for<PB>([&](auto kc_) {
        constexpr int k = decltype(kc_)::value;
        // rows k and (prefetch) k+1 of L_jj with their reciprocals: pivots 0 .. k+1 published
        wait_prog(k + 2 < PB ? k + 2 : PB);
        double s0 = x[k], s1 = 0.0, s2 = 0.0, s3 = 0.0;
        static_for<4>([&](auto gc_) {
            constexpr int g = decltype(gc_)::value;
            if constexpr (g < trsm_ng(k)) {
                if constexpr (g + 1 < trsm_ng(k)) load_wide(kc_, std::integral_constant<int, g + 1>{});
                else load_tail(kc_);
                PANEL_FENCE();
                constexpr int buf = trsm_gcount(k, g) & 1;
#pragma unroll
                for (int i = 0; i < 8; i += 2)
                    if (16 * g + 2 * i + 3 < k) {
                        const int m = 16 * g + 2 * i;
                        s0 = fma(-x[m], lq[buf][i].x, s0);
                        s1 = fma(-x[m + 1], lq[buf][i].y, s1);
                        s2 = fma(-x[m + 2], lq[buf][i + 1].x, s2);
                        s3 = fma(-x[m + 3], lq[buf][i + 1].y, s3);
                    }
                asm volatile("" : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3));
                PANEL_FENCE();
            }
        });
        load_first_of(std::integral_constant<int, k + 1>{});
        PANEL_FENCE();
        {
            const double tl[4] = {tq[k & 1][0].x, tq[k & 1][0].y, tq[k & 1][1].x, tq[k & 1][1].y};
#pragma unroll
            for (int m = k & ~3; m < k; ++m) s0 = fma(-x[m], tl[m - (k & ~3)], s0);
        }
        x[k] = ((s0 + s1) + (s2 + s3)) * ivq[k & 1];
        asm volatile("" : "+v"(x[k]));
        PANEL_FENCE();
    });
    PANEL_STAMP(4);
    if (has_row) {
        double* ap = a.A + row * a.lda + j0;
#pragma unroll
        for (int k = 0; k < PB; ++k) ap[k] = x[k];
        if (a.rhs) {
            wait_prog(PB + 1);                 // zblk published
            double d0 = 0.0, d1 = 0.0;
#pragma unroll
            for (int k = 0; k < PB; k += 2) {
                d0 = fma(x[k], zblk[k], d0);
                d1 = fma(x[k + 1], zblk[k + 1], d1);
            }
            a.rhs[row] -= d0 + d1;
        }
    }
    PANEL_STAMP(5);
}

// trailing update: tile (bi, bk), bi >= bk, of the blocks below/right of column block j:
// A[ri.., rk..] -= L[ri.., j0..j0+64) * L[rk.., j0..j0+64)^T
__global__ __launch_bounds__(256) void potrf_update_kernel(PotrfArgs a) {
    potrf_select(a);
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
    info += blockIdx.x;
    if (*(unsigned int*)info == 0xffffffffu) *info = 0;
}

// `batch` matrices A + b * batch_A (right-hand sides y - shifts[b] -> z + b * n) factorised by
// the same launches: gridDim.y = batch.  The Cholesky of one small matrix is a chain of
// latency-bound steps that leaves most of the chip idle, so a batch costs little more than one.
static int potrf_run(double* A, int64_t n, int64_t lda, int64_t batch, int64_t batch_A, const double* y,
                     const double* shifts, double* z, int32_t* info_dev, hipStream_t s) {
    // info = UINT_MAX means "no failure yet"; normalised to 0 by the caller-visible finish kernel
    if (hipMemsetAsync(info_dev, 0xff, sizeof(int32_t) * batch, s) != hipSuccess) {
        apgp_set_error("apgp_potrf: memset failed");
        return -2;
    }
    PotrfArgs a;
    a.A = A; a.rhs = z; a.n = n; a.lda = lda; a.shift = 0.0; a.info = info_dev;
    a.batch_A = batch_A; a.batch_rhs = n;
    if (z)
        for (int64_t b = 0; b < batch; ++b)
            hipLaunchKernelGGL(potrf_rhs_init_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, y, shifts[b],
                               z + b * n, (long long)n);
    const long long nb = (n + PB - 1) / PB;
    for (long long jb = 0; jb < nb; ++jb) {
        a.j0 = jb * PB;
        const long long below = n - (a.j0 + PB);
        const unsigned pg = below > 0 ? (unsigned)((below + PB - 1) / PB) : 1u;
        hipLaunchKernelGGL(potrf_panel_kernel, dim3(pg, (unsigned)batch), dim3(2 * PB), 0, s, a);
        if (below > 0) {
            const long long tb = (below + PB - 1) / PB;
            hipLaunchKernelGGL(potrf_update_kernel, dim3((unsigned)(tb * (tb + 1) / 2), (unsigned)batch), dim3(256), 0, s, a);
        }
    }
    hipLaunchKernelGGL(potrf_finish_kernel, dim3((unsigned)batch), dim3(1), 0, s, info_dev);
    APGP_CHECK_LAUNCH();
    return 0;
}

extern "C" int apgp_potrf(double* A, int64_t n, int64_t lda, const double* y, double shift, double* z,
                          int32_t* info_dev, void* stream) {
    APGP_CHECK_ARG(A && info_dev, "null pointer");
    APGP_CHECK_ARG((y == NULL) == (z == NULL), "y and z must be given together");
    APGP_CHECK_ARG(n >= 1 && lda >= n, "n >= 1 and lda >= n required");
    return potrf_run(A, n, lda, 1, 0, y, &shift, z, info_dev, (hipStream_t)stream);
}

// One gpUtils._nll evaluation (gpUtils.py:46-80) as ONE library call: Gram matrix ->
// Cholesky with the forward solve riding along -> summary record -> 40-byte D2H -> one
// stream synchronisation.  Powell / Nelder-Mead call this hundreds to thousands of times
// per fit; at N = 50 the separate calls' host overhead was as long
// as the kernels themselves.
extern "C" int apgp_gram(const double* X, int64_t n, const apgp_kernel_t* kern, double* K, int64_t ldk, void* stream);
extern "C" int apgp_fit_summary(const double* L, int64_t n, int64_t ldl, const double* z, const int32_t* info_dev,
                                double* out5, void* stream);
extern "C" int apgp_nll_eval(const double* X, int64_t n, const apgp_kernel_t* kern, const double* y, double mean,
                             double* K, double* z, int32_t* info_dev, double* out5_dev, double* out5_host,
                             void* stream) {
    APGP_CHECK_ARG(X && kern && y && K && z && info_dev && out5_dev && out5_host, "null pointer");
    int rc = apgp_gram(X, n, kern, K, n, stream);
    if (rc != 0) return rc;
    rc = apgp_potrf(K, n, n, y, mean, z, info_dev, stream);
    if (rc != 0) return rc;
    rc = apgp_fit_summary(K, n, n, z, info_dev, out5_dev, stream);
    if (rc != 0) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemcpyAsync(out5_host, out5_dev, 5 * sizeof(double), hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess) {
        apgp_set_error("apgp_nll_eval: D2H copy failed");
        return -2;
    }
    return 0;
}

// `batch` _nll evaluations at different hyper-parameters of the SAME training set in one call
// (several hyper-vectors per launch for optimizeGP's restarts):
// batch Gram launches, ONE batched Cholesky (gridDim.y = batch), batch summary launches, one
// 40 * batch byte copy, one synchronisation.  Every matrix goes through exactly the code path
// of apgp_nll_eval, so the values are bit-identical to `batch` single calls.
extern "C" int apgp_nll_eval_batch(const double* X, int64_t n, int64_t batch, const apgp_kernel_t* kerns,
                                   const double* y, const double* means, double* K, double* z,
                                   int32_t* info_dev, double* out5_dev, double* out5_host, void* stream) {
    APGP_CHECK_ARG(X && kerns && y && means && K && z && info_dev && out5_dev && out5_host, "null pointer");
    APGP_CHECK_ARG(batch >= 1 && batch <= 65535, "1 <= batch <= 65535 required");
    APGP_CHECK_ARG(n >= 1, "n >= 1 required");
    int rc;
    for (int64_t b = 0; b < batch; ++b)
        if ((rc = apgp_gram(X, n, kerns + b, K + b * n * n, n, stream)) != 0) return rc;
    if ((rc = potrf_run(K, n, n, batch, n * n, y, means, z, info_dev, (hipStream_t)stream)) != 0) return rc;
    for (int64_t b = 0; b < batch; ++b)
        if ((rc = apgp_fit_summary(K + b * n * n, n, n, z + b * n, info_dev + b, out5_dev + 5 * b, stream)) != 0) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemcpyAsync(out5_host, out5_dev, 5 * sizeof(double) * batch, hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess) {
        apgp_set_error("apgp_nll_eval_batch: D2H copy failed");
        return -2;
    }
    return 0;
}
