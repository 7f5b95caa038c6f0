// K5/K6 "sweep": fused candidate tile -> k* row -> mu -> L^-1 k* contraction on
// the f64 matrix cores -> predictive variance -> AGP / BAPE / Jones utility ->
// arg-min.  The batched counterpart of
//   george.GP.predict(y, t, return_var=True)        (utility.py:131,178,224)
//   utility.AGPUtility / BAPEUtility / JonesUtility (utility.py:99-250)
//   utility.minimizeObjective's arg-min             (utility.py:369-371)
//
// Data flow per workgroup (256 threads = 4 wavefronts, one per SIMD, 64 candidates):
//   * wavefront w owns candidates [16 w, 16 w + 16) (one MFMA column block);
//     the scaled candidate coordinates live in registers for the whole kernel.
//   * the packed factor W = L^-1 is streamed tile by tile (512 rows x 16 k,
//     64 KiB, A-fragment order) L2/MALL -> registers -> LDS, double buffered,
//     shared by the four wavefronts (one barrier per tile).
//   * K* is NEVER materialised: each lane generates the k*(t_m, x_k) value it
//     must feed as the MFMA B operand (lane -> candidate lane&15, k lane>>4)
//     on the VALU (D sub + D fma + table-driven exp).
//   * V = W K*^T is accumulated 512 rows x 16 candidates per wavefront in 128
//     v_mfma_f64_4x4x4_4b_f64 accumulators (AGPRs); at the end of a row block
//     the squares are folded into a per-candidate sum; V is never stored.
//   * mu is a VALU by-product of the last row block (which visits every k).
#include "apgp_common.h"
#include <stdlib.h>
#include <type_traits>
#include <string.h>

#define SW_ROWS APGP_ROW_BLOCK       // 512 rows of W per tile
#define SW_KC APGP_K_CHUNK           // 16 k per tile
#define SW_TILE (SW_ROWS * SW_KC)    // doubles per tile (64 KiB)
#define SW_CAND 64                   // candidates per workgroup (16 per wavefront)
#define SW_THREADS 256

struct SweepArgs {
    const double* T;
    const double* linv;
    const double* xs;
    const unsigned char* mask;
    double* mu;
    double* var;
    double* u;
    double* part_u;
    long long* part_i;
    long long m, idx_offset;
    int ndim, nrb, kind, has_box, n;
    double mean, amp, zeta, ybest;
    double sc[APGP_MAX_DIM], lo[APGP_MAX_DIM], hi[APGP_MAX_DIM];
    unsigned long long* dbg;   // phase cycle counters (APGP_SWEEP_TIMING=1 builds only)
};

__device__ __forceinline__ double util_value(int kind, double mu, double var, double zeta,
                                             double ybest) {
    if (kind == APGP_UTIL_AGP) {
        // utility.py:136  -(mu + 0.5*log(2*pi*e*var)); var < 0 -> NaN as in NumPy
        return -(mu + 0.5 * log(2.0 * M_PI * M_E * var));
    } else if (kind == APGP_UTIL_BAPE) {
        // utility.py:183 with logsubexp(var, 0) (utility.py:85-88):
        // var <= 0 -> -inf -> utility +inf; else var + log(1 - exp(-var))
        double lse = (var <= 0.0) ? -INFINITY : var + log(1.0 - exp(0.0 - var));
        return -((2.0 * mu + var) + lse);
    } else {
        // utility.py:229-244; std <= 0 or NaN -> 0.0
        double sd = sqrt(var);
        if (sd > 0.0) {
            double imp = mu - ybest - zeta;
            double z = imp / sd;
            double cdf = 0.5 * erfc(-z * M_SQRT1_2);
            double pdf = exp(-0.5 * z * z) * 0.3989422804014326779399461;
            return -(imp * cdf + sd * pdf);
        }
        return 0.0;
    }
}

__device__ __forceinline__ void best_merge(double& bu, long long& bi, double u, long long i) {
    // NaN and +inf (inadmissible) never win; ties resolve to the lowest global index
    if (i >= 0 && u < INFINITY && (u < bu || (u == bu && (bi < 0 || i < bi)))) { bu = u; bi = i; }
}

// Matrix-core instruction choice (measured on MI355X, tools/mfma_peak.hip and
// tools/mfma_probe.hip):
//   v_mfma_f64_16x16x4_f64     36 TF/chip  (~138 cycles, 7.4 MAC/clk/SIMD)
//   v_mfma_f64_4x4x4_4b_f64    70-75 TF    (16.5 cycles, 15.5 MAC/clk/SIMD = FP64 FMA peak)
//   v_fma_f64 (VALU)           64-70 TF, and MFMA + VALU f64 do NOT overlap: they
//                              share the DP pipes (sum stays ~70 TF for any mix).
// So the contraction uses the 4x4x4 four-block instruction.  Its blocks are
// independent (CBSZ/ABID broadcast is ignored for f64 on gfx950), so a
// 16 x 16 x 4 product takes four instructions whose A operand is the same
// 16 x 4 fragment with its 4-row groups rotated across the blocks:
//   A lane = i + 4 b + 16 k,  B lane = j + 4 b + 16 k,  D lane = j + 4 b + 16 i.
// The candidate of a lane is lane & 15 for every rotation, which is all the
// variance reduction needs (sum over rows of V^2).
// Because K* generation costs DP cycles too, each generated value must feed as
// many rows as possible: a wavefront owns 512 rows x 16 candidates (128 f64
// accumulators per lane = the whole AGPR file), so one generated B value feeds
// 128 MFMAs; one wavefront per SIMD, everything latency-critical is prefetched.
template <int DPAD, bool TIMING = false>
__global__ __launch_bounds__(SW_THREADS, 1) void sweep_kernel(SweepArgs a) {
    unsigned long long tph[5] = {0, 0, 0, 0, 0}, tq = 0;
#define SW_TICK(i) do { if (TIMING) { unsigned long long n_ = __builtin_amdgcn_s_memtime(); tph[i] += n_ - tq; tq = n_; } } while (0)
    constexpr int XS = DPAD + 2;
    constexpr int RS = SW_ROWS / 16;           // 16-row sub-blocks per tile (32)
    constexpr int NKK = SW_KC / 4;             // k-steps per tile (4)
    static_assert(NKK == 4 && (RS % 2) == 0, "tile layout");
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* Abuf = smem;                       // 2 x SW_TILE
    double* Xbuf = smem + 2 * SW_TILE;         // 2 x SW_KC x XS
    double* Etab = Xbuf + 2 * SW_KC * XS;      // exp table
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int cl = lane & 15, kq = lane >> 4;
    apgp_exp_tab_load(Etab);

    // ---- candidate of this lane -------------------------------------------
    double tt[DPAD];
    const long long crow = (long long)blockIdx.x * SW_CAND + w * 16 + cl;
    bool adm = crow < a.m;
    bool has_nan = false;
    {
        const bool inb = adm;
#pragma unroll
        for (int d = 0; d < DPAD; ++d) {
            double v = 0.0;
            if (inb && d < a.ndim) {
                v = a.T[crow * a.ndim + d];
                if (a.has_box && !(v >= a.lo[d] && v <= a.hi[d])) adm = false;
                if (v != v) has_nan = true;
            }
            tt[d] = v * a.sc[d];
        }
        if (inb && a.mask && a.mask[crow] == 0) adm = false;
    }

    double qpart = 0.0, mupart = 0.0;
    // lane index of the A element this lane feeds for rotation r
    int rot[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) rot[r] = (lane & ~12) | ((((lane >> 2) + r) & 3) << 2);

    constexpr int CPB = SW_ROWS / SW_KC;       // k-chunks per row-block width (32)
    const int kc_lim = (a.n + SW_KC - 1) / SW_KC;     // chunks that hold real columns
    const int sb_lim = (a.n + 15) / 16;               // sub-blocks (global) that hold real rows

    // Tile staging through registers (global_load_dwordx4 early, ds_write_b128
    // late): an LDS-DMA (global_load_lds) costs ~55 issue cycles per KiB on the
    // issuing wavefront, and with one wavefront per SIMD nothing hides that.
    // The 64 KiB tile moves in two halves of 8 x 16 B per thread.
    constexpr int HALF16 = SW_TILE / 2 / 2;          // 16-byte pieces per half tile (2048)
    constexpr int NST = HALF16 / SW_THREADS;         // pieces per thread per half (8)
    constexpr int XCHUNK16 = SW_KC * XS / 2;         // 16-byte pieces of the x chunk
    static_assert(XCHUNK16 <= SW_THREADS, "x chunk staging");
    f64x2 st[NST];
    f64x2 xpend;
    auto half_load = [&](long long tile, int half) {
        const f64x2* g = (const f64x2*)(a.linv + tile * SW_TILE) + half * HALF16 + t;
#pragma unroll
        for (int i = 0; i < NST; ++i) st[i] = g[i * SW_THREADS];
    };
    auto half_store = [&](int buf, int half) {
        f64x2* l = (f64x2*)(Abuf + buf * SW_TILE) + half * HALF16 + t;
#pragma unroll
        for (int i = 0; i < NST; ++i) l[i * SW_THREADS] = st[i];
    };
    // two pieces (j0, j0+1) of a half tile: the staging traffic is dealt two
    // instructions per sub-block pair into the MFMA stream (non-fp64 instructions
    // cost ~1 cycle there; a cluster of them idles the matrix pipe)
    auto piece_load = [&](long long tile, int half, int j0) {
        const f64x2* g = (const f64x2*)(a.linv + tile * SW_TILE) + half * HALF16 + t;
        st[j0] = g[j0 * SW_THREADS];
        st[j0 + 1] = g[(j0 + 1) * SW_THREADS];
    };
    auto piece_store = [&](int buf, int half, int j0) {
        f64x2* l = (f64x2*)(Abuf + buf * SW_TILE) + half * HALF16 + t;
        l[j0 * SW_THREADS] = st[j0];
        l[(j0 + 1) * SW_THREADS] = st[j0 + 1];
    };
    // NOTE: every load below is UNCONDITIONAL (indices are clamped instead): a load
    // under an `if` makes hipcc merge "loaded or old" values and drain vmcnt(0) at
    // the join, exposing the full memory latency (cdna guide, .s-level trap (c)).
    const int xt = t < XCHUNK16 ? t : 0;
    auto x_load = [&](int kc) {
        xpend = *((const f64x2*)(a.xs + (long long)kc * SW_KC * XS) + xt);
    };
    auto x_store = [&](int buf) {
        if (t < XCHUNK16) *((f64x2*)(Xbuf + buf * SW_KC * XS) + t) = xpend;
    };

    // tile sequence: row block ib, chunks kc = 0 .. min((ib+1)*CPB, kc_lim) - 1;
    // packed tile index = CPB*ib*(ib+1)/2 + kc
    auto tile_index = [&](int ib, int kc) { return (long long)CPB * ib * (ib + 1) / 2 + kc; };
    auto nkc_of = [&](int ib) { const int v = CPB * (ib + 1); return v < kc_lim ? v : kc_lim; };

    half_load(0, 0); half_store(0, 0);
    half_load(0, 1); half_store(0, 1);
    x_load(0); x_store(0);
    __syncthreads();

    int buf = 0;
    for (int ib = 0; ib < a.nrb; ++ib) {
        const int nkc = nkc_of(ib);
        double acc[RS][4];
#pragma unroll
        for (int s = 0; s < RS; ++s)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[s][r] = 0.0;
        // mu = k* . alpha: only the last row block's pass (which visits every k)
        // survives -- mupart is reset at the top of each pass
        mupart = 0.0;
        int kc = 0;
        auto do_tile = [&](auto diag_tag) {
            constexpr bool DIAG = decltype(diag_tag)::value;
            if (TIMING) tq = __builtin_amdgcn_s_memtime();
            // next tile in the sequence (possibly the first of the next row block)
            int nib = ib, nk = kc + 1;
            if (nk >= nkc) { nib = ib + 1; nk = 0; }
            const bool more = nib < a.nrb;
            if (!more) { nib = 0; nk = 0; }          // harmless dummy prefetch on the last tile
            const long long ntile = tile_index(nib, nk);
            x_load(nk);
            const double* Xb = Xbuf + buf * SW_KC * XS;
            const f64x2* A2 = (const f64x2*)(Abuf + buf * SW_TILE);
            SW_TICK(0);
            // ---- generate the B operands of the tile's four k-steps:
            //      k*(candidate cl, x_k), k = kc*16 + 4 kk + kq
            double bfv[NKK];
            {
                // written across the four k-steps so the four dependent fp64 chains
                // interleave; two partial sums per chain halve its length
                double s2[NKK], s3[NKK], al[NKK];
#pragma unroll
                for (int kk = 0; kk < NKK; ++kk) { s2[kk] = 0.0; s3[kk] = 0.0; }
#pragma unroll
                for (int d = 0; d < DPAD; d += 2) {
#pragma unroll
                    for (int kk = 0; kk < NKK; ++kk) {
                        const double* xr = Xb + (kk * 4 + kq) * XS;
                        const double df0 = tt[d] - xr[d];
                        const double df1 = tt[d + 1] - xr[d + 1];
                        s2[kk] = fma(df0, df0, s2[kk]);
                        s3[kk] = fma(df1, df1, s3[kk]);
                    }
                }
                double ex[NKK];
#pragma unroll
                for (int kk = 0; kk < NKK; ++kk) {
                    ex[kk] = -(s2[kk] + s3[kk]);
                    al[kk] = Xb[(kk * 4 + kq) * XS + DPAD];
                }
                apgp_exp4(ex, bfv, Etab);
                // the amplitude multiplies the exponential (folding log(amp) into the
                // exponent would perturb every entry by ~|log amp| ulps, which matters
                // once cond(K) approaches 1/eps)
#pragma unroll
                for (int kk = 0; kk < NKK; ++kk) {
                    bfv[kk] *= a.amp;
                    mupart = fma(bfv[kk], al[kk], mupart);
                }
            }
            // keep the A-fragment prefetch below the generation phase (register pressure)
            __builtin_amdgcn_sched_barrier(0);
            if (TIMING) { asm volatile("" :: "v"(bfv[0]), "v"(bfv[1]), "v"(bfv[2]), "v"(bfv[3])); SW_TICK(1); }
            // ---- A fragments: per 16-row sub-block, eight ds_read_b128 fetch the
            //      four rotations x four k-steps (tile layout [s][kp][lane][q]).
            auto load_a = [&](f64x2 (&dst)[4][2], int sb) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int kp = 0; kp < 2; ++kp) dst[r][kp] = A2[(sb * 2 + kp) * 64 + rot[r]];
            };
            if (!DIAG) {
                // sub-blocks in pairs: 8 independent accumulators per (kk, r) sweep keep
                // the dependent-accumulate distance at 8 MFMAs; A fragments are
                // software-pipelined one pair ahead.
                f64x2 av[2][2][4][2];
                load_a(av[0][0], 0);
                load_a(av[0][1], 1);
#pragma unroll
                for (int pr = 0; pr < RS / 2; ++pr) {
                    auto mfma_pair = [&](int kk) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            acc[2 * pr][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(
                                av[pr & 1][0][r][kk >> 1][kk & 1], bfv[kk], acc[2 * pr][r], 0, 0, 0);
                            acc[2 * pr + 1][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(
                                av[pr & 1][1][r][kk >> 1][kk & 1], bfv[kk], acc[2 * pr + 1][r], 0, 0, 0);
                        }
                    };
                    // The LDS counter saturates at 15: issuing the next pair's 16 reads
                    // BEFORE this pair's first MFMA would force a wait on fresh reads.
                    // So: first k-step (its operands landed during the previous pair),
                    // then the prefetch, then the remaining three k-steps.
                    mfma_pair(0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (pr + 1 < RS / 2) {
                        load_a(av[(pr + 1) & 1][0], 2 * pr + 2);
                        load_a(av[(pr + 1) & 1][1], 2 * pr + 3);
                    }
                    mfma_pair(1);
                    mfma_pair(2);
                    mfma_pair(3);
                    // staging of the next tile, two instructions per pair:
                    // pairs 0-3 load half 0, 4-7 store it, 8-11 load half 1, 12-15 store it
                    if (pr < 4) piece_load(ntile, 0, 2 * pr);
                    else if (pr < 8) piece_store(buf ^ 1, 0, 2 * (pr - 4));
                    else if (pr < 12) piece_load(ntile, 1, 2 * (pr - 8));
                    else piece_store(buf ^ 1, 1, 2 * (pr - 12));
                    if (pr == RS / 2 - 1) x_store(buf ^ 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                // diagonal row block: W is lower triangular, so 16-row sub-blocks above
                // the chunk's k range are structurally zero, and rows >= N are padding.
                const int smin = kc - CPB * ib;
                const int smax = sb_lim - RS * ib;
                f64x2 av[2][4][2];
                load_a(av[0], 0);
#pragma unroll
                for (int sb = 0; sb < RS; ++sb) {
                    if (sb + 1 < RS) load_a(av[(sb + 1) & 1], sb + 1);
                    if (sb >= smin && sb < smax) {
#pragma unroll
                        for (int kk = 0; kk < NKK; ++kk)
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                acc[sb][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(
                                    av[sb & 1][r][kk >> 1][kk & 1], bfv[kk], acc[sb][r], 0, 0, 0);
                    }
                    if ((sb & 1) == 1) {
                        const int pr = sb >> 1;
                        if (pr < 4) piece_load(ntile, 0, 2 * pr);
                        else if (pr < 8) piece_store(buf ^ 1, 0, 2 * (pr - 4));
                        else if (pr < 12) piece_load(ntile, 1, 2 * (pr - 8));
                        else piece_store(buf ^ 1, 1, 2 * (pr - 12));
                        if (pr == RS / 2 - 1) x_store(buf ^ 1);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            SW_TICK(2);
            SW_TICK(3);
            __syncthreads();
            SW_TICK(4);
            buf ^= 1;
        };
        // first chunk that needs the predicated path: the diagonal block, or every
        // chunk when the row block has padded rows (N not a multiple of 512)
        const int ndiag0 = (sb_lim - RS * ib < RS) ? 0 : CPB * ib;
        for (; kc < nkc && kc < ndiag0; ++kc) do_tile(std::false_type{});
        for (; kc < nkc; ++kc) do_tile(std::true_type{});
        // row block finished: fold ||V||^2 into the per-candidate sum
#pragma unroll
        for (int s = 0; s < RS; ++s)
#pragma unroll
            for (int r = 0; r < 4; ++r) qpart = fma(acc[s][r], acc[s][r], qpart);
    }

    if (TIMING && a.dbg && blockIdx.x == 0 && t == 0) {
        for (int i = 0; i < 5; ++i) a.dbg[i] = tph[i];
        long long nt = 0;
        for (int ib = 0; ib < a.nrb; ++ib) nt += nkc_of(ib);
        a.dbg[5] = (unsigned long long)nt;
    }
    // ---- reduce over the four k-quarters / row-quarters of the wavefront ------
    qpart += __shfl_xor(qpart, 16);
    qpart += __shfl_xor(qpart, 32);
    mupart += __shfl_xor(mupart, 16);
    mupart += __shfl_xor(mupart, 32);
    double bu = INFINITY;
    long long bi = -1;
    if (kq == 0 && crow < a.m) {
        double mu = mupart + a.mean;
        double var = a.amp - qpart;
        if (has_nan) { mu = NAN; var = NAN; }     // george propagates NaN coordinates
        if (a.mu) a.mu[crow] = mu;
        if (a.var) a.var[crow] = var;
        if (a.kind != APGP_UTIL_NONE) {
            double uu = adm ? util_value(a.kind, mu, var, a.zeta, a.ybest) : INFINITY;
            if (a.u) a.u[crow] = uu;
            best_merge(bu, bi, uu, a.idx_offset + crow);
        }
    }
    if (a.kind == APGP_UTIL_NONE) return;
    // wavefront arg-min, then workgroup arg-min through LDS
    for (int o = 8; o > 0; o >>= 1) {
        double ou = __shfl_xor(bu, o);
        long long oi = __shfl_xor(bi, o);
        best_merge(bu, bi, ou, oi);
    }
    double* red_u = smem;                    // LDS is free again (all tiles consumed)
    long long* red_i = (long long*)(smem + 8);
    if (lane == 0) { red_u[w] = bu; red_i[w] = bi; }
    __syncthreads();
    if (t == 0) {
        for (int i = 1; i < SW_THREADS / 64; ++i) best_merge(bu, bi, red_u[i], red_i[i]);
        a.part_u[blockIdx.x] = bu;
        a.part_i[blockIdx.x] = bi;
    }
}

// K6: final arg-min over the per-workgroup partials (one workgroup).
__global__ __launch_bounds__(1024) void argmin_final_kernel(const double* part_u, const long long* part_i,
                                                            long long nparts, apgp_best_t* best) {
    __shared__ double su[16];
    __shared__ long long si[16];
    double bu = INFINITY;
    long long bi = -1;
    for (long long p = threadIdx.x; p < nparts; p += 1024) best_merge(bu, bi, part_u[p], part_i[p]);
    for (int o = 32; o > 0; o >>= 1) {
        double ou = __shfl_xor(bu, o);
        long long oi = __shfl_xor(bi, o);
        best_merge(bu, bi, ou, oi);
    }
    if ((threadIdx.x & 63) == 0) { su[threadIdx.x >> 6] = bu; si[threadIdx.x >> 6] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 16; ++i) best_merge(bu, bi, su[i], si[i]);
        best->u = bu;
        best->index = bi;
    }
}

template <int DPAD>
static int launch_sweep(const SweepArgs& a, hipStream_t s) {
    const size_t lds = (2 * SW_TILE + 2 * SW_KC * (DPAD + 2) + APGP_EXP_TAB_N) * sizeof(double);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)sweep_kernel<DPAD, false>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    const unsigned nblk = (unsigned)((a.m + SW_CAND - 1) / SW_CAND);
    static int timing = -1;
    if (timing < 0) { const char* e = getenv("APGP_SWEEP_TIMING"); timing = (e && e[0] == '1') ? 1 : 0; }
    if (timing && DPAD == 8) {
        // developer instrumentation: per-phase s_memtime cycles of block 0 / wave 0
        static unsigned long long* dbg = nullptr;
        if (!dbg) (void)hipMalloc(&dbg, 8 * sizeof(unsigned long long));
        SweepArgs b = a;
        b.dbg = dbg;
        (void)hipFuncSetAttribute((const void*)sweep_kernel<8, true>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((sweep_kernel<8, true>), dim3(nblk), dim3(SW_THREADS), lds, s, b);
        unsigned long long h[8];
        (void)hipMemcpyAsync(h, dbg, sizeof(h), hipMemcpyDeviceToHost, s);
        (void)hipStreamSynchronize(s);
        fprintf(stderr, "[apgp sweep timing] tiles %llu | per tile cycles: stage %.0f gen %.0f mfma %.0f commit %.0f wait+barrier %.0f\n",
                h[5], (double)h[0] / h[5], (double)h[1] / h[5], (double)h[2] / h[5], (double)h[3] / h[5], (double)h[4] / h[5]);
        return 0;
    }
    hipLaunchKernelGGL((sweep_kernel<DPAD, false>), dim3(nblk), dim3(SW_THREADS), lds, s, a);
    return 0;
}

extern "C" int apgp_acquire(const double* T, int64_t m, int64_t idx_offset, const double* packed_linv,
                            const double* xs, int64_t n, const apgp_kernel_t* kern, double mean,
                            int32_t kind, const double* lo, const double* hi, const uint8_t* mask,
                            double zeta, double ybest, double* mu, double* var, double* u, void* part,
                            apgp_best_t* best, void* stream) {
    APGP_CHECK_ARG(T && packed_linv && xs && kern, "null pointer");
    APGP_CHECK_ARG(m >= 1 && n >= 1, "m >= 1 and n >= 1 required");
    APGP_CHECK_ARG(kind >= APGP_UTIL_AGP && kind <= APGP_UTIL_NONE, "unknown utility kind");
    APGP_CHECK_ARG(kind == APGP_UTIL_NONE || (part && best), "part/best required for an acquisition");
    APGP_CHECK_ARG((lo == NULL) == (hi == NULL), "lo and hi must be given together");
    KernConst kc;
    APGP_CHECK_ARG(apgp_make_kernconst(kern, &kc) == 0, "kernel parameters");
    SweepArgs a;
    a.T = T; a.linv = packed_linv; a.xs = xs; a.mask = mask; a.dbg = NULL;
    a.mu = mu; a.var = var; a.u = u;
    const long long nblk = (m + SW_CAND - 1) / SW_CAND;
    a.part_u = (double*)part;
    a.part_i = part ? (long long*)((double*)part + nblk) : NULL;
    a.m = m; a.idx_offset = idx_offset;
    a.ndim = kc.ndim; a.nrb = (int)(apgp_npad(n) / APGP_ROW_BLOCK); a.kind = kind; a.n = (int)n;
    a.has_box = lo != NULL;
    a.mean = mean; a.amp = kc.amp; a.zeta = zeta; a.ybest = ybest;
    for (int d = 0; d < APGP_MAX_DIM; ++d) {
        a.sc[d] = kc.sc[d];
        a.lo[d] = (lo && d < kc.ndim) ? lo[d] : 0.0;
        a.hi[d] = (hi && d < kc.ndim) ? hi[d] : 0.0;
    }
    hipStream_t s = (hipStream_t)stream;
    switch (kc.dpad) {
        case 2: launch_sweep<2>(a, s); break;
        case 4: launch_sweep<4>(a, s); break;
        case 8: launch_sweep<8>(a, s); break;
        default: launch_sweep<16>(a, s); break;
    }
    if (kind != APGP_UTIL_NONE)
        hipLaunchKernelGGL(argmin_final_kernel, dim3(1), dim3(1024), 0, s, a.part_u, a.part_i, nblk, best);
    APGP_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------
// Solve-based variant of the sweep for ill-conditioned factors.
// The explicit-inverse contraction above is as accurate as a triangular solve
// while cond(K) <~ 1e10; beyond that (the reference's own fitAmp=True optimum
// reaches cond ~ 8e15) only the solve-based form keeps the
// error at the level of george's cho_solve.  Here one wavefront owns one
// candidate: k* lives in LDS, v = L^-1 k* by forward substitution against the
// row-major factor (row i read coalesced, one wave reduction per row), then
// sigma^2 = amp - |v|^2.  O(N^2/64) cycles per candidate and the factor is
// re-read from L2 for every candidate, so this is a correctness path for the
// small, badly conditioned training sets that need it, not a throughput path.
// ---------------------------------------------------------------------------
struct SolveArgs {
    const double* T;
    const double* L;
    const double* xs;
    const unsigned char* mask;
    double* mu;
    double* var;
    double* u;
    double* part_u;
    long long* part_i;
    long long m, idx_offset, ldl;
    int ndim, n, kind, has_box;
    double mean, amp, zeta, ybest;
    double sc[APGP_MAX_DIM], lo[APGP_MAX_DIM], hi[APGP_MAX_DIM];
};

template <int DPAD>
__global__ __launch_bounds__(256) void sweep_solve_kernel(SolveArgs a) {
    constexpr int XS = DPAD + 2;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ double etab[APGP_EXP_TAB_N];
    __shared__ double red_u[4];
    __shared__ long long red_i[4];
    apgp_exp_tab_load(etab);
    __syncthreads();
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    double* v = smem + (size_t)w * a.n;
    const long long crow = (long long)blockIdx.x * 4 + w;
    const bool inb = crow < a.m;
    bool adm = inb, has_nan = false;
    double tt[DPAD];
#pragma unroll
    for (int d = 0; d < DPAD; ++d) {
        double x = 0.0;
        if (inb && d < a.ndim) {
            x = a.T[crow * a.ndim + d];
            if (a.has_box && !(x >= a.lo[d] && x <= a.hi[d])) adm = false;
            if (x != x) has_nan = true;
        }
        tt[d] = x * a.sc[d];
    }
    if (inb && a.mask && a.mask[crow] == 0) adm = false;
    double bu = INFINITY;
    long long bi = -1;
    if (inb) {
        double mup = 0.0;
        for (int k = lane; k < a.n; k += 64) {
            const double* xr = a.xs + (long long)k * XS;
            double s2 = 0.0, s3 = 0.0;
#pragma unroll
            for (int d = 0; d < DPAD; d += 2) {
                const double df0 = tt[d] - xr[d], df1 = tt[d + 1] - xr[d + 1];
                s2 = fma(df0, df0, s2);
                s3 = fma(df1, df1, s3);
            }
            const double kv = a.amp * apgp_exp(-(s2 + s3), etab);
            v[k] = kv;
            mup = fma(kv, xr[DPAD], mup);
        }
        for (int o = 32; o > 0; o >>= 1) mup += __shfl_xor(mup, o);
        double q = 0.0;
        for (int i = 0; i < a.n; ++i) {
            const double* lrow = a.L + (long long)i * a.ldl;
            double p = 0.0;
            for (int j = lane; j < i; j += 64) p = fma(lrow[j], v[j], p);
            for (int o = 32; o > 0; o >>= 1) p += __shfl_xor(p, o);
            const double vi = (v[i] - p) / lrow[i];
            if (lane == 0) v[i] = vi;       // same wavefront: LDS ops are ordered
            q = fma(vi, vi, q);
        }
        double mu = mup + a.mean;
        double var = a.amp - q;
        if (has_nan) { mu = NAN; var = NAN; }
        if (lane == 0) {
            if (a.mu) a.mu[crow] = mu;
            if (a.var) a.var[crow] = var;
            if (a.kind != APGP_UTIL_NONE) {
                const double uu = adm ? util_value(a.kind, mu, var, a.zeta, a.ybest) : INFINITY;
                if (a.u) a.u[crow] = uu;
                best_merge(bu, bi, uu, a.idx_offset + crow);
            }
        }
    }
    if (a.kind == APGP_UTIL_NONE) return;
    if (lane == 0) { red_u[w] = bu; red_i[w] = bi; }
    __syncthreads();
    if (t == 0) {
        for (int i = 1; i < 4; ++i) best_merge(bu, bi, red_u[i], red_i[i]);
        a.part_u[blockIdx.x] = bu;
        a.part_i[blockIdx.x] = bi;
    }
}

extern "C" int apgp_acquire_solve(const double* T, int64_t m, int64_t idx_offset, const double* L,
                                  int64_t ldl, const double* xs, int64_t n, const apgp_kernel_t* kern,
                                  double mean, int32_t kind, const double* lo, const double* hi,
                                  const uint8_t* mask, double zeta, double ybest, double* mu, double* var,
                                  double* u, void* part, apgp_best_t* best, void* stream) {
    APGP_CHECK_ARG(T && L && xs && kern, "null pointer");
    APGP_CHECK_ARG(m >= 1 && n >= 1 && ldl >= n, "m >= 1, n >= 1 and ldl >= n required");
    APGP_CHECK_ARG(n <= 4096, "solve-based sweep keeps k* in LDS: n <= 4096");
    APGP_CHECK_ARG(kind >= APGP_UTIL_AGP && kind <= APGP_UTIL_NONE, "unknown utility kind");
    APGP_CHECK_ARG(kind == APGP_UTIL_NONE || (part && best), "part/best required for an acquisition");
    APGP_CHECK_ARG((lo == NULL) == (hi == NULL), "lo and hi must be given together");
    KernConst kc;
    APGP_CHECK_ARG(apgp_make_kernconst(kern, &kc) == 0, "kernel parameters");
    SolveArgs a;
    a.T = T; a.L = L; a.xs = xs; a.mask = mask; a.mu = mu; a.var = var; a.u = u;
    const long long nblk = (m + 3) / 4;
    a.part_u = (double*)part;
    a.part_i = part ? (long long*)((double*)part + nblk) : NULL;
    a.m = m; a.idx_offset = idx_offset; a.ldl = ldl;
    a.ndim = kc.ndim; a.n = (int)n; a.kind = kind; a.has_box = lo != NULL;
    a.mean = mean; a.amp = kc.amp; a.zeta = zeta; a.ybest = ybest;
    for (int d = 0; d < APGP_MAX_DIM; ++d) {
        a.sc[d] = kc.sc[d];
        a.lo[d] = (lo && d < kc.ndim) ? lo[d] : 0.0;
        a.hi[d] = (hi && d < kc.ndim) ? hi[d] : 0.0;
    }
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = (size_t)4 * n * sizeof(double);
    dim3 grid((unsigned)nblk), block(256);
#define APGP_LAUNCH_SOLVE(DP)                                                                           \
    do {                                                                                                \
        (void)hipFuncSetAttribute((const void*)sweep_solve_kernel<DP>,                                  \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);              \
        hipLaunchKernelGGL(sweep_solve_kernel<DP>, grid, block, lds, s, a);                             \
    } while (0)
    switch (kc.dpad) {
        case 2: APGP_LAUNCH_SOLVE(2); break;
        case 4: APGP_LAUNCH_SOLVE(4); break;
        case 8: APGP_LAUNCH_SOLVE(8); break;
        default: APGP_LAUNCH_SOLVE(16); break;
    }
#undef APGP_LAUNCH_SOLVE
    if (kind != APGP_UTIL_NONE)
        hipLaunchKernelGGL(argmin_final_kernel, dim3(1), dim3(1024), 0, s, a.part_u, a.part_i, nblk, best);
    APGP_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------
// mean-only prediction: the batched ApproxPosterior._gpll (approx.py:178-180).
// One wavefront per candidate; lanes stride over the training points.
// ---------------------------------------------------------------------------
struct MeanArgs {
    const double* T;
    const double* xs;
    double* mu;
    long long m, npad;
    int ndim;
    double mean, amp;
    double sc[APGP_MAX_DIM];
};

template <int DPAD>
__global__ __launch_bounds__(256) void predict_mean_kernel(MeanArgs a) {
    constexpr int XS = DPAD + 2;
    __shared__ double etab[APGP_EXP_TAB_N];
    apgp_exp_tab_load(etab);
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long long row = (long long)blockIdx.x * 4 + w;
    if (row >= a.m) return;
    double tt[DPAD];
#pragma unroll
    for (int d = 0; d < DPAD; ++d) tt[d] = d < a.ndim ? a.T[row * a.ndim + d] * a.sc[d] : 0.0;
    double acc = 0.0;
    for (long long k = lane; k < a.npad; k += 64) {
        const double* xr = a.xs + k * XS;
        double s = 0.0, s3 = 0.0;
#pragma unroll
        for (int d = 0; d < DPAD; d += 2) {
            double df0 = tt[d] - xr[d];
            double df1 = tt[d + 1] - xr[d + 1];
            s = fma(df0, df0, s);
            s3 = fma(df1, df1, s3);
        }
        acc = fma(a.amp * apgp_exp(-(s + s3), etab), xr[DPAD], acc);
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    bool bad = false;
    for (int d = 0; d < DPAD; ++d) bad = bad || (tt[d] != tt[d]);
    if (lane == 0) a.mu[row] = bad ? NAN : acc + a.mean;
}

extern "C" int apgp_predict_mean(const double* T, int64_t m, const double* xs, int64_t n,
                                 const apgp_kernel_t* kern, double mean, double* mu, void* stream) {
    APGP_CHECK_ARG(T && xs && kern && mu, "null pointer");
    APGP_CHECK_ARG(m >= 1 && n >= 1, "m >= 1 and n >= 1 required");
    KernConst kc;
    APGP_CHECK_ARG(apgp_make_kernconst(kern, &kc) == 0, "kernel parameters");
    MeanArgs a;
    a.T = T; a.xs = xs; a.mu = mu; a.m = m; a.npad = apgp_npad(n); a.ndim = kc.ndim;
    a.mean = mean; a.amp = kc.amp;
    for (int d = 0; d < APGP_MAX_DIM; ++d) a.sc[d] = kc.sc[d];
    dim3 grid((unsigned)((m + 3) / 4)), block(256);
    hipStream_t s = (hipStream_t)stream;
    switch (kc.dpad) {
        case 2: hipLaunchKernelGGL(predict_mean_kernel<2>, grid, block, 0, s, a); break;
        case 4: hipLaunchKernelGGL(predict_mean_kernel<4>, grid, block, 0, s, a); break;
        case 8: hipLaunchKernelGGL(predict_mean_kernel<8>, grid, block, 0, s, a); break;
        default: hipLaunchKernelGGL(predict_mean_kernel<16>, grid, block, 0, s, a); break;
    }
    APGP_CHECK_LAUNCH();
    return 0;
}
