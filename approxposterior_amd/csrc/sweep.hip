// K5/K6 "sweep": fused candidate tile -> k* row -> mu -> L^-1 k* contraction on
// the f64 matrix cores -> predictive variance -> AGP / BAPE / Jones utility ->
// arg-min.  The batched counterpart of
//   george.GP.predict(y, t, return_var=True)        (utility.py:131,178,224)
//   utility.AGPUtility / BAPEUtility / JonesUtility (utility.py:99-250)
//   utility.minimizeObjective's arg-min             (utility.py:369-371)
//
// One kernel implements it: sweep2_kernel, the two-role kernel further down (matrix wavefronts
// + feeder wavefronts); a short last round of the persistent grid is split by row block and
// finished by sweep_finish_kernel.
//
// Data flow per workgroup (512 threads = 4 matrix + 4 feeder wavefronts, 64 candidates):
//   * matrix / feeder wavefront w owns candidates [16 w, 16 w + 16) (one MFMA column block).
//   * the packed factor W = L^-1 is streamed tile by tile (the 256-row halves of the packed
//     512 rows x 16 k tiles, 32 KiB, A-fragment order) L2/MALL -> LDS by LDS-DMA (ring of
//     three slots), shared by the four matrix wavefronts.
//   * the feeder generates the k*(t_m, x_k) values the MFMA B operand needs (lane ->
//     candidate lane&15, k lane>>4) on the VALU (D sub + D fma + table-driven exp) the
//     FIRST time chunk k is visited (the diagonal row block); later row blocks need the same
//     operands again, and because the fp64 VALU shares the DP pipe with the matrix cores,
//     regenerating them costs ~12 % of a tile.  They are parked in a per-workgroup-slot
//     scratch stream instead (32 B per lane per chunk, written once, read back by LDS-DMA)
//     -- K* is still never materialised as a matrix the host sees.  The grid is persistent
//     (one workgroup per CU, candidate blocks dealt round-robin) so the scratch is SW_GRID
//     slots, not M/64.
//   * V = W K*^T is accumulated 256 rows x 16 candidates per matrix wavefront in 64
//     v_mfma_f64_4x4x4_4b_f64 accumulators; at the end of a row block the squares are folded
//     into a per-candidate sum; V is never stored.
//   * mu is a VALU by-product of the generating tiles (every k exactly once).
#include "apgp_common.h"
#include "mma16.h"
#include "scratch.h"
#include <chrono>
#include <mutex>
#include <stdlib.h>
#include <type_traits>
#include <utility>
#include <string.h>

#define SW_ROWS APGP_ROW_BLOCK       // 512 rows of W per tile
#define SW_KC APGP_K_CHUNK           // 16 k per tile
#define SW_TILE (SW_ROWS * SW_KC)    // doubles per tile (64 KiB)
#define SW_CAND 64                   // candidates per workgroup (16 per wavefront)
#define SW_GRID 256                  // persistent workgroups = K* scratch slots (one per CU)
#define SW_BCH 1024                  // doubles of parked B operands per chunk per slot (4 wavefronts x 64 lanes x 4 k)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// cache policy of the parked-operand stream (aux bits of its buffer loads / stores): 0 = default.
// Each workgroup slot is written once and read back once per later row block and 256 slots x
// 2 MB never fit the L2, so a non-temporal hint (aux 2) looked right -- measured alternating on
// one box with the two-role kernel it is 0.5 % slower than the default policy (256.5 vs 257.9 ms).
#ifndef SW_KAUX
#define SW_KAUX 0
#endif

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
    double* kcache;            // parked B operands: SW_GRID slots x ncache chunks x SW_BCH
    unsigned linv_bytes, xs_bytes, kslot_bytes;   // buffer-descriptor extents
    // candidate blocks [blk_begin, blk_end) of this launch; split != 0: one workgroup per
    // (candidate block, row block), partial sums to sp_q / sp_mu (the short last round)
    long long blk_begin, blk_end;
    double* sp_q;
    double* sp_mu;
    int split;
    long long m, idx_offset;
    int ndim, nrb, kind, has_box, n, ncache, lin_order;
    double mean, amp, zeta, ybest, lin_coef;
    double sc[APGP_MAX_DIM], lo[APGP_MAX_DIM], hi[APGP_MAX_DIM], lw[APGP_MAX_DIM];
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
// independent (CBSZ/ABID broadcast is ignored for f64 on gfx950):
//   A lane = i + 4 b + 16 k,  B lane = j + 4 b + 16 k,  D lane = j + 4 b + 16 i.
// A 16 x 16 x 4 product is four instructions with the SAME A fragment and the B
// operand rotated by 4 r lanes inside each row of 16 lanes (DPP row_ror): block b
// then multiplies row group b with candidate group (b + r) mod 4.  The candidate
// of an accumulator depends on r, which the variance reduction (sum over rows of
// V^2 per candidate) undoes once per row block with the inverse rotation.
// (Rotating the A fragment instead keeps lane <-> candidate fixed but needs four
// times the LDS reads and A registers: it ran the LDS at 73 % of its bandwidth.)

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

// ===========================================================================
// Two-role sweep (round 1e): 8 wavefronts per workgroup, two per SIMD.
//   wavefronts 0-3 ("matrix" role, one per SIMD): nothing but the MFMA stream, the LDS reads
//     of its operands and one barrier per tile.  256 rows x 16 candidates per wavefront
//     (64 accumulators): the 256 unified registers of a two-wave-per-SIMD kernel hold them.
//   wavefronts 4-7 ("feeder" role): everything else -- the packed-factor stream L2 -> registers
//     -> LDS, the B operands (generated on the first visit of a chunk, otherwise fetched from
//     the parked stream) written to LDS for the matrix wavefront of the same candidates, mu,
//     the candidate setup and the utility / arg-min epilogue.
// tools/mfma_pair.hip: the SIMD issues the older wavefront first, so a companion's LDS / VMEM /
// integer traffic costs the MFMA wavefront nothing (16.25 -> 16.4 cycles per MFMA), while in
// the one-wave design every such instruction's issue time is lost to the matrix pipe (~15 %).
// Only the feeder's fp64 VALU work (k* generation) still competes for the shared DP pipe.
// Tiles are the 256-row HALVES of the packed 512 x 16 tiles (same packed format): row block jb
// = half jb & 1 of packed row block jb >> 1, chunks 0 .. 16 (jb + 1) - 1.
// Synchronisation: tile i+1 (A image into slot (i+1) % 3, B operands into buffer (i+1) & 1) is
// written by the feeders between barrier i-1 and barrier i; the matrix wavefronts execute
// barrier i after the first k-step of pair 4 of tile i and read tile i+1 only after it.
// ===========================================================================
#define S2_THREADS 512
// LDS reads of the matrix role are issued one per S2_SPREAD MFMAs instead of in clusters of four
// (sched_group_barrier): clustered, the reads' issue time exceeds one MFMA's 16-cycle shadow and
// the matrix pipe idles (same box, alternating: 249.5 vs 255.0 ms at C3).  0 = compiler's order.
#ifndef S2_SPREAD
#define S2_SPREAD 1
#endif
#define S2_ROWS 256
#define S2_TILE (S2_ROWS * SW_KC)        // doubles per tile image (32 KiB)
#define S2_CPB (S2_ROWS / SW_KC)         // chunks per row-block width (16)
#define S2_NP (S2_ROWS / 32)             // sub-block pairs per tile (8)
// substitution form: the statically unrolled diagonal tiles are used up to this many 256-row blocks
// (N <= 2048), the run-time-indexed body above that (see the matrix role)
#define S2_STATIC_DIAG_NRB 8

// compile-time loop (the body sees its index as a constant expression)
template <int... Is, class F>
__device__ __forceinline__ void s2_static_for_impl(std::integer_sequence<int, Is...>, F&& f) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void s2_static_for(F&& f) {
    s2_static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

// which form of the feeder loop an instantiation gets (see the feeder role)
// (Dpad = 32, round 5: the plain feeder -- an x chunk is 4352 bytes there, more than the four feeder wavefronts' four 1 KiB
// LDS-DMA pieces of the structured form, whose hand-counted vmcnt waits assume exactly one piece per wavefront)
static constexpr bool s2_structured_feeder(int dpad, bool solve) { return dpad == 32 ? false : (solve ? dpad == 2 : dpad != 8); }

// MODE 0: inverse form (A = packed L^-1); 1: substitution form; 2: substitution form with the statically
// unrolled diagonal tiles (N <= 256 S2_STATIC_DIAG_NRB) -- a separate instantiation: the 65 KiB of
// unrolled code slowed the run-time-indexed path of the SAME kernel by 20 % at N = 4096 (333 vs 276 ms;
// identical hot loops, only their placement differs), so large N run the kernel that does not contain it.
template <int DPAD, bool LIN, int MODE>
__global__ __launch_bounds__(S2_THREADS, 1) void sweep2_kernel(SweepArgs a) {
    constexpr bool SOLVE = MODE != 0;
    constexpr bool STATIC_DIAG = MODE == 2;
    constexpr int XS = DPAD + 2;
    constexpr int NKK = SW_KC / 4;
    constexpr int NP = S2_NP;
    constexpr int XCHUNK16 = SW_KC * XS / 2;
    // doubles between two x buffers: the chunk itself, or (structured feeder: whole-wavefront LDS-DMA
    // pieces of 1 KiB, ring of three buffers) the chunk rounded up to 1 KiB
    constexpr bool SF = s2_structured_feeder(DPAD, SOLVE);
    constexpr int XSTRIDE = SF ? (SW_KC * XS * 8 + 1023) / 1024 * 128 : SW_KC * XS;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* Aring = smem;                         // 3 tile images
    double* Bbuf = Aring + 3 * S2_TILE;           // 2 x 4 wavefronts x [2][64] x 16 B
    double* Xbuf = Bbuf + 2 * 4 * 256;            // 2 x SW_KC x XS
    double* Etab = Xbuf + (SF ? 3 : 2) * XSTRIDE;
    double* Shq = Etab + APGP_EXP_TAB_N;          // sum V^2 per candidate (matrix -> feeder)
    double* red_u = Shq + SW_CAND;
    long long* red_i = (long long*)(red_u + 4);
    double* Cst = red_u + 8;                      // sc | lo | hi | lw (4 x APGP_MAX_DIM), feeder only
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int cl = lane & 15, kq = lane >> 4;
    apgp_exp_tab_load(Etab);
    if (t == 256) {
#pragma unroll
        for (int d = 0; d < APGP_MAX_DIM; ++d) {
            Cst[d] = a.sc[d];
            Cst[APGP_MAX_DIM + d] = a.lo[d];
            Cst[2 * APGP_MAX_DIM + d] = a.hi[d];
            Cst[3 * APGP_MAX_DIM + d] = a.lw[d];
        }
    }

    const int kc_lim = (a.n + SW_KC - 1) / SW_KC;
    const int nrb2 = (a.n + S2_ROWS - 1) / S2_ROWS;
    auto nkc_of = [&](int jb) { const int v = S2_CPB * (jb + 1); return v < kc_lim ? v : kc_lim; };
    // byte offset of tile (jb, kc) in the packed factor: half jb & 1 of packed tile (jb >> 1, kc)
    auto tile_off = [&](int jb, int kc) {
        const int ib = jb >> 1;
        const unsigned tile = (unsigned)((SW_ROWS / SW_KC) * ib * (ib + 1) / 2 + kc);
        return tile * (unsigned)(SW_TILE * 8) + (unsigned)((jb & 1) * S2_TILE * 8);
    };
    // (jb, kc, block) -> successor in the stream: next chunk, next row block, next candidate block
    // work of this workgroup: row blocks jb_lo .. jb_hi-1 of candidate blocks blk0, blk0 +
    // blk_step, ...  Persistent mode: every row block of every gridDim-th candidate block.
    // Split mode (short last round / small launches): ONE (candidate block, row block) item,
    // heaviest row blocks first; the shares go to sp_q / sp_mu and sweep_finish_kernel.
    int jb_lo = 0, jb_hi = nrb2;
    long long blk0 = a.blk_begin + blockIdx.x, blk_step = gridDim.x;
    if (a.split) {
        const int nsplit = (int)(a.blk_end - a.blk_begin);
        blk0 = a.blk_begin + (int)blockIdx.x % nsplit;
        jb_lo = nrb2 - 1 - (int)blockIdx.x / nsplit;
        jb_hi = jb_lo + 1;
        blk_step = a.blk_end;
    }
    auto successor = [&](int& jb, int& kc, long long& blk) {
        ++kc;
        if (kc >= nkc_of(jb)) { ++jb; kc = 0; }
        if (jb >= jb_hi) { jb = jb_lo; blk += blk_step; }
    };

    if (w < 4) {
        // =============================== matrix role ===============================
        // lane whose B value this lane multiplies in rotation r (block b meets candidate
        // group b + r): the rotation is an LDS read address, not a register move
        int lr[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) lr[r] = (lane & 48) | ((lane + 4 * r) & 15);
        f64x2 av[2][2][2];
        auto load_a = [&](f64x2 (&dst)[2][2], int slot, int p) {
            const f64x2* A2 = (const f64x2*)(Aring + slot * S2_TILE) + lane;
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int kp = 0; kp < 2; ++kp) dst[h][kp] = A2[((p * 2 + h) * 2 + kp) * 64];
        };
        __syncthreads();                          // C : constants staged
        __syncthreads();                          // P0: x chunks staged (feeders)
        __syncthreads();                          // P : tile 0 published
        int slot = 0, bpar = 0;
        // B operands of the current tile in all four rotations; half 0 = k-steps 0-1, 1 = 2-3
        double brot[4][NKK];
        auto load_b = [&](int buf, int half) {
            const f64x2* Bw = (const f64x2*)(Bbuf + buf * 1024 + w * 256) + half * 64;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const f64x2 bv = Bw[lr[r]];
                brot[r][2 * half] = bv.x; brot[r][2 * half + 1] = bv.y;
            }
        };
        load_a(av[0], 0, 0);
        load_b(0, 0);
        // SOLVE: the solved blocks V are parked by the matrix wavefronts themselves (same slot
        // layout the feeders of the inverse form write: [half][wavefront][lane] x 16 B)
        const __amdgpu_buffer_rsrc_t rs_kv = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(a.kcache + (long long)blockIdx.x * a.ncache * SW_BCH), 0, (int)a.kslot_bytes, 0x00020000);
        for (long long blk = blk0; blk < a.blk_end; blk += blk_step) {
            double qtot = 0.0;
            for (int jb = jb_lo; jb < jb_hi; ++jb) {
                const int nkc = nkc_of(jb);
                const int ndiag0 = S2_CPB * jb;
                double qmic = 0.0;                // SOLVE: sum V^2 of the 16-row blocks solved on the diagonal
                double acc[2 * NP][4];
#pragma unroll
                for (int s_ = 0; s_ < 2 * NP; ++s_)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[s_][r] = 0.0;
                int kc = 0;
                auto do_tile_n = [&](auto pred_tag, auto npa_tag) {
                    constexpr bool PRED = decltype(pred_tag)::value;
                    // NPA: sub-block pairs this body works through (8 = the whole tile; 4 / 6: the straight tiles
                    // of a partial last row block, whose remaining rows are zero in the packed factor; even,
                    // so that the last pair's prefetch of the next tile lands in the free fragment buffer)
                    constexpr int NPA = decltype(npa_tag)::value;
                    const int nslot = slot == 2 ? 0 : slot + 1;
                    auto mfma_pair = [&](int pr, int kk) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            acc[2 * pr][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(
                                av[pr & 1][0][kk >> 1][kk & 1], brot[r][kk], acc[2 * pr][r], 0, 0, 0);
                            acc[2 * pr + 1][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(
                                av[pr & 1][1][kk >> 1][kk & 1], brot[r][kk], acc[2 * pr + 1][r], 0, 0, 0);
                        }
                    };
                    int p0 = (kc - ndiag0) >> 1;
                    if (p0 < 0) p0 = 0;
                    int p1 = (a.n - S2_ROWS * jb + 31) >> 5;
                    if (p1 > NP) p1 = NP;
#pragma unroll
                    for (int pr = 0; pr < NPA; ++pr) {
                        const bool act = !PRED || (pr >= p0 && pr < p1);
                        if (act) mfma_pair(pr, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        if (pr == NPA / 2) {
                            // (SOLVE: the park stores of the last diagonal tile are acknowledged before this
                            // wavefront arrives -- their readers request them >= 14 barriers later)
                            if constexpr (SOLVE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                            __syncthreads();                    // barrier i: tile i+1 is complete
                        }
                        if (pr + 1 < NPA) {
                            // (first pair: the k-step 2-3 half of this tile's B operands; its
                            // registers were still in use when the 0-1 half was prefetched)
                            if (pr == 0) load_b(bpar, 1);
                            load_a(av[(pr + 1) & 1], slot, pr + 1);
                            if (act) {
                                mfma_pair(pr, 1);
                                mfma_pair(pr, 2);
                                mfma_pair(pr, 3);
                            }
#if S2_SPREAD > 0
                            if (!PRED) {
#pragma unroll
                                for (int q_ = 0; q_ < (pr == 0 ? 8 : 4); ++q_) {
                                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                                    __builtin_amdgcn_sched_group_barrier(0x008, S2_SPREAD, 0);
                                }
                            }
#endif
                        } else {
                            // last pair: the next tile's first A fragments and, as soon as this
                            // tile's k-step 0-1 B registers fall free, that half of the next
                            // tile's B operands (published by this tile's barrier): their LDS
                            // latency hides behind the 16 MFMAs of k-steps 2-3
                            load_a(av[0], nslot, 0);
                            if (act) mfma_pair(pr, 1);
                            __builtin_amdgcn_sched_barrier(0);
                            load_b(bpar ^ 1, 0);
                            if (act) {
                                mfma_pair(pr, 2);
                                mfma_pair(pr, 3);
                            }
#if S2_SPREAD > 0
                            if (!PRED) {
#pragma unroll
                                for (int q_ = 0; q_ < 4; ++q_) {
                                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                                    __builtin_amdgcn_sched_group_barrier(0x008, S2_SPREAD, 0);
                                }
                            }
#endif
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    slot = nslot;
                    bpar ^= 1;
                };
                auto do_tile = [&](auto pred_tag) { do_tile_n(pred_tag, std::integral_constant<int, NP>{}); };
                const int nstraight = (a.n - S2_ROWS * jb >= S2_ROWS) ? ndiag0 : 0;
                for (; kc < nstraight; ++kc) do_tile(std::false_type{});
                if constexpr (!SOLVE) {
                    // partial last row block of <= 128 / <= 192 rows: its straight tiles with four / six pairs at
                    // compile time (the predicated body costs 3.4 k cycles for four pairs' 2.05 k of MFMAs,
                    // profiles/r03aa; N = 1152: 23.9 -> 22.8 ms, bit-identical).  Plain loops in sequence: as
                    // if / else alternatives that join with the 128 accumulators live, hipcc spilled them.
                    const int npart4 = (nstraight == 0 && a.n - S2_ROWS * jb <= 128) ? ndiag0 : 0;
                    for (; kc < npart4; ++kc) do_tile_n(std::false_type{}, std::integral_constant<int, 4>{});
                    const int npart6 = (nstraight == 0 && a.n - S2_ROWS * jb <= 192) ? ndiag0 : 0;
                    for (; kc < npart6; ++kc) do_tile_n(std::false_type{}, std::integral_constant<int, 6>{});
                    for (; kc < nkc; ++kc) do_tile(std::true_type{});
                } else {
                    for (; kc < ndiag0 && kc < nkc; ++kc) do_tile(std::true_type{});   // (partial last row block)
                    // ---- diagonal tiles of the substitution form ---------------------------------
                    // Tile (jb, 16 jb + c): the image holds -L (rows below the 16 x 16 diagonal block c) and,
                    // in sub-block slot c, that block prepared for a 4 x 4-blocked solve (pack_lsolve_kernel):
                    //   Ts[(4 q + q') 16 + 4 i + k] = -L_c[4q+i][4q'+k] (q' < q), inverse of the 4 x 4 diagonal
                    //   block (q' = q).  With acc[c] = -sum_{j<c} L_cj V_j from the tiles so far,
                    //   V_c = L_cc^-1 (K*_c + acc[c]) is solved in registers, four rows at a time, on the
                    //   four-block MFMA with the blocks used as CANDIDATE groups (A replicated, B/D lane
                    //   = cand + 16 row): the D fragment of one step is the B fragment of the next, and
                    //   V_c in that layout is, lane for lane, the tile's B operand (rotation 0) -- the other
                    //   three rotations are DPP moves, the parked copy two 16-byte stores.  No LDS round trip.
                    auto do_diag = [&]() {
                        const int nslot = slot == 2 ? 0 : slot + 1;
                        const int c = kc - ndiag0;
                        int s1 = (a.n - S2_ROWS * jb + 15) >> 4;
                        if (s1 > 2 * NP) s1 = 2 * NP;
                        load_b(bpar, 1);                        // K*_c, k-steps 2-3 (0-1: prefetched)
                        double ta[10];
                        {
                            const double* Ts = Aring + slot * S2_TILE + c * 256 + (lane & 3) * 4 + (lane >> 4);
                            int e = 0;
#pragma unroll
                            for (int q = 0; q < 4; ++q)
#pragma unroll
                                for (int q2 = 0; q2 <= q; ++q2) ta[e++] = Ts[(q * 4 + q2) * 16];
                        }
                        const bool park_now = a.ncache > 0 && jb + 1 < nrb2;
                        auto half = [&](int pr, int h, int kk) {
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                acc[2 * pr + h][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(
                                    av[pr & 1][h][kk >> 1][kk & 1], brot[r][kk], acc[2 * pr + h][r], 0, 0, 0);
                        };
                        auto micro = [&](int S) {     // (S is a constant after the pair loop is unrolled)
                            // rows 4 q + i of sub-block S sit in lane block q of the four rotation registers
                            // (candidate group (q + r) & 3): ror 4 r brings group b' to lane block b'
                            double rot[4];
                            rot[0] = acc[S][0];
                            rot[1] = apgp_row_ror4<1>(acc[S][1]);
                            rot[2] = apgp_row_ror4<2>(acc[S][2]);
                            rot[3] = apgp_row_ror4<3>(acc[S][3]);
                            const int bb = (lane >> 2) & 3;
                            double R[4], V[4];
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const int rr = (bb - q) & 3;
                                R[q] = (rr == 0 ? rot[0] : rr == 1 ? rot[1] : rr == 2 ? rot[2] : rot[3]) + brot[0][q];
                            }
                            V[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(ta[0], R[0], 0.0, 0, 0, 0);
                            R[1] = __builtin_amdgcn_mfma_f64_4x4x4f64(ta[1], V[0], R[1], 0, 0, 0);
                            R[2] = __builtin_amdgcn_mfma_f64_4x4x4f64(ta[3], V[0], R[2], 0, 0, 0);
                            R[3] = __builtin_amdgcn_mfma_f64_4x4x4f64(ta[6], V[0], R[3], 0, 0, 0);
                            V[1] = __builtin_amdgcn_mfma_f64_4x4x4f64(ta[2], R[1], 0.0, 0, 0, 0);
                            R[2] = __builtin_amdgcn_mfma_f64_4x4x4f64(ta[4], V[1], R[2], 0, 0, 0);
                            R[3] = __builtin_amdgcn_mfma_f64_4x4x4f64(ta[7], V[1], R[3], 0, 0, 0);
                            V[2] = __builtin_amdgcn_mfma_f64_4x4x4f64(ta[5], R[2], 0.0, 0, 0, 0);
                            R[3] = __builtin_amdgcn_mfma_f64_4x4x4f64(ta[8], V[2], R[3], 0, 0, 0);
                            V[3] = __builtin_amdgcn_mfma_f64_4x4x4f64(ta[9], R[3], 0.0, 0, 0, 0);
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                brot[0][q] = V[q];
                                brot[1][q] = apgp_row_ror4<3>(V[q]);     // lane l <- lane l + 4
                                brot[2][q] = apgp_row_ror4<2>(V[q]);
                                brot[3][q] = apgp_row_ror4<1>(V[q]);
                                qmic = fma(V[q], V[q], qmic);
                            }
#pragma unroll
                            for (int r = 0; r < 4; ++r) acc[S][r] = 0.0;    // (counted through qmic)
                            if (park_now) {
                                const unsigned soff = (unsigned)kc * (unsigned)(SW_BCH * 8);
                                const unsigned voff = (unsigned)(w * 64 + lane) * 16u;
                                f64x2 q0, q1;
                                q0.x = V[0]; q0.y = V[1]; q1.x = V[2]; q1.y = V[3];
                                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, q0), rs_kv, voff, soff, 0);
                                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, q1), rs_kv, voff, soff + 4096u, 0);
                            }
                        };
#pragma unroll
                        for (int pr = 0; pr < NP; ++pr) {
                            if (2 * pr == c) micro(2 * pr);
                            if (2 * pr + 1 == c) micro(2 * pr + 1);
                            const bool a0 = 2 * pr > c && 2 * pr < s1, a1 = 2 * pr + 1 > c && 2 * pr + 1 < s1;
                            if (a0) half(pr, 0, 0);
                            if (a1) half(pr, 1, 0);
                            __builtin_amdgcn_sched_barrier(0);
                            if (pr == 4) {
                                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                                __syncthreads();                // barrier i: tile i+1 is complete
                            }
                            if (pr + 1 < NP) {
                                load_a(av[(pr + 1) & 1], slot, pr + 1);
                                if (a0) { half(pr, 0, 1); half(pr, 0, 2); half(pr, 0, 3); }
                                if (a1) { half(pr, 1, 1); half(pr, 1, 2); half(pr, 1, 3); }
                            } else {
                                load_a(av[0], nslot, 0);
                                if (a0) half(pr, 0, 1);
                                if (a1) half(pr, 1, 1);
                                __builtin_amdgcn_sched_barrier(0);
                                load_b(bpar ^ 1, 0);
                                if (a0) { half(pr, 0, 2); half(pr, 0, 3); }
                                if (a1) { half(pr, 1, 2); half(pr, 1, 3); }
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        slot = nslot;
                        bpar ^= 1;
                    };
                    // The same tile with the chunk index C a compile-time constant, for a complete row block (its
                    // sixteen diagonal tiles always come in order): only the sub-blocks below the diagonal block
                    // are touched -- no predicated pairs, no A-fragment reads or joins for the inactive ones, acc[C]
                    // indexed statically --, the first active pair's fragments land during the in-register solve.
                    // Generic form above: ~123 k cycles per row block (the inverse form's diagonal phase: 66 k) --
                    // the substitution form's whole deficit; static form: ~75 k at N = 512.  But the sixteen bodies
                    // are 65 KiB of straight-line code executed once per row block: once the workgroups have
                    // drifted apart and stop sharing instruction-cache lines the phase is fetch-bound (N = 4096:
                    // 310 ms against 276 ms for the generic form; N = 2048: 74 vs 77 ms; N = 1152: 27 vs 31 ms;
                    // N = 512: 5.9 vs 8.5 ms), so it is used up to S2_STATIC_DIAG_NRB row blocks.  (A version
                    // templated on the pair index only, parity as a uniform branch -- a third of the code --
                    // made hipcc spill the accumulators at the join: 2,900 scratch accesses.)
                    auto diag_c = [&](auto c_) {
                        constexpr int C = decltype(c_)::value;
                        constexpr int PF = (C + 1) >> 1;            // first pair with a sub-block below block C
                        const int nslot = slot == 2 ? 0 : slot + 1;
                        load_b(bpar, 1);                            // K*_C, k-steps 2-3 (0-1: prefetched)
                        double ta[10];
                        {
                            const double* Ts = Aring + slot * S2_TILE + C * 256 + (lane & 3) * 4 + (lane >> 4);
                            int e = 0;
#pragma unroll
                            for (int q = 0; q < 4; ++q)
#pragma unroll
                                for (int q2 = 0; q2 <= q; ++q2) ta[e++] = Ts[(q * 4 + q2) * 16];
                        }
                        if constexpr (C < 15) load_a(av[PF & 1], slot, PF);
                        {
                            double rot[4];
                            rot[0] = acc[C][0];
                            rot[1] = apgp_row_ror4<1>(acc[C][1]);
                            rot[2] = apgp_row_ror4<2>(acc[C][2]);
                            rot[3] = apgp_row_ror4<3>(acc[C][3]);
                            const int bb = (lane >> 2) & 3;
                            double R[4], V[4];
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const int rr = (bb - q) & 3;
                                R[q] = (rr == 0 ? rot[0] : rr == 1 ? rot[1] : rr == 2 ? rot[2] : rot[3]) + brot[0][q];
                            }
                            V[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(ta[0], R[0], 0.0, 0, 0, 0);
                            R[1] = __builtin_amdgcn_mfma_f64_4x4x4f64(ta[1], V[0], R[1], 0, 0, 0);
                            R[2] = __builtin_amdgcn_mfma_f64_4x4x4f64(ta[3], V[0], R[2], 0, 0, 0);
                            R[3] = __builtin_amdgcn_mfma_f64_4x4x4f64(ta[6], V[0], R[3], 0, 0, 0);
                            V[1] = __builtin_amdgcn_mfma_f64_4x4x4f64(ta[2], R[1], 0.0, 0, 0, 0);
                            R[2] = __builtin_amdgcn_mfma_f64_4x4x4f64(ta[4], V[1], R[2], 0, 0, 0);
                            R[3] = __builtin_amdgcn_mfma_f64_4x4x4f64(ta[7], V[1], R[3], 0, 0, 0);
                            V[2] = __builtin_amdgcn_mfma_f64_4x4x4f64(ta[5], R[2], 0.0, 0, 0, 0);
                            R[3] = __builtin_amdgcn_mfma_f64_4x4x4f64(ta[8], V[2], R[3], 0, 0, 0);
                            V[3] = __builtin_amdgcn_mfma_f64_4x4x4f64(ta[9], R[3], 0.0, 0, 0, 0);
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                brot[0][q] = V[q];
                                brot[1][q] = apgp_row_ror4<3>(V[q]);
                                brot[2][q] = apgp_row_ror4<2>(V[q]);
                                brot[3][q] = apgp_row_ror4<1>(V[q]);
                                qmic = fma(V[q], V[q], qmic);
                            }
#pragma unroll
                            for (int r = 0; r < 4; ++r) acc[C][r] = 0.0;
                            if (jb + 1 < nrb2) {
                                const unsigned soff = (unsigned)kc * (unsigned)(SW_BCH * 8);
                                const unsigned voff = (unsigned)(w * 64 + lane) * 16u;
                                f64x2 q0, q1;
                                q0.x = V[0]; q0.y = V[1]; q1.x = V[2]; q1.y = V[3];
                                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, q0), rs_kv, voff, soff, 0);
                                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, q1), rs_kv, voff, soff + 4096u, 0);
                            }
                        }
                        auto half = [&](int pr, int h, int kk) {
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                acc[2 * pr + h][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(
                                    av[pr & 1][h][kk >> 1][kk & 1], brot[r][kk], acc[2 * pr + h][r], 0, 0, 0);
                        };
                        // (the two park stores of THIS tile may stay in flight at its barrier -- VMEM completes in
                        // issue order, "at most two outstanding" = the previous tile's are acknowledged; their
                        // readers request them >= 14 barriers later)
                        constexpr int PBAR = PF > 4 ? PF : 4;       // pair after whose first k-step the tile's barrier stands
                        if constexpr (C == 15) {
                            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                            __syncthreads();                        // barrier i: tile i+1 is complete
                            load_a(av[0], nslot, 0);
                            load_b(bpar ^ 1, 0);
                        }
                        s2_static_for<NP>([&](auto pr_) {
                            constexpr int pr = decltype(pr_)::value;
                            if constexpr (pr >= PF && C < 15) {
                                constexpr bool a0 = 2 * pr > C;
                                if constexpr (a0) half(pr, 0, 0);
                                half(pr, 1, 0);
                                __builtin_amdgcn_sched_barrier(0);
                                if constexpr (pr == PBAR) {
                                    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                                    __syncthreads();                // barrier i: tile i+1 is complete
                                }
                                if constexpr (pr + 1 < NP) {
                                    load_a(av[(pr + 1) & 1], slot, pr + 1);
                                    if constexpr (a0) { half(pr, 0, 1); half(pr, 0, 2); half(pr, 0, 3); }
                                    half(pr, 1, 1); half(pr, 1, 2); half(pr, 1, 3);
                                } else {
                                    load_a(av[0], nslot, 0);
                                    if constexpr (a0) half(pr, 0, 1);
                                    half(pr, 1, 1);
                                    __builtin_amdgcn_sched_barrier(0);
                                    load_b(bpar ^ 1, 0);
                                    if constexpr (a0) { half(pr, 0, 2); half(pr, 0, 3); }
                                    half(pr, 1, 2); half(pr, 1, 3);
                                }
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        });
                        slot = nslot;
                        bpar ^= 1;
                        ++kc;
                    };
                    if constexpr (STATIC_DIAG) {
                        if (a.n - S2_ROWS * jb >= S2_ROWS) s2_static_for<S2_CPB>(diag_c);
                    }
                    for (; kc < nkc; ++kc) do_diag();
                }
                // this row block's share of sum V^2: rotation r's accumulators belong to the
                // candidate of lane lr[r]; gather them back, reduce over the 4 row-lanes
                double qr[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    qr[r] = 0.0;
#pragma unroll
                    for (int s_ = 0; s_ < 2 * NP; ++s_) qr[r] = fma(acc[s_][r], acc[s_][r], qr[r]);
                }
                double qs = qr[0];
#pragma unroll
                for (int r = 1; r < 4; ++r) qs += __shfl(qr[r], (lane & 48) | ((lane - 4 * r) & 15));
                if constexpr (SOLVE) qs += qmic;  // (lane = candidate + 16 row there as well)
                qs += __shfl_xor(qs, 16);
                qs += __shfl_xor(qs, 32);
                qtot += qs;
            }
            if (kq == 0) Shq[w * 16 + cl] = qtot;
            __syncthreads();                      // E1: sums visible to the feeders
            __syncthreads();                      // E2: block result written
        }
        return;
    }

    // ================================= feeder role =================================
    const int hw = w - 4, ht = t - 256;
    const unsigned hoff = (unsigned)ht * 16u;
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.linv, 0, (int)a.linv_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)a.xs, 0, (int)a.xs_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_k = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.kcache + (long long)blockIdx.x * a.ncache * SW_BCH), 0, (int)a.kslot_bytes, 0x00020000);
    // an x chunk is XCHUNK16 16-byte pieces, one per feeder thread -- two for the first threads when Dpad = 32 (272 pieces,
    // 256 threads): XQ carries the second piece, dead code for the other instantiations
    constexpr bool X2 = XCHUNK16 > 256;
    struct XQ { f64x2 a, b; };
    const unsigned xoff = ht < XCHUNK16 ? hoff : 0u;
    const unsigned xoff2 = (X2 && ht + 256 < XCHUNK16) ? hoff + 256u * 16u : 0u;
    auto x_fetch = [&](int kc) {
        XQ q;
        q.a = __builtin_bit_cast(f64x2, __builtin_amdgcn_raw_buffer_load_b128(rs_x, xoff, (unsigned)kc * (unsigned)(SW_KC * XS * 8), 0));
        if constexpr (X2) q.b = __builtin_bit_cast(f64x2, __builtin_amdgcn_raw_buffer_load_b128(rs_x, xoff2, (unsigned)kc * (unsigned)(SW_KC * XS * 8), 0));
        return q;
    };
    auto x_put = [&](int xb, const XQ& v) {
        if (ht < XCHUNK16) {
            if (xb == 0) *((f64x2*)Xbuf + ht) = v.a;
            else *((f64x2*)(Xbuf + XSTRIDE) + ht) = v.a;
        }
        if constexpr (X2) {
            if (ht + 256 < XCHUNK16) {
                if (xb == 0) *((f64x2*)Xbuf + ht + 256) = v.b;
                else *((f64x2*)(Xbuf + XSTRIDE) + ht + 256) = v.b;
            }
        }
    };
    const bool park = a.ncache > 0;
    // per-candidate state of the block being fed (cur) and of the finished one (fin)
    double tt[DPAD];
    double ktt_cur = a.amp, ktt_fin = a.amp, mu_cur = 0.0, mu_tot = 0.0, mu_fin = 0.0;
    int fl_cur = 0, fl_fin = 0;
    long long blk_cur = -1, blk_fin = -1;
    auto load_candidates = [&](long long blk) {
        // snapshot the finished block, then set up the new one
        mu_fin = mu_tot; ktt_fin = ktt_cur; fl_fin = fl_cur; blk_fin = blk_cur;
        blk_cur = blk; mu_cur = 0.0; mu_tot = 0.0;
        const long long crow = blk * SW_CAND + hw * 16 + cl;
        const bool inb = blk < a.blk_end && crow < a.m;
        bool adm = inb, has_nan = false;
        double ktl = (LIN && a.lin_order == 0) ? (double)a.ndim : 0.0;
#pragma unroll
        for (int d = 0; d < DPAD; ++d) {
            double v = 0.0;
            if (inb && d < a.ndim) {
                v = a.T[crow * a.ndim + d];
                if (a.has_box && !(v >= Cst[APGP_MAX_DIM + d] && v <= Cst[2 * APGP_MAX_DIM + d])) adm = false;
                if (v != v) has_nan = true;
            }
            tt[d] = v * Cst[d];
            if (LIN && a.lin_order > 0) {
                const double p = v * v;
                double q = p;
                for (int e = 1; e < a.lin_order; ++e) q *= p;
                ktl += q;
            }
        }
        if (inb && a.mask && a.mask[crow] == 0) adm = false;
        fl_cur = (adm ? 1 : 0) | (has_nan ? 2 : 0) | (inb ? 4 : 0);
        ktt_cur = LIN ? fma(a.lin_coef, ktl, a.amp) : a.amp;
    };
    // generate the B operands of tile (jb, kc) of the current block -> LDS buffer bb (and, on
    // the chunk's first visit, the parked stream and this row block's share of mu)
    auto produce_b = [&](int jb, int kc, int bb, int xb) {
        double bfv[NKK];
        {
            const double* Xb = Xbuf + xb * XSTRIDE;
            double s2[NKK], s3[NKK], al[NKK], lsum[NKK];
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) {
                s2[kk] = 0.0; s3[kk] = 0.0;
                lsum[kk] = (LIN && a.lin_order == 0) ? (double)a.ndim : 0.0;
                al[kk] = Xb[(kk * 4 + kq) * XS + DPAD];
            }
#pragma unroll
            for (int d = 0; d < DPAD; d += 2)
#pragma unroll
                for (int kk = 0; kk < NKK; ++kk) {
                    const f64x2 xa = *(const f64x2*)(Xb + (kk * 4 + kq) * XS + d);
                    const double df0 = tt[d] - xa.x, df1 = tt[d + 1] - xa.y;
                    s2[kk] = fma(df0, df0, s2[kk]);
                    s3[kk] = fma(df1, df1, s3[kk]);
                    if (LIN && a.lin_order > 0) {
                        const double p0_ = tt[d] * xa.x * Cst[3 * APGP_MAX_DIM + d], p1_ = tt[d + 1] * xa.y * Cst[3 * APGP_MAX_DIM + d + 1];
                        double q0 = p0_, q1 = p1_;
                        for (int e = 1; e < a.lin_order; ++e) { q0 *= p0_; q1 *= p1_; }
                        lsum[kk] += q0 + q1;
                    }
                }
            double ex[NKK];
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) ex[kk] = -(s2[kk] + s3[kk]);
            apgp_exp4(ex, bfv, Etab);
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) bfv[kk] = LIN ? fma(a.lin_coef, lsum[kk], bfv[kk] * a.amp) : bfv[kk] * a.amp;
            if (kc >= S2_CPB * jb) {
#pragma unroll
                for (int kk = 0; kk < NKK; ++kk) mu_cur = fma(bfv[kk], al[kk], mu_cur);
                if (kc == nkc_of(jb) - 1) {
                    // row block complete: its share of mu, reduced over the 4 k-lanes, joins the
                    // total in row-block order (the order sweep_finish_kernel adds split shares)
                    double m = mu_cur;
                    m += __shfl_xor(m, 16);
                    m += __shfl_xor(m, 32);
                    mu_tot += m;
                    mu_cur = 0.0;
                }
                if (!SOLVE && park && jb + 1 < nrb2) {
                    const unsigned soff = (unsigned)kc * (unsigned)(SW_BCH * 8);
                    f64x2 q0, q1;
                    q0.x = bfv[0]; q0.y = bfv[1]; q1.x = bfv[2]; q1.y = bfv[3];
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, q0), rs_k, hoff, soff, SW_KAUX);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, q1), rs_k, hoff, soff + 256 * 16, SW_KAUX);
                }
            }
            f64x2 o0, o1;
            o0.x = bfv[0]; o0.y = bfv[1]; o1.x = bfv[2]; o1.y = bfv[3];
            if (bb == 0) { f64x2* Bw = (f64x2*)(Bbuf + hw * 256); Bw[lane] = o0; Bw[64 + lane] = o1; }
            else { f64x2* Bw = (f64x2*)(Bbuf + 1024 + hw * 256); Bw[lane] = o0; Bw[64 + lane] = o1; }
        }
    };
    auto epilogue = [&]() {
        __syncthreads();                          // E1
        if (a.split) {
            if (kq == 0) {
                const long long e = ((blk_fin - a.blk_begin) * SW_CAND + hw * 16 + cl) * nrb2 + jb_lo;
                a.sp_q[e] = Shq[hw * 16 + cl];
                a.sp_mu[e] = mu_fin;
            }
            __syncthreads();                      // E2
            return;
        }
        double bu = INFINITY;
        long long bi = -1;
        const long long crow = blk_fin * SW_CAND + hw * 16 + cl;
        if (kq == 0 && (fl_fin & 4)) {
            double mu = mu_fin + a.mean;
            double var = ktt_fin - Shq[hw * 16 + cl];
            if (fl_fin & 2) { mu = NAN; var = NAN; }
            if (a.mu) a.mu[crow] = mu;
            if (a.var) a.var[crow] = var;
            if (a.kind != APGP_UTIL_NONE) {
                const double uu = (fl_fin & 1) ? util_value(a.kind, mu, var, a.zeta, a.ybest) : INFINITY;
                if (a.u) a.u[crow] = uu;
                best_merge(bu, bi, uu, a.idx_offset + crow);
            }
        }
        for (int o = 8; o > 0; o >>= 1) {
            double ou = __shfl_xor(bu, o);
            long long oi = __shfl_xor(bi, o);
            best_merge(bu, bi, ou, oi);
        }
        if (lane == 0) { red_u[hw] = bu; red_i[hw] = bi; }
        __syncthreads();                          // E2
        if (ht == 0 && a.kind != APGP_UTIL_NONE) {
            for (int i = 1; i < 4; ++i) best_merge(bu, bi, red_u[i], red_i[i]);
            a.part_u[blk_fin] = bu;
            a.part_i[blk_fin] = bi;
        }
    };

    // ---- the feeder loop -------------------------------------------------------------------
    // Next to a saturated MFMA stream every feeder instruction waits for an issue slot (~10
    // cycles for a scalar one, ~100 for one that reads or writes VGPRs; tools/mfma_pair.hip),
    // so the per-tile work is a handful of LDS-DMA requests (buffer_load ... lds: no data
    // VGPRs, no ds_write) and scalar bookkeeping.  Tile i+1's image and parked operands are
    // requested right after barrier i-1 into slot (i+1) % 3 / buffer (i+1) & 1 -- both free
    // since barrier i-1 -- and the vmcnt(0) hipcc places in front of barrier i publishes them.
    // Two forms of the feeder loop, chosen per instantiation by same-box A/B (profiles/r03g): next to a
    // saturated MFMA stream the feeder's cost is a matter of how hipcc allocates this loop's scalars (a
    // spilled one read back with v_readlane costs 30-100 cycles), and that differs between the DPAD
    // instantiations.  Structured: +8-13 % at N = 1024-2048 for DPAD 2 / 4 / 16 (inverse form) and for
    // DPAD 2 (substitution form); the single loop is 1.5-2.5 % faster for DPAD 8 (C3, C5).
    if constexpr (SF) {
        const int hw_s = __builtin_amdgcn_readfirstlane(hw);
        typedef __attribute__((address_space(3))) void lds_void;
        // image of tile (jb, kc): 8 pieces of 4 KiB = the 8 sub-block pairs (32 rows each), a quarter
        // of each piece per feeder wavefront.  (Requesting only the non-zero pairs q >= (kc - 16 jb) / 2
        // of a lower-triangular diagonal tile -- 288 instead of 512 KiB per row block -- changes
        // nothing, 248.9 vs 247.4 ms: the diagonal tiles do not wait for the image stream.)
        auto dma_tile = [&](int slot, int jb, int kc) {
            double* dst = Aring + slot * S2_TILE + hw_s * 128;
            const unsigned toffs = tile_off(jb, kc);
#pragma unroll
            for (int q = 0; q < 8; ++q)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void*)(dst + q * 512), 16, (unsigned)lane * 16u,
                                                         toffs + (unsigned)(q * 4096) + (unsigned)hw_s * 1024u, 0, 0);
        };
        auto dma_parked = [&](int par, int c) {
            double* dst = Bbuf + par * 1024 + hw_s * 256;
            const unsigned soff = (unsigned)c * (unsigned)(SW_BCH * 8) + (unsigned)hw_s * 1024u;
            // (SOLVE: the parked blocks were stored by the matrix wavefronts of this workgroup: sc1 = served by
            // the L2, never by a line this CU's vector cache kept from the previous candidate block)
            constexpr int kaux = SOLVE ? 16 : SW_KAUX;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_k, (lds_void*)dst, 16, (unsigned)lane * 16u, soff, 0, kaux);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_k, (lds_void*)(dst + 128), 16, (unsigned)lane * 16u, soff + 4096u, 0, kaux);
        };
        // x chunk kc of the packed training stream straight into x buffer xb by LDS-DMA (whole wavefronts
        // only: the chunk is rounded up to 1 KiB pieces, the extra lanes fetch the bytes that follow it in
        // the stream -- zeros past its end -- into the buffer's padding)
        auto x_dma = [&](int xb, int kc) {
            if (hw_s * 128 < XSTRIDE)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_void*)(Xbuf + xb * XSTRIDE + hw_s * 128), 16,
                                                         (unsigned)ht * 16u, (unsigned)kc * (unsigned)(SW_KC * XS * 8), 0, 0);
        };
        auto nd0_of = [&](int jb) { const int v = S2_CPB * jb, n = nkc_of(jb); return park ? (v < n ? v : n) : 0; };
        auto next_gen = [&](int& jb, int& kc) {
            if (kc + 1 < nkc_of(jb)) { ++kc; return; }
            jb = jb + 1 < jb_hi ? jb + 1 : jb_lo;
            kc = nd0_of(jb);
        };
        const bool xw = hw_s * 128 < XSTRIDE;
#define S2_BARRIER(n) asm volatile("s_waitcnt vmcnt(" #n ") lgkmcnt(0)\n\ts_barrier" ::: "memory")
        __syncthreads();                              // C : constants visible
        load_candidates(blk0);
        dma_tile(0, jb_lo, 0);
        {
            int gj = jb_lo, gk = 0;
            x_dma(0, gk);
            next_gen(gj, gk);
            x_dma(1, gk);
            next_gen(gj, gk);
            x_dma(2, gk);
            __syncthreads();                          // P0 (vmcnt(0): image 0 and the x chunks have landed)
            produce_b(jb_lo, 0, 0, 0);
        }
        __syncthreads();                              // P
        int slot = 1, par = 1, gpar = 1;
        bool first_tile = true;
        for (long long blk = blk0; blk < a.blk_end; blk += blk_step) {
            for (int jb = jb_lo; jb < jb_hi; ++jb) {
                const int nkc = nkc_of(jb), nd0 = nd0_of(jb);
                // ---- run of parked tiles: image + parked operands by LDS-DMA, nothing else ----
                for (int kc = 0; kc < nd0; ++kc) {
                    dma_tile(slot, jb, kc);
                    dma_parked(par, kc);
                    slot = slot == 2 ? 0 : slot + 1;
                    par ^= 1;
                    S2_BARRIER(0);
                }
                // ---- run of generating tiles ----
                for (int kc = nd0; kc < nkc; ++kc) {
                    if (first_tile) { first_tile = false; continue; }      // T_0: the prologue's
                    const bool newblk = (jb == jb_lo && kc == 0);
                    __builtin_amdgcn_s_setprio(3);
                    dma_tile(slot, jb, kc);
                    if (newblk) load_candidates(blk);
                    int xk = kc;                      // chunk of the generating tile after next
                    {
                        int gj = jb;
                        next_gen(gj, xk);
                        next_gen(gj, xk);
                    }
                    const bool stores = !SOLVE && park && kc >= S2_CPB * jb && jb + 1 < nrb2;
                    produce_b(jb, kc, par, gpar);
                    x_dma(gpar == 0 ? 2 : gpar - 1, xk);
                    slot = slot == 2 ? 0 : slot + 1;
                    par ^= 1;
                    gpar = gpar == 2 ? 0 : gpar + 1;
                    __builtin_amdgcn_s_setprio(0);
                    if (stores) { if (xw) S2_BARRIER(3); else S2_BARRIER(2); }
                    else { if (xw) S2_BARRIER(1); else S2_BARRIER(0); }
                    // the matrix wavefronts have passed the barrier of the previous block's last tile
                    if (newblk) epilogue();
                }
            }
        }
        load_candidates(a.blk_end);                   // (no next block: only snapshots the finished one)
        S2_BARRIER(0);                                // barrier of the last tile
        epilogue();
#undef S2_BARRIER
    } else {
        struct Pos { int jb, kc; long long bl; };
        Pos p0 = {jb_lo, 0, blk0}, p1, p2, p3;
        auto next_of = [&](const Pos& p) { Pos q = p; successor(q.jb, q.kc, q.bl); return q; };
        p1 = next_of(p0); p2 = next_of(p1); p3 = next_of(p2);
        const int hw_s = __builtin_amdgcn_readfirstlane(hw);
        typedef __attribute__((address_space(3))) void lds_void;
        // image of tile (jb, kc): 8 pieces of 4 KiB = the 8 sub-block pairs (32 rows each), a quarter
        // of each piece per feeder wavefront.  (Requesting only the non-zero pairs q >= (kc - 16 jb) / 2
        // of a lower-triangular diagonal tile -- 288 instead of 512 KiB per row block -- changes
        // nothing, 248.9 vs 247.4 ms: the diagonal tiles do not wait for the image stream.)
        auto dma_tile = [&](int slot, int jb, int kc) {
            double* dst = Aring + slot * S2_TILE + hw_s * 128;
            const unsigned toffs = tile_off(jb, kc);
#pragma unroll
            for (int q = 0; q < 8; ++q)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void*)(dst + q * 512), 16, (unsigned)lane * 16u,
                                                         toffs + (unsigned)(q * 4096) + (unsigned)hw_s * 1024u, 0, 0);
        };
        auto dma_parked = [&](int par, int c) {
            double* dst = Bbuf + par * 1024 + hw_s * 256;
            const unsigned soff = (unsigned)c * (unsigned)(SW_BCH * 8) + (unsigned)hw_s * 1024u;
            // (SOLVE: the parked blocks were stored by the matrix wavefronts of this workgroup: sc1 = served by
            // the L2, never by a line this CU's vector cache kept from the previous candidate block)
            constexpr int kaux = SOLVE ? 16 : SW_KAUX;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_k, (lds_void*)dst, 16, (unsigned)lane * 16u, soff, 0, kaux);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_k, (lds_void*)(dst + 128), 16, (unsigned)lane * 16u, soff + 4096u, 0, kaux);
        };
        auto is_gen = [&](const Pos& p) { return !park || p.kc >= S2_CPB * p.jb; };
        // ---- prologue: tile 0 and its operands, x chunks of tiles 0 and 1, requests for tile 1 ----
        __syncthreads();                              // C : constants visible
        load_candidates(p0.bl);
        dma_tile(0, p0.jb, p0.kc);
        x_put(0, x_fetch(p0.kc));
        x_put(1, x_fetch(p1.kc));
        __syncthreads();                              // P0
        produce_b(p0.jb, p0.kc, 0, 0);
        XQ xq = x_fetch(p2.kc);
        __syncthreads();                              // P
        // total tiles of this workgroup
        long long ntile_blk = 0;
        for (int jb = jb_lo; jb < jb_hi; ++jb) ntile_blk += nkc_of(jb);
        long long nblk_mine = 0;
        for (long long b = blk0; b < a.blk_end; b += blk_step) ++nblk_mine;
        const long long ntot = ntile_blk * nblk_mine;
        bool last_m1 = (ntile_blk == 1), last_m2 = false;
        int slot = 1, par = 1;
        for (long long i = 0;; ++i) {
            // the matrix wavefronts have just passed barrier i-1: if tile i-1 closed a candidate
            // block, finish that block
            if (last_m2) epilogue();
            if (i == ntot) break;
            // ---- produce tile i+1 = p1 ----
            // (a generating iteration is what the matrix wavefronts wait for on the diagonal tiles: it
            // runs at raised priority -- same box, alternating, 244.4 vs 245.5 ms; raising it for the
            // image requests only is neutral)
            if (is_gen(p1)) __builtin_amdgcn_s_setprio(3);
            dma_tile(slot, p1.jb, p1.kc);
            if (p1.jb == jb_lo && p1.kc == 0) load_candidates(p1.bl);
            if (is_gen(p1)) produce_b(p1.jb, p1.kc, par, par);
            else dma_parked(par, p1.kc);
            x_put(par ^ 1, xq);                       // x chunk of tile i+2
            // VMEM requests of this iteration that barrier i need NOT wait for: the park stores of a
            // first-visit tile (read back a row block later) and the x chunk fetched for tile i+3 (a
            // register load hipcc tracks itself).  They are the LAST requests issued, and gfx9 VMEM
            // completes in issue order, so "at most n outstanding" still means the images and parked
            // operands of tile i+1 have landed -- without the ~1-2 k cycles of store acknowledgement.
            int n_pend = (!SOLVE && is_gen(p1) && park && p1.kc >= S2_CPB * p1.jb && p1.jb + 1 < nrb2) ? 2 : 0;
            if (is_gen(p3)) { xq = x_fetch(p3.kc); n_pend += X2 ? 2 : 1; }
            last_m2 = last_m1;
            last_m1 = (p1.jb == jb_hi - 1 && p1.kc == nkc_of(p1.jb) - 1);
            p0 = p1; p1 = p2; p2 = p3; p3 = next_of(p3);
            slot = slot == 2 ? 0 : slot + 1;
            par ^= 1;
            __builtin_amdgcn_s_setprio(0);
            // barrier i
            if (n_pend == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else if (n_pend == 1) asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else if (n_pend == 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else if (n_pend == 3) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    }
}

// The short last round of the persistent grid (and every launch with fewer candidate blocks
// than CUs) is split by ROW BLOCK: one workgroup per (candidate block, row block) writes its
// share of sum V^2 and of mu, this kernel adds the shares in row-block order (deterministic)
// and finishes mu / sigma^2 / utility / block arg-min exactly like the sweep's own epilogue.
// A candidate block is a serial stream of all its tiles otherwise: 9 blocks left over for
// 256 CUs would keep 247 of them idle for a whole block time.
__global__ __launch_bounds__(64) void sweep_finish_kernel(SweepArgs a) {
    const int c = threadIdx.x;
    const long long blk = a.blk_begin + blockIdx.x;
    const long long crow = blk * SW_CAND + c;
    double bu = INFINITY;
    long long bi = -1;
    if (crow < a.m) {
        bool adm = true, has_nan = false;
        double ktl = a.lin_order == 0 ? (double)a.ndim : 0.0;
        for (int d = 0; d < a.ndim; ++d) {
            const double v = a.T[crow * a.ndim + d];
            if (a.has_box && !(v >= a.lo[d] && v <= a.hi[d])) adm = false;
            if (v != v) has_nan = true;
            if (a.lin_coef != 0.0 && a.lin_order > 0) {
                const double p = v * v;
                double q = p;
                for (int e = 1; e < a.lin_order; ++e) q *= p;
                ktl += q;
            }
        }
        if (a.mask && a.mask[crow] == 0) adm = false;
        const long long e0 = ((long long)blockIdx.x * SW_CAND + c) * a.nrb;
        double q = 0.0, mup = 0.0;
        for (int ib = 0; ib < a.nrb; ++ib) { q += a.sp_q[e0 + ib]; mup += a.sp_mu[e0 + ib]; }
        double mu = mup + a.mean;
        double var = fma(a.lin_coef, ktl, a.amp) - q;
        if (has_nan) { mu = NAN; var = NAN; }
        if (a.mu) a.mu[crow] = mu;
        if (a.var) a.var[crow] = var;
        if (a.kind != APGP_UTIL_NONE) {
            const double uu = adm ? util_value(a.kind, mu, var, a.zeta, a.ybest) : INFINITY;
            if (a.u) a.u[crow] = uu;
            best_merge(bu, bi, uu, a.idx_offset + crow);
        }
    }
    if (a.kind == APGP_UTIL_NONE) return;
    for (int o = 32; o > 0; o >>= 1) {
        double ou = __shfl_xor(bu, o);
        long long oi = __shfl_xor(bi, o);
        best_merge(bu, bi, ou, oi);
    }
    if (c == 0) { a.part_u[blk] = bu; a.part_i[blk] = bi; }
}

// blocks of the last round that are split by row block (measured break-even at N = 4096:
// ~190 of 256 -- the split launch regenerates every k* and its largest item is 1/nrb of a block)
#define S2_SPLIT_MAX 184

static inline int s2_nrb(int64_t n) { return (int)((n + S2_ROWS - 1) / S2_ROWS); }
// chunks per workgroup slot whose B operands are revisited by a later 256-row block
static inline long long s2_ncache(int64_t n) { return (long long)S2_CPB * (s2_nrb(n) - 1); }

// dynamic LDS of sweep2_kernel<DPAD, *>
template <int DPAD>
static constexpr size_t s2_lds_bytes() {      // (the larger of the two feeder forms' x staging)
    return (3 * S2_TILE + 2 * 4 * 256 + 3 * ((SW_KC * (DPAD + 2) * 8 + 1023) / 1024 * 128) + APGP_EXP_TAB_N + SW_CAND + 8 +
            4 * APGP_MAX_DIM) * sizeof(double);
}

// The kernel needs > 64 KiB of dynamic LDS: the attribute is per device (and per kernel
// instantiation), so it is set -- and its result checked -- once for each device this process
// launches on.
template <int DPAD>
static int s2_prepare_device() {
    static bool done[64] = {false};
    static std::mutex mu;                 // (ctypes releases the GIL: host threads may race here)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
        apgp_set_error("apgp_acquire: hipGetDevice failed");
        return -2;
    }
    std::lock_guard<std::mutex> lock(mu);
    if (done[dev]) return 0;
    const int lds = (int)s2_lds_bytes<DPAD>();
    hipError_t e = hipSuccess;
    const void* kernels[6] = {(const void*)sweep2_kernel<DPAD, false, 0>, (const void*)sweep2_kernel<DPAD, true, 0>,
                              (const void*)sweep2_kernel<DPAD, false, 1>, (const void*)sweep2_kernel<DPAD, true, 1>,
                              (const void*)sweep2_kernel<DPAD, false, 2>, (const void*)sweep2_kernel<DPAD, true, 2>};
    for (int i = 0; i < 6 && e == hipSuccess; ++i)
        e = hipFuncSetAttribute(kernels[i], hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) {
        apgp_set_error("apgp_acquire: hipFuncSetAttribute(%d B of LDS) failed on device %d: %s", lds, dev,
                       hipGetErrorString(e));
        return -2;
    }
    done[dev] = true;
    return 0;
}

template <int DPAD>
static int launch_sweep(const SweepArgs& a0, hipStream_t s, bool solve) {
    const int rc = s2_prepare_device<DPAD>();
    if (rc != 0) return rc;
    SweepArgs a = a0;
    const size_t lds = s2_lds_bytes<DPAD>();
    const long long ncb = (a.m + SW_CAND - 1) / SW_CAND;
    const int nrb2 = s2_nrb(a.n);
    a.nrb = nrb2;
    auto launch = [&](unsigned grid) {
        if (solve && nrb2 <= S2_STATIC_DIAG_NRB) {
            if (a.lin_coef != 0.0)
                hipLaunchKernelGGL((sweep2_kernel<DPAD, true, 2>), dim3(grid), dim3(S2_THREADS), lds, s, a);
            else
                hipLaunchKernelGGL((sweep2_kernel<DPAD, false, 2>), dim3(grid), dim3(S2_THREADS), lds, s, a);
        } else if (solve) {
            if (a.lin_coef != 0.0)
                hipLaunchKernelGGL((sweep2_kernel<DPAD, true, 1>), dim3(grid), dim3(S2_THREADS), lds, s, a);
            else
                hipLaunchKernelGGL((sweep2_kernel<DPAD, false, 1>), dim3(grid), dim3(S2_THREADS), lds, s, a);
        } else if (a.lin_coef != 0.0)
            hipLaunchKernelGGL((sweep2_kernel<DPAD, true, 0>), dim3(grid), dim3(S2_THREADS), lds, s, a);
        else
            hipLaunchKernelGGL((sweep2_kernel<DPAD, false, 0>), dim3(grid), dim3(S2_THREADS), lds, s, a);
    };
    // full rounds on the persistent grid, then the remainder split by row block (the substitution
    // form cannot split: its row blocks depend on each other -- the last round just runs short)
    long long rest = ncb % SW_GRID;
    if (solve || nrb2 < 2 || rest > S2_SPLIT_MAX || a.sp_q == NULL) rest = 0;
    const long long full = ncb - rest;
    if (full > 0) {
        a.blk_begin = 0; a.blk_end = full; a.split = 0;
        launch((unsigned)(full < SW_GRID ? full : SW_GRID));
    }
    if (rest > 0) {
        a.blk_begin = full; a.blk_end = ncb; a.split = 1;
        a.ncache = 0;                             // no parking across workgroups
        a.kslot_bytes = SW_BCH * 8;
        launch((unsigned)(rest * nrb2));
        hipLaunchKernelGGL(sweep_finish_kernel, dim3((unsigned)rest), dim3(64), 0, s, a);
    }
    return 0;
}

// Layout of the caller's scratch (doubles): [2 ncb arg-min partials | slots x ncache x SW_BCH
// parked operands | 2 x S2_SPLIT_MAX x SW_CAND x nrb row-block shares of the split last round]
extern "C" int64_t apgp_acquire_work_len(int64_t m, int64_t n) {
    if (m < 1 || n < 1) return 0;
    if (m > APGP_MAX_M || n > APGP_MAX_N) return -1;
    const long long ncb = (m + SW_CAND - 1) / SW_CAND;
    const long long slots = ncb < SW_GRID ? ncb : SW_GRID;
    return 2 * ncb + slots * s2_ncache(n) * SW_BCH + 2 * (long long)S2_SPLIT_MAX * SW_CAND * s2_nrb(n);
}

static int acquire_impl(bool solve, const double* T, int64_t m, int64_t idx_offset, const double* packed_linv,
                        const double* xs, int64_t n, const apgp_kernel_t* kern, double mean,
                        int32_t kind, const double* lo, const double* hi, const uint8_t* mask,
                        double zeta, double ybest, double* mu, double* var, double* u, void* part,
                        apgp_best_t* best, void* stream) {
    APGP_CHECK_ARG(T && packed_linv && xs && kern, "null pointer");
    APGP_CHECK_ARG(m >= 1 && m <= APGP_MAX_M && n >= 1 && n <= APGP_MAX_N, "m >= 1 and n >= 1 required");
    APGP_CHECK_ARG(kind >= APGP_UTIL_AGP && kind <= APGP_UTIL_NONE, "unknown utility kind");
    APGP_CHECK_ARG(kind == APGP_UTIL_NONE || best, "best required for an acquisition");
    APGP_CHECK_ARG(part || (kind == APGP_UTIL_NONE && n <= S2_ROWS),
                   "part (apgp_acquire_work_len doubles) required");
    (void)solve;
    APGP_CHECK_ARG((lo == NULL) == (hi == NULL), "lo and hi must be given together");
    KernConst kc;
    APGP_CHECK_ARG(apgp_make_kernconst(kern, &kc) == 0, "kernel parameters");
    SweepArgs a;
    a.T = T; a.linv = packed_linv; a.xs = xs; a.mask = mask;
    a.mu = mu; a.var = var; a.u = u;
    const long long nblk = (m + SW_CAND - 1) / SW_CAND;
    const long long slots = nblk < SW_GRID ? nblk : SW_GRID;
    a.part_u = (double*)part;
    a.part_i = part ? (long long*)((double*)part + nblk) : NULL;
    a.ncache = (int)s2_ncache(n);
    // N <= 256: nothing is parked; the descriptor then points at the factor (never dereferenced)
    a.kcache = a.ncache > 0 ? (double*)part + 2 * nblk : (double*)packed_linv;
    // row-block shares of the split last round (after the parked-operand slots)
    a.sp_q = part ? (double*)part + 2 * nblk + slots * s2_ncache(n) * SW_BCH : NULL;
    a.sp_mu = a.sp_q ? a.sp_q + (long long)S2_SPLIT_MAX * SW_CAND * s2_nrb(n) : NULL;
    a.blk_begin = 0; a.blk_end = nblk; a.split = 0;
    {
        const long long wb = apgp_packed_linv_len(n) * 8, xb = apgp_packed_train_len(n, kc.ndim) * 8;
        const long long kb = (long long)(a.ncache > 0 ? a.ncache : 1) * SW_BCH * 8;
        APGP_CHECK_ARG(wb < (1ll << 31) && kb < (1ll << 31), "n too large for the sweep's 32-bit stream offsets");
        a.linv_bytes = (unsigned)wb; a.xs_bytes = (unsigned)xb; a.kslot_bytes = (unsigned)kb;
    }
    a.m = m; a.idx_offset = idx_offset;
    a.ndim = kc.ndim; a.nrb = s2_nrb(n); a.kind = kind; a.n = (int)n;
    a.has_box = lo != NULL;
    a.mean = mean; a.amp = kc.amp; a.zeta = zeta; a.ybest = ybest;
    a.lin_coef = kc.lin_coef; a.lin_order = kc.lin_order;
    for (int d = 0; d < APGP_MAX_DIM; ++d) {
        a.sc[d] = kc.sc[d];
        a.lw[d] = kc.lw[d];
        a.lo[d] = (lo && d < kc.ndim) ? lo[d] : 0.0;
        a.hi[d] = (hi && d < kc.ndim) ? hi[d] : 0.0;
    }
    hipStream_t s = (hipStream_t)stream;
    int rc;
    switch (kc.dpad) {
        case 2: rc = launch_sweep<2>(a, s, solve); break;
        case 4: rc = launch_sweep<4>(a, s, solve); break;
        case 8: rc = launch_sweep<8>(a, s, solve); break;
        case 16: rc = launch_sweep<16>(a, s, solve); break;
        default: rc = launch_sweep<32>(a, s, solve); break;
    }
    if (rc != 0) return rc;
    if (kind != APGP_UTIL_NONE)
        hipLaunchKernelGGL(argmin_final_kernel, dim3(1), dim3(1024), 0, s, a.part_u, a.part_i, nblk, best);
    APGP_CHECK_LAUNCH();
    return 0;
}

extern "C" int apgp_acquire(const double* T, int64_t m, int64_t idx_offset, const double* packed_linv,
                            const double* xs, int64_t n, const apgp_kernel_t* kern, double mean,
                            int32_t kind, const double* lo, const double* hi, const uint8_t* mask,
                            double zeta, double ybest, double* mu, double* var, double* u, void* part,
                            apgp_best_t* best, void* stream) {
    return acquire_impl(false, T, m, idx_offset, packed_linv, xs, n, kern, mean, kind, lo, hi, mask, zeta, ybest,
                        mu, var, u, part, best, stream);
}

// ---------------------------------------------------------------------------
// Substitution form of the sweep: sigma^2 = amp - |v|^2 with v = L^-1 k* by BLOCKED FORWARD
// SUBSTITUTION against the factor itself (what george's cho_solve does, utility.py:131,178,224)
// instead of a product with the explicit inverse -- the only trustworthy formulation once
// cond(K) approaches 1e16 (the reference's own fitAmp=True optimum), and
// the same matrix-core stream otherwise: sweep2_kernel<.., SOLVE = true> reads the tiles of
// apgp_pack_lsolve (-L off the diagonal 16 x 16 blocks, those blocks prepared for a 4 x 4-blocked
// in-register solve), its B operands are the solved blocks V_j parked by the matrix wavefronts,
// and k* enters once per 16-row block.  No n limit beyond the inverse form's.
// ---------------------------------------------------------------------------
extern "C" int apgp_acquire_solve(const double* T, int64_t m, int64_t idx_offset, const double* packed_lsolve,
                                  const double* xs, int64_t n, const apgp_kernel_t* kern, double mean,
                                  int32_t kind, const double* lo, const double* hi, const uint8_t* mask,
                                  double zeta, double ybest, double* mu, double* var, double* u, void* part,
                                  apgp_best_t* best, void* stream) {
    return acquire_impl(true, T, m, idx_offset, packed_lsolve, xs, n, kern, mean, kind, lo, hi, mask, zeta, ybest,
                        mu, var, u, part, best, stream);
}

// Packed tiles of the substitution form, same geometry as apgp_trtri_pack's (512 x 16 tiles, MFMA
// A-fragment order, row block ib holds chunks 0 .. 32 (ib + 1) - 1):
//   * below the diagonal 16 x 16 block of a chunk: -L;
//   * sub-block slot of the diagonal block C itself (rows 16 C .. 16 C + 15 of chunk C): 256 doubles
//     Ts[(4 q + q') 16 + 4 i + k] = -L[16C+4q+i][16C+4q'+k] for q' < q, (L_qq)^-1[i][k] for q' = q, 0 above --
//     the 4 x 4 diagonal blocks are inverted by substitution with reciprocal pivots; rows / columns
//     past n are zero (so the padded rows of V are exact zeros).
// 4 x 4 is the largest diagonal block whose explicit inverse keeps the solve in cho_solve's error
// class at cond 8.5e15 (8 x 8: borderline, 16 x 16: 10-100x worse; tools/cond_blocking_study.py).
__global__ __launch_bounds__(256) void pack_lsolve_kernel(const double* L, long long ldl, long long n,
                                                          double* packed) {
    constexpr int CPB = APGP_ROW_BLOCK / APGP_K_CHUNK;
    const long long tile = blockIdx.x;
    long long ib = (long long)((sqrt(8.0 * (double)tile / CPB + 1.0) - 1.0) * 0.5);
    while (CPB * (ib + 1) * (ib + 2) / 2 <= tile) ++ib;
    while (CPB * ib * (ib + 1) / 2 > tile) --ib;
    const long long kc = tile - CPB * ib * (ib + 1) / 2;
    double* out = packed + tile * (long long)(APGP_ROW_BLOCK * APGP_K_CHUNK);
    const long long sdiag = kc - ib * CPB;                 // sub-block slot of the diagonal block (if in this tile)
    for (int e = threadIdx.x; e < APGP_ROW_BLOCK * APGP_K_CHUNK; e += 256) {
        const int s = e >> 8;
        double v = 0.0;
        if (s == sdiag) {
            const int f = e & 255, q = f >> 6, q2 = (f >> 4) & 3, i = (f >> 2) & 3, k = f & 3;
            const long long r0 = kc * APGP_K_CHUNK + 4 * q, c0 = kc * APGP_K_CHUNK + 4 * q2;
            if (q2 < q) {
                if (r0 + i < n) v = -L[(r0 + i) * ldl + c0 + k];
            } else if (q2 == q && k <= i && r0 + i < n) {
                // column k of the inverse of the 4 x 4 lower-triangular block, rows k .. i
                double x[4] = {0.0, 0.0, 0.0, 0.0};
                x[k] = 1.0 / L[(r0 + k) * ldl + r0 + k];
                for (int r = k + 1; r <= i; ++r) {
                    double sacc = 0.0;
                    for (int c = k; c < r; ++c) sacc = fma(L[(r0 + r) * ldl + r0 + c], x[c], sacc);
                    x[r] = -sacc * (1.0 / L[(r0 + r) * ldl + r0 + r]);
                }
                v = x[i];
            }
        } else {
            const int q = e & 1, lane = (e >> 1) & 63, kp = (e >> 7) & 1;
            const int kk = 2 * kp + q;
            const long long row = ib * APGP_ROW_BLOCK + 16 * s + (lane & 15);
            const long long col = kc * APGP_K_CHUNK + 4 * kk + (lane >> 4);
            if (row < n && (col >> 4) < (row >> 4)) v = -L[row * ldl + col];
        }
        out[e] = v;
    }
}

extern "C" int64_t apgp_packed_lsolve_len(int64_t n) { return apgp_packed_linv_len(n); }

extern "C" int apgp_pack_lsolve(const double* L, int64_t n, int64_t ldl, double* packed, void* stream) {
    APGP_CHECK_ARG(L && packed, "null pointer");
    APGP_CHECK_ARG(n >= 1 && n <= APGP_MAX_N && ldl >= n, "n >= 1 and ldl >= n required");
    const long long nrb = apgp_npad(n) / APGP_ROW_BLOCK;
    const long long ntiles = (APGP_ROW_BLOCK / APGP_K_CHUNK) * nrb * (nrb + 1) / 2;
    hipLaunchKernelGGL(pack_lsolve_kernel, dim3((unsigned)ntiles), dim3(256), 0, (hipStream_t)stream, L,
                       (long long)ldl, (long long)n, packed);
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
    int ndim, lin_order;
    double mean, amp, lin_coef;
    double sc[APGP_MAX_DIM], lw[APGP_MAX_DIM];
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
        double kv = a.amp * apgp_exp(-(s + s3), etab);
        if (a.lin_coef != 0.0) {
            double ls;
            APGP_LIN_SUM(ls, DPAD, a.ndim, a.lin_order, tt[d_] * xr[d_] * a.lw[d_]);
            kv = fma(a.lin_coef, ls, kv);
        }
        acc = fma(kv, xr[DPAD], acc);
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    bool bad = false;
    for (int d = 0; d < DPAD; ++d) bad = bad || (tt[d] != tt[d]);
    if (lane == 0) a.mu[row] = bad ? NAN : acc + a.mean;
}

extern "C" int apgp_predict_mean(const double* T, int64_t m, const double* xs, int64_t n,
                                 const apgp_kernel_t* kern, double mean, double* mu, void* stream) {
    APGP_CHECK_ARG(T && xs && kern && mu, "null pointer");
    APGP_CHECK_ARG(m >= 1 && m <= APGP_MAX_M && n >= 1 && n <= APGP_MAX_N, "m >= 1 and n >= 1 required");
    KernConst kc;
    APGP_CHECK_ARG(apgp_make_kernconst(kern, &kc) == 0, "kernel parameters");
    MeanArgs a;
    a.T = T; a.xs = xs; a.mu = mu; a.m = m; a.npad = apgp_npad(n); a.ndim = kc.ndim;
    a.mean = mean; a.amp = kc.amp; a.lin_coef = kc.lin_coef; a.lin_order = kc.lin_order;
    for (int d = 0; d < APGP_MAX_DIM; ++d) { a.sc[d] = kc.sc[d]; a.lw[d] = kc.lw[d]; }
    dim3 grid((unsigned)((m + 3) / 4)), block(256);
    hipStream_t s = (hipStream_t)stream;
    switch (kc.dpad) {
        case 2: hipLaunchKernelGGL(predict_mean_kernel<2>, grid, block, 0, s, a); break;
        case 4: hipLaunchKernelGGL(predict_mean_kernel<4>, grid, block, 0, s, a); break;
        case 8: hipLaunchKernelGGL(predict_mean_kernel<8>, grid, block, 0, s, a); break;
        case 16: hipLaunchKernelGGL(predict_mean_kernel<16>, grid, block, 0, s, a); break;
        default: hipLaunchKernelGGL(predict_mean_kernel<32>, grid, block, 0, s, a); break;
    }
    APGP_CHECK_LAUNCH();
    return 0;
}

// Host-buffer form of apgp_predict_mean for the latency-bound caller: an ensemble
// sampler's half-step evaluates a few dozen points per call (approx.py:178-180 through
// emcee), 4e4 calls per chain.  One entry point; round 3: the points are staged in -- and
// the means written to -- a pinned, device-mapped area of the stream that the kernel reads
// and writes in place, and a one-thread kernel behind it posts the sequence word the host
// polls: no H2D / D2H copy, no stream synchronisation (without pinned memory: the copies
// through `work` and one synchronisation, as in round 2).
__global__ void mailbox_post_kernel(double* mail, long long seq) {
    __hip_atomic_store((long long*)(mail + 5), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

extern "C" int apgp_predict_mean_host(const double* T_host, int64_t m, const double* xs, int64_t n,
                                      const apgp_kernel_t* kern, double mean, double* mu_host,
                                      double* work, void* stream) {
    APGP_CHECK_ARG(T_host && mu_host && work && kern, "null pointer");
    APGP_CHECK_ARG(m >= 1 && m <= APGP_MAX_M, "m >= 1 required");
    APGP_CHECK_ARG(kern->ndim >= 1 && kern->ndim <= APGP_MAX_DIM, "kernel parameters");   // (sizes the staging area below)
    hipStream_t s = (hipStream_t)stream;
    const size_t tn = (size_t)m * (size_t)kern->ndim;
    {
        std::lock_guard<std::mutex> lock(apgp_stream_lock(s));
        double* io_dev = nullptr;
        double* io = apgp_stream_pinned_io(s, tn + (size_t)m, &io_dev);
        ApgpMailbox* mb = io ? apgp_stream_mailbox(s) : nullptr;
        if (io && mb && mb->host) {
            memcpy(io, T_host, tn * sizeof(double));
            const int rc = apgp_predict_mean(io_dev, m, xs, n, kern, mean, io_dev + tn, stream);
            if (rc != 0) return rc;
            const long long seq = ++mb->seq;
            hipLaunchKernelGGL(mailbox_post_kernel, dim3(1), dim3(1), 0, s, mb->dev, seq);
            APGP_CHECK_LAUNCH();
            volatile long long* flag = (volatile long long*)(mb->host + 5);
            const auto t0 = std::chrono::steady_clock::now();
            unsigned spins = 0;
            while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) {
                if ((++spins & 255u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(400)) {
                    if (hipStreamSynchronize(s) != hipSuccess || __atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) {
                        apgp_set_error("apgp_predict_mean_host: result not posted");
                        return -2;
                    }
                    break;
                }
            }
            memcpy(mu_host, io + tn, (size_t)m * sizeof(double));
            return 0;
        }
    }
    const size_t tb = tn * sizeof(double);
    double* T_dev = work;
    double* mu_dev = work + tn;
    if (hipMemcpyAsync(T_dev, T_host, tb, hipMemcpyHostToDevice, s) != hipSuccess) {
        apgp_set_error("apgp_predict_mean_host: H2D copy failed");
        return -2;
    }
    const int rc = apgp_predict_mean(T_dev, m, xs, n, kern, mean, mu_dev, stream);
    if (rc != 0) return rc;
    if (hipMemcpyAsync(mu_host, mu_dev, (size_t)m * sizeof(double), hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess) {
        apgp_set_error("apgp_predict_mean_host: D2H copy failed");
        return -2;
    }
    return 0;
}
