// Blocked right-looking Cholesky factorisation (lower, row-major, in place) for
// the GP fit: replaces scipy.linalg.cholesky inside george BasicSolver.compute
// (every gpUtils._nll evaluation, gpUtils.py:74-78, and GP.compute,
// gpUtils.py:178; approx.py:717).  rocSOLVER's dpotrf reaches ~2 TFLOP/s at
// N = 4096 on MI355X (11.8 ms, dozens of tiny kernels); this version is ONE
// launch per 64-column block step (1.9 ms):
//   potrf_panel_kernel (block column 0 only) : the panel step alone;
//   potrf_step_kernel(j)                     : trailing update of block column j, A_ik -= L_ij L_kj^T
//           (64x64x64 tiles on the four-block f64 MFMA, mma16.h), AND the panel step of block
//           column j + 1 by the workgroups of the first trailing tile column (look-ahead);
//   panel step: every workgroup re-factorises the 64x64 diagonal block in registers (redundantly
//           -- it is the critical path either way, and it removes every dependency between
//           workgroups) while its second wavefront solves 64 panel rows against it and a third
//           applies the first column groups to the right half of the diagonal block;
//   potrf_finish_kernel: info, and the factored diagonal blocks from the scratch into the matrix.
// info follows LAPACK: 0 = OK, k > 0 = leading minor of order k not positive
// definite (first failing pivot).
#include "apgp_common.h"
#include "mma16.h"
#include "scratch.h"
#include <atomic>
#include <chrono>
#include <mutex>
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
#define CB 4     // columns per step of the diagonal-block factorisation (potrf_panel_kernel)

struct PotrfArgs {
    double* A;
    double* rhs;         // optional right-hand side carried through the factorisation
    long long n, lda;
    long long j0;        // first column of the current block
    double shift;        // rhs is used as (rhs - shift); applied at block step 0 .. as read
    int* info;
    // factored diagonal blocks, nb x 64 x 64 per matrix (NULL: written straight into A).  A panel
    // workgroup reads the raw block from A when it starts and stores rows of the factor when it ends;
    // sibling workgroups are not guaranteed to start together, so the factor goes here and is copied
    // into A by potrf_finish_kernel.
    double* out5;            // optional (apgp_nll_eval): potrf_finish_kernel also writes the 5-value fit summary
    double* mail;            // optional: ... and (matrix 0) into the stream's pinned mailbox, sequence word last
    long long seq;
    double* dscr;
    long long batch_dscr;    // per matrix: nb * 64 * 64 factor blocks, then nb * 64 entries of the forward solve (same hazard)
    long long zoff;          // offset of the latter
    // batched factorisation (apgp_nll_eval_batch): blockIdx.y selects the matrix
    long long batch_A, batch_rhs;   // element strides between consecutive matrices / right-hand sides
    // persistent factorisation (potrf_persist.h): *abort_word == abort_id <=> the launch gave up; potrf_finish_kernel
    // then reports PP_ABORTED in the summary's info slot and the host re-runs on the multi-launch path
    const unsigned long long* abort_word;
    unsigned long long abort_id;
    // batched persistent factorisation (round 6): matrix y's abort word / mailbox record lie y * batch_abort words /
    // y * batch_mail doubles further on (0, 0: one abort word, the record of matrix 0 only -- everything before round 6)
    long long batch_abort = 0, batch_mail = 0;
    // hybrid factorisation (large n: the first block columns by the launch-per-step path, the rest by ONE persistent
    // launch on the trailing matrix): the step that hands over applies its block column to the whole trailing matrix and
    // writes the first tile column back instead of carrying on with the panel (no_panel); the persistent launch sees the
    // trailing matrix as its own (A, rhs, dscr shifted) and reports failing pivots in the full matrix's numbering (info_j0)
    int no_panel;
    long long info_j0;
    // paired trailing updates (large trailing matrices, where a step is bound by the update's memory traffic): a NARROW
    // step (pair_mode 1) applies its block column to the first TWO tile columns only, the WIDE step after it (2) applies
    // both block columns to every other tile in ONE pass over the tile -- two products, each accumulated from zero and
    // subtracted in turn: the bits of two separate passes, half their traffic.  0 = every step applies its own column.
    int pair_mode;
    // DEFERRED tiles (round 5): a narrow step is a latency chain (first tile + panel, ~25 us) on ~2 tb workgroups with the
    // rest of the chip idle, the wide step before it is bound by its update.  A wide step therefore leaves `defer8` eighths
    // of its tiles right of the next narrow step's two tile columns (block columns >= 3 of its trailing matrix; tile e of
    // that triangle is deferred iff e % 8 < defer8) to the NEXT launch, whose `deferred8` says so: extra workgroups of the
    // narrow step apply the two earlier block columns to them, exactly as the wide step would have (same products, same
    // order: same bits), while the panel chain runs.  0 = nothing deferred.
    int defer8, deferred8;
};

// tile e (row r, column c of the lower triangle, 0 <= c <= r) of a triangle, e = r (r + 1) / 2 + c
__device__ __forceinline__ void potrf_tri(long long e, long long& r, long long& c) {
    long long b = (long long)((sqrt(8.0 * (double)e + 1.0) - 1.0) * 0.5);
    while ((b + 1) * (b + 2) / 2 <= e) ++b;
    while (b * (b + 1) / 2 > e) --b;
    r = b;
    c = e - b * (b + 1) / 2;
}
// the k-th deferred (e % 8 < d8) / kept (e % 8 >= d8) index
__device__ __host__ __forceinline__ long long potrf_deferred_index(long long k, int d8) { return (k / d8) * 8 + (k % d8); }
__device__ __host__ __forceinline__ long long potrf_kept_index(long long k, int d8) { return (k / (8 - d8)) * 8 + d8 + (k % (8 - d8)); }
__device__ __host__ __forceinline__ long long potrf_deferred_count(long long E, int d8) {
    return E <= 0 || d8 <= 0 ? 0 : (E / 8) * d8 + ((E % 8) < d8 ? (E % 8) : d8);
}

// matrix of this workgroup in a batched launch (gridDim.y = batch size; strides 0 otherwise)
__device__ __forceinline__ void potrf_select(PotrfArgs& a) {
    a.A += (long long)blockIdx.y * a.batch_A;
    if (a.rhs) a.rhs += (long long)blockIdx.y * a.batch_rhs;
    a.info += blockIdx.y;
    if (a.dscr) a.dscr += (long long)blockIdx.y * a.batch_dscr;
    if (a.out5) a.out5 += 5 * (long long)blockIdx.y;
    if (a.abort_word) a.abort_word += (long long)blockIdx.y * a.batch_abort;
    if (a.mail) a.mail += (long long)blockIdx.y * a.batch_mail;
}

// Panel step of block column j, three wavefronts per workgroup; workgroup b owns the 64 panel rows
// [j0 + 64 + 64 b, +64) and re-factorises the 64x64 diagonal block itself (it is the critical path
// either way; redundancy removes every dependency between workgroups).
//
// Wavefront 0 (panel_factor_wave) factorises the diagonal block in REGISTERS: lane i holds row i,
// right-looking, fully unrolled, in groups of CB = 4 columns.  Per group: the 4x4 diagonal
// sub-block is broadcast (v_readlane) and factorised BY EVERY LANE ALIKE -- v_rsq_f64 + one
// third-order step per pivot, no cross-lane step, no LDS round trip, no branch --, every lane solves
// its own four entries against it (rows of the sub-block reproduce the factor bit for bit), the
// group is published (columns of L_jj in Ls, reciprocal pivots in invd, progress counter), and the
// columns to the right get their rank-4 update from broadcast ds_read_b128 (panel_trailing).  The
// forward solve z = L^-1 (y - mean) rides along in the same groups.  One wavefront, in-order LDS:
// no barrier anywhere.  (Round 1 / early round 2: one pivot at a time through an LDS line, 670
// cycles per pivot; a 64-lane store to ONE LDS address serialises -- publishing invd from every
// lane cost ~1000 cycles per group.)
// Wavefront 1 (panel_solve_wave) solves the workgroup's 64 panel rows, x L_jj^T = a, one row per
// lane, the same way and one group behind.
// Wavefront 2 (panel_helper_wave) applies groups 0 .. 7 to columns 32 .. 63 of the diagonal block's
// rows while wavefront 0 applies them to columns < 32 only, and hands the columns back before
// group 8: the rank-4 updates are bound by uniform-address LDS reads PER WAVEFRONT (~20 cycles per
// ds_read_b128, tools/probes/lat_probe.hip), so a second wavefront nearly halves them (factorisation at
// column 2048: 16.5 -> 12.8 us).
// Same arithmetic in the same order per element whichever wavefront applies it.
//
// Scheduling fences for the straight-line code: the asm memory clobber stops the SelectionDAG
// from hoisting the (address-independent) LDS reads of later steps, the sched_barrier stops the
// machine scheduler -- without both, hundreds of reads are in flight at once and the 64-double
// register rows spill.
#ifdef APGP_PANEL_TIMING
__device__ unsigned long long apgp_panel_stamps[16];
__device__ unsigned long long apgp_step_stamps[16];   // 100 MHz wall clock, fused step at column 2048: workgroups 0 and 1
#define STEP_STAMP(i) do { if (base == 2048 && bi < 2 && (threadIdx.x & 63) == 0) apgp_step_stamps[(i) + 6 * bi] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define PANEL_STAMP(i) do { if (wb_index == 0 && (threadIdx.x & 63) == 0 && j0 == 2048) apgp_panel_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PANEL_STAMP(i) do { } while (0)
#define STEP_STAMP(i) do { } while (0)
#endif
// broadcast of lane `src` (compile-time constant): two v_readlane_b32 (a few cycles) --
// __shfl with a constant lane still goes through ds_bpermute (an LDS round trip)
__device__ __forceinline__ double bcast_lane(double v, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}
#ifdef PP_STAMPS     // (persistent kernel, stamped build: the factorising wavefront of the last row workgroup)
__device__ unsigned long long pp_fstamps[64 * 8];
#define PANEL_HOOK(i_) do { if ((int)blockIdx.x == (int)((a.n + PB - 1) / PB) - 1 && lane == 0) pp_fstamps[(j0 / PB) * 8 + (i_)] = __builtin_amdgcn_s_memrealtime(); } while (0)
__device__ unsigned long long pp_gstamps[64 * 8];   // shader-clock stamps inside the group of columns 52 .. 55
#define PANEL_GHOOK(i_) do { if (c0 == 52 && (int)blockIdx.x == (int)((a.n + PB - 1) / PB) - 1 && lane == 0) pp_gstamps[(j0 / PB) * 8 + (i_)] = __builtin_amdgcn_s_memtime(); } while (0)
extern "C" int apgp_debug_read_gstamps(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(pp_gstamps), sizeof(unsigned long long) * 64 * 8) == hipSuccess ? 0 : -2;
}
extern "C" int apgp_debug_read_fstamps(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(pp_fstamps), sizeof(unsigned long long) * 64 * 8) == hipSuccess ? 0 : -2;
}
#else
#define PANEL_HOOK(i_) do { } while (0)
#define PANEL_GHOOK(i_) do { } while (0)
#endif
#define PANEL_FENCE() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
// every LDS spin between wavefronts is bounded (2^18 polls, tens of ms): a role that never publishes makes the result wrong,
// never the GPU hang
// PANEL_SPIN_WHILE_T: an expired guard also raises the LDS word `trip_` (if any): the persistent kernel turns that into its
// global abort word, so that the host re-runs the evaluation instead of trusting values computed from stale operands
#define PANEL_SPIN_WHILE_T(cond, trip_) do { unsigned guard_ = 0; while (cond) { __builtin_amdgcn_s_sleep(1); if (++guard_ > (1u << 18)) { if (trip_) lds_store_volatile((trip_), 1); break; } } } while (0)
#define PANEL_SPIN_WHILE(cond) PANEL_SPIN_WHILE_T(cond, (int*)nullptr)
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
// rank-CB update of columns J0 .. J1 - 1 of a register row with the row's new entries xs (columns
// C0 .. C0 + CB - 1): row[j] -= sum_k xs[k] L[j][C0 + k], k ascending.  Row j of the new
// columns is a broadcast read of the published factor.  Four columns advance together (a chain of
// dependent f64 FMAs issues one per ~10 cycles, four independent ones one per ~4.4), two k per
// step; the reads of a step are requested TR_AHEAD steps before their use.
#define TR_AHEAD 4
template <int C0, int J0, int J1, int OFF, int LEN>
__device__ __forceinline__ void panel_trailing(double (&row)[LEN], const double (&xs)[CB], const double (*Ls)[PB + 2]) {
    // columns J0 .. J1 - 1; row[j - OFF] holds column j
    constexpr int NJ = J1 - J0, NS = (NJ > 0 ? NJ / 4 : 0) * (CB / 2);
    static_assert(NJ <= 0 || NJ % 4 == 0, "column groups of four");
    if constexpr (NS > 0) {
        f64x2 lb[TR_AHEAD][4];
        auto request = [&](auto s_) {
            constexpr int s = decltype(s_)::value, g = s / (CB / 2), kp = s % (CB / 2);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) lb[s % TR_AHEAD][jj] = *(const f64x2*)(&Ls[J0 + 4 * g + jj][C0 + 2 * kp]);
        };
        static_for<(TR_AHEAD < NS ? TR_AHEAD : NS)>(request);
        PANEL_FENCE();
        static_for<NS>([&](auto s_) {
            constexpr int s = decltype(s_)::value, g = s / (CB / 2), kp = s % (CB / 2), slot = s % TR_AHEAD;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) row[J0 - OFF + 4 * g + jj] = fma(-xs[2 * kp], lb[slot][jj].x, row[J0 - OFF + 4 * g + jj]);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) row[J0 - OFF + 4 * g + jj] = fma(-xs[2 * kp + 1], lb[slot][jj].y, row[J0 - OFF + 4 * g + jj]);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) asm volatile("" : "+v"(row[J0 - OFF + 4 * g + jj]));
            if constexpr (s + TR_AHEAD < NS) request(std::integral_constant<int, s + TR_AHEAD>{});
            PANEL_FENCE();
        });
    }
}
// Helper wavefront of a panel step: while the factorising wavefront works through the column
// groups left of HELPER_COL0 (and applies them to the columns left of it only), this one applies
// them to columns HELPER_COL0 .. PB - 1 of the same 64 rows, one row per lane, from the groups as
// they are published -- the rank-4 updates are bound by uniform-address LDS reads per wavefront
// (~20 cycles per ds_read_b128), and a second wavefront has its own.  The updated columns go back
// through the staging positions Ls[lane][HELPER_COL0 ..] (not yet published at that point).
#define HELPER_COL0 32
__device__ __forceinline__ void panel_helper_wave(const int bs, const int lane, double (*Ls)[PB + 2], int* prog_p, int* hflag_p,
                                                  int* trip_p = nullptr) {
#define prog (*prog_p)
    double hi[PB - HELPER_COL0];
#pragma unroll
    for (int k = HELPER_COL0; k < PB; k += 2) {
        const f64x2 q = *(const f64x2*)(&Ls[lane][k]);
        hi[k - HELPER_COL0] = (lane < bs && k <= lane) ? q.x : ((k == lane) ? 1.0 : 0.0);
        hi[k + 1 - HELPER_COL0] = (lane < bs && k + 1 <= lane) ? q.y : ((k + 1 == lane) ? 1.0 : 0.0);
    }
    PANEL_FENCE();
    static_for<HELPER_COL0 / CB>([&](auto cc_) {
        constexpr int c0 = CB * decltype(cc_)::value;
        PANEL_SPIN_WHILE_T(lds_load_volatile(&prog) < c0 + CB, trip_p);
        PANEL_FENCE();
        double xs[CB];
#pragma unroll
        for (int k = 0; k < CB; k += 2) {
            const f64x2 q = *(const f64x2*)(&Ls[lane][c0 + k]);
            xs[k] = q.x;
            xs[k + 1] = q.y;
        }
        PANEL_FENCE();
        if constexpr (c0 + CB < HELPER_COL0) panel_trailing<c0, HELPER_COL0, PB, HELPER_COL0, PB - HELPER_COL0>(hi, xs, Ls);
        else {
            // the last group: the factorising wavefront is about to wait for these columns -- the four it needs for its
            // next group go back first (hflag = 1), the rest while it factorises that group (hflag = 2).  Handing all 32
            // back at once left it waiting 1.5-2 us per panel step.
            panel_trailing<c0, HELPER_COL0, HELPER_COL0 + CB, HELPER_COL0, PB - HELPER_COL0>(hi, xs, Ls);
#pragma unroll
            for (int k = HELPER_COL0; k < HELPER_COL0 + CB; k += 2) *(f64x2*)(&Ls[lane][k]) = (f64x2){hi[k - HELPER_COL0], hi[k + 1 - HELPER_COL0]};
            lds_store_volatile(hflag_p, 1);              // (same wavefront: LDS stores stay in order)
            PANEL_FENCE();
            panel_trailing<c0, HELPER_COL0 + CB, PB, HELPER_COL0, PB - HELPER_COL0>(hi, xs, Ls);
        }
    });
#pragma unroll
    for (int k = HELPER_COL0 + CB; k < PB; k += 2) *(f64x2*)(&Ls[lane][k]) = (f64x2){hi[k - HELPER_COL0], hi[k + 1 - HELPER_COL0]};
    lds_store_volatile(hflag_p, 2);
#undef prog
}

// The two wavefront roles of a panel step (block column j0, bs = its width).  The caller loads
// the operands: `ar` = row `lane` of the diagonal block (zero above the diagonal, identity rows
// past bs), `ri` = the right-hand-side entry of that row; `x` = panel row `row` of the workgroup.
// Ls / invd / zblk / prog: the workgroup's LDS exchange area (prog zeroed before the first use).
// RHS = false: no right-hand side rides along (the persistent kernel's forward solve is a pass of its own)
template <bool RHS = true>
__device__ __forceinline__ void panel_factor_wave(PotrfArgs& a, const long long j0, const int bs, const int lane,
                                                  double (&ar)[PB], double ri, double (*Ls)[PB + 2], double* invd,
                                                  double* zblk, int* prog_p, const int* hflag_p, const int wb_index,
                                                  const int wb_count, int* trip_p = nullptr) {
#define prog (*prog_p)
    PANEL_FENCE();
    PANEL_STAMP(1);
    PANEL_HOOK(0);
    int firstbad = 0x7fffffff;               // (uniform) first pivot that is not positive and finite
    static_for<PB / CB>([&](auto cc_) {
        constexpr int c0 = CB * decltype(cc_)::value;
        if constexpr (c0 == HELPER_COL0) {
            // columns HELPER_COL0 .. of this row come back from the helper wavefront, updated with
            // the groups before this one (panel_helper_wave)
            // (the four columns of this group first; the others are awaited before this group's trailing update)
            PANEL_HOOK(1);
            PANEL_SPIN_WHILE_T(lds_load_volatile(hflag_p) == 0, trip_p);
            PANEL_FENCE();
            PANEL_HOOK(2);
#pragma unroll
            for (int k = HELPER_COL0; k < HELPER_COL0 + CB; k += 2) {
                const f64x2 q = *(const f64x2*)(&Ls[lane][k]);
                ar[k] = q.x;
                ar[k + 1] = q.y;
            }
            PANEL_FENCE();
        }
        // (1) the CB x CB diagonal block, as updated so far, and the CB right-hand-side entries
        // into every lane (uniform registers)
        double d[CB][CB], zb[CB];
        if (c0 == 52) { PANEL_FENCE(); PANEL_GHOOK(0); PANEL_FENCE(); }
#pragma unroll
        for (int r = 0; r < CB; ++r) {
#pragma unroll
            for (int q = 0; q <= r; ++q) d[r][q] = bcast_lane(ar[c0 + q], c0 + r);
            if constexpr (RHS) zb[r] = bcast_lane(ri, c0 + r);
        }
        if (c0 == 0) { PANEL_FENCE(); PANEL_STAMP(6); }
        if (c0 == 52) { PANEL_FENCE(); PANEL_GHOOK(1); PANEL_FENCE(); }
        // (2) its factor, computed by all lanes alike: the serial chain of CB pivots runs on
        // registers alone -- no cross-lane step, no LDS round trip, no branch per pivot.  Per
        // pivot: v_rsq_f64 seed, one third-order step (r (1 + e/2 + 3 e^2/8), e = 1 - p r^2:
        // error e^3, four dependent operations), scale, update.  A pivot that is not positive
        // and finite is recorded, not replaced: the factor is then undefined, as in LAPACK.
        double inv[CB], sq[CB];
#pragma unroll
        for (int k = 0; k < CB; ++k) {
            const double pv = d[k][k];
            const bool bad = !(pv > 0.0) || !(pv < INFINITY);
            firstbad = (bad && firstbad == 0x7fffffff) ? c0 + k + 1 : firstbad;
            double r = __builtin_amdgcn_rsq(pv);
            {
                const double e = fma(-pv * r, r, 1.0);
                r = fma(r * e, fma(0.375, e, 0.5), r);
            }
            inv[k] = r;
#pragma unroll
            for (int i = k + 1; i < CB; ++i) d[i][k] *= r;
#pragma unroll
            for (int j = k + 1; j < CB; ++j)
#pragma unroll
                for (int i = j; i < CB; ++i) d[i][j] = fma(-d[i][k], d[j][k], d[i][j]);
            double dd = pv * r;                                  // (off the chain) sqrt(p), one Newton step
            sq[k] = fma(0.5 * r, fma(-dd, dd, pv), dd);
            if constexpr (RHS) {
                double zacc = zb[k];                             // forward solve riding along
#pragma unroll
                for (int m = 0; m < k; ++m) zacc = fma(-zb[m], d[k][m], zacc);
                zb[k] = zacc * r;
            }
        }
        if (c0 == 0) { PANEL_FENCE(); PANEL_STAMP(7); }
        if (c0 == 52) { PANEL_FENCE(); PANEL_GHOOK(2); PANEL_FENCE(); }
        // (3) every lane solves its own row against it (rows of the block itself reproduce the
        // factor bit for bit: same operations in the same order)
        double x[CB];
        // (the lane comparisons from an opaque copy of the lane number: step- and launch-invariant, hipcc computes all 128
        // masks of the 16 groups once, keeps them in SGPRs spilled to VGPR lanes and restores each with two v_readlane +
        // s_nop -- three instructions where one v_cmp does)
        int ln = lane;
        asm volatile("" : "+v"(ln));
#pragma unroll
        for (int k = 0; k < CB; ++k) {
            double sacc = ar[c0 + k];
#pragma unroll
            for (int m = 0; m < k; ++m) sacc = fma(-x[m], d[k][m], sacc);
            sacc *= inv[k];
            x[k] = ln == c0 + k ? sq[k] : (ln > c0 + k ? sacc : 0.0);
            ar[c0 + k] = x[k];
        }
        if constexpr (RHS) {
            double racc = ri;
#pragma unroll
            for (int k = 0; k < CB; ++k) racc = fma(-x[k], zb[k], racc);
            ri = lane >= c0 + CB ? racc : ri;
#pragma unroll
            for (int k = 0; k < CB; ++k) ri = lane == c0 + k ? zb[k] : ri;
        }
        if (c0 == 0) { PANEL_FENCE(); PANEL_STAMP(8); }
        if (c0 == 52) { PANEL_FENCE(); PANEL_GHOOK(3); PANEL_FENCE(); }
        // (4) publish columns c0 .. c0 + CB - 1 of L_jj and their reciprocal pivots (one lane:
        // 64 lanes storing to one address serialise)
#pragma unroll
        for (int k = 0; k < CB; k += 2) *(f64x2*)(&Ls[lane][c0 + k]) = (f64x2){x[k], x[k + 1]};
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < CB; k += 2) *(f64x2*)(&invd[c0 + k]) = (f64x2){inv[k], inv[k + 1]};
            lds_store_volatile(&prog, c0 + CB);     // (same wavefront: LDS stores stay in order)
        }
        PANEL_FENCE();
        if (c0 == 0) { PANEL_STAMP(9); }
        if (c0 == 52) { PANEL_GHOOK(4); PANEL_FENCE(); }
        if (c0 == 16) PANEL_HOOK(3);
        if (c0 == 48) PANEL_HOOK(4);
        if (c0 == 60) PANEL_HOOK(5);
        if constexpr (c0 == HELPER_COL0) {
            PANEL_SPIN_WHILE_T(lds_load_volatile(hflag_p) < 2, trip_p);
            PANEL_FENCE();
#pragma unroll
            for (int k = HELPER_COL0 + CB; k < PB; k += 2) {
                const f64x2 q = *(const f64x2*)(&Ls[lane][k]);
                ar[k] = q.x;
                ar[k + 1] = q.y;
            }
            PANEL_FENCE();
        }
        // (5) rank-CB update of the columns to the right; row j of the new columns comes back
        // as broadcast reads, requested TR_AHEAD columns before their use
        panel_trailing<c0, c0 + CB, (c0 < HELPER_COL0 ? HELPER_COL0 : PB), 0, PB>(ar, x, Ls);
        if (c0 == 0) { PANEL_STAMP(10); }
        if (c0 == 52) { PANEL_FENCE(); PANEL_GHOOK(5); PANEL_FENCE(); }
    });
    PANEL_STAMP(2);
    if (firstbad != 0x7fffffff && wb_index == 0 && lane == 0)
        atomicMin((unsigned int*)a.info, (unsigned int)(a.info_j0 + j0 + firstbad));
    if constexpr (RHS) {
        if (a.rhs) {
            zblk[lane] = ri;
            if (wb_index == 0 && lane < bs) (a.dscr ? a.dscr + a.zoff : a.rhs)[j0 + lane] = ri;
        }
    }
    lds_store_volatile(&prog, PB + 1);
    {
        // L_jj out, coalesced, from the published copy (every workgroup holds it: workgroup number
        // wb_index of wb_count stores the rows r = wb_index mod wb_count) -- into the scratch blocks
        // when siblings exist (PotrfArgs::dscr), else straight into the matrix
        double* dst = a.dscr ? a.dscr + (j0 / PB) * (PB * PB) + lane : a.A + j0 * a.lda + j0 + lane;
        const long long ldd = a.dscr ? PB : a.lda;
        for (int r = wb_index; r < bs; r += wb_count)
            if (lane <= r) dst[(long long)r * ldd] = Ls[r][lane];
    }
    PANEL_STAMP(3);
#undef prog
}

__device__ __forceinline__ void panel_solve_wave(PotrfArgs& a, const long long j0, const long long row, const bool has_row,
                                                 const int lane, double (&x)[PB], const double (*Ls)[PB + 2],
                                                 const double* invd, const double* zblk, int* prog_p, const int wb_index) {
#define prog (*prog_p)
    auto wait_prog = [&](int need) {
        PANEL_SPIN_WHILE(lds_load_volatile(&prog) < need);
        PANEL_FENCE();
    };
    static_for<PB / CB>([&](auto cc_) {
        constexpr int c0 = CB * decltype(cc_)::value;
        wait_prog(c0 + CB);                 // columns c0 .. c0 + CB - 1 of L_jj published
        f64x2 dq[CB][CB / 2], iq[CB / 2];
#pragma unroll
        for (int r = 0; r < CB; ++r)
#pragma unroll
            for (int q = 0; 2 * q < r; ++q) dq[r][q] = *(const f64x2*)(&Ls[c0 + r][c0 + 2 * q]);
#pragma unroll
        for (int q = 0; q < CB / 2; ++q) iq[q] = *(const f64x2*)(&invd[c0 + 2 * q]);
        PANEL_FENCE();
        double xs[CB];
#pragma unroll
        for (int k = 0; k < CB; ++k) {
            double sacc = x[c0 + k];
#pragma unroll
            for (int m = 0; m < k; ++m) sacc = fma(-xs[m], (m & 1) ? dq[k][m >> 1].y : dq[k][m >> 1].x, sacc);
            xs[k] = sacc * ((k & 1) ? iq[k >> 1].y : iq[k >> 1].x);
            x[c0 + k] = xs[k];
        }
        PANEL_FENCE();
        panel_trailing<c0, c0 + CB, PB, 0, PB>(x, xs, Ls);
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
#undef prog
}

__global__ __launch_bounds__(192) void potrf_panel_kernel(PotrfArgs a) {
    potrf_select(a);
    __shared__ __attribute__((aligned(16))) double Ls[PB][PB + 2];
    __shared__ __attribute__((aligned(16))) double invd[PB];
    __shared__ double zblk[PB];
    __shared__ int prog;                 // columns published so far; PB + 1 once zblk is published too
    __shared__ int hflag;                // the helper wavefront's columns are back in Ls
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long j0 = a.j0;
    const int bs = (int)((a.n - j0) < PB ? (a.n - j0) : PB);
    const long long row = j0 + PB + (long long)blockIdx.x * PB + lane;
    const bool has_row = row < a.n;
    const int wb_index = blockIdx.x;
    if (threadIdx.x == 0) { prog = 0; hflag = 0; }

    if (wv == 0) {
        // ---------------- wavefront 0: the diagonal block ----------------
        PANEL_STAMP(0);
        double ar[PB];
        const double ri = (a.rhs && lane < bs) ? a.rhs[j0 + lane] : 0.0;
        if (bs == PB) {
            // coalesced: lane = column (one 512-byte row of the block per load, all 64 in flight
            // together), transposed to lane = row through Ls (free until the first columns are
            // published).  A row-per-lane load touches 64 cache lines per instruction.
            const double* src = a.A + j0 * a.lda + j0 + lane;
            double t[PB];
#pragma unroll
            for (int r = 0; r < PB; ++r) t[r] = src[(long long)r * a.lda];
#pragma unroll
            for (int r = 0; r < PB; ++r) Ls[r][lane] = t[r];
        } else {
            const double* src = a.A + (j0 + (lane < bs ? lane : 0)) * a.lda + j0;
#pragma unroll
            for (int k = 0; k < PB; ++k) Ls[lane][k] = (lane < bs && k <= lane) ? src[k] : 0.0;
        }
        __syncthreads();                                   // the block is staged (the helper reads its columns too)
#pragma unroll
        for (int k = 0; k < PB; k += 2) {
            const f64x2 v = *(const f64x2*)(&Ls[lane][k]);
            ar[k] = (lane < bs && k <= lane) ? v.x : ((k == lane) ? 1.0 : 0.0);
            ar[k + 1] = (lane < bs && k + 1 <= lane) ? v.y : ((k + 1 == lane) ? 1.0 : 0.0);
        }
        panel_factor_wave(a, j0, bs, lane, ar, ri, Ls, invd, zblk, &prog, &hflag, wb_index, (int)gridDim.x);
        return;
    }
    if (wv == 2) {
        __syncthreads();
        panel_helper_wave(bs, lane, Ls, &prog, &hflag);
        return;
    }

    // ---------------- wavefront 1: the 64 panel rows of this workgroup ----------------
    double x[PB];
    {
        const double* src = a.A + (has_row ? row : j0) * a.lda + j0;
#pragma unroll
        for (int k = 0; k < PB; ++k) x[k] = (bs == PB && has_row) ? src[k] : 0.0;
    }
    __syncthreads();
    PANEL_FENCE();
    panel_solve_wave(a, j0, row, has_row, lane, x, Ls, invd, zblk, &prog, wb_index);
}

// ---------------------------------------------------------------------------
// One gpUtils._nll evaluation (gpUtils.py:46-80) of a training set of n <= 64 points as ONE
// single-workgroup launch: Gram block -> LDS (never through memory), the register-resident
// factorisation of the panel step with the forward solve riding along (panel_factor_wave +
// panel_helper_wave, exactly the code and operation order of the blocked path), and the
// 5-value fit summary in potrf_finish_kernel's reduction order -- so the values are the bits
// the three-launch path (gram + panel + finish) produces, without its two launch boundaries
// (~7 us each) and with K never leaving the CU.  This is where the README
// configuration (C1: N = 50 .. 90, 722-3,122 evaluations per optimizeGP) lives.
// ---------------------------------------------------------------------------
struct NllSmallArgs {
    const double* X;
    const double* y;
    double* K;           // n x n (ld n): the factor on return (lower triangle)
    double* z;           // n: L^-1 (y - shift)
    int* info;
    double* out5;
    long long n;
    double shift;
    KernConst kc;
    double* mail;        // optional: pinned, device-mapped host record (5 doubles + sequence word)
    long long seq;
    // batched launch (apgp_nll_eval_batch at n <= 128: gridDim.x matrices of the SAME training set, one workgroup each):
    // matrix b takes its kernel constants from bkc[b] and its shift from bshift[b] (device-mapped host memory, read once
    // into LDS), works in K + b n^2, z + b n, info + b, out5 + 5 b and posts its record + sequence word at brec + 8 b
    const KernConst* bkc;
    const double* bshift;
    double* brec;
};
// this workgroup's matrix of a batched launch: the constants into LDS, the pointers moved on
template <bool BATCH>
__device__ __forceinline__ void nll_batch_select(NllSmallArgs& q, KernConst* kcb) {
    if constexpr (BATCH) {
        const int b = blockIdx.x;
        for (int e = threadIdx.x; e < (int)(sizeof(KernConst) / sizeof(double)); e += blockDim.x)
            ((double*)kcb)[e] = ((const double*)(q.bkc + b))[e];
        q.shift = q.bshift[b];
        q.K += (long long)b * q.n * q.n;
        q.z += (long long)b * q.n;
        q.info += b;
        q.out5 += 5 * b;
        q.mail = q.brec + 8 * b;
        __syncthreads();
    }
}
static_assert(sizeof(KernConst) % sizeof(double) == 0, "KernConst is copied as doubles");

template <int DPAD, bool BATCH>
__global__ __launch_bounds__(192) void nll_small_kernel(NllSmallArgs q) {
    __shared__ KernConst kcb;
    nll_batch_select<BATCH>(q, &kcb);
    const KernConst& kc = BATCH ? kcb : q.kc;
    __shared__ __attribute__((aligned(16))) double Ls[PB][PB + 2];
    __shared__ __attribute__((aligned(16))) double invd[PB];
    __shared__ double zblk[PB];
    __shared__ int prog, hflag;
    __shared__ double xs[PB][DPAD + 1];
    __shared__ double etab[APGP_EXP_TAB_N];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int bs = (int)q.n;
    apgp_exp_tab_load(etab);
    if (t == 0) { prog = 0; hflag = 0; *(unsigned int*)q.info = 0xffffffffu; }
    for (int e = t; e < PB * DPAD; e += 192) {
        const int r = e / DPAD, d = e % DPAD;
        xs[r][d] = (r < bs && d < kc.ndim) ? q.X[(long long)r * kc.ndim + d] * kc.sc[d] : 0.0;
    }
    __syncthreads();
    // Gram block, lower triangle (zero above it and past n): thread = (column c, rows r = g, g + 3, ...)
    {
        const int c = lane, g = wv;
        double xc[DPAD];
#pragma unroll
        for (int d = 0; d < DPAD; ++d) xc[d] = xs[c][d];
        for (int r = g; r < PB; r += 3) {
            double k = 0.0;
            if (r < bs && c <= r) k = apgp_gram_value<DPAD>(xs[r], xc, kc, r == c, etab);
            Ls[r][c] = k;
        }
    }
    __syncthreads();
    if (wv == 1) {
        // (otherwise idle: the strict upper triangle of the factor in memory is written as zeros)
        for (int r = 0; r < bs; ++r)
            if (lane > r && lane < bs) q.K[(long long)r * q.n + lane] = 0.0;
        return;
    }
    if (wv == 2) {
        panel_helper_wave(bs, lane, Ls, &prog, &hflag);
        return;
    }
    PotrfArgs a;
    a.A = q.K; a.rhs = q.z; a.n = q.n; a.lda = q.n; a.j0 = 0; a.shift = 0.0; a.info = q.info; a.out5 = nullptr;
    a.mail = nullptr; a.seq = 0;
    a.dscr = nullptr; a.batch_dscr = 0; a.zoff = 0; a.batch_A = 0; a.batch_rhs = 0;
    a.abort_word = nullptr; a.abort_id = 0; a.no_panel = 0; a.info_j0 = 0; a.pair_mode = 0; a.defer8 = 0; a.deferred8 = 0;
    double ar[PB];
    const double ri = lane < bs ? q.y[lane] - q.shift : 0.0;
#pragma unroll
    for (int k = 0; k < PB; k += 2) {
        const f64x2 v = *(const f64x2*)(&Ls[lane][k]);
        ar[k] = (lane < bs && k <= lane) ? v.x : ((k == lane) ? 1.0 : 0.0);
        ar[k + 1] = (lane < bs && k + 1 <= lane) ? v.y : ((k + 1 == lane) ? 1.0 : 0.0);
    }
    panel_factor_wave(a, 0, bs, lane, ar, ri, Ls, invd, zblk, &prog, &hflag, 0, 1);
    // fit summary (potrf_finish_kernel's operations in its order: one element per lane, the
    // butterfly over the wavefront; its sum over sixteen wavefront partials adds zeros here)
    double sl = 0.0, mn = INFINITY, mx = -INFINITY, zz = 0.0;
    if (lane < bs) {
        const double d = Ls[lane][lane];
        sl += log(d);
        mn = fmin(mn, d);
        mx = fmax(mx, d);
        zz = fma(zblk[lane], zblk[lane], zz);
    }
    for (int o = 32; o > 0; o >>= 1) {
        sl += __shfl_xor(sl, o);
        zz += __shfl_xor(zz, o);
        mn = fmin(mn, __shfl_xor(mn, o));
        mx = fmax(mx, __shfl_xor(mx, o));
    }
    if (lane == 0) {
        int inf = *q.info;                      // (this lane's own atomicMin, if any, precedes the read)
        if ((unsigned int)inf == 0xffffffffu) inf = 0;
        *q.info = inf;
        q.out5[0] = 2.0 * sl;
        q.out5[1] = mn;
        q.out5[2] = mx;
        q.out5[3] = zz;
        q.out5[4] = (double)inf;
        if (q.mail) {
            // the record straight into host memory, then the sequence word with system-scope release: the
            // host polls it instead of a D2H copy + stream synchronisation (~10 us of a 40 us evaluation)
            q.mail[0] = 2.0 * sl;
            q.mail[1] = mn;
            q.mail[2] = mx;
            q.mail[3] = zz;
            q.mail[4] = (double)inf;
            __hip_atomic_store((long long*)(q.mail + 5), q.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// ---------------------------------------------------------------------------
// The same for 64 < n <= 128 (round 5; the README configuration grows from N = 50 to 90): TWO block columns in ONE
// single-workgroup launch -- the Gram tiles (0,0) and (1,0) into LDS, block column 0 by the panel step's three
// roles (panel_factor_wave + panel_helper_wave, panel_solve_wave for rows 64 ..: the code of potrf_panel_kernel), the
// update of block (1,1) by the product of the launch-per-step path (apgp_gemm64_tile on the rows just written, the
// block's own Gram values generated in the accumulators' layout), block column 1 by the panel step again, and
// potrf_finish_kernel's summary in its order (two "virtual wavefronts" of 64 diagonal entries).  Every value is computed
// by the code and in the order of gram + panel + step + finish: the same bits, without three launch boundaries
// (51 -> ~35 us at N = 65 .. 128).  The factor goes straight into K (no sibling workgroups: no scratch blocks), z into q.z.
// ---------------------------------------------------------------------------
template <int DPAD>
static constexpr size_t nll_two_lds_doubles() {
    return (size_t)PB * (PB + 2) + GEMM64_LDS_DOUBLES + (size_t)2 * PB * (DPAD + 1) + 5 * PB + APGP_EXP_TAB_N + 2;
}
template <int DPAD, bool BATCH>
__global__ __launch_bounds__(256) void nll_two_kernel(NllSmallArgs q) {
    __shared__ KernConst kcb;
    nll_batch_select<BATCH>(q, &kcb);
    const KernConst& kc = BATCH ? kcb : q.kc;
    extern __shared__ __attribute__((aligned(16))) double two_lds[];
    double (*Ls)[PB + 2] = (double (*)[PB + 2])two_lds;                          // the block being factorised
    double* gl = two_lds + PB * (PB + 2);                                         // Gram tile (1,0), later the product's buffers
    double (*T10)[PB + 2] = (double (*)[PB + 2])gl;
    double (*xs)[DPAD + 1] = (double (*)[DPAD + 1])(gl + GEMM64_LDS_DOUBLES);      // scaled coordinates of all n points
    double* invd = (double*)(xs + 2 * PB);
    double* zblk = invd + PB;
    double* dsave = zblk + PB;                                                    // diagonal / z of block column 0 (for the summary)
    double* zsave = dsave + PB;
    double* rhs1 = zsave + PB;                                                    // running right-hand side of rows 64 ..
    double* etab = rhs1 + PB;
    int* prog_p = (int*)(etab + APGP_EXP_TAB_N);
    int* hflag_p = prog_p + 1;
    static_assert(GEMM64_LDS_DOUBLES >= PB * (PB + 2), "the Gram tile fits the product's buffers");
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int n = (int)q.n, bs1 = n - PB;                                         // rows of block column 1: 1 .. 64
    apgp_exp_tab_load(etab);
    if (t == 0) { *prog_p = 0; *hflag_p = 0; *(unsigned int*)q.info = 0xffffffffu; }
    for (int e = t; e < 2 * PB * DPAD; e += 256) {
        const int r = e / DPAD, d = e % DPAD;
        xs[r][d] = (r < n && d < kc.ndim) ? q.X[(long long)r * kc.ndim + d] * kc.sc[d] : 0.0;
    }
    __syncthreads();
    // Gram tiles (0,0) (lower triangle) and (1,0) (rows past n: zero): thread = (column c, rows g, g + 4, ...)
    {
        const int c = lane, g = wv;
        double xc[DPAD];
#pragma unroll
        for (int d = 0; d < DPAD; ++d) xc[d] = xs[c][d];
        for (int r = g; r < PB; r += 4) {
            Ls[r][c] = c <= r ? apgp_gram_value<DPAD>(xs[r], xc, kc, r == c, etab) : 0.0;
            T10[r][c] = r < bs1 ? apgp_gram_value<DPAD>(xs[PB + r], xc, kc, false, etab) : 0.0;
        }
    }
    __syncthreads();
    PotrfArgs a;
    a.A = q.K; a.rhs = q.z; a.n = q.n; a.lda = q.n; a.j0 = 0; a.shift = 0.0; a.info = q.info; a.out5 = nullptr;
    a.mail = nullptr; a.seq = 0;
    a.dscr = nullptr; a.batch_dscr = 0; a.zoff = 0; a.batch_A = 0; a.batch_rhs = 0;
    a.abort_word = nullptr; a.abort_id = 0; a.no_panel = 0; a.info_j0 = 0; a.pair_mode = 0; a.defer8 = 0; a.deferred8 = 0;
    // ---------------- block column 0: potrf_panel_kernel's roles ----------------
    if (wv == 0) {
        double ar[PB];
        const double ri = q.y[lane] - q.shift;
#pragma unroll
        for (int k = 0; k < PB; k += 2) {
            const f64x2 v = *(const f64x2*)(&Ls[lane][k]);
            ar[k] = k <= lane ? v.x : ((k == lane) ? 1.0 : 0.0);
            ar[k + 1] = k + 1 <= lane ? v.y : ((k + 1 == lane) ? 1.0 : 0.0);
        }
        panel_factor_wave(a, 0, PB, lane, ar, ri, Ls, invd, zblk, prog_p, hflag_p, 0, 1);
        dsave[lane] = Ls[lane][lane];
        zsave[lane] = zblk[lane];
    } else if (wv == 2) {
        panel_helper_wave(PB, lane, Ls, prog_p, hflag_p);
    } else if (wv == 1) {
        const long long row = PB + lane;
        const bool has_row = lane < bs1;
        double x[PB];
#pragma unroll
        for (int k = 0; k < PB; k += 2) {
            const f64x2 v = *(const f64x2*)(&T10[lane][k]);
            x[k] = has_row ? v.x : 0.0;
            x[k + 1] = has_row ? v.y : 0.0;
        }
        if (has_row) q.z[row] = q.y[row] - q.shift;                              // (the running right-hand side of this row)
        PANEL_FENCE();
        panel_solve_wave(a, 0, row, has_row, lane, x, Ls, invd, zblk, prog_p, 0);
        rhs1[lane] = has_row ? q.z[row] : 0.0;                                   // (this lane's own store)
    }
    __threadfence();                          // (L(1,0) is read back from memory by all four wavefronts)
    __syncthreads();
    // ---------------- block (1,1) -= L(1,0) L(1,0)^T: potrf_step_kernel's first tile (bi = 0) ----------------
    const int wr = (wv >> 1) * 32, wc = (wv & 1) * 32;
    double v[2][2][4];
    {
        double cin[2][2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                double xc[DPAD];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int lr = wr + 16 * i + apgp_mma16_row(lane), lc = wc + 16 * j + apgp_mma16_col(lane, r);
#pragma unroll
                    for (int d = 0; d < DPAD; ++d) xc[d] = xs[PB + lc][d];
                    cin[i][j][r] = (lr < bs1 && lc < bs1 && lc <= lr) ? apgp_gram_value<DPAD>(xs[PB + lr], xc, kc, lr == lc, etab) : 0.0;
                    v[i][j][r] = 0.0;
                }
            }
        apgp_gemm64_tile<false, false>(q.K + (long long)PB * q.n, q.n, bs1, q.K + (long long)PB * q.n, q.n, bs1, 0, PB, gl, v);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[i][j][r] = cin[i][j][r] - v[i][j][r];
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                Ls[wr + 16 * i + apgp_mma16_row(lane)][wc + 16 * j + apgp_mma16_col(lane, r)] = v[i][j][r];
    if (t == 0) { *prog_p = 0; *hflag_p = 0; }
    __syncthreads();
    // ---------------- block column 1: the panel step of potrf_step_kernel ----------------
    if (wv == 2) {
        panel_helper_wave(bs1, lane, Ls, prog_p, hflag_p);
        return;
    }
    if (wv != 0) return;
    {
        double rowv[PB];
        const double rhs_i = lane < bs1 ? rhs1[lane] : 0.0;
#pragma unroll
        for (int k = 0; k < PB; k += 2) {
            const f64x2 qv = *(const f64x2*)(&Ls[lane][k]);
            rowv[k] = (lane < bs1 && k <= lane) ? qv.x : ((k == lane) ? 1.0 : 0.0);
            rowv[k + 1] = (lane < bs1 && k + 1 <= lane) ? qv.y : ((k + 1 == lane) ? 1.0 : 0.0);
        }
        panel_factor_wave(a, PB, bs1, lane, rowv, rhs_i, Ls, invd, zblk, prog_p, hflag_p, 0, 1);
    }
    // fit summary: potrf_finish_kernel's operations in its order -- virtual wavefront 0 = entries 0 .. 63, 1 = 64 .. n - 1,
    // each through the butterfly, the sixteen partials summed in order (fourteen of them zeros / infinities)
    double p_sl[2], p_zz[2], p_mn[2], p_mx[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        double sl = 0.0, mn = INFINITY, mx = -INFINITY, zz = 0.0;
        if (h == 0 || lane < bs1) {
            const double d = h == 0 ? dsave[lane] : Ls[lane][lane];
            const double zv = h == 0 ? zsave[lane] : zblk[lane];
            sl += log(d);
            mn = fmin(mn, d);
            mx = fmax(mx, d);
            zz = fma(zv, zv, zz);
        }
        for (int o = 32; o > 0; o >>= 1) {
            sl += __shfl_xor(sl, o);
            zz += __shfl_xor(zz, o);
            mn = fmin(mn, __shfl_xor(mn, o));
            mx = fmax(mx, __shfl_xor(mx, o));
        }
        p_sl[h] = sl; p_zz[h] = zz; p_mn[h] = mn; p_mx[h] = mx;
    }
    if (lane == 0) {
        double sl = 0.0, zz = 0.0, mn = INFINITY, mx = -INFINITY;
#pragma unroll
        for (int h = 0; h < 2; ++h) { sl += p_sl[h]; zz += p_zz[h]; mn = fmin(mn, p_mn[h]); mx = fmax(mx, p_mx[h]); }
        int inf = *q.info;                      // (this wavefront's own atomicMin, if any, precedes the read)
        if ((unsigned int)inf == 0xffffffffu) inf = 0;
        *q.info = inf;
        q.out5[0] = 2.0 * sl;
        q.out5[1] = mn;
        q.out5[2] = mx;
        q.out5[3] = zz;
        q.out5[4] = (double)inf;
        if (q.mail) {
            q.mail[0] = 2.0 * sl;
            q.mail[1] = mn;
            q.mail[2] = mx;
            q.mail[3] = zz;
            q.mail[4] = (double)inf;
            __hip_atomic_store((long long*)(q.mail + 5), q.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// the kernel's dynamic LDS exceeds 64 KiB: the attribute is set once per device and instantiation
template <int DPAD, bool BATCH>
static int nll_two_launch_t(const NllSmallArgs& q, int batch, hipStream_t s) {
    static bool done[64] = {false};
    static std::mutex mu;
    const int lds = (int)(nll_two_lds_doubles<DPAD>() * sizeof(double));
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
        apgp_set_error("apgp_nll_eval: hipGetDevice failed");
        return -2;
    }
    {
        std::lock_guard<std::mutex> lock(mu);
        if (!done[dev]) {
            const hipError_t e = hipFuncSetAttribute((const void*)nll_two_kernel<DPAD, BATCH>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            if (e != hipSuccess) {
                apgp_set_error("apgp_nll_eval: hipFuncSetAttribute(%d B of LDS) failed on device %d: %s", lds, dev, hipGetErrorString(e));
                return -2;
            }
            done[dev] = true;
        }
    }
    hipLaunchKernelGGL((nll_two_kernel<DPAD, BATCH>), dim3((unsigned)batch), dim3(256), lds, s, q);
    APGP_CHECK_LAUNCH();
    return 0;
}
// one launch of the fused evaluation: n <= 64 (nll_small_kernel) or 64 < n <= 128 (nll_two_kernel); q.bkc != NULL:
// `batch` matrices, one workgroup each
template <bool BATCH>
static int nll_fused_launch_q(const NllSmallArgs& q, int dpad, int batch, hipStream_t s) {
    if (q.n > PB) {
        switch (dpad) {
            case 2: return nll_two_launch_t<2, BATCH>(q, batch, s);
            case 4: return nll_two_launch_t<4, BATCH>(q, batch, s);
            case 8: return nll_two_launch_t<8, BATCH>(q, batch, s);
            case 16: return nll_two_launch_t<16, BATCH>(q, batch, s);
            default: return nll_two_launch_t<32, BATCH>(q, batch, s);
        }
    }
    const dim3 grid((unsigned)batch), block(192);
    switch (dpad) {
        case 2: hipLaunchKernelGGL((nll_small_kernel<2, BATCH>), grid, block, 0, s, q); break;
        case 4: hipLaunchKernelGGL((nll_small_kernel<4, BATCH>), grid, block, 0, s, q); break;
        case 8: hipLaunchKernelGGL((nll_small_kernel<8, BATCH>), grid, block, 0, s, q); break;
        case 16: hipLaunchKernelGGL((nll_small_kernel<16, BATCH>), grid, block, 0, s, q); break;
        default: hipLaunchKernelGGL((nll_small_kernel<32, BATCH>), grid, block, 0, s, q); break;
    }
    APGP_CHECK_LAUNCH();
    return 0;
}
static int nll_fused_launch(const double* X, int64_t n, const apgp_kernel_t* kern, const double* y, double mean,
                            double* K, double* z, int32_t* info_dev, double* out5_dev, hipStream_t s,
                            double* mail = nullptr, long long seq = 0) {
    NllSmallArgs q;
    q.mail = mail; q.seq = seq; q.bkc = nullptr; q.bshift = nullptr; q.brec = nullptr;
    if (apgp_make_kernconst(kern, &q.kc) != 0) {
        apgp_set_error("apgp_nll_eval: bad argument: kernel parameters");
        return -1;
    }
    q.X = X; q.y = y; q.K = K; q.z = z; q.info = info_dev; q.out5 = out5_dev; q.n = n; q.shift = mean;
    return nll_fused_launch_q<false>(q, q.kc.dpad, 1, s);
}

// One launch per block step after the first: the trailing update of block column j AND the panel
// step of block column j + 1 (look-ahead).  Workgroups 0 .. tb-1 own the first trailing tile
// column -- the next diagonal block (workgroup 0) and the next panel rows (workgroups 1 ..) -- and
// carry on with the panel step as soon as their tile is updated, while the other workgroups are
// still updating the rest of the trailing matrix: a step costs max(update, first tile + panel)
// instead of their sum.  There is no dependency between workgroups inside the launch: a panel
// workgroup updates the next diagonal block ITSELF (a second 64x64x64 product, redundantly, as it
// re-factorises it redundantly anyway), so nothing waits on a flag or a memory round trip.  The
// updated tiles never go through memory either: accumulators -> LDS -> one row per lane.
__global__ __launch_bounds__(256, 2) void potrf_step_kernel(PotrfArgs a) {
    potrf_select(a);
    __shared__ __attribute__((aligned(16))) double lds[GEMM64_LDS_DOUBLES];   // GEMM buffers / tile staging / Ls of the panel step
    __shared__ __attribute__((aligned(16))) double invd[PB];
    __shared__ double zblk[PB];
    __shared__ int prog, hflag;
    static_assert(GEMM64_LDS_DOUBLES >= PB * (PB + 2), "Ls aliases the GEMM buffers");
    const long long base = a.j0 + PB;                     // first row / column of the trailing matrix
    const long long tb = (a.n - base + PB - 1) / PB;      // its size in blocks
    const bool first = (long long)blockIdx.x < tb;
    long long bi, bk;
    bool deferred = false;                                // a tile the previous (wide) step left to this launch
    if (first) { bi = blockIdx.x; bk = 0; }
    else if (a.pair_mode == 1) {
        // narrow step: the second tile column only ...
        const long long k = (long long)blockIdx.x - tb;
        if (k < tb - 1) { bi = k + 1; bk = 1; }
        else {
            // ... and the previous wide step's deferred tiles: that step's trailing matrix starts at j0 (one block earlier
            // than this one's), its eligible triangle at block (3, 3)
            deferred = true;
            long long r, c;
            potrf_tri(potrf_deferred_index(k - (tb - 1), a.deferred8), r, c);
            bi = r + 3 - 1; bk = c + 3 - 1;               // (in THIS step's block numbering: one less)
        }
    }
    else if (a.defer8 > 0) {
        // wide step that defers: tile columns 1 and 2 in full, then the kept tiles of the triangle from block (3, 3) on
        const long long k = (long long)blockIdx.x - tb;
        if (k < tb - 1) { bi = k + 1; bk = 1; }
        else if (k < 2 * tb - 3) { bi = k - (tb - 1) + 2; bk = 2; }
        else {
            long long r, c;
            potrf_tri(potrf_kept_index(k - (2 * tb - 3), a.defer8), r, c);
            bi = r + 3; bk = c + 3;
        }
    }
    else {
        // the other tiles: lower triangle of the (tb - 1) x (tb - 1) blocks below / right of tile (0, 0)
        const long long tix = (long long)blockIdx.x - tb;
        long long b2 = (long long)((sqrt(8.0 * (double)tix + 1.0) - 1.0) * 0.5);
        while ((b2 + 1) * (b2 + 2) / 2 <= tix) ++b2;
        while (b2 * (b2 + 1) / 2 > tix) --b2;
        bi = b2 + 1;
        bk = tix - b2 * (b2 + 1) / 2 + 1;
    }
    const long long ri = base + bi * PB, rk = base + bk * PB;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int wr = (w >> 1) * 32, wc = (w & 1) * 32;
    STEP_STAMP(0);
    const int bs = (int)((a.n - base) < PB ? (a.n - base) : PB);
    const double rhs_i = (first && w == 0 && a.rhs && lane < bs) ? a.rhs[base + lane] : 0.0;   // (needed late: requested now)
    if (!first) {
        // plain trailing tile; its own elements are requested first: they are needed last
        double cin[2][2][4], v[2][2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const long long gr = ri + wr + 16 * i + apgp_mma16_row(lane);
                    const long long gc = rk + wc + 16 * j + apgp_mma16_col(lane, r);
                    cin[i][j][r] = (gr < a.n && gc < a.n && gc <= gr) ? a.A[gr * a.lda + gc] : 0.0;
                    v[i][j][r] = 0.0;
                }
        // (a deferred tile: the wide step's two block columns are the two BEFORE this step's)
        const long long jc = deferred ? a.j0 - PB : a.j0;
        if (a.pair_mode == 2 || deferred) {
            // wide step: the previous block column first (its narrow step left these tiles alone), then this one
            apgp_gemm64_tile<false, false>(a.A + ri * a.lda + jc - PB, a.lda, a.n - ri, a.A + rk * a.lda + jc - PB, a.lda, a.n - rk,
                                           0, PB, lds, v);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        cin[i][j][r] = cin[i][j][r] - v[i][j][r];
                        v[i][j][r] = 0.0;
                    }
        }
        apgp_gemm64_tile<false, false>(a.A + ri * a.lda + jc, a.lda, a.n - ri, a.A + rk * a.lda + jc, a.lda, a.n - rk,
                                       0, PB, lds, v);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const long long gr = ri + wr + 16 * i + apgp_mma16_row(lane);
                    const long long gc = rk + wc + 16 * j + apgp_mma16_col(lane, r);
                    if (gr < a.n && gc < a.n && gc <= gr) a.A[gr * a.lda + gc] = cin[i][j][r] - v[i][j][r];
                }
        return;
    }
    // first tile column.  A panel workgroup (bi > 0) also needs tile (0, 0), the next diagonal
    // block: B B^T of the same B rows, riding along in the same product.
    const bool two = bi != 0;
    double v[2][2][4], v2[2][2][4];
    {
        double cin[2][2][4], cin2[2][2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int lr = wr + 16 * i + apgp_mma16_row(lane), lc = wc + 16 * j + apgp_mma16_col(lane, r);
                    const long long gr = ri + lr, gc = rk + lc;
                    cin[i][j][r] = (gr < a.n && gc < a.n && gc <= gr) ? a.A[gr * a.lda + gc] : 0.0;
                    cin2[i][j][r] = (two && lc <= lr) ? a.A[(base + lr) * a.lda + base + lc] : 0.0;   // (two: the block is full)
                    v[i][j][r] = 0.0;
                    v2[i][j][r] = 0.0;
                }
        apgp_gemm64_tile2<false, false, true>(a.A + ri * a.lda + a.j0, a.lda, a.n - ri, a.A + rk * a.lda + a.j0, a.lda,
                                              a.n - rk, 0, PB, lds, v, v2);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[i][j][r] = cin[i][j][r] - v[i][j][r];
                    v2[i][j][r] = cin2[i][j][r] - v2[i][j][r];
                }
    }
    if (a.no_panel) {
        // hand-over to the persistent launch (hybrid): the updated first tile column goes to memory like any other tile
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const long long gr = ri + wr + 16 * i + apgp_mma16_row(lane);
                    const long long gc = rk + wc + 16 * j + apgp_mma16_col(lane, r);
                    if (gr < a.n && gc < a.n && gc <= gr) a.A[gr * a.lda + gc] = v[i][j][r];
                }
        return;
    }
    // ---------------- first tile column: the panel step of block column `base` ----------------
    STEP_STAMP(1);
    double (*Ls)[PB + 2] = (double (*)[PB + 2])lds;       // (the GEMM ended with a barrier: its buffers are free)
    const long long row = ri + lane;
    const bool has_row = row < a.n;
    double rowv[PB];        // panel row (solving wavefront) / diagonal-block row (factorising wavefront) of this lane
    if (two) {
        // own tile = 64 panel rows: through LDS into the solving wavefront's registers
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    Ls[wr + 16 * i + apgp_mma16_row(lane)][wc + 16 * j + apgp_mma16_col(lane, r)] = v[i][j][r];
        __syncthreads();
        if (w == 1) {
#pragma unroll
            for (int k = 0; k < PB; k += 2) {
                const f64x2 q = *(const f64x2*)(&Ls[lane][k]);
                rowv[k] = has_row ? q.x : 0.0;
                rowv[k + 1] = has_row ? q.y : 0.0;
            }
        }
        __syncthreads();
    }
    // the updated diagonal block, same way, for the factorising wavefront (two loops, not one
    // with a select between the arrays: a select of element ADDRESSES keeps both arrays in memory)
    if (two) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    Ls[wr + 16 * i + apgp_mma16_row(lane)][wc + 16 * j + apgp_mma16_col(lane, r)] = v2[i][j][r];
    } else {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    Ls[wr + 16 * i + apgp_mma16_row(lane)][wc + 16 * j + apgp_mma16_col(lane, r)] = v[i][j][r];
    }
    if (t == 0) { prog = 0; hflag = 0; }
    __syncthreads();
    STEP_STAMP(2);
    if (w == 3 || (w == 1 && bi == 0)) return;           // (the diagonal tile has no panel rows)
    if (w == 2) {
        panel_helper_wave(bs, lane, Ls, &prog, &hflag);
        return;
    }
    if (w == 0) {
#pragma unroll
        for (int k = 0; k < PB; k += 2) {
            const f64x2 q = *(const f64x2*)(&Ls[lane][k]);
            rowv[k] = (lane < bs && k <= lane) ? q.x : ((k == lane) ? 1.0 : 0.0);
            rowv[k + 1] = (lane < bs && k + 1 <= lane) ? q.y : ((k + 1 == lane) ? 1.0 : 0.0);
        }
        STEP_STAMP(3);
        panel_factor_wave(a, base, bs, lane, rowv, rhs_i, Ls, invd, zblk, &prog, &hflag, (int)bi, (int)tb);
        STEP_STAMP(4);
    } else {
        PANEL_FENCE();
        panel_solve_wave(a, base, row, has_row, lane, rowv, Ls, invd, zblk, &prog, (int)bi);
        STEP_STAMP(5);
    }
}

__global__ __launch_bounds__(256) void potrf_rhs_init_kernel(const double* y, double shift, double* rhs, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) rhs[i] = y[i] - shift;
}

// info = UINT_MAX ("no failure yet") -> 0; factored diagonal blocks from the scratch into the matrix;
// with a.out5 the fit summary of apgp_fit_summary (2 sum log L_ii | min L_ii | max L_ii | z.z | info)
// as well, by workgroup 0, with the operations of linalg.hip's logdet_kernel in the same order: its
// 1024 threads are four "virtual" threads per thread here (v = t + 256 q belongs to wavefront
// v / 64 = t / 64 + 4 q, lane t % 64), so every butterfly and the final sum over 16 wavefront
// partials see the same operands.
__global__ __launch_bounds__(256) void potrf_finish_kernel(PotrfArgs a) {
    potrf_select(a);
    __shared__ double ssum[16], smin[16], smax[16], szz[16];
    __shared__ int sinfo;
    const bool aborted = a.abort_word && *a.abort_word == a.abort_id;     // (persistent launch gave up: potrf_persist.h)
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (*(unsigned int*)a.info == 0xffffffffu) *a.info = 0;
        sinfo = *a.info;
    }
    if (blockIdx.x == 0 && a.out5) {
        const int t = threadIdx.x;
        const double* zsrc = a.rhs ? (a.dscr ? a.dscr + a.zoff : a.rhs) : nullptr;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            double sl = 0.0, mn = INFINITY, mx = -INFINITY, zz = 0.0;
            for (long long i = t + 256 * q; i < a.n; i += 1024) {
                const double d = a.dscr ? a.dscr[(i / PB) * (PB * PB) + (i % PB) * (PB + 1)] : a.A[i * a.lda + i];
                sl += log(d);
                mn = fmin(mn, d);
                mx = fmax(mx, d);
                if (zsrc) zz = fma(zsrc[i], zsrc[i], zz);
            }
            for (int o = 32; o > 0; o >>= 1) {
                sl += __shfl_xor(sl, o);
                zz += __shfl_xor(zz, o);
                mn = fmin(mn, __shfl_xor(mn, o));
                mx = fmax(mx, __shfl_xor(mx, o));
            }
            const int w = (t >> 6) + 4 * q;
            if ((t & 63) == 0) { ssum[w] = sl; smin[w] = mn; smax[w] = mx; szz[w] = zz; }
        }
        __syncthreads();
        if (t == 0) {
            double sl = 0.0, zz = 0.0, mn = INFINITY, mx = -INFINITY;
            for (int i = 0; i < 16; ++i) { sl += ssum[i]; zz += szz[i]; mn = fmin(mn, smin[i]); mx = fmax(mx, smax[i]); }
            a.out5[0] = 2.0 * sl;
            a.out5[1] = mn;
            a.out5[2] = mx;
            a.out5[3] = zz;
            a.out5[4] = aborted ? -7777.0 /* PP_ABORTED */ : (double)sinfo;
            if (a.mail && (blockIdx.y == 0 || a.batch_mail != 0)) {
                a.mail[0] = 2.0 * sl;
                a.mail[1] = mn;
                a.mail[2] = mx;
                a.mail[3] = zz;
                a.mail[4] = aborted ? -7777.0 : (double)sinfo;
                __hip_atomic_store((long long*)(a.mail + 5), a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
    if (!a.dscr) return;
    const long long j0 = (long long)blockIdx.x * PB;
    const int bs = (int)((a.n - j0) < PB ? (a.n - j0) : PB);
    const double* src = a.dscr + (long long)blockIdx.x * (PB * PB);
    for (int e = threadIdx.x; e < PB * PB; e += 256) {
        const int r = e >> 6, c = e & 63;
        if (r < bs && c <= r) a.A[(j0 + r) * a.lda + j0 + c] = src[e];
    }
    if (a.rhs && threadIdx.x < bs) a.rhs[j0 + threadIdx.x] = a.dscr[a.zoff + j0 + threadIdx.x];
}


#include "potrf_persist.h"

// ---------------------------------------------------------------------------
// host side of the persistent factorisation
// ---------------------------------------------------------------------------
// 0 = persistent where it applies (default) | 1 = multi-launch path only | 2 = persistent, workgroup 0 gives up at
// once (exercises the fallback) | 3 = persistent wherever it can run (n <= 4096).  A test / profiling switch
// (apgp_potrf_mode), not read from the environment.
static std::atomic<int> g_potrf_mode{0};
// paired trailing updates of the launch-per-step path (PotrfArgs::pair_mode) from this many trailing block rows on;
// apgp_potrf_mode(mode | 16) switches them off (A/B and bit-identity tests), plain modes switch them on again
#define POTRF_PAIR_MIN_TB 24
static std::atomic<int> g_potrf_pairs{1};
// eighths of a wide step's eligible tiles left to the narrow step after it (PotrfArgs::defer8); apgp_potrf_mode(mode | 32)
// switches the deferral off
#ifndef POTRF_DEFER8
#define POTRF_DEFER8 4
#endif
static std::atomic<int> g_potrf_defer8{POTRF_DEFER8};
static std::atomic<long long> g_potrf_fallbacks{0};
// Back-off after a persistent launch gave up (its workgroups were not all resident within the timeout: another stream
// or process holds CUs): every such call costs the timeout AND the re-run, so the next PP_BACKOFF_BASE << (streak - 1)
// evaluations on that device (at most PP_BACKOFF_MAX) take the launch-per-step path straight away; a persistent launch
// that completes ends the streak.  (Per device: contention is a property of the device, not of the stream.)
#define PP_BACKOFF_BASE 64
#define PP_BACKOFF_MAX 4096
static std::atomic<long long> g_pp_skip[64];
static std::atomic<int> g_pp_streak[64];
static std::atomic<long long> g_pp_skipped{0};
extern "C" int64_t apgp_potrf_backoff_skips(void) { return g_pp_skipped.load(); }
static bool pp_backoff_take(int dev) {          // true: this evaluation skips the persistent launch
    if (dev < 0 || dev >= 64) return false;
    long long left = g_pp_skip[dev].load();
    while (left > 0)
        if (g_pp_skip[dev].compare_exchange_weak(left, left - 1)) { g_pp_skipped.fetch_add(1); return true; }
    return false;
}
static void pp_backoff_report(int dev, bool gave_up) {
    if (dev < 0 || dev >= 64) return;
    if (!gave_up) { g_pp_streak[dev].store(0); return; }
    const int streak = g_pp_streak[dev].fetch_add(1) + 1;
    long long skip = (long long)PP_BACKOFF_BASE << (streak > 7 ? 6 : streak - 1);
    if (skip > PP_BACKOFF_MAX) skip = PP_BACKOFF_MAX;
    g_pp_skip[dev].store(skip);
}
extern "C" int apgp_potrf_mode(int mode) {
    if (mode < 0) return g_potrf_mode.load() | (g_potrf_pairs.load() ? 0 : 16) | (g_potrf_defer8.load() ? 0 : 32);
    if ((mode & ~48) > 3) { apgp_set_error("apgp_potrf_mode: bad argument: mode 0 .. 3 (+ 16: no paired trailing updates, + 32: no deferred tiles)"); return -1; }
    const int prev = g_potrf_mode.exchange(mode & 15) | (g_potrf_pairs.exchange((mode & 16) ? 0 : 1) ? 0 : 16) |
                     (g_potrf_defer8.exchange((mode & 32) ? 0 : POTRF_DEFER8) ? 0 : 32);
    for (int d = 0; d < 64; ++d) { g_pp_skip[d].store(0); g_pp_streak[d].store(0); }   // (an explicit mode ends any back-off)
    return prev;
}
extern "C" int64_t apgp_potrf_fallbacks(void) { return g_potrf_fallbacks.load(); }

static int potrf_device_cus(int dev) {
    static std::mutex mu;
    static int cus[64] = {0};
    if (dev < 0 || dev >= 64) return 0;
    std::lock_guard<std::mutex> lock(mu);
    if (cus[dev] == 0) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = -1;
        cus[dev] = v;
    }
    return cus[dev];
}

// the persistent launch serves one matrix of 64 < n <= 4096 whose byte offsets fit the buffer descriptor, on a device
// with a CU for every row workgroup and at least one more
// How an n x n factorisation (n > 64) is run: -1 = a launch per 64-column step; 0 = ONE persistent launch;
// s0 > 0 = hybrid: the first s0 block columns a launch per step, the trailing (nb - s0) x (nb - s0) blocks as one
// persistent launch.
// (measured, apgp_nll_eval, persistent vs launch per step: 0.165 vs 0.208 ms at n = 512, 0.357 vs 0.464 at 1152, 0.670 vs 0.859 at 2048,
// 1.13 vs 1.34 at 3072, 1.63 vs 1.71 at 3712, but 2.03 vs 1.96 at 4096: while the trailing matrix is large a step is bound by the
// update (45 us at 4096 against 23.5 once it is small), and there the launch-per-step path has all 256 CUs on it -- so above
// PP_AUTO_NB block columns the first steps are its launches and the persistent launch takes over for the last PP_HYBRID_NB
// block columns, where a step is a latency chain.  Thresholds measured: (50, 44) against (58, 48): 1.28 vs 1.31 ms at n = 3328,
// 1.53 vs 1.63 at 3712, 1.79 vs 1.80 at 4096, 2.59 vs 2.62 at 5000)
#ifndef PP_HYBRID_NB
#define PP_HYBRID_NB 44
#endif
static long long potrf_plan(int64_t n, int64_t lda, hipStream_t s) {
    const int mode = g_potrf_mode.load();
    if (mode == 1) return -1;
    const int64_t nb = (n + PB - 1) / PB;
    if (nb < 2) return -1;
    const int cus = potrf_device_cus(apgp_stream_device(s));
    if (nb <= PP_MAX_NB && (nb <= PP_AUTO_NB || mode == 3)) {
        if (lda * n * 8 >= (1ll << 31) || cus < nb + 1) return -1;
        return 0;
    }
    if (mode == 2) return -1;                                            // (the fallback test runs whole persistent launches)
    const long long s0 = nb - PP_HYBRID_NB;
    if (lda * (n - s0 * PB) * 8 >= (1ll << 31) || cus < PP_HYBRID_NB + 1) return -1;
    return s0;
}

// gram (pre_init) -> ONE persistent launch -> potrf_finish_kernel (summary, factored diagonal blocks into place).
// Caller holds apgp_stream_lock(s).  The summary's info slot reads PP_ABORTED if the launch gave up.
// s0 > 0 (hybrid): block columns 0 .. s0 - 1 were factorised and applied to the whole trailing matrix by the
// launch-per-step path (z blocks and running right-hand side included); this launch factorises the trailing matrix.
// cu_budget > 0: the CUs this launch may count on (apgp_nll_eval_batch runs several launches side by side).
static int potrf_persist_locked(double* A, int64_t n, int64_t lda, double* z, int32_t* info_dev, hipStream_t s,
                                double* out5, double* mail, long long seq, long long s0 = 0, int cu_budget = 0) {
    const int dev = apgp_stream_device(s);
    const long long nb_all = (n + PB - 1) / PB, nb = nb_all - s0;
    PersistArgs q;
    PotrfArgs& a = q.a;
    a.A = A; a.rhs = z; a.n = n; a.lda = lda; a.j0 = 0; a.shift = 0.0; a.info = info_dev; a.out5 = out5;
    a.mail = mail; a.seq = seq; a.batch_A = 0; a.batch_rhs = n; a.no_panel = 0; a.info_j0 = 0; a.pair_mode = 0; a.defer8 = 0; a.deferred8 = 0;
    a.zoff = nb_all * (long long)(PB * PB); a.batch_dscr = a.zoff + nb_all * PB;
    a.dscr = apgp_stream_scratch(0, s, (size_t)a.batch_dscr);
    bool fresh = false;
    unsigned long long* calls = nullptr;
    pp_u64* pw = (pp_u64*)apgp_stream_scratch_ex(2, s, (size_t)PP_SCRATCH_WORDS, &fresh, &calls);
    if (!a.dscr || !pw) {
        apgp_set_error("apgp_potrf: scratch allocation failed");
        return -2;
    }
    unsigned long long id = ++*calls;
    if ((unsigned)id == 0u) { fresh = true; id = ++*calls; }             // (the 32-bit granule tag wrapped: start over)
    if (fresh && hipMemsetAsync(pw, 0, (size_t)PP_SCRATCH_WORDS * 8, s) != hipSuccess) {
        apgp_set_error("apgp_potrf: memset failed");
        return -2;
    }
    q.ctl = pw; q.strm = pw + PP_CTL_WORDS; q.zstrm = q.strm + PP_STRM_WORDS;
    q.call_id = id;
    q.timeout = 5000000ull;                                             // 50 ms of the 100 MHz clock
    q.nb = (int)nb;
    q.debug = g_potrf_mode.load() == 2 ? 1 : 0;
    a.abort_word = pw + PP_CTL_ABORT; a.abort_id = id;
    const PotrfArgs full = a;                                           // (the finish kernel works on the whole matrix)
    if (s0 > 0) {
        const long long off = s0 * PB;
        a.A = A + off * lda + off;
        a.n = n - off;
        if (z) a.rhs = z + off;
        a.zoff = full.zoff - s0 * (long long)(PB * PB) + off;           // (dscr + zoff addresses the same z array)
        a.dscr = full.dscr + s0 * (long long)(PB * PB);
        a.info_j0 = off;
    }
    {
        static std::mutex attr_mu;
        static bool attr_set[64] = {false};
        std::lock_guard<std::mutex> lock(attr_mu);
        if (dev >= 0 && dev < 64 && !attr_set[dev]) {
            if (hipFuncSetAttribute((const void*)potrf_persist_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS_BYTES) != hipSuccess) {
                apgp_set_error("apgp_potrf: hipFuncSetAttribute failed");
                return -2;
            }
            attr_set[dev] = true;
        }
    }
    const long long tiles0 = (nb - 1) * (nb - 2) / 2;                   // tiles of the first update step: one per workgroup
    long long nupd = tiles0;
    const long long room = (cu_budget > 0 ? cu_budget : potrf_device_cus(dev)) - nb;
    if (nupd > room) nupd = room;
    // (not one workgroup per tile of the first update step: a tile takes ~3.5 us of a 20 us step, and every further
    // update workgroup costs the row workgroups memory latency -- 30 % of the first step's tiles + 8 measured best:
    // 0.361 vs 0.369 ms at n = 1152, 0.521 vs 0.550 at 1600, 0.689 vs 0.716 at 2048, unchanged from 2560 on, where the
    // CUs left beside the row workgroups are the limit)
#ifndef PP_NUPD_PCT
#define PP_NUPD_PCT 30
#endif
    { const long long want = (tiles0 * PP_NUPD_PCT + 99) / 100 + 8; if (nupd > want) nupd = want; }
    if (nupd < 1) nupd = 1;
    hipLaunchKernelGGL(potrf_persist_kernel, dim3((unsigned)(nb + nupd)), dim3(PP_THREADS), PP_LDS_BYTES, s, q);
    hipLaunchKernelGGL(potrf_finish_kernel, dim3((unsigned)nb_all, 1u), dim3(256), 0, s, full);
    APGP_CHECK_LAUNCH();
    return 0;
}

// `batch` matrices A + b * batch_A (right-hand sides y - shifts[b] -> z + b * n) factorised by
// the same launches: gridDim.y = batch.  The Cholesky of one small matrix is a chain of
// latency-bound steps that leaves most of the chip idle, so a batch costs little more than one.
// pre_init: info and z = y - shift were initialised by the caller's Gram launch (apgp_gram_with_rhs);
// out5 != NULL: the finish kernel also writes the fit summary (5 doubles per matrix)
// (caller holds apgp_stream_lock(s))
static int potrf_run_locked(double* A, int64_t n, int64_t lda, int64_t batch, int64_t batch_A, const double* y,
                            const double* shifts, double* z, int32_t* info_dev, hipStream_t s, bool pre_init = false,
                            double* out5 = nullptr, double* mail = nullptr, long long seq = 0, long long stop_at = 0) {
    // stop_at = s0 > 0 (hybrid, batch 1): block columns 0 .. s0 - 1 only; the last step applies column s0 - 1 to the whole
    // trailing matrix, writes the first tile column back and leaves the rest -- and the finish kernel -- to the caller
    // info = UINT_MAX means "no failure yet"; normalised to 0 by the caller-visible finish kernel
    if (!pre_init && hipMemsetAsync(info_dev, 0xff, sizeof(int32_t) * batch, s) != hipSuccess) {
        apgp_set_error("apgp_potrf: memset failed");
        return -2;
    }
    PotrfArgs a;
    a.A = A; a.rhs = z; a.n = n; a.lda = lda; a.shift = 0.0; a.info = info_dev; a.out5 = out5;
    a.mail = mail; a.seq = seq;
    a.batch_A = batch_A; a.batch_rhs = n;
    a.abort_word = nullptr; a.abort_id = 0; a.no_panel = 0; a.info_j0 = 0; a.pair_mode = 0; a.defer8 = 0; a.deferred8 = 0;
    if (z && !pre_init)
        for (int64_t b = 0; b < batch; ++b)
            hipLaunchKernelGGL(potrf_rhs_init_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, y, shifts[b],
                               z + b * n, (long long)n);
    const long long nb = (n + PB - 1) / PB;
    // block column 0: panel step alone; then one launch per step = trailing update of column jb +
    // panel step of column jb + 1 (potrf_step_kernel)
    a.dscr = nullptr; a.zoff = nb * (long long)(PB * PB); a.batch_dscr = a.zoff + nb * PB;
    if (nb > 1) {
        a.dscr = apgp_stream_scratch(0, s, (size_t)a.batch_dscr * (size_t)batch);
        if (!a.dscr) {
            apgp_set_error("apgp_potrf: scratch allocation failed");
            return -2;
        }
    }
    {
        a.j0 = 0;
        const long long below = n - PB;
        const unsigned pg = below > 0 ? (unsigned)((below + PB - 1) / PB) : 1u;
        hipLaunchKernelGGL(potrf_panel_kernel, dim3(pg, (unsigned)batch), dim3(3 * PB), 0, s, a);
    }
    // steps are paired (narrow, wide) while the trailing matrix is large; the step that hands over to the persistent
    // launch -- and the step before an unpaired one -- is never a narrow one (its deferred tiles would be missing)
    const long long last = (stop_at > 0 ? stop_at : nb - 1) - 1;         // last step launched here
    // a wide step defers part of its tiles to the narrow step that follows it, if one does (PotrfArgs::defer8)
    const bool pairs = g_potrf_pairs.load() != 0;
    const int defer_cfg = g_potrf_defer8.load();
    auto narrow_at = [&](long long jb_) {                                  // would step jb_ be a narrow one (after a wide one)?
        const long long tb_ = (n - (jb_ * PB + PB) + PB - 1) / PB;
        return pairs && jb_ + 1 < nb && tb_ >= POTRF_PAIR_MIN_TB && jb_ + 1 <= last;
    };
    int prev_mode = 0, prev_defer = 0;
    for (long long jb = 0; jb + 1 < nb; ++jb) {
        a.j0 = jb * PB;
        a.no_panel = (stop_at > 0 && jb == stop_at - 1) ? 1 : 0;
        const long long tb = (n - (a.j0 + PB) + PB - 1) / PB;
        a.pair_mode = prev_mode == 1 ? 2 : ((pairs && tb >= POTRF_PAIR_MIN_TB && jb + 1 <= last) ? 1 : 0);
        prev_mode = a.pair_mode;
        a.deferred8 = a.pair_mode == 1 ? prev_defer : 0;
        const long long E = (tb - 3) * (tb - 2) / 2;                       // this step's eligible triangle (from block (3, 3) on)
        a.defer8 = 0;
        if (a.pair_mode == 2 && !a.no_panel && tb >= 4 && narrow_at(jb + 1)) a.defer8 = defer_cfg;
        prev_defer = a.defer8;
        const long long Eprev = (tb - 2) * (tb - 1) / 2;                   // the previous wide step's, one block larger
        long long tiles;
        if (a.pair_mode == 1) tiles = tb + (tb - 1) + potrf_deferred_count(Eprev, a.deferred8);
        else if (a.defer8 > 0) tiles = tb + (2 * tb - 3) + (E - potrf_deferred_count(E, a.defer8));
        else tiles = tb + tb * (tb - 1) / 2;
        hipLaunchKernelGGL(potrf_step_kernel, dim3((unsigned)tiles, (unsigned)batch), dim3(256), 0, s, a);
        if (a.no_panel) {
            APGP_CHECK_LAUNCH();
            return 0;
        }
    }
    hipLaunchKernelGGL(potrf_finish_kernel, dim3((unsigned)nb, (unsigned)batch), dim3(256), 0, s, a);
    APGP_CHECK_LAUNCH();
    return 0;
}

// One factorisation's launches are enqueued as a unit: two host threads on the same stream (ctypes
// releases the GIL) must not interleave theirs -- they share the stream's scratch (scratch.h).
// (one lock per (device, stream): other streams and devices enqueue concurrently)
static int potrf_run(double* A, int64_t n, int64_t lda, int64_t batch, int64_t batch_A, const double* y,
                     const double* shifts, double* z, int32_t* info_dev, hipStream_t s, bool pre_init = false,
                     double* out5 = nullptr) {
    std::lock_guard<std::mutex> enqueue_lock(apgp_stream_lock(s));
    return potrf_run_locked(A, n, lda, batch, batch_A, y, shifts, z, info_dev, s, pre_init, out5);
}

// the host side of the mailbox: polls the sequence word (bounded spin, then the ordinary stream
// synchronisation) and copies the record out.  Caller holds apgp_stream_lock(s).
static int mailbox_wait(ApgpMailbox* mb, long long seq, hipStream_t s, double* out5_host) {
    volatile long long* flag = (volatile long long*)(mb->host + 5);
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) {
        if ((++spins & 255u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(400)) {
            if (hipStreamSynchronize(s) != hipSuccess) {
                apgp_set_error("apgp_nll_eval: stream synchronisation failed");
                return -2;
            }
            if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) {
                apgp_set_error("apgp_nll_eval: result record not written");
                return -2;
            }
            break;
        }
    }
    for (int i = 0; i < 5; ++i) out5_host[i] = mb->host[i];
    return 0;
}

// the 5-value record of an evaluation: through the mailbox, or the 40-byte copy + synchronisation
static int nll_fetch(ApgpMailbox* mb, bool mail, long long seq, hipStream_t s, const double* out5_dev, double* out5_host) {
    if (mail) return mailbox_wait(mb, seq, s, out5_host);
    if (hipMemcpyAsync(out5_host, out5_dev, 5 * sizeof(double), hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess) {
        apgp_set_error("apgp_nll_eval: D2H copy failed");
        return -2;
    }
    return 0;
}

extern "C" int apgp_potrf(double* A, int64_t n, int64_t lda, const double* y, double shift, double* z,
                          int32_t* info_dev, void* stream) {
    APGP_CHECK_ARG(A && info_dev, "null pointer");
    APGP_CHECK_ARG((y == NULL) == (z == NULL), "y and z must be given together");
    APGP_CHECK_ARG(n >= 1 && n <= APGP_MAX_N && lda >= n, "n >= 1 and lda >= n required");
    return potrf_run(A, n, lda, 1, 0, y, &shift, z, info_dev, (hipStream_t)stream);
}

// One gpUtils._nll evaluation (gpUtils.py:46-80) as ONE library call: Gram matrix ->
// Cholesky with the forward solve riding along -> summary record -> 40-byte D2H -> one
// stream synchronisation.  Powell / Nelder-Mead call this hundreds to thousands of times
// per fit; at N = 50 the separate calls' host overhead was as long
// as the kernels themselves.
int apgp_gram_with_rhs(const double* X, int64_t n, const apgp_kernel_t* kern, double* K, int64_t ldk,
                       const double* y, double shift, double* z, int32_t* info_dev, void* stream);   // gram.hip
extern "C" int apgp_nll_eval(const double* X, int64_t n, const apgp_kernel_t* kern, const double* y, double mean,
                             double* K, double* z, int32_t* info_dev, double* out5_dev, double* out5_host,
                             void* stream) {
    APGP_CHECK_ARG(X && kern && K && info_dev && out5_dev && out5_host, "null pointer");
    APGP_CHECK_ARG((y == NULL) == (z == NULL), "y and z must be given together");
    APGP_CHECK_ARG(n >= 1 && n <= APGP_MAX_N, "n >= 1 required");
    hipStream_t s = (hipStream_t)stream;
    int rc;
    // The record comes back through the stream's pinned mailbox: the last kernel's last lane writes it straight
    // into host memory and the host polls the sequence word -- no D2H copy, no stream synchronisation while
    // the evaluation is short (bounded spin, then the ordinary synchronisation): 42.6 -> 26.9 us at N = 50.
    ApgpMailbox* mb = apgp_stream_mailbox(s);
    const bool mail = mb && mb->host;
    std::lock_guard<std::mutex> lock(apgp_stream_lock(s));             // (one mailbox, one scratch per stream)
    const long long seq = mail ? ++mb->seq : 0;
    if (n <= PB && y) {
        // one single-workgroup launch (nll_small_kernel): same values, two launch boundaries fewer
        rc = nll_fused_launch(X, n, kern, y, mean, K, z, info_dev, out5_dev, s, mail ? mb->dev : nullptr, seq);
        if (rc != 0) return rc;
    } else if (n <= 2 * PB && y && (g_potrf_mode.load() == 0 || g_potrf_mode.load() == 3)) {
        // two block columns, still one single-workgroup launch (nll_two_kernel; modes 1 / 2 keep the separate launches)
        rc = nll_fused_launch(X, n, kern, y, mean, K, z, info_dev, out5_dev, s, mail ? mb->dev : nullptr, seq);
        if (rc != 0) return rc;
    } else {
        // three launches fewer than the separate calls: the Gram launch initialises the right-hand side
        // and the info word, the Cholesky's last launch writes the summary
        rc = apgp_gram_with_rhs(X, n, kern, K, n, y, mean, z, info_dev, stream);
        if (rc != 0) return rc;
        long long plan = potrf_plan(n, n, s);
        const int dev = apgp_stream_device(s);
        if (plan >= 0 && g_potrf_mode.load() != 2 && pp_backoff_take(dev)) plan = -1;   // (mode 2 = the fallback test: always launches)
        const bool persist = plan >= 0;
        if (plan > 0) {
            rc = potrf_run_locked(K, n, n, 1, 0, y, &mean, z, info_dev, s, true, nullptr, nullptr, 0, plan);
            if (rc != 0) return rc;
        }
        rc = persist ? potrf_persist_locked(K, n, n, z, info_dev, s, out5_dev, mail ? mb->dev : nullptr, seq, plan)
                     : potrf_run_locked(K, n, n, 1, 0, y, &mean, z, info_dev, s, true, out5_dev, mail ? mb->dev : nullptr, seq);
        if (rc != 0) return rc;
        if (persist) {
            rc = nll_fetch(mb, mail, seq, s, out5_dev, out5_host);
            if (rc != 0) return rc;
            if (out5_host[4] != PP_ABORTED) { pp_backoff_report(dev, false); return rc; }
            // the persistent launch gave up (its workgroups were not all resident in time): the evaluation again,
            // from the Gram matrix, on the multi-launch path
            g_potrf_fallbacks.fetch_add(1);
            if (g_potrf_mode.load() != 2) pp_backoff_report(dev, true);
            const long long seq2 = mail ? ++mb->seq : 0;
            rc = apgp_gram_with_rhs(X, n, kern, K, n, y, mean, z, info_dev, stream);
            if (rc != 0) return rc;
            rc = potrf_run_locked(K, n, n, 1, 0, y, &mean, z, info_dev, s, true, out5_dev, mail ? mb->dev : nullptr, seq2);
            if (rc != 0) return rc;
            return nll_fetch(mb, mail, seq2, s, out5_dev, out5_host);
        }
    }
    return nll_fetch(mb, mail, seq, s, out5_dev, out5_host);
}

// ---- a small batch of mid-size evaluations: their persistent factorisations SIDE BY SIDE in one launch (round 6) -----
// One persistent Cholesky occupies nb + ~30 % of its first step's tiles workgroups (67 of 256 CUs at n = 1152) and is a
// latency chain: several of them next to each other take little more than the time of one (measured, N = 1152:
// 2 / 3 / 4 matrices 366 / 382 / 402 us against 342 for one).  Three launches, as for a single evaluation: the batched
// Gram launch (gram.hip), potrf_persist_batch_kernel (gridDim.y = batch; every matrix with its own scratch region, flags
// and granule streams, its update workgroups sized to 1 / batch of the CUs so that all workgroups of all matrices are
// resident together) and the batched finish launch, whose records come back through the stream's pinned staging area
// (one sequence word per matrix).  Every matrix runs exactly the code of the single call: the same bits.  A matrix whose
// workgroups give up (another process holds CUs) sends the whole batch to the batched launch-per-step path.
int apgp_gram_with_rhs_batch(const double* X, int64_t n, int64_t batch, const apgp_kernel_t* kerns, double* K,
                             const double* y, const double* shifts, double* z, int32_t* info_dev, void* stream);   // gram.hip
static std::atomic<long long> g_pp_side_batches{0};
extern "C" int64_t apgp_nll_side_batches(void) { return g_pp_side_batches.load(); }

// 0: done (out5_host filled); 1: not applicable or gave up -- nothing of the result is valid, the caller runs the
// batched launch-per-step path from the Gram matrices; < 0: error
static int nll_batch_side_by_side(const double* X, int64_t n, int64_t batch, const apgp_kernel_t* kerns, const double* y,
                                  const double* means, double* K, double* z, int32_t* info_dev, double* out5_dev,
                                  double* out5_host, hipStream_t s) {
    const int mode = g_potrf_mode.load();
    if (mode != 0 && mode != 3) return 1;
    if (batch < 2 || batch > PP_BATCH_MAX) return 1;
    if (potrf_plan(n, n, s) != 0) return 1;                               // one persistent launch per matrix, no hybrid
    const int dev = apgp_stream_device(s);
    const int cus = potrf_device_cus(dev);
    const long long nb = (n + PB - 1) / PB;
    const long long budget = cus / batch;                                 // CUs per matrix
    if (budget < nb + 2) return 1;                                        // its row workgroups + >= 2 update workgroups
    std::lock_guard<std::mutex> lock(apgp_stream_lock(s));
    double* io_dev = nullptr;
    double* io = apgp_stream_pinned_io(s, (size_t)batch * 8, &io_dev);
    ApgpMailbox* mb = io ? apgp_stream_mailbox(s) : nullptr;
    if (!io || !mb || !mb->host) return 1;
    if (pp_backoff_take(dev)) return 1;
    int rc = apgp_gram_with_rhs_batch(X, n, batch, kerns, K, y, means, z, info_dev, (void*)s);
    if (rc != 0) return rc;
    PersistBatchArgs qb;
    PotrfArgs full;
    {
        PotrfArgs& a = full;
        a.A = K; a.rhs = z; a.n = n; a.lda = n; a.j0 = 0; a.shift = 0.0; a.info = info_dev; a.out5 = out5_dev;
        a.batch_A = n * n; a.batch_rhs = n; a.no_panel = 0; a.info_j0 = 0; a.pair_mode = 0; a.defer8 = 0; a.deferred8 = 0;
        a.zoff = nb * (long long)(PB * PB); a.batch_dscr = a.zoff + nb * PB;
        a.dscr = apgp_stream_scratch(0, s, (size_t)a.batch_dscr * (size_t)batch);
    }
    bool fresh = false;
    unsigned long long* calls = nullptr;
    pp_u64* pw = (pp_u64*)apgp_stream_scratch_ex(2, s, (size_t)PP_SCRATCH_WORDS * (size_t)batch, &fresh, &calls);
    if (!full.dscr || !pw) {
        apgp_set_error("apgp_nll_eval_batch: scratch allocation failed");
        return -2;
    }
    unsigned long long id = ++*calls;
    if ((unsigned)id == 0u) { fresh = true; id = ++*calls; }             // (the 32-bit granule tag wrapped: start over)
    if (fresh && hipMemsetAsync(pw, 0, (size_t)PP_SCRATCH_WORDS * 8 * (size_t)batch, s) != hipSuccess) {
        apgp_set_error("apgp_nll_eval_batch: memset failed");
        return -2;
    }
    const long long seq = ++mb->seq;
    full.abort_word = pw + PP_CTL_ABORT; full.abort_id = id; full.batch_abort = PP_SCRATCH_WORDS;
    full.mail = io_dev; full.batch_mail = 8; full.seq = seq;
    for (int64_t b = 0; b < PP_BATCH_MAX; ++b) {
        const int64_t m = b < batch ? b : 0;                              // (unused records: copies of matrix 0's, never read)
        PersistArgs& q = qb.m[b];
        q.a = full;
        q.a.A = K + m * n * n; q.a.rhs = z + m * n; q.a.info = info_dev + m; q.a.out5 = out5_dev + 5 * m;
        q.a.dscr = full.dscr + m * full.batch_dscr;
        q.a.batch_A = 0; q.a.batch_rhs = n; q.a.batch_abort = 0; q.a.batch_mail = 0; q.a.mail = nullptr;
        q.ctl = pw + m * (long long)PP_SCRATCH_WORDS; q.strm = q.ctl + PP_CTL_WORDS; q.zstrm = q.strm + PP_STRM_WORDS;
        q.a.abort_word = q.ctl + PP_CTL_ABORT; q.a.abort_id = id;
        q.call_id = id;
        q.timeout = 5000000ull;                                           // 50 ms of the 100 MHz clock
        q.nb = (int)nb;
        q.debug = 0;
    }
    {
        static std::mutex attr_mu;
        static bool attr_set[64] = {false};
        std::lock_guard<std::mutex> alock(attr_mu);
        if (dev >= 0 && dev < 64 && !attr_set[dev]) {
            if (hipFuncSetAttribute((const void*)potrf_persist_batch_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS_BYTES) != hipSuccess) {
                apgp_set_error("apgp_nll_eval_batch: hipFuncSetAttribute failed");
                return -2;
            }
            attr_set[dev] = true;
        }
    }
    const long long tiles0 = (nb - 1) * (nb - 2) / 2;
    long long nupd = (tiles0 * PP_NUPD_PCT + 99) / 100 + 8;               // as the single launch (potrf_persist_locked) ...
    if (nupd > tiles0) nupd = tiles0;
    if (nupd > budget - nb) nupd = budget - nb;                           // ... within this matrix's share of the CUs
    if (nupd < 1) nupd = 1;
    hipLaunchKernelGGL(potrf_persist_batch_kernel, dim3((unsigned)(nb + nupd), (unsigned)batch), dim3(PP_THREADS), PP_LDS_BYTES, s, qb);
    hipLaunchKernelGGL(potrf_finish_kernel, dim3((unsigned)nb, (unsigned)batch), dim3(256), 0, s, full);
    APGP_CHECK_LAUNCH();
    // the records: one sequence word per matrix (bounded spin, then the ordinary synchronisation)
    bool gave_up = false, synced = false;
    const auto t0 = std::chrono::steady_clock::now();
    for (int64_t b = 0; b < batch; ++b) {
        volatile long long* flag = (volatile long long*)(io + 8 * b + 5);
        unsigned spins = 0;
        while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) {
            if ((++spins & 255u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(800)) {
                if (synced || hipStreamSynchronize(s) != hipSuccess || __atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) {
                    apgp_set_error("apgp_nll_eval_batch: result record not written");
                    return -2;
                }
                synced = true;
            }
        }
        for (int i = 0; i < 5; ++i) out5_host[5 * b + i] = io[8 * b + i];
        if (out5_host[5 * b + 4] == PP_ABORTED) gave_up = true;
    }
    if (gave_up) {
        g_potrf_fallbacks.fetch_add(1);
        pp_backoff_report(dev, true);
        if (hipStreamSynchronize(s) != hipSuccess) return -2;             // (nothing of the launch is still running when the re-run starts)
        return 1;
    }
    pp_backoff_report(dev, false);
    g_pp_side_batches.fetch_add(1);
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
    APGP_CHECK_ARG(n >= 1 && n <= APGP_MAX_N, "n >= 1 required");
    int rc;
    if (n <= 2 * PB && batch <= 64 && (g_potrf_mode.load() == 0 || g_potrf_mode.load() == 3)) {
        // README sizes (round 5): ONE launch of the fused evaluation, a workgroup per matrix (nll_small_kernel /
        // nll_two_kernel); constants and shifts go in -- and the records come back -- through the stream's pinned,
        // device-mapped staging area, each workgroup posting its own sequence word: no copy, no synchronisation.
        // Every matrix runs the code of the single call: the same bits.
        hipStream_t s = (hipStream_t)stream;
        std::lock_guard<std::mutex> lock(apgp_stream_lock(s));
        constexpr size_t KCD = sizeof(KernConst) / sizeof(double);
        double* io_dev = nullptr;
        double* io = apgp_stream_pinned_io(s, (size_t)batch * (KCD + 1 + 8), &io_dev);
        ApgpMailbox* mb = io ? apgp_stream_mailbox(s) : nullptr;
        if (io && mb && mb->host) {
            KernConst* kcs = (KernConst*)io;
            for (int64_t b = 0; b < batch; ++b) {
                APGP_CHECK_ARG(apgp_make_kernconst(kerns + b, kcs + b) == 0, "kernel parameters");
                APGP_CHECK_ARG(kcs[b].dpad == kcs[0].dpad && kcs[b].ndim == kcs[0].ndim, "kernels of one batch share their dimension");
            }
            double* sh = io + (size_t)batch * KCD;
            for (int64_t b = 0; b < batch; ++b) sh[b] = means[b];
            double* rec = sh + batch;
            const long long seq = ++mb->seq;
            NllSmallArgs q;
            q.X = X; q.y = y; q.K = K; q.z = z; q.info = info_dev; q.out5 = out5_dev; q.n = n; q.shift = 0.0;
            q.kc = kcs[0]; q.mail = nullptr; q.seq = seq;
            q.bkc = (const KernConst*)io_dev; q.bshift = io_dev + (size_t)batch * KCD; q.brec = io_dev + (size_t)batch * (KCD + 1);
            if ((rc = nll_fused_launch_q<true>(q, kcs[0].dpad, (int)batch, s)) != 0) return rc;
            const auto t0 = std::chrono::steady_clock::now();
            bool synced = false;
            for (int64_t b = 0; b < batch; ++b) {
                volatile long long* flag = (volatile long long*)(rec + 8 * b + 5);
                unsigned spins = 0;
                while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) {
                    if ((++spins & 255u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(600)) {
                        if (synced || hipStreamSynchronize(s) != hipSuccess || __atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) {
                            apgp_set_error("apgp_nll_eval_batch: result record not written");
                            return -2;
                        }
                        synced = true;
                    }
                }
                for (int i = 0; i < 5; ++i) out5_host[5 * b + i] = rec[8 * b + i];
            }
            return 0;
        }
    }
    hipStream_t s = (hipStream_t)stream;
    if (batch >= 2 && batch <= PP_BATCH_MAX) {
        rc = nll_batch_side_by_side(X, n, batch, kerns, y, means, K, z, info_dev, out5_dev, out5_host, s);
        if (rc <= 0) return rc;                                            // (1: not applicable / gave up -- the batched launches below)
    }
    for (int64_t b = 0; b < batch; ++b)
        if ((rc = apgp_gram_with_rhs(X, n, kerns + b, K + b * n * n, n, y, means[b], z + b * n, info_dev + b, stream)) != 0) return rc;
    if ((rc = potrf_run(K, n, n, batch, n * n, y, means, z, info_dev, s, true, out5_dev)) != 0) return rc;
    if (hipMemcpyAsync(out5_host, out5_dev, 5 * sizeof(double) * batch, hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess) {
        apgp_set_error("apgp_nll_eval_batch: D2H copy failed");
        return -2;
    }
    return 0;
}
