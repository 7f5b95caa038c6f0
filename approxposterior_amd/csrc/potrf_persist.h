// The blocked Cholesky of potrf.hip as ONE persistent launch (round 4): no launch boundary between the 64-column
// steps, the trailing update off the critical path, and the 64 x 64 products of the next block column fed group by
// group while the current one is still being factorised.  Included by potrf.hip (same translation unit: it reuses
// the panel wavefront roles unchanged, so every element sees the arithmetic of the multi-launch path in the same
// order and the two factorisations are bit-identical -- tests/test_gpu_parity.py compares them).
//
// Replaces, per gpUtils._nll evaluation (gpUtils.py:46-80 -> george GP.log_likelihood -> BasicSolver.compute),
// the chain of 18 (N = 1152) .. 64 (N = 4096) dependent potrf_step_kernel launches.
//
// Grid: nb "row" workgroups (one per 64-row block, 512 threads = 8 wavefronts, one per CU) + "update" workgroups.
//
// Row workgroup r lives through the steps s = 0 .. r.  At step s < r it holds T = tile (r, s) and D = the diagonal
// block (s, s), both updated with the block columns < s, and runs the panel step of the multi-launch path:
//   wavefront 0  factorises D (redundantly in every row workgroup, as before: no workgroup waits for a factor);
//   wavefront 2  is its helper (columns 32 .. 63 of groups 0 .. 7; it hands the next group's four columns back first);
//                afterwards it stages the NEXT step's two tiles -- tile (r, s+1) and the diagonal block (s+1, s+1),
//                final in memory once the update workgroups have flagged them -- in LDS, coalesced, exactly where the
//                hand-over puts the differences (pp_stage_tiles);
//   wavefront 1  solves the 64 rows of T one 4-column group behind and publishes each solved group to LDS (the A
//                operand of this workgroup's next products); once done it stores L(r, s) to memory (write-through;
//                the flag for the update workgroups follows at the head of the next step, when the stores have drained);
//   wavefront 3  in workgroup s + 1 -- whose rows are the B operand of everybody's next products -- forwards every
//                solved group to global memory as data-tagged granules (16 bytes per value: {low word, tag, high
//                word, tag}; the consumer needs no flag, the producer no fence or wait); in every other workgroup it
//                receives them (four 16-byte write-through-coherent loads per lane and look) into LDS; it also
//                carries this block row's right-hand side of the forward solve z = L^-1 (y - mean);
//   wavefronts 4-7 multiply: tile (r, s+1) -= L(r, s) L(s+1, s)^T and the next diagonal block (s+1, s+1) -=
//                L(s+1, s) L(s+1, s)^T, one k = 4 step of v_mfma_f64_4x4x4_4b per published group (no memory access),
//                so that when the last group of the factorisation is out only ONE k-step, the subtraction from the
//                staged values and the workgroup's one barrier per step remain.
//   The critical path per 64-column step is then the factorisation itself + one granule hand-off + one k-step
//   (the multi-launch path: launch boundary + operand loads + two 64^3 products + factorisation).
//   At step r the workgroup factorises its own diagonal block for the output (L_rr, info, z_r): nobody waits for that.
// Update workgroups apply block column s to the tiles (i, q), q >= s + 2 ("update step" s), one tile at a time on all
//   eight wavefronts, as soon as all rows of L(:, s) are in memory -- one whole row step ahead of their use: tile
//   (i, s + 2) is needed at the END of row step s + 1.  Tiles are owned statically (column-major index modulo the
//   number of update workgroups), column s + 2 first.
//
// Hand-offs (MI355X_MICROARCH.md, inter-workgroup visibility; measured with tools/probes/handoff_probe.hip):
//   payload stores are write-through (sc1), every storing wavefront drains (s_waitcnt vmcnt(0)) before ONE relaxed
//   agent-scope flag store; consumers poll ONE word relaxed, then read with sc1 (write-through-coherent) loads -- no
//   acquire anywhere: buffer_inv sc1 empties the XCD's whole L2 for everybody (update workgroups took one per step until
//   it was measured: 4-5 % of the evaluation at n = 2048 .. 3712).
//   Flags and granule tags are call-unique (a per-stream call counter), so nothing is zeroed between calls.
// Every global spin is bounded: on timeout (the workgroups are not all resident -- a foreign kernel holds CUs --
//   or a bug) the call is marked aborted and the host re-runs the evaluation on the multi-launch path.
#pragma once

typedef unsigned long long pp_u64;
typedef unsigned int pp_u32x4 __attribute__((ext_vector_type(4)));

#define PP_MAX_NB 64
#ifndef PP_AUTO_NB
#define PP_AUTO_NB 50
#endif
//                              // block columns up to which the persistent launch is the default
#define PP_W_RECV 3                               // wavefront of the receiver role (see pp_row_role)
#define PP_THREADS 512
#define PP_CHUNK (64 * 18)                        // a 16-column chunk of a 64-row tile, rows padded to 18 (apgp_gemm64_tile's)
// LDS map of a row workgroup (doubles)
#define PP_LS 0                                   // [64][66]  diagonal block: staged rows, then the published factor
#define PP_AS (PP_LS + 64 * 66)                   // [2][4][64][18]  own solved tile L(r, s), by step parity
#define PP_BS (PP_AS + 2 * 4 * PP_CHUNK)          // [4][64][18]     received L(s+1, s)
#define PP_INVD (PP_BS + 4 * PP_CHUNK)
#define PP_ZBLK (PP_INVD + 64)
#define PP_ZROW (PP_ZBLK + 64)
#define PP_ZST (PP_ZROW + 64)
#define PP_INTS (PP_ZST + 64)                     // 64 ints: 2 x 8 step-parity counters, 17 / 18 "give up", from 24: 2 x 16 super-group flags
#define PP_XFER (PP_INTS + 32)                    // [64][18]  the tile's columns of the next super-group, matrix wavefronts -> solver (potrf_persist_sg.h)
#define PP_LDS_DOUBLES (160 * 1024 / 8)
#define PP_LDS_BYTES (PP_LDS_DOUBLES * 8)
static_assert(PP_XFER + 64 * 18 <= PP_LDS_DOUBLES, "one row workgroup per CU");
// update role: two buffers (current / next tile) of the two whole operand tiles of a product (chunk layout), counters
#define PP_UPD_HALF (2 * 4 * PP_CHUNK)
#define PP_UPD_INTS (2 * PP_UPD_HALF)
static_assert(PP_UPD_INTS + 4 <= PP_LDS_DOUBLES, "update role: two buffers of two whole tiles + counters");
static_assert(64 * 66 <= 4 * PP_CHUNK, "the handed-over tile fits one parity of As");
// control block (pp_u64 words) in the stream's scratch
#define PP_CTL_ABORT 0
#define PP_CTL_ROWDONE 8                          // [64]      (call << 8) | block columns of this block row in memory
#define PP_CTL_TILEFINAL 128                      // [64][64]  call id once tile (i, q) carries the block columns < q - 1
#define PP_CTL_WORDS (128 + 64 * 64)
#define PP_STRM_WORDS ((long long)PP_MAX_NB * 16 * 64 * 8)   // [step][group][lane][4 doubles x 2 granules]
#define PP_ZSTRM_WORDS ((long long)PP_MAX_NB * 64 * 2)       // [step][lane][2 granules]
#define PP_SCRATCH_WORDS (PP_CTL_WORDS + PP_STRM_WORDS + PP_ZSTRM_WORDS)
#define PP_ABORTED (-7777.0)                      // out5[4] of an aborted call

// -DPP_STAMPS (tools/persist_stamps.py builds its own copy of the library with it): 100 MHz wall-clock stamps of
// the LAST row workgroup, per step -- 0 step start | 1 factor done | 2 solve done | 3 last group received |
// 4 k-step 15 done | 5 tile values requested (flags seen) | 6 hand-over written | 7 z applied | 8 L(r, s-1) stored |
// 9 k-step 5 done | 10 first group received | 11 k-step 0 done | 12-14 k-steps 12-14 done | 15 group 14 received
#ifdef PP_STAMPS
__device__ unsigned long long pp_stamps[64 * 24];
#define PP_STAMPP(s_, i_) do { if ((int)blockIdx.x == (s_) + 1 && (threadIdx.x & 63) == 0) pp_stamps[(s_) * 24 + (i_)] = __builtin_amdgcn_s_memrealtime(); } while (0)   /* the step's producer workgroup */
#define PP_STAMP(s_, i_) do { if ((int)blockIdx.x == q->nb - 1 && (threadIdx.x & 63) == 0) pp_stamps[(s_) * 24 + (i_)] = __builtin_amdgcn_s_memrealtime(); } while (0)
extern "C" int apgp_debug_read_stamps(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(pp_stamps), sizeof(unsigned long long) * 64 * 24) == hipSuccess ? 0 : -2;
}
// the update half-workgroup that owns tile (s + 2, s + 2), per update step: 0 rows seen | 1 acquired | 2 first tile pair: operands in LDS | 3 products done |
// 4 stored and drained | 5 step done
__device__ unsigned long long pp_wstamps[64 * 8];      // arrival of each wavefront of the last row workgroup at the step's barrier
#define PP_WSTAMP(s_) do { if ((int)blockIdx.x == q->nb - 1 && (threadIdx.x & 63) == 0) pp_wstamps[(s_) * 8 + (threadIdx.x >> 6)] = __builtin_amdgcn_s_memrealtime(); } while (0)
extern "C" int apgp_debug_read_wstamps(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(pp_wstamps), sizeof(unsigned long long) * 64 * 8) == hipSuccess ? 0 : -2;
}
__device__ unsigned long long pp_ustamps[64 * 8];
#define PP_USTAMP(s_, i_) do { if (((i_) >= 2 && (i_) <= 4) ? (ustamp_me && (threadIdx.x & 255) == 0) : (ustamp_wg && threadIdx.x == 0)) pp_ustamps[(s_) * 8 + (i_)] = __builtin_amdgcn_s_memrealtime(); } while (0)
extern "C" int apgp_debug_read_ustamps(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(pp_ustamps), sizeof(unsigned long long) * 64 * 8) == hipSuccess ? 0 : -2;
}
// (round 5, super-group step) the factorising / solving wavefront of the last row workgroup: 0 entry | 1 super-group 0 done |
// 2 boundary 1 received | 3 super-group 1 done | 4 boundary 2 received | 5 super-group 2 done | 6 boundary 3 received | 7 done
#define PP_FSTAMP(s_, i_) do { if ((int)blockIdx.x == q->nb - 1 && (threadIdx.x & 63) == 0) pp_fstamps[(s_) * 8 + (i_)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define PP_GSTAMP(s_, i_) do { if ((int)blockIdx.x == q->nb - 1 && (threadIdx.x & 63) == 0) pp_gstamps[(s_) * 8 + (i_)] = __builtin_amdgcn_s_memrealtime(); } while (0)
// event log of the four matrix wavefronts of the last row workgroup at step 5: [mw][event][0 start | 1 end | 2 kind]
// (kind: 100 + H catch-up D | 200 + H catch-up T | g k-step)
__device__ unsigned long long pp_estamps[4 * 32 * 3];
#define PP_ESTAMP_BEGIN(kind_) do { if (s == 5 && (int)blockIdx.x == q->nb - 1 && lane == 0 && nev < 32) { pp_estamps[(mw * 32 + nev) * 3 + 0] = __builtin_amdgcn_s_memrealtime(); pp_estamps[(mw * 32 + nev) * 3 + 2] = (kind_); } } while (0)
#define PP_ESTAMP_END() do { if (s == 5 && (int)blockIdx.x == q->nb - 1 && lane == 0 && nev < 32) { pp_estamps[(mw * 32 + nev) * 3 + 1] = __builtin_amdgcn_s_memrealtime(); } ++nev; } while (0)
extern "C" int apgp_debug_read_estamps(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(pp_estamps), sizeof(unsigned long long) * 4 * 32 * 3) == hipSuccess ? 0 : -2;
}
#else
#define PP_ESTAMP_BEGIN(kind_) do { } while (0)
#define PP_ESTAMP_END() do { } while (0)
#define PP_FSTAMP(s_, i_) do { } while (0)
#define PP_GSTAMP(s_, i_) do { } while (0)
#define PP_STAMP(s_, i_) do { } while (0)
#define PP_WSTAMP(s_) do { } while (0)
#define PP_STAMPP(s_, i_) do { } while (0)
#define PP_USTAMP(s_, i_) do { } while (0)
#endif

struct PersistArgs {
    PotrfArgs a;
    pp_u64* ctl;
    pp_u64* strm;
    pp_u64* zstrm;
    pp_u64 call_id;
    pp_u64 timeout;       // 100 MHz ticks
    int nb;
    int debug;            // 1: workgroup 0 aborts at once (exercises the host's fallback)
};

__device__ __forceinline__ pp_u64 pp_ld(const pp_u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void pp_st(pp_u64* p, pp_u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double pp_ld_f64(const double* p) { return __longlong_as_double((long long)pp_ld((const pp_u64*)p)); }
__device__ __forceinline__ void pp_st_f64(double* p, double v) { pp_st((pp_u64*)p, (pp_u64)__double_as_longlong(v)); }
// 16-byte buffer store whose data registers stay untouched for two more issue slots.  gfx9's rule "a VMEM store of more
// than 64 bits followed by a VALU write of its data registers needs a wait state" has an exception in the ISA manual --
// and in LLVM's hazard recogniser -- for MUBUF stores whose soffset is an SGPR; on this chip the store still reads its
// data late: hipcc recycled the forwarder's data registers in the very next instruction (v_mov into the first dword)
// and, whenever no other wavefront's instruction happened to fall in between, lanes 12-15 of every 16 stored the NEXT
// value's low word under THIS value's tag (found when the receiver moved to a SIMD it shares with a wavefront that
// rarely yields: factor off by 1e-7 relative, silently).  The empty-looking asm keeps the registers live across the nop.
#define PP_STORE16(data_, rs_, voff_, soff_, aux_) do {                                              \
        __builtin_amdgcn_raw_buffer_store_b128((data_), (rs_), (voff_), (soff_), (aux_));            \
        asm volatile("s_nop 1" : : "v"(data_));                                                      \
    } while (0)
__device__ __forceinline__ void pp_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// one failed poll of a global spin: sleeps; every 32nd time looks at the abort word and the clock.
// true = give up (the abort word carries this call's id, set here on timeout or by somebody else)
struct PpSpin { pp_u64 t0 = 0; unsigned it = 0; };
__device__ __forceinline__ bool pp_give_up(PpSpin& sp, pp_u64* ctl, const pp_u64 call_id, const pp_u64 timeout) {
    __builtin_amdgcn_s_sleep(1);
    if ((++sp.it & 31u) != 0) return false;
    if (pp_ld(ctl + PP_CTL_ABORT) == call_id) return true;
    const pp_u64 now = __builtin_amdgcn_s_memrealtime();
    if (sp.t0 == 0) { sp.t0 = now; return false; }
    if (now - sp.t0 > timeout) { pp_st(ctl + PP_CTL_ABORT, call_id); return true; }
    return false;
}
// (an expired guard raises `trip` -- this step's "give up" word: the workgroup leaves at the next step head / at the role's
// exit and marks the CALL aborted, PP_ROLE_EXIT; before round 5 the wait just fell through and the wavefront went on with
// stale operands while the launch could still finish "successfully")
__device__ __forceinline__ void pp_lds_wait_ge(const int* p, int need, int* trip) {
    PANEL_SPIN_WHILE_T(lds_load_volatile(p) < need, trip);
    PANEL_FENCE();
}

// wavefront 1 of a row workgroup: panel_solve_wave with every solved 4-column group published at once to As (this
// workgroup's A operand, chunk layout).  In the workgroup whose rows are everybody's B operand wavefront 3 forwards
// the groups from there to the granule stream (the stores would cost the solving wavefront ~1 us per step, and its
// groups pace everybody).  The rows themselves go to memory from As once all sixteen groups are solved (pp_role_solve).
__device__ __forceinline__ void pp_solve_wave(const int lane, double (&x)[PB], const double (*Ls)[PB + 2], const double* invd,
                                              int* prog_p, double* As_par, int* xprog_p, int* trip) {
    static_for<PB / CB>([&](auto cc_) {
        constexpr int cc = decltype(cc_)::value, c0 = CB * cc;
        pp_lds_wait_ge(prog_p, c0 + CB, trip);           // columns c0 .. c0 + CB - 1 of L_jj published
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
        {
            double* dst = As_par + (cc >> 2) * PP_CHUNK + lane * 18 + 4 * (cc & 3);
            *(f64x2*)dst = (f64x2){xs[0], xs[1]};
            *(f64x2*)(dst + 2) = (f64x2){xs[2], xs[3]};
            lds_store_volatile(xprog_p, cc + 1);         // (same wavefront: LDS stores stay in order)
        }
        PANEL_FENCE();
        panel_trailing<c0, c0 + CB, PB, 0, PB>(x, xs, Ls);
    });
}

// one k = 4 step (group g of the current block column) of a matrix wavefront's 32 x 32 quadrant:
// acc += A_g B_g^T (own rows x received rows) and, with TWO, acc2 += B_g B_g^T -- the operations and operand
// order of apgp_gemm64_tile2's inner loop, so the sums carry the same bits
template <bool TWO>
__device__ __forceinline__ void pp_kstep(const double* Ach, const double* Bch, const int ks, const int lane, const int wr, const int wc,
                                         const int (&bcol)[4], double (&acc)[2][2][4], double (&acc2)[2][2][4]) {
    double af[2], bf[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i) af[i] = Ach[(wr + 16 * i + (lane & 15)) * 18 + ks * 4 + (lane >> 4)];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < 2; ++j) bf[j][r] = Bch[(wc + 16 * j + bcol[r]) * 18 + ks * 4 + (lane >> 4)];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(af[i], bf[j][r], acc[i][j][r], 0, 0, 0);
    if constexpr (TWO) {
        double bfa[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) bfa[i] = Bch[(wr + 16 * i + (lane & 15)) * 18 + ks * 4 + (lane >> 4)];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc2[i][j][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(bfa[i], bf[j][r], acc2[i][j][r], 0, 0, 0);
    }
}

// ---------------------------------------------------------------------------
// row workgroup.  Each wavefront role is its own (not inlined) function: one register allocation per role instead
// of one for a 30,000-instruction body (which spilled in the matrix wavefronts' loop).  The roles read the call's
// arguments from the kernel-argument segment (scalar loads: uniform) and rebuild their LDS pointers from the byte
// offset of the dynamic region.
// ---------------------------------------------------------------------------
// (the kernel-argument segment pointer is only defined in the kernel itself -- in a called function the builtin
// reads as NULL on this toolchain -- so the kernel passes it down as two 32-bit halves, made uniform again here)
typedef const __attribute__((address_space(4))) PersistArgs* pp_args_ptr;
struct PpKarg { unsigned lo, hi; };
__device__ __forceinline__ pp_args_ptr pp_args(PpKarg k) {
    const unsigned long long v = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)k.hi) << 32) |
                                 (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)k.lo);
    return (pp_args_ptr)v;
}
__device__ __forceinline__ double* pp_lds_base(unsigned off) {
    return (double*)(__attribute__((address_space(3))) double*)(size_t)__builtin_amdgcn_readfirstlane(off);
}
#define PP_NOINLINE __attribute__((noinline))
// every role runs the same loop over the steps s = 0 .. r, with the workgroup's ONE barrier per step at its end (the
// roles are functions that are entered once: a function's callee-saved registers are stored and reloaded through
// scratch memory at entry and exit -- once per launch, not once per step)
// A wavefront that finds a "give up" word raised -- by a global spin that timed out (which has marked the call aborted
// already) or by an LDS wait whose guard expired (which has not) -- marks the call aborted itself before it leaves:
// whatever this workgroup has or has not stored, the host must not trust the launch (the last row workgroup
// is awaited by nobody, its stale L(nb-1, .), z and logdet would have passed as a result).
#define PP_ROLE_ABORT(q_) do { if ((threadIdx.x & 63) == 0) pp_st((q_)->ctl + PP_CTL_ABORT, (q_)->call_id); } while (0)
#define PP_STEP_LOOP_BEGIN(lds_)                                                                     \
    for (int s_in = 0, r_ = (int)blockIdx.x; s_in <= r_; ++s_in) {                                   \
        if (lds_load_volatile((int*)((lds_) + PP_INTS) + 17 + ((s_in + 1) & 1))) { PP_ROLE_ABORT(q); return; }   /* somebody gave up during the previous step (uniform) */
#define PP_STEP_LOOP_END()                                                                           \
        PP_WSTAMP(s_in);                                                                             \
        if (s_in < r_) __syncthreads();                                                              \
    }                                                                                                \
    PP_ROLE_EXIT(lds, q);
// after a role's last step: a guard that expired during it (either parity) still aborts the call
#define PP_ROLE_EXIT(lds_, q_) do {                                                                  \
        const int* gi_ = (const int*)((lds_) + PP_INTS) + 17;                                        \
        if (lds_load_volatile(gi_) | lds_load_volatile(gi_ + 1)) PP_ROLE_ABORT(q_);                  \
    } while (0)

struct PpStep {            // what every role derives from (r, s)
    int r, s, bs;
    long long j0;
    bool producer;
    int* cnt;              // this step's counters: 0 prog | 1 hflag | 2 xprog | 3 bprog | 4 zflag | 5 cflag | 6 As_prev free
    int* cnt2;             // super-group flags, one word per matrix wavefront: 0-3 diagonal block handed back | 4-7 tile | 8-11 staged values in registers
    int* abl;
    double* As_cur;
    double* As_prev;
};
__device__ __forceinline__ PpStep pp_step(double* lds, int s_in, long long n) {
    PpStep p;
    p.r = (int)blockIdx.x;
    p.s = __builtin_amdgcn_readfirstlane(s_in);
    p.j0 = (long long)p.s * PB;
    p.bs = (int)((n - p.j0) < PB ? (n - p.j0) : PB);
    p.producer = p.r == p.s + 1;                               // this workgroup's rows are the B operand of the step
    int* ints = (int*)(lds + PP_INTS);
    // step-parity counters: [par][0 prog | 1 hflag | 2 xprog | 3 bprog | 4 zflag | 5 cflag | 6 As_prev free]; others: 17, 18 "give up" by
    // step parity (set during step s, read at the head of step s + 1: never while it may still be written)
    p.cnt = ints + 8 * (p.s & 1);
    p.cnt2 = ints + 24 + 16 * (p.s & 1);
    p.abl = ints + 17 + (p.s & 1);
    p.As_cur = lds + PP_AS + (p.s & 1) * 4 * PP_CHUNK;         // this step's solved tile; at the step's start: the handed-over tile [64][66]
    p.As_prev = lds + PP_AS + ((p.s & 1) ^ 1) * 4 * PP_CHUNK;  // L(r, s-1); at the step's end: the hand-over of the next tile
    return p;
}



// ---------------- wavefront 3: receiver -- z of the previous block column, then the groups of L(s+1, s) ----------------
// rhs_r: the running right-hand side of the forward solve for this lane's matrix row
__device__ PP_NOINLINE void pp_role_recv(unsigned lds_off, PpKarg karg, double rhs_r) {
    pp_args_ptr q = pp_args(karg);
    double* lds = pp_lds_base(lds_off);
    const int lane = threadIdx.x & 63;
    const long long n = q->a.n;
    PP_STEP_LOOP_BEGIN(lds)
    const PpStep p = pp_step(lds, s_in, n);
    const int s = p.s, r = p.r;
    const bool has_row = (long long)r * PB + lane < n;
    const unsigned tag = (unsigned)q->call_id;
    const pp_u64 call_id = q->call_id, timeout = q->timeout;
    pp_u64* ctl = q->ctl;
    double* Bs = lds + PP_BS;
    double* zrow = lds + PP_ZROW;
    double* zst = lds + PP_ZST;
    int* bprog = p.cnt + 3;
    int* zflag = p.cnt + 4;
    bool dead = false;
    if (q->a.rhs && s >= 1) {
        const pp_u64* src = q->zstrm + ((long long)(s - 1) * 64 + lane) * 2;
        pp_u64 g0, g1;
        PpSpin sp;
        for (;;) {
            g0 = pp_ld(src); g1 = pp_ld(src + 1);
            if (__all((unsigned)(g0 >> 32) == tag && (unsigned)(g1 >> 32) == tag)) break;
            if (pp_give_up(sp, ctl, call_id, timeout)) { dead = true; break; }
        }
        if (!dead) {
            zst[lane] = __hiloint2double((int)(unsigned)g1, (int)(unsigned)g0);
            // rhs_row -= L(row, block s-1) . z_(s-1): the two-accumulator sum of panel_solve_wave
            double d0 = 0.0, d1 = 0.0;
#pragma unroll
            for (int k = 0; k < PB; k += 2) {
                const f64x2 xv = *(const f64x2*)(p.As_prev + (k >> 4) * PP_CHUNK + lane * 18 + (k & 15));
                const f64x2 zv = *(const f64x2*)(zst + k);
                d0 = fma(xv.x, zv.x, d0);
                d1 = fma(xv.y, zv.y, d1);
            }
            if (has_row) rhs_r -= d0 + d1;
        }
        PP_STAMP(s, 7);
    }
    lds_store_volatile(p.cnt + 6, 1);                          // (done with As_prev = L(r, s-1): wavefront 2 may stage the next tile's values there)
    if (!dead && r == s) {
        zrow[lane] = rhs_r;
        lds_store_volatile(zflag, 1);
    } else if (!dead && !p.producer) {
        const __amdgpu_buffer_rsrc_t rs_strm = __builtin_amdgcn_make_buffer_rsrc((void*)q->strm, 0, (int)(PP_STRM_WORDS * 8), 0x00020000);
        // (measured: 0.647 vs 0.670 ms at n = 2048, 1.09 vs 1.13 at 3072; neutral at 1152, +1 % at 512)
        const bool poll_last = q->nb >= 20;
        static_for<PB / CB>([&](auto cc_) {
            constexpr int cc = decltype(cc_)::value;
            if (dead) return;
            // four 16-byte write-through-coherent loads per lane and look ({lo, tag, hi, tag} per value), no sleep between
            // looks: the group this wavefront waits for paces the whole step
            pp_u32x4 g[4];
            unsigned it = 0;
            pp_u64 t0 = 0;
            for (;;) {
                bool ok = true;
                if (poll_last) {
                    // look at the granule the producer stores LAST only, all four once that one is there (a quarter of the poll
                    // traffic: with twenty and more workgroups polling, the looks themselves slow the stream down)
                    g[3] = __builtin_amdgcn_raw_buffer_load_b128(rs_strm, (unsigned)(lane * 64 + 48), (unsigned)((s * 16 + cc) * 4096), 16);
                    ok = g[3].y == tag && g[3].w == tag;
                    if (__all(ok)) {
#pragma unroll
                        for (int k = 0; k < 3; ++k) {
                            g[k] = __builtin_amdgcn_raw_buffer_load_b128(rs_strm, (unsigned)(lane * 64 + k * 16), (unsigned)((s * 16 + cc) * 4096), 16);
                            ok = ok && g[k].y == tag && g[k].w == tag;
                        }
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        g[k] = __builtin_amdgcn_raw_buffer_load_b128(rs_strm, (unsigned)(lane * 64 + k * 16), (unsigned)((s * 16 + cc) * 4096), 16);
                        ok = ok && g[k].y == tag && g[k].w == tag;
                    }
                }
                if (__all(ok)) break;
                if ((++it & 31u) == 0) {
                    if (pp_ld(ctl + PP_CTL_ABORT) == call_id) { dead = true; return; }
                    const pp_u64 now = __builtin_amdgcn_s_memrealtime();
                    if (t0 == 0) t0 = now;
                    else if (now - t0 > timeout) { pp_st(ctl + PP_CTL_ABORT, call_id); dead = true; return; }
                }
            }
            double* dst = Bs + (cc >> 2) * PP_CHUNK + lane * 18 + 4 * (cc & 3);
            *(f64x2*)dst = (f64x2){__hiloint2double((int)g[0].z, (int)g[0].x), __hiloint2double((int)g[1].z, (int)g[1].x)};
            *(f64x2*)(dst + 2) = (f64x2){__hiloint2double((int)g[2].z, (int)g[2].x), __hiloint2double((int)g[3].z, (int)g[3].x)};
            lds_store_volatile(bprog, cc + 1);
            if (cc == 0) PP_STAMP(s, 10);
            if (cc == 14) PP_STAMP(s, 15);
        });
        PP_STAMP(s, 3);
    }
    else if (!dead && p.producer) {
        // this workgroup's rows are the step's B operand: every group wavefront 1 has solved goes out as granules
        const __amdgpu_buffer_rsrc_t rs_strm = __builtin_amdgcn_make_buffer_rsrc((void*)q->strm, 0, (int)(PP_STRM_WORDS * 8), 0x00020000);
        static_for<PB / CB>([&](auto cc_) {
            constexpr int cc = decltype(cc_)::value;
            pp_lds_wait_ge(p.cnt + 2, cc + 1, p.abl);
            const double* src = p.As_cur + (cc >> 2) * PP_CHUNK + lane * 18 + 4 * (cc & 3);
            const f64x2 a0 = *(const f64x2*)src, a1 = *(const f64x2*)(src + 2);
            const double xs[4] = {a0.x, a0.y, a1.x, a1.y};
#pragma unroll
            for (int k = 0; k < CB; ++k) {
                const pp_u32x4 g = {(unsigned)__double2loint(xs[k]), tag, (unsigned)__double2hiint(xs[k]), tag};
                PP_STORE16(g, rs_strm, (unsigned)(lane * 64 + k * 16), (unsigned)((s * 16 + cc) * 4096), 16);
            }
            if (cc == 15) PP_STAMPP(s, 22);
        });
    }
    if (dead) {
        // release whoever waits for this wavefront; everybody leaves at the next step's head
        lds_store_volatile(p.abl, 1);
        lds_store_volatile(zflag, 1);
    }
    lds_store_volatile(bprog, PB);                             // (>= 16: also "this wavefront is done with As_prev")
    PP_STEP_LOOP_END()
}


// ---------------- wavefront 2: the factorisation's helper for the first eight column groups; then -- idle otherwise -- it
// watches the flags of the two tiles the matrix wavefronts subtract their products from (tile (r, s+1) and the next
// diagonal block, finished by the update workgroups during this step), fetches their values and stages them in LDS
// exactly where the hand-over puts the differences: the tile in the free parity of As as soon as wavefront 3 is done
// with it, the diagonal block -- held in registers meanwhile -- in Ls once the factorisation and the solving wavefront
// have left it.  The matrix wavefronts then touch no memory at all (with the requests in their own stream -- sixteen
// cache lines per accumulator-layout load -- the k-steps stood still for 2.5-3 us and were three groups behind when
// the last group arrived).  The staging is a function of its own, called once per step: inlined beside
// panel_helper_wave its 128 registers of tile values made the helper's loop spill, and the factorisation waits for
// the helper; the call's register saves fall into idle time.
__device__ PP_NOINLINE void pp_stage_tiles(unsigned lds_off, PpKarg karg, int s_arg) {
    pp_args_ptr q = pp_args(karg);
    double* lds = pp_lds_base(lds_off);
    const int lane = threadIdx.x & 63;
    const int r = (int)blockIdx.x;
    const int s = __builtin_amdgcn_readfirstlane(s_arg);
    const long long n = q->a.n, lda = q->a.lda;
    pp_u64* ctl = q->ctl;
    const pp_u64 call_id = q->call_id, timeout = q->timeout;
    int* ints = (int*)(lds + PP_INTS);
    int* cnt = ints + 8 * (s & 1);
    const long long base_n = (long long)(s + 1) * PB;
    bool up = s + 1 < 2;
    if (!up) {
        const pp_u64* f0 = ctl + PP_CTL_TILEFINAL + (long long)r * 64 + (s + 1);
        const pp_u64* f1 = ctl + PP_CTL_TILEFINAL + (long long)(s + 1) * 64 + (s + 1);
        PpSpin sp;
        for (;;) {
            const pp_u64 f = lane == 0 ? pp_ld(f0) : (lane == 1 ? pp_ld(f1) : call_id);
            if (__all(f == call_id)) { up = true; break; }
            if (pp_give_up(sp, ctl, call_id, timeout)) break;
        }
    }
    PP_STAMP(s, 17);
    if (!up) lds_store_volatile(ints + 17 + (s & 1), 1);       // (gave up: everybody leaves at the next step's head)
    else {
        const __amdgpu_buffer_rsrc_t rs_A = __builtin_amdgcn_make_buffer_rsrc((void*)q->a.A, 0, (int)(lda * n * 8), 0x00020000);
        // (address arithmetic from an opaque lane copy: step-invariant, it would be hoisted, spilled and reloaded between
        // the loads -- each scratch reload waits for every earlier load; loads unconditional, masks at the LDS writes)
        int ln = lane;
        asm volatile("" : "+v"(ln));
        // (32-bit index arithmetic throughout: the byte offsets fit the 2 GiB buffer descriptor, and 64-bit row / column
        // numbers beside the 128 registers of tile values made the diagonal block's writes spill -- each scratch reload
        // a memory round trip, 4 us for the block)
        const int ni = (int)n, ldai = (int)lda, bn = (int)base_n;
        f64x2 tv[32];
        if (r > s + 1) {
            double (*St)[PB + 2] = (double (*)[PB + 2])(lds + PP_AS + ((s & 1) ^ 1) * 4 * PP_CHUNK);
            const int ri0 = r * PB, nrow = ni - ri0;           // rows of the tile inside the matrix
#pragma unroll
            for (int it = 0; it < 32; ++it) {
                const int e = it * 64 + ln, rw = e >> 5, col = 2 * (e & 31);
                const int gr = rw < nrow ? ri0 + rw : ni - 1;
                tv[it] = __builtin_bit_cast(f64x2, __builtin_amdgcn_raw_buffer_load_b128(rs_A, (unsigned)(gr * ldai + bn + col) * 8u, 0, 16));
            }
            pp_lds_wait_ge(cnt + 6, 1, ints + 17 + (s & 1));                        // (wavefront 3 is done with L(r, s-1) in this parity of As)
#pragma unroll
            for (int it = 0; it < 32; ++it) {
                const int e = it * 64 + ln, rw = e >> 5, col = 2 * (e & 31);
                const f64x2 zero = {0.0, 0.0};
                *(f64x2*)(&St[rw][col]) = rw < nrow ? tv[it] : zero;
            }
            PP_STAMP(s, 18);
        }
        double (*Ls)[PB + 2] = (double (*)[PB + 2])(lds + PP_LS);
        const int nrem = ni - bn;                              // rows = columns of the next diagonal block inside the matrix
#pragma unroll
        for (int it = 0; it < 32; ++it) {
            const int e = it * 64 + ln, rw = e >> 5, col = 2 * (e & 31);
            const int gr = rw < nrem ? bn + rw : ni - 1;
            const int gc = col + 1 < nrem ? bn + col : (ni >= 2 ? ni - 2 : 0);   // (a pair past the matrix: one column to the left)
            tv[it] = __builtin_bit_cast(f64x2, __builtin_amdgcn_raw_buffer_load_b128(rs_A, (unsigned)(gr * ldai + gc) * 8u, 0, 16));
        }
        // the masks while the factorisation is still running (this wavefront shares its SIMD with a matrix wavefront: its
        // VALU instructions get a slot every ~32 cycles -- done after the wait they cost 2.5 us of critical path) ...
#pragma unroll
        for (int it = 0; it < 32; ++it) {
            const int e = it * 64 + ln, rw = e >> 5, col = 2 * (e & 31);
            const int lim = rw < nrem ? (rw < nrem - 1 ? rw : nrem - 1) : -1;   // last column kept in this row (lower triangle, inside the matrix)
            f64x2 x;
            x.x = col <= lim ? (col + 1 == nrem ? tv[it].y : tv[it].x) : 0.0;   // (col + 1 == nrem: the pair was loaded one column to the left)
            x.y = col + 1 <= lim ? tv[it].y : 0.0;
            tv[it] = x;
        }
        unsigned la[32];                                       // (... and the LDS addresses)
#pragma unroll
        for (int it = 0; it < 32; ++it) {
            const int e = it * 64 + ln, rw = e >> 5, col = 2 * (e & 31);
            la[it] = (unsigned)(size_t)(__attribute__((address_space(3))) double*)&Ls[rw][col];
        }
#pragma unroll
        for (int it = 0; it < 32; ++it) asm volatile("" : "+v"(tv[it]), "+v"(la[it]));
        // (prog = PB: the factorisation has published its last group and touches Ls no more in a workgroup that is not the
        // step's diagonal one -- not PB + 1: hipcc sinks the 64 pivot checks behind the last publish, 1.1 us before that)
        pp_lds_wait_ge(cnt + 0, PB, ints + 17 + (s & 1));
        pp_lds_wait_ge(cnt + 2, PB / CB, ints + 17 + (s & 1));                      // (... and the solving wavefront has read its last group of the factor)
        __builtin_amdgcn_s_setprio(3);
#pragma unroll
        for (int it = 0; it < 32; ++it)
            *(__attribute__((address_space(3))) f64x2*)(size_t)la[it] = tv[it];
        __builtin_amdgcn_s_setprio(0);
        PP_STAMP(s, 19);
    }
    lds_store_volatile(cnt + 5, up ? 1 : 2);                   // 1: the tiles' values are where the differences go | 2: gave up
    PP_WSTAMP(s);
    __syncthreads();                                           // the step's barrier (see pp_role_helper)
}

#include "potrf_persist_sg.h"

// karg_off: byte offset of `q` in the kernel-argument segment (the batched kernel passes one record per matrix)
__device__ __forceinline__ void pp_row_role(const PersistArgs& q, double* lds, const unsigned lds_off, const unsigned karg_off = 0) {
    const PotrfArgs& a = q.a;
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int r = (int)blockIdx.x;
    double (*Ls)[PB + 2] = (double (*)[PB + 2])(lds + PP_LS);
    int* ints = (int*)(lds + PP_INTS);
    const long long n = a.n, lda = a.lda;
    if (t < 64) ints[t] = 0;
    double rhs_r = 0.0;     // receiving wavefront: running right-hand side of the forward solve for matrix row r * 64 + lane
    if (w == 0) {
        // block (0, 0), coalesced (lane = column), transposed to lane = row through Ls (n > 64: the block is full)
        const double* src = a.A + lane;
        double tt[PB];
#pragma unroll
        for (int k = 0; k < PB; ++k) tt[k] = src[(long long)k * lda];
#pragma unroll
        for (int k = 0; k < PB; ++k) Ls[k][lane] = tt[k];
    } else if (w == PP_W_RECV) {
        const long long row = (long long)r * PB + lane;
        rhs_r = (a.rhs && row < n) ? a.rhs[row] : 0.0;
    }
    if (q.debug == 1 && r == 0 && t == 0) pp_st(q.ctl + PP_CTL_ABORT, q.call_id);
    __syncthreads();

    PpKarg karg;
    {
        const unsigned long long kp = (unsigned long long)(size_t)__builtin_amdgcn_kernarg_segment_ptr() + karg_off;
        karg.lo = (unsigned)kp; karg.hi = (unsigned)(kp >> 32);
    }
#ifdef PP_STAMPS
    if (r == q.nb - 1 && lane == 0) pp_stamps[63 * 24 + w] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_ID: SIMD in bits 5:4
#endif
    // (the LDS offset through an opaque register: passed as the constant it is, hipcc propagates the dynamic-LDS symbol into
    // the role functions, where every use becomes a lookup in its dynamic-LDS offset table -- an s_load + s_waitcnt
    // lgkmcnt(0) twice per 4-column group of the factorisation, behind the LDS stores just issued)
    unsigned lo = lds_off;
    asm volatile("" : "+s"(lo));
    // Roles by wavefront.  Wavefronts w and w + 4 share a SIMD (HW_ID: SIMD = {0, 2, 1, 3}[w & 3]), and a burst of MFMAs holds
    // that SIMD's double-precision pipe against the other wavefront's fp64 instructions whatever the priorities: the
    // factorisation loses ~500 cycles whenever a k-step's 32 MFMAs meet one of its column groups.  Other pairings were
    // measured (factorisation + receiver, solver + helper, matrix wavefronts with each other; receiver and helper swapped;
    // ...): the factorisation gets faster, the helper or the last k-step slower, the step stays within 2 %.
    if (w == 0) pp_role_factor(lo, karg);
    else if (w == 1) pp_role_solve(lo, karg);
    else if (w == 3) pp_role_recv(lo, karg, rhs_r);
    else if (w >= 4) pp_role_matrix(lo, karg);
    else pp_role_helper(lo, karg);
}

// ---------------------------------------------------------------------------
// update workgroup: block column s applied to the tiles (i, q), q >= s + 2, ONE tile at a time on all eight
// wavefronts (32 x 16 outputs each: a tile's products take 1.7 us of the four matrix pipes instead of 3.4 for two
// tiles side by side -- the tiles of column s + 2 are awaited by the row workgroups).  A tile's two 64 x 64 operands
// and its own values are requested in ONE round of loads, one tile AHEAD (registers -> the other LDS buffer while the
// current tile is multiplied: the memory latency of a tile hides behind its predecessor), staged whole in LDS (chunk
// layout) and multiplied with the k-step order of apgp_gemm64_tile, so the values are those of the multi-launch path.
// Tiles are owned statically: column-major index of (i, q), 2 <= q <= i < nb, modulo the number of update workgroups.
// ---------------------------------------------------------------------------
struct PpTile { long long ri, rk; int q; bool active; };
__device__ __forceinline__ PpTile pp_tile(long long idx, long long F, long long total, int s, int nb) {
    PpTile tl;
    tl.active = idx >= F && idx < total;
    long long ti = 0, tq = 0;
    if (tl.active) {
        long long rem = idx - F;
        tq = s + 2;
        while (rem >= nb - tq) { rem -= nb - tq; ++tq; }
        ti = tq + rem;
    }
    tl.ri = ti * PB; tl.rk = tq * PB; tl.q = (int)tq;
    return tl;
}

__device__ __forceinline__ void pp_update_role(const PersistArgs& q, double* lds) {
    const PotrfArgs& a = q.a;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int nb = q.nb;
    const int u = (int)blockIdx.x - nb, nupd = (int)gridDim.x - nb;
    int* ints = (int*)(lds + PP_UPD_INTS);
    const long long n = a.n, lda = a.lda;
    const pp_u64 base = q.call_id << 8;
    const int wr = (w >> 2) * 32, wc = (w & 3) * 16;          // this wavefront's 32 x 16 outputs
    int bcol[4];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) bcol[rr] = ((lane & 15) - 4 * rr) & 15;
    const long long total = (long long)(nb - 1) * (nb - 2) / 2;
    // operand rows: thread -> row e >> 5, columns 2 (e & 31), + 1 of each operand, four rounds
    // (a second register set / loads two tiles ahead measured slower: 0.85 vs 0.81 ms at n = 2048, 3.8 vs 2.2 at 4096)
    f64x2 ra[4], rb[4];
    double cin[2][4];
    const __amdgpu_buffer_rsrc_t rs_Au = __builtin_amdgcn_make_buffer_rsrc((void*)a.A, 0, (int)(lda * n * 8), 0x00020000);
    auto request = [&](const PpTile& tl, long long j0) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int e = it * 512 + t, rw = e >> 5, col = 2 * (e & 31);
            // (unconditional loads from clamped rows, masked when staged / used: see the matrix wavefronts' request)
            const long long ar = tl.ri + rw < n ? tl.ri + rw : n - 1, br = tl.rk + rw < n ? tl.rk + rw : n - 1;
            // (write-through-coherent loads, no acquire: an agent-scope acquire per update step and workgroup -- buffer_inv sc1
            // -- empties the XCD's L2 for everybody some thirty times per row step; with it and plain loads: 0.692 / 0.917 /
            // 1.18 ms at n = 2048 / 2560 / 3072 instead of 0.670 / 0.874 / 1.13)
            ra[it] = __builtin_bit_cast(f64x2, __builtin_amdgcn_raw_buffer_load_b128(rs_Au, (unsigned)((ar * lda + j0 + col) * 8), 0, 16));
            rb[it] = __builtin_bit_cast(f64x2, __builtin_amdgcn_raw_buffer_load_b128(rs_Au, (unsigned)((br * lda + j0 + col) * 8), 0, 16));
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const long long gr = tl.ri + wr + 16 * i + apgp_mma16_row(lane);
                const long long gc = tl.rk + wc + apgp_mma16_col(lane, rr);
                cin[i][rr] = pp_ld_f64(a.A + (gr < n ? gr : n - 1) * lda + (gc < n ? gc : n - 1));
            }
    };
    auto stage = [&](double* buf, const PpTile& tl) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int e = it * 512 + t, rw = e >> 5, col = 2 * (e & 31);
            const f64x2 zero = {0.0, 0.0};
            *(f64x2*)(buf + (col >> 4) * PP_CHUNK + rw * 18 + (col & 15)) = (tl.active && tl.ri + rw < n) ? ra[it] : zero;
            *(f64x2*)(buf + 4 * PP_CHUNK + (col >> 4) * PP_CHUNK + rw * 18 + (col & 15)) = (tl.active && tl.rk + rw < n) ? rb[it] : zero;
        }
    };
    for (int s = 0; s + 2 < nb; ++s) {
#ifdef PP_STAMPS
        const long long F_ = (long long)s * nb - ((long long)(s + 1) * (s + 2) / 2 - 1);
        const bool ustamp_wg = (F_ % nupd) == u, ustamp_me = ustamp_wg;
#endif
        // every block row below s has L(:, s) in memory
        if (w == 0) {
            const int i = s + 1 + lane;
            PpSpin sp;
            int ok = 1;
            for (;;) {
                const bool have = i >= nb || pp_ld(q.ctl + PP_CTL_ROWDONE + i) >= (base | (pp_u64)(s + 1));
                if (__all(have)) break;
                __builtin_amdgcn_s_sleep(16);       // (a whole row step of slack: poll gently -- 0.5-1 % for the row workgroups)
                if (pp_give_up(sp, q.ctl, q.call_id, q.timeout)) { ok = 0; break; }
            }
            PP_USTAMP(s, 0);
            if (lane == 0) {
                // (no acquire: the operands are read with write-through-coherent loads -- see request)
                ints[s & 1] = ok;
            }
        }
        __syncthreads();
        if (!ints[s & 1]) return;
        PP_USTAMP(s, 1);
        const long long j0 = (long long)s * PB;
        const long long F = (long long)s * nb - ((long long)(s + 1) * (s + 2) / 2 - 1);   // index of tile (s + 2, s + 2)
        long long m = F <= u ? 0 : (F - u + nupd - 1) / nupd;              // first m with u + nupd m >= F
        if (u + (long long)nupd * m >= total) continue;
        PpTile cur = pp_tile(u + (long long)nupd * m, F, total, s, nb);
        request(cur, j0);
        stage(lds, cur);
        double ccur[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) ccur[i][rr] = cin[i][rr];
        __syncthreads();
        bool first = true;
        for (int par = 0;; par ^= 1, ++m) {
            // the next tile's loads fly during this tile's products (requested unconditionally -- past the end: clamped,
            // unused -- because a load under a condition has to land at the join)
            const PpTile nxt = pp_tile(u + (long long)nupd * (m + 1), F, total, s, nb);
            request(nxt, j0);
            if (first) PP_USTAMP(s, 2);
            const double* Aop = lds + par * PP_UPD_HALF;
            const double* Bop = Aop + 4 * PP_CHUNK;
            double v[2][4];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) v[i][rr] = 0.0;
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const double* Ach = Aop + (g >> 2) * PP_CHUNK;
                const double* Bch = Bop + (g >> 2) * PP_CHUNK;
                const int ks = g & 3;
                double af[2], bf[4];
#pragma unroll
                for (int i = 0; i < 2; ++i) af[i] = Ach[(wr + 16 * i + (lane & 15)) * 18 + ks * 4 + (lane >> 4)];
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) bf[rr] = Bch[(wc + bcol[rr]) * 18 + ks * 4 + (lane >> 4)];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) v[i][rr] = __builtin_amdgcn_mfma_f64_4x4x4f64(af[i], bf[rr], v[i][rr], 0, 0, 0);
            }
            if (first) PP_USTAMP(s, 3);
            if (cur.active) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const long long gr = cur.ri + wr + 16 * i + apgp_mma16_row(lane);
                        const long long gc = cur.rk + wc + apgp_mma16_col(lane, rr);
                        if (gr < n && gc < n && gc <= gr) pp_st_f64(a.A + gr * lda + gc, ccur[i][rr] - v[i][rr]);   // (masked elements: never stored)
                    }
            }
            if (nxt.active) {
                stage(lds + (par ^ 1) * PP_UPD_HALF, nxt);               // (waits for the next tile's operands)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) ccur[i][rr] = cin[i][rr];
            }
            // column s + 2 is awaited by the row workgroups: its stores are drained (every wavefront, then the barrier) before
            // the flag.  The other tiles' stores only have to be in memory before this workgroup reads the tiles again, an
            // update step later: ONE drain at the end of the step (a write-through store's acknowledgement takes 1-2 us)
            const bool fin = cur.active && cur.q == s + 2;
            if (fin) pp_drain();
            __syncthreads();                                             // (the next operands staged; this buffer free)
            if (first) PP_USTAMP(s, 4);
            first = false;
            if (fin && t == 0) pp_st(q.ctl + PP_CTL_TILEFINAL + (cur.ri / PB) * 64 + cur.q, q.call_id);
            if (!nxt.active) break;
            cur = nxt;
        }
        pp_drain();
        PP_USTAMP(s, 5);
    }
}

__global__ __launch_bounds__(PP_THREADS) void potrf_persist_kernel(PersistArgs q) {
    extern __shared__ __attribute__((aligned(16))) double pp_lds[];
    if ((int)blockIdx.x < q.nb) pp_row_role(q, pp_lds, (unsigned)(size_t)(__attribute__((address_space(3))) double*)pp_lds);
    else pp_update_role(q, pp_lds);
}

// Round 6: up to PP_BATCH_MAX matrices of the SAME size factorised side by side by ONE launch (gridDim.y = batch;
// apgp_nll_eval_batch: the look-ahead points of a Powell line search).  One persistent factorisation is a latency chain
// on nb + a few dozen CUs; matrix y's workgroups (blockIdx.x as in the single launch) read THEIR argument record --
// own matrix, right-hand side, info word, scratch (factor blocks, flags, granule streams) -- from the kernel-argument
// segment and never look at another matrix: the code and the bits of `batch` single launches, at little more than the
// time of one.  The host sizes gridDim.x so that all workgroups of all matrices are resident at once.
#define PP_BATCH_MAX 6
struct PersistBatchArgs { PersistArgs m[PP_BATCH_MAX]; };
static_assert(sizeof(PersistBatchArgs) <= 4096, "kernel-argument segment");
__global__ __launch_bounds__(PP_THREADS) void potrf_persist_batch_kernel(PersistBatchArgs qb) {
    extern __shared__ __attribute__((aligned(16))) double pp_lds[];
    const unsigned y = blockIdx.y;
    const PersistArgs& q = qb.m[y];
    if ((int)blockIdx.x < q.nb)
        pp_row_role(q, pp_lds, (unsigned)(size_t)(__attribute__((address_space(3))) double*)pp_lds, y * (unsigned)sizeof(PersistArgs));
    else pp_update_role(q, pp_lds);
}
