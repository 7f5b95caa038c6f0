// Round 5: the 64-column step of the persistent Cholesky in SUPER-GROUPS of 16 columns (included by potrf_persist.h).
//
// Round 4's step was bound by the register-resident factorisation of the 64 x 64 diagonal block (12.9-15.6 us of a 19 us
// step) and the triangular solve of the workgroup's 64 tile rows behind it: one row per lane, every 4-column group applied
// to ALL columns to its right with uniform-address ds_read_b128 -- 480 column updates per wavefront and step at ~40 cycles
// each, three quarters of them wasted on the LDS return path (a broadcast read still moves 1 KB to deliver 16 bytes).
// tools/probes/mfma_fma_chain.hip (profiles/r05a_mfma_fma_chain_probe.txt) shows that v_mfma_f64_4x4x4_4b IS the
// ascending chain of four IEEE FMAs, c -> fma(a0, b0, c) -> ... -> fma(a3, b3, .), bit for bit: a rank-4 update done on
// the matrix cores carries exactly the bits of panel_trailing's VALU chain row[j] = fma(-x[k], l[j][k], row[j]).
// So the row-per-lane wavefronts now work through 16 columns at a time (16 registers of row instead of 64; updates only
// inside the super-group: 24 column updates per super-group instead of 120 on average), and everything a finished
// super-group owes to the columns right of it is applied by the four matrix wavefronts -- which used to idle between
// their k-steps -- to accumulators they hold from the step's start (the staged values of super-groups 1 .. 3 of the
// diagonal block and of the tile, in the 16 x 16 MFMA layout, one rotation per wavefront):
//   * at the end of super-group h - 1 (boundary h) its 16 columns are applied to every later super-group's accumulators,
//     super-group h first; those tiles are handed back at once -- the diagonal block's in place in Ls, the tile's through
//     a 64 x 16 transfer buffer -- and the factorising / solving wavefront goes on with super-group h;
//   * element (i, j) still sees a_ij, then fma(-l_ik, l_jk, .) for k = 0, 1, 2, ... in ascending order, then its
//     reciprocal pivot: the launch-per-step path's arithmetic (potrf.hip panel_factor_wave / panel_solve_wave, which the
//     other kernels keep), the same bits -- tests/test_gpu_parity.py compares factor, z, record and info word.
// The matrix wavefronts are event-driven (catch-up of the diagonal block > catch-up of the tile > k-step of the next
// step's products): none of their waits blocks another.  The helper wavefront no longer shares the factorisation
// (its columns 32 .. 63 are the matrix cores' now); it stages the next step's tiles as before.
#pragma once

// four flag words written by the four matrix wavefronts, read with one ds_read_b128
__device__ __forceinline__ void pp_lds_wait4_ge(const int* p4, int need, int* trip) {
    unsigned guard_ = 0;
    const bool first = (threadIdx.x & 63) == 0;
    for (;;) {
        // (ONE lane looks: a uniform-address read by 64 lanes still moves 1 KB over the LDS return path -- with the four
        // matrix wavefronts polling like that the factorisation's own LDS traffic ran at half speed)
        pp_u32x4 v = {0u, 0u, 0u, 0u};
        if (first) asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned)(size_t)(__attribute__((address_space(3))) const int*)p4) : "memory");
        const int lo = __builtin_amdgcn_readfirstlane(min(min((int)v.x, (int)v.y), min((int)v.z, (int)v.w)));
        if (lo >= need) break;
        __builtin_amdgcn_s_sleep(1);
        if (++guard_ > (1u << 18)) { lds_store_volatile(trip, 1); break; }
    }
    PANEL_FENCE();
}

#define PP_SG 16                                  // columns per super-group
#define PP_NSG (PB / PP_SG)

// ---------------- wavefront 0: the diagonal block (s, s), 16 columns at a time; at step r also z_r ----------------
__device__ PP_NOINLINE void pp_role_factor(unsigned lds_off, PpKarg karg) {
    pp_args_ptr q = pp_args(karg);
    double* lds = pp_lds_base(lds_off);
    const int lane = threadIdx.x & 63;
    const long long n = q->a.n;
    __builtin_amdgcn_s_setprio(3);         // (shares its SIMD with a matrix wavefront: the critical path goes first)
    PP_STEP_LOOP_BEGIN(lds)
    const PpStep p = pp_step(lds, s_in, n);
    {
        int* nx = (int*)(lds + PP_INTS) + 8 * ((s_in + 1) & 1);   // the next step's counters (nobody uses them before the barrier)
        int* nx2 = (int*)(lds + PP_INTS) + 24 + 16 * ((s_in + 1) & 1);
        if (lane < 8) nx[lane] = 0;
        if (lane < 16) nx2[lane] = 0;
    }
    double (*Ls)[PB + 2] = (double (*)[PB + 2])(lds + PP_LS);
    double* invd = lds + PP_INVD;
    const double* zrow = lds + PP_ZROW;
    const int bs = p.bs, s = p.s, r = p.r;
    int* prog_p = p.cnt + 0;
    PP_STAMP(s, 0);
    PP_STAMPP(s, 20);
    int firstbad = 0x7fffffff;               // (uniform) first pivot that is not positive and finite
    // (the lane comparisons from an opaque copy of the lane number: see panel_factor_wave)
    int ln = lane;
    asm volatile("" : "+v"(ln));
    double xl[CB];                           // this row's entries of the group factorised last
    static_for<PP_NSG>([&](auto h_) {
        constexpr int h = decltype(h_)::value, b0 = PP_SG * h;
        // columns b0 .. b0 + 15 of row `lane`, updated with every column left of b0: the staged values (h = 0) or what
        // the matrix wavefronts have just written back (their boundary-h catch-up)
        if constexpr (h == 0) PP_FSTAMP(s, 0);
        if constexpr (h > 0) { PP_FSTAMP(s, 2 * h - 1); pp_lds_wait4_ge(p.cnt2 + 0, h, p.abl); PP_FSTAMP(s, 2 * h); }
        double ar[PP_SG];
#pragma unroll
        for (int k = 0; k < PP_SG; k += 2) {
            // (masks and addresses from the opaque lane copy: step-invariant, hipcc would hoist all 64 of them out of the step
            // loop, spill them to scratch and reload them here -- a memory round trip each, on the critical path)
            const f64x2 v = *(const f64x2*)(&Ls[ln][b0 + k]);
            ar[k] = (ln < bs && b0 + k <= ln) ? v.x : ((b0 + k == ln) ? 1.0 : 0.0);
            ar[k + 1] = (ln < bs && b0 + k + 1 <= ln) ? v.y : ((b0 + k + 1 == ln) ? 1.0 : 0.0);
        }
        PANEL_FENCE();
        // what came back carries the columns < b0 - 4; the last group of the previous super-group -- factorised while the
        // matrix wavefronts were at it -- is applied here, from this wavefront's own registers (k ascending: same chain)
        if constexpr (h > 0) panel_trailing<b0 - CB, b0, b0 + PP_SG, b0, PP_SG>(ar, xl, Ls);
        static_for<PP_SG / CB>([&](auto cc_) {
            constexpr int l0 = CB * decltype(cc_)::value, c0 = b0 + l0;
            // (1) .. (4): panel_factor_wave's group, operation for operation
            double d[CB][CB];
#pragma unroll
            for (int rr = 0; rr < CB; ++rr)
#pragma unroll
                for (int qq = 0; qq <= rr; ++qq) d[rr][qq] = bcast_lane(ar[l0 + qq], c0 + rr);
            double inv[CB], sq[CB];
#pragma unroll
            for (int k = 0; k < CB; ++k) {
                const double pv = d[k][k];
                const bool bad = !(pv > 0.0) || !(pv < INFINITY);
                firstbad = (bad && firstbad == 0x7fffffff) ? c0 + k + 1 : firstbad;
                double rq = __builtin_amdgcn_rsq(pv);
                {
                    const double e = fma(-pv * rq, rq, 1.0);
                    rq = fma(rq * e, fma(0.375, e, 0.5), rq);
                }
                inv[k] = rq;
#pragma unroll
                for (int i = k + 1; i < CB; ++i) d[i][k] *= rq;
#pragma unroll
                for (int j = k + 1; j < CB; ++j)
#pragma unroll
                    for (int i = j; i < CB; ++i) d[i][j] = fma(-d[i][k], d[j][k], d[i][j]);
                double dd = pv * rq;                                  // (off the chain) sqrt(p), one Newton step
                sq[k] = fma(0.5 * rq, fma(-dd, dd, pv), dd);
            }
            double x[CB];
#pragma unroll
            for (int k = 0; k < CB; ++k) {
                double sacc = ar[l0 + k];
#pragma unroll
                for (int m = 0; m < k; ++m) sacc = fma(-x[m], d[k][m], sacc);
                sacc *= inv[k];
                x[k] = ln == c0 + k ? sq[k] : (ln > c0 + k ? sacc : 0.0);
                ar[l0 + k] = x[k];
            }
#pragma unroll
            for (int k = 0; k < CB; k += 2) *(f64x2*)(&Ls[ln][c0 + k]) = (f64x2){x[k], x[k + 1]};
            if (ln == 0) {
#pragma unroll
                for (int k = 0; k < CB; k += 2) *(f64x2*)(&invd[c0 + k]) = (f64x2){inv[k], inv[k + 1]};
                lds_store_volatile(prog_p, c0 + CB);     // (same wavefront: LDS stores stay in order)
            }
            PANEL_FENCE();
            // (5) the rank-4 update of the columns right of the group INSIDE the super-group (the others: matrix cores)
            panel_trailing<c0, c0 + CB, b0 + PP_SG, b0, PP_SG>(ar, x, Ls);
            if constexpr (l0 == PP_SG - CB) {
#pragma unroll
                for (int k = 0; k < CB; ++k) xl[k] = x[k];
            }
        });
    });
    PP_STAMP(s, 1);
    PP_FSTAMP(s, 7);
    if (firstbad != 0x7fffffff && r == s && lane == 0)
        atomicMin((unsigned int*)q->a.info, (unsigned int)(q->a.info_j0 + p.j0 + firstbad));
    lds_store_volatile(prog_p, PB + 1);
    if (r == s) {
        // L_ss out, coalesced, from the published copy, into the scratch blocks (potrf_finish_kernel moves them into place)
        double* dst = q->a.dscr + (p.j0 / PB) * (PB * PB) + lane;
        for (int rr = 0; rr < bs; ++rr)
            if (lane <= rr) dst[(long long)rr * PB] = Ls[rr][lane];
    }
    if (r == s && q->a.rhs) {
        // z_s = L_ss^-1 (rhs block s): the operations of the pass that rides along in panel_factor_wave, in its
        // order, on the finished factor (Ls, invd) -- the 4 x 4 blocks of Ls ARE its d[][] bit for bit
        pp_lds_wait_ge(p.cnt + 4, 1, p.abl);
        double ri = lane < bs ? zrow[lane] : 0.0;
        static_for<PB / CB>([&](auto cc_) {
            constexpr int c0 = CB * decltype(cc_)::value;
            double d[CB][CB], zb[CB], inv[CB], x[CB];
#pragma unroll
            for (int rr = 0; rr < CB; ++rr) {
#pragma unroll
                for (int qq = 0; qq < rr; ++qq) d[rr][qq] = Ls[c0 + rr][c0 + qq];
                zb[rr] = bcast_lane(ri, c0 + rr);
                inv[rr] = invd[c0 + rr];
                x[rr] = Ls[lane][c0 + rr];
            }
#pragma unroll
            for (int k = 0; k < CB; ++k) {
                double zacc = zb[k];
#pragma unroll
                for (int m = 0; m < k; ++m) zacc = fma(-zb[m], d[k][m], zacc);
                zb[k] = zacc * inv[k];
            }
            double racc = ri;
#pragma unroll
            for (int k = 0; k < CB; ++k) racc = fma(-x[k], zb[k], racc);
            ri = lane >= c0 + CB ? racc : ri;
#pragma unroll
            for (int k = 0; k < CB; ++k) ri = lane == c0 + k ? zb[k] : ri;
        });
        if (lane < bs) (q->a.dscr + q->a.zoff)[p.j0 + lane] = ri;
        if (s + 1 < q->nb) {
            const unsigned tag = (unsigned)q->call_id;
            const pp_u32x4 g = {(unsigned)__double2loint(ri), tag, (unsigned)__double2hiint(ri), tag};
            const __amdgpu_buffer_rsrc_t rs_z = __builtin_amdgcn_make_buffer_rsrc((void*)q->zstrm, 0, (int)(PP_ZSTRM_WORDS * 8), 0x00020000);
            PP_STORE16(g, rs_z, (unsigned)(lane * 16), (unsigned)(s * 1024), 16);
        }
    }
    PP_STEP_LOOP_END()
}

// ---------------- wavefront 1: the 64 rows of tile (r, s), 16 columns at a time ----------------
__device__ PP_NOINLINE void pp_role_solve(unsigned lds_off, PpKarg karg) {
    pp_args_ptr q = pp_args(karg);
    double* lds = pp_lds_base(lds_off);
    const int lane = threadIdx.x & 63;
    const long long n = q->a.n, lda = q->a.lda;
    __builtin_amdgcn_s_setprio(3);
    PP_STEP_LOOP_BEGIN(lds)
    const PpStep p = pp_step(lds, s_in, n);
    if (p.s >= 1) {
        // L(r, s-1) went to memory at the end of the previous step (below): drained by now -- publish it
        pp_drain();
        if (lane == 0) pp_st(q->ctl + PP_CTL_ROWDONE + p.r, (q->call_id << 8) | (pp_u64)p.s);
        PP_STAMP(p.s, 8);
    }
    if (p.r > p.s) {
    const double (*Ls)[PB + 2] = (const double (*)[PB + 2])(lds + PP_LS);
    const double* invd = lds + PP_INVD;
    const double* xfer = lds + PP_XFER;
    const long long row = (long long)p.r * PB + lane;
    const bool has_row = row < n;
    double xl[CB];                                                 // this row's entries of the group solved last
    static_for<PP_NSG>([&](auto h_) {
        constexpr int h = decltype(h_)::value, b0 = PP_SG * h;
        double x16[PP_SG];                                         // columns b0 .. b0 + 15 of row `lane` of tile (r, s)
        if constexpr (h == 0) {
            if (p.s == 0) {
                const double* src = q->a.A + (has_row ? row : 0) * lda;
#pragma unroll
                for (int k = 0; k < PP_SG; ++k) x16[k] = has_row ? src[k] : 0.0;
            } else {
                const double (*St)[PB + 2] = (const double (*)[PB + 2])p.As_cur;
#pragma unroll
                for (int k = 0; k < PP_SG; k += 2) {
                    const f64x2 v = *(const f64x2*)(&St[lane][k]);
                    x16[k] = has_row ? v.x : 0.0;
                    x16[k + 1] = has_row ? v.y : 0.0;
                }
            }
            PANEL_FENCE();
            // the solved groups go where the staged tile lies (chunk layout over [64][66]): not before the matrix
            // wavefronts hold the staged values of super-groups 1 .. 3 in their accumulators
            pp_lds_wait4_ge(p.cnt2 + 8, 1, p.abl);
            PP_GSTAMP(p.s, 0);
        } else {
            PP_GSTAMP(p.s, 2 * h - 1);
            pp_lds_wait4_ge(p.cnt2 + 4, h, p.abl);                 // boundary h: the tile's catch-up is in the transfer buffer
            PP_GSTAMP(p.s, 2 * h);
#pragma unroll
            for (int k = 0; k < PP_SG; k += 2) {
                const f64x2 v = *(const f64x2*)(xfer + lane * 18 + k);
                x16[k] = has_row ? v.x : 0.0;
                x16[k + 1] = has_row ? v.y : 0.0;
            }
            PANEL_FENCE();
            // (the transfer buffer carries the columns < b0 - 4: the group solved last is applied here, from registers)
            panel_trailing<b0 - CB, b0, b0 + PP_SG, b0, PP_SG>(x16, xl, Ls);
        }
        static_for<PP_SG / CB>([&](auto cc_) {
            constexpr int cc = decltype(cc_)::value, l0 = CB * cc, c0 = b0 + l0;
            pp_lds_wait_ge(p.cnt + 0, c0 + CB, p.abl);             // columns c0 .. c0 + CB - 1 of L_ss published
            f64x2 dq[CB][CB / 2], iq[CB / 2];
#pragma unroll
            for (int rr = 0; rr < CB; ++rr)
#pragma unroll
                for (int qq = 0; 2 * qq < rr; ++qq) dq[rr][qq] = *(const f64x2*)(&Ls[c0 + rr][c0 + 2 * qq]);
#pragma unroll
            for (int qq = 0; qq < CB / 2; ++qq) iq[qq] = *(const f64x2*)(&invd[c0 + 2 * qq]);
            PANEL_FENCE();
            double xs[CB];
#pragma unroll
            for (int k = 0; k < CB; ++k) {
                double sacc = x16[l0 + k];
#pragma unroll
                for (int m = 0; m < k; ++m) sacc = fma(-xs[m], (m & 1) ? dq[k][m >> 1].y : dq[k][m >> 1].x, sacc);
                xs[k] = sacc * ((k & 1) ? iq[k >> 1].y : iq[k >> 1].x);
                x16[l0 + k] = xs[k];
            }
            {
                double* dst = p.As_cur + h * PP_CHUNK + lane * 18 + l0;
                *(f64x2*)dst = (f64x2){xs[0], xs[1]};
                *(f64x2*)(dst + 2) = (f64x2){xs[2], xs[3]};
                lds_store_volatile(p.cnt + 2, 4 * h + cc + 1);     // (same wavefront: LDS stores stay in order)
            }
            PANEL_FENCE();
            panel_trailing<c0, c0 + CB, b0 + PP_SG, b0, PP_SG>(x16, xs, Ls);
            if constexpr (l0 == PP_SG - CB) {
#pragma unroll
                for (int k = 0; k < CB; ++k) xl[k] = xs[k];
            }
        });
    });
    PP_STAMP(p.s, 2);
    PP_GSTAMP(p.s, 7);
    PP_STAMPP(p.s, 21);
    {
        // L(r, s) from LDS (this wavefront's published groups) to memory, coalesced write-through stores: this wavefront
        // is idle until the step ends, and the update workgroups need the rows as early as possible.  The flag follows at
        // the head of the next step, once the stores have drained.
        const int r = p.r, s = p.s;
        const __amdgpu_buffer_rsrc_t rs_A = __builtin_amdgcn_make_buffer_rsrc((void*)q->a.A, 0, (int)(q->a.lda * n * 8), 0x00020000);
        const unsigned row_b = (unsigned)(lda * 8);
        const unsigned off0 = (unsigned)((((long long)r * PB + (lane >> 5)) * lda + (long long)s * PB + 2 * (lane & 31)) * 8);
        const double* src0 = p.As_cur + ((lane & 31) >> 3) * PP_CHUNK + (lane >> 5) * 18 + (2 * (lane & 31) & 15);
#pragma unroll
        for (int b8 = 0; b8 < 4; ++b8) {
            f64x2 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = *(const f64x2*)(src0 + (b8 * 8 + k) * 2 * 18);
            PANEL_FENCE();
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const pp_u32x4 w4 = __builtin_bit_cast(pp_u32x4, v[k]);
                PP_STORE16(w4, rs_A, off0 + (unsigned)(b8 * 8 + k) * 2u * row_b, 0, 16);
            }
            PANEL_FENCE();
        }
    }
    }
    PP_STEP_LOOP_END()
}

// ---------------- wavefronts 4-7: catch-ups of this step's super-groups; the next tile and the next diagonal block ----------------
// accumulator tiles of the diagonal block (16 x 16 tiles (ti, tj), 1 <= tj <= ti <= 3): index
__device__ __forceinline__ constexpr int pp_dt(int ti, int tj) { return tj == 1 ? ti - 1 : (tj == 2 ? ti + 1 : 5); }

// Event H (1 .. 3) of the diagonal block, triggered when the first THREE groups of super-group H - 1 are out (the fourth is
// still being factorised): the published columns k in [K0, K1) = [16 (H - 1) - 4 (0 for H = 1), 16 H - 4) -- the last group
// of super-group H - 2 and the first three of H - 1 -- go into every accumulator tile of the super-groups >= H, those of
// super-group H first, which are handed back in place (Ls) at once.  What is then still missing in them, the columns
// [16 H - 4, 16 H) of the group in flight, the factorising wavefront adds itself from its registers when that group is done
// (16 column updates by broadcast reads, its own instruction stream: no hand-off on the critical path).
template <int H>
__device__ __forceinline__ void pp_catch_D(double (*Ls)[PB + 2], const int lane, const int rot, const int er, const int ec,
                                           double (&accD)[6], int* flag) {
    constexpr int K0 = H == 1 ? 0 : PP_SG * (H - 1) - CB, K1 = PP_SG * H - CB, NK = (K1 - K0) / CB;
    double af[4][NK], bf[NK];                                      // af[ti][ks] = -L[16 ti + (lane & 15)][K0 + 4 ks + (lane >> 4)]
#pragma unroll
    for (int ti = H; ti < 4; ++ti)
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) af[ti][ks] = -Ls[16 * ti + (lane & 15)][K0 + 4 * ks + (lane >> 4)];
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) bf[ks] = Ls[16 * H + rot][K0 + 4 * ks + (lane >> 4)];
#pragma unroll
    for (int ti = H; ti < 4; ++ti)
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) accD[pp_dt(ti, H)] = __builtin_amdgcn_mfma_f64_4x4x4f64(af[ti][ks], bf[ks], accD[pp_dt(ti, H)], 0, 0, 0);
#pragma unroll
    for (int ti = H; ti < 4; ++ti) Ls[16 * ti + er][16 * H + ec] = accD[pp_dt(ti, H)];
    lds_store_volatile(flag, H);                                   // (same wavefront: LDS stores stay in order)
    PANEL_FENCE();
#pragma unroll
    for (int tj = H + 1; tj < 4; ++tj) {
        double bl[NK];
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) bl[ks] = Ls[16 * tj + rot][K0 + 4 * ks + (lane >> 4)];
#pragma unroll
        for (int ti = tj; ti < 4; ++ti)
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) accD[pp_dt(ti, tj)] = __builtin_amdgcn_mfma_f64_4x4x4f64(af[ti][ks], bl[ks], accD[pp_dt(ti, tj)], 0, 0, 0);
    }
}

// the same for the tile: solved columns [K0, K1) (chunk layout of As) applied to the tile's super-groups >= H; super-group H
// goes to the solver through the transfer buffer, which adds the columns of its group in flight itself
template <int H>
__device__ __forceinline__ void pp_catch_T(const double (*Ls)[PB + 2], const double* As_cur, double* xfer, const int lane, const int rot,
                                           const int er, const int ec, double (&accT)[12], int* flag) {
    constexpr int K0 = H == 1 ? 0 : PP_SG * (H - 1) - CB, K1 = PP_SG * H - CB, NK = (K1 - K0) / CB;
    double af[4][NK], bf[NK];
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
            const int kc = K0 + 4 * ks;                            // first column of the group: chunk kc / 16, position kc % 16
            af[ti][ks] = -As_cur[(kc >> 4) * PP_CHUNK + (16 * ti + (lane & 15)) * 18 + (kc & 15) + (lane >> 4)];
        }
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) bf[ks] = Ls[16 * H + rot][K0 + 4 * ks + (lane >> 4)];
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) accT[(H - 1) * 4 + ti] = __builtin_amdgcn_mfma_f64_4x4x4f64(af[ti][ks], bf[ks], accT[(H - 1) * 4 + ti], 0, 0, 0);
#pragma unroll
    for (int ti = 0; ti < 4; ++ti) xfer[(16 * ti + er) * 18 + ec] = accT[(H - 1) * 4 + ti];
    lds_store_volatile(flag, H);
    PANEL_FENCE();
#pragma unroll
    for (int tj = H + 1; tj < 4; ++tj) {
        double bl[NK];
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) bl[ks] = Ls[16 * tj + rot][K0 + 4 * ks + (lane >> 4)];
#pragma unroll
        for (int ti = 0; ti < 4; ++ti)
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) accT[(tj - 1) * 4 + ti] = __builtin_amdgcn_mfma_f64_4x4x4f64(af[ti][ks], bl[ks], accT[(tj - 1) * 4 + ti], 0, 0, 0);
    }
}

// pp_kstep in two halves (same operands, same order of products): the operand reads of up to two k-steps are issued together
struct PpFragK { double af[2], bf[2][4], bfa[2]; };
template <bool TWO>
__device__ __forceinline__ void pp_load_K(PpFragK& f, const double* Ach, const double* Bch, const int ks, const int lane, const int wr, const int wc,
                                          const int (&bcol)[4]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) f.af[i] = Ach[(wr + 16 * i + (lane & 15)) * 18 + ks * 4 + (lane >> 4)];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < 2; ++j) f.bf[j][r] = Bch[(wc + 16 * j + bcol[r]) * 18 + ks * 4 + (lane >> 4)];
    if constexpr (TWO) {
#pragma unroll
        for (int i = 0; i < 2; ++i) f.bfa[i] = Bch[(wr + 16 * i + (lane & 15)) * 18 + ks * 4 + (lane >> 4)];
    }
}
// DIAG: which 16 x 16 blocks of the quadrant's share of the NEXT DIAGONAL block are computed -- only its lower triangle is
// ever read: 0 = none (the quadrant above the diagonal), 1 = blocks (0,0), (1,0), (1,1) (a quadrant on the diagonal),
// 2 = all four (the quadrant below it)
// (TILE: the same selection for the first product -- 2 for a tile; the step's producer workgroup multiplies its own rows with
// themselves there, which IS its diagonal product)
template <bool TWO, int DIAG, int TILE = 2>
__device__ __forceinline__ void pp_mfma_K(const PpFragK& f, double (&acc)[2][2][4], double (&acc2)[2][2][4]) {
    if constexpr (TILE > 0) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (TILE == 1 && i == 0 && j == 1) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(f.af[i], f.bf[j][r], acc[i][j][r], 0, 0, 0);
            }
    }
    if constexpr (TWO && DIAG > 0) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (DIAG == 1 && i == 0 && j == 1) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) acc2[i][j][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(f.bfa[i], f.bf[j][r], acc2[i][j][r], 0, 0, 0);
            }
    }
}

__device__ PP_NOINLINE void pp_role_matrix(unsigned lds_off, PpKarg karg) {
    pp_args_ptr q = pp_args(karg);
    double* lds = pp_lds_base(lds_off);
    const int t = threadIdx.x, lane = t & 63;
    const long long n = q->a.n;
    PP_STEP_LOOP_BEGIN(lds)
    const PpStep p = pp_step(lds, s_in, n);
    const int s = p.s, r = p.r;
    double (*Ls)[PB + 2] = (double (*)[PB + 2])(lds + PP_LS);
    const double* Bs = lds + PP_BS;
    double* xfer = lds + PP_XFER;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    // wavefront 4 shares its SIMD with the factorising wavefront: it takes the quadrant ABOVE the diagonal (wr = 0, wc = 32),
    // whose share of the next diagonal block is never read and is skipped -- half the MFMAs of the others per k-step
    const int mw = (wv - 4) ^ 1;
    const int wr = (mw >> 1) * 32, wc = (mw & 1) * 32;         // this wavefront's 32 x 32 quadrant of both products
    bool dead = false;
    {
    int bcol[4];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) bcol[rr] = ((lane & 15) - 4 * rr) & 15;
    // this wavefront's element of every 16 x 16 catch-up tile: rotation mw of the MFMA layout
    const int rot = ((lane & 15) - 4 * mw) & 15;               // row of the B fragment under rotation mw
    const int er = apgp_mma16_row(lane), ec = apgp_mma16_col(lane, mw);
    double accD[6], accT[12];
#pragma unroll
    for (int tj = 1; tj < 4; ++tj)
#pragma unroll
        for (int ti = tj; ti < 4; ++ti) accD[pp_dt(ti, tj)] = Ls[16 * ti + er][16 * tj + ec];
    const bool tile = r > s;
    if (tile) {
        if (s == 0) {
            // (first step: the tile is still in memory -- rows past the matrix read as the last row and are masked by the solver)
            const long long lda = q->a.lda;
#pragma unroll
            for (int tj = 1; tj < 4; ++tj)
#pragma unroll
                for (int ti = 0; ti < 4; ++ti) {
                    const long long gr = (long long)r * PB + 16 * ti + er;
                    const double v = q->a.A[(gr < n ? gr : n - 1) * lda + 16 * tj + ec];
                    accT[(tj - 1) * 4 + ti] = gr < n ? v : 0.0;
                }
        } else {
            const double (*St)[PB + 2] = (const double (*)[PB + 2])p.As_cur;
#pragma unroll
            for (int tj = 1; tj < 4; ++tj)
#pragma unroll
                for (int ti = 0; ti < 4; ++ti) accT[(tj - 1) * 4 + ti] = St[16 * ti + er][16 * tj + ec];
        }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) asm volatile("" : "+v"(accD[i]));
    if (tile) {
#pragma unroll
        for (int i = 0; i < 12; ++i) asm volatile("" : "+v"(accT[i]));
    }
    PANEL_FENCE();
    lds_store_volatile(p.cnt2 + 8 + mw, 1);                    // the staged values are in registers: the solver may publish
    double v[2][2][4], v2[2][2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) { v[i][j][rr] = 0.0; v2[i][j][rr] = 0.0; }
    const bool producer = p.producer;
#ifdef PP_STAMPS
#define PP_KSTAMPS() do { if (mw == 0) { if (g == 0) PP_STAMP(s, 11); if (g == 5) PP_STAMP(s, 9); if (g == 12) PP_STAMP(s, 12); if (g == 14) PP_STAMP(s, 14); if (g == 15) PP_STAMP(s, 4); } } while (0)
#else
#define PP_KSTAMPS() do { } while (0)
#endif
    // event loop: whatever is ready, the critical path first -- event of the diagonal block (the factorisation is one group
    // away from needing the tiles), event of the tile, then up to TWO k = 4 steps of the next step's products with their
    // operand reads issued together (a k-step alone is 0.2 us of MFMAs behind an LDS round trip of about the same)
    int nD = 1, nT = tile ? 1 : 4, g = tile ? 0 : PB / CB;
    unsigned guard = 0;
    int nev = 0; (void)nev;
    while (nD < 4 || nT < 4 || g < PB / CB) {
        pp_u32x4 c = {0u, 0u, 0u, 0u};
        if (lane == 0) asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(c) : "v"((unsigned)(size_t)(__attribute__((address_space(3))) const int*)p.cnt) : "memory");
        const int prog = __builtin_amdgcn_readfirstlane((int)c.x), xprog = __builtin_amdgcn_readfirstlane((int)c.z);
        const int bprog = producer ? PB : __builtin_amdgcn_readfirstlane((int)c.w);
        if (nD < 4 && prog >= PP_SG * nD - CB) {
            PP_ESTAMP_BEGIN(1000 * nD);
            if (nD == 1) pp_catch_D<1>(Ls, lane, rot, er, ec, accD, p.cnt2 + 0 + mw);
            else if (nD == 2) pp_catch_D<2>(Ls, lane, rot, er, ec, accD, p.cnt2 + 0 + mw);
            else pp_catch_D<3>(Ls, lane, rot, er, ec, accD, p.cnt2 + 0 + mw);
            PP_ESTAMP_END();
            ++nD; guard = 0;
            continue;
        }
        if (nT < 4 && xprog >= 4 * nT - 1) {
            PP_ESTAMP_BEGIN(100 * nT);
            if (nT == 1) pp_catch_T<1>(Ls, p.As_cur, xfer, lane, rot, er, ec, accT, p.cnt2 + 4 + mw);
            else if (nT == 2) pp_catch_T<2>(Ls, p.As_cur, xfer, lane, rot, er, ec, accT, p.cnt2 + 4 + mw);
            else pp_catch_T<3>(Ls, p.As_cur, xfer, lane, rot, er, ec, accT, p.cnt2 + 4 + mw);
            PP_ESTAMP_END();
            ++nT; guard = 0;
            continue;
        }
        const int nk = min(min(xprog, bprog), PB / CB);
        if (g < nk) {
            PP_ESTAMP_BEGIN(g + 1);
            // (up to two k-steps per look, operand reads together.  A rolled, software-pipelined batch of up to four was
            // measured: the fragment copy between iterations made the loop spill -- 0.49 ms at n = 1152 against 0.336)
            const bool two = g + 1 < nk;
            PpFragK f0, f1;
            if (producer) {
                pp_load_K<false>(f0, p.As_cur + (g >> 2) * PP_CHUNK, p.As_cur + (g >> 2) * PP_CHUNK, g & 3, lane, wr, wc, bcol);
                if (two) pp_load_K<false>(f1, p.As_cur + ((g + 1) >> 2) * PP_CHUNK, p.As_cur + ((g + 1) >> 2) * PP_CHUNK, (g + 1) & 3, lane, wr, wc, bcol);
                // (its own rows with themselves: the quadrant above the diagonal owes nothing, the two on it three blocks of four)
                if (wr < wc) { }
                else if (wr == wc) { pp_mfma_K<false, 0, 1>(f0, v, v2); if (two) pp_mfma_K<false, 0, 1>(f1, v, v2); }
                else { pp_mfma_K<false, 0, 2>(f0, v, v2); if (two) pp_mfma_K<false, 0, 2>(f1, v, v2); }
            } else {
                pp_load_K<true>(f0, p.As_cur + (g >> 2) * PP_CHUNK, Bs + (g >> 2) * PP_CHUNK, g & 3, lane, wr, wc, bcol);
                if (two) pp_load_K<true>(f1, p.As_cur + ((g + 1) >> 2) * PP_CHUNK, Bs + ((g + 1) >> 2) * PP_CHUNK, (g + 1) & 3, lane, wr, wc, bcol);
                // (the quadrant above the diagonal owes the next diagonal block nothing, the two on it three blocks of four)
                if (wr < wc) { pp_mfma_K<true, 0>(f0, v, v2); if (two) pp_mfma_K<true, 0>(f1, v, v2); }
                else if (wr == wc) { pp_mfma_K<true, 1>(f0, v, v2); if (two) pp_mfma_K<true, 1>(f1, v, v2); }
                else { pp_mfma_K<true, 2>(f0, v, v2); if (two) pp_mfma_K<true, 2>(f1, v, v2); }
            }
            g += two ? 2 : 1;
            PP_ESTAMP_END();
            guard = 0;
            continue;
        }
        __builtin_amdgcn_s_sleep(1);
        if (++guard > (1u << 18)) { lds_store_volatile(p.abl, 1); dead = true; break; }
    }
    if (tile) {
    // hand-over: the staged values minus the products, in place -- the next tile in the free parity of As ([64][66]) for
    // wavefront 1, the next diagonal block in Ls (staged there by wavefront 2)
    pp_lds_wait_ge(p.cnt + 5, 1, p.abl);
    if (lds_load_volatile(p.cnt + 5) != 1) dead = true;
    if (mw == 0) PP_STAMP(s, 5);
    double (*St)[PB + 2] = (double (*)[PB + 2])p.As_prev;
    {
        // all the reads first, then the differences, then the stores (St and Ls may alias as far as the compiler knows)
        double sv[2][2][4], lv[2][2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int lr = wr + 16 * i + apgp_mma16_row(lane), lc = wc + 16 * j + apgp_mma16_col(lane, rr);
                    lv[i][j][rr] = Ls[lr][lc];
                    sv[i][j][rr] = St[lr][lc];       // (the producer's own tile is not in St: read, unused)
                }
        PANEL_FENCE();
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int lr = wr + 16 * i + apgp_mma16_row(lane), lc = wc + 16 * j + apgp_mma16_col(lane, rr);
                    if (producer) Ls[lr][lc] = lv[i][j][rr] - v[i][j][rr];
                    else {
                        St[lr][lc] = sv[i][j][rr] - v[i][j][rr];
                        Ls[lr][lc] = lv[i][j][rr] - v2[i][j][rr];
                    }
                }
    }
    if (mw == 0) PP_STAMP(s, 6);
    }
    if (dead) lds_store_volatile(p.abl, 1);
    }
    PP_STEP_LOOP_END()
}

// ---------------- wavefront 2: stages the next step's two tiles (pp_stage_tiles) ----------------
__device__ PP_NOINLINE void pp_role_helper(unsigned lds_off, PpKarg karg) {
    pp_args_ptr q = pp_args(karg);
    double* lds = pp_lds_base(lds_off);
    PP_STEP_LOOP_BEGIN(lds)
    PP_STAMP(s_in, 16);
    // (the step's barrier is the last thing pp_stage_tiles does -- it is called exactly when the step has one)
    if (r_ > s_in) pp_stage_tiles(lds_off, karg, s_in);
    }
    PP_ROLE_EXIT(lds, q);
}
