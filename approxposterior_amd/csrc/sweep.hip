// K5/K6 "sweep": fused candidate tile -> k* row -> mu -> L^-1 k* contraction on
// the f64 matrix cores -> predictive variance -> AGP / BAPE / Jones utility ->
// arg-min.  The batched counterpart of
//   george.GP.predict(y, t, return_var=True)        (utility.py:131,178,224)
//   utility.AGPUtility / BAPEUtility / JonesUtility (utility.py:99-250)
//   utility.minimizeObjective's arg-min             (utility.py:369-371)
//
// Data flow per workgroup (256 threads = 4 wavefronts, 64 candidates):
//   * wavefront w owns candidates [16 w, 16 w + 16) (one MFMA column block);
//     the scaled candidate coordinates live in registers for the whole kernel.
//   * the packed factor W = L^-1 is streamed tile by tile (256 rows x 16 k,
//     32 KiB, A-fragment order) HBM/L2 -> LDS with global_load_lds (no VGPR
//     round trip), double buffered, shared by the four wavefronts.
//   * K* is NEVER materialised: each lane generates the k*(t_m, x_k) value it
//     must feed as the MFMA B operand (lane -> candidate lane&15, k lane>>4)
//     on the VALU (D sub + D fma + exp).
//   * V = W K*^T is accumulated 256 rows x 16 candidates per wavefront in 64
//     v_mfma_f64_4x4x4_4b_f64 accumulators; at the end of a row block the
//     squares are folded into a per-candidate sum; V itself is never stored.
//   * mu is a VALU by-product of the last row block (which visits every k).
#include "apgp_common.h"
#include <stdlib.h>
#include <type_traits>
#include <string.h>

#define SW_ROWS APGP_ROW_BLOCK       // 256 rows of W per tile
#define SW_KC APGP_K_CHUNK           // 16 k per tile
#define SW_TILE (SW_ROWS * SW_KC)    // doubles per tile (32 KiB)
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
    int ndim, nrb, kind, has_box;
    double mean, amp, log_amp, zeta, ybest;
    double sc[APGP_MAX_DIM], lo[APGP_MAX_DIM], hi[APGP_MAX_DIM];
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
    // NaN never wins; ties resolve to the lowest global index
    if (i >= 0 && (u < bu || (u == bu && (bi < 0 || i < bi)))) { bu = u; bi = i; }
}

// Matrix-core instruction choice (measured on MI355X, tools/mfma_peak.hip and
// tools/mfma_probe.hip):
//   v_mfma_f64_16x16x4_f64     36 TF/chip  (~138 cycles, 7.4 MAC/clk/SIMD)
//   v_mfma_f64_4x4x4_4b_f64    70-74 TF    (16.5 cycles, 15.5 MAC/clk/SIMD = FP64 FMA peak)
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
// many rows as possible: a wavefront owns 256 rows x 16 candidates (64 f64
// accumulators per lane), so one generated B fragment feeds 64 MFMAs.
template <int DPAD>
__global__ __launch_bounds__(SW_THREADS, 1) void sweep_kernel(SweepArgs a) {
    constexpr int XS = DPAD + 2;
    constexpr int RS = SW_ROWS / 16;           // 16-row sub-blocks per tile
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* Abuf = smem;                       // 2 x SW_TILE
    double* Xbuf = smem + 2 * SW_TILE;         // 2 x SW_KC x XS
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int cl = lane & 15, kq = lane >> 4;

    // ---- candidate of this lane -------------------------------------------
    double tt[DPAD];
    const long long crow = (long long)blockIdx.x * SW_CAND + w * 16 + cl;
    bool adm = crow < a.m;
    {
        const bool inb = adm;
#pragma unroll
        for (int d = 0; d < DPAD; ++d) {
            double v = 0.0;
            if (inb && d < a.ndim) {
                v = a.T[crow * a.ndim + d];
                if (a.has_box && !(v >= a.lo[d] && v <= a.hi[d])) adm = false;
            }
            tt[d] = v * a.sc[d];
        }
        if (inb && a.mask && a.mask[crow] == 0) adm = false;
    }

    // V accumulators: 256 rows x 16 candidates per wavefront
    double acc[RS][4];
#pragma unroll
    for (int s = 0; s < RS; ++s)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[s][r] = 0.0;
    double qpart = 0.0, mupart = 0.0;
    // lane index of the A element this lane feeds for rotation r
    int rot[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) rot[r] = (lane & ~12) | ((((lane >> 2) + r) & 3) << 2);

    constexpr int CPB = SW_ROWS / SW_KC;       // k-chunks per row-block width
    const long long ntiles = (long long)CPB * a.nrb * (a.nrb + 1) / 2;
    // stage_issue(tile, kc, buf): W tile via LDS-DMA (each wavefront moves 8 KiB
    // as 8 x 1 KiB, no VGPR round trip); the small training-stream chunk is
    // fetched into registers now and written to LDS by stage_commit() after the
    // tile's matrix work, so its HBM/L2 latency hides under the MFMAs.
    constexpr int XCHUNK16 = SW_KC * XS / 2;                    // 16-byte pieces per chunk
    constexpr int XNP = (XCHUNK16 + SW_THREADS - 1) / SW_THREADS;
    f64x2 xpend[XNP];
    auto stage_issue = [&](long long tile, int kc, int buf) {
        const char* g = (const char*)(a.linv + tile * SW_TILE) + w * 8192 + lane * 16;
        char* l = (char*)(Abuf + buf * SW_TILE) + w * 8192;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(g + i * 1024),
                (__attribute__((address_space(3))) void*)(l + i * 1024), 16, 0, 0);
#pragma unroll
        for (int p = 0; p < XNP; ++p) {
            const int e = t + p * SW_THREADS;
            if (e < XCHUNK16) xpend[p] = *((const f64x2*)(a.xs + (long long)kc * SW_KC * XS) + e);
        }
    };
    auto stage_commit = [&](int buf) {
#pragma unroll
        for (int p = 0; p < XNP; ++p) {
            const int e = t + p * SW_THREADS;
            if (e < XCHUNK16) *((f64x2*)(Xbuf + buf * SW_KC * XS) + e) = xpend[p];
        }
    };

    stage_issue(0, 0, 0);
    stage_commit(0);
    __builtin_amdgcn_s_waitcnt(0);   // vmcnt(0) lgkmcnt(0) expcnt(0)
    __syncthreads();

    long long tile = 0;
    int buf = 0;
    for (int ib = 0; ib < a.nrb; ++ib) {
        const int nkc = CPB * (ib + 1);
        const bool last_rb = (ib == a.nrb - 1);
        for (int kc = 0; kc < nkc; ++kc, ++tile) {
            // prefetch the next tile (possibly the first tile of the next row block)
            const bool more = tile + 1 < ntiles;
            if (more) stage_issue(tile + 1, (kc + 1 < nkc) ? kc + 1 : 0, buf ^ 1);
            const double* Ab = Abuf + buf * SW_TILE;
            const double* Xb = Xbuf + buf * SW_KC * XS;
            // W is lower triangular: inside the diagonal row block the 16-row
            // sub-blocks above the chunk's k range are structurally zero and are
            // skipped (DIAG path); off-diagonal tiles run the branch-free path.
            const int smin = (kc - CPB * ib) * (SW_KC / 16);   // <= 0 off the diagonal
            auto tile_body = [&](auto diag_tag) {
                constexpr bool DIAG = decltype(diag_tag)::value;
                constexpr int NKK = SW_KC / 4, NG = NKK * RS;
                // ---- generate the B operands of the tile's k-steps:
                //      k*(candidate cl, x_k), k = kc*KC + 4 kk + kq
                double bfv[NKK];
#pragma unroll
                for (int kk = 0; kk < NKK; ++kk) {
                    const double* xr = Xb + (kk * 4 + kq) * XS;
                    double s2 = -a.log_amp;
#pragma unroll
                    for (int d = 0; d < DPAD; ++d) {
                        const double df = tt[d] - xr[d];
                        s2 = fma(df, df, s2);
                    }
                    bfv[kk] = exp(-s2);
                    if (last_rb) mupart = fma(bfv[kk], xr[DPAD], mupart);
                }
                // ---- A fragments (W rows, four rotations each) software-pipelined two
                //      groups ahead of the matrix-core steps that consume them
                double afr[3][4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    afr[0][r] = Ab[0 * 64 + rot[r]];
                    afr[1][r] = Ab[1 * 64 + rot[r]];
                }
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    if (g + 2 < NG) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) afr[(g + 2) % 3][r] = Ab[(g + 2) * 64 + rot[r]];
                    }
                    const int kk = g / RS, sb = g % RS;
                    if (!DIAG || sb >= smin) {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            acc[sb][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(afr[g % 3][r], bfv[kk], acc[sb][r], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            if (smin > 0) tile_body(std::true_type{});
            else tile_body(std::false_type{});
            if (kc == nkc - 1) {
                // row block finished: fold ||V||^2 into the per-candidate sum
#pragma unroll
                for (int s = 0; s < RS; ++s)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        qpart = fma(acc[s][r], acc[s][r], qpart);
                        acc[s][r] = 0.0;
                    }
            }
            if (more) stage_commit(buf ^ 1);
            __builtin_amdgcn_s_waitcnt(0);
            __syncthreads();
            buf ^= 1;
        }
    }

    // ---- reduce over the four k-quarters / row-quarters of the wavefront ------
    qpart += __shfl_xor(qpart, 16);
    qpart += __shfl_xor(qpart, 32);
    mupart += __shfl_xor(mupart, 16);
    mupart += __shfl_xor(mupart, 32);
    double bu = INFINITY;
    long long bi = -1;
    if (kq == 0 && crow < a.m) {
        const double mu = mupart + a.mean;
        const double var = a.amp - qpart;
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
    const size_t lds = (2 * SW_TILE + 2 * SW_KC * (DPAD + 2)) * sizeof(double);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)sweep_kernel<DPAD>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    const unsigned nblk = (unsigned)((a.m + SW_CAND - 1) / SW_CAND);
    hipLaunchKernelGGL(sweep_kernel<DPAD>, dim3(nblk), dim3(SW_THREADS), lds, s, a);
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
    a.T = T; a.linv = packed_linv; a.xs = xs; a.mask = mask;
    a.mu = mu; a.var = var; a.u = u;
    const long long nblk = (m + SW_CAND - 1) / SW_CAND;
    a.part_u = (double*)part;
    a.part_i = part ? (long long*)((double*)part + nblk) : NULL;
    a.m = m; a.idx_offset = idx_offset;
    a.ndim = kc.ndim; a.nrb = (int)(apgp_npad(n) / APGP_ROW_BLOCK); a.kind = kind;
    a.has_box = lo != NULL;
    a.mean = mean; a.amp = kc.amp; a.log_amp = kc.log_amp; a.zeta = zeta; a.ybest = ybest;
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
// mean-only prediction: the batched ApproxPosterior._gpll (approx.py:178-180).
// One wavefront per candidate; lanes stride over the training points.
// ---------------------------------------------------------------------------
struct MeanArgs {
    const double* T;
    const double* xs;
    double* mu;
    long long m, npad;
    int ndim;
    double mean, log_amp;
    double sc[APGP_MAX_DIM];
};

template <int DPAD>
__global__ __launch_bounds__(256) void predict_mean_kernel(MeanArgs a) {
    constexpr int XS = DPAD + 2;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long long row = (long long)blockIdx.x * 4 + w;
    if (row >= a.m) return;
    double tt[DPAD];
#pragma unroll
    for (int d = 0; d < DPAD; ++d) tt[d] = d < a.ndim ? a.T[row * a.ndim + d] * a.sc[d] : 0.0;
    double acc = 0.0;
    for (long long k = lane; k < a.npad; k += 64) {
        const double* xr = a.xs + k * XS;
        double s = -a.log_amp;
#pragma unroll
        for (int d = 0; d < DPAD; ++d) {
            double df = tt[d] - xr[d];
            s = fma(df, df, s);
        }
        acc = fma(exp(-s), xr[DPAD], acc);
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) a.mu[row] = acc + a.mean;
}

extern "C" int apgp_predict_mean(const double* T, int64_t m, const double* xs, int64_t n,
                                 const apgp_kernel_t* kern, double mean, double* mu, void* stream) {
    APGP_CHECK_ARG(T && xs && kern && mu, "null pointer");
    APGP_CHECK_ARG(m >= 1 && n >= 1, "m >= 1 and n >= 1 required");
    KernConst kc;
    APGP_CHECK_ARG(apgp_make_kernconst(kern, &kc) == 0, "kernel parameters");
    MeanArgs a;
    a.T = T; a.xs = xs; a.mu = mu; a.m = m; a.npad = apgp_npad(n); a.ndim = kc.ndim;
    a.mean = mean; a.log_amp = kc.log_amp;
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
