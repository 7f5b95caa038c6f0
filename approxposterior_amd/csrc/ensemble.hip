// On-device affine-invariant ensemble sampler over the GP-mean surrogate: the
// whole MCMC loop of ApproxPosterior.runMCMC (approx.py:839-846) -- emcee's
// stretch move (Goodman & Weare 2010; red/blue split, a = 2) calling
// ApproxPosterior._gpll (approx.py:148-189: log-probability = GP predictive
// mean, -inf outside the prior) once per walker per half-step -- as ONE
// persistent kernel.  The chain is sequential (every half-step depends on the
// previous one) and a half-step is only W/2 mean-only predictions, so on a GPU
// the reference's structure is pure launch/sync latency (~0.1 ms per half-step
// from the host, 4e4 half-steps for the README example); here proposal, RNG
// (Philox4x32-10), prior gate, GP mean and accept test all stay on the device.
// One workgroup = one independent ensemble (grid.x ensembles = replicas, the
// only way this path shards); a wavefront evaluates one
// walker's GP mean with its lanes striding the training points.
// Only a box prior can be evaluated on the device (arbitrary Python priors
// cannot): the caller asserts lnprior == const inside [lo, hi], -inf outside.
#include "apgp_common.h"
#include "scratch.h"
#include <atomic>
#include <mutex>

struct EnsArgs {
    const double* xs;      // packed training stream: Npad x (DPAD+2): scaled x | alpha | 0
    double* coords;        // E x W x D  (in: initial state, out: final state)
    double* logp;          // E x W      (out: final log-probability)
    double* chain;         // iterations x E x W x D or NULL
    double* logp_chain;    // iterations x E x W or NULL
    long long* naccept;    // E x W
    long long n, iterations;
    int ndim, nwalkers, lin_order;
    unsigned long long seed;
    double mean, amp, a_stretch, lin_coef;
    double sc[APGP_MAX_DIM], lo[APGP_MAX_DIM], hi[APGP_MAX_DIM], lw[APGP_MAX_DIM];
};

__device__ __forceinline__ void philox4x32(unsigned int (&c)[4], unsigned int k0, unsigned int k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c[0];
        const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c[2];
        const unsigned int n0 = (unsigned int)(p1 >> 32) ^ c[1] ^ k0;
        const unsigned int n1 = (unsigned int)p1;
        const unsigned int n2 = (unsigned int)(p0 >> 32) ^ c[3] ^ k1;
        const unsigned int n3 = (unsigned int)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}

__device__ __forceinline__ double u01(unsigned int a, unsigned int b) {
    // 53-bit uniform in (0, 1)
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6) + 0.5) * (1.0 / 9007199254740992.0);
}

#define ENS_MAXW 256

// XLDS: the packed training stream (Npad x (DPAD+2) doubles) is staged into LDS once
// (it is re-read by every walker of every half-step); falls back to L2 when it does
// not fit beside the sampler state.
template <int DPAD, bool XLDS>
__global__ __launch_bounds__(1024) void ensemble_kernel(EnsArgs a) {
    constexpr int XS = DPAD + 2;
    extern __shared__ __attribute__((aligned(16))) double xsl[];
    __shared__ double etab[APGP_EXP_TAB_N];
    __shared__ double cs[ENS_MAXW][DPAD];      // scaled walker coordinates
    __shared__ double lp[ENS_MAXW];
    __shared__ double qs[ENS_MAXW / 2][DPAD];  // scaled proposals of the active half
    __shared__ double lpq[ENS_MAXW / 2];
    __shared__ double fac[ENS_MAXW / 2];       // (D-1) log z
    __shared__ double uacc[ENS_MAXW / 2];
    __shared__ int qok[ENS_MAXW / 2];
    __shared__ int nacc[ENS_MAXW];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int W = a.nwalkers, H = W / 2, D = a.ndim;
    const long long ens = blockIdx.x;
    apgp_exp_tab_load(etab);
    double* gc = a.coords + ens * W * D;
    for (int e = t; e < W * DPAD; e += 1024) {
        const int w = e / DPAD, d = e % DPAD;
        cs[w][d] = d < D ? gc[w * D + d] * a.sc[d] : 0.0;
    }
    if (t < W) nacc[t] = 0;
    if (XLDS) {
        for (long long e = t; e < a.n * XS; e += 1024) xsl[e] = a.xs[e];
    }
    __syncthreads();

    // GP means of TWO points (scaled coordinates p0 / p1[0..DPAD)) by one wavefront: every training
    // row read from LDS serves both (the stream is 80 B per row and is read by all 16 wavefronts of
    // every half-step: with one point per pass the LDS, not the arithmetic, set the pace).  Each
    // point's sum runs in the same order as a single-point pass.
    auto gp_mean2 = [&](const double* p0, const double* p1, double& m0, double& m1) {
        double t0[DPAD], t1[DPAD];
#pragma unroll
        for (int d = 0; d < DPAD; ++d) { t0[d] = p0[d]; t1[d] = p1[d]; }
        double acc0 = 0.0, acc1 = 0.0;
        const int nn = (int)a.n;
        for (int k = lane; k < nn; k += 64) {
            const double* xr = (XLDS ? (const double*)xsl : a.xs) + k * XS;
            double xv[DPAD];
#pragma unroll
            for (int d = 0; d < DPAD; ++d) xv[d] = xr[d];
            const double al = xr[DPAD];
            double s0 = 0.0, s03 = 0.0, s1 = 0.0, s13 = 0.0;
#pragma unroll
            for (int d = 0; d < DPAD; d += 2) {
                const double a0 = t0[d] - xv[d], a1 = t0[d + 1] - xv[d + 1];
                const double b0 = t1[d] - xv[d], b1 = t1[d + 1] - xv[d + 1];
                s0 = fma(a0, a0, s0); s03 = fma(a1, a1, s03);
                s1 = fma(b0, b0, s1); s13 = fma(b1, b1, s13);
            }
            double kv0 = a.amp * apgp_exp(-(s0 + s03), etab);
            double kv1 = a.amp * apgp_exp(-(s1 + s13), etab);
            if (a.lin_coef != 0.0) {
                double ls;
                APGP_LIN_SUM(ls, DPAD, a.ndim, a.lin_order, t0[d_] * xv[d_] * a.lw[d_]);
                kv0 = fma(a.lin_coef, ls, kv0);
                APGP_LIN_SUM(ls, DPAD, a.ndim, a.lin_order, t1[d_] * xv[d_] * a.lw[d_]);
                kv1 = fma(a.lin_coef, ls, kv1);
            }
            acc0 = fma(kv0, al, acc0);
            acc1 = fma(kv1, al, acc1);
        }
        for (int o = 32; o > 0; o >>= 1) { acc0 += __shfl_xor(acc0, o); acc1 += __shfl_xor(acc1, o); }
        m0 = acc0 + a.mean;
        m1 = acc1 + a.mean;
    };
    // one point per pass (start-up; and D > 8, where two points' registers do not fit 16 wavefronts)
    auto gp_mean = [&](const double* p) {
        double tt[DPAD];
#pragma unroll
        for (int d = 0; d < DPAD; ++d) tt[d] = p[d];
        double acc = 0.0;
        const int nn = (int)a.n;
        for (int k = lane; k < nn; k += 64) {
            const double* xr = (XLDS ? (const double*)xsl : a.xs) + k * XS;
            double s = 0.0, s3 = 0.0;
#pragma unroll
            for (int d = 0; d < DPAD; d += 2) {
                const double df0 = tt[d] - xr[d], df1 = tt[d + 1] - xr[d + 1];
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
        return acc + a.mean;
    };

    for (int w = wv; w < W; w += 16) {
        double m = gp_mean(cs[w]);
        // walkers that start outside the prior have zero probability (as _gpll returns -inf)
        for (int d = 0; d < D; ++d)
            if (!(cs[w][d] >= a.lo[d] && cs[w][d] <= a.hi[d])) m = -INFINITY;
        if (lane == 0) lp[w] = m;
    }
    __syncthreads();

    const unsigned int k0 = (unsigned int)a.seed, k1 = (unsigned int)(a.seed >> 32) ^ (unsigned int)(ens * 0x9E3779B9u);
    for (long long it = 0; it < a.iterations; ++it) {
        // random cyclic offset of the red/blue partition for this iteration
        unsigned int cr[4] = {(unsigned int)it, (unsigned int)(it >> 32), 0xFFFFFFFFu, 0x5u};
        philox4x32(cr, k0, k1);
        const int rot = (int)(cr[0] % (unsigned int)W);
        for (int split = 0; split < 2; ++split) {
            // walker index of slot i of the active half S and of the complement C
            auto s_idx = [&](int i) { int v = i + split * H + rot; return v >= W ? v - W : v; };
            auto c_idx = [&](int i) { int v = i + (1 - split) * H + rot; return v >= W ? v - W : v; };
            if (t < H) {
                unsigned int c1[4] = {(unsigned int)it, (unsigned int)(it >> 32), (unsigned int)(split * ENS_MAXW + t), 0x1u};
                philox4x32(c1, k0, k1);
                const double u = u01(c1[0], c1[1]);
                const double z = ((a.a_stretch - 1.0) * u + 1.0);
                const double zz = z * z / a.a_stretch;
                const int j = c_idx((int)(c1[2] % (unsigned int)H));
                const int s = s_idx(t);
                bool ok = true;
#pragma unroll
                for (int d = 0; d < DPAD; ++d) {
                    const double q = cs[j][d] - (cs[j][d] - cs[s][d]) * zz;
                    qs[t][d] = q;
                    // prior gate in scaled coordinates (lo/hi were scaled on the host)
                    if (d < D && !(q >= a.lo[d] && q <= a.hi[d])) ok = false;
                }
                qok[t] = ok ? 1 : 0;
                fac[t] = (D - 1.0) * log(zz);
                unsigned int c2[4] = {(unsigned int)it, (unsigned int)(it >> 32), (unsigned int)(split * ENS_MAXW + t), 0x2u};
                philox4x32(c2, k0, k1);
                uacc[t] = u01(c2[0], c2[1]);
            }
            __syncthreads();
            if constexpr (DPAD > 8) {
                for (int i = wv; i < H; i += 16) {
                    double m = -INFINITY;
                    if (qok[i]) m = gp_mean(qs[i]);         // wave-uniform branch
                    if (lane == 0) lpq[i] = m;
                }
            } else
            for (int i = wv; i < H; i += 32) {
                // proposals i and i + 16 of this wavefront in one pass over the training stream
                const int i1 = i + 16;
                const bool ok0 = qok[i] != 0, ok1 = i1 < H && qok[i1] != 0;     // wave-uniform
                double m0 = -INFINITY, m1 = -INFINITY;
                if (ok0 && ok1) gp_mean2(qs[i], qs[i1], m0, m1);
                else if (ok0) m0 = gp_mean(qs[i]);
                else if (ok1) m1 = gp_mean(qs[i1]);
                if (lane == 0) {
                    lpq[i] = m0;
                    if (i1 < H) lpq[i1] = m1;
                }
            }
            __syncthreads();
            if (t < H) {
                const int s = s_idx(t);
                const double diff = fac[t] + lpq[t] - lp[s];
                if (qok[t] && lpq[t] == lpq[t] && log(uacc[t]) < diff) {
#pragma unroll
                    for (int d = 0; d < DPAD; ++d) cs[s][d] = qs[t][d];
                    lp[s] = lpq[t];
                    nacc[s] += 1;
                }
            }
            __syncthreads();
        }
        if (a.chain) {
            double* out = a.chain + ((it * gridDim.x + ens) * W) * D;
            for (int e = t; e < W * D; e += 1024) {
                const int w = e / D, d = e % D;
                out[e] = cs[w][d] / a.sc[d];
            }
        }
        if (a.logp_chain && t < W) a.logp_chain[(it * gridDim.x + ens) * W + t] = lp[t];
    }
    for (int e = t; e < W * D; e += 1024) {
        const int w = e / D, d = e % D;
        gc[e] = cs[w][d] / a.sc[d];
    }
    if (t < W) {
        a.logp[ens * W + t] = lp[t];
        a.naccept[ens * W + t] = nacc[t];
    }
}

// ---------------------------------------------------------------------------
// Round 5: ONE ensemble over SEVERAL workgroups.  The single-workgroup kernel above keeps 255 of 256 compute units idle for
// a lone ensemble (BASELINE config 5: 64 walkers x 2e4 iterations at N = 1152 took 1.4 s, 35 us per half-step, all of it the
// 32 GP means of the half-step on one CU).  Here G workgroups (256 threads = 4 wavefronts each, one per SIMD) share an
// ensemble: every workgroup keeps the whole sampler state and draws the same proposals (the RNG is counter-based: pure
// functions of (seed, iteration, walker)), evaluates the GP mean of ITS proposals (g, g + G, ...; rows split over its four
// wavefronts, partial sums added in a fixed order), publishes them as data-tagged granules ({low word, tag, high word, tag},
// tag = half-step number + 1, write-through stores, two buffers by half-step parity) and reads everybody else's; the accept
// step then runs identically in every workgroup.  One exchange of H doubles per half-step, no flag, no fence.
// All G x E workgroups must be resident at once (the host limits G x E to the device's CUs); every poll is bounded: on
// timeout the workgroup writes NaN log-probabilities and leaves -- the host wrapper re-runs on the single-workgroup kernel.
// ---------------------------------------------------------------------------
struct EnsMwArgs {
    EnsArgs e;
    unsigned long long* xchg;      // [E][2][ENS_MAXW / 2] granules of 2 x 8 bytes
    unsigned long long* status;    // sticky "a workgroup gave up" word of this launch (zeroed with xchg)
    unsigned long long timeout;    // 100 MHz ticks
    int G;
};
typedef unsigned int ens_u32x4 __attribute__((ext_vector_type(4)));

template <int DPAD, bool XLDS>
__global__ __launch_bounds__(256) void ensemble_mw_kernel(EnsMwArgs q) {
    const EnsArgs& a = q.e;
    constexpr int XS = DPAD + 2;
    extern __shared__ __attribute__((aligned(16))) double xsl[];
    __shared__ double etab[APGP_EXP_TAB_N];
    __shared__ double cs[ENS_MAXW][DPAD];
    __shared__ double lp[ENS_MAXW];
    __shared__ double qs[ENS_MAXW / 2][DPAD];
    __shared__ double lpq[ENS_MAXW / 2];
    __shared__ double fac[ENS_MAXW / 2];
    __shared__ double uacc[ENS_MAXW / 2];
    __shared__ double part[4][2];              // partial sums of the (at most two) proposals in flight, per wavefront
    __shared__ int qok[ENS_MAXW / 2];
    __shared__ int nacc[ENS_MAXW];
    __shared__ int gaveup;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int W = a.nwalkers, H = W / 2, D = a.ndim, G = q.G;
    const long long ens = blockIdx.x / G;
    const int g = (int)(blockIdx.x % G);
    apgp_exp_tab_load(etab);
    double* gc = a.coords + ens * W * D;
    for (int e = t; e < W * DPAD; e += 256) {
        const int w = e / DPAD, d = e % DPAD;
        cs[w][d] = d < D ? gc[w * D + d] * a.sc[d] : 0.0;
    }
    if (t < W) nacc[t] = 0;
    if (t == 0) gaveup = 0;
    if (XLDS) {
        for (long long e = t; e < a.n * XS; e += 256) xsl[e] = a.xs[e];
    }
    __syncthreads();
    const int nn = (int)a.n;
    // GP means of one or two points by the WHOLE workgroup: row k goes to thread k mod 256 (wavefront (k / 64) mod 4), each
    // wavefront butterflies its lanes, the four partial sums are added 0 + 1 + 2 + 3
    auto gp_mean_wg = [&](const double* p0, const double* p1, const bool two, double& m0, double& m1) {
        double t0[DPAD], t1[DPAD];
#pragma unroll
        for (int d = 0; d < DPAD; ++d) { t0[d] = p0[d]; t1[d] = two ? p1[d] : p0[d]; }
        double acc0 = 0.0, acc1 = 0.0;
        for (int k = t; k < nn; k += 256) {
            const double* xr = (XLDS ? (const double*)xsl : a.xs) + k * XS;
            double xv[DPAD];
#pragma unroll
            for (int d = 0; d < DPAD; ++d) xv[d] = xr[d];
            const double al = xr[DPAD];
            double s0 = 0.0, s03 = 0.0, s1 = 0.0, s13 = 0.0;
#pragma unroll
            for (int d = 0; d < DPAD; d += 2) {
                const double a0 = t0[d] - xv[d], a1 = t0[d + 1] - xv[d + 1];
                const double b0 = t1[d] - xv[d], b1 = t1[d + 1] - xv[d + 1];
                s0 = fma(a0, a0, s0); s03 = fma(a1, a1, s03);
                s1 = fma(b0, b0, s1); s13 = fma(b1, b1, s13);
            }
            double kv0 = a.amp * apgp_exp(-(s0 + s03), etab);
            double kv1 = a.amp * apgp_exp(-(s1 + s13), etab);
            if (a.lin_coef != 0.0) {
                double ls;
                APGP_LIN_SUM(ls, DPAD, a.ndim, a.lin_order, t0[d_] * xv[d_] * a.lw[d_]);
                kv0 = fma(a.lin_coef, ls, kv0);
                APGP_LIN_SUM(ls, DPAD, a.ndim, a.lin_order, t1[d_] * xv[d_] * a.lw[d_]);
                kv1 = fma(a.lin_coef, ls, kv1);
            }
            acc0 = fma(kv0, al, acc0);
            acc1 = fma(kv1, al, acc1);
        }
        for (int o = 32; o > 0; o >>= 1) { acc0 += __shfl_xor(acc0, o); acc1 += __shfl_xor(acc1, o); }
        if (lane == 0) { part[wv][0] = acc0; part[wv][1] = acc1; }
        __syncthreads();
        m0 = ((part[0][0] + part[1][0]) + part[2][0]) + part[3][0] + a.mean;
        m1 = ((part[0][1] + part[1][1]) + part[2][1]) + part[3][1] + a.mean;
        __syncthreads();
    };
    // start-up: every workgroup evaluates every walker (once per launch; keeps the state replicated without an exchange)
    for (int w = 0; w < W; w += 2) {
        double m0, m1;
        gp_mean_wg(cs[w], cs[w + 1 < W ? w + 1 : w], w + 1 < W, m0, m1);
        if (t == 0) {
            for (int d = 0; d < D; ++d) {
                if (!(cs[w][d] >= a.lo[d] && cs[w][d] <= a.hi[d])) m0 = -INFINITY;
                if (w + 1 < W && !(cs[w + 1][d] >= a.lo[d] && cs[w + 1][d] <= a.hi[d])) m1 = -INFINITY;
            }
            lp[w] = m0;
            if (w + 1 < W) lp[w + 1] = m1;
        }
    }
    __syncthreads();

    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)(q.xchg + ens * 2 * (ENS_MAXW / 2) * 2), 0,
                                                                          2 * (ENS_MAXW / 2) * 16, 0x00020000);
    const unsigned int k0 = (unsigned int)a.seed, k1 = (unsigned int)(a.seed >> 32) ^ (unsigned int)(ens * 0x9E3779B9u);
    bool dead = false;
    for (long long it = 0; it < a.iterations && !dead; ++it) {
        unsigned int cr[4] = {(unsigned int)it, (unsigned int)(it >> 32), 0xFFFFFFFFu, 0x5u};
        philox4x32(cr, k0, k1);
        const int rot = (int)(cr[0] % (unsigned int)W);
        for (int split = 0; split < 2; ++split) {
            auto s_idx = [&](int i) { int v = i + split * H + rot; return v >= W ? v - W : v; };
            auto c_idx = [&](int i) { int v = i + (1 - split) * H + rot; return v >= W ? v - W : v; };
            if (t < H) {
                unsigned int c1[4] = {(unsigned int)it, (unsigned int)(it >> 32), (unsigned int)(split * ENS_MAXW + t), 0x1u};
                philox4x32(c1, k0, k1);
                const double u = u01(c1[0], c1[1]);
                const double z = ((a.a_stretch - 1.0) * u + 1.0);
                const double zz = z * z / a.a_stretch;
                const int j = c_idx((int)(c1[2] % (unsigned int)H));
                const int sw = s_idx(t);
                bool ok = true;
#pragma unroll
                for (int d = 0; d < DPAD; ++d) {
                    const double qv = cs[j][d] - (cs[j][d] - cs[sw][d]) * zz;
                    qs[t][d] = qv;
                    if (d < D && !(qv >= a.lo[d] && qv <= a.hi[d])) ok = false;
                }
                qok[t] = ok ? 1 : 0;
                fac[t] = (D - 1.0) * log(zz);
                unsigned int c2[4] = {(unsigned int)it, (unsigned int)(it >> 32), (unsigned int)(split * ENS_MAXW + t), 0x2u};
                philox4x32(c2, k0, k1);
                uacc[t] = u01(c2[0], c2[1]);
            }
            __syncthreads();
            // this workgroup's proposals g, g + G, ... (two per pass over the training stream), published as they are done
            const unsigned step = (unsigned)(2 * it + split);
            const unsigned tag = step + 1u;
            const unsigned par = step & 1u;
            for (int i = g; i < H; i += 2 * G) {
                const int i1 = i + G;
                const bool ok0 = qok[i] != 0, ok1 = i1 < H && qok[i1] != 0;
                double m0 = -INFINITY, m1 = -INFINITY;
                if (ok0 || ok1) {
                    double r0, r1;
                    gp_mean_wg(qs[ok0 ? i : i1], qs[ok1 ? i1 : i], ok0 && ok1, r0, r1);
                    if (ok0) m0 = r0;
                    if (ok1) m1 = ok0 ? r1 : r0;
                }
                if (t < 2 && (t == 0 || i1 < H)) {
                    const double mv = t == 0 ? m0 : m1;
                    const int slot = t == 0 ? i : i1;
                    const ens_u32x4 gr = {(unsigned)__double2loint(mv), tag, (unsigned)__double2hiint(mv), tag};
                    __builtin_amdgcn_raw_buffer_store_b128(gr, rs_x, (unsigned)((par * (ENS_MAXW / 2) + slot) * 16), 0, 16);
                    asm volatile("s_nop 1" : : "v"(gr));        // (16-byte store: its data registers stay live -- see PP_STORE16)
                }
            }
            // everybody's values (one granule per thread, polled until its tag is this half-step's)
            if (t < H) {
                ens_u32x4 gr;
                unsigned long long t0 = 0;
                unsigned itn = 0;
                for (;;) {
                    gr = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (unsigned)((par * (ENS_MAXW / 2) + t) * 16), 0, 16);
                    if (gr.y == tag && gr.w == tag) break;
                    if ((++itn & 63u) == 0) {
                        if (*(volatile int*)&gaveup) break;
                        const unsigned long long now = __builtin_amdgcn_s_memrealtime();
                        if (t0 == 0) t0 = now;
                        else if (now - t0 > q.timeout) { gaveup = 1; break; }
                    }
                }
                lpq[t] = __hiloint2double((int)gr.z, (int)gr.x);
            }
            __syncthreads();
            if (gaveup) { dead = true; break; }
            if (t < H) {
                const int sw = s_idx(t);
                const double diff = fac[t] + lpq[t] - lp[sw];
                if (qok[t] && lpq[t] == lpq[t] && log(uacc[t]) < diff) {
#pragma unroll
                    for (int d = 0; d < DPAD; ++d) cs[sw][d] = qs[t][d];
                    lp[sw] = lpq[t];
                    nacc[sw] += 1;
                }
            }
            __syncthreads();
        }
        if (dead) break;
        // the chain: every workgroup writes its share of the walkers
        const long long E_ = gridDim.x / G;
        if (a.chain) {
            double* out = a.chain + ((it * E_ + ens) * W) * D;
            for (int e = g * 256 + t; e < W * D; e += G * 256) {
                const int w = e / D, d = e % D;
                out[e] = cs[w][d] / a.sc[d];
            }
        }
        if (a.logp_chain && g == 0 && t < W) a.logp_chain[(it * E_ + ens) * W + t] = lp[t];
    }
    // A workgroup that gave up raises the launch's sticky word: its NaN marker in logp alone could be overwritten by the
    // ensemble's workgroup 0 finishing normally a moment later (it has every granule it needs) while part of the last chain
    // rows was never written; ens_mark_failed_kernel, behind this launch on the stream, turns the word
    // into NaN log-probabilities for every ensemble -- the marker the host looks at can no longer be lost.
    if (dead && t == 0) atomicOr(q.status, 1ull);
    if (g == 0 || dead) {
        for (int e = t; e < W * D; e += 256) {
            const int w = e / D, d = e % D;
            gc[e] = cs[w][d] / a.sc[d];
        }
        if (t < W) {
            a.logp[ens * W + t] = dead ? __longlong_as_double(0x7ff8000000000000ll) : lp[t];
            a.naccept[ens * W + t] = nacc[t];
        }
    }
}

__global__ __launch_bounds__(256) void ens_mark_failed_kernel(const unsigned long long* status, double* logp, int count) {
    if (*status == 0ull) return;
    for (int i = threadIdx.x; i < count; i += 256) logp[i] = __longlong_as_double(0x7ff8000000000000ll);
}

// 0 = several workgroups per ensemble where that helps and fits (default) | 1 = the single-workgroup kernel only.
// A test / profiling switch, not read from the environment.  Returns the previous value.
static std::atomic<int> g_ens_mode{0};
static thread_local int tl_ens_mode = -1;       // >= 0: this thread's current call overrides the switch (apgp_ensemble_sample_ex)
extern "C" int apgp_ensemble_mode(int mode) {
    if (mode < 0) return g_ens_mode.load();
    return g_ens_mode.exchange(mode ? 1 : 0);
}

extern "C" int apgp_ensemble_sample(const double* xs, int64_t n, const apgp_kernel_t* kern, double mean,
                                    const double* lo, const double* hi, int32_t nwalkers,
                                    int32_t nensembles, int64_t iterations, double a_stretch,
                                    uint64_t seed, double* coords, double* logp, double* chain,
                                    double* logp_chain, int64_t* naccept, void* stream) {
    APGP_CHECK_ARG(xs && kern && lo && hi && coords && logp && naccept, "null pointer");
    APGP_CHECK_ARG(n >= 1 && n <= APGP_MAX_N && iterations >= 0 && nensembles >= 1, "n, iterations, nensembles");
    KernConst kc;
    APGP_CHECK_ARG(apgp_make_kernconst(kern, &kc) == 0, "kernel parameters");
    APGP_CHECK_ARG(nwalkers >= 2 && nwalkers % 2 == 0 && nwalkers <= ENS_MAXW, "nwalkers must be even and <= 256");
    APGP_CHECK_ARG(nwalkers >= 2 * kc.ndim, "nwalkers must be at least twice the dimension");
    APGP_CHECK_ARG(a_stretch > 1.0, "stretch scale a must be > 1");
    for (int d = 0; d < kc.ndim; ++d) APGP_CHECK_ARG(kc.sc[d] > 0.0, "inverse metric must be positive");
    EnsArgs a;
    a.xs = xs; a.coords = coords; a.logp = logp; a.chain = chain; a.logp_chain = logp_chain;
    a.naccept = (long long*)naccept; a.n = apgp_npad(n); a.iterations = iterations;
    a.ndim = kc.ndim; a.nwalkers = nwalkers; a.seed = seed; a.mean = mean; a.amp = kc.amp;
    a.a_stretch = a_stretch;
    a.lin_coef = kc.lin_coef; a.lin_order = kc.lin_order;
    for (int d = 0; d < APGP_MAX_DIM; ++d) {
        a.lw[d] = kc.lw[d];
        a.sc[d] = d < kc.ndim ? kc.sc[d] : 1.0;
        a.lo[d] = d < kc.ndim ? lo[d] * kc.sc[d] : 0.0;     // bounds in scaled coordinates
        a.hi[d] = d < kc.ndim ? hi[d] * kc.sc[d] : 0.0;
    }
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((unsigned)nensembles), block(1024);
    const size_t xbytes = (size_t)a.n * (kc.dpad + 2) * sizeof(double);
    // (D > 16: the sampler state alone takes 100 KB of LDS -- the training stream stays in L2)
    const bool xlds = xbytes <= 96 * 1024 && kc.dpad <= 16;
    {
        // several workgroups per ensemble (ensemble_mw_kernel) when the ensembles alone leave compute units idle: G = the
        // proposals of a half-step, at most (CUs / ensembles), at least 2; every workgroup must be resident (one per CU
        // with the training stream in LDS)
        int devn = 0, cus = 0;
        if ((tl_ens_mode >= 0 ? tl_ens_mode : g_ens_mode.load()) == 0 && hipGetDevice(&devn) == hipSuccess &&
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, devn) == hipSuccess && devn >= 0 && devn < 64) {
            int G = nwalkers / 2;
            if ((long long)G * nensembles > cus) G = cus / nensembles;
            if (G >= 2 && iterations > 0) {
                std::lock_guard<std::mutex> lock(apgp_stream_lock(s));
                const size_t words = (size_t)nensembles * 2 * (ENS_MAXW / 2) * 2 + 8;      // (+ the status word)
                unsigned long long* xchg = (unsigned long long*)apgp_stream_scratch(4, s, words);
                if (!xchg || hipMemsetAsync(xchg, 0, words * 8, s) != hipSuccess) {
                    apgp_set_error("apgp_ensemble_sample: exchange buffer");
                    return -2;
                }
                EnsMwArgs q;
                q.e = a; q.xchg = xchg; q.status = xchg + (words - 8); q.timeout = 5000000ull; q.G = G;      // 50 ms of the 100 MHz clock
                dim3 gridm((unsigned)(nensembles * G)), blockm(256);
                static std::mutex attr_mu_m;
                static bool attr_done_m[5][64] = {{false}};
#define APGP_LAUNCH_ENS_MW(DP, SLOT)                                                                   \
    do {                                                                                               \
        if (xlds) {                                                                                    \
            {                                                                                          \
                std::lock_guard<std::mutex> lk(attr_mu_m);                                             \
                if (!attr_done_m[SLOT][devn]) {                                                        \
                    if (hipFuncSetAttribute((const void*)ensemble_mw_kernel<DP, true>,                 \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) != hipSuccess) { \
                        apgp_set_error("apgp_ensemble_sample: hipFuncSetAttribute(96 KiB of LDS) failed"); \
                        return -2;                                                                     \
                    }                                                                                  \
                    attr_done_m[SLOT][devn] = true;                                                    \
                }                                                                                      \
            }                                                                                          \
            hipLaunchKernelGGL((ensemble_mw_kernel<DP, true>), gridm, blockm, xbytes, s, q);           \
        } else {                                                                                       \
            hipLaunchKernelGGL((ensemble_mw_kernel<DP, false>), gridm, blockm, 0, s, q);               \
        }                                                                                              \
    } while (0)
                switch (kc.dpad) {
                    case 2: APGP_LAUNCH_ENS_MW(2, 0); break;
                    case 4: APGP_LAUNCH_ENS_MW(4, 1); break;
                    case 8: APGP_LAUNCH_ENS_MW(8, 2); break;
                    case 16: APGP_LAUNCH_ENS_MW(16, 3); break;
                    default: APGP_LAUNCH_ENS_MW(32, 4); break;
                }
#undef APGP_LAUNCH_ENS_MW
                hipLaunchKernelGGL(ens_mark_failed_kernel, dim3(1), dim3(256), 0, s, q.status, logp, (int)(nensembles * nwalkers));
                APGP_CHECK_LAUNCH();
                return 0;
            }
        }
    }
    // the LDS-resident form needs > 64 KiB of dynamic LDS: per-device attribute, set and CHECKED once per
    // device and instantiation
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
        apgp_set_error("apgp_ensemble_sample: hipGetDevice failed");
        return -2;
    }
    static std::mutex attr_mu;
    static bool attr_done[5][64] = {{false}};
#define APGP_LAUNCH_ENS(DP, SLOT)                                                                      \
    do {                                                                                               \
        if (xlds) {                                                                                    \
            {                                                                                          \
                std::lock_guard<std::mutex> lock(attr_mu);                                             \
                if (!attr_done[SLOT][dev]) {                                                           \
                    const hipError_t e_ = hipFuncSetAttribute((const void*)ensemble_kernel<DP, true>,  \
                                                              hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024); \
                    if (e_ != hipSuccess) {                                                            \
                        apgp_set_error("apgp_ensemble_sample: hipFuncSetAttribute(96 KiB of LDS) failed on device %d: %s", \
                                       dev, hipGetErrorString(e_));                                    \
                        return -2;                                                                     \
                    }                                                                                  \
                    attr_done[SLOT][dev] = true;                                                       \
                }                                                                                      \
            }                                                                                          \
            hipLaunchKernelGGL((ensemble_kernel<DP, true>), grid, block, xbytes, s, a);                \
        } else {                                                                                       \
            hipLaunchKernelGGL((ensemble_kernel<DP, false>), grid, block, 0, s, a);                    \
        }                                                                                              \
    } while (0)
    switch (kc.dpad) {
        case 2: APGP_LAUNCH_ENS(2, 0); break;
        case 4: APGP_LAUNCH_ENS(4, 1); break;
        case 8: APGP_LAUNCH_ENS(8, 2); break;
        case 16: APGP_LAUNCH_ENS(16, 3); break;
        default: APGP_LAUNCH_ENS(32, 4); break;
    }
#undef APGP_LAUNCH_ENS
    APGP_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------
// Candidate matrix of the acquisition sweep drawn ON the device (round 5; opt-in, ApproxPosterior(deviceCandidates)):
// the reference's point search starts from ``priorSample`` draws (utility.py:334-338); the batched counterpart
// sweeps M of them, and drawing 1e6 x 8 uniforms with NumPy + the H2D copy cost more than the sweep itself at
// C5's sizes (26 ms vs 18 ms).  Row i of the GLOBAL matrix is a pure function of (seed, i): counter-based
// Philox4x32-10, counter = (i low, i high, d / 2, 0x43414e44 "CAND"), key = seed -- so rank r of a sharded sweep
// generates exactly its rows [lo, hi) with idx_offset = lo, and anybody can regenerate the winning row alone.
//   T[i][d] = lo[d] + (hi[d] - lo[d]) * u,  u = 53-bit uniform in (0, 1) (u01 above).
// ---------------------------------------------------------------------------
struct BoxArgs {
    double* T;
    long long m, idx_offset;
    int ndim;
    unsigned long long seed;
    double lo[APGP_MAX_DIM], span[APGP_MAX_DIM];
};

__global__ __launch_bounds__(256) void box_candidates_kernel(BoxArgs a) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.m) return;
    const unsigned long long g = (unsigned long long)(a.idx_offset + i);
    double* row = a.T + i * a.ndim;
    for (int d = 0; d < a.ndim; d += 2) {
        unsigned int c[4] = {(unsigned int)g, (unsigned int)(g >> 32), (unsigned int)(d >> 1), 0x43414e44u};
        philox4x32(c, (unsigned int)a.seed, (unsigned int)(a.seed >> 32));
        row[d] = fma(a.span[d], u01(c[0], c[1]), a.lo[d]);
        if (d + 1 < a.ndim) row[d + 1] = fma(a.span[d + 1], u01(c[2], c[3]), a.lo[d + 1]);
    }
}

extern "C" int apgp_box_candidates(double* T, int64_t m, int32_t ndim, const double* lo, const double* hi,
                                   uint64_t seed, int64_t idx_offset, void* stream) {
    APGP_CHECK_ARG(T && lo && hi, "null pointer");
    APGP_CHECK_ARG(m >= 0 && idx_offset >= 0, "m >= 0 and idx_offset >= 0 required");
    APGP_CHECK_ARG(ndim >= 1 && ndim <= APGP_MAX_DIM, "1 <= ndim <= APGP_MAX_DIM required");
    if (m == 0) return 0;
    BoxArgs a;
    a.T = T; a.m = m; a.idx_offset = idx_offset; a.ndim = ndim; a.seed = seed;
    for (int d = 0; d < APGP_MAX_DIM; ++d) {
        a.lo[d] = d < ndim ? lo[d] : 0.0;
        a.span[d] = d < ndim ? hi[d] - lo[d] : 0.0;
        APGP_CHECK_ARG(d >= ndim || (a.span[d] >= 0.0 && a.span[d] < INFINITY), "bounds must be finite with lo <= hi");
    }
    hipLaunchKernelGGL(box_candidates_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    APGP_CHECK_LAUNCH();
    return 0;
}

// apgp_ensemble_sample with the kernel chosen PER CALL (mode 0: several workgroups per ensemble where that helps, 1: the
// single-workgroup kernel, < 0: the process-wide switch apgp_ensemble_mode): the re-run after a give-up (NaN in logp) does
// not change what other threads' calls get.
extern "C" int apgp_ensemble_sample_ex(const double* xs, int64_t n, const apgp_kernel_t* kern, double mean,
                                       const double* lo, const double* hi, int32_t nwalkers,
                                       int32_t nensembles, int64_t iterations, double a_stretch,
                                       uint64_t seed, double* coords, double* logp, double* chain,
                                       double* logp_chain, int64_t* naccept, int mode, void* stream) {
    const int saved = tl_ens_mode;
    tl_ens_mode = mode < 0 ? -1 : (mode ? 1 : 0);
    const int rc = apgp_ensemble_sample(xs, n, kern, mean, lo, hi, nwalkers, nensembles, iterations, a_stretch, seed, coords,
                                        logp, chain, logp_chain, naccept, stream);
    tl_ens_mode = saved;
    return rc;
}
