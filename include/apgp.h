/*
 * apgp.h -- C ABI of the MI355X-native GP-surrogate hot path (libapgp.so).
 *
 * This is the drop-in boundary UNDER the duck-typed ``george.GP`` object that
 * dflemin3/approxposterior calls (the reference has no FFI of its own: its
 * seam is the Python object passed as ``gp=``, approx.py:77,140-144, and the
 * arithmetic lives in the third-party george wheel).  Every entry point below
 * replaces one piece of what george does for the reference's call sites; the
 * Python class ``approxposterior_amd.gp.GP`` (host side, ctypes) strings them
 * together behind george's method names.  INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - plain C: pointers + sizes, no torch / C++ types.  All matrices are IEEE
 *     fp64, row-major.  Pointers are DEVICE pointers owned by the caller unless
 *     the parameter is documented as "host".  ``stream`` is a hipStream_t
 *     passed as void* (NULL = the default stream).  Calls only ENQUEUE work;
 *     they do not synchronise unless documented.
 *   - return value: 0 = OK, <0 = bad argument (-1) / HIP error (-2);
 *     apgp_last_error() gives a thread-local message.  No exceptions cross
 *     the ABI.  Not thread-safe per stream.
 *   - gfx950 (MI355X) only; there is no CPU fallback in this library.
 */
#ifndef APGP_H
#define APGP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define APGP_ABI_VERSION 8
#define APGP_MAX_DIM 32          /* feature dimension D supported by the kernels (instantiated for D padded to 2 / 4 / 8 / 16 / 32) */
#define APGP_ROW_BLOCK 512       /* rows per packed L^-1 row block (sweep tile)  */
#define APGP_K_CHUNK 16          /* contraction depth per packed tile            */
#define APGP_CAND_BLOCK 64       /* candidates per sweep workgroup                */

/* Acquisition kinds (utility.py:99-250). */
#define APGP_UTIL_AGP 0          /* utility.AGPUtility   utility.py:99-142  */
#define APGP_UTIL_BAPE 1         /* utility.BAPEUtility  utility.py:145-189 */
#define APGP_UTIL_JONES 2        /* utility.JonesUtility utility.py:192-250 */
#define APGP_UTIL_NONE 3         /* predict only (george.GP.predict)        */

/*
 * ExpSquared(+Constant)(+Constant*Linear) kernel hyper-parameters in evaluated
 * form (george.kernels.ExpSquaredKernel with axis-aligned metric, optionally
 * ``c * kernel``, optionally ``+ c2 * LinearKernel(log_gamma2, order)``;
 * gpUtils.py:160-173):
 *   k(x,x') = amp * exp(-0.5 * sum_d (x_d-x'_d)^2 * inv_metric[d])
 *           + lin_coef * sum_d (x_d * x'_d)^lin_order
 *   amp        = ndim*exp(log_constant) when fitAmp, else 1
 *   inv_metric = exp(-log_M_d_d)
 *   lin_coef   = ndim*exp(log_constant2) * exp(-log_gamma2); 0 = no linear term.
 *                The per-axis sum is george's convention for non-stationary
 *                kernels; no reference test pins it for order > 1.
 *   diag_add   = yerr^2 + exp(white_noise), added to K_ii only (never to k(t,t))
 */
typedef struct apgp_kernel {
    int32_t ndim;
    int32_t lin_order;   /* integer power P >= 0 of the linear-regression term */
    double amp;
    double diag_add;
    double inv_metric[APGP_MAX_DIM];
    double lin_coef;
} apgp_kernel_t;

/* Result record of apgp_acquire (device or host memory, 16 bytes). */
typedef struct apgp_best {
    double u;        /* smallest utility found (+inf if no admissible candidate) */
    int64_t index;   /* its GLOBAL candidate index (idx_offset + row), -1 if none */
} apgp_best_t;

int apgp_abi_version(void);
const char* apgp_last_error(void);

/* ---- sizes ------------------------------------------------------------- */
/* N rounded up to the packed row block (512).                               */
int64_t apgp_npad(int64_t n);
/* doubles in the packed lower-triangular L^-1 image for N training points.  */
int64_t apgp_packed_linv_len(int64_t n);
/* doubles in the packed training stream (scaled X | alpha) for N, D.        */
int64_t apgp_packed_train_len(int64_t n, int32_t ndim);
/* doubles of scratch apgp_trtri_pack needs (two dense Np64 x Np64 panels).  */
int64_t apgp_trtri_work_len(int64_t n);

/* ---- K1: Gram matrix ------------------------------------------------------
 * Replaces george ``kernel.get_value(X)`` + the diagonal update inside
 * ``GP.compute`` (called from gpUtils.py:178,244,254; approx.py:717 and every
 * _nll evaluation, gpUtils.py:74-78).  Writes the LOWER triangle of the symmetric
 * N x N matrix K (leading dimension ldk >= N), diagonal included; nothing above the
 * diagonal is touched (allocate K zeroed if a dense lower-triangular array is wanted):
 * apgp_potrf, the only consumer, reads the lower triangle only, and half the HBM writes
 * is half the kernel's time.                                                          */
int apgp_gram(const double* X, int64_t n, const apgp_kernel_t* kern /*host*/,
              double* K, int64_t ldk, void* stream);

/* ---- cross kernel matrix ------------------------------------------------------
 * C (m x n, ld ldc) = k(X1_i, X2_j) without the diagonal term: george
 * ``kernel.get_value(x1, x2)``.  Feeds the incremental factor update when
 * ApproxPosterior.findNextPoint appends a design point (approx.py:693-717).     */
int apgp_kernel_cross(const double* X1, int64_t m, const double* X2, int64_t n,
                      const apgp_kernel_t* kern /*host*/, double* C, int64_t ldc, void* stream);

/* ---- Cholesky factorisation (lower, row-major, in place) -------------------
 * Replaces scipy.linalg.cholesky inside george BasicSolver.compute (every
 * gpUtils._nll evaluation, gpUtils.py:74-78; GP.compute, gpUtils.py:178,
 * approx.py:717).  Only the lower triangle of A is read and written.
 * Optionally carries a right-hand side through the factorisation: with
 * y, z != NULL, z = L^-1 (y - shift) comes out of the same launches (what
 * GP.log_likelihood needs right after a refactorisation: r^T K^-1 r = z.z), so
 * an _nll evaluation needs no separate triangular solve.
 * *info_dev (device int32): 0 = OK, k > 0 = leading minor of order k is not
 * positive definite (LAPACK dpotrf convention; george/SciPy raise LinAlgError).
 * For n > 64 the library keeps a stream-ordered scratch (hipMallocAsync) of
 * (n/64) (64^2 + 64) doubles per matrix per (device, stream), grown on demand. */
int apgp_potrf(double* A, int64_t n, int64_t lda, const double* y, double shift, double* z,
               int32_t* info_dev, void* stream);

/* ---- K2: log-determinant and diagonal range of the Cholesky factor -------
 * Replaces BasicSolver.compute's ``2*sum(log(diag(U)))`` (george; feeds
 * GP._const used by gpUtils._nll, gpUtils.py:78).  L is the lower factor
 * (row-major).  out[0] = log det K = 2*sum log L_ii, out[1] = min L_ii,
 * out[2] = max L_ii (so (out[2]/out[1])^2 is a condition estimate); the buffer
 * must hold 5 doubles (out[3], out[4] are written as 0).
 * apgp_fit_summary additionally returns out[3] = z.z (if z != NULL) and
 * out[4] = *info_dev, so one 40-byte copy fetches everything an _nll needs.   */
int apgp_logdet(const double* L, int64_t n, int64_t ldl, double* out5, void* stream);
int apgp_fit_summary(const double* L, int64_t n, int64_t ldl, const double* z,
                     const int32_t* info_dev, double* out5, void* stream);

/* ---- stream scratch ------------------------------------------------------------
 * apgp_potrf / apgp_nll_eval* (n > 64) and apgp_trsv (n >= 256) keep stream-ordered
 * device scratch per (device of the stream, stream), grown with hipMallocAsync -- so
 * they must not be called while the stream is being captured into a graph.
 * apgp_release_scratch frees (stream-ordered) what `stream` holds and returns the
 * number of buffers released: call it before destroying a stream.               */
int apgp_release_scratch(void* stream);

/* ---- one gpUtils._nll evaluation (gpUtils.py:46-80) in one call ---------------
 * apgp_gram -> apgp_potrf (z = L^-1 (y - mean) riding along) -> apgp_fit_summary, and the
 * 5-value record of apgp_fit_summary in out5_host when the call returns (the call
 * synchronises with its own work only).  n <= 64: ONE single-workgroup launch (Gram block
 * in LDS, register-resident factorisation, summary; same bits as the separate calls); 64 < n <= 128
 * (with y): both block columns in one single-workgroup launch, same bits again.
 * The record travels through 64 bytes of pinned, device-mapped host memory kept per
 * (device, stream): the last kernel's last lane writes it and a sequence word, the host
 * polls the word (400 us, then an ordinary stream synchronisation) -- no D2H copy; if
 * pinned memory is unavailable the call falls back to copy + synchronisation.
 * K: n x n work (holds the factor once the stream has passed the call's work), z: n,
 * info_dev / out5_dev: device scratch (both written as before).  Only the 5-value record in
 * out5_host is host-visible when the call returns: the host stops polling as soon as the
 * record lands, which may be before the posting kernel has finished -- K, z, *info_dev and
 * out5_dev are STREAM-ordered (read them from work enqueued on `stream`, or after
 * synchronising it), not host- or other-stream-visible on return.
 * Status as the parts'; a non-PD matrix is reported in out5_host[4] (> 0), not in the status.
 * 64 < n <= 3200 (50 block columns): the Cholesky is ONE persistent launch (csrc/potrf_persist.h:
 * row workgroups chained by in-launch hand-offs instead of a launch per 64-column step); above
 * that a hybrid -- a launch per step for the first block columns, ONE persistent launch for the
 * trailing 44 x 44 blocks; all bit-identical to the multi-launch path.  The persistent launch
 * needs one compute unit per workgroup (512 threads, 160 KB of LDS each: up to all 256 CUs), all
 * resident at once; if they are not within 50 ms (a kernel of another stream or process holds
 * compute units or LDS) the launch gives up -- every expired wait inside it marks the call
 * aborted -- and the call transparently re-runs the evaluation on the multi-launch path
 * (counted by apgp_potrf_fallbacks).  LATENCY CLIFF: such a call costs the 50 ms plus the
 * re-run; after it the next 64 evaluations on that device (doubling per consecutive give-up,
 * at most 4096) go straight to the multi-launch path (counted by apgp_potrf_backoff_skips), and
 * the first persistent launch that completes ends the back-off.
 * y == NULL and z == NULL (round 5; GP.compute / recompute): the factorisation alone on the same
 * plan, out5[3] = 0.                                                                          */
int apgp_nll_eval(const double* X, int64_t n, const apgp_kernel_t* kern /*host*/,
                  const double* y, double mean, double* K, double* z,
                  int32_t* info_dev, double* out5_dev, double* out5_host /*host*/,
                  void* stream);

/* Test / profiling switch for the Cholesky inside apgp_nll_eval (not read from the
 * environment): 0 = default: ONE persistent launch for 64 < n <= 3200, above that a hybrid
 * (a launch per 64-column step for the first block columns, one persistent launch for the
 * trailing 44 x 44 blocks); 1 = a launch per step only; 2 = persistent launch that gives up
 * at once (exercises the fallback); 3 = persistent launch wherever it can run (64 < n <= 4096).
 * + 16: the launch-per-step path without its paired trailing updates (two block columns per
 * pass over a tile while the trailing matrix is large); + 32: without the deferred tiles (a paired
 * wide step leaves half of its far tiles to the narrow step after it, whose launch is a latency
 * chain on a few workgroups) -- all variants return the same bits.
 * mode < 0 only queries; setting a mode also ends any back-off.  Returns the previous mode (-1: bad argument).
 * apgp_potrf_fallbacks: evaluations re-run on the multi-launch path so far (process-wide);
 * apgp_potrf_backoff_skips: evaluations that skipped the persistent launch while backing off. */
int apgp_potrf_mode(int mode);
int64_t apgp_potrf_fallbacks(void);
int64_t apgp_potrf_backoff_skips(void);

/* ---- `batch` _nll evaluations at different hyper-parameters, one call ----------
 * (the restarts of gpUtils.optimizeGP, gpUtils.py:223-247,
 * evaluated in lock-step.)  Same training set X, y; kerns[batch] / means[batch] on the
 * host; K: batch x n x n work (the factors on return), z: batch x n, info_dev: batch
 * int32, out5_dev / out5_host: batch x 5 doubles laid out as apgp_fit_summary's record.
 * One batched Cholesky (gridDim.y = batch): values bit-identical to single calls.
 * n <= 128 and batch <= 64 (round 5; the README configuration's restarts): ONE launch of the fused
 * single-workgroup evaluation of apgp_nll_eval, a workgroup per matrix; kernel constants, shifts and
 * the records travel through the stream's pinned, device-mapped staging area (each workgroup posts
 * its own sequence word: no copy, no synchronisation).  Same bits again.
 * 128 < n <= 3200 and 2 <= batch <= 6 (round 6; the look-ahead points of a Powell line search,
 * gpUtils.py:238): the matrices' persistent factorisations run SIDE BY SIDE in one launch (gridDim.y
 * = batch; a batched Gram launch before it, the batched finish launch after it: three launches as
 * for one evaluation), each matrix on 1 / batch of the CUs with its own scratch, flags and record:
 * a batch costs little more than one evaluation (a persistent factorisation is a latency chain on a
 * quarter of the chip).  Every matrix runs the code of the single call: same bits.  A matrix that
 * gives up sends the batch to the batched launch-per-step path (as n > 3200 and batch > 6 go
 * anyway).  apgp_nll_side_batches: batches served side by side so far (process-wide).            */
int64_t apgp_nll_side_batches(void);
int apgp_nll_eval_batch(const double* X, int64_t n, int64_t batch,
                        const apgp_kernel_t* kerns /*host*/, const double* y,
                        const double* means /*host*/, double* K, double* z,
                        int32_t* info_dev, double* out5_dev, double* out5_host /*host*/,
                        void* stream);

/* ---- K3: triangular solves for z = L^-1 (b - shift), alpha = L^-T z --------
 * Replaces BasicSolver.apply_inverse / dot_solve on a vector (scipy cho_solve;
 * george GP.log_likelihood and _compute_alpha; gpUtils.py:78, utility.py:131).
 * trans = 0: solve L x = (b - shift); trans = 1: solve L^T x = (b - shift).
 * If sumsq != NULL, *sumsq = x.x (device scalar).  x may alias b.  Below n = 256: one workgroup,
 * right-hand side in LDS.  256 <= n <= 16384: ONE persistent launch (a workgroup per 64-row block,
 * solved blocks handed on as data-tagged granules; stream-ordered scratch per (device, stream):
 * slot 3 of csrc/scratch.h).  Above, or with apgp_trsv_mode(1): one launch per 256 rows with n
 * doubles of stream-ordered scratch.  All paths return the same bits.  If the persistent launch's
 * workgroups are not all resident within its timeout, x and *sumsq are written as NaN (no
 * automatic re-run: callers that may share the device re-issue with apgp_trsv_mode(1)).      */
int apgp_trsv(const double* L, int64_t n, int64_t ldl, const double* b, double shift,
              int trans, double* x, double* sumsq, void* stream);
/* Test / profiling switch (not read from the environment): 1 = the multi-launch path, 0 = default
 * (persistent launch where it applies); < 0 queries.  Returns the previous value.             */
int apgp_trsv_mode(int multi_launch);
/* apgp_trsv with the path chosen for THIS call (0: persistent launch where it applies, 1: a launch per
 * 256 rows, < 0: the process-wide switch): the re-run of one solve that came back NaN, without changing
 * what other threads' or streams' calls get (round 6).                                */
int apgp_trsv_ex(const double* L, int64_t n, int64_t ldl, const double* b, double shift,
                 int trans, double* x, double* sumsq, int mode, void* stream);

/* ---- pivot of an appended factor row -----------------------------------------
 * Incremental fit when ApproxPosterior.findNextPoint appends a design point
 * (approx.py:693-717): with l = L^-1 k(x_new, X_old) already in the new row (apgp_kernel_cross
 * + apgp_trsv, *ss = l.l on the device), writes *ljj = sqrt(kdiag - *ss), kdiag = k(x_new,
 * x_new) + diag_add.  *info_dev (device int32, caller-initialised to 0) keeps the FIRST
 * non-positive pivot's 1-based leading-minor order (LAPACK convention; rows are appended in
 * stream order) and 1.0 is stored so that later rows stay finite.                        */
int apgp_append_diag(double* ljj, const double* ss, double kdiag, int32_t* info_dev, int64_t order,
                     void* stream);

/* ---- K3 through the resident explicit inverse -------------------------------
 * x = W (b - shift) (trans = 0) or x = W^T b (trans = 1, shift must be 0) for the dense
 * lower-triangular W = L^-1 that apgp_trtri_pack leaves in the first panel of its work
 * buffer (winv, leading dimension ldw = n rounded up to 64): z and alpha of george's
 * GP._compute_alpha (behind every GP.predict, utility.py:131,178,224; approx.py:178) as two
 * HBM-rate matrix-vector products once the sweep's W exists, instead of two triangular
 * solves.  If sumsq != NULL, *sumsq = x.x (device scalar).  x must not alias b.
 * work: apgp_winv_apply_work_len(n) doubles (trans = 1 only).  Use apgp_trsv instead when the
 * condition estimate exceeds ~1e10 (same rule as apgp_acquire vs apgp_acquire_solve).     */
int64_t apgp_winv_apply_work_len(int64_t n);
int apgp_winv_apply(const double* winv, int64_t ldw, int64_t n, const double* b, double shift,
                    int trans, double* x, double* sumsq, double* work, void* stream);

/* ---- L^-1 in the sweep's packed tile layout --------------------------------
 * Computes W = L^-1 (blocked recursive triangular inversion, MFMA-f64 GEMM
 * merges) and writes it as lower-triangular 512 x 16 tiles in MFMA A-fragment
 * order (see DESIGN.md "packed factor").  This is what lets the sweep evaluate
 * BasicSolver.apply_inverse(Kxs.T) (george GP.predict, utility.py:131,178,224)
 * for millions of candidates without materialising Kxs.
 * work: apgp_trtri_work_len(n) doubles; on return its first panel holds the
 * dense row-major L^-1 with leading dimension n rounded up to 64.
 * packed: apgp_packed_linv_len(n) doubles (may be NULL).
 * If winv_dense != NULL (n x n, ld n) a compact copy of L^-1 is also stored. */
int apgp_trtri_pack(const double* L, int64_t n, int64_t ldl, double* work,
                    double* packed, double* winv_dense, void* stream);

/* ---- packed training stream ------------------------------------------------
 * rows k = 0..npad-1 of [ x_k * sqrt(inv_metric/2) (Dpad) | alpha_k | 0 ].   */
int apgp_pack_train(const double* X, const double* alpha, int64_t n,
                    const apgp_kernel_t* kern /*host*/, double* xs, void* stream);

/* ---- K5/K6: fused predict + acquisition sweep + arg-min --------------------
 * For every candidate row t of T (m x ndim):
 *   mu  = k(t,X).alpha + mean                     (george GP.predict)
 *   var = amp - || L^-1 k(t,X)^T ||^2             (return_var=True)
 *   u   = utility(kind)(mu, var)                  (utility.py:136,183,229-244)
 *   u   = +inf if t is outside [lo,hi] (box prior; utility.py:126,173,219) or
 *         mask[i] == 0
 * and the arg-min over candidates (ties -> lowest index, NaN never wins):
 * the batched counterpart of utility.minimizeObjective (utility.py:253-372).
 * mu / var / u may be NULL (not stored).  lo/hi are host arrays or NULL.
 * part: scratch of apgp_acquire_work_len(m, n) doubles, 16-byte aligned: the
 *       per-block arg-min partials plus, for n > APGP_ROW_BLOCK, the stream in
 *       which each persistent workgroup parks the k* operands it generated for
 *       one row block so that later row blocks do not regenerate them, and the
 *       per-row-block shares of the candidate blocks of a short last round
 *       (those are split over one workgroup per row block).
 * best: device apgp_best_t, written by the final reduction kernel.
 * ybest = max(y) and zeta are used by JONES only.                            */
int64_t apgp_acquire_work_len(int64_t m, int64_t n);
int apgp_acquire(const double* T, int64_t m, int64_t idx_offset,
                 const double* packed_linv, const double* xs, int64_t n,
                 const apgp_kernel_t* kern /*host*/, double mean, int32_t kind,
                 const double* lo /*host*/, const double* hi /*host*/,
                 const uint8_t* mask, double zeta, double ybest,
                 double* mu, double* var, double* u,
                 void* part, apgp_best_t* best, void* stream);

/* ---- substitution form of apgp_acquire (no explicit inverse) ------------------
 * Same semantics, outputs, scratch (apgp_acquire_work_len) and arg-min contract as
 * apgp_acquire, but sigma^2 = amp - |L^-1 k*|^2 comes from a BLOCKED FORWARD
 * SUBSTITUTION against the factor L itself -- what george's cho_solve does
 * (BasicSolver.apply_inverse, reached from utility.py:131,178,224) -- instead of
 * a product with the packed explicit inverse.  Use it when the condition
 * estimate of apgp_fit_summary, (out[2]/out[1])^2, exceeds ~1e10 (the reference's
 * own fitAmp=True optimum, tests/test_OptimizeGP.py:50, has cond(K) 8.5e15: the
 * explicit inverse is 200x off there, this form stays in cho_solve's error
 * class); it runs on the same matrix-core stream at about the same rate, for any n
 * the inverse form takes.
 * apgp_pack_lsolve writes the tiles it streams (apgp_packed_lsolve_len(n) doubles,
 * same geometry as the packed inverse): -L below the diagonal 16 x 16 blocks, and
 * those blocks prepared for the in-register solve (their 4 x 4 diagonal blocks
 * inverted, the rest negated).  One O(n^2) pass, no triangular inversion.     */
int64_t apgp_packed_lsolve_len(int64_t n);
int apgp_pack_lsolve(const double* L, int64_t n, int64_t ldl, double* packed, void* stream);
int apgp_acquire_solve(const double* T, int64_t m, int64_t idx_offset,
                       const double* packed_lsolve, const double* xs, int64_t n,
                       const apgp_kernel_t* kern /*host*/, double mean, int32_t kind,
                       const double* lo /*host*/, const double* hi /*host*/,
                       const uint8_t* mask, double zeta, double ybest,
                       double* mu, double* var, double* u,
                       void* part, apgp_best_t* best, void* stream);

/* ---- ONE candidate: george GP.predict(y, t[1 x D], return_var=True) ----------
 * What the reference's scalar utilities evaluate once per Nelder-Mead step
 * (utility.py:131,178,224 <- minimizeObjective, utility.py:336-372; the default
 * point search of ApproxPosterior.findNextPoint).  t_host: the candidate (host,
 * ndim doubles; it travels in the kernel arguments).  xs: packed training stream
 * (with alpha).  winv / ldw: the dense L^-1 left by apgp_trtri_pack in its work
 * buffer -> sigma^2 by one matrix-vector product; or winv = NULL and L / ldl: the
 * factor -> one triangular solve (use it above the conditioning gate).  work:
 * apgp_predict1_work_len(n) doubles (device).  out2_host: mu, sigma^2 -- through the
 * stream's pinned mailbox when the call returns (no allocation, copy or stream
 * synchronisation; falls back to copy + synchronisation without pinned memory).
 * Three small launches; through winv at n <= 256 ONE single-workgroup launch (round 5,
 * same bits; apgp_potrf_mode 1 keeps the three).                                       */
int64_t apgp_predict1_work_len(int64_t n);
int apgp_predict1_host(const double* t_host /*host*/, const double* xs, int64_t n,
                       const apgp_kernel_t* kern /*host*/, double mean,
                       const double* winv, int64_t ldw, const double* L, int64_t ldl,
                       double* work, double* out2_host /*host*/, void* stream);

/* ---- mean-only prediction (the batched ApproxPosterior._gpll path) --------
 * mu_i = k(t_i,X).alpha + mean for m candidates (approx.py:178-180).         */
int apgp_predict_mean(const double* T, int64_t m, const double* xs, int64_t n,
                      const apgp_kernel_t* kern /*host*/, double mean,
                      double* mu, void* stream);

/* Same, from / to HOST buffers in one call (H2D, kernel, D2H, one stream
 * synchronisation): the per-half-step call of an ensemble sampler
 * (approx.py:839-846 -> _gpll, approx.py:178-180) is latency-bound.
 * work: device scratch of m * (ndim + 1) doubles.                            */
int apgp_predict_mean_host(const double* T_host, int64_t m, const double* xs, int64_t n,
                           const apgp_kernel_t* kern /*host*/, double mean,
                           double* mu_host, double* work, void* stream);

/* ---- on-device ensemble MCMC over the GP-mean surrogate ---------------------
 * The whole loop of ApproxPosterior.runMCMC (approx.py:839-846): emcee's stretch
 * move (a = a_stretch, red/blue halves with a random cyclic offset per iteration)
 * with log-probability = GP predictive mean (ApproxPosterior._gpll,
 * approx.py:148-189) and a box prior [lo, hi] (host arrays, required: -inf
 * outside), as one persistent kernel; nensembles independent ensembles run as
 * one workgroup each (replicas).  coords: nensembles x nwalkers x ndim, in =
 * initial state, out = final state; logp / naccept: nensembles x nwalkers;
 * chain (iterations x nensembles x nwalkers x ndim) and logp_chain may be NULL.
 * nwalkers even, >= 2 ndim, <= 256.  RNG: Philox4x32-10 keyed by seed.         */
/* Round 5: when the ensembles alone leave compute units idle (nensembles x nwalkers / 2 <= CUs, or fewer workgroups per
 * ensemble when not), ONE ensemble runs on several workgroups: each evaluates its share of a half-step's proposals
 * and the log-probabilities are exchanged as data-tagged granules in memory (stream-ordered scratch, slot 4 of
 * csrc/scratch.h); same RNG streams and proposals as the single-workgroup kernel, GP means summed in another order
 * (chains agree to rounding, not bit for bit).  All workgroups must be resident at once; if they are not within
 * 50 ms the launch writes NaN into logp[] -- the caller re-runs with apgp_ensemble_mode(1) (the Python wrapper does).
 * apgp_ensemble_mode: 0 = default, 1 = single-workgroup kernel only; < 0 queries; returns the previous value.   */
int apgp_ensemble_mode(int mode);
/* apgp_ensemble_sample with the kernel chosen for THIS call (0: several workgroups per ensemble where
 * that helps, 1: the single-workgroup kernel, < 0: the process-wide switch).  Round 6: a workgroup
 * that gives up also raises a sticky word of the launch, and a one-workgroup kernel behind it turns
 * that word into NaN in ALL of logp[] -- the marker cannot be overwritten by a workgroup that finishes
 * normally afterwards.                                                                                */
int apgp_ensemble_sample_ex(const double* xs, int64_t n, const apgp_kernel_t* kern /*host*/, double mean,
                            const double* lo /*host, MAX_DIM*/, const double* hi /*host, MAX_DIM*/,
                            int32_t nwalkers, int32_t nensembles, int64_t iterations, double a_stretch,
                            uint64_t seed, double* coords, double* logp, double* chain,
                            double* logp_chain, int64_t* naccept, int mode, void* stream);
int apgp_ensemble_sample(const double* xs, int64_t n, const apgp_kernel_t* kern /*host*/, double mean,
                         const double* lo /*host*/, const double* hi /*host*/,
                         int32_t nwalkers, int32_t nensembles, int64_t iterations,
                         double a_stretch, uint64_t seed,
                         double* coords, double* logp, double* chain, double* logp_chain,
                         int64_t* naccept, void* stream);

/* ---- candidate matrix of the sweep drawn on the device (round 5, opt-in) ------
 * The batched counterpart of the ``sampleFn`` draws utility.minimizeObjective starts from
 * (utility.py:334-338) when the prior is the box: T (m x ndim, device) row i =
 * lo + (hi - lo) * u(seed, idx_offset + i), u = 53-bit uniforms in (0, 1) from counter-based
 * Philox4x32-10 (counter = (row low, row high, d / 2, "CAND"), key = seed).  Row g of the global
 * matrix depends on (seed, g) only: a rank of a sharded sweep generates its own rows with
 * idx_offset = first row, and any rank regenerates the winning row alone.               */
int apgp_box_candidates(double* T, int64_t m, int32_t ndim, const double* lo /*host*/,
                        const double* hi /*host*/, uint64_t seed, int64_t idx_offset, void* stream);

/* ---- K4: gradient of the log-likelihood wrt kernel hyper-parameters -------
 * Replaces george GP.grad_log_likelihood (gpUtils._grad_nll, gpUtils.py:110):
 *   g_k = 0.5 * sum_ij (alpha alpha^T - K^-1)_ij dK_ij/dtheta_k.
 * winv/ldw: the dense L^-1 that apgp_trtri_pack leaves in the first panel of
 * its work buffer (ldw = n rounded up to 64).  work: apgp_grad_work_len(n)
 * doubles (K^-1 = W^T W is formed there).  out (device, 2+ndim doubles):
 *   out[0] = sum(alpha) (d/d mean), out[1] = d/d log_constant (amp part),
 *   out[2+d] = d/d log_M_d_d (d < ndim),
 *   out[2+APGP_MAX_DIM] = 0.5 * trace(alpha alpha^T - K^-1): times exp(white_noise)
 *   it is d/d white_noise (george fit_white_noise=True),
 *   out[3+APGP_MAX_DIM] = 0.5 * sum_ij (alpha alpha^T - K^-1)_ij K_lin_ij: d/d log_constant
 *   of the linear term, and minus d/d log_gamma2.  out: 4 + APGP_MAX_DIM doubles.   */
int64_t apgp_grad_work_len(int64_t n);
/* K^-1 by the SOLVE route, for factors whose explicit inverse must not be trusted (the host's
 * conditioning gate): what george does -- K^-1 = cho_solve(L, I) inside grad_log_likelihood
 * (gpUtils.py:110) -- two blocked triangular solves against the identity, no product of
 * inverses.  xwork: apgp_kinv_solve_work_len(n) doubles; kinv: n x n (ld n), the lower
 * 64 x 64 tiles are written -- pass it as apgp_grad_loglik's `work` with winv = NULL.       */
int64_t apgp_kinv_solve_work_len(int64_t n);
int apgp_kinv_solve(const double* L, int64_t n, int64_t ldl, double* xwork, double* kinv, void* stream);
/* winv == NULL: `work` already holds K^-1 (apgp_kinv_solve); otherwise K^-1 = W^T W is formed there. */
int apgp_grad_loglik(const double* X, const double* alpha, const double* winv, int64_t ldw,
                     int64_t n, const apgp_kernel_t* kern /*host*/,
                     double* work, double* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* APGP_H */
