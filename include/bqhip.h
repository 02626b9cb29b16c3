/*
 * bqhip.h -- C ABI of libbqhip.so, the MI355X (gfx950) Bayesian-quadrature
 * GP engine.  This is the drop-in boundary: plain pointers and sizes, no
 * torch / numpy types.  The Python host package binds it with ctypes
 * (bayesian-quadrature_amd/_lib.py); INTEGRATION.md shows the stub a
 * maintainer of the reference would add.
 *
 * Reference interfaces replaced (jhamrick/bayesian-quadrature v0.2.0):
 *   bq_cho_factor      <- linalg_c.pyx:55-93   cho_factor(C, L)
 *   bq_cho_solve       <- linalg_c.pyx:96-179  cho_solve_vec / cho_solve_mat
 *   bq_logdet          <- linalg_c.pyx:182-210 logdet(L)
 *   bq_gram_gauss      <- gp.GP.Kxx / gp.GaussianKernel.__call__ (third-party
 *                         `gp` package, requirements.txt:2; used bq.py:147-162,465)
 *   bq_gp_fit          <- gp.GP.Lxx, .inv_Kxx_y, .log_lh   (bq.py:282,334-335,546)
 *   bq_gp_predict      <- gp.GP.mean(xo), diag(gp.GP.cov(xo)) (bq.py:200,227-228,942-943)
 *   bq_gp_logml_grid   <- the hyper-parameter loop body  (bq.py:536-550)
 *   bq_batch_fit_predict <- a Python loop over independent BQ problems
 *
 * Conventions
 *   - all floating point data is fp64; matrices are COLUMN-MAJOR (the
 *     reference's float64_t[::1, :]), points are d x n (gauss_c.pyx:116-117)
 *   - "host" pointers are caller-owned host memory; "_dev" entry points take
 *     device pointers obtained from bq_dev_alloc (or any HIP allocation on the
 *     context's device) and only enqueue work on the context's stream
 *   - every function returns a status code; bq_last_error() gives the text
 *   - a context is bound to one device and one stream and is not thread-safe;
 *     use one context per thread / per GPU
 */
#ifndef BQHIP_H
#define BQHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* status codes; the Python layer maps them like linalg_c.pyx:49-53,88-91 */
#define BQ_OK 0
#define BQ_ERR_NOT_PD 1  /* dpotrf info > 0  -> numpy.linalg.LinAlgError */
#define BQ_ERR_BAD_ARG 2 /* shape / illegal value -> ValueError */
#define BQ_ERR_HIP 3     /* HIP runtime failure -> RuntimeError */
#define BQ_ERR_NOMEM 4   /* allocation failure -> MemoryError */

#define BQ_MAX_DIM 8 /* largest input dimension d */

typedef struct bq_ctx bq_ctx;
typedef struct bq_fit bq_fit;

/* ---- contexts ------------------------------------------------------ */
int bq_device_count(int *count);
int bq_ctx_create(int device, bq_ctx **out);
/* adopt an existing hipStream_t (e.g. a torch stream); not destroyed by us */
int bq_ctx_create_on_stream(int device, void *hip_stream, bq_ctx **out);
void bq_ctx_destroy(bq_ctx *ctx);
int bq_ctx_sync(bq_ctx *ctx);
const char *bq_last_error(const bq_ctx *ctx);
/* name: at least 64 bytes.  cus = compute units, hbm_bytes = total memory */
int bq_device_info(bq_ctx *ctx, char *name, int *cus, size_t *hbm_bytes, int *clock_khz);
/* outer Cholesky block (multiple of 64; 0 = automatic from the size) */
int bq_set_block(bq_ctx *ctx, int nb);
/* look-ahead of one panel on a second, high-priority stream (default on; the
 * environment variable BQ_LOOKAHEAD=0 also disables it) */
int bq_set_lookahead(bq_ctx *ctx, int on);
/* the look-ahead runs while the bulk trailing update still has at least `min_rows` rows;
 * below that the sweep continues with sequential launches (default 3072, the measured
 * cross-over on MI355X; 0 = look-ahead to the end; also BQ_LA_MIN) */
int bq_set_lookahead_rows(bq_ctx *ctx, int min_rows);
/* the three settings above as they stand (a caller that changes them for a while reads them
 * first and puts them back; any pointer may be NULL) */
int bq_get_config(bq_ctx *ctx, int *nb, int *lookahead, int *min_rows);
/* counters of the context since its creation, out[0 .. n): [0] single-vector solves that were
 * re-issued on the per-block sweeps after a hand-off of the one-launch sweeps timed out (the
 * call still returned BQ_OK with the right answer; non-zero on a healthy, unshared device means
 * something is wrong with it).  Entries beyond the ones defined are set to 0.  No reference
 * counterpart (linalg_c.pyx:96-136 is a LAPACK call) */
int bq_ctx_stats(bq_ctx *ctx, int64_t *out, int n);
/* bq_batch_fit_predict and bq_gp_logml_grid keep their device workspace (up to half of
 * the free HBM) in the context between calls, so that a hyper-parameter loop
 * (bq.py:536-550) does not allocate and release it on every evaluation; this releases it */
int bq_ctx_trim(bq_ctx *ctx);

/* ---- device memory -------------------------------------------------- */
int bq_dev_alloc(bq_ctx *ctx, size_t bytes, void **dptr);
int bq_dev_free(bq_ctx *ctx, void *dptr);
int bq_upload(bq_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);   /* synchronous */
int bq_download(bq_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes); /* synchronous */
int bq_memset(bq_ctx *ctx, void *dst_dev, int byte, size_t bytes);

/* ---- stream timers (HIP events on the context's stream) ------------- */
int bq_timer_start(bq_ctx *ctx);
int bq_timer_stop_ms(bq_ctx *ctx, float *ms); /* records, synchronises, returns elapsed */

/* per-kernel-class device time of everything enqueued since the last reset,
 * measured with HIP events around each launch (slows the pipeline: use in a
 * separate, instrumented pass).  Classes: */
#define BQ_K_GRAM 0       /* gram_gauss / assemble kernels            (work: bytes) */
#define BQ_K_POTF2 1      /* 64x64 diagonal factor                    (work: flop)  */
#define BQ_K_TRSM 2       /* panel solve                              (work: flop)  */
#define BQ_K_GEMM 3       /* in-panel (left-looking) MFMA update      (work: flop)  */
#define BQ_K_SYRK 4       /* trailing MFMA update, gemm_sub_kernel<4,4> launches    */
#define BQ_K_SYRK_SMALL 5 /* trailing MFMA update, the smaller-tile launches        */
#define BQ_K_REDUCE 6     /* finalize / reductions / predict          (work: bytes) */
#define BQ_K_NCLASS 7
int bq_profile_enable(bq_ctx *ctx, int on);
int bq_profile_reset(bq_ctx *ctx);
/* ms[BQ_K_NCLASS], launches[BQ_K_NCLASS], work[BQ_K_NCLASS] = the ALGORITHMIC flops
 * or bytes of the launches of each class (trailing update: the lower half only,
 * m^2 k; Gram: 8 N^2 + 8 d N); synchronises.  Any pointer may be NULL. */
int bq_profile_read(bq_ctx *ctx, double *ms, int64_t *launches, double *work);
/* the bracketed launches as a timeline: rows {class, stream (0 main / 1 second), start ms,
 * end ms, algorithmic work}, times since the first bracketed launch.  keep = 1 with out = NULL
 * arms (and clears) the recording; keep = 0 stops it -- with out = NULL that call only reports
 * *nrows, with out it copies up to max_rows rows. */
int bq_profile_timeline(bq_ctx *ctx, int keep, double *out, int64_t max_rows, int64_t *nrows);

/* ---- linalg_c drop-ins: host buffers in and out --------------------- */
/* L <- lower Cholesky factor of C (n x n, ld = n).  C == L allowed (in place).
 * The strict upper triangle of L is left as it was in C ("upper values could
 * be anything", linalg_c.pyx:58-59).  BQ_ERR_NOT_PD when dpotrf would fail;
 * *info (may be NULL) receives the 1-based failing column. */
int bq_cho_factor(bq_ctx *ctx, const double *C, double *L, int64_t n, int64_t *info);
/* X <- (L L^T)^-1 B, B and X are n x nrhs, ld = n.  B == X allowed. */
int bq_cho_solve(bq_ctx *ctx, const double *L, const double *B, double *X, int64_t n,
                 int64_t nrhs);
/* *out <- 2 sum_i log L[i,i] */
int bq_logdet(bq_ctx *ctx, const double *L, int64_t n, double *out);

/* ---- Gaussian-kernel Gram ------------------------------------------- */
/* K[i,j] = h^2 N(x_i | x_j, diag(w^2)) + s^2 [i==j]; full symmetric n x n
 * matrix, host in / host out.  x is d x n, w has d entries. */
int bq_gram_gauss(bq_ctx *ctx, const double *x, int64_t d, int64_t n, double h, const double *w,
                  double s, double *K_out);
/* same on device-resident data; only enqueues.  x_dev: d x n, K_dev: n x n
 * with leading dimension ldk >= n. */
int bq_gram_gauss_dev(bq_ctx *ctx, const double *x_dev, int64_t d, int64_t n, double h,
                      const double *w, double s, double *K_dev, int64_t ldk);
/* cross Gram K[i,j] = k(x1_i, x2_j), n1 x n2, host in / out (gp.GP.Kxxo, .K) */
int bq_gram_gauss_cross(bq_ctx *ctx, const double *x1, int64_t n1, const double *x2, int64_t n2,
                        int64_t d, double h, const double *w, double *K_out);

/* ---- device-resident Cholesky (the MFMA roofline path) -------------- */
/* In-place lower Cholesky of the n x n device matrix (ld = lda).  n must be a
 * multiple of 64 (callers pad with an identity block).  Only enqueues (the first call at a
 * new largest n also allocates 1024 n + 32768 bytes of panel scratch in the context); the
 * failing column (0 = success) is written to info_dev[0] (device int32). */
int bq_potrf_dev(bq_ctx *ctx, double *A_dev, int64_t n, int64_t lda, int32_t *info_dev);

/* ---- GP fit objects (device resident) ------------------------------- */
/* Fit a GP: Gram, Cholesky, z = L^-1 y, log marginal likelihood.  x is d x n
 * (host), y has n entries (host).  The factor stays on the device. */
int bq_gp_fit(bq_ctx *ctx, const double *x, const double *y, int64_t d, int64_t n, double h,
              const double *w, double s, bq_fit **out);
/* same data, new hyper-parameters (the hyper-parameter loop, bq.py:933-965) */
int bq_gp_refit(bq_ctx *ctx, bq_fit *fit, double h, const double *w, double s);
/* same points, new targets y (n entries, host): the hyper-parameter loop gives GP2 new targets
 * on every evaluation, `self.gp_l.y = self.l_sc`, bq.py:948-954.  The fit holds no valid factor
 * until its next bq_gp_refit / bq_gp_refit_predict. */
int bq_gp_set_y(bq_ctx *ctx, bq_fit *fit, const double *y);
/* new hyper-parameters and the posterior mean / marginal variance at M points xo (d x M, host)
 * in ONE sweep: the points ride as border rows of the fit's own bordered system.  This is the
 * body of the reference's hyper-parameter loop, bq.py:933-947 (`_set_gp_log_l_params`: set the
 * parameters, then gp.mean(x_c) and diag gp.cov(x_c)).  mean / var may be NULL (not both);
 * M > 63 falls back to bq_gp_refit + bq_gp_predict. */
int bq_gp_refit_predict(bq_ctx *ctx, bq_fit *fit, double h, const double *w, double s,
                        const double *xo, int64_t M, double *mean, double *var);
void bq_fit_destroy(bq_ctx *ctx, bq_fit *fit);
int bq_gp_logml(bq_ctx *ctx, bq_fit *fit, double *out);
/* which: 0 = L (n x n, strict upper zeroed), 1 = alpha = Kxx^-1 y (n),
 * 2 = z = L^-1 y (n), 3 = Kxx (n x n, recomputed) */
int bq_gp_get(bq_ctx *ctx, bq_fit *fit, int which, double *out_host);
/* posterior at M points xo (d x M, host): mean[M], var[M] (marginal, prior
 * scale minus explained part; either may be NULL), cov (M x M full posterior
 * covariance, may be NULL). */
int bq_gp_predict(bq_ctx *ctx, bq_fit *fit, const double *xo, int64_t M, double *mean,
                  double *var, double *cov);

/* One pass "fit + posterior + log-ML" of ONE problem without keeping a fit
 * object: the bordered-Cholesky pipeline.  Host in / host out. */
int bq_fit_predict(bq_ctx *ctx, const double *x, const double *y, int64_t d, int64_t n,
                   double h, const double *w, double s, const double *xo, int64_t M,
                   double *mean, double *var, double *logml);

/* log marginal likelihood on a grid of G hyper-parameter points sharing the
 * data (x, y): h[G], w[G*d] (point g uses w[g*d .. g*d+d-1]), one s.
 * out[G]; a point whose Gram is not positive definite yields -inf
 * (bq.py:542-548).  chunk = problems factored per batched launch (0 = auto).
 * With s == 0 only the distinct w are factored (at h = 1) and every h is derived from
 * chol(h^2 G) = h chol(G). */
int bq_gp_logml_grid(bq_ctx *ctx, const double *x, const double *y, int64_t d, int64_t n,
                     const double *h, const double *w, double s, int64_t G, double *out,
                     int64_t chunk);

/* nprob independent problems of equal size: x[p] is d x n, y[p] has n entries,
 * xo[p] is d x M; all concatenated problem after problem.  Hyper-parameters
 * are shared.  mean/var: nprob x M, logml: nprob, status: nprob (0 ok, >0 the
 * failing column).  This is the unit that shards across GPUs: each rank calls
 * it on its own slice with its own context; there is no collective. */
int bq_batch_fit_predict(bq_ctx *ctx, int64_t nprob, const double *x, const double *y,
                         int64_t d, int64_t n, double h, const double *w, double s,
                         const double *xo, int64_t M, double *mean, double *var, double *logml,
                         int32_t *status);

/* ---- closed-form Gaussian-kernel integrals (gauss_c.pyx) ------------ */
/* Host buffers in and out; points d x n, w / mu length d, cov d x d (symmetric).
 * out_i = h^2 N(x_i | mu, diag(w^2) + cov)                       gauss_c.pyx:95-164 */
int bq_int_K(bq_ctx *ctx, const double *x, int64_t d, int64_t n, double h, const double *w,
             const double *mu, const double *cov, double *out);
/* out (n1 x n2) = int K1(x1, x') K2(x', x2) N(x' | mu, cov) dx'    gauss_c.pyx:235-339 */
int bq_int_K1_K2(bq_ctx *ctx, const double *x1, int64_t n1, const double *x2, int64_t n2,
                 int64_t d, double h1, const double *w1, double h2, const double *w2,
                 const double *mu, const double *cov, double *out);
/* out (n x n) = int int K1 K2 K1                                   gauss_c.pyx:416-531 */
int bq_int_int_K1_K2_K1(bq_ctx *ctx, const double *x, int64_t d, int64_t n, double h1,
                        const double *w1, double h2, const double *w2, const double *mu,
                        const double *cov, double *out);
/* out (n) = int int K1 K2                                          gauss_c.pyx:617-713 */
int bq_int_int_K1_K2(bq_ctx *ctx, const double *x, int64_t d, int64_t n, double h1,
                     const double *w1, double h2, const double *w2, const double *mu,
                     const double *cov, double *out);

/* ---- BQ moments on device-resident fits (bq_c.pyx) ------------------- */
/* X <- Kxx^-1 B with the resident factor of `fit`; B, X are n x nrhs (host) */
int bq_gp_solve(bq_ctx *ctx, bq_fit *fit, const double *B, int64_t nrhs, double *X);
/* E[Z] = (int K_l(x, x_sc) p(x) dx) . alpha_l, fused on the device: gp_l is the fit
 * of the second GP (its points are x_sc)                           bq_c.pyx:157-213 */
int bq_bq_Z_mean(bq_ctx *ctx, bq_fit *gp_l, const double *mu, const double *cov, double *out);
/* V(Z) = alpha' (int int K_l K_tl K_l) alpha - beta' K_tl^-1 beta, beta =
 * (int K_tl K_l) alpha; nothing n x n is materialised              bq_c.pyx:264-355 */
int bq_bq_Z_var(bq_ctx *ctx, bq_fit *gp_log_l, bq_fit *gp_l, const double *mu, const double *cov,
                double *out);

/* Active-sampling acquisition, batched over M candidate locations x_a (1-D):
 * for each a, the Gram of [x_sc, x_a[a]] (no noise, the reference's jitter on the
 * candidates within `thresh` of x_a[a] and on the new point) is factored and
 * A = K^-1 int K p is reduced to A_a[a] = A[last] and A_sc_l[a] = A[:nsc] . l_sc
 * -- the two numbers bq_c._esm_and_em needs (bq.py:447-527, bq_c.pyx:425-490).
 * One batched bordered Cholesky instead of M sequential refactorisations.
 * status[a] > 0: the system of candidate a is not positive definite. */
int bq_esm_batch(bq_ctx *ctx, const double *x_sc, const double *l_sc, int64_t ns, int64_t nsc,
                 const double *x_a, int64_t M, double h, double w, double thresh, const double *mu,
                 const double *cov, double *A_a, double *A_sc_l, int32_t *status);
/* The same quantities as bq_esm_batch, as a bordered UPDATE of gp_l's resident factor
 * instead of M refactorisations (bq.py:447-527 with the jitter rule of bq_c.pyx:136): one
 * multi-right-hand-side solve for the M borders, the candidate unit vectors, int K p and
 * l_sc, then a c x c Woodbury system per candidate (c = candidates within `thresh`).
 * gp_l: the fit over (x_sc, l_sc), samples first (ns of them), noise-free (s = 0) --
 * BQ_ERR_BAD_ARG otherwise.  status[a] = 1 where the bordered matrix is not positive
 * definite (the reference's LinAlgError fallback, bq.py:481-490). */
int bq_esm_border(bq_ctx *ctx, bq_fit *gp_l, int64_t ns, const double *x_a, int64_t M,
                  double thresh, const double *mu, const double *cov, double *A_a,
                  double *A_sc_l, int32_t *status);

/* ---- resident batch pipeline (what bench.py times) ------------------ */
/* A plan owns device copies of the inputs and all workspaces, so that a run
 * starts with everything resident in HBM and only enqueues kernels. */
/* ---- the stacked pair of GPs at S hyper-parameter sets in one batched pass --------------
 * bq.py:536-550, 933-965 (the hyper-parameter objective) and bq.py:604-662 (marginalize /
 * choose_next over sampled hyper-parameters) evaluate one parameter set after the other; the
 * sets are independent and run here as one batch.  GP1 is the GP over tl_s = log l_s at the
 * samples x_s, GP2 the GP over [l_s, exp(mean of GP1 at x_c)] at [x_s, x_c].  A pair object
 * keeps the points resident for S parameter sets; with ma > 0 acquisition points x_a it serves
 * bq_pair_esm, with ma = 0 bq_pair_llh.  Parameters are S x 3 row-major: (h, w, s) per set. */
typedef struct bq_pair bq_pair;
int bq_pair_create(bq_ctx *ctx, const double *x_s, const double *tl_s, const double *l_s, int64_t ns,
                   const double *x_c, int64_t nc, const double *x_a, int64_t ma, int64_t S,
                   bq_pair **out);
void bq_pair_destroy(bq_ctx *ctx, bq_pair *pair);
/* llh[b] = log_lh(GP1) + log_lh(GP2) under set b, -inf where the reference's closure returns
 * -inf (bq.py:542-548); status[b]: 0 ok, 1 GP1 not positive definite, 2 "GP mean is too large"
 * (bq.py:945-947), 3 GP2 not positive definite.  l_c (S x nc, may be NULL): the candidates'
 * values exp(mean) under every set. */
int bq_pair_llh(bq_ctx *ctx, bq_pair *pair, const double *p_tl, const double *p_l, double *llh,
                double *l_c, int32_t *status);
/* the acquisition's ingredients (bq.py:447-527, bq_c.pyx:425-490) under every set b and for every
 * acquisition point a (element b * ma + a): A_a, A_sc_l with status (1: singular bordered system,
 * the fallback of bq.py:481-490), GP1's posterior mean / variance tm_a, tC_a; per set: l_c (S x nc,
 * may be NULL) and sstatus (1: GP1 not positive definite, 2: GP mean too large). */
int bq_pair_esm(bq_ctx *ctx, bq_pair *pair, const double *p_tl, const double *p_l, double thresh,
                const double *mu, const double *cov, double *A_a, double *A_sc_l, int32_t *status,
                double *tm_a, double *tC_a, double *l_c, int32_t *sstatus);

typedef struct bq_plan bq_plan;
int bq_plan_create(bq_ctx *ctx, int64_t nprob, int64_t d, int64_t n, int64_t M, bq_plan **out);
void bq_plan_destroy(bq_ctx *ctx, bq_plan *plan);
/* upload inputs (host) of all nprob problems; hyper-parameters per problem:
 * h[nprob], w[nprob*d], s[nprob] */
int bq_plan_set_inputs(bq_ctx *ctx, bq_plan *plan, const double *x, const double *y,
                       const double *xo, const double *h, const double *w, const double *s);
/* enqueue one full pass (assemble -> bordered Cholesky -> finalize) */
int bq_plan_run(bq_ctx *ctx, bq_plan *plan);
/* synchronise and copy results out (any pointer may be NULL) */
int bq_plan_results(bq_ctx *ctx, bq_plan *plan, double *mean, double *var, double *logml,
                    int32_t *status);
/* bytes of device memory the plan holds */
int bq_plan_bytes(bq_plan *plan, size_t *bytes);
/* Test aid.  bq_set_guard(1): every device buffer the library allocates from now on carries a
 * 4 KiB sentinel band behind its last byte (also: BQ_GUARD=1 in the environment when the library
 * is loaded).  bq_plan_check_guards synchronises and reports how many of the plan's buffers carry
 * a band and how many sentinel bytes a pass has overwritten (0 for a correct launch sequence).
 * No reference counterpart: the reference's workspaces are numpy arrays (linalg_c.pyx:55-93
 * factors in place); this checks that the batched sweep's block rules size theirs correctly. */
int bq_set_guard(int on);
int bq_plan_check_guards(bq_ctx *ctx, bq_plan *plan, int64_t *guarded, int64_t *damaged);

/* ---- hardware probes (tools/probe.py, bench.py peak denominators) --- */
/* sustained v_mfma_f64_16x16x4_f64 rate in TFLOP/s over all CUs */
int bq_probe_mfma_f64(bq_ctx *ctx, double *tflops);
/* sustained v_fma_f64 rate in TFLOP/s */
int bq_probe_fma_f64(bq_ctx *ctx, double *tflops);
/* streaming fp64 write / copy bandwidth in GB/s over `bytes` */
int bq_probe_hbm(bq_ctx *ctx, size_t bytes, double *write_gbs, double *copy_gbs);
/* ns per hand-off between two one-wave workgroups (eight ping-pong pairs, the slowest pair):
 * mode 0 = partners on one XCD, plain payload stores + sc1 flag, sc1 loads (no cache maintenance);
 * 1 = partners on different XCDs behind agent-scope release / acquire; 2 = as 1 on one XCD.  kib:
 * KiB of payload per hand-off.  xcc16: the XCC id each of the 16 workgroups ran on; bad_words:
 * payload words that arrived stale.  Bounded spins (status 3 if a partner never answers).
 * Measurement behind docs/LABBOOK.md round 6 (the C2 chain as one launch); no reference
 * counterpart (linalg_c.pyx:55-93 is one LAPACK call). */
int bq_probe_xcd_hop(bq_ctx *ctx, int mode, int64_t iters, int64_t kib, double *ns_per_hop,
                     int32_t *xcc16, int64_t *bad_words);
/* C (m x n) -= P (m x k) Q (n x k)^T on scratch operands through the engine's own kernel
 * selection (lower: only the lower trapezoid; qt: Q given k-contiguous): average ms over `reps`
 * back-to-back launches -- the tuning probe behind tools/gemm_probe.py */
int bq_probe_gemm(bq_ctx *ctx, int64_t m, int64_t n, int64_t k, int lower, int64_t batch, int qt,
                  int64_t reps, double *ms);
/* `reps` read-only passes over `bytes` with 8-byte-per-lane loads, 512 contiguous bytes per
 * wave (the access pattern of the single-vector sweeps): a known byte count for calibrating
 * the profiler's FETCH_SIZE counter on that pattern; read_gbs may be NULL */
int bq_probe_hbm_read8(bq_ctx *ctx, size_t bytes, int64_t reps, double *read_gbs);
/* MFMA issue study: kind 0 = v_mfma_f64_16x16x4_f64, 1 = v_mfma_f64_4x4x4_4b_f64; nacc
 * independent accumulators per wave (1,2,4,8); blocks_per_cu = waves per SIMD */
int bq_probe_mfma_variant(bq_ctx *ctx, int kind, int nacc, int blocks_per_cu, double *tflops);
/* operand map of v_mfma_f64_4x4x4_4b_f64 under the CBSZ / ABID broadcast controls:
 * out[2*(la*64 + lb) + {0,1}] = low / high half of the 64-bit mask of D lanes fed by A lane la
 * and B lane lb */
int bq_probe_mfma444_layout(bq_ctx *ctx, int cbsz, int abid, int32_t *out8192);
/* relative error of v_rsq_f64 raw / after one / after two Newton steps at x[0..n):
 * err3[3*i + {0,1,2}] */
int bq_probe_rsq(bq_ctx *ctx, const double *x, int64_t n, double *err3);
/* out[i] = the Gram kernels' own exp (exp_gauss, csrc/common.h) of x[i] <= 0: per-element
 * accuracy probe (tests/test_gpu_parity.py::test_exp_gauss_accuracy). */
int bq_probe_exp(bq_ctx *ctx, const double *x, int64_t n, double *out);
/* device time per launch of a chain of n empty, dependent kernels (us) */
int bq_probe_launch(bq_ctx *ctx, int64_t n, double *us_per_launch);
/* One eager pass of a plan (one or two problems, outer block 64) with the profiling
 * instantiation of the one-launch slab step: 160 s_memtime stamps of workgroup 0 per step
 * (10 phase boundaries, then the diagonal factor's per-wave barrier stamps). */
int bq_probe_c2_timeline(bq_ctx *ctx, bq_plan *plan, int64_t *stamps, int64_t nsteps);
/* The 64 x 64 diagonal factor alone (the launch that heads every panel step): A is a
 * 64 x 64 host matrix, factored `reps` times from a resident copy (from_lds != 0: handed
 * over through LDS as the one-launch steps do).  Last launch's factor, its
 * BQ_DINV_HALF-double record (64 reciprocal pivots + four 16 x 16 block inverses), info,
 * HIP-event microseconds per launch and 136 in-kernel s_memtime stamps (5 phase
 * boundaries, then per panel and wave the arrival at / release from the panel barrier). */
int bq_probe_potf2(bq_ctx *ctx, const double *A, int from_lds, int64_t reps, double *L_out,
                   double *dinv_out, int32_t *info_out, double *us_per_launch,
                   int64_t *stamps136);
/* The batched panel solve of one outer block alone (potrf.hip, enqueue_panel_solve): X (m x kb per
 * problem, column-major, in / out) <- X L^-T against `batch` dense lower-triangular kb x kb
 * factors L; mode 0: as the context is configured, 1: recursive products + 64-column solves,
 * 2: the one-launch sweep (trsm_sweep_kernel).  m, kb multiples of 64.  reps > 0: the call is
 * repeated reps times on its own output and timed (HIP events, ms per call); X is then not
 * written back. */
int bq_probe_panel_solve(bq_ctx *ctx, int64_t m, int64_t kb, int64_t batch, const double *L,
                         double *X, int mode, int64_t reps, double *ms_per_call);
/* dump of the f64 MFMA D-register layout: out[64*4] receives, for lane l and
 * register r, the value row*16+col of the D element it holds */
int bq_probe_mfma_layout(bq_ctx *ctx, double *out256);

#ifdef __cplusplus
}
#endif
#endif /* BQHIP_H */
