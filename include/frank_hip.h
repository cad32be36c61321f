/*
 * frank_hip.h -- C ABI of the MI355X (gfx950) visibility-fitting hot path.
 *
 * Drop-in boundary for discsim/frank v1.2.3.  The reference is pure Python with no
 * FFI layer of its own; each entry point below names the reference interface
 * (file:line under the reference tree) whose arithmetic it replaces, and
 * INTEGRATION.md shows the ctypes binding a frank maintainer would add.
 *
 * Conventions
 *   - plain C types, caller-owned buffers, no torch / numpy types in signatures;
 *   - every function returns FH_OK (0) or a negative FH_ERR_* code and records a
 *     message retrievable with fh_last_error() (thread-local);
 *   - all matrices are row-major (NumPy C order), IEEE fp64;
 *   - "host" pointers are ordinary malloc'd memory, "device" pointers are HIP
 *     allocations on the context's device;
 *   - handles are thread-compatible, not thread-safe; one HIP stream per fh_ctx;
 *   - there is NO CPU fallback for device work: with no usable GPU the device entry
 *     points fail with FH_ERR_HIP.
 */
#ifndef FRANK_HIP_H
#define FRANK_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FH_OK 0
#define FH_ERR_INVALID (-1)       /* bad argument (ValueError / AttributeError in the reference)              */
#define FH_ERR_QRANGE (-2)        /* q_k[-1] < max deprojected baseline: statistical_models.py:526-535        */
#define FH_ERR_BAD_P (-3)         /* non-positive / NaN power spectrum: statistical_models.py:688-698         */
#define FH_ERR_NOT_SPD (-4)       /* Cholesky failed and the SVD route could not be taken                     */
#define FH_ERR_NOMEM (-5)
#define FH_ERR_HIP (-6)           /* no GPU, or a HIP / rocBLAS / rocSOLVER / RCCL runtime error              */
#define FH_ERR_UNSUPPORTED (-7)   /* e.g. nu != 0                                                             */
#define FH_ERR_NUMERIC (-8)       /* ValueError("Round off in slope calculation"): minimizer.py:136-137         */

/* vis_model: statistical_models.py:71-73, 486-496 */
#define FH_VIS_OPT_THICK 0 /* H scaled by cos(inc) */
#define FH_VIS_OPT_THIN 1  /* no scaling           */
#define FH_VIS_DEBRIS 2    /* geometrically thick: exp(-kz^2 H2[k]) per visibility and column, see fh_ctx_set_scale_height */

typedef struct fh_dht fh_dht; /* DiscreteHankelTransform, hankel.py:25-294        */
typedef struct fh_vis fh_vis; /* a visibility table resident in HBM               */
typedef struct fh_ctx fh_ctx; /* device + stream + workspaces for one DHT size    */
typedef struct fh_comm fh_comm; /* RCCL communicator (one rank per GPU)           */

/* SourceGeometry / FixedGeometry parameters, geometry.py:196-200, 372-396 (degrees, arcsec). */
typedef struct fh_geometry {
    double inc_deg, PA_deg, dRA_arcsec, dDec_arcsec;
} fh_geometry;

const char *fh_last_error(void);
/* Hardware queues.  HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4; read when the ROCm runtime comes up, at the
 * first HIP call of the process) and kernels whose streams share a queue serialise; a pipeline of fits (fh_fit_submit) uses up to
 * six launch streams beside the binning stream.  Loading the library -- and importing the Python package -- changes nothing in the
 * process.  The first entry point that is about to touch the device (fh_ctx_create, fh_vis_upload*, fh_device_count) exports
 * GPU_MAX_HW_QUEUES=24 if the runtime is not up yet and the variable is unset; fh_init() does the same on request (an embedding
 * application with HIP code of its own calls it, or exports the variable, before that code runs).  Returns the number of queues the
 * runtime came up with, as far as the library can tell (the variable's value, or 4 when somebody else initialised the runtime with
 * the variable unset).  A context created on fewer than eight leaves a message in fh_last_warning() ("" otherwise; valid until the
 * next fh_ctx_create on this thread).                                                                                          */
int fh_init(void);
const char *fh_last_warning(void);
const char *fh_version(void);
/* Number of usable HIP devices (0 => every device entry point returns FH_ERR_HIP). */
int fh_device_count(int *count);

/* ---- a1/a2/a4: DiscreteHankelTransform set-up -- host, O(N^2), once per fitter ----------------------------
 * hankel.py:55-93.  Rmax in RADIANS (radial_fitters.py:441 converts from arcsec).  nu must be 0.            */
int fh_dht_create(double Rmax_rad, int N, int nu, fh_dht **out);
void fh_dht_destroy(fh_dht *dht);
int fh_dht_size(const fh_dht *dht);
/* Any output may be NULL.  r, q, scale_factor: N; zeros: N+1 (j_{0,1..N+1}); Ykm: N*N; scalars: 1.         */
int fh_dht_get(const fh_dht *dht, double *r, double *q, double *zeros, double *Ykm, double *scale_factor,
               double *Qmax, double *Rmax);

/* The Taylor tables behind bin_gram's evaluation of J0((q/Qmax) j_k) (hankel.py:201-202), for inspection and tests
 * (host only, no GPU needed).  Visibilities are grouped into buckets of s = q/Qmax of width *delta = 1/(2 j_N); in
 * bucket b, with tau = (s - (b + 1/2) delta) / (delta / 2) in [-1, 1],
 *     J0(s j_k) = sum_{n < 12} table[(b - b0) * 12 * N + n * N + k] * tau^n        (truncation error 1.2e-16).
 * table: (b1 - b0) * 12 * N doubles, or NULL to query delta only.                                                    */
int fh_dht_bucket_tables(const fh_dht *dht, int b0, int b1, double *table, double *delta);
/* The same tables as a context holds them on the device for its first nb buckets, [bucket][12][N] -- since round 5 they are built
 * THERE (host seeds every 16th bucket, double-double Taylor marching in between; FRANK_AMD_K1_TABLES=host keeps the long-double
 * construction): the tests hold them to fh_dht_bucket_tables ABSOLUTELY -- |device - host| <= 2.2e-16 (one ulp of 1) per entry,
 * 4 ulp of 1 per (bucket, column): every entry multiplies |tau|^n <= 1; relatively the high orders of the first buckets differ
 * by ~1e-7 (they are 1e-7-relative round-off in the host's tables too).                                                     */
int fh_ctx_bucket_tables(fh_ctx *ctx, int nb, double *table);

/* ---- contexts ------------------------------------------------------------------------------------------- */
int fh_ctx_create(const fh_dht *dht, int device, fh_ctx **out);
void fh_ctx_destroy(fh_ctx *ctx);
int fh_ctx_synchronize(fh_ctx *ctx);
/* The context's hipStream_t (as void*), e.g. to record HIP events around the binning kernel. */
void *fh_ctx_stream(fh_ctx *ctx);

/* ---- a3/a7: design block H(q) on the GPU -------------------------------------------------------------------
 * DHT.coefficients(q, direction) * scale: hankel.py:187-204, statistical_models.py:483-509.
 * q: n host doubles (lambda for forward, radians for backward); H: n*N host doubles, row-major.
 * direction 0 = forward, 1 = backward.  scale multiplies every entry (cos(inc), 1, or 1/cos(inc)).         */
int fh_dht_coefficients(fh_ctx *ctx, const double *q, int64_t n, int direction, double scale, double *H);

/* predict_visibilities: V = H(q) . I, chunk-free (statistical_models.py:279-329). q, V: n host doubles.     */
int fh_predict_visibilities(fh_ctx *ctx, const double *q, int64_t n, const double *I, double scale, double *V);

/* ---- visibility tables ------------------------------------------------------------------------------------
 * Upload (u, v, Re V, Im V, w) to HBM as five fp64 columns (40 B / visibility).  Vim may be NULL (real V).
 * n_w == 1 broadcasts a scalar weight (statistical_models.py:173).                                         */
int fh_vis_upload(int device, const double *u, const double *v, const double *Vre, const double *Vim,
                  const double *w, int64_t n_w, int64_t n, fh_vis **out);
/* fh_vis_upload_c128: the same table from a complex128 array of visibilities as NumPy holds it -- Vc = n (re, im) pairs --: one
 * contiguous copy and a split on the device instead of two strided host copies (the reference's map_visibilities takes V complex,
 * statistical_models.py:109).                                                                                            */
int fh_vis_upload_c128(int device, const double *u, const double *v, const double *Vc, const double *w, int64_t n_w, int64_t n,
                       fh_vis **out);
/* The same table stored as five fp32 columns (20 B / visibility; BASELINE configs with fp32 data).  The columns are
 * widened to fp64 as the pre-pass reads them and everything downstream is the fp64 path: the result equals that of
 * fh_vis_upload on the widened values bit for bit (the reference also computes in fp64 whatever dtype it is handed:
 * NumPy promotes in geometry.py:69-79, 111-131).                                                              */
int fh_vis_upload_f32(int device, const float *u, const float *v, const float *Vre, const float *Vim,
                      const float *w, int64_t n_w, int64_t n, fh_vis **out);
void fh_vis_destroy(fh_vis *vis);
int64_t fh_vis_size(const fh_vis *vis);
/* Bootstrap resampling without moving data: counts[i] (n int32, host) = how many times row i was drawn by
 * draw_bootstrap_sample (utilities.py:632-666; counts = bincount(idxs)).  Subsequent fh_bin_visibilities calls weigh
 * row i by counts[i] in M, j, H0 and leave rows with count 0 out of the q range -- the sums over the resampled table,
 * streamed in place instead of gathered.  counts = NULL restores every row once.                                  */
int fh_vis_set_multiplicity(fh_vis *vis, const int32_t *counts);

/* ---- a5-a8: map_visibilities = K1 `bin_gram` ---------------------------------------------------------------
 * statistical_models.py:109-237 with geometry.py:69-79,111-131 and hankel.py:201-202 fused:
 * phase-centre + deproject each visibility, q = hypot(u', v'), J0((q/Qmax) j_k) for all k, and accumulate the
 * Bessel Gram G = X^T diag(w) X, g = X^T diag(w) Re V', sum w V'^2, sum log w, min/max q on the device.
 *   fh_bin_reset      zero the context's sufficient statistics
 *   fh_bin_visibilities  add rows [first, first+count) of `vis`.  The pre-pass reads the baseline range back once (it
 *                     sizes the bucket sort, and it is what _check_uv_range needs before any binning,
 *                     statistical_models.py:166-169); binning the SAME rows of the SAME table under the SAME geometry and the
 *                     SAME multiplicities again (pipelines, sweeps) re-uses that range and the call does not wait for the
 *                     device (fh_ctx_set_range_cache(ctx, 0) or FRANK_AMD_NO_RANGE_CACHE=1 switches this off).
 *                     Default path (bin_prepass.hip): the rows of a J0 bucket enter the Gram through their 13 x 13 moment
 *                     matrix; N <= 1023.  Rows path (debris model, single-precision arithmetic, FRANK_AMD_K1=rows): every
 *                     visibility through the design-block + Gram kernel (N <= 511), rows-to-memory + rocBLAS beyond
 *   fh_stats_device   device pointer / length (doubles) of the packed statistics, for an RCCL all-reduce
 *   fh_stats_finalize apply the DHT scaling, unpack to M (N*N), j (N), H0, qmin, qmax (host, any may be NULL);
 *                     the device copies of M and j stay in the context for fh_fit_normal(M = NULL).  With every output
 *                     NULL and check_qbounds == 0 the call only queues the finalisation and returns (pipelines:
 *                     fh_fit_submit takes M, j on the device).
 *                     Returns FH_ERR_QRANGE iff check_qbounds and q_k[-1] < qmax (outputs are still written). */
int fh_bin_reset(fh_ctx *ctx);
int fh_bin_visibilities(fh_ctx *ctx, const fh_geometry *geom, const fh_vis *vis, int64_t first, int64_t count);
/* Device time (ms, HIP events on the context's stream) of the most recent bin_gram launch alone. */
int fh_bin_last_kernel_ms(fh_ctx *ctx, float *ms);
/* Device time (ms, HIP events) of the binning pass in front of its Gram kernel: the (u, v) histogram + scan when the rows are not
 * the ones binned last, the fused deprojection + bucket sort and the bucket moments (default path), or deprojection + bucket sort
 * (rows path).  Since round 6 WITHOUT the look at the baseline range (the _check_uv_range input, statistical_models.py:166-169):
 * that kernel runs in front of the pass -- on the look-ahead stream after fh_bin_prefetch_range -- and has events of its own:  */
int fh_bin_last_prepass_ms(fh_ctx *ctx, float *ms);
/* ... device time (ms) of the range kernel the most recent binning pass needed; 0 when it took the range from the cache.      */
int fh_bin_last_range_ms(fh_ctx *ctx, float *ms);
/* Duration of the fit_loop kernel of the last fh_fit_normal call, by HIP events on the context's stream (bench.py). */
int fh_fit_last_kernel_ms(fh_ctx *ctx, float *ms);
/* Arithmetic of bin_gram (BASELINE configs[2], "fp32").  fp32 != 0: the Bessel design block and the tile products of the
 * Gram run in single precision on the matrix pipe (v_mfma_f32_16x16x4_f32, twice the fp64 rate), with the argument of
 * J0 still reduced in fp64 (bucket centre + offset) and the single-precision accumulators added into fp64 sums every
 * 1024 visibilities; everything downstream (M, j, the fit) is fp64.  The reference has no such mode (NumPy promotes to
 * fp64, geometry.py:69-79): the brightness profile then agrees with the fp64 path to ~1e-5 of its maximum, inside the
 * 1e-3 BASELINE.json states for fp32.  Default 0 (fp64 arithmetic whatever the storage type of the table).
 * Limited to tables of at most 2e6 visibilities: fh_bin_visibilities returns FH_ERR_UNSUPPORTED beyond (measured: at 1e7 rows the
 * single-precision Gram is no longer positive definite, and the pass is 25 x slower than the fp64 moments pass).  What
 * BASELINE configs[2] calls "fp32" at full size is single-precision STORAGE: fh_vis_upload_f32, 20 B per visibility, binned
 * by the fp64 path.                                                                                                      */
int fh_ctx_set_arithmetic(fh_ctx *ctx, int fp32);
/* Line search of the LogNormal Newton solves (minimizer.py:70-187 evaluates H(x + lam p) afresh for every trial step).
 * The prior precision S^-1 = Y^T diag(1/p) Y has entries ~1/p_0 = 1e35 that cancel in S^-1 x, so a freshly multiplied
 * S^-1 (x + lam p) carries round-off far above that of the objective itself and the Armijo test of the reference mostly
 * compares noise: ~15 trial points per Newton step, most MAP solves end with "neither direction improves".
 *   reference_products == 0 (default): S^-1 (x + lam p) = S^-1 x + lam S^-1 p along a search (S^-1 is linear; one product
 *       per search instead of one per trial).  The round-off of S^-1 x is then the same at every trial point, the
 *       searches accept the Newton step, the solves converge in ~10 steps.  The profile differs from the reference's by
 *       no more than the reference's own fit moves when M is perturbed by 1e-15 (tests/test_gpu_configs.py).
 *   reference_products != 0: every trial point multiplied out, the reference's arithmetic; Newton step and evaluation
 *       counts then track the reference's (tests/golden/lognormal_*.npz), at ~8x the time.                              */
int fh_ctx_set_lognormal_linesearch(fh_ctx *ctx, int reference_products);
/* Work hand-out of bin_gram.  By default a synchronous fit deals contiguous ranges of the sorted table to the workgroups
 * (the sums come out bit for bit the same in every run) and a pipeline of fits (fh_fit_submit outstanding) lets the
 * workgroups pull work from a counter, which is faster while fit loops occupy compute units but makes the last bits
 * depend on the run.  on != 0 forces the first, reproducible hand-out everywhere, as the reference's single-threaded
 * sums are (statistical_models.py:200-214).                                                                          */
int fh_ctx_set_reproducible(fh_ctx *ctx, int on);
/* The baseline range of a (table, row range, geometry, multiplicities) is remembered by the context, so that binning the same
 * rows again needs no second look at (u, v) and no host round trip before the sort is sized (what _check_uv_range computes
 * before the chunk loop, statistical_models.py:166-169).  on = 0 forgets it and measures the range on every pass -- the cost
 * of binning a table the context has not seen (bench.py `distinct_tables`).  Default: on.                                  */
int fh_ctx_set_range_cache(fh_ctx *ctx, int on);
/* Look-ahead for a pipeline that knows its next table (round 6): measure the baseline range of these rows NOW, on a stream of the
 * context's own beside whatever the binning stream is doing, so that the fh_bin_visibilities call that follows for the SAME rows,
 * geometry and multiplicities finds the range on the host when it gets there instead of waiting for the pass in front of it (the
 * one look at (u, v), 16 B per row, and the host round trip of _check_uv_range, statistical_models.py:166-169, are still paid --
 * overlapped).  Ask for the next pass's BEFORE queueing the current pass: its kernel then runs in front of that pass.  Where the
 * bucket count has an upper bound that keeps the launch geometry (N <~ 320) the look also leaves the table's (u, v) histograms
 * for the pass -- ONE look at (u, v) instead of two.  Look-aheads that no pass has taken yet are kept (up to 48; the oldest is
 * dropped beyond); nothing of an EARLIER pass is remembered.  A no-op outside the default (moments) binning path.                                                                                 */
int fh_bin_prefetch_range(fh_ctx *ctx, const fh_geometry *geom, const fh_vis *vis, int64_t first, int64_t count);
/* Pipelines of fits (fh_fit_submit): confine the binning pass to the first bin_cus compute units and the fit loops to the
 * others, so that a fit loop -- one workgroup that holds a compute unit for the whole iteration (radial_fitters.py:765-785) --
 * does not share its unit with the workgroups of the passes that follow.  Before the first fh_fit_submit of the context.     */
int fh_ctx_set_cu_partition(fh_ctx *ctx, int bin_cus);
int fh_stats_device(fh_ctx *ctx, double **sum_stats, int64_t *n_sum, double **minmax_stats);
int fh_stats_finalize(fh_ctx *ctx, const fh_geometry *geom, int vis_model, int check_qbounds, double *M, double *j,
                      double *H0, double *qmin, double *qmax);
/* One-shot convenience with host arrays (what VisibilityMapping.map_visibilities binds to).  With check_qbounds the
 * baseline range is checked BEFORE any binning, as _check_uv_range is in the reference (statistical_models.py:166-169):
 * FH_ERR_QRANGE is returned after the deprojection pre-pass, qmin / qmax are set, M, j, H0 are not.                     */
int fh_map_visibilities(fh_ctx *ctx, const fh_geometry *geom, int vis_model, int check_qbounds, const double *u,
                        const double *v, const double *Vre, const double *Vim, const double *w, int64_t n_w,
                        int64_t n, double *M, double *j, double *H0, double *qmin, double *qmax);
/* The same with the visibilities as ONE complex128 array (n (re, im) pairs, as NumPy holds `V`): fh_vis_upload_c128 underneath. */
int fh_map_visibilities_c128(fh_ctx *ctx, const fh_geometry *geom, int vis_model, int check_qbounds, const double *u,
                             const double *v, const double *Vc, const double *w, int64_t n_w, int64_t n, double *M, double *j,
                             double *H0, double *qmin, double *qmax);

/* ---- a11-a13: GaussianModel -------------------------------------------------------------------------------
 * statistical_models.py:650-781.  p may be NULL (no prior).  M, j, p: host; M = j = NULL: the statistics a preceding
 * fh_stats_finalize(..., M = NULL, j = NULL, ...) left on the device.  Outputs (host, any may be NULL):
 * mu (N), chol (N*N, upper factor U with Dinv = U^T U in the upper triangle, as scipy.linalg.cho_factor),
 * Sinv (N*N).  *used_svd is set when the Cholesky failed and the SVD pseudo-inverse (:747-755) was used.    */
int fh_gaussian_model(fh_ctx *ctx, const double *M, const double *j, const double *p, double *mu, double *chol,
                      double *Sinv, int *used_svd);
/* Dsolve(b) with a previously returned factor (statistical_models.py:762-781). B: N*nrhs row-major, in place. */
int fh_cho_solve(fh_ctx *ctx, const double *chol, double *B, int nrhs);

/* The reference's route when cho_factor raises (statistical_models.py:747-755, 779-781, 1150-1158, 1181-1182):
 * U, s, V = svd(A); X = V^T diag(where(s > 0, 1/s, 0)) U^T B, on the device (rocSOLVER gesvd + rocBLAS).
 * A: N*N row-major host; B: N*nrhs row-major host, overwritten with X.                                          */
int fh_svd_solve(fh_ctx *ctx, const double *A, double *B, int nrhs);
/* Dsolve(b) of a posterior whose Cholesky failed, EXACTLY as the reference evaluates it (statistical_models.py:779-781,
 * 1181-1182): np.dot(V.T, np.multiply(np.dot(U.T, b), s1)).  NumPy broadcasts s1 over the LAST axis: for a vector b
 * (nrhs = 1) this is fh_svd_solve; for the N x N right-hand side of update_power_spectrum (filter.py:168) it scales
 * right-hand side c by s1[c] instead of singular direction k by s1[k] -- and the reference's iteration runs on that.
 * Other widths cannot be broadcast: FH_ERR_INVALID (the reference raises ValueError).                            */
int fh_svd_solve_as_reference(fh_ctx *ctx, const double *A, double *B, int nrhs);

/* ---- a10/a14/a15/a16: the power-spectrum iteration = K2 `fit_iterate` --------------------------------------
 * FrankFitter._fit, method='Normal' (radial_fitters.py:737-832) with CriticalFilter.update_power_spectrum /
 * check_convergence (filter.py:154-181), spectral_smoothing_matrix (filter.py:23-62) and GaussianModel
 * (statistical_models.py:700-760), entirely on the device.
 * M, j host (N*N, N) or both NULL to use the context's device copies from fh_stats_finalize.
 * Outputs (host): mu (N), p (N), *niter = `count` at loop exit (caller applies the convergence_failure policy:
 * success iff niter < max_iter, radial_fitters.py:788).  diag_p / diag_mu: NULL or (max_iter+1)*N receiving pI /
 * MAP of every loop pass (:781-783).                                                                        */
int fh_fit_normal(fh_ctx *ctx, const double *M, const double *j, double alpha, double p0, double wsmooth,
                  double tol, int max_iter, double *mu, double *p, int *niter, double *diag_p, double *diag_mu);

/* Pipelined form for independent fits (hyper-parameter sweeps, bootstraps, many sources -- fit.py:534-548,
 * 770-782 call the fitter in a plain loop): fh_fit_submit stages the iteration of the context's device-resident
 * M, j (from fh_stats_finalize) in one of the context's fit slots and returns at once; a fit_loop workgroup occupies
 * ONE compute unit, so the following fh_bin_visibilities calls overlap with it.  Submissions are launched in batches
 * (up to 64 fit loops per launch by default, the launches taking turns on four streams; a launch carries one tol /
 * max_iter, alpha, p0 and w_smooth are per fit):
 * fh_fit_flush launches what has been staged so far -- call it after the last submission; fh_fit_collect on a fit
 * whose launch is still open does the same.  fh_fit_collect waits for that fit and returns mu, p, niter exactly as
 * fh_fit_normal does.  Tickets are collected in any order; at most fh_fit_slots() fits may be outstanding.
 * A caller whose host thread is busy between submissions (uploading the next table: bench.py extra.from_host_pipelined) should
 * flush every few submissions: a collect that meets a fit still staged launches it and then waits a whole fit for it, and a context
 * runs about four launches at a time (one fit per launch = four fits in flight).  fh_vis_destroy of a table waits for the streams
 * that read tables (binning, look-ahead), not for the fit loops in flight.                                            */
/* The packed statistics of the last binning pass (what fh_comm_allreduce_stats reduces in place on the device: n doubles, n from
 * fh_stats_device; minmax = (-qmin, qmax), NaN where nothing was binned) copied to / replaced from the host: for a reduction over
 * ranks that does not go through RCCL (two ranks on one device, any torch.distributed backend -- frank_amd.distributed.HostComm);
 * the sums are those of statistical_models.py:210-211, 218.                                                                  */
int fh_stats_get_packed(fh_ctx *ctx, double *sum_stats, int64_t n, double *minmax);
int fh_stats_set_packed(fh_ctx *ctx, const double *sum_stats, int64_t n, const double *minmax);
/* fh_stats_upload: M (N*N, row-major) and j (N) from the HOST become the context's device-resident normal equations -- what
 * fh_stats_finalize leaves there after a binning pass -- so that fh_fit_submit / fh_fit_normal(ctx, NULL, NULL, ..) can run on
 * statistics computed elsewhere (a saved mapping, the sum of several tables: statistical_models.py:220-237 hands them around as
 * a dict).                                                                                                           */
int fh_stats_upload(fh_ctx *ctx, const double *M, const double *j);
int fh_fit_slots(void);
int fh_fit_submit(fh_ctx *ctx, double alpha, double p0, double wsmooth, double tol, int max_iter, int *ticket);
int fh_fit_flush(fh_ctx *ctx);
int fh_fit_collect(fh_ctx *ctx, int ticket, double *mu, double *p, int *niter);
/* Posterior extras of a whole sweep, batched on the device: for each of `batch` points (p, mu: batch x N, the MAP power spectra
 * and brightness profiles fh_fit_normal_batched returned; alpha, p0, wsmooth: their hyper-parameters) over ONE mapping (M, j from
 * the host, or NULL, NULL = the context's device-resident statistics; H0 its null likelihood)
 *   sol_log_likelihood[b] = GaussianModel.log_likelihood()        1/2 j.mu + 1/2 log det(D S^-1) + H0   statistical_models.py:836-841
 *   log_prior[b]          = CriticalFilter.log_prior(p)                                                filter.py:253-261
 *   log_evidence[b]       = FrankFitter.log_evidence_laplace()    log P(p, V) - 1/2 log det(H / 2 pi)  radial_fitters.py:951-967
 *   pscov_diag[b][N]      = diag of CriticalFilter.covariance_MAP (H^-1)                               filter.py:184-227
 * (any output may be NULL; FrankFitter.log_likelihood() is log_prior + sol_log_likelihood).  The reference forms Y D Y^T and the
 * Hessian H with dense host products per point; here Y D Y^T = (A + diag(1/p))^-1 in the basis of the fit loop, and a point is
 * two Cholesky factorisations and inversions of N x N matrices in rocSOLVER's strided-batched routines, 128 points at a time --
 * ranking a 512-point sweep by evidence (what fit.py:534-548 exists for) without 512 host O(N^3) passes.  A point whose
 * Hessian is not positive definite gets NaN evidence.                                                                  */
int fh_sweep_evidence(fh_ctx *ctx, const double *M, const double *j, double H0, int batch, const double *p, const double *mu,
                      const double *alpha, const double *p0, const double *wsmooth, double *sol_log_likelihood,
                      double *log_prior, double *log_evidence, double *pscov_diag);

/* Cluster ("latency") mode of the fit loop.  A fit whose pipeline is shallow -- fh_fit_normal on an idle context, the first
 * launches of a pipeline -- runs on `workgroups` compute units of one XCD instead of one: the first factors the posterior
 * precision and runs the loop (GaussianModel._fit, statistical_models.py:732-760; filter.py:154-181), the others form the
 * block columns of the inverse of the factor that Tr2 and the mean need (one wave per column); the arithmetic of every tile
 * is the one the single-workgroup kernel does, so the results are the same bits.  FRANK_AMD_K2_CLUSTER = 1 turns the mode
 * off, 2 .. 8 set the size (default 5: two helpers of the inverse, two of the trailing update; 3 for N > 335).  *workgroups: what the last fh_fit_normal ran on; *fallbacks: cluster launches of this
 * context that did not assemble on one XCD within 3 ms and were repeated on one compute unit (either may be NULL).     */
int fh_fit_cluster_info(fh_ctx *ctx, int *workgroups, int64_t *fallbacks);

/* The columns of the visibility tables (fh_vis_upload, fh_vis_upload_c128; fh_map_visibilities underneath) come from a cache of
 * freed allocations -- exact size, per device, at most 1.5 GB held -- so that a caller that maps one table after another
 * (VisibilityMapping.map_visibilities, fit.py:455-471) does not pay six hipMalloc + six hipFree per call.  This empties it.   */
int fh_cache_release(void);

/* The development switches of the binning pass (FRANK_AMD_K1_*, FRANK_AMD_NO_RANGE_CACHE) are read from the environment ONCE,
 * when a context is created; this reads them again (tests that switch them inside one process).  It also makes the context
 * forget that its pipeline has once held 128 fits (from then on its one-unit fit loops keep the form that is faster on a loaded
 * device, fit_loop_rr.hip; FRANK_AMD_K2_RR, read at every launch, overrides the choice either way).                        */
int fh_ctx_reload_env(fh_ctx *ctx);

/* Measurement aid (no counterpart in the reference): the clock the fit loops of this context ran at.  on != 0 switches a
 * probe on -- every fit loop then adds its shader-clock cycles, its ticks of the constant 100 MHz wall clock and its passes
 * (posterior solves) to three device counters --, on == 0 off; out3 (may be NULL) receives the sums since the last call and
 * clears them: mean clock = 100 MHz x out3[0] / out3[1], mean pass = out3[1] / 100 / out3[2] microseconds.                  */
int fh_ctx_loop_clocks(fh_ctx *ctx, int on, int64_t *out3);

/* Batched form for hyper-parameter sweeps over ONE mapping (fit.py:534-548 re-runs the whole fit per (alpha,
 * w_smooth) point although M, j do not depend on them): `batch` fits of the same M, j (host, or NULL for the
 * context's device copies) with per-fit alpha[b], p0[b], wsmooth[b]; one fit_loop workgroup (one CU) per fit, all in
 * one launch.  Outputs: mu, p (batch*N, row per fit), niter (batch), status (batch: FH_OK / FH_ERR_BAD_P /
 * FH_ERR_NOT_SPD per fit).                                                                                      */
int fh_fit_normal_batched(fh_ctx *ctx, const double *M, const double *j, int batch, const double *alpha,
                          const double *p0, const double *wsmooth, double tol, int max_iter, double *mu, double *p,
                          int *niter, int *status);

/* One pass of the loop body for a caller-supplied p: fit = GaussianModel(M, j, p); p_new =
 * CriticalFilter.update_power_spectrum(fit) (filter.py:154-177).  M, j, p host; mu (N, posterior mean for p) and
 * p_new (N) host outputs, either may be NULL.                                                                */
int fh_update_power_spectrum(fh_ctx *ctx, const double *M, const double *j, const double *p, double alpha, double p0,
                             double wsmooth, double *mu, double *p_new);

/* ---- method='LogNormal' (radial_fitters.py:754-763, statistical_models.py:907-1160, minimizer.py) -------------
 * fh_lognormal_model: LogNormalMAPModel(DHT, M, j, p, guess=guess, s0=s0) for one field / one frequency: the MAP of
 * s = log(I) - s0 by MinimizeNewton(H, jac, hess, guess, LineSearch(reduce_step=limit_step), tol=1e-7)
 * (statistical_models.py:1064-1145, minimizer.py:190-283).  M, j host (or NULL, NULL for the context's device
 * copies); p, guess (N) host.  Outputs (host): s_map (N); Dinv (N*N row-major, hess(s_MAP), statistical_models.py:
 * 1147; may be NULL); stats (9 x int64, may be NULL): MAP solves, Newton steps, function evaluations, Hessian
 * factorisations, then the number of MinimizeNewton exits with status 0 (converged), 1 (no improvement), 2 (max
 * steps), 3 (max Hessians), 4 (slope round-off -> FH_ERR_NUMERIC).  As in the reference the exit status of the
 * minimiser is otherwise ignored (statistical_models.py:1142-1145).  N <= 320: one persistent kernel (lognormal.hip);
 * 320 < N <= 1023: the minimiser's control flow on the host over device kernels (lognormal_wide.hip), the reference's
 * line search; the same holds for fh_fit_lognormal, fh_fit_lognormal_batched and fh_posterior_update.           */
int fh_lognormal_model(fh_ctx *ctx, const double *M, const double *j, const double *p, const double *guess, double s0,
                       double *s_map, double *Dinv, int64_t *stats);

/* fh_fit_lognormal: FrankFitter._fit with method='LogNormal' (radial_fitters.py:737-832): the two Normal seed fits,
 * the log-space seed (:756-763), then `while not converged and count <= max_iter` of LogNormalMAPModel +
 * CriticalFilter.update_power_spectrum, all device-resident.  I_scale as FrankFitter(I_scale=...) (:712).
 * Outputs (host): s_map (N; I = exp(s_map + log I_scale), radial_fitters.py:392), p (N), niter (`count` at exit),
 * Dinv (N*N, Hessian at the final MAP, may be NULL), stats (9 x int64 as above, may be NULL), diag_p / diag_s
 * ((max_iter+1)*N each, both or neither): p and s of every pass (store_iteration_diagnostics).                  */
int fh_fit_lognormal(fh_ctx *ctx, const double *M, const double *j, double alpha, double p0, double wsmooth,
                     double tol, int max_iter, double I_scale, double *s_map, double *p, int *niter, double *Dinv,
                     int64_t *stats, double *diag_p, double *diag_s);

/* Batched form for hyper-parameter sweeps of LogNormal fits over ONE mapping (fit.py:534-548): `batch` fits with
 * per-fit alpha[b], p0[b], wsmooth[b]; the Normal seed fits are shared; one lognormal workgroup (one CU) per fit, all
 * in one launch.  Outputs: s_map, p (batch*N, row per fit), niter (batch), status (batch: FH_OK / FH_ERR_BAD_P /
 * FH_ERR_NUMERIC; may be NULL), stats (batch*9 int64 as in fh_lognormal_model; may be NULL).                    */
int fh_fit_lognormal_batched(fh_ctx *ctx, const double *M, const double *j, int batch, const double *alpha,
                             const double *p0, const double *wsmooth, double tol, int max_iter, double I_scale,
                             double *s_map, double *p, int *niter, int *status, int64_t *stats);

/* fh_posterior_update: CriticalFilter.update_power_spectrum(fit) (filter.py:154-177) for ANY posterior object the
 * caller holds: map = fit.MAP (N), Dinv = the posterior precision (N*N row-major; fit.Dsolve applies its inverse),
 * p = fit.power_spectrum.  The inverse is applied through a partial-pivoting LU on the device.  Output p_new (N). */
int fh_posterior_update(fh_ctx *ctx, const double *map, const double *Dinv, const double *p, double alpha, double p0,
                        double wsmooth, double *p_new);

/* ---- geometry fits (geometry.py:404-763): the residual functions an optimiser calls, on the resident table ---------
 * The reference fits (inc, PA, dRA, dDec) with scipy.optimize.least_squares(method='lm') over a residual function that
 * is evaluated on the whole table at every step.  These two entry points are those functions; the optimiser stays where it
 * is (frank_amd.geometry hands them to the same SciPy routine).
 *
 * fh_vis_residuals: FitGeometryFourierBessel._residual (geometry.py:660-694) after its FBF.fit -- which is
 *   fh_bin_reset / fh_bin_visibilities / fh_stats_finalize / fh_gaussian_model under the trial geometry --:
 *   out[i] = sqrt(w_i) Re(Vm_i - V_i), out[count + i] = sqrt(w_i) Im(Vm_i - V_i), Vm = sol.predict(u, v)
 *   (radial_fitters.py:56-98: deproject, H(q) I, x cos(inc) for vis_model 0 ('opt_thick'), exp(-kz^2 H2[k]) per column for
 *   2 ('debris', fh_ctx_set_scale_height), re-phased by the phase centre).  I: N host doubles.  out (host, 2 count
 *   doubles) and sumsq (the sum of squares of out) may each be NULL.
 * fh_gauss_residuals: _gauss_fun / _gauss_jac of _fit_geometry_gaussian (geometry.py:535-585).  params = (inc [rad],
 *   PA [rad], dRA [arcsec], dDec [arcsec], norm, scal) as the optimiser holds them (with a given phase centre pass it in
 *   params[2..3] and fit_phase = 0: it is applied, its Jacobian columns are zero; fit_inc_pa = 0 zeroes columns 0, 1).
 *   fun: 2 n host doubles (real parts, then imaginary parts) or NULL; jac: [2 n][6] row-major host doubles or NULL.   */
int fh_vis_residuals(fh_ctx *ctx, const fh_geometry *g, int vis_model, const fh_vis *vis, int64_t first, int64_t count,
                     const double *I, double *out, double *sumsq);
int fh_gauss_residuals(const fh_vis *vis, const double *params, int fit_inc_pa, int fit_phase, double *fun, double *jac,
                       double *sumsq);

/* fh_predict_sky: FrankRadialFit.predict(u, v, I, geometry) (radial_fitters.py:56-98) in one pass: the sky-plane baselines are
 *   deprojected, V = H(q) I (x cos(inc) for vis_model 0, exp(-kz^2 H2[k]) per column for 2), and re-phased by the phase centre.
 *   u, v: n host doubles; Vre, Vim: n host doubles each (the complex model visibilities). */
int fh_predict_sky(fh_ctx *ctx, const fh_geometry *g, int vis_model, const double *u, const double *v, int64_t n, const double *I,
                   double *Vre, double *Vim);

/* The same fits with nothing of size n leaving the device: Levenberg-Marquardt needs the residual norm of a trial point and,
 * at an accepted point, J^T J and J^T r -- a few doubles (frank_amd.geometry, optimizer='device': MINPACK's lmdif / lmder
 * algorithm on the normal equations).  For the reference's optimiser itself use the two entry points above.
 * fh_vis_residuals_slot: as fh_vis_residuals over the whole table, the vector written to one of FH_RESIDUAL_SLOTS device
 *   buffers of 2 n doubles owned by the table; only its sum of squares is returned.  I == NULL: the profile the context's
 *   last solve left on the device (fh_gaussian_model with M = j = NULL after fh_stats_finalize with M = j = NULL keeps the
 *   whole evaluation -- statistics, solve, residuals -- on the device).
 * fh_residual_normal_equations: forward-difference Jacobian columns d_k = (slot col_slots[k] - slot base_slot) / h[k],
 *   k < ncol <= 4 (MINPACK fdjac2), reduced to JtJ [ncol x ncol] and Jtr = J^T r(base) [ncol].
 * fh_gauss_normal_equations: JtJ [6 x 6], Jtr [6] and the sum of squares of the Gaussian's residual with its analytic
 *   Jacobian (arguments as fh_gauss_residuals).                                                                        */
#define FH_RESIDUAL_SLOTS 8
int fh_vis_residuals_slot(fh_ctx *ctx, const fh_geometry *g, int vis_model, const fh_vis *vis, const double *I, int slot,
                          double *sumsq);
int fh_residual_normal_equations(fh_ctx *ctx, const fh_vis *vis, int base_slot, int ncol, const int *col_slots,
                                 const double *h, double *JtJ, double *Jtr);
int fh_gauss_normal_equations(const fh_vis *vis, const double *params, int fit_inc_pa, int fit_phase, double *JtJ, double *Jtr,
                              double *sumsq);

/* ---- utilities.UVDataBinner (utilities.py:180-400): uv-data averaged in bins of equal width ---------------------
 * fh_uvbin_create: UVDataBinner(uv, V, weights, bin_width): uv, Vre, Vim (NULL for real V), w: n host doubles.
 *   nbins = ceil(max(uv) / bin_width) (+1 under the rounding guard of :206-208); per bin the weighted means of uv
 *   and V, the summed weight, the number of rows, and the error of the mean (NaN for bins with fewer than two rows,
 *   as the reference leaves it: see the note at :256-261 in DESIGN.md).  Bin indices and counts are bit-exact.
 * fh_uvbin_get: copies out nbins entries of each (any pointer may be NULL); empty bins hold 0 sums / NaN errors and
 *   count 0 (the Python class masks them).
 * fh_uvbin_determine: determine_uv_bin(uv) (:271-298): bin of each baseline, -1 exactly past the last edge;
 *   FH_ERR_INVALID where the reference raises IndexError (baseline >= (nbins + 1) * bin_width).
 * fh_uvbin_quantities: bin_quantities(uv, w, qty[, bin_counts]) (:300-366) for one real or complex quantity.   */
typedef struct fh_uvbin fh_uvbin;
int fh_uvbin_create(int device, const double *uv, const double *Vre, const double *Vim, const double *w, int64_t n,
                    double bin_width, fh_uvbin **out);
void fh_uvbin_destroy(fh_uvbin *h);
int fh_uvbin_nbins(const fh_uvbin *h);
float fh_uvbin_kernel_ms(const fh_uvbin *h); /* HIP-event time of the three streaming passes of fh_uvbin_create */
int fh_uvbin_get(const fh_uvbin *h, double *uv, double *Vre, double *Vim, double *w, int64_t *count, double *err_re,
                 double *err_im);
int fh_uvbin_determine(fh_uvbin *h, const double *uv, int64_t n, int32_t *idx);
int fh_uvbin_quantities(fh_uvbin *h, const double *uv, const double *w, const double *qre, const double *qim, int64_t n,
                        double *out_re, double *out_im, int64_t *counts);

/* vis_model='debris' (statistical_models.py:96-102, 494-496): H2[k] = 0.5 * (2 pi scale_height(r_k) / rad_to_arcsec)^2,
 * N host doubles.  While set, fh_bin_visibilities scales each row by exp(-kz_i^2 H2[k]) (kz = the vertical uv-distance
 * of the 3-D deprojection, geometry.py:128): on the fused rows kernel for N <= 511 (the generated design block is scaled in
 * registers; the bucket moments do not apply, the factor does not split into row x column), through rows-to-memory + rocBLAS
 * beyond; fh_stats_finalize / fh_map_visibilities must then be called with FH_VIS_DEBRIS.  H2 = NULL switches back.          */
int fh_ctx_set_scale_height(fh_ctx *ctx, const double *H2);

/* ---- multi-GPU: RCCL all-reduce of the sufficient statistics (one rank per GPU) ------------------------------
 * The reduction being distributed is `Ms[i] += ...; js[i] += ...` (statistical_models.py:210-211) and the
 * sum at :218; min/max q feed _check_uv_range (:512-535).                                                    */
int fh_comm_unique_id(char id[128]);
int fh_comm_create(const char id[128], int rank, int world, int device, fh_comm **out);
void fh_comm_destroy(fh_comm *comm);
/* Sums the context's statistics over the ranks in place, asynchronously on the context's stream: the packed tile
 * triangle (the fused paths: N <= 1023 by default) or the dense (N+1)^2 Gram of the rows + rocBLAS path, each with its two
 * scalars, plus a 2-double max-reduce of (-qmin, qmax).  fh_stats_finalize afterwards yields the unsharded M, j, H0. */
int fh_comm_allreduce_stats(fh_comm *comm, fh_ctx *ctx);
/* Device time (ms, HIP events on the context's stream) of the most recent fh_comm_allreduce_stats. */
int fh_comm_last_allreduce_ms(fh_comm *comm, float *ms);
int fh_comm_size(const fh_comm *comm); /* number of ranks */

#ifdef __cplusplus
}
#endif
#endif /* FRANK_HIP_H */
