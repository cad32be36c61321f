// Internal launch interface between the C-ABI glue (capi_*.hip, capi_internal.h) and the kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct BinParams {
    // visibility columns (device), rows [first, first+count)
    const double *u, *v, *Vre, *Vim, *w;
    // the same columns stored as fp32 (fh_vis_upload_f32: 20 B / visibility, widened to fp64 as they are read);
    // non-NULL u32 selects them
    const float *u32, *v32, *Vre32, *Vim32, *w32;
    const int *mult;  // optional per-row multiplicity (bootstrap resampling), NULL = 1
    int w_scalar;
    int64_t first, count;
    // geometry (geometry.py:69-70,111-115): dRA, dDec already multiplied by 2 pi / rad_to_arcsec
    double dRA, dDec, cos_t, sin_t, cos_i, sin_i;
    // DHT
    int N;
    double inv_Qmax;        // k = 1./Qmax, hankel.py:189
    const double *zeros;    // j_k, N entries (device)
    const double *j0_table; // FH_J0_TAYLOR (device)
    // outputs
    // K1a output (device, `count` entries each): s = q/Qmax, sqrt(w), sqrt(w) Re V'
    double *prep_s, *prep_sw, *prep_swV;
    // debris model only: kz^2 per visibility (K1a output) and H2[k] = 0.5 (2 pi H(r_k))^2 (device); NULL otherwise
    double *prep_k2;
    const double *H2;
    // throughput mode: two ints (one per part) handing out super-chunks; NULL = static, reproducible split
    int *work_counter;
    // grid split between the tile parts (bin_gram.hip): part_blocks[0] + part_blocks[1] workgroups
    int part_blocks[2];
    double *partials[2];     // per part: [part_blocks][part_ntiles][256]
    double *partial_scalars; // [deproject blocks][4]  (sum log(w/2pi), qmin, qmax, -)
};

struct ReduceParams {
    int nparts, ntiles, scalar_blocks;
    int part_blocks[3], part_tile0[3], part_ntiles[3];
    const double *partials[3];
    const double *partial_scalars;
    double *scratch;  // 8 * ntiles * 256 doubles: level-1 sums of the slab reduction
};

// ---- K1 v2 (bin_gram2.hip): bucket sort + design block generated on the matrix pipe ------------------------------
struct SortParams {
    const double *s, *sw, *swV;  // K1a output, n rows
    const double *k2;            // debris model: kz^2 of every row (K1a output), else NULL
    int64_t n;
    double inv_delta, delta;     // bucket width in s (j0_buckets.h)
    int nb, blocks;              // buckets; workgroups of the histogram / scatter passes
    int *hist;                   // [blocks][nb]
    int *totals;                 // [nb]
    int *starts;                 // [nb + 1]
    int *info;                   // [0] = number of 16-row chunks of the sorted table
    double *rows;                // sorted table, 4 doubles per row (tau, sqrt(w), sqrt(w) Re V', -), n + 16 nb rows
    int *chunk_bucket;           // bucket of every chunk
};
struct Bin2Params {
    int N;
    const double *rows;
    const int *chunk_bucket;
    const int *info;
    const double *table;         // [bucket][12][xstride]: Taylor tables (fh_k1_bucket_table)
    const float *table32;        // the same tables rounded to fp32; non-NULL selects the single-precision kernel
    const double *H2;            // debris model: H2[k] (N doubles); non-NULL selects the kernel that scales by exp(-kz^2 H2[k])
    int *work_counter;           // NULL = static contiguous ranges (bitwise reproducible); else one int per part
    int virtual_rows;            // rows = the compressed rows of fh_k1v2_launch_compress: [chunk][16][16] doubles, P[0..11] and
                                 // the data column at [12]; chunk_bucket = bucket of each compressed chunk, info[0] = their number
    int part_blocks[3];
    double *partials[3];         // per part: [part_blocks][part_ntiles][256]
};
int fh_k1v2_nbt_for(int N);
int fh_k1v2_xstride(int NBT);
int fh_k1v2_ntiles(int NBT);
int fh_k1v2_nparts(int NBT);
int fh_k1v2_part_tile0(int NBT, int P);
int fh_k1v2_part_ntiles(int NBT, int P);
int fh_k1v2_part_block0(int NBT, int P);
hipError_t fh_k1v2_launch_sort(const SortParams &sp, hipStream_t stream);
hipError_t fh_k1v2_launch_bin(int NBT, const Bin2Params &p, hipStream_t stream);
// Bucket compression (bin_gram2.hip): one 16-row chunk per non-empty bucket.
struct CompressParams {
    const double *rows;   // sorted rows (Row32), bucket starts aligned to 16 rows
    const int *starts, *totals;
    int nb;
    int *cidx;            // [nb] index of the bucket among the non-empty ones
    int *info;            // info[1] = number of non-empty buckets
    double *vrows;        // [non-empty bucket][16][16]
    int *vbucket;         // [non-empty bucket]
    int parts;            // slices per bucket of the moment sums
    double *partial;      // [nb][parts][fh_k1v2_moment_doubles()]
};
// ---- pre-pass of the moments path (bin_prepass.hip) -------------------------------------------------------------------
struct PrepassParams {
    BinParams bin;            // table columns, row range, geometry, 1/Qmax, multiplicities
    double inv_delta, delta;  // bucket width in s (j0_buckets.h)
    int nb;                   // buckets
    int wpb, blocks;          // waves per workgroup and workgroups of P1 / P2 (fh_prepass_geometry)
    int unroll;               // rows per lane and tile in P1 / P2 (1 or 2): a tile is 64 x wpb x unroll rows
    int seg_rows;             // rows per segment of the sorted table in P3 (a multiple of 128)
    int safe_trig;            // phases u dRA + v dDec beyond 1e5 rad may occur: P2 takes the library's sincos
    int64_t dummy_row;        // a row behind the sorted table that lanes past the end of the visibility table write to
    int *hist;                // [nb][hist_stride] rows per bucket and workgroup (zero beyond `blocks`), then (scan) in earlier workgroups
    int hist_stride;          // blocks rounded up to a multiple of 256 (four counts per lane and 16-byte alignment in the scan)
    int *totals;              // [nb]
    int *starts;              // [nb + 1] first sorted row of every bucket (multiples of 16)
    int *cidx;                // [nb] index of the bucket among the non-empty ones
    int *info;                // [0] 16-row chunks of the sorted table, [1] non-empty buckets, [2] segments, [3] ticket (zero)
    int *piece0;              // [nb + 1] first slot of every bucket's partial moments
    double *rows;             // sorted table, 3 doubles per row (tau, sqrt(w), sqrt(w) Re V'), count + 16 nb + 16 rows
    double *partial;          // [pieces][fh_prepass_moment_doubles()]
    double *vrows;            // [non-empty bucket][16][16]: the input of bin_gram2_kernel<.., VR = true>
    int *vbucket;             // [non-empty bucket]
    double *partial_scalars;  // [workgroup][4]: sum log(w / 2 pi), qmin, qmax (rows of multiplicity > 0), qmax (all rows)
    int hist_zeroed;          // P1 of a one-look pass: `hist` was cleared in front of the kernel, only the non-zero counts are stored
    int fused;                // bin_fused.hip: every bucket's moments come from `partial` (piece0[b] = b x workgroups), no sorted rows
};
// Gram of the virtual rows, one workgroup (x split) per output tile (bin_prepass.hip)
struct VrGramParams {
    int N, NBT, XS, ntiles;
    int split, waves;         // workgroups per tile (<= 8: scratch), waves per workgroup (<= 16): the chunks are dealt to split x waves
    const double *vrows;      // [chunk][16][16]
    const int *vbucket;       // [chunk]
    const int *info;          // info[0] = chunks
    const double *table;      // [bucket][12][XS]
    double *scratch;          // [split][ntiles * 256]
    const double *partial_scalars;
    int scalar_blocks;
    int fresh;                // 1: the sums start with this launch (store, do not add: fh_bin_reset's fills were deferred)
};
hipError_t fh_vr_gram_launch(const VrGramParams &G, double *stats_sum, double *stats_minmax, hipStream_t stream);
int fh_prepass_moment_doubles();
void fh_prepass_geometry(int nb, int num_cu, int *wpb, int *blocks);
int64_t fh_prepass_max_pieces(int64_t count, int nb, int seg_rows);
hipError_t fh_prepass_launch_range(const PrepassParams &P, hipStream_t stream);  // baseline range only (partial_scalars)
// P1, scan, P2, P3, factor; skip_hist = 1: the histograms, their scan and the layout of the last pass still describe these rows;
// 2: the histograms of these rows exist (fh_prepass_launch_look), scan and layout do not
hipError_t fh_prepass_launch(const PrepassParams &P, hipStream_t stream, int skip_hist = 0);
// the fused form (bin_fused.hip): P1 + scan as above (skip_hist), then the layout of the accumulator slots and ONE pass over the table
hipError_t fh_prepass_launch_hist(const PrepassParams &P, hipStream_t stream);
hipError_t fh_prepass_launch_zero(int *p, size_t n, hipStream_t stream);
hipError_t fh_prepass_launch_look(const PrepassParams &P, hipStream_t stream);  // P1 alone, P.nb an upper bound: range + histograms in one look
hipError_t fh_prepass_launch_scan(const PrepassParams &P, hipStream_t stream);  // the scan + layout over histograms that exist
hipError_t fh_prepass_launch_factor(const PrepassParams &P, hipStream_t stream);
int fh_fused_slot_doubles();
int fh_fused_max_slots(int nb);
hipError_t fh_fused_launch_layout(const PrepassParams &P, int G, int max_slots, int *slot_tab, int *nslots_out, hipStream_t stream);
hipError_t fh_fused_launch(const PrepassParams &P, const int *slot_tab, int max_slots, int G, int scalar_blocks, hipStream_t stream);

int fh_k1v2_moment_doubles();
hipError_t fh_k1v2_launch_compress(const CompressParams &cp, hipStream_t stream);
hipError_t fh_k1v2_launch_max(const double *q, int64_t n, double *out, hipStream_t stream);
// the Taylor tables of the buckets b0 <= b < b1 built on the device from host seeds (j0_buckets_device.hip)
int fh_k1_seed_stride();
int fh_k1_seed_chains(int b0, int b1);   // chains of the range [b0, b1) ...
int fh_k1_seed_bucket(int b0, int c);    // ... and the bucket the seeds of chain c belong to
hipError_t fh_k1_bucket_table_device(const double *zeros_dev, int N, int XS, int b0, int b1, double Delta, const double *seeds_dev,
                                     double *table_dev, hipStream_t stream);
hipError_t fh_k1v2_launch_predict_coef(const double *table, int XS, int N, int nb, const double *pref, const double *I, double scale,
                                       double *coef, hipStream_t stream);
hipError_t fh_k1v2_launch_predict(const double *table, int XS, int N, int nb, const double *pref, const double *I, double scale,
                                  double *coef, const double *q, int64_t n, double inv_Q, double delta, double *V,
                                  hipStream_t stream);

int fh_k1_nbt_for(int N);
int fh_k1_ntiles(int NBT);
int fh_k1_nparts(int NBT);
int fh_k1_part_tile0(int NBT, int P);
int fh_k1_part_ntiles(int NBT, int P);
int fh_k1_super();
hipError_t fh_k1_launch_deproject(const BinParams &p, int blocks, hipStream_t stream);
hipError_t fh_k1_launch_bin(int NBT, const BinParams &p, hipStream_t stream);
hipError_t fh_k1_launch_reduce(const ReduceParams &rp, double *stats_sum, double *stats_minmax, hipStream_t stream);
hipError_t fh_k1_launch_wide_rows(const BinParams &p, int64_t first, int64_t rows, double *X, hipStream_t stream);
hipError_t fh_k1_launch_wide_scalars(const double *partial_scalars, int blocks, double *tail, double *stats_minmax,
                                     hipStream_t stream);
hipError_t fh_k1_launch_wide_finalize(const double *G, int N, const double *a, double *M, double *j, double *sumwV2,
                                      hipStream_t stream);
hipError_t fh_k1_launch_finalize(const double *stats_sum, int NBT, int N, const double *a, double *M, double *j,
                                 double *sumwV2, hipStream_t stream);
hipError_t fh_k1_launch_coefficients(const double *q, int64_t n, int N, const double *zeros, const double *pref,
                                     double inv_Q, double scale, const double *j0_table, double *H,
                                     hipStream_t stream);
hipError_t fh_k1_launch_predict(const double *q, int64_t n, int N, const double *zeros, const double *pref,
                                double inv_Q, double scale, const double *I, const double *j0_table, double *V,
                                hipStream_t stream);

// ---- K2 ------------------------------------------------------------------------------------------------
#define FIT_MAX_N 1024
enum { FIT_FLAG_DONE = 0, FIT_FLAG_COUNT = 1, FIT_FLAG_BAD_P = 2, FIT_FLAG_NOT_SPD = 3, FIT_FLAG_INFO = 4, FIT_NFLAGS = 8 };

struct FitState {
    int N, max_iter;
    double alpha, p0, tol, transform_norm;  // transform_norm = 2 pi Rmax^2 / j_nN (hankel.py:155)
    const double *Y, *Ykm, *q;              // DHT.coefficients(), _Ykm, collocation q (device)
    const double *M, *j;                    // normal equations (device)
    const double *band_lu;                  // 5*N: LU factors of the pentadiagonal T + I (host-factorised)
    double *W, *D, *Z;                      // N*N work: diag(1/p) Y, Dinv -> Cholesky factor, triangular-solve rhs
    double *p, *p_old, *mu;
    int *flags;                             // FIT_NFLAGS
    int *info;                              // rocSOLVER potrf info
    double *diag_p, *diag_mu;               // optional (max_iter+1)*N each
};

hipError_t fh_k2_launch_init(const FitState &st, hipStream_t s);
hipError_t fh_k2_launch_prep(const FitState &st, hipStream_t s);
hipError_t fh_k2_launch_powerlaw(const FitState &st, hipStream_t s);
hipError_t fh_k2_launch_update(const FitState &st, hipStream_t s);
hipError_t fh_k2_launch_record(const FitState &st, hipStream_t s);
hipError_t fh_k2_launch_pinv_scale(const double *s, int n, double *s1, hipStream_t st);

// ---- method='LogNormal' for 320 < N <= 1023 (lognormal_wide.hip; the host drives MinimizeNewton, capi_lognormal.hip) ----------------
struct LnWideParams {
    int N;
    double s0, transform_norm;
    const double *M, *j, *Y, *Ykm, *q;  // device
    const double *mu;                   // the Normal seed fit (seed kernel)
    double *Sinv, *W;                   // N*N each: S^-1 = Y^T diag(1/p) Y, its left factor diag(1/p) Y
    double *p, *p_old;                  // the power spectrum (the fit loop's buffers)
    int *flags;                         // FIT_NFLAGS (the library loop's)
    double *x, *xn, *I, *t1, *t2, *fr, *jx, *dx;  // N each: the point, the last trial point and its I, S^-1 xn, M I, summands
    double *Sx, *Sp;                    // N each: S^-1 x of the point, S^-1 of the current search's direction
    double *scal;                       // 8: [0] H(xn), [1] xn == x, [2] max |jac||x|, [3] jac.p, [4] jac.dir
};
hipError_t fh_lnw_launch_seed(const LnWideParams &P, hipStream_t s);
hipError_t fh_lnw_launch_scale(const LnWideParams &P, hipStream_t s);
hipError_t fh_lnw_launch_eval(const LnWideParams &P, const double *x, const double *dir, double lam, int mode, hipStream_t s);
hipError_t fh_lnw_launch_jac(const LnWideParams &P, hipStream_t s);
hipError_t fh_lnw_launch_hess(const LnWideParams &P, double *H, hipStream_t s);
hipError_t fh_lnw_launch_limit_step(const LnWideParams &P, const double *x, const double *dir, double *p, hipStream_t s);
hipError_t fh_lnw_launch_matvec(int N, const double *A, const double *x, double alpha, const double *z, double *y, hipStream_t s);
hipError_t fh_lnw_launch_identity(double *A, int N, hipStream_t s);

// ---- K2 v2: single persistent kernel (fit_loop.hip) ---------------------------------------------------------
enum { FIT_MODE_FULL = 0, FIT_MODE_STEP = 1, FIT_MODE_SOLVE = 2, FIT_MODE_RESUME = 3 };
enum { FIT_STATUS_OK = 0, FIT_STATUS_BAD_P = 1, FIT_STATUS_NOT_SPD = 2, FIT_STATUS_CLUSTER = 3, FIT_STATUS_PAUSED = 4 };

struct FitLoopParams {
    int N, NP, max_iter, mode;
    double alpha, p0, tol;
    double pl_scale;        // DHT.transform(MAP) = pl_scale * m, m = Y mu  (hankel.py:155,199)
    const double *A;        // NP*NP, symmetric, zero padded: Y^-T M Y^-1
    const double *bq;       // N: Y^-T j
    const double *Yinv;     // N*N row-major: mu = Yinv m
    const double *q;        // N collocation frequencies
    const double *band_lu;  // 5N: LU factors of the pentadiagonal T + I
    const double *p_init;   // N or NULL (= ones)
    double *C, *W;          // NP*NP work: Cholesky factor (+ mirror), its inverse
    double *WdT;            // (NP/16)*256: transposed inverses of the diagonal tiles
    double *cs;             // (NP/16)^2 * 16: per-tile column sums of squares of W
    double *mu_out, *p_out; // N
    int *result;            // [0] count, [1] status
    double *diag_p, *diag_mu;
    long long *timing;      // debug builds (FIT_LOOP_TIMING): cycles per phase
    int trace_on;           // debug builds: record per-wave time stamps of this pass behind timing[16]
    unsigned long long *clk_out;  // fh_ctx_loop_clocks(ctx, 1): every fit adds [0] shader-clock cycles, [1] ticks of the 100 MHz
                                  // wall clock, [2] passes of its loop (the clock the compute units ran at with the device loaded)
    // batched launch (one workgroup per fit; A, bq, Yinv, q shared): per-fit alpha / p0, band_lu[f][5N], and the
    // work / output buffers strided by fit
    int batch;
    int *batch_counter;     // zeroed before the launch; workgroups pull fit indices from it
    const double *batch_alpha, *batch_p0;
    // slot launch (pipelined fits, one workgroup per fit, every operand per fit): the pointers above are those of slot 0,
    // slot i lives slot_stride doubles further (results: 2 ints further); workgroup b runs slot slot_ids[b]
    size_t slot_stride;
    unsigned long long slot_words[32];  // 128 slot ids of 16 bits, four to a word (FIT_MAX_BATCH)
    // pinned host mirrors of a slot's outputs (the pipeline reads them after the launch's completion event, no copy on any
    // stream): [mu (N), p (N)] and [count, status] per slot, strides 2 N doubles / 2 ints; NULL: none
    double *out_host;
    int *result_host;
    // cluster ("latency") mode, fit_loop.hip: `cluster` workgroups of ONE XCD per fit -- the first runs the loop and the
    // factorisation, the others the block columns of the inverse (one wave per column, no barriers), handed over through a
    // progress word in global memory.  The exchange area is the fit's WdT buffer: [0, NP) Tr2, [NP, 2 NP) m = Y mu from the
    // helpers, then the control words (ints, zero between fits: the first workgroup leaves them so).  Workgroup b of the
    // launch: XCD x = b & 7, index i = b >> 3 on it; fit (i / cluster) * 8 + x of the launch, member i % cluster.
    int cluster;            // 0 / 1: none
    int cluster_inv;        // helpers of the inverse among the cluster - 1 helpers (the others: helpers of the trailing update)
    int cluster_break;      // tests (FRANK_AMD_K2_CLUSTER_BREAK=1): the helpers leave at once, the cluster never assembles
    int cluster_xcd0;       // the XCD the first fit of the launch goes to (fit f sits on XCD (cluster_xcd0 + f) & 7): the host
                            // deals the small launches of a filling pipeline round the XCDs (a cluster wants an L2 to itself)
    int nfits;              // fits of a cluster launch (slot launch: entries of slot_words; single fit: 1)
    // Pause / resume (the sweeps' two-stage schedule, capi_fit.hip: sweep_staged).  pass_cap > 0: a fit that has made
    // pass_cap passes without converging stops with FIT_STATUS_PAUSED, its power spectrum in p_out and the one before in mu_out
    // -- the whole state of the iteration (radial_fitters.py:769-785 carries nothing else from pass to pass).
    // mode FIT_MODE_RESUME continues from such a state: `resume` = [p (N), p_old (N), passes made] per fit (batched launch:
    // 2 N + 1 doubles per fit; slot launch: behind the slot's band LU and hyper-parameters, the pointer is set by the kernel).
    int pass_cap;
    const double *resume;
    // ... or adaptively (batched launches): pause_when_left > 0 -- a fit pauses, at a pass that is a multiple of 16, once every fit
    // of the batch has been handed out (batch_counter >= batch) and at most pause_when_left of them have not ended (done_counter
    // counts the ended ones): the stragglers of a sweep stop together, when they are few enough for the clusters
    int pause_when_left;
    int *done_counter;
    int loaded;             // host hint: fit loops already resident on the device when this launch starts (launch_loop picks the
                            // form of the one-workgroup kernel that suits a full device: the rows of the inverse in pairs)
};
#define FIT_MAX_BATCH 128
#define FIT_CLUSTER_MAX 8
// doubles of a fit's cs buffer: (NP / 16)^2 x 16 column sums of the tiles of W, and -- cluster mode -- room for NP / 16 packed
// diagonal tiles that come back from the helpers of the trailing update (256 doubles each: more than the column sums below
// sixteen block rows; the first version wrote them past the end of the buffer there -- a memory fault at N = 128 and 150 that
// tools/size_sweep_cluster.py found), and for the vectors of the widest instantiation (14 NP + 3 072)
constexpr size_t fh_k2_cs_doubles(int NP) {
    const size_t nb = (size_t)NP / 16, a = nb * nb * 16, b = nb * 256, c = 14 * (size_t)NP + 3072;
    return (a > b ? a : b) > c ? (a > b ? a : b) : c;
}

size_t fh_k2_loop_smem_bytes(int NP);
int fh_k2_loop_max_np();  // largest padded size NP the persistent fit loop covers (640: N <= 639)
hipError_t fh_k2_launch_loop_rr(const FitLoopParams &P, int blocks, hipStream_t s);  // fit_loop_rr.hip: the matrix in registers (NP <= 304)
size_t fh_k2_loop_rr_smem_bytes(int NP);
hipError_t fh_k2_launch_loop(const FitLoopParams &P, hipStream_t s);        // P.cluster > 1: one fit on a cluster
hipError_t fh_k2_launch_loop_slots(const FitLoopParams &P, int nslots, hipStream_t s);  // P.cluster > 1: every fit on one
size_t fh_k2_exchange_doubles(int NP);  // doubles of a fit's WdT buffer (exchange area of the cluster mode)
hipError_t fh_k2_launch_loop_batched(const FitLoopParams &P, int batch, hipStream_t s);
hipError_t fh_k2_launch_symmetrize(const double *Araw, const double *bq, int N, int NP, double *A, hipStream_t s);

// ---- posterior extras of a sweep (evidence.hip) ---------------------------------------------------------------------
hipError_t fh_evidence_launch_build_c(const double *Araw, const double *p, int N, int batch, double *C, hipStream_t s);
hipError_t fh_evidence_launch_logdet(const double *L, int N, int batch, double *out, hipStream_t s);
hipError_t fh_evidence_launch_hessian(const double *Dqq, const double *mq, const double *p, const double *p0, const double *ws,
                                      const double *Tband, int N, int batch, double *H, hipStream_t s);
hipError_t fh_evidence_launch_diag(const double *A, int N, int batch, double *out, hipStream_t s);
hipError_t fh_launch_split_complex(const double *vc, int64_t n, double *re, double *im, hipStream_t s);  // (re, im) pairs -> columns

// ---- LogNormal (lognormal.hip): Newton MAP of the log-brightness + the power-spectrum loop, one workgroup per fit
enum { LN_MODE_MAP = 0, LN_MODE_FIT = 1, LN_MODE_UPDATE = 2 };
enum { LN_STATUS_OK = 0, LN_STATUS_BAD_P = 1, LN_STATUS_SLOPE = 2, LN_STATUS_CLUSTER = 3, LN_STATUS_PAUSED = 4, LN_STATUS_NOT_SPD = 5 };
// control words of a cluster of LogNormal workgroups (ints per group): [0, 8) command / counters / flags (lognormal.hip), then,
// each in a 128-byte line of its own, the progress word of a distributed Cholesky (panels stored, cumulative over the
// factorisations of the launch) and the counters of the block columns its helpers have handed back
constexpr int LN_CTL_WORDS = 128, LN_CTL_PROG = 32, LN_CTL_COL = 64;

struct LogNormalParams {
    int N, max_iter, mode, lu_in_lds, lu_nb;  // lu_nb: panel width of the blocked LU (set by fh_ln_launch)
    int max_step, max_hev;        // MinimizeNewton limits (minimizer.py:190-191: 10**5, 1000)
    double newton_tol;            // 1e-7 (statistical_models.py:1141)
    double alpha, p0, tol, s0;    // CriticalFilter hyper-parameters, loop tolerance, s0 = log(I_scale)
    double pl_scale;              // DHT.transform(f) = pl_scale * (Y f)
    const double *M, *j;          // normal equations (device)
    const double *Y, *q;          // DHT.coefficients() row-major, collocation q
    const double *band_lu;        // 5N: LU factors of the pentadiagonal T + I
    const double *p_in;           // LN_MODE_MAP: power spectrum
    const double *guess;          // LN_MODE_MAP: starting s;  LN_MODE_FIT: MAP of the Normal seed fit (radial_fitters.py:752)
    double *Sinv, *H, *LU;        // N*N work: prior precision, Hessian at the MAP (output Dinv), LU factors when N > 112
    int NP;                       // N rounded up to a multiple of 16.  LU holds fh_ln_lu_doubles(N, NP) doubles per workgroup:
                                  // [N*N] factors, [NP*NP] the padded copy of the Hessian that the tiled Cholesky factors in
                                  // place, [NP*NP] the solved tiles of the Tr2 triangular solve, [16*NP] inverses of the
                                  // diagonal tiles, [6*NP + 3072] bands and scan tables of the pentadiagonal solve (band_scan.h), [2*NP + NP*NP/8] vectors and
                                  // partial sums of the objective evaluations
    int dist_cholesky;            // 1: the trailing tiles of the tiled Cholesky live in the registers of the cluster's helpers (lognormal.hip:
                                  // chol_helper; the same bits; FRANK_AMD_LN_CLUSTER_CHOL=0 keeps the factorisation on the first workgroup)
    int no_cholesky;              // 1: skip the tiled Cholesky attempts, always the pivoted LU (FRANK_AMD_LN_PIVOTED=1: the route a
                                  // non-positive pivot takes, kept testable)
    int fresh_products;           // 1: every trial point of the line search gets its own S^-1 x product, as the reference's
                                  // H(x) evaluates it (statistical_models.py:1075-1085); 0: S^-1 (x + lam p) by linearity
    double *Hinv;                 // N*N work: explicit inverse of a Hessian that keeps being re-used
    double *s_out, *p_out;        // N
    int *result;                  // [0] count, [1] status
    long long *stats;             // [0] MAP solves, [1] Newton steps, [2] function evaluations, [3] Hessians, [4..8] exits 0-4
    double *diag_p, *diag_s;      // optional (max_iter+1)*N each
    // batched launch (sweeps): M, j, Y, q and the seed shared; Sinv / LU / Hinv strided by workgroup; H, s_out, p_out,
    // result (2), stats (17) strided by fit; per-fit alpha, p0, band_lu[f][5N]
    int batch;
    int *batch_counter;
    const double *batch_alpha, *batch_p0;
    // cluster (single fits, N > 112): `cluster` workgroups share the two per-pass pieces that are plain parallel work -- S^-1 =
    // Y^T diag(1/p) Y and the Tr2 triangular solve -- through flags in global memory; the first of them runs everything else
    int cluster;                  // workgroups of the cluster (1: none)
    int *ctl;                     // [0] sequence number, [1] command, [2] helpers done, [3] helpers alive, [4] disbanded (zeroed per launch)
    double *rk_g, *dvec_g, *tr2_g;  // N: 1 / p; NP: diagonal of the Cholesky factor; N: Tr2 -- the operands the helpers cannot read from the first workgroup's LDS
    // several clusters in one launch (round 6: the stragglers of a batched sweep): `groups` clusters, group g on XCD g % 8; ctl
    // (8 ints), the three exchange vectors (group_vec_stride doubles) and the Sinv / LU / Hinv work buffers are strided by group;
    // a group pulls fits from the batch counter like a lone workgroup does
    int groups, group_vec_stride;
    // staged sweeps: a fit pauses behind an update of p -- its state is (s, p, the p before, the count): s_out, p_out, the first N
    // entries of its H, result[0]; status LN_STATUS_PAUSED -- once every fit of the batch has been handed out and at most
    // pause_when_left have not ended (checked every 16 passes); resume: [fit][3 N + 1] = s, p, p_old, count of a paused fit
    int pause_when_left;
    int *done_counter;
    const double *resume;
};

constexpr size_t fh_ln_lu_doubles(int N, int NP) {
    return (size_t)N * N + 2 * (size_t)NP * NP + 16 * (size_t)NP + 6 * (size_t)NP + 3072  // (3072: bandscan::kTableDoubles)
           + 2 * (size_t)NP + 2 * ((size_t)NP / 16) * NP  // two vectors and [2][NP / 16][NP] partial sums of the evaluations
           + (N > 320 ? 29 * (size_t)N + 64 : 0);         // WIDE kernel (320 < N <= 639): its twenty vectors, the solve vectors, the permutation
}
size_t fh_ln_smem_bytes(int N, int *lu_in_lds, int *lu_nb);
hipError_t fh_ln_launch(const LogNormalParams &P, int nblocks, hipStream_t s);

// ---- geometry fits (vis_residual.hip) ----------------------------------------------------------------------------
// b: table columns, row range and the trial geometry as the binning pass takes them (zeros, j0_table, inv_Qmax, H2 too)
struct VisResidualParams {
    BinParams b;
    const double *pref;  // norm * scale_factor[k] of the forward transform (hankel.py:201)
    const double *I;     // brightness profile, N entries (device)
    double scale;        // cos(inc) for the optically thick model, 1 otherwise (statistical_models.py:486-490)
    double *out;         // [2 count]: real parts, then imaginary parts; NULL = sum of squares only
    double *partial;     // one sum of squares per workgroup (fh_residual_max_blocks())
    // through the bucket tables of the binning pass (bin_gram2.hip, predict_bucket_coef_kernel): [nb][FH_K1_TERMS] Taylor
    // coefficients of V(s) per bucket of s = q / Qmax; NULL = N Bessel evaluations per row
    const double *coef;
    int nb;
    double delta;
    int predict_only;    // no data: out = the model visibilities themselves (only b.u / b.v of the table columns are read)
};
struct GaussResidualParams {
    BinParams b;         // table + (cos, sin) of PA and inc in cos_t, sin_t, cos_i, sin_i; dRA, dDec in radians per wavelength
    double norm, scal, rad_to_arcsec, fac;  // fac = 2 pi / rad_to_arcsec
    int fit_inc_pa, fit_phase;              // which Jacobian columns are filled (geometry.py:556-577)
    double *fun;         // [2 count] or NULL
    double *jac;         // [2 count][6] row-major or NULL
    double *partial;
};
// forward-difference normal equations from residual vectors on the device (fh_residual_normal_equations)
struct FdNormalParams {
    const double *base;    // r(x), len entries
    const double *col[4];  // r(x + h_k e_k)
    double inv_h[4];
    int ncol;
    int64_t len;
    double *partial;       // [workgroups][14]
};
int fh_residual_max_blocks();
int fh_residual_sums_max();  // widest row of partial sums any of these kernels writes per workgroup
hipError_t fh_launch_fd_normal(const FdNormalParams &P, double *out14, hipStream_t stream);
hipError_t fh_launch_gauss_normal(const GaussResidualParams &P, double *out28, hipStream_t stream);
hipError_t fh_launch_vis_residual(const VisResidualParams &P, double *sumsq, hipStream_t stream);
hipError_t fh_launch_gauss_residual(const GaussResidualParams &P, double *sumsq, hipStream_t stream);

// ---- UVDataBinner (uvbin.hip) -----------------------------------------------------------------------------------
struct UvBinParams {
    const double *uv, *w;   // baselines and weights of the rows (device)
    const double *qty[4];   // up to four quantities to sum as w * qty (NULL = ones)
    int nq;                 // number of quantities
    int count;              // also count rows per bin
    int64_t n;
    double bin_width, norm; // norm = 1 / bin_width (utilities.py:211)
    int nbins, use_lds;
    const double *mu_re, *mu_im;  // per-bin means for the error pass
    double *scratch;              // per-workgroup slabs (fh_uvbin_scratch_doubles); NULL: global atomics into sums
    double *sums;                 // nq * nbins, zeroed by the caller
    unsigned long long *counts;   // nbins, zeroed by the caller
};

hipError_t fh_uvbin_launch_max(const double *uv, int64_t n, unsigned long long *out2, int num_cu, hipStream_t s);
size_t fh_uvbin_scratch_doubles(int nq_plus, int nbins, int64_t n, int num_cu);
hipError_t fh_uvbin_launch_sum(const UvBinParams &p, int num_cu, hipStream_t s);
hipError_t fh_uvbin_launch_err(const UvBinParams &p, int num_cu, hipStream_t s);
hipError_t fh_uvbin_launch_lookup(const double *uv, int64_t n, double bin_width, int nbins, int *out, int num_cu,
                                  hipStream_t s);
