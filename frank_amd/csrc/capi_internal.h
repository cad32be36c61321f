// C-ABI glue (include/frank_hip.h), shared part: error handling, device buffers, the handle types (fh_vis, fh_ctx, fh_comm) and the
// helpers the entry-point families call across files.  The entry points themselves:
//   capi_core.hip       library / DHT / context life cycle, visibility tables, the allocation cache
//   capi_map.hip        K1: H(q), predict, the binning pass, statistics, fh_map_visibilities
//   capi_fit.hip        K2: GaussianModel solves, the fit loop (single, batched, staged sweeps, pipelined), evidence
//   capi_lognormal.hip  K3: method='LogNormal' (persistent kernel and the host-driven route beyond N = 320)
//   capi_callers.hip    UVDataBinner, the residual functions of the geometry fits, sky-plane predict
//   capi_comm.hip       RCCL all-reduce of the packed statistics
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <utility>
#include <string>
#include <vector>

#include "../../include/frank_hip.h"
#include "bessel.h"
#include "dht_host.h"
#include "j0_buckets.h"
#include "kernels.h"

extern thread_local std::string g_err;       // capi_core.hip
int fail(int code, const char *fmt, ...);  // records the message fh_last_error returns; returns code

#define HIP_TRY(expr)                                                                               \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess) return fail(FH_ERR_HIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)
#define ROC_TRY(expr)                                                                               \
    do {                                                                                            \
        rocblas_status s_ = (expr);                                                                 \
        if (s_ != rocblas_status_success)                                                           \
            return fail(FH_ERR_HIP, "%s: rocblas status %d (%s:%d)", #expr, (int)s_, __FILE__, __LINE__); \
    } while (0)

static const double kRadToArcsec = 3600.0 * 180.0 / M_PI;  // frank/constants.py:23
static const double kDegToRad = M_PI / 180.0;              // frank/constants.py:25

// Functions that queue copies into CALLER memory must not return (on an error path) while those copies are in flight.
struct SyncOnExit {
    hipStream_t s;
    ~SyncOnExit() { (void)hipStreamSynchronize(s); }
};

// ---- the allocation cache of the visibility tables' columns (see DevBuf::alloc_pooled) ----------------------------------------
constexpr size_t kPoolBytes = (size_t)3 << 29;  // 1.5 GB held at most
struct PoolEntry {
    void *p;
    size_t bytes;
    int device;
};
// The streams on which kernels READ the columns of a visibility table: every context's binning stream and its look-ahead stream
// (fit loops read the operands of their slot, never a table).  fh_vis_destroy waits for these and the null stream, not for the
// device: a pipeline that frees a table behind its fit no longer waits for every fit loop in flight (45 ms apiece).
void reader_stream_add(int device, hipStream_t s);  // capi_core.hip
void reader_stream_remove(hipStream_t s);
void reader_streams_sync(int device);
void *pool_take(size_t bytes, int device);   // capi_core.hip
void pool_put(void *p, size_t bytes, int device);
void pool_clear();

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    bool owned = true;
    hipError_t alloc(size_t count) {
        release();
        owned = true;
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&p), count * sizeof(T));
        if (e != hipSuccess) {  // (up to kPoolBytes of freed table columns may be what is in the way: give them back and try once more)
            pool_clear();
            e = hipMalloc(reinterpret_cast<void **>(&p), count * sizeof(T));
        }
        if (e != hipSuccess) {  // leave the buffer empty so that a later grow-on-demand check allocates again
            p = nullptr;
            return e;
        }
        n = count;
        return e;
    }
    void adopt(T *slice, size_t count) {  // a slice of somebody else's allocation
        release();
        p = slice;
        n = count;
        owned = false;
    }
    void release() {
        if (p && owned) {
            if (pooled) pool_put(p, n * sizeof(T), pool_device);
            else (void)hipFree(p);
        }
        p = nullptr;
        n = 0;
        pooled = false;
    }
    // The columns of a visibility table go through a small cache of freed allocations (per device, exact size, at most
    // kPoolBytes held): VisibilityMapping.map_visibilities(u, v, V, w) uploads a table, bins it and frees it at every call, and
    // six hipMalloc + six hipFree of 80-160 MB were ~10 ms of its 19 ms at 1e7 rows.  fh_cache_release() empties the cache.
    bool pooled = false;
    int pool_device = 0;
    hipError_t alloc_pooled(size_t count, int device) {
        release();
        owned = true;
        void *q = pool_take(count * sizeof(T), device);
        if (!q) {
            const hipError_t e = hipMalloc(&q, count * sizeof(T));
            if (e != hipSuccess) {
                pool_clear();  // (the cache may be what is in the way)
                const hipError_t e2 = hipMalloc(&q, count * sizeof(T));
                if (e2 != hipSuccess) {
                    p = nullptr;
                    return e2;
                }
            }
        }
        p = reinterpret_cast<T *>(q);
        n = count;
        pooled = true;
        pool_device = device;
        return hipSuccess;
    }
    ~DevBuf() { release(); }
};

extern std::atomic<unsigned long long> g_vis_serial;  // capi_core.hip
struct fh_vis {
    unsigned long long serial = g_vis_serial.fetch_add(1);  // identifies the table in the baseline-range cache of a context
    int device = 0;
    int64_t n = 0;
    int w_scalar = 0, has_im = 0;
    DevBuf<double> u, v, Vre, Vim, w;
    DevBuf<float> u32, v32, Vre32, Vim32, w32;  // fh_vis_upload_f32: the same columns in fp32
    bool f32 = false;
    DevBuf<int> mult;  // bootstrap multiplicities (fh_vis_set_multiplicity), empty = every row once
    bool use_mult = false;
    unsigned long long mult_gen = 0;  // changes with every fh_vis_set_multiplicity: rows drawn zero times leave the range
    mutable DevBuf<double> resid;     // geometry fits: residuals (2 n), Jacobian (12 n), partial sums -- grown on first use
    mutable DevBuf<double> slots;     // geometry fits on the normal equations: FH_RESIDUAL_SLOTS residual vectors of 2 n
};

struct FitSlot {
    DevBuf<double> Aq, bq, Cq, Wq, WdT, cs, mu_out, p_out, band_lu;
    DevBuf<int> result;
    std::vector<double> lu_host;  // stays alive while the asynchronous copy of the band LU may still read it
    std::vector<double> resume_host;  // the state of a paused fit on its way to the slot (behind the band LU), likewise
    double lu_key[3] = {0, 0, 0};  // (w_smooth, alpha, p0) of the factors the device copy holds
    bool lu_valid = false;
    bool busy = false;
    int batch = -1;               // the launch this fit belongs to
};
// Pipelined fits are launched in BATCHES: one fit_loop launch with one workgroup per fit (kernels.h: slot launch).
//  * A single dispatch deals its workgroups evenly over the XCDs and their shader engines; fit loops started one by one
//    land wherever the dispatcher's pointers happen to be, and the binning kernels, whose workgroups are dealt IN ORDER,
//    stop at the first engine without a free CU (round 1: 0.40 ms of binning time per co-running fit loop started singly,
//    0.25 ms in a batch of 16).
//  * The fits in flight are not limited by the hardware queues (a stream per launch, not per fit).
// HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4; raised to 24 below, so that the streams of several
// contexts never share one) and two kernels whose streams share a queue serialise (seen in the kernel trace as 190 ms
// stalls with 32 streams).  But the command processor SERVES about four queues at a time: see FitBatch below.
constexpr int kFitSlots = 512;  // capacity; fh_fit_slots() is what a context hands out at a time (FRANK_AMD_FIT_SLOTS)
constexpr int kFitBatchMax = FIT_MAX_BATCH;
constexpr int kFitBatches = 16;
constexpr int kLaunchStreamsMax = 8;
// The launches of the pipeline share a FEW streams (four by default): the command processor serves four hardware queues at a
// time; with eight launches of sixteen fit loops on eight queues an empty kernel on the binning stream took 31 us instead of 3
// and a 10 us kernel 75 (tools/microbench/boundary_cost.hip) -- every one of the ~16 kernels of a pipelined step paid that.
// Launch i goes to stream i mod 3 and starts when launch i - 3 has ended; its completion is an event, its results land in
// pinned host memory, so collecting a fit waits for ITS launch only and puts nothing on any stream.
struct FitBatch {
    hipStream_t stream = nullptr;  // (one of the context's launch streams; not owned)
    hipEvent_t ready = nullptr, done = nullptr;
    unsigned short slots[kFitBatchMax];
    int n = 0, outstanding = 0;
    bool active = false, launched = false;
    double alpha = 0, p0 = 0, tol = 0;
    int max_iter = 0;
    int cluster = 1;  // workgroups per fit of this launch (fit_loop.hip, cluster mode)
    int mode = FIT_MODE_FULL;  // FIT_MODE_RESUME: every fit of the launch continues from a paused state (the sweeps' second stage)
};

struct fh_ctx {
    const fh_dht *dht = nullptr;
    int device = 0, N = 0, NBT = 0, ntiles = 0, nparts = 1, num_cu = 0;
    hipStream_t stream = nullptr;
    rocblas_handle blas = nullptr;
    // DHT constants on the device
    DevBuf<double> zeros, j0_table, Y, Ykm, q, pref_fwd, pref_bwd;
    // K1
    int part_blocks[3] = {0, 0, 0};
    DevBuf<double> partials[3], partial_scalars, stats_sum, stats_minmax, a_scale, sumwV2, prep, reduce_scratch;
    DevBuf<int> work_counter;
    int deproject_blocks = 0;
    // K1 v2 (bin_gram2.hip): bucket sort workspaces and the Taylor tables of the buckets seen so far
    bool v2 = false, force_static = false;
    bool rows_ok = true;             // false: 511 < N <= 1023, only the moments path of the fused kernels exists
    bool check_q_before_bin = false;  // fh_map_visibilities(check_qbounds): _check_uv_range before any binning, as the reference
    double prepass_qmin = 0, prepass_qmax = 0, prepass_qmax_all = 0;
    int XS = 0, k1_nb_built = 0, sort_blocks = 0;
    double k1_delta = 0;
    DevBuf<double> k1_table, k1_rows;
    DevBuf<double> predict_coef;    // [bucket][12]: fh_predict_visibilities through the tables
    DevBuf<float> k1_table32;       // the tables rounded to fp32 (fh_ctx_set_arithmetic)
    int k1_nb_built32 = 0;
    bool arith32 = false;
    bool ln_fresh_products = false;  // fh_ctx_set_lognormal_linesearch
    DevBuf<int> k1_hist, k1_totals, k1_starts, k1_info, k1_chunk_bucket;
    DevBuf<double> k1_partial;      // partial moments of the bucket slices
    DevBuf<double> k1_vrows;        // compressed rows (fh_k1v2_launch_compress): one 16 x 16 chunk per non-empty bucket
    DevBuf<int> k1_cidx, k1_vbucket;
    DevBuf<int> k1_piece0;          // bin_prepass.hip: first partial-moment slot of every bucket
    DevBuf<int> k1_slot_tab;        // bin_fused.hip: accumulator slots of every bucket, [nb] then the number of slots
    int bin_cus = 0;                // fh_ctx_set_cu_partition
    bool no_range_cache = false;    // fh_ctx_set_range_cache(ctx, 0): look at (u, v) on every pass (benchmarks of distinct tables)
    // development switches of the binning pass (FRANK_AMD_K1_*, FRANK_AMD_NO_RANGE_CACHE), read ONCE when the context is created
    // -- a pass used to make a dozen getenv calls -- and again on fh_ctx_reload_env (tests that switch them inside one process)
    struct K1Env {
        int unroll = 2, seg = 4096, wpb = 0, blocks = 0, vrwaves = 8, vrsplit = 8, vrblocks = 0;
        bool no_range_cache = false, safe_trig = false, no_hist_cache = false, vr_slabs = false, dynamic = false;
        double reserve_mult = -1.0;
        int fused = 0;  // FRANK_AMD_K1_FUSED: the one-pass form of the moments pre-pass (bin_fused.hip)
    } k1env;
    // baseline range of the last pre-pass, keyed by (table, row range, geometry): binning the same rows under the same geometry
    // again (bootstrap draws, pipelines of fits, sweeps) needs no second look at the range before the sort is sized
    unsigned long long range_vis = 0, range_mult_gen = 0;
    int64_t range_first = -1, range_count = -1;
    double range_geom[6] = {0, 0, 0, 0, 0, 0};
    bool range_valid = false;
    // fh_bin_prefetch_range: the range pass of the NEXT table on a stream of its own; (-, qmin, qmax, qmax_all) per workgroup land in
    // pinned host memory behind an event
    hipStream_t pf_stream = nullptr;
    struct LookAhead {  // (the one for the next pass is asked for BEFORE the current pass is queued, so that its kernel runs in front of
                        //  that pass and not behind it)
        hipEvent_t event = nullptr, ev0 = nullptr, ev1 = nullptr;  // the copy has landed; around the range kernel (fh_bin_last_range_ms)
        DevBuf<double> dev;
        DevBuf<int> hist;           // one look: the per-workgroup bucket histograms of these rows for nb_cap buckets (has_hist)
        bool has_hist = false;
        hipEvent_t consumed = nullptr;  // on the binning stream, behind the kernels that read `hist`: the next look into this slot waits for it
        bool consumed_pending = false;
        int nb_cap = 0, hist_stride = 0, wpb = 0, unroll = 0;
        double *host = nullptr;
        int blocks = 0;
        bool valid = false;
        unsigned long long seq = 0;  // order of the fh_bin_prefetch_range calls
        unsigned long long vis = 0, mult_gen = 0;
        int64_t first = -1, count = -1;
        double geom[6] = {0, 0, 0, 0, 0, 0};
    };
    // A pool: a look-ahead is free once the pass that took its histograms has run on the binning stream (`consumed`); a pipeline
    // whose host runs K passes ahead of the device has K in use (the driver's 20 steps: ~20 x 6 MB).  Capped at kMaxLookAheads:
    // beyond, the oldest is waited for.
    static constexpr int kMaxLookAheads = 48;
    std::vector<std::unique_ptr<LookAhead>> pf;
    unsigned long long pf_seq = 0;
    LookAhead *rng_la = nullptr;  // the look-ahead whose range the last pass took (its events time the kernel)
    // the per-workgroup bucket histograms, their scan and the layout of the sorted table (bin_prepass.hip: P1 + scan) of the LAST
    // pre-pass of the moments path, valid for exactly the rows / geometry / multiplicities of the range key above and this
    // launch geometry: a pass over the same rows skips P1 and the scan (16 of its 104 bytes per row)
    bool hist_valid = false;
    int hist_nb = 0, hist_blocks = 0, hist_wpb = 0, hist_unroll = 0, hist_seg = 0;
    bool hist_fused = false;
    std::vector<double> a_host;      // finalize scale vector (stays alive behind an asynchronous copy)
    bool a_scale_valid = false;
    double a_scale_value = 0.0;
    bool k1_moments = true;         // FRANK_AMD_K1=rows: bin the visibilities themselves (the v2 path, kept for cross-checks)
    std::vector<double> k1_scalars_host;
    hipEvent_t ev_pre0 = nullptr;
    hipEvent_t ev_rng0 = nullptr, ev_rng1 = nullptr;  // around the range pass of the last binning pass that needed one, on whichever stream
    bool rng_timed = false;
    hipEvent_t ev_loop0 = nullptr, ev_loop1 = nullptr;  // around the fit_loop kernel of the last fh_fit_normal
    bool loop_timed = false;
    bool have_device_mu = false;  // a solve of this context has left a profile in `mu` (fh_vis_residuals_slot with I = NULL reads it)
    float last_prepass_ms = 0.f;
    // N > 303: rows to memory + rocBLAS dsyrk; stats_sum then holds the dense (N+1)^2 Gram (upper triangle) + 2 scalars
    bool wide = false;
    size_t tail_offset = 0;      // index of sum log(w / 2 pi) in stats_sum
    bool stats_reset_pending = false;  // fh_bin_reset came, its two fills have not run: the moments path's last kernel then
                                       // STORES its sums (settle_reset() runs the fills for everybody else)
    int64_t wide_rows = 0;       // rows per dsyrk chunk
    DevBuf<double> wide_X, wide_G;  // wide_G: dense Gram + 2 scalars when the tile workspace exists too (debris, N <= 303)
    DevBuf<double> debris_H2;    // vis_model 'debris': H2[k]; set by fh_ctx_set_scale_height, forces the rows + dgemm path
    bool debris = false;
    // normal equations + K2 work
    DevBuf<double> M, j, W, D, Z, p, p_old, mu, band_lu, diag_p, diag_mu;
    DevBuf<int> flags, info;
    // K2 v2 (fit_loop): q-space operands and work buffers
    int NP = 0;
    bool use_rocsolver_loop = false;
    DevBuf<double> Yinv, T1, Araw, Aq, bq, Cq, Wq, WdT, cs, mu_out, p_out, p_init;
    DevBuf<int> loop_result;
    DevBuf<long long> loop_timing;  // FIT_LOOP_TIMING debug builds only
    DevBuf<unsigned long long> loop_clocks;  // fh_ctx_loop_clocks: [cycles, 100 MHz ticks, passes] summed over the fits since the last read
    DevBuf<double> slot_pool;   // backing store of every slot's buffers
    DevBuf<int> slot_results;
    FitSlot slots[kFitSlots];
    int n_slots = 0;  // slots this context has carved (fit_slots_wanted() when its pool was made; 0: no pool yet)
    FitBatch batches[kFitBatches];
    hipStream_t launch_streams[kLaunchStreamsMax] = {};
    int n_launch_streams = 0;
    unsigned long long launches = 0;      // launches so far (stream of the next one: launches mod n_launch_streams)
    double *slot_out_host = nullptr;      // pinned: [slot][mu (N), p (N)]
    int *slot_result_host = nullptr;      // pinned: [slot][count, status]
    int pending_batch = -1;  // batch that is still collecting submissions (not launched)
    int fit_batch = kFitBatchMax;
    int burst_next = 1;      // fits that trigger the next launch: 1, 2, 4, .. up to fit_batch while a pipeline fills up
    int next_xcd = 0;        // XCD of the first fit of the next cluster launch
    bool qspace_shared = false;  // fit_submit_impl copies the context's q-space operands instead of forming them again (sweeps)
    bool force_cluster_launch = false;  // the launch being flushed runs on clusters whatever is outstanding (fh_fit_normal_batched)
    hipEvent_t stream_last_done[kLaunchStreamsMax] = {};  // completion event of the last launch each launch stream was given
    size_t slot_stride = 0;
    int slots_busy = 0;
    bool throughput_context = false;      // the pipeline of this context has had 128 fits in flight: its launches take the form of
                                          // the fit loop that is faster on a loaded device (capi_fit.hip, fit_loop.hip: launch_loop)
    int last_fit_cluster = 1;             // workgroups the last fh_fit_normal ran on
    unsigned long long cluster_fallbacks = 0;  // cluster launches that ended with FIT_STATUS_CLUSTER and were repeated on one CU
    bool have_device_Mj = false;
    hipEvent_t ev_bin0 = nullptr, ev_bin1 = nullptr;
    bool bin_timed = false;
    // scratch for coefficient / predict calls
    DevBuf<double> scratch_q, scratch_out, scratch_I;
    // LogNormal (lognormal.hip)
    DevBuf<double> ln_Sinv, ln_H, ln_LU, ln_Hinv, ln_s, ln_p, ln_pin, ln_guess, ln_diag_p, ln_diag_s;
    DevBuf<int> ln_result, ln_ctl;
    DevBuf<double> ln_cluster_vecs;  // 1 / p, diag(L), Tr2: what the helper workgroups of a cluster exchange with the first
    DevBuf<long long> ln_stats;
    // LogNormal beyond N = 320 (lognormal_wide.hip)
    DevBuf<double> lnw_Sinv, lnw_H, lnw_Hinv, lnw_Hc, lnw_vec, lnw_scal, lnw_diag_s;
    DevBuf<rocblas_int> lnw_ipiv;
};

// bits [first, last) of a 256-bit compute-unit mask
static inline void cu_mask(int first, int last, uint32_t mask[8]) {
    for (int w = 0; w < 8; ++w) mask[w] = 0;
    for (int b = first; b < last && b < 256; ++b) mask[b >> 5] |= 1u << (b & 31);
}

struct fh_comm {
    void *lib = nullptr;
    void *comm = nullptr;
    int rank = 0, world = 1, device = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;  // around the most recent all-reduce, on the context's stream
    bool timed = false;
    int (*allreduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*destroy)(void *) = nullptr;
    const char *(*errstr)(int) = nullptr;
};


static inline int env_int(const char *name, int dflt) {
    const char *e = getenv(name);
    return e && *e ? atoi(e) : dflt;
}
// Environment switches.  Two kinds:
//   * FORM switches select among forms of a kernel that give the same results -- the parity tests compare them inside one process
//     (FRANK_AMD_K1, _K1_FUSED, _K1_NO_HIST_CACHE, _K1_SAFE_TRIG, _NO_RANGE_CACHE, _K2, _K2_CLUSTER, _K2_CLUSTER_BREAK, _K2_RR, _K2_DEFER,
//     _K2_LL, _SWEEP_CAP, _SWEEP_LEFT, _SWEEP_STAGE2_CLUSTERS, _SWEEP_NO_CLUSTERS, _LN_CLUSTER, _LN_PIVOTED, _LN_WIDE, _RESIDUAL_DIRECT): read with
//     env_int / getenv, listed in INTEGRATION.md;
//   * DEVELOPMENT switches (tuning knobs, probes, kill-line forms) exist only in builds with -DFRANK_AMD_DEV (`make dev`: the tools
//     under tools/ that sweep them load libfrank_hip_dev.so through FRANK_AMD_LIB): the shipped library reads none of them.
#ifdef FRANK_AMD_DEV
#define FH_DEV_INT(name, dflt) env_int(name, dflt)
#define FH_DEV_STR(name) getenv(name)
#else
#define FH_DEV_INT(name, dflt) (dflt)
#define FH_DEV_STR(name) (static_cast<const char *>(nullptr))
#endif
#define FH_DEV_SET(name) (FH_DEV_STR(name) != nullptr)

// ---- helpers shared by the entry-point families (defined in the file named) ---------------------------------------------
extern "C" {
int k1v2_ensure_table(fh_ctx *c, int nb_needed);                       // capi_map.hip: Taylor tables of the buckets the data reach
void load_k1_env(fh_ctx *c);                                           // capi_map.hip: the FRANK_AMD_K1_* switches, once per context
int settle_reset(fh_ctx *c);                                           // capi_map.hip
void table_columns(BinParams &p, const fh_vis *vis, int64_t first, int64_t count);  // capi_map.hip
int stage_q(fh_ctx *c, const double *q, int64_t n);                    // capi_map.hip
void smoothing_band_lu(const fh_dht &d, double wsmooth, std::vector<double> &lu);    // capi_fit.hip
FitState make_state(fh_ctx *c);                                        // capi_fit.hip
int solve_posterior(fh_ctx *c, const FitState &st, bool with_prior, bool want_tr2);  // capi_fit.hip
int svd_pinv_solve_device(fh_ctx *c, double *A_dev, double *B_dev, int nrhs, bool last_axis_scaling = false);  // capi_fit.hip
int prepare_qspace(fh_ctx *c, double *Aq, double *bq);                 // capi_fit.hip
int fit_cluster_size(const fh_ctx *c);                                 // capi_fit.hip
FitLoopParams make_loop_params(fh_ctx *c, int mode, double alpha, double p0, double tol, int max_iter);  // capi_fit.hip
int running_fit_loops(fh_ctx *c);                                      // capi_map.hip
std::vector<int> sweep_launch_order(const double *alpha, const double *wsmooth, int batch);  // capi_fit.hip
bool use_wide(const fh_ctx *c);                                        // capi_map.hip: rows to memory + dsyrk (N > 303, debris)
double *dense_gram(fh_ctx *c);                                         // capi_map.hip
size_t dense_tail(const fh_ctx *c);                                    // capi_map.hip
}
