// C-ABI glue (include/frank_hip.h): handles, device buffers, kernel / rocBLAS / rocSOLVER / RCCL calls.
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

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

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

const double kRadToArcsec = 3600.0 * 180.0 / M_PI;  // frank/constants.py:23
const double kDegToRad = M_PI / 180.0;              // frank/constants.py:25

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
static std::mutex g_pool_mutex;
static std::vector<PoolEntry> g_pool;
static size_t g_pool_held = 0;
static void *pool_take(size_t bytes, int device) {
    std::lock_guard<std::mutex> lk(g_pool_mutex);
    for (size_t i = 0; i < g_pool.size(); ++i)
        if (g_pool[i].bytes == bytes && g_pool[i].device == device) {
            void *p = g_pool[i].p;
            g_pool_held -= bytes;
            g_pool.erase(g_pool.begin() + (long)i);
            return p;
        }
    return nullptr;
}
static void pool_put(void *p, size_t bytes, int device) {
    std::lock_guard<std::mutex> lk(g_pool_mutex);
    if (bytes < ((size_t)1 << 20) || bytes > kPoolBytes / 2) {  // small ones are cheap to allocate; huge ones are not worth holding
        (void)hipFree(p);
        return;
    }
    while (!g_pool.empty() && g_pool_held + bytes > kPoolBytes) {  // oldest out
        (void)hipFree(g_pool.front().p);
        g_pool_held -= g_pool.front().bytes;
        g_pool.erase(g_pool.begin());
    }
    g_pool.push_back({p, bytes, device});
    g_pool_held += bytes;
}
static void pool_clear() {
    std::lock_guard<std::mutex> lk(g_pool_mutex);
    for (const PoolEntry &e : g_pool) (void)hipFree(e.p);
    g_pool.clear();
    g_pool_held = 0;
}

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    bool owned = true;
    hipError_t alloc(size_t count) {
        release();
        owned = true;
        const hipError_t e = hipMalloc(reinterpret_cast<void **>(&p), count * sizeof(T));
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

}  // namespace

static std::atomic<unsigned long long> g_vis_serial{1};
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
    int bin_cus = 0;                // fh_ctx_set_cu_partition
    bool no_range_cache = false;    // fh_ctx_set_range_cache(ctx, 0): look at (u, v) on every pass (benchmarks of distinct tables)
    // development switches of the binning pass (FRANK_AMD_K1_*, FRANK_AMD_NO_RANGE_CACHE), read ONCE when the context is created
    // -- a pass used to make a dozen getenv calls -- and again on fh_ctx_reload_env (tests that switch them inside one process)
    struct K1Env {
        int unroll = 2, seg = 4096, wpb = 0, blocks = 0, vrwaves = 8, vrsplit = 8, vrblocks = 0;
        bool no_range_cache = false, safe_trig = false, no_hist_cache = false, vr_slabs = false, dynamic = false;
        double reserve_mult = -1.0;
    } k1env;
    // baseline range of the last pre-pass, keyed by (table, row range, geometry): binning the same rows under the same geometry
    // again (bootstrap draws, pipelines of fits, sweeps) needs no second look at the range before the sort is sized
    unsigned long long range_vis = 0, range_mult_gen = 0;
    int64_t range_first = -1, range_count = -1;
    double range_geom[6] = {0, 0, 0, 0, 0, 0};
    bool range_valid = false;
    // the per-workgroup bucket histograms, their scan and the layout of the sorted table (bin_prepass.hip: P1 + scan) of the LAST
    // pre-pass of the moments path, valid for exactly the rows / geometry / multiplicities of the range key above and this
    // launch geometry: a pass over the same rows skips P1 and the scan (16 of its 104 bytes per row)
    bool hist_valid = false;
    int hist_nb = 0, hist_blocks = 0, hist_wpb = 0, hist_unroll = 0, hist_seg = 0;
    std::vector<double> a_host;      // finalize scale vector (stays alive behind an asynchronous copy)
    bool a_scale_valid = false;
    double a_scale_value = 0.0;
    bool k1_moments = true;         // FRANK_AMD_K1=rows: bin the visibilities themselves (the v2 path, kept for cross-checks)
    std::vector<double> k1_scalars_host;
    hipEvent_t ev_pre0 = nullptr;
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
    bool force_cluster_launch = false;  // the launch being flushed runs on clusters whatever is outstanding (fh_fit_normal_batched)
    hipEvent_t stream_last_done[kLaunchStreamsMax] = {};  // completion event of the last launch each launch stream was given
    size_t slot_stride = 0;
    int slots_busy = 0;
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
static void cu_mask(int first, int last, uint32_t mask[8]) {
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

// HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and kernels whose streams share a queue serialise: the
// launches of a pipeline of fits (up to six streams beside the binning stream) want at least eight.  The variable is read when
// the HIP runtime initialises (the first HIP call of the process), so it can only be set before that -- by the embedding
// application, or by an explicit fh_init(); loading this library changes nothing in the process (rounds 1-3 did it in a
// constructor).  A context created with fewer queues records a warning (fh_last_warning).
static thread_local std::string g_warn;
constexpr int kHwQueuesWanted = 8;
static int hw_queues_env() {
    const char *e = getenv("GPU_MAX_HW_QUEUES");
    return e ? atoi(e) : 4;  // (the runtime's default)
}

extern "C" {

int fh_init(void) {
    setenv("GPU_MAX_HW_QUEUES", "24", 0);  // never overrides a value the user chose
    return hw_queues_env();
}
const char *fh_last_warning(void) { return g_warn.c_str(); }

const char *fh_last_error(void) { return g_err.c_str(); }
// (the build stamp ties a profile under profiles/ to the binary it was taken from: tools/profile_r04.sh records it, bench.py
//  prints the loaded library's beside the profile's)
extern "C" const char *fh_build_stamp(void);  // version_stamp.cpp: compiled again whenever any object of the library changes
const char *fh_version(void) { return fh_build_stamp(); }

int fh_device_count(int *count) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) n = 0;
    if (count) *count = n;
    return FH_OK;
}

// ---- DHT (host) ---------------------------------------------------------------------------------------------
int fh_dht_create(double Rmax_rad, int N, int nu, fh_dht **out) {
    if (!out) return fail(FH_ERR_INVALID, "fh_dht_create: out is NULL");
    if (nu != 0) return fail(FH_ERR_UNSUPPORTED, "fh_dht_create: only nu = 0 is implemented (got %d)", nu);
    if (N < 1 || !(Rmax_rad > 0)) return fail(FH_ERR_INVALID, "fh_dht_create: need N >= 1 and Rmax > 0");
    fh_dht *d = new fh_dht();
    int rc = fh_dht_build(Rmax_rad, N, d);
    if (rc != FH_OK) {
        delete d;
        return fail(rc, "fh_dht_create: set-up failed");
    }
    *out = d;
    return FH_OK;
}
void fh_dht_destroy(fh_dht *dht) { delete dht; }
int fh_dht_size(const fh_dht *dht) { return dht ? dht->N : 0; }
int fh_dht_get(const fh_dht *d, double *r, double *q, double *zeros, double *Ykm, double *scale_factor, double *Qmax,
               double *Rmax) {
    if (!d) return fail(FH_ERR_INVALID, "fh_dht_get: dht is NULL");
    const int N = d->N;
    if (r) memcpy(r, d->r.data(), sizeof(double) * N);
    if (q) memcpy(q, d->q.data(), sizeof(double) * N);
    if (zeros) memcpy(zeros, d->zeros.data(), sizeof(double) * (N + 1));
    if (Ykm) memcpy(Ykm, d->Ykm.data(), sizeof(double) * (size_t)N * N);
    if (scale_factor) memcpy(scale_factor, d->scale_factor.data(), sizeof(double) * N);
    if (Qmax) *Qmax = d->Qmax;
    if (Rmax) *Rmax = d->Rmax;
    return FH_OK;
}

int fh_dht_bucket_tables(const fh_dht *d, int b0, int b1, double *table, double *delta) {
    if (!d || b0 < 0 || b1 < b0) return fail(FH_ERR_INVALID, "fh_dht_bucket_tables: bad argument");
    if (delta) *delta = fh_k1_bucket_width(d->zeros.data(), d->N);
    if (table && fh_k1_bucket_table(d->zeros.data(), d->N, d->N, b0, b1, table) != 0)
        return fail(FH_ERR_INVALID, "fh_dht_bucket_tables: table construction failed");
    return FH_OK;
}

// ---- contexts -------------------------------------------------------------------------------------------------
static void load_k1_env(fh_ctx *c);  // (the FRANK_AMD_K1_* switches, read once per context)
int fh_ctx_create(const fh_dht *dht, int device, fh_ctx **out) {
    if (!dht || !out) return fail(FH_ERR_INVALID, "fh_ctx_create: NULL argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(FH_ERR_HIP, "no HIP device available: frank_amd has no CPU fallback for device work");
    if (device < 0 || device >= ndev) return fail(FH_ERR_INVALID, "fh_ctx_create: device %d of %d", device, ndev);
    HIP_TRY(hipSetDevice(device));
    // released to the caller only on success: every early return below destroys what has been created so far
    std::unique_ptr<fh_ctx, void (*)(fh_ctx *)> guard(new fh_ctx(), fh_ctx_destroy);
    fh_ctx *c = guard.get();
    c->dht = dht;
    c->device = device;
    load_k1_env(c);
    const int N = c->N = dht->N;
    const size_t NN = (size_t)N * N;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    c->num_cu = prop.multiProcessorCount;
    HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&c->ev_bin0, hipEventDisableSystemFence));
    HIP_TRY(hipEventCreateWithFlags(&c->ev_bin1, hipEventDisableSystemFence));
    ROC_TRY(rocblas_create_handle(&c->blas));
    ROC_TRY(rocblas_set_stream(c->blas, c->stream));
    ROC_TRY(rocblas_set_pointer_mode(c->blas, rocblas_pointer_mode_host));

    // constants
    std::vector<double> tab(FH_J0_TABLE_DOUBLES), Y(NN), pf(N), pb(N);
    fh_j0_fill_table(tab.data());
    fh_dht_self_coefficients(*dht, Y.data());
    const double norm_f = 1 / (M_PI * dht->Qmax * dht->Qmax), norm_b = 1 / (M_PI * dht->Rmax * dht->Rmax);
    for (int k = 0; k < N; ++k) {
        pf[k] = norm_f * dht->scale_factor[k];  // (norm * self._scale_factor), hankel.py:201
        pb[k] = norm_b * dht->scale_factor[k];
    }
    HIP_TRY(c->zeros.alloc(N + 1));
    HIP_TRY(c->j0_table.alloc(tab.size()));
    HIP_TRY(c->Y.alloc(NN));
    HIP_TRY(c->Ykm.alloc(NN));
    HIP_TRY(c->q.alloc(N));
    HIP_TRY(c->pref_fwd.alloc(N));
    HIP_TRY(c->pref_bwd.alloc(N));
    HIP_TRY(hipMemcpy(c->zeros.p, dht->zeros.data(), sizeof(double) * (N + 1), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->j0_table.p, tab.data(), sizeof(double) * tab.size(), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->Y.p, Y.data(), sizeof(double) * NN, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->Ykm.p, dht->Ykm.data(), sizeof(double) * NN, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->q.p, dht->q.data(), sizeof(double) * N, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->pref_fwd.p, pf.data(), sizeof(double) * N, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->pref_bwd.p, pb.data(), sizeof(double) * N, hipMemcpyHostToDevice));

    // K1 workspaces.  v2 (bin_gram2.hip, design block generated on the matrix pipe) covers N <= 383; beyond that, and
    // for the debris model, rows go to memory and rocBLAS forms the Gram.  Development switches: FRANK_AMD_K1=v1 selects
    // the first kernel (J0 on the vector ALU, N <= 303), FRANK_AMD_K1=wide forces the rows + dgemm path.
    const char *k1env = getenv("FRANK_AMD_K1");
    const bool want_v1 = k1env && !strcmp(k1env, "v1"), want_wide = k1env && !strcmp(k1env, "wide");
    // 511 < N <= 1023: the moments path has no register-resident kernel in it (bin_prepass.hip: one workgroup per output
    // tile), so it runs at any basis size; what cannot go through moments there (debris model, FRANK_AMD_K1=rows) takes the
    // rows-to-memory + rocBLAS path
    const bool generic = !want_v1 && !want_wide && fh_k1v2_nbt_for(N) == 0 && N <= 1023;
    c->v2 = !want_v1 && !want_wide && (fh_k1v2_nbt_for(N) != 0 || generic);
    c->rows_ok = !generic;
    c->k1_moments = !(k1env && !strcmp(k1env, "rows"));  // FRANK_AMD_K1=rows: the v2 kernel on the visibilities themselves
    c->NBT = want_wide ? 0 : (generic ? (N + 1 + 15) / 16 : (c->v2 ? fh_k1v2_nbt_for(N) : fh_k1_nbt_for(N)));
    HIP_TRY(hipEventCreateWithFlags(&c->ev_pre0, hipEventDisableSystemFence));
    HIP_TRY(hipEventCreateWithFlags(&c->ev_loop0, hipEventDisableSystemFence));
    HIP_TRY(hipEventCreateWithFlags(&c->ev_loop1, hipEventDisableSystemFence));
    if (c->NBT) {
        const int G = c->num_cu > 0 ? c->num_cu : 256;
        if (generic) {
            c->ntiles = c->NBT * (c->NBT + 1) / 2;
            c->nparts = 1;
            c->XS = fh_k1v2_xstride(c->NBT);
            c->k1_delta = fh_k1_bucket_width(dht->zeros.data(), N);
            HIP_TRY(c->k1_info.alloc(4));
            HIP_TRY(hipMemset(c->k1_info.p, 0, 4 * sizeof(int)));
        } else if (c->v2) {
            c->ntiles = fh_k1v2_ntiles(c->NBT);
            c->nparts = fh_k1v2_nparts(c->NBT);
            c->XS = fh_k1v2_xstride(c->NBT);
            c->k1_delta = fh_k1_bucket_width(dht->zeros.data(), N);
            // split the CUs over the parts by their MFMA work per 16 rows: 4 per tile + 3 per generated column block
            double wsum = 0, wp[3] = {0, 0, 0};
            for (int P = 0; P < c->nparts; ++P)
                wsum += wp[P] = 4.0 * fh_k1v2_part_ntiles(c->NBT, P) + 3.0 * (c->NBT - fh_k1v2_part_block0(c->NBT, P));
            int left = G;
            for (int P = 0; P < c->nparts; ++P) {
                int g = P == c->nparts - 1 ? left : (int)llround(G * wp[P] / wsum);
                if (g < 1) g = 1;
                if (g > left - (c->nparts - 1 - P)) g = left - (c->nparts - 1 - P);
                c->part_blocks[P] = g;
                left -= g;
            }
            // (the per-workgroup Gram slabs -- 100 MB at N = 300 -- are allocated when the rows path first runs: ensure_slabs)
            HIP_TRY(c->k1_info.alloc(4));
            HIP_TRY(hipMemset(c->k1_info.p, 0, 4 * sizeof(int)));  // ([3]: the ticket of bucket_scan_kernel starts at zero)
        } else {
            c->ntiles = fh_k1_ntiles(c->NBT);
            c->nparts = fh_k1_nparts(c->NBT);
            if (c->nparts == 1) {
                c->part_blocks[0] = G;
                c->part_blocks[1] = 0;
            } else {
                // split the CUs in proportion to the parts' work per visibility: their tiles (MFMA) plus the J0 column
                // blocks they have to evaluate (part 0 all 19, part 1 the last 12); block weight from sweeps at N = 300
                const int t0 = fh_k1_part_ntiles(c->NBT, 0), t1 = fh_k1_part_ntiles(c->NBT, 1);
                const double w0 = t0 + 3.3 * c->NBT, w1 = t1 + 3.3 * (c->NBT - 7);
                int g0 = (int)llround((double)G * w0 / (w0 + w1));
                if (const char *e = getenv("FRANK_AMD_K1_SPLIT")) g0 = atoi(e);  // development: workgroups of part 0
                if (g0 < 1) g0 = 1;
                if (g0 > G - 1) g0 = G - 1;
                c->part_blocks[0] = g0;
                c->part_blocks[1] = G - g0;
            }
            for (int P = 0; P < c->nparts; ++P)
                HIP_TRY(c->partials[P].alloc((size_t)c->part_blocks[P] * fh_k1_part_ntiles(c->NBT, P) * 256));
        }
        c->deproject_blocks = 8 * G;
        HIP_TRY(c->partial_scalars.alloc((size_t)c->deproject_blocks * 4));
        HIP_TRY(c->work_counter.alloc(4));
        HIP_TRY(c->stats_sum.alloc((size_t)c->ntiles * 256 + 2));
        HIP_TRY(c->reduce_scratch.alloc(8 * (size_t)c->ntiles * 256));
        HIP_TRY(c->stats_minmax.alloc(2));
        HIP_TRY(c->a_scale.alloc(N));
        HIP_TRY(c->sumwV2.alloc(1));
        c->tail_offset = (size_t)c->ntiles * 256;
    } else {
        c->wide = true;
        const int G = c->num_cu > 0 ? c->num_cu : 256;
        const size_t N1 = (size_t)N + 1;
        c->deproject_blocks = 8 * G;
        c->wide_rows = 65536;
        HIP_TRY(c->partial_scalars.alloc((size_t)c->deproject_blocks * 4));
        HIP_TRY(c->stats_sum.alloc(N1 * N1 + 2));
        HIP_TRY(c->stats_minmax.alloc(2));
        HIP_TRY(c->a_scale.alloc(N));
        HIP_TRY(c->sumwV2.alloc(1));
        HIP_TRY(c->wide_X.alloc((size_t)c->wide_rows * N1));
        c->tail_offset = N1 * N1;
    }
    // K2
    HIP_TRY(c->M.alloc(NN));
    HIP_TRY(c->j.alloc(N));
    HIP_TRY(c->W.alloc(NN));
    HIP_TRY(c->D.alloc(NN));
    HIP_TRY(c->Z.alloc(NN));
    HIP_TRY(c->p.alloc(N));
    HIP_TRY(c->p_old.alloc(N));
    HIP_TRY(c->mu.alloc(N));
    HIP_TRY(c->band_lu.alloc(5 * (size_t)N));
    HIP_TRY(c->flags.alloc(FIT_NFLAGS));
    HIP_TRY(c->info.alloc(1));
    HIP_TRY(hipMemset(c->info.p, 0, sizeof(int)));
    HIP_TRY(hipMemset(c->flags.p, 0, sizeof(int) * FIT_NFLAGS));
    // K2 v2: Y^-1 (cond(Y) ~ 1e2), q-space work buffers
    {
        c->NP = 16 * ((N + 1 + 15) / 16);  // at least one padding row: row N carries b (fit_loop.hip)
        const size_t PP = (size_t)c->NP * c->NP;
        // Y^-1 by LU on the device (the inverse of the column-major view is the row-major inverse): getrf, then getrs on the
        // identity.  NOT getri: rocSOLVER 3.32 (ROCm 7.2) returns a wrong inverse -- |inv A - I| = 1 with info = 0 -- for every
        // N = 127 mod 128 from 255 on (255, 383, 511, 639; checked on random matrices, round 3), which made the q-space
        // operands garbage, the first seed Cholesky "fail" and every fit of those sizes fall back to the slow route, silently.
        // The residual of the inverse is checked once, here, so that a library misbehaving at some other size cannot do that again.
        DevBuf<rocblas_int> ipiv;
        DevBuf<double> lu_y;
        HIP_TRY(ipiv.alloc(N));
        HIP_TRY(lu_y.alloc(NN));
        HIP_TRY(c->Yinv.alloc(NN));
        HIP_TRY(c->T1.alloc(NN));
        HIP_TRY(hipMemcpy(lu_y.p, Y.data(), sizeof(double) * NN, hipMemcpyHostToDevice));
        {
            std::vector<double> eye(NN, 0.0);
            for (int k = 0; k < N; ++k) eye[(size_t)k * N + k] = 1.0;
            HIP_TRY(hipMemcpy(c->Yinv.p, eye.data(), sizeof(double) * NN, hipMemcpyHostToDevice));
        }
        ROC_TRY(rocsolver_dgetrf(c->blas, N, N, lu_y.p, N, ipiv.p, c->info.p));
        int inv_info = 0;
        HIP_TRY(hipMemcpyAsync(&inv_info, c->info.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (inv_info != 0) return fail(FH_ERR_INVALID, "DHT coefficient matrix is singular (getrf info %d)", inv_info);
        ROC_TRY(rocsolver_dgetrs(c->blas, rocblas_operation_none, N, N, lu_y.p, N, ipiv.p, c->Yinv.p, N));
        {   // residual: (Y^-1 Y - I) in the row-major reading == the column-major product Y_buf * Yinv_buf
            const double one = 1.0, zero = 0.0;
            HIP_TRY(hipStreamSynchronize(c->stream));  // (the solve still reads the factors in lu_y)
            HIP_TRY(hipMemcpy(lu_y.p, Y.data(), sizeof(double) * NN, hipMemcpyHostToDevice));
            ROC_TRY(rocblas_dgemm(c->blas, rocblas_operation_none, rocblas_operation_none, N, N, N, &one, lu_y.p, N, c->Yinv.p, N, &zero,
                                  c->T1.p, N));
            std::vector<double> r(NN);
            HIP_TRY(hipMemcpyAsync(r.data(), c->T1.p, sizeof(double) * NN, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            double worst = 0.0;
            for (int a = 0; a < N; ++a)
                for (int b = 0; b < N; ++b) worst = std::max(worst, std::fabs(r[(size_t)a * N + b] - (a == b ? 1.0 : 0.0)));
            if (!(worst < 1e-9))
                return fail(FH_ERR_HIP, "the device inverse of the DHT coefficient matrix is wrong (|Y^-1 Y - I| = %.3g at N = %d)", worst, N);
        }
        HIP_TRY(c->Araw.alloc(NN));
        HIP_TRY(c->Aq.alloc(PP));
        HIP_TRY(c->Cq.alloc(PP));
        HIP_TRY(c->Wq.alloc(PP));
        HIP_TRY(c->WdT.alloc(fh_k2_exchange_doubles(c->NP)));
        HIP_TRY(c->cs.alloc(fh_k2_cs_doubles(c->NP)));
        HIP_TRY(hipMemsetAsync(c->WdT.p, 0, sizeof(double) * fh_k2_exchange_doubles(c->NP), c->stream));  // (control words of the cluster mode: zero between fits)
        HIP_TRY(hipMemsetAsync(c->Cq.p, 0, sizeof(double) * PP, c->stream));
        HIP_TRY(hipMemsetAsync(c->Wq.p, 0, sizeof(double) * PP, c->stream));
        HIP_TRY(c->bq.alloc(N));
        HIP_TRY(c->mu_out.alloc(N));
        HIP_TRY(c->p_out.alloc(N));
        HIP_TRY(c->p_init.alloc(N));
        HIP_TRY(c->loop_result.alloc(2));
        const char *env = getenv("FRANK_AMD_K2");
        c->use_rocsolver_loop = env && strcmp(env, "rocsolver") == 0;
    }
    const int rc = fh_bin_reset(c);
    if (rc != FH_OK) return rc;
    g_warn.clear();
    if (hw_queues_env() < kHwQueuesWanted) {
        char buf[320];
        snprintf(buf, sizeof buf, "GPU_MAX_HW_QUEUES=%d: pipelined fits (fh_fit_submit) put their launches on up to six streams beside "
                 "the binning stream; with fewer than %d hardware queues HIP lets streams share a queue and their kernels serialise. "
                 "Call fh_init() -- or export GPU_MAX_HW_QUEUES=24 -- before the first HIP call of the process.", hw_queues_env(),
                 kHwQueuesWanted);
        g_warn = buf;
    }
    *out = guard.release();
    return FH_OK;
}

void fh_ctx_destroy(fh_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (int i = 0; i < c->n_launch_streams; ++i) {
        (void)hipStreamSynchronize(c->launch_streams[i]);
        (void)hipStreamDestroy(c->launch_streams[i]);
    }
    for (auto &b : c->batches) {
        if (b.ready) (void)hipEventDestroy(b.ready);
        if (b.done) (void)hipEventDestroy(b.done);
    }
    if (c->slot_out_host) (void)hipHostFree(c->slot_out_host);
    if (c->slot_result_host) (void)hipHostFree(c->slot_result_host);
    if (c->blas) rocblas_destroy_handle(c->blas);
    if (c->ev_bin0) (void)hipEventDestroy(c->ev_bin0);
    if (c->ev_bin1) (void)hipEventDestroy(c->ev_bin1);
    if (c->ev_pre0) (void)hipEventDestroy(c->ev_pre0);
    if (c->ev_loop0) (void)hipEventDestroy(c->ev_loop0);
    if (c->ev_loop1) (void)hipEventDestroy(c->ev_loop1);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}
int fh_ctx_synchronize(fh_ctx *c) {
    if (!c) return fail(FH_ERR_INVALID, "ctx is NULL");
    HIP_TRY(hipStreamSynchronize(c->stream));
    return FH_OK;
}
void *fh_ctx_stream(fh_ctx *c) { return c ? (void *)c->stream : nullptr; }

// ---- a3/a7: H(q), predict ------------------------------------------------------------------------------------
static int stage_q(fh_ctx *c, const double *q, int64_t n) {
    if (c->scratch_q.n < (size_t)n) HIP_TRY(c->scratch_q.alloc((size_t)n));
    HIP_TRY(hipMemcpyAsync(c->scratch_q.p, q, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
    return FH_OK;
}

int fh_dht_coefficients(fh_ctx *c, const double *q, int64_t n, int direction, double scale, double *H) {
    if (!c || !q || !H || n < 0) return fail(FH_ERR_INVALID, "fh_dht_coefficients: bad argument");
    if (direction != 0 && direction != 1) return fail(FH_ERR_INVALID, "direction must be one of ['forward', 'backward']");
    if (n == 0) return FH_OK;
    HIP_TRY(hipSetDevice(c->device));
    const int N = c->N;
    int rc = stage_q(c, q, n);
    if (rc) return rc;
    if (c->scratch_out.n < (size_t)n * N) HIP_TRY(c->scratch_out.alloc((size_t)n * N));
    const double inv = direction == 0 ? 1. / c->dht->Qmax : 1. / c->dht->Rmax;  // hankel.py:189,192
    HIP_TRY(fh_k1_launch_coefficients(c->scratch_q.p, n, N, c->zeros.p, direction == 0 ? c->pref_fwd.p : c->pref_bwd.p,
                                      inv, scale, c->j0_table.p, c->scratch_out.p, c->stream));
    HIP_TRY(hipMemcpyAsync(H, c->scratch_out.p, sizeof(double) * (size_t)n * N, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return FH_OK;
}

static int k1v2_ensure_table(fh_ctx *c, int nb_needed);

int fh_predict_visibilities(fh_ctx *c, const double *q, int64_t n, const double *I, double scale, double *V) {
    if (!c || !q || !I || !V || n < 0) return fail(FH_ERR_INVALID, "fh_predict_visibilities: bad argument");
    if (n == 0) return FH_OK;
    HIP_TRY(hipSetDevice(c->device));
    const int N = c->N;
    int rc = stage_q(c, q, n);
    if (rc) return rc;
    if (c->scratch_out.n < (size_t)n) HIP_TRY(c->scratch_out.alloc((size_t)n));
    if (c->scratch_I.n < (size_t)N + 1) HIP_TRY(c->scratch_I.alloc((size_t)N + 1));  // (+ one scratch double)
    HIP_TRY(hipMemcpyAsync(c->scratch_I.p, I, sizeof(double) * N, hipMemcpyHostToDevice, c->stream));
    if (c->v2 && n >= 4096) {
        // through the bucket tables of bin_gram: 12 coefficients per bucket, then a degree-11 polynomial per visibility
        // instead of N Bessel evaluations (bin_gram2.hip); small calls keep the direct kernel (no table to build)
        double *mx = c->scratch_I.p + N;
        HIP_TRY(fh_k1v2_launch_max(c->scratch_q.p, n, mx, c->stream));
        double qmax = 0.0;
        HIP_TRY(hipMemcpyAsync(&qmax, mx, sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        const double smax = qmax / c->dht->Qmax;
        if (qmax == qmax && smax / c->k1_delta < 15000.0) {
            const int nb = (int)(smax / c->k1_delta) + 2;
            rc = k1v2_ensure_table(c, nb);
            if (rc) return rc;
            if (c->predict_coef.n < (size_t)c->k1_nb_built * FH_K1_TERMS)
                HIP_TRY(c->predict_coef.alloc((size_t)c->k1_nb_built * FH_K1_TERMS));
            HIP_TRY(fh_k1v2_launch_predict(c->k1_table.p, c->XS, N, nb, c->pref_fwd.p, c->scratch_I.p, scale, c->predict_coef.p,
                                           c->scratch_q.p, n, 1. / c->dht->Qmax, c->k1_delta, c->scratch_out.p, c->stream));
            HIP_TRY(hipMemcpyAsync(V, c->scratch_out.p, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            return FH_OK;
        }
    }
    HIP_TRY(fh_k1_launch_predict(c->scratch_q.p, n, N, c->zeros.p, c->pref_fwd.p, 1. / c->dht->Qmax, scale,
                                 c->scratch_I.p, c->j0_table.p, c->scratch_out.p, c->stream));
    HIP_TRY(hipMemcpyAsync(V, c->scratch_out.p, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return FH_OK;
}

// ---- visibility tables -------------------------------------------------------------------------------------------
int fh_vis_upload(int device, const double *u, const double *v, const double *Vre, const double *Vim, const double *w,
                  int64_t n_w, int64_t n, fh_vis **out) {
    if (!out || n < 0 || (n > 0 && (!u || !v || !Vre || !w))) return fail(FH_ERR_INVALID, "fh_vis_upload: bad argument");
    if (n_w != 1 && n_w != n) return fail(FH_ERR_INVALID, "fh_vis_upload: weights must have 1 or n entries");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(FH_ERR_HIP, "no HIP device available: frank_amd has no CPU fallback for device work");
    HIP_TRY(hipSetDevice(device));
    fh_vis *t = new fh_vis();
    t->device = device;
    t->n = n;
    t->w_scalar = (n_w == 1 && n != 1) ? 1 : 0;
    t->has_im = Vim ? 1 : 0;
    const size_t nn = (size_t)(n > 0 ? n : 1);
    hipError_t e = t->u.alloc_pooled(nn, device);
    if (e == hipSuccess) e = t->v.alloc_pooled(nn, device);
    if (e == hipSuccess) e = t->Vre.alloc_pooled(nn, device);
    if (e == hipSuccess && Vim) e = t->Vim.alloc_pooled(nn, device);
    if (e == hipSuccess) e = t->w.alloc_pooled(t->w_scalar ? 1 : nn, device);
    if (e != hipSuccess) {
        delete t;
        return fail(FH_ERR_NOMEM, "fh_vis_upload: hipMalloc failed: %s", hipGetErrorString(e));
    }
    if (n > 0) {
        const size_t b = sizeof(double) * (size_t)n;
        e = hipMemcpy(t->u.p, u, b, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(t->v.p, v, b, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(t->Vre.p, Vre, b, hipMemcpyHostToDevice);
        if (e == hipSuccess && Vim) e = hipMemcpy(t->Vim.p, Vim, b, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(t->w.p, w, t->w_scalar ? sizeof(double) : b, hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            delete t;
            return fail(FH_ERR_HIP, "fh_vis_upload: copy failed: %s", hipGetErrorString(e));
        }
    }
    *out = t;
    return FH_OK;
}
// The same table from a complex128 array as NumPy holds it (re, im interleaved): one contiguous copy of 16 n bytes and a split on
// the device instead of two strided host copies into separate columns (30 ms of a 45 ms mapping call at 1e7 visibilities).
int fh_vis_upload_c128(int device, const double *u, const double *v, const double *Vc, const double *w, int64_t n_w, int64_t n,
                       fh_vis **out) {
    if (!out || n < 0 || (n > 0 && (!u || !v || !Vc || !w))) return fail(FH_ERR_INVALID, "fh_vis_upload_c128: bad argument");
    if (n_w != 1 && n_w != n) return fail(FH_ERR_INVALID, "fh_vis_upload_c128: weights must have 1 or n entries");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(FH_ERR_HIP, "no HIP device available: frank_amd has no CPU fallback for device work");
    HIP_TRY(hipSetDevice(device));
    std::unique_ptr<fh_vis> t(new fh_vis());
    t->device = device;
    t->n = n;
    t->w_scalar = (n_w == 1 && n != 1) ? 1 : 0;
    t->has_im = 1;
    const size_t nn = (size_t)(n > 0 ? n : 1);
    DevBuf<double> tmp;
    if (t->u.alloc_pooled(nn, device) != hipSuccess || t->v.alloc_pooled(nn, device) != hipSuccess ||
        t->Vre.alloc_pooled(nn, device) != hipSuccess || t->Vim.alloc_pooled(nn, device) != hipSuccess ||
        t->w.alloc_pooled(t->w_scalar ? 1 : nn, device) != hipSuccess || tmp.alloc_pooled(2 * nn, device) != hipSuccess)
        return fail(FH_ERR_NOMEM, "fh_vis_upload_c128: hipMalloc failed");
    if (n > 0) {
        const size_t b = sizeof(double) * (size_t)n;
        HIP_TRY(hipMemcpy(tmp.p, Vc, 2 * b, hipMemcpyHostToDevice));
        HIP_TRY(fh_launch_split_complex(tmp.p, n, t->Vre.p, t->Vim.p, nullptr));
        HIP_TRY(hipMemcpy(t->u.p, u, b, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(t->v.p, v, b, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(t->w.p, w, t->w_scalar ? sizeof(double) : b, hipMemcpyHostToDevice));
        HIP_TRY(hipDeviceSynchronize());  // (the split runs on the null stream; tmp goes away with this scope)
    }
    *out = t.release();
    return FH_OK;
}

int fh_vis_upload_f32(int device, const float *u, const float *v, const float *Vre, const float *Vim, const float *w,
                      int64_t n_w, int64_t n, fh_vis **out) {
    if (!out || n < 0 || (n > 0 && (!u || !v || !Vre || !w))) return fail(FH_ERR_INVALID, "fh_vis_upload_f32: bad argument");
    if (n_w != 1 && n_w != n) return fail(FH_ERR_INVALID, "fh_vis_upload_f32: weights must have 1 or n entries");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(FH_ERR_HIP, "no HIP device available: frank_amd has no CPU fallback for device work");
    HIP_TRY(hipSetDevice(device));
    fh_vis *t = new fh_vis();
    t->device = device;
    t->n = n;
    t->f32 = true;
    t->w_scalar = (n_w == 1 && n != 1) ? 1 : 0;
    t->has_im = Vim ? 1 : 0;
    const size_t nn = (size_t)(n > 0 ? n : 1);
    hipError_t e = t->u32.alloc(nn);
    if (e == hipSuccess) e = t->v32.alloc(nn);
    if (e == hipSuccess) e = t->Vre32.alloc(nn);
    if (e == hipSuccess && Vim) e = t->Vim32.alloc(nn);
    if (e == hipSuccess) e = t->w32.alloc(t->w_scalar ? 1 : nn);
    if (e != hipSuccess) {
        delete t;
        return fail(FH_ERR_NOMEM, "fh_vis_upload_f32: hipMalloc failed: %s", hipGetErrorString(e));
    }
    if (n > 0) {
        const size_t b = sizeof(float) * (size_t)n;
        e = hipMemcpy(t->u32.p, u, b, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(t->v32.p, v, b, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(t->Vre32.p, Vre, b, hipMemcpyHostToDevice);
        if (e == hipSuccess && Vim) e = hipMemcpy(t->Vim32.p, Vim, b, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(t->w32.p, w, t->w_scalar ? sizeof(float) : b, hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            delete t;
            return fail(FH_ERR_HIP, "fh_vis_upload_f32: copy failed: %s", hipGetErrorString(e));
        }
    }
    *out = t;
    return FH_OK;
}
// empties the cache of freed table columns (DevBuf::alloc_pooled: at most 1.5 GB of device memory held between calls)
int fh_cache_release(void) {
    pool_clear();
    return FH_OK;
}

void fh_vis_destroy(fh_vis *vis) {
    if (!vis) return;
    (void)hipSetDevice(vis->device);
    // (hipFree waited for the device; the columns now go back to a cache and may be handed out again at once: kernels of any
    //  stream that still read them must have ended)
    (void)hipDeviceSynchronize();
    delete vis;
}
int64_t fh_vis_size(const fh_vis *vis) { return vis ? vis->n : 0; }

int fh_vis_set_multiplicity(fh_vis *vis, const int32_t *counts) {
    if (!vis) return fail(FH_ERR_INVALID, "fh_vis_set_multiplicity: vis is NULL");
    vis->mult_gen = g_vis_serial.fetch_add(1);  // the baseline range of the drawn rows is not the one a context remembers
    if (!counts) {
        vis->use_mult = false;
        return FH_OK;
    }
    HIP_TRY(hipSetDevice(vis->device));
    const size_t nn = (size_t)(vis->n > 0 ? vis->n : 1);
    if (!vis->mult.p && vis->mult.alloc(nn) != hipSuccess) return fail(FH_ERR_NOMEM, "fh_vis_set_multiplicity: hipMalloc failed");
    if (vis->n > 0) HIP_TRY(hipMemcpy(vis->mult.p, counts, sizeof(int) * (size_t)vis->n, hipMemcpyHostToDevice));
    vis->use_mult = true;
    return FH_OK;
}

// ---- K1 ----------------------------------------------------------------------------------------------------------
// the rows-to-memory + dgemm path is taken for N > 303 always and for the debris model at any N
static bool use_wide(const fh_ctx *c) {
    return c->wide || (c->debris && !c->v2) || (c->v2 && !c->rows_ok && (c->debris || !c->k1_moments));
}
static double *dense_gram(fh_ctx *c) { return c->wide ? c->stats_sum.p : c->wide_G.p; }  // (N+1)^2 + 2 scalars
static size_t dense_tail(const fh_ctx *c) { return ((size_t)c->N + 1) * ((size_t)c->N + 1); }

// slabs of the rows path (bin_gram2_kernel: every workgroup holds all tiles of its part), on first use
static int ensure_slabs(fh_ctx *c) {
    if (!c->v2) return FH_OK;
    for (int P = 0; P < c->nparts; ++P)
        if (!c->partials[P].p &&
            c->partials[P].alloc((size_t)c->part_blocks[P] * fh_k1v2_part_ntiles(c->NBT, P) * 256) != hipSuccess)
            return fail(FH_ERR_NOMEM, "hipMalloc of the Gram slabs failed");
    return FH_OK;
}

static int ensure_wide(fh_ctx *c) {
    if (c->wide_X.p) return FH_OK;
    const size_t N1 = (size_t)c->N + 1;
    c->wide_rows = 65536;
    if (c->wide_X.alloc((size_t)c->wide_rows * N1) != hipSuccess || c->wide_G.alloc(N1 * N1 + 2) != hipSuccess)
        return fail(FH_ERR_NOMEM, "device allocation for the rows + dgemm path failed");
    HIP_TRY(hipMemsetAsync(c->wide_G.p, 0, sizeof(double) * c->wide_G.n, c->stream));  // (fh_bin_reset came before it existed)
    return FH_OK;
}

int fh_ctx_set_scale_height(fh_ctx *c, const double *H2) {
    if (!c) return fail(FH_ERR_INVALID, "ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    if (!H2) {
        c->debris = false;
        return FH_OK;
    }
    for (int k = 0; k < c->N; ++k)
        if (!(H2[k] >= 0.0)) return fail(FH_ERR_INVALID, "H2[%d] = %g: the squared scale height must be >= 0", k, H2[k]);
    if (!c->v2) {  // the fused kernel scales its generated design block itself; otherwise rows go to memory + rocBLAS
        int rc = ensure_wide(c);
        if (rc) return rc;
    }
    if (!c->debris_H2.p) HIP_TRY(c->debris_H2.alloc(c->N));
    HIP_TRY(hipMemcpy(c->debris_H2.p, H2, sizeof(double) * c->N, hipMemcpyHostToDevice));
    c->debris = true;
    return FH_OK;
}

// ---- K1 v2: Taylor tables of the buckets (j0_buckets.h) ------------------------------------------------------------
// The host copy is shared by every context of the same basis size in the process (the zeros depend on N only) and only
// ever grows; a context's device copy is re-uploaded when a table with more buckets is needed.
namespace {
struct K1TableCache {
    std::mutex mu;
    std::map<std::pair<int, int>, std::shared_ptr<std::vector<double>>> tabs;  // (N, XS) -> [nb][12][XS]
} g_k1_tables;
}  // namespace

static int k1v2_upload_table32(fh_ctx *c, const std::vector<double> &tab, int nb) {
    const size_t n = (size_t)nb * FH_K1_TERMS * c->XS;
    std::vector<float> t32(n);
    for (size_t i = 0; i < n; ++i) t32[i] = (float)tab[i];
    c->k1_nb_built32 = 0;
    if (c->k1_table32.alloc(n) != hipSuccess) return fail(FH_ERR_NOMEM, "hipMalloc of the fp32 bucket tables failed");
    HIP_TRY(hipMemcpy(c->k1_table32.p, t32.data(), sizeof(float) * n, hipMemcpyHostToDevice));
    c->k1_nb_built32 = nb;
    return FH_OK;
}

static int k1v2_ensure_table(fh_ctx *c, int nb_needed) {
    if (nb_needed <= c->k1_nb_built && (!c->arith32 || nb_needed <= c->k1_nb_built32)) return FH_OK;
    const size_t per = (size_t)FH_K1_TERMS * c->XS;
    // 25 % headroom so that fits of similar tables do not rebuild; bounded so that one absurd baseline cannot ask for
    // an absurd table (s = q/Qmax < 1 whenever the q-range check of statistical_models.py:526 would pass)
    int nb_new = nb_needed + nb_needed / 4 + 8;
    const size_t cap_bytes = (size_t)4 << 30;
    if ((size_t)nb_needed * per * sizeof(double) > cap_bytes)
        return fail(FH_ERR_UNSUPPORTED, "baselines reach %.1f x Qmax: the bucket tables of bin_gram would need %.1f GB",
                    nb_needed * c->k1_delta, nb_needed * per * 8e-9);
    if ((size_t)nb_new * per * sizeof(double) > cap_bytes) nb_new = nb_needed;
    std::shared_ptr<std::vector<double>> tab;
    {
        std::lock_guard<std::mutex> lk(g_k1_tables.mu);
        auto &slot = g_k1_tables.tabs[{c->N, c->XS}];
        if (!slot) slot = std::make_shared<std::vector<double>>();
        const int have = (int)(slot->size() / per);
        if (have < nb_new) {
            // a NEW vector (readers of the old one keep their shared_ptr): old buckets copied, new ones computed
            auto grown = std::make_shared<std::vector<double>>((size_t)nb_new * per);
            if (have) memcpy(grown->data(), slot->data(), sizeof(double) * (size_t)have * per);
            if (fh_k1_bucket_table(c->dht->zeros.data(), c->N, c->XS, have, nb_new, grown->data() + (size_t)have * per) != 0)
                return fail(FH_ERR_INVALID, "fh_k1_bucket_table failed");
            slot = grown;
        }
        tab = slot;
    }
    const int nb_up = (int)(tab->size() / per);
    HIP_TRY(hipStreamSynchronize(c->stream));  // nothing in flight may still read the old device table
    c->k1_nb_built = 0;  // DevBuf::alloc releases the old table first: after a failed allocation there is none
    if (c->k1_table.alloc((size_t)nb_up * per) != hipSuccess) return fail(FH_ERR_NOMEM, "hipMalloc of the bucket tables failed");
    HIP_TRY(hipMemcpy(c->k1_table.p, tab->data(), sizeof(double) * (size_t)nb_up * per, hipMemcpyHostToDevice));
    c->k1_nb_built = nb_up;
    if (c->arith32) return k1v2_upload_table32(c, *tab, nb_up);
    return FH_OK;
}

// The two fills of fh_bin_reset.  On a device whose other compute units run fit loops every kernel boundary of the binning
// stream costs ~40 us (the L2 write-backs between dependent kernels find the caches full of the loops' dirty tiles): a step of
// the pipeline was sixteen kernels, two of them these fills.  fh_bin_reset only notes that the sums are to start from
// zero; the last kernel of the moments path (vr_finish_kernel) then stores instead of adding; every other reader or writer
// of the sums calls settle_reset() first.
static int settle_reset(fh_ctx *c) {
    if (!c->stats_reset_pending) return FH_OK;
    c->stats_reset_pending = false;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemsetAsync(c->stats_sum.p, 0, sizeof(double) * c->stats_sum.n, c->stream));
    // (-qmin, qmax) under max start at -infinity: 0xFFF0000000000000 is not a byte pattern, but 0xFFFFFFFF words are a
    // NaN, and fmax(NaN, x) = x -- the same neutral element, set without a host-side source buffer or a wait
    HIP_TRY(hipMemsetAsync(c->stats_minmax.p, 0xFF, 2 * sizeof(double), c->stream));
    return FH_OK;
}

int fh_bin_reset(fh_ctx *c) {
    if (!c) return fail(FH_ERR_INVALID, "ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    c->stats_reset_pending = true;
    if (c->wide_G.p) HIP_TRY(hipMemsetAsync(c->wide_G.p, 0, sizeof(double) * c->wide_G.n, c->stream));
    c->have_device_Mj = false;
    return FH_OK;
}

// fit_loop kernels of earlier fits that are still RUNNING each hold a CU (a slot stays "busy" until it is collected, long
// after its kernel has finished: counting those would leave CUs idle)
static int running_fit_loops(fh_ctx *c) {
    int running = 0;
    if (c->slots_busy > 0)
        for (auto &b : c->batches)
            if (b.active && b.launched && hipEventQuery(b.done) == hipErrorNotReady) running += b.n;
    (void)hipGetLastError();  // hipErrorNotReady is not an error here
    return running;
}

// K1 v2: deproject -> (host: baseline range, bucket tables) -> bucket sort -> bin_gram2 -> slab reduction.
// The one host round trip (64 KB of per-block scalars) is what _check_uv_range needs before any binning in the reference
// too (statistical_models.py:166-169); it costs the stream ~20 us of idle time per call.
static int bin_visibilities_v4(fh_ctx *c, BinParams &p, int64_t count, unsigned long long vis_serial,
                               unsigned long long mult_gen);

static int bin_visibilities_v2(fh_ctx *c, BinParams &p, int64_t count, unsigned long long vis_serial,
                               unsigned long long mult_gen) {
    if (count > 0x7fffffff - 16 * 65536) return fail(FH_ERR_UNSUPPORTED, "more than 2^31 visibilities in one call: split it");
    // the default: the rows of a bucket enter the Gram through 12 x 12 moments (bin_prepass.hip + bin_gram2.hip); not for the
    // debris model (its design block is not a product of a row factor and a column factor) and not in single precision
    if (c->k1_moments && !c->debris && !c->arith32) return bin_visibilities_v4(c, p, count, vis_serial, mult_gen);
    c->hist_valid = false;  // (this path sorts through the same workspaces)
    {
        const int rcs = settle_reset(c);
        if (rcs) return rcs;
    }
    const size_t cnt1 = (size_t)(count > 0 ? count : 1);  // K1a scratch: 24 B per visibility (32 B with the debris model's kz^2)
    const size_t need = cnt1 * (c->debris ? 4 : 3);
    if (c->prep.n < need) HIP_TRY(c->prep.alloc(need));
    p.prep_s = c->prep.p;
    p.prep_sw = c->prep.p + cnt1;
    p.prep_swV = c->prep.p + 2 * cnt1;
    p.prep_k2 = c->debris ? c->prep.p + 3 * cnt1 : nullptr;
    int dblocks = (int)((count + 255) / 256);
    if (dblocks > c->deproject_blocks) dblocks = c->deproject_blocks;
    if (dblocks < 1) dblocks = 1;
    p.partial_scalars = c->partial_scalars.p;
    HIP_TRY(hipEventRecord(c->ev_pre0, c->stream));
    HIP_TRY(fh_k1_launch_deproject(p, dblocks, c->stream));
    const double gkey[6] = {p.dRA, p.dDec, p.cos_t, p.sin_t, p.cos_i, p.sin_i};
    const bool known = c->range_valid && c->range_vis == vis_serial && c->range_mult_gen == mult_gen && c->range_first == p.first && c->range_count == count &&
                       memcmp(gkey, c->range_geom, sizeof gkey) == 0 && !c->no_range_cache && !c->k1env.no_range_cache;
    // qmax_all: over every row of the range whatever its multiplicity -- the sort is sized from it, because rows drawn
    // zero times are still sorted (with weight 0) and must land in a bucket of their own argument
    double qmax = 0.0, qmin = INFINITY, qmax_all = 0.0;
    if (known) {  // same rows, same geometry: the range is the one read back last time, no host round trip
        qmin = c->prepass_qmin;
        qmax = c->prepass_qmax;
        qmax_all = c->prepass_qmax_all;
    } else {
        c->k1_scalars_host.resize((size_t)dblocks * 4);
        HIP_TRY(hipMemcpyAsync(c->k1_scalars_host.data(), c->partial_scalars.p, sizeof(double) * (size_t)dblocks * 4,
                               hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        for (int b = 0; b < dblocks; ++b) {
            const double m = c->k1_scalars_host[(size_t)b * 4 + 2], mn = c->k1_scalars_host[(size_t)b * 4 + 1];
            if (m > qmax) qmax = m;  // (-inf for blocks without rows; NaN baselines never win)
            if (mn < qmin) qmin = mn;
            const double ma = c->k1_scalars_host[(size_t)b * 4 + 3];
            if (ma > qmax_all) qmax_all = ma;
        }
    }
    if (!(qmax < INFINITY) || !(qmax_all < INFINITY)) return fail(FH_ERR_INVALID, "non-finite baseline in the visibility table");
    c->prepass_qmin = qmin;
    c->prepass_qmax = qmax;
    c->prepass_qmax_all = qmax_all;
    c->range_vis = vis_serial;
    c->range_mult_gen = mult_gen;
    c->range_first = p.first;
    c->range_count = count;
    memcpy(c->range_geom, gkey, sizeof gkey);
    c->range_valid = true;
    // statistical_models.py:166-169: the range check comes BEFORE the chunk loop -- nothing is binned for a table that fails it
    if (c->check_q_before_bin && c->dht->q[c->N - 1] < qmax)
        return fail(FH_ERR_QRANGE, "last collocation point %.3e < longest deprojected baseline %.3e", c->dht->q[c->N - 1], qmax);
    const double delta = c->k1_delta, inv_delta = 1.0 / delta;
    const double smax = qmax_all * p.inv_Qmax;
    if (smax * inv_delta > 2.0e9) return fail(FH_ERR_UNSUPPORTED, "baselines reach %.3g x Qmax", smax);
    const int nb = (int)(smax * inv_delta) + 2;  // one spare bucket: the device recomputes s * inv_delta itself
    if (nb > 16000)  // the sort keeps one counter per bucket in 64 KB of LDS
        return fail(FH_ERR_UNSUPPORTED, "baselines reach %.1f x Qmax (%d buckets of J0 arguments): cut the (u, v) distribution or "
                    "raise N", smax, nb);
    int rc = k1v2_ensure_table(c, nb);
    if (rc) return rc;
    rc = ensure_slabs(c);
    if (rc) return rc;
    // sort workspaces (grow on demand)
    int sblocks = (int)((count + 255) / 256);
    if (sblocks > 512) sblocks = 512;
    if (sblocks < 1) sblocks = 1;
    const size_t nrows = (size_t)count + 16 * (size_t)nb + 16, nchunks_max = nrows / 16 + 1;
    if (c->k1_rows.n < nrows * 4 && c->k1_rows.alloc(nrows * 4) != hipSuccess) return fail(FH_ERR_NOMEM, "hipMalloc (sorted rows) failed");
    if (c->k1_chunk_bucket.n < nchunks_max && c->k1_chunk_bucket.alloc(nchunks_max) != hipSuccess)
        return fail(FH_ERR_NOMEM, "hipMalloc (chunk map) failed");
    if (c->k1_hist.n < (size_t)sblocks * nb && c->k1_hist.alloc((size_t)sblocks * nb + 1024) != hipSuccess)
        return fail(FH_ERR_NOMEM, "hipMalloc (histograms) failed");
    if (c->k1_totals.n < (size_t)nb && c->k1_totals.alloc((size_t)nb + 256) != hipSuccess) return fail(FH_ERR_NOMEM, "hipMalloc failed");
    if (c->k1_starts.n < (size_t)nb + 1 && c->k1_starts.alloc((size_t)nb + 257) != hipSuccess) return fail(FH_ERR_NOMEM, "hipMalloc failed");
    SortParams sp{};
    sp.s = p.prep_s;
    sp.sw = p.prep_sw;
    sp.swV = p.prep_swV;
    sp.k2 = c->debris ? p.prep_k2 : nullptr;
    sp.n = count;
    sp.inv_delta = inv_delta;
    sp.delta = delta;
    sp.nb = nb;
    sp.blocks = sblocks;
    sp.hist = c->k1_hist.p;
    sp.totals = c->k1_totals.p;
    sp.starts = c->k1_starts.p;
    sp.info = c->k1_info.p;
    sp.rows = c->k1_rows.p;
    sp.chunk_bucket = c->k1_chunk_bucket.p;
    HIP_TRY(fh_k1v2_launch_sort(sp, c->stream));

    const int running = running_fit_loops(c);
    // throughput mode while fit_loop kernels hold CUs (a workgroup that starts late simply takes fewer runs); fits of a
    // pipeline are run-dependent in their last bits anyway; synchronous fits stay static = bitwise reproducible
    const bool dynamic = (running > 0 || c->slots_busy > 0 || c->k1env.dynamic) && !c->force_static;
    Bin2Params bp{};
    bp.N = c->N;
    bp.rows = c->k1_rows.p;
    bp.chunk_bucket = c->k1_chunk_bucket.p;
    bp.info = c->k1_info.p;
    bp.table = c->k1_table.p;
    bp.table32 = c->arith32 ? c->k1_table32.p : nullptr;
    bp.H2 = c->debris ? c->debris_H2.p : nullptr;
    bp.work_counter = dynamic ? c->work_counter.p : nullptr;
    if (dynamic) HIP_TRY(hipMemsetAsync(c->work_counter.p, 0, 4 * sizeof(int), c->stream));
    ReduceParams rp{};
    rp.nparts = c->nparts;
    rp.ntiles = c->ntiles;
    int reserve = running;
    if (reserve > c->num_cu / 4) reserve = c->num_cu / 4;
    int G = 0;
    for (int P = 0; P < c->nparts; ++P) G += c->part_blocks[P];
    for (int P = 0; P < 3; ++P) {
        int blocks = P < c->nparts ? c->part_blocks[P] : 0;
        if (P < c->nparts && reserve > 0 && G > 0) blocks -= (reserve * c->part_blocks[P] + G - 1) / G;
        if (P < c->nparts && blocks < 1) blocks = 1;
        bp.part_blocks[P] = blocks;
        bp.partials[P] = c->partials[P].p;
        rp.part_blocks[P] = blocks;
        rp.part_tile0[P] = P < c->nparts ? fh_k1v2_part_tile0(c->NBT, P) : 0;
        rp.part_ntiles[P] = P < c->nparts ? fh_k1v2_part_ntiles(c->NBT, P) : 0;
        rp.partials[P] = c->partials[P].p;
    }
    rp.partial_scalars = c->partial_scalars.p;
    rp.scratch = c->reduce_scratch.p;
    rp.scalar_blocks = dblocks;
    HIP_TRY(hipEventRecord(c->ev_bin0, c->stream));
    HIP_TRY(fh_k1v2_launch_bin(c->NBT, bp, c->stream));
    HIP_TRY(hipEventRecord(c->ev_bin1, c->stream));
    c->bin_timed = true;
    HIP_TRY(fh_k1_launch_reduce(rp, c->stats_sum.p, c->stats_minmax.p, c->stream));
    c->have_device_Mj = false;
    return FH_OK;
}

// The moments path (default): range (first sight of a table only) -> P1 histogram of (u, v) -> scan + layout -> P2 deproject +
// scatter -> P3 segment moments -> factor -> bin_gram2 on the 13 virtual rows per bucket -> slab reduction (bin_prepass.hip).
static int env_int(const char *name, int dflt) {
    const char *e = getenv(name);
    return e && *e ? atoi(e) : dflt;
}
static void load_k1_env(fh_ctx *c) {
    fh_ctx::K1Env e;
    e.unroll = env_int("FRANK_AMD_K1_UNROLL", 2) == 2 ? 2 : 1;
    e.seg = env_int("FRANK_AMD_K1_SEG", 4096);
    e.wpb = env_int("FRANK_AMD_K1_WPB", 0);
    e.blocks = env_int("FRANK_AMD_K1_BLOCKS", 0);
    e.vrwaves = env_int("FRANK_AMD_K1_VRWAVES", 8);
    e.vrsplit = env_int("FRANK_AMD_K1_VRSPLIT", 8);
    e.vrblocks = env_int("FRANK_AMD_K1_VRBLOCKS", 0);
    e.no_range_cache = getenv("FRANK_AMD_NO_RANGE_CACHE") != nullptr;
    e.safe_trig = getenv("FRANK_AMD_K1_SAFE_TRIG") != nullptr;
    e.no_hist_cache = getenv("FRANK_AMD_K1_NO_HIST_CACHE") != nullptr;
    const char *vr = getenv("FRANK_AMD_K1_VR");
    e.vr_slabs = vr && !strcmp(vr, "slabs");
    e.dynamic = getenv("FRANK_AMD_K1_DYNAMIC") != nullptr;
    if (const char *r = getenv("FRANK_AMD_K1_RESERVE_MULT")) e.reserve_mult = atof(r);
    c->k1env = e;
}
int fh_ctx_reload_env(fh_ctx *c) {
    if (!c) return fail(FH_ERR_INVALID, "fh_ctx_reload_env: NULL argument");
    load_k1_env(c);
    return FH_OK;
}
static int bin_visibilities_v4(fh_ctx *c, BinParams &p, int64_t count, unsigned long long vis_serial,
                               unsigned long long mult_gen) {
    PrepassParams P{};
    P.bin = p;
    P.partial_scalars = c->partial_scalars.p;
    const fh_ctx::K1Env &E = c->k1env;
    P.unroll = E.unroll;
    HIP_TRY(hipEventRecord(c->ev_pre0, c->stream));
    const double gkey[6] = {p.dRA, p.dDec, p.cos_t, p.sin_t, p.cos_i, p.sin_i};
    const bool known = c->range_valid && c->range_vis == vis_serial && c->range_mult_gen == mult_gen && c->range_first == p.first &&
                       c->range_count == count && memcmp(gkey, c->range_geom, sizeof gkey) == 0 && !c->no_range_cache &&
                       !E.no_range_cache;
    // qmax_all: over every row of the range whatever its multiplicity -- the sort is sized from it, because rows drawn
    // zero times are still sorted (with weight 0) and must land in a bucket of their own argument
    double qmax = 0.0, qmin = INFINITY, qmax_all = 0.0;
    if (known) {  // same rows, same geometry: the range is the one read back last time, no host round trip
        qmin = c->prepass_qmin;
        qmax = c->prepass_qmax;
        qmax_all = c->prepass_qmax_all;
    } else {  // one look at (u, v): 16 B per visibility and the one host round trip of the pass (64 KB of per-workgroup scalars)
        fh_prepass_geometry(0, c->num_cu, &P.wpb, &P.blocks);
        const int rblocks = P.blocks;
        HIP_TRY(fh_prepass_launch_range(P, c->stream));
        c->k1_scalars_host.resize((size_t)rblocks * 4);
        HIP_TRY(hipMemcpyAsync(c->k1_scalars_host.data(), c->partial_scalars.p, sizeof(double) * (size_t)rblocks * 4,
                               hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        for (int b = 0; b < rblocks; ++b) {
            const double mn = c->k1_scalars_host[(size_t)b * 4 + 1], m = c->k1_scalars_host[(size_t)b * 4 + 2],
                         ma = c->k1_scalars_host[(size_t)b * 4 + 3];
            if (m > qmax) qmax = m;  // (-inf for workgroups without rows; NaN baselines never win)
            if (mn < qmin) qmin = mn;
            if (ma > qmax_all) qmax_all = ma;
        }
    }
    if (!(qmax < INFINITY) || !(qmax_all < INFINITY)) return fail(FH_ERR_INVALID, "non-finite baseline in the visibility table");
    c->prepass_qmin = qmin;
    c->prepass_qmax = qmax;
    c->prepass_qmax_all = qmax_all;
    c->range_vis = vis_serial;
    c->range_mult_gen = mult_gen;
    c->range_first = p.first;
    c->range_count = count;
    memcpy(c->range_geom, gkey, sizeof gkey);
    c->range_valid = true;
    // statistical_models.py:166-169: the range check comes BEFORE the chunk loop -- nothing is binned for a table that fails it
    if (c->check_q_before_bin && c->dht->q[c->N - 1] < qmax)
        return fail(FH_ERR_QRANGE, "last collocation point %.3e < longest deprojected baseline %.3e", c->dht->q[c->N - 1], qmax);
    const double delta = c->k1_delta, inv_delta = 1.0 / delta;
    const double smax = qmax_all * p.inv_Qmax;
    if (smax * inv_delta > 2.0e9) return fail(FH_ERR_UNSUPPORTED, "baselines reach %.3g x Qmax", smax);
    const int nb = (int)(smax * inv_delta) + 2;  // one spare bucket: the device recomputes s * inv_delta itself
    if (nb > 16000)  // a wave of the sort keeps one counter per bucket in LDS
        return fail(FH_ERR_UNSUPPORTED, "baselines reach %.1f x Qmax (%d buckets of J0 arguments): cut the (u, v) distribution or "
                    "raise N", smax, nb);
    int rc = k1v2_ensure_table(c, nb);
    if (rc) return rc;
    int seg = E.seg;
    seg = seg < 128 ? 128 : ((seg + 127) & ~127);
    fh_prepass_geometry(nb, c->num_cu, &P.wpb, &P.blocks);
    if (const int w = E.wpb) {  // development: waves per workgroup / workgroups of P1, P2
        P.wpb = w;
        if (E.blocks > 0) P.blocks = E.blocks;
    }
    if (P.blocks > c->deproject_blocks) P.blocks = c->deproject_blocks;  // (entries of partial_scalars)
    // workspaces (grow on demand)
    const size_t nrows = (size_t)count + 16 * (size_t)nb + 16, max_pc = (size_t)fh_prepass_max_pieces(count, nb, seg);
    const size_t md = (size_t)fh_prepass_moment_doubles();
    if (c->k1_rows.n < nrows * 3 && c->k1_rows.alloc(nrows * 3 + 1024) != hipSuccess) return fail(FH_ERR_NOMEM, "hipMalloc (sorted rows) failed");
    P.hist_stride = (P.blocks + 255) & ~255;
    if (c->k1_hist.n < (size_t)P.hist_stride * nb) {
        c->hist_valid = false;
        if (c->k1_hist.alloc((size_t)P.hist_stride * nb + 1024) != hipSuccess) return fail(FH_ERR_NOMEM, "hipMalloc (histograms) failed");
        HIP_TRY(hipMemsetAsync(c->k1_hist.p, 0, sizeof(int) * c->k1_hist.n, c->stream));  // (the padding of the rows stays zero)
    }
    if (c->k1_totals.n < (size_t)nb && c->k1_totals.alloc((size_t)nb + 256) != hipSuccess) return fail(FH_ERR_NOMEM, "hipMalloc failed");
    if (c->k1_starts.n < (size_t)nb + 1 && c->k1_starts.alloc((size_t)nb + 257) != hipSuccess) return fail(FH_ERR_NOMEM, "hipMalloc failed");
    if (c->k1_cidx.n < (size_t)nb && c->k1_cidx.alloc((size_t)nb + 256) != hipSuccess) return fail(FH_ERR_NOMEM, "hipMalloc failed");
    if (c->k1_vbucket.n < (size_t)nb && c->k1_vbucket.alloc((size_t)nb + 256) != hipSuccess) return fail(FH_ERR_NOMEM, "hipMalloc failed");
    if (c->k1_vrows.n < (size_t)nb * 256 && c->k1_vrows.alloc((size_t)nb * 256 + 4096) != hipSuccess)
        return fail(FH_ERR_NOMEM, "hipMalloc (compressed rows) failed");
    if (c->k1_piece0.n < (size_t)nb + 1 && c->k1_piece0.alloc((size_t)nb + 257) != hipSuccess) return fail(FH_ERR_NOMEM, "hipMalloc failed");
    if (c->k1_partial.n < max_pc * md && c->k1_partial.alloc(max_pc * md + 4096) != hipSuccess)
        return fail(FH_ERR_NOMEM, "hipMalloc (moments) failed");
    P.inv_delta = inv_delta;
    P.delta = delta;
    P.nb = nb;
    P.seg_rows = seg;
    P.dummy_row = (int64_t)nrows;  // (the buffer holds nrows + 341 rows)
    // |u|, |v| <= q / cos(inc), so |phase| <= (|dRA| + |dDec|) qmax / |cos(inc)|: beyond 1e5 rad the library's sincos
    P.safe_trig = !((fabs(p.dRA) + fabs(p.dDec)) * qmax_all < 1.0e5 * fabs(p.cos_i)) || E.safe_trig;
    P.hist = c->k1_hist.p;
    P.totals = c->k1_totals.p;
    P.starts = c->k1_starts.p;
    P.cidx = c->k1_cidx.p;
    P.info = c->k1_info.p;
    P.piece0 = c->k1_piece0.p;
    P.rows = c->k1_rows.p;
    P.partial = c->k1_partial.p;
    P.vrows = c->k1_vrows.p;
    P.vbucket = c->k1_vbucket.p;
    // the same rows under the same geometry as the last pass of this context (bootstrap-free pipelines, sweeps that re-bin, the
    // bench's steps): the histograms, their scan and the table layout are still in place -- P1 and the scan are skipped
    const bool reuse = known && c->hist_valid && c->hist_nb == nb && c->hist_blocks == P.blocks && c->hist_wpb == P.wpb &&
                       c->hist_unroll == P.unroll && c->hist_seg == seg && !E.no_hist_cache;
    c->hist_valid = false;
    HIP_TRY(fh_prepass_launch(P, c->stream, reuse ? 1 : 0));
    c->hist_valid = true;
    c->hist_nb = nb;
    c->hist_blocks = P.blocks;
    c->hist_wpb = P.wpb;
    c->hist_unroll = P.unroll;
    c->hist_seg = seg;

    // the Gram of the virtual rows: one 16-row chunk per non-empty bucket (a few hundred to a few thousand chunks); one
    // workgroup (or a few) per output tile, no slabs (vr_gram_kernel).  FRANK_AMD_K1_VR=slabs keeps bin_gram2_kernel<.., VR>.
    if (!(E.vr_slabs && c->rows_ok)) {  // (bin_gram2_kernel's tile maps stop at N = 511)
        VrGramParams G{};
        G.N = c->N;
        G.NBT = c->NBT;
        G.XS = c->XS;
        G.ntiles = c->ntiles;
        // eight workgroups per tile -- workgroup ids go round the eight XCDs, so an XCD's L2 holds one eighth of the tables
        // (7 MB at N = 300: read once per workgroup they came from memory, 42 us) -- of eight waves each
        G.waves = E.vrwaves;
        G.waves = G.waves < 4 ? 4 : (G.waves > 16 ? 16 : G.waves);  // (the tile is folded by the workgroup's first 256 threads)
        int split = E.vrsplit;
        G.split = split < 1 ? 1 : (split > 8 ? 8 : split);
        G.vrows = c->k1_vrows.p;
        G.vbucket = c->k1_vbucket.p;
        G.info = c->k1_info.p + 1;
        G.table = c->k1_table.p;
        G.scratch = c->reduce_scratch.p;
        G.partial_scalars = c->partial_scalars.p;
        G.scalar_blocks = P.blocks;
        G.fresh = c->stats_reset_pending ? 1 : 0;  // (the sums start here: vr_finish_kernel stores them, the fills of fh_bin_reset never run)
        c->stats_reset_pending = false;
        HIP_TRY(hipEventRecord(c->ev_bin0, c->stream));
        HIP_TRY(fh_vr_gram_launch(G, c->stats_sum.p, c->stats_minmax.p, c->stream));
        HIP_TRY(hipEventRecord(c->ev_bin1, c->stream));
        c->bin_timed = true;
        c->have_device_Mj = false;
        return FH_OK;
    }
    // the Gram of the virtual rows: one 16-row chunk per non-empty bucket (a few hundred to a few thousand chunks), so a
    // few dozen workgroups -- every workgroup writes a slab of all its tiles that the reduction reads back
    rc = ensure_slabs(c);
    if (rc) return rc;
    rc = settle_reset(c);
    if (rc) return rc;
    Bin2Params bp{};
    bp.N = c->N;
    bp.table = c->k1_table.p;
    bp.virtual_rows = 1;
    bp.rows = c->k1_vrows.p;
    bp.chunk_bucket = c->k1_vbucket.p;
    bp.info = c->k1_info.p + 1;
    bp.work_counter = nullptr;  // static hand-out: the same sums in every run
    ReduceParams rp{};
    rp.nparts = c->nparts;
    rp.ntiles = c->ntiles;
    int vr_blocks = E.vrblocks > 0 ? E.vrblocks : (nb < 512 ? 32 : 64);
    int G = 0;
    for (int Pt = 0; Pt < c->nparts; ++Pt) G += c->part_blocks[Pt];
    if (vr_blocks > G) vr_blocks = G;
    for (int Pt = 0; Pt < 3; ++Pt) {
        int blocks = 0;
        if (Pt < c->nparts) {
            blocks = (int)(((long long)c->part_blocks[Pt] * vr_blocks + G - 1) / G);
            if (blocks < 1) blocks = 1;
            if (blocks > c->part_blocks[Pt]) blocks = c->part_blocks[Pt];
        }
        bp.part_blocks[Pt] = blocks;
        bp.partials[Pt] = c->partials[Pt].p;
        rp.part_blocks[Pt] = blocks;
        rp.part_tile0[Pt] = Pt < c->nparts ? fh_k1v2_part_tile0(c->NBT, Pt) : 0;
        rp.part_ntiles[Pt] = Pt < c->nparts ? fh_k1v2_part_ntiles(c->NBT, Pt) : 0;
        rp.partials[Pt] = c->partials[Pt].p;
    }
    rp.partial_scalars = c->partial_scalars.p;
    rp.scratch = c->reduce_scratch.p;
    rp.scalar_blocks = P.blocks;
    HIP_TRY(hipEventRecord(c->ev_bin0, c->stream));
    HIP_TRY(fh_k1v2_launch_bin(c->NBT, bp, c->stream));
    HIP_TRY(hipEventRecord(c->ev_bin1, c->stream));
    c->bin_timed = true;
    HIP_TRY(fh_k1_launch_reduce(rp, c->stats_sum.p, c->stats_minmax.p, c->stream));
    c->have_device_Mj = false;
    return FH_OK;
}

// columns and row range of a resident table, as every kernel that streams it takes them
static void table_columns(BinParams &p, const fh_vis *vis, int64_t first, int64_t count) {
    p.u = vis->u.p;
    p.v = vis->v.p;
    p.Vre = vis->Vre.p;
    p.Vim = vis->has_im ? vis->Vim.p : nullptr;
    p.w = vis->w.p;
    if (vis->f32) {
        p.u32 = vis->u32.p;
        p.v32 = vis->v32.p;
        p.Vre32 = vis->Vre32.p;
        p.Vim32 = vis->has_im ? vis->Vim32.p : nullptr;
        p.w32 = vis->w32.p;
    }
    p.w_scalar = vis->w_scalar;
    p.mult = vis->use_mult ? vis->mult.p : nullptr;
    p.first = first;
    p.count = count;
}

int fh_bin_visibilities(fh_ctx *c, const fh_geometry *g, const fh_vis *vis, int64_t first, int64_t count) {
    if (!c || !g || !vis) return fail(FH_ERR_INVALID, "fh_bin_visibilities: NULL argument");
    if (first < 0 || count < 0 || first + count > vis->n) return fail(FH_ERR_INVALID, "fh_bin_visibilities: bad range");
    if (vis->device != c->device) return fail(FH_ERR_INVALID, "visibility table lives on another device");
    // single-precision arithmetic of the design block (fh_ctx_set_arithmetic): its Gram is off by ~1e-8 of the largest entry and,
    // measured, no longer positive definite at 1e7 rows (the first seed solve has unit prior precision, radial_fitters.py:744);
    // it is also 25 x slower than the fp64 moments pass.  Kept for tables up to 2e6 rows, refused beyond: hand the table over
    // in single precision instead (fh_vis_upload_f32 -- 20 B per visibility, fp64 arithmetic).
    if (c->arith32 && count > 2000000)
        return fail(FH_ERR_UNSUPPORTED, "arithmetic='fp32' covers tables up to 2e6 visibilities (%lld given): beyond, the "
                    "single-precision Gram loses positive definiteness; pass float32 arrays (fp32 storage, fp64 arithmetic) "
                    "or use the default arithmetic", (long long)count);
    HIP_TRY(hipSetDevice(c->device));
    BinParams p{};
    table_columns(p, vis, first, count);
    // geometry.py:69-70 (dRA *= 2 pi / rad_to_arcsec), :111-115
    p.dRA = g->dRA_arcsec * (2. * M_PI / kRadToArcsec);
    p.dDec = g->dDec_arcsec * (2. * M_PI / kRadToArcsec);
    const double inc = g->inc_deg * kDegToRad, PA = g->PA_deg * kDegToRad;
    p.cos_t = cos(PA);
    p.sin_t = sin(PA);
    p.cos_i = cos(inc);
    p.sin_i = sin(inc);
    p.N = c->N;
    p.inv_Qmax = 1. / c->dht->Qmax;
    p.zeros = c->zeros.p;
    p.j0_table = c->j0_table.p;
    const int64_t nsuper = (count + fh_k1_super() - 1) / fh_k1_super();
    if (nsuper > 0x7fffffff / 2) return fail(FH_ERR_UNSUPPORTED, "more than 2^39 visibilities in one call");
    p.H2 = c->debris ? c->debris_H2.p : nullptr;
    if (c->arith32 && !c->rows_ok)
        return fail(FH_ERR_UNSUPPORTED, "arithmetic='fp32' exists for N <= 511 (N = %d)", c->N);
    if (c->v2 && !use_wide(c)) return bin_visibilities_v2(c, p, count, vis->serial, vis->use_mult ? vis->mult_gen : 0);
    c->hist_valid = false;  // (the paths below write the per-workgroup scalars the moments path keeps between passes)
    {
        const int rcs = settle_reset(c);
        if (rcs) return rcs;
    }
    if (use_wide(c)) {
        const int rcw = ensure_wide(c);
        if (rcw) return rcw;
    }
    // K1a scratch: 24 B per visibility (32 B with the debris model's kz^2)
    const size_t cnt1 = (size_t)(count > 0 ? count : 1);
    const size_t need = cnt1 * (c->debris ? 4 : 3);
    if (c->prep.n < need) HIP_TRY(c->prep.alloc(need));
    p.prep_s = c->prep.p;
    p.prep_sw = c->prep.p + cnt1;
    p.prep_swV = c->prep.p + 2 * cnt1;
    p.prep_k2 = c->debris ? c->prep.p + 3 * cnt1 : nullptr;
    if (use_wide(c)) {
        // N > 303 / debris: sqrt(w)-scaled rows to memory, chunk by chunk, and G += X^T X by rocBLAS (fp64 MFMA inside)
        double *G = dense_gram(c);
        int dblocks = (int)((count + 255) / 256);
        if (dblocks > c->deproject_blocks) dblocks = c->deproject_blocks;
        if (dblocks < 1) dblocks = 1;
        p.partial_scalars = c->partial_scalars.p;
        HIP_TRY(fh_k1_launch_deproject(p, dblocks, c->stream));
        HIP_TRY(hipEventRecord(c->ev_bin0, c->stream));
        const int N1 = c->N + 1;
        const double one = 1.0;
        for (int64_t r0 = 0; r0 < count; r0 += c->wide_rows) {
            const int64_t rows = count - r0 < c->wide_rows ? count - r0 : c->wide_rows;
            HIP_TRY(fh_k1_launch_wide_rows(p, r0, rows, c->wide_X.p, c->stream));
            if (getenv("FRANK_AMD_WIDE_SYRK")) {
                ROC_TRY(rocblas_dsyrk(c->blas, rocblas_fill_upper, rocblas_operation_none, N1, (rocblas_int)rows, &one,
                                      c->wide_X.p, N1, &one, G, N1));
            } else {
                // the full product: rocBLAS's dsyrk is ~2 orders of magnitude slower than its dgemm for this shape
                // (n = N + 1 small, k = 65536); only the upper triangle of G is read afterwards
                ROC_TRY(rocblas_dgemm(c->blas, rocblas_operation_none, rocblas_operation_transpose, N1, N1, (rocblas_int)rows,
                                      &one, c->wide_X.p, N1, c->wide_X.p, N1, &one, G, N1));
            }
        }
        HIP_TRY(hipEventRecord(c->ev_bin1, c->stream));
        c->bin_timed = true;
        HIP_TRY(fh_k1_launch_wide_scalars(c->partial_scalars.p, dblocks, G + dense_tail(c), c->stats_minmax.p, c->stream));
        c->have_device_Mj = false;
        return FH_OK;
    }
    // fit_loop kernels of earlier fits that are still RUNNING each hold a CU (a slot stays "busy" until it is collected,
    // long after its kernel has finished: counting those would leave CUs idle)
    int running = 0;
    if (c->slots_busy > 0)
        for (auto &b : c->batches)
            if (b.active && b.launched && hipEventQuery(b.done) == hipErrorNotReady) running += b.n;
    (void)hipGetLastError();  // hipErrorNotReady is not an error here
    // throughput mode while such kernels hold CUs (see bin_gram.hip)
    // (also while fits of a pipeline are merely outstanding: the dynamic hand-out is 2 % faster even on an empty GPU,
    // 26.7 vs 27.2 ms, and a pipeline's sums are run-dependent in their last bits anyway; synchronous fits stay static)
    const bool dynamic = (running > 0 || c->slots_busy > 0 || c->k1env.dynamic) && !c->force_static;
    p.work_counter = dynamic ? c->work_counter.p : nullptr;
    if (dynamic) HIP_TRY(hipMemsetAsync(c->work_counter.p, 0, 2 * sizeof(int), c->stream));
    ReduceParams rp{};
    rp.nparts = c->nparts;
    rp.ntiles = c->ntiles;
    // leave one CU per outstanding fit_loop kernel (each occupies a whole CU) so every bin_gram workgroup is resident
    int reserve = running;
    if (c->k1env.reserve_mult >= 0.0) reserve = (int)(running * c->k1env.reserve_mult);  // development switch
    if (reserve > c->num_cu / 4) reserve = c->num_cu / 4;
    const int G = c->part_blocks[0] + c->part_blocks[1];
    for (int P = 0; P < 2; ++P) {
        int blocks = c->part_blocks[P];
        if (reserve > 0 && G > 0) blocks -= (reserve * c->part_blocks[P] + G - 1) / G;
        if (blocks < 1 && P < c->nparts) blocks = 1;
        if (P < c->nparts && nsuper < blocks) blocks = (int)(nsuper > 0 ? nsuper : 1);
        p.part_blocks[P] = P < c->nparts ? blocks : 0;
        p.partials[P] = c->partials[P].p;
        rp.part_blocks[P] = p.part_blocks[P];
        rp.part_tile0[P] = P < c->nparts ? fh_k1_part_tile0(c->NBT, P) : 0;
        rp.part_ntiles[P] = P < c->nparts ? fh_k1_part_ntiles(c->NBT, P) : 0;
        rp.partials[P] = c->partials[P].p;
    }
    p.partial_scalars = c->partial_scalars.p;
    rp.partial_scalars = c->partial_scalars.p;
    rp.scratch = c->reduce_scratch.p;
    int dblocks = (int)((count + 255) / 256);
    if (dblocks > c->deproject_blocks) dblocks = c->deproject_blocks;
    if (dblocks < 1) dblocks = 1;
    rp.scalar_blocks = dblocks;
    HIP_TRY(fh_k1_launch_deproject(p, dblocks, c->stream));
    HIP_TRY(hipEventRecord(c->ev_bin0, c->stream));
    HIP_TRY(fh_k1_launch_bin(c->NBT, p, c->stream));
    HIP_TRY(hipEventRecord(c->ev_bin1, c->stream));
    c->bin_timed = true;
    HIP_TRY(fh_k1_launch_reduce(rp, c->stats_sum.p, c->stats_minmax.p, c->stream));
    c->have_device_Mj = false;
    return FH_OK;
}

int fh_bin_last_kernel_ms(fh_ctx *c, float *ms) {
    if (!c || !ms) return fail(FH_ERR_INVALID, "fh_bin_last_kernel_ms: NULL argument");
    if (!c->bin_timed) return fail(FH_ERR_INVALID, "no bin_gram launch recorded yet");
    HIP_TRY(hipEventSynchronize(c->ev_bin1));
    HIP_TRY(hipEventElapsedTime(ms, c->ev_bin0, c->ev_bin1));
    return FH_OK;
}

int fh_bin_last_prepass_ms(fh_ctx *c, float *ms) {
    if (!c || !ms) return fail(FH_ERR_INVALID, "fh_bin_last_prepass_ms: NULL argument");
    if (!c->bin_timed || !c->v2) return fail(FH_ERR_INVALID, "no bin_gram (v2) launch recorded yet");
    HIP_TRY(hipEventSynchronize(c->ev_bin0));
    HIP_TRY(hipEventElapsedTime(ms, c->ev_pre0, c->ev_bin0));
    return FH_OK;
}

int fh_fit_last_kernel_ms(fh_ctx *c, float *ms) {
    if (!c || !ms) return fail(FH_ERR_INVALID, "fh_fit_last_kernel_ms: NULL argument");
    if (!c->loop_timed) return fail(FH_ERR_INVALID, "no fit_loop launch of fh_fit_normal recorded yet");
    HIP_TRY(hipEventSynchronize(c->ev_loop1));
    HIP_TRY(hipEventElapsedTime(ms, c->ev_loop0, c->ev_loop1));
    return FH_OK;
}

int fh_ctx_set_arithmetic(fh_ctx *c, int fp32) {
    if (!c) return fail(FH_ERR_INVALID, "ctx is NULL");
    if (fp32 && !c->v2) return fail(FH_ERR_UNSUPPORTED, "single-precision binning exists for the fused kernel only (N <= 383)");
    c->arith32 = fp32 != 0;
    return FH_OK;
}

int fh_ctx_set_lognormal_linesearch(fh_ctx *c, int reference_products) {
    if (!c) return fail(FH_ERR_INVALID, "ctx is NULL");
    c->ln_fresh_products = reference_products != 0;
    return FH_OK;
}

int fh_ctx_set_reproducible(fh_ctx *c, int on) {
    if (!c) return fail(FH_ERR_INVALID, "ctx is NULL");
    c->force_static = on != 0;
    return FH_OK;
}

// Compute-unit partition of a pipeline of fits.  A fit loop holds one compute unit for the ~0.1 s of its iteration; the binning
// passes of the following fits would otherwise put their workgroups on the same units (a fit loop leaves registers and LDS
// free) and take instruction slots and L1 lines from it.  bin_cus > 0: the context's stream -- every kernel of the binning
// pass -- is confined to the first bin_cus units of the mask (the bits go round the eight XCDs, so every XCD gives the same
// share), the streams of the fit loops to the rest.  Call before the first fh_fit_submit of the context.
int fh_ctx_set_cu_partition(fh_ctx *c, int bin_cus) {
    if (!c) return fail(FH_ERR_INVALID, "ctx is NULL");
    if (c->slot_pool.p) return fail(FH_ERR_INVALID, "fh_ctx_set_cu_partition: the fit slots of this context exist already");
    if (bin_cus < 8 || bin_cus > c->num_cu - 8) return fail(FH_ERR_INVALID, "fh_ctx_set_cu_partition: %d of %d compute units", bin_cus, c->num_cu);
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    uint32_t mask[8];
    cu_mask(0, bin_cus, mask);
    hipStream_t st = nullptr;
    HIP_TRY(hipExtStreamCreateWithCUMask(&st, 8, mask));
    ROC_TRY(rocblas_set_stream(c->blas, st));
    (void)hipStreamDestroy(c->stream);
    c->stream = st;
    c->bin_cus = bin_cus;
    return FH_OK;
}

int fh_ctx_set_range_cache(fh_ctx *c, int on) {
    if (!c) return fail(FH_ERR_INVALID, "ctx is NULL");
    c->no_range_cache = on == 0;
    c->range_valid = false;
    return FH_OK;
}

int fh_stats_device(fh_ctx *c, double **sum_stats, int64_t *n_sum, double **minmax_stats) {
    if (!c || !c->stats_sum.p) return fail(FH_ERR_INVALID, "fh_stats_device: no binning workspace");
    {
        const int rcs = settle_reset(c);
        if (rcs) return rcs;
    }
    if (sum_stats) *sum_stats = use_wide(c) ? dense_gram(c) : c->stats_sum.p;
    if (n_sum) *n_sum = use_wide(c) ? (int64_t)dense_tail(c) + 2 : (int64_t)c->stats_sum.n;
    if (minmax_stats) *minmax_stats = c->stats_minmax.p;
    return FH_OK;
}

int fh_stats_finalize(fh_ctx *c, const fh_geometry *g, int vis_model, int check_qbounds, double *M, double *j,
                      double *H0, double *qmin, double *qmax) {
    if (!c || !g) return fail(FH_ERR_INVALID, "fh_stats_finalize: NULL argument");
    if (vis_model != FH_VIS_OPT_THICK && vis_model != FH_VIS_OPT_THIN && vis_model != FH_VIS_DEBRIS)
        return fail(FH_ERR_INVALID, "vis_model must be one of ['opt_thick', 'opt_thin', 'debris']");
    if ((vis_model == FH_VIS_DEBRIS) != c->debris)
        return fail(FH_ERR_INVALID, "vis_model 'debris' goes with fh_ctx_set_scale_height (and only with it)");
    HIP_TRY(hipSetDevice(c->device));
    {
        const int rcs = settle_reset(c);  // (a reset that no binning pass followed)
        if (rcs) return rcs;
    }
    const int N = c->N;
    // a_k = ((norm * sf_k)) * scale : hankel.py:201 and statistical_models.py:490,507
    const double scale = vis_model == FH_VIS_OPT_THICK ? cos(g->inc_deg * kDegToRad) : 1.0;
    const double norm = 1 / (M_PI * c->dht->Qmax * c->dht->Qmax);
    if (!(c->a_scale_valid && c->a_scale_value == scale)) {  // (the vector on the device depends on `scale` only)
        std::vector<double> &a = c->a_host;
        HIP_TRY(hipStreamSynchronize(c->stream));  // an earlier asynchronous copy may still read the host vector
        a.resize(N);
        for (int k = 0; k < N; ++k) a[k] = (norm * c->dht->scale_factor[k]) * scale;
        HIP_TRY(hipMemcpyAsync(c->a_scale.p, a.data(), sizeof(double) * N, hipMemcpyHostToDevice, c->stream));
        c->a_scale_valid = true;
        c->a_scale_value = scale;
    }
    if (use_wide(c))
        HIP_TRY(fh_k1_launch_wide_finalize(dense_gram(c), N, c->a_scale.p, c->M.p, c->j.p, c->sumwV2.p, c->stream));
    else
        HIP_TRY(fh_k1_launch_finalize(c->stats_sum.p, c->NBT, N, c->a_scale.p, c->M.p, c->j.p, c->sumwV2.p, c->stream));
    if (!M && !j && !H0 && !qmin && !qmax && !check_qbounds) {  // nothing asked for on the host: M, j stay on the device, no wait
        c->have_device_Mj = true;
        return FH_OK;
    }
    double tail[2], mm[2], swv2;
    const double *tail_src = use_wide(c) ? dense_gram(c) + dense_tail(c) : c->stats_sum.p + c->tail_offset;
    HIP_TRY(hipMemcpyAsync(tail, tail_src, sizeof tail, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(mm, c->stats_minmax.p, sizeof mm, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(&swv2, c->sumwV2.p, sizeof swv2, hipMemcpyDeviceToHost, c->stream));
    if (M) HIP_TRY(hipMemcpyAsync(M, c->M.p, sizeof(double) * (size_t)N * N, hipMemcpyDeviceToHost, c->stream));
    if (j) HIP_TRY(hipMemcpyAsync(j, c->j.p, sizeof(double) * N, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->have_device_Mj = true;
    if (H0) *H0 = 0.5 * (tail[0] - swv2);  // statistical_models.py:218
    if (mm[0] != mm[0]) mm[0] = -INFINITY;  // nothing binned since the reset (the reset leaves the NaN neutral element)
    if (mm[1] != mm[1]) mm[1] = -INFINITY;
    const double qmn = -mm[0], qmx = mm[1];
    if (qmin) *qmin = qmn;
    if (qmax) *qmax = qmx;
    if (check_qbounds && c->dht->q[N - 1] < qmx)  // statistical_models.py:526
        return fail(FH_ERR_QRANGE, "last collocation point %.3e < longest deprojected baseline %.3e", c->dht->q[N - 1], qmx);
    return FH_OK;
}

int fh_map_visibilities(fh_ctx *c, const fh_geometry *g, int vis_model, int check_qbounds, const double *u,
                        const double *v, const double *Vre, const double *Vim, const double *w, int64_t n_w, int64_t n,
                        double *M, double *j, double *H0, double *qmin, double *qmax) {
    if (!c) return fail(FH_ERR_INVALID, "ctx is NULL");
    fh_vis *vis = nullptr;
    // Vre == NULL: Vim holds the visibilities as n (re, im) pairs -- a NumPy complex128 array as it is (fh_map_visibilities_c128)
    int rc = Vre ? fh_vis_upload(c->device, u, v, Vre, Vim, w, n_w, n, &vis) : fh_vis_upload_c128(c->device, u, v, Vim, w, n_w, n, &vis);
    if (rc) return rc;
    rc = fh_bin_reset(c);
    c->check_q_before_bin = check_qbounds != 0;
    if (!rc) rc = fh_bin_visibilities(c, g, vis, 0, n);
    c->check_q_before_bin = false;
    if (!rc) {
        rc = fh_stats_finalize(c, g, vis_model, check_qbounds, M, j, H0, qmin, qmax);
    } else {
        (void)hipStreamSynchronize(c->stream);
        if (rc == FH_ERR_QRANGE) {  // stopped before the binning: the range is what the caller's message needs
            if (qmin) *qmin = c->prepass_qmin;
            if (qmax) *qmax = c->prepass_qmax;
        }
    }
    fh_vis_destroy(vis);
    return rc;
}

int fh_map_visibilities_c128(fh_ctx *c, const fh_geometry *g, int vis_model, int check_qbounds, const double *u, const double *v,
                             const double *Vc, const double *w, int64_t n_w, int64_t n, double *M, double *j, double *H0,
                             double *qmin, double *qmax) {
    if (!Vc) return fail(FH_ERR_INVALID, "fh_map_visibilities_c128: V is NULL");
    return fh_map_visibilities(c, g, vis_model, check_qbounds, u, v, nullptr, Vc, w, n_w, n, M, j, H0, qmin, qmax);
}

// ---- K2 -----------------------------------------------------------------------------------------------------------
static FitState make_state(fh_ctx *c) {
    FitState st{};
    st.N = c->N;
    st.Y = c->Y.p;
    st.Ykm = c->Ykm.p;
    st.q = c->q.p;
    st.M = c->M.p;
    st.j = c->j.p;
    st.band_lu = c->band_lu.p;
    st.W = c->W.p;
    st.D = c->D.p;
    st.Z = c->Z.p;
    st.p = c->p.p;
    st.p_old = c->p_old.p;
    st.mu = c->mu.p;
    st.flags = c->flags.p;
    st.info = c->info.p;
    st.transform_norm = (2 * M_PI * c->dht->Rmax * c->dht->Rmax) / c->dht->j_nN;  // hankel.py:155
    return st;
}

// D = M + W^T Y (upper triangle in row-major terms is what potrf reads), factor, solve for mu.
// Row-major buffers are column-major transposes: C_cm = D^T = Y^T W + M^T -> dgemm(N, T) on (Y_rm, W_rm) gives
// C_cm[i + k*N] = sum_j Y_rm[j*N+i] ... we want D[i][k] = sum_j W[j][i] Y[j][k]; as column-major (ld N):
// A_cm = W_rm viewed (N x N, A_cm[i + j*N] = W[j][i]) and B_cm = Y_rm (B_cm[k + j*N] = Y[j][k]) ->
// D_cm[k + i*N]  (= row-major D[i][k]) = sum_j B_cm[k + j*N] * A_cm[i + j*N] = (B * A^T)[k][i].
static int solve_posterior(fh_ctx *c, const FitState &st, bool with_prior, bool want_tr2) {
    const int N = c->N;
    const double one = 1.0;
    c->have_device_mu = true;  // (whatever the factorisation says: the callers replace a failed solve by the SVD route's)
    if (with_prior) {
        HIP_TRY(fh_k2_launch_prep(st, c->stream));
        ROC_TRY(rocblas_dgemm(c->blas, rocblas_operation_none, rocblas_operation_transpose, N, N, N, &one, c->Y.p, N,
                              c->W.p, N, &one, c->D.p, N));
    } else {
        HIP_TRY(hipMemcpyAsync(c->D.p, c->M.p, sizeof(double) * (size_t)N * N, hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(c->mu.p, c->j.p, sizeof(double) * N, hipMemcpyDeviceToDevice, c->stream));
    }
    // scipy.linalg.cho_factor(Dinv) = LAPACK dpotrf('U') on the row-major array == 'L' on its column-major view
    ROC_TRY(rocsolver_dpotrf(c->blas, rocblas_fill_lower, N, c->D.p, N, c->info.p));
    ROC_TRY(rocsolver_dpotrs(c->blas, rocblas_fill_lower, N, 1, c->D.p, N, c->mu.p, N));
    if (want_tr2)  // Z_cm <- L^-1 Z_cm with Z_cm = Y^T  (the buffer holds row-major Y)
        ROC_TRY(rocblas_dtrsm(c->blas, rocblas_side_left, rocblas_fill_lower, rocblas_operation_none,
                              rocblas_diagonal_non_unit, N, N, &one, c->D.p, N, c->Z.p, N));
    return FH_OK;
}

// SVD pseudo-inverse solve on the device, the reference's route when cho_factor raises (statistical_models.py:747-755,
// 1150-1158):  U, s, V = svd(A);  X = V^T diag(where(s > 0, 1/s, 0)) U^T B.   A_dev: N*N row-major (destroyed),
// B_dev: N*nrhs row-major, overwritten with X.  rocSOLVER factorises the column-major view A^T = U' S Vt', so
// pinv(A) = U' S^+ Vt' and, on the column-major view of B (nrhs x N), X^T = B^T Vt'^T S^+ U'^T.
static int svd_pinv_solve_device(fh_ctx *c, double *A_dev, double *B_dev, int nrhs, bool last_axis_scaling = false) {
    const int N = c->N;
    const size_t NN = (size_t)N * N;
    DevBuf<double> S, U, Vt, E, T;
    if (S.alloc(2 * (size_t)N) != hipSuccess || U.alloc(NN) != hipSuccess || Vt.alloc(NN) != hipSuccess ||
        E.alloc((size_t)N) != hipSuccess || T.alloc((size_t)N * nrhs) != hipSuccess)
        return fail(FH_ERR_NOMEM, "svd_pinv_solve: device allocation failed");
    ROC_TRY(rocsolver_dgesvd(c->blas, rocblas_svect_all, rocblas_svect_all, N, N, A_dev, N, S.p, U.p, N, Vt.p, N, E.p,
                             rocblas_outofplace, c->info.p));
    int info = 0;
    HIP_TRY(hipMemcpyAsync(&info, c->info.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (info != 0) return fail(FH_ERR_NOT_SPD, "SVD did not converge (info %d)", info);
    HIP_TRY(fh_k2_launch_pinv_scale(S.p, N, S.p + N, c->stream));
    const double one = 1.0, zero = 0.0;
    ROC_TRY(rocblas_dgemm(c->blas, rocblas_operation_none, rocblas_operation_transpose, nrhs, N, N, &one, B_dev, nrhs, Vt.p,
                          N, &zero, T.p, nrhs));
    // T (nrhs x N column-major) = (U^T B)^T.  The pseudo-inverse scales singular direction i (a column here) by s1[i];
    // the reference's `(U^T b) * s1` broadcasts s1 over the LAST axis of an N x N right-hand side instead, i.e. scales
    // right-hand side c (a row here) by s1[c] (fh_svd_solve_as_reference)
    if (last_axis_scaling && nrhs == N) ROC_TRY(rocblas_ddgmm(c->blas, rocblas_side_left, nrhs, N, T.p, nrhs, S.p + N, 1, T.p, nrhs));
    else ROC_TRY(rocblas_ddgmm(c->blas, rocblas_side_right, nrhs, N, T.p, nrhs, S.p + N, 1, T.p, nrhs));
    ROC_TRY(rocblas_dgemm(c->blas, rocblas_operation_none, rocblas_operation_transpose, nrhs, N, N, &one, T.p, nrhs, U.p, N,
                          &zero, B_dev, nrhs));
    return FH_OK;
}


int fh_gaussian_model(fh_ctx *c, const double *M, const double *j, const double *p, double *mu, double *chol,
                      double *Sinv, int *used_svd) {
    if (!c || ((M == nullptr) != (j == nullptr))) return fail(FH_ERR_INVALID, "fh_gaussian_model: pass both M and j or neither");
    if (!M && !c->have_device_Mj) return fail(FH_ERR_INVALID, "fh_gaussian_model: no device-resident M, j (run fh_stats_finalize)");
    SyncOnExit drain{c->stream};
    HIP_TRY(hipSetDevice(c->device));
    const int N = c->N;
    const size_t NN = (size_t)N * N;
    if (used_svd) *used_svd = 0;
    if (p)
        for (int k = 0; k < N; ++k)
            if (!(p[k] > 0.0)) return fail(FH_ERR_BAD_P, "Bad value in power spectrum (p[%d] = %g)", k, p[k]);
    if (M) {  // (else: the statistics fh_stats_finalize left on the device)
        HIP_TRY(hipMemcpyAsync(c->M.p, M, sizeof(double) * NN, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(c->j.p, j, sizeof(double) * N, hipMemcpyHostToDevice, c->stream));
        c->have_device_Mj = false;
    }
    if (p) HIP_TRY(hipMemcpyAsync(c->p.p, p, sizeof(double) * N, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemsetAsync(c->flags.p, 0, sizeof(int) * FIT_NFLAGS, c->stream));
    FitState st = make_state(c);
    if (Sinv) {
        if (p) {
            const double one = 1.0, zero = 0.0;
            HIP_TRY(fh_k2_launch_prep(st, c->stream));
            ROC_TRY(rocblas_dgemm(c->blas, rocblas_operation_none, rocblas_operation_transpose, N, N, N, &one, c->Y.p,
                                  N, c->W.p, N, &zero, c->Z.p, N));
            HIP_TRY(hipMemcpyAsync(Sinv, c->Z.p, sizeof(double) * NN, hipMemcpyDeviceToHost, c->stream));
        } else {
            memset(Sinv, 0, sizeof(double) * NN);
        }
    }
    int rc = solve_posterior(c, st, p != nullptr, false);
    if (rc) return rc;
    int info = 0;
    HIP_TRY(hipMemcpyAsync(&info, c->info.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    if (mu) HIP_TRY(hipMemcpyAsync(mu, c->mu.p, sizeof(double) * N, hipMemcpyDeviceToHost, c->stream));
    if (chol) HIP_TRY(hipMemcpyAsync(chol, c->D.p, sizeof(double) * NN, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (info != 0) {
        // not positive definite: rebuild Dinv and take the SVD route, as the reference does (rocSOLVER gesvd)
        if (p) {
            const double one = 1.0;
            HIP_TRY(fh_k2_launch_prep(st, c->stream));
            ROC_TRY(rocblas_dgemm(c->blas, rocblas_operation_none, rocblas_operation_transpose, N, N, N, &one, c->Y.p,
                                  N, c->W.p, N, &one, c->D.p, N));
        } else {
            HIP_TRY(hipMemcpyAsync(c->D.p, c->M.p, sizeof(double) * NN, hipMemcpyDeviceToDevice, c->stream));
        }
        HIP_TRY(hipMemcpyAsync(c->mu.p, c->j.p, sizeof(double) * N, hipMemcpyDeviceToDevice, c->stream));
        rc = svd_pinv_solve_device(c, c->D.p, c->mu.p, 1);
        if (rc) return rc;
        if (mu) HIP_TRY(hipMemcpyAsync(mu, c->mu.p, sizeof(double) * N, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (used_svd) *used_svd = 1;
    }
    return FH_OK;
}

int fh_cho_solve(fh_ctx *c, const double *chol, double *B, int nrhs) {
    if (!c || !chol || !B || nrhs < 1) return fail(FH_ERR_INVALID, "fh_cho_solve: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    const int N = c->N;
    const size_t NN = (size_t)N * N, nb = (size_t)N * nrhs;
    if (c->scratch_out.n < nb) HIP_TRY(c->scratch_out.alloc(nb));
    HIP_TRY(hipMemcpyAsync(c->D.p, chol, sizeof(double) * NN, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->scratch_out.p, B, sizeof(double) * nb, hipMemcpyHostToDevice, c->stream));
    // row-major B (N x nrhs) is the column-major (nrhs x N) matrix B^T:  X^T (L L^T) = B^T
    const double one = 1.0;
    ROC_TRY(rocblas_dtrsm(c->blas, rocblas_side_right, rocblas_fill_lower, rocblas_operation_transpose,
                          rocblas_diagonal_non_unit, nrhs, N, &one, c->D.p, N, c->scratch_out.p, nrhs));
    ROC_TRY(rocblas_dtrsm(c->blas, rocblas_side_right, rocblas_fill_lower, rocblas_operation_none,
                          rocblas_diagonal_non_unit, nrhs, N, &one, c->D.p, N, c->scratch_out.p, nrhs));
    HIP_TRY(hipMemcpyAsync(B, c->scratch_out.p, sizeof(double) * nb, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return FH_OK;
}

int fh_svd_solve(fh_ctx *c, const double *A, double *B, int nrhs) {
    if (!c || !A || !B || nrhs < 1) return fail(FH_ERR_INVALID, "fh_svd_solve: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    const int N = c->N;
    const size_t NN = (size_t)N * N, nb = (size_t)N * nrhs;
    if (c->scratch_out.n < nb) HIP_TRY(c->scratch_out.alloc(nb));
    HIP_TRY(hipMemcpyAsync(c->D.p, A, sizeof(double) * NN, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->scratch_out.p, B, sizeof(double) * nb, hipMemcpyHostToDevice, c->stream));
    int rc = svd_pinv_solve_device(c, c->D.p, c->scratch_out.p, nrhs);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(B, c->scratch_out.p, sizeof(double) * nb, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return FH_OK;
}

int fh_svd_solve_as_reference(fh_ctx *c, const double *A, double *B, int nrhs) {
    if (!c || !A || !B || nrhs < 1) return fail(FH_ERR_INVALID, "fh_svd_solve_as_reference: bad argument");
    if (nrhs != 1 && nrhs != c->N)
        return fail(FH_ERR_INVALID, "operands could not be broadcast together with shapes (%d,%d) (%d,)", c->N, nrhs, c->N);
    HIP_TRY(hipSetDevice(c->device));
    const int N = c->N;
    const size_t NN = (size_t)N * N, nb = (size_t)N * nrhs;
    if (c->scratch_out.n < nb) HIP_TRY(c->scratch_out.alloc(nb));
    HIP_TRY(hipMemcpyAsync(c->D.p, A, sizeof(double) * NN, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->scratch_out.p, B, sizeof(double) * nb, hipMemcpyHostToDevice, c->stream));
    int rc = svd_pinv_solve_device(c, c->D.p, c->scratch_out.p, nrhs, true);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(B, c->scratch_out.p, sizeof(double) * nb, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return FH_OK;
}

// spectral_smoothing_matrix (filter.py:23-62) as bands, then LU of (T + I) without pivoting (T + I is SPD).
// out: f1[N] (multiplier of row i-1), f2[N] (row i-2), d0[N] (pivots), u1[N], u2[N] (upper bands of U).
// the five bands of T / weights_smooth (filter.py:23-62): band[(d + 2) N + i] = T_unit[i][i + d], d = -2 .. 2
static void smoothing_bands(const fh_dht &d, std::vector<double> &band) {
    const int N = d.N;
    std::vector<double> lq(N), dc(N, 0.0), de(N, 0.0), D0(N, 0.0), D1(N, 0.0), D2(N, 0.0);
    band.assign(5 * (size_t)N, 0.0);
    for (int i = 0; i < N; ++i) lq[i] = log(d.q[i]);
    for (int i = 0; i + 2 < N; ++i) dc[i] = (lq[i + 2] - lq[i]) / 2;  // filter.py:42
    for (int i = 0; i + 1 < N; ++i) de[i] = lq[i + 1] - lq[i];        // filter.py:43
    for (int i = 1; i + 1 < N; ++i) {                                 // filter.py:48-50
        D0[i] = 1 / (dc[i - 1] * de[i - 1]);
        D1[i] = -(1 / de[i] + 1 / de[i - 1]) / dc[i - 1];
        D2[i] = 1 / (dc[i - 1] * de[i]);
    }
    for (int i = 1; i + 1 < N; ++i) {  // T = Delta^T (dce Delta), filter.py:55-60
        const double dce = dc[i - 1];
        const int cols[3] = {i - 1, i, i + 1};
        const double vals[3] = {D0[i], D1[i], D2[i]};
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) band[(size_t)(cols[b] - cols[a] + 2) * N + cols[a]] += vals[a] * (dce * vals[b]);
    }
}

static void smoothing_band_lu(const fh_dht &d, double weights, std::vector<double> &out) {
    const int N = d.N;
    std::vector<double> lq(N), dc(N, 0.0), de(N, 0.0), D0(N, 0.0), D1(N, 0.0), D2(N, 0.0);
    std::vector<double> band(5 * (size_t)N, 0.0);
    for (int i = 0; i < N; ++i) lq[i] = log(d.q[i]);
    for (int i = 0; i + 2 < N; ++i) dc[i] = (lq[i + 2] - lq[i]) / 2;  // filter.py:42
    for (int i = 0; i + 1 < N; ++i) de[i] = lq[i + 1] - lq[i];        // filter.py:43
    for (int i = 1; i + 1 < N; ++i) {                                 // filter.py:48-50
        D0[i] = 1 / (dc[i - 1] * de[i - 1]);
        D1[i] = -(1 / de[i] + 1 / de[i - 1]) / dc[i - 1];
        D2[i] = 1 / (dc[i - 1] * de[i]);
    }
    for (int i = 1; i + 1 < N; ++i) {  // T = Delta^T (dce Delta), filter.py:55-60
        const double dce = dc[i - 1];
        const int cols[3] = {i - 1, i, i + 1};
        const double vals[3] = {D0[i], D1[i], D2[i]};
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) band[(size_t)(cols[b] - cols[a] + 2) * N + cols[a]] += vals[a] * (dce * vals[b]);
    }
    std::vector<double> A(5 * (size_t)N);
    for (int i = 0; i < N; ++i)
        for (int dd = -2; dd <= 2; ++dd) A[(size_t)i * 5 + dd + 2] = weights * band[(size_t)(dd + 2) * N + i] + (dd == 0 ? 1.0 : 0.0);
    out.assign(5 * (size_t)N, 0.0);
    double *f1 = out.data(), *f2 = f1 + N, *d0 = f2 + N, *u1 = d0 + N, *u2 = u1 + N;
    for (int k = 0; k < N; ++k) {
        const double piv = A[(size_t)k * 5 + 2];
        for (int i = k + 1; i <= k + 2 && i < N; ++i) {
            const int dk = k - i;
            const double f = A[(size_t)i * 5 + dk + 2] / piv;
            (dk == -1 ? f1 : f2)[i] = f;
            if (f == 0) continue;
            for (int cc = k + 1; cc <= k + 2 && cc < N; ++cc) A[(size_t)i * 5 + (cc - i + 2)] -= f * A[(size_t)k * 5 + (cc - k + 2)];
            A[(size_t)i * 5 + dk + 2] = 0;
        }
    }
    for (int i = 0; i < N; ++i) {
        d0[i] = A[(size_t)i * 5 + 2];
        u1[i] = A[(size_t)i * 5 + 3];
        u2[i] = A[(size_t)i * 5 + 4];
    }
}

static int fit_normal_rocsolver(fh_ctx *c, const double *M, const double *j, double alpha, double p0, double wsmooth, double tol,
                  int max_iter, double *mu, double *p, int *niter, double *diag_p, double *diag_mu) {
    if (!c || !mu || !p || !niter) return fail(FH_ERR_INVALID, "fh_fit_normal: NULL argument");
    if ((M == nullptr) != (j == nullptr)) return fail(FH_ERR_INVALID, "fh_fit_normal: pass both M and j or neither");
    if (!M && !c->have_device_Mj) return fail(FH_ERR_INVALID, "fh_fit_normal: no device-resident M, j (run fh_stats_finalize)");
    if (max_iter < 0) return fail(FH_ERR_INVALID, "max_iter must be >= 0");
    if (c->N > FIT_MAX_N) return fail(FH_ERR_UNSUPPORTED, "N = %d > %d", c->N, FIT_MAX_N);
    HIP_TRY(hipSetDevice(c->device));
    const int N = c->N;
    const size_t NN = (size_t)N * N;
    if (M) {
        HIP_TRY(hipMemcpyAsync(c->M.p, M, sizeof(double) * NN, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(c->j.p, j, sizeof(double) * N, hipMemcpyHostToDevice, c->stream));
        c->have_device_Mj = false;
    }
    std::vector<double> lu;
    smoothing_band_lu(*c->dht, wsmooth, lu);
    HIP_TRY(hipMemcpyAsync(c->band_lu.p, lu.data(), sizeof(double) * lu.size(), hipMemcpyHostToDevice, c->stream));
    const bool want_diag = diag_p || diag_mu;
    if (want_diag) {
        const size_t need = (size_t)(max_iter + 1) * N;
        if (c->diag_p.n < need) HIP_TRY(c->diag_p.alloc(need));
        if (c->diag_mu.n < need) HIP_TRY(c->diag_mu.alloc(need));
    }
    FitState st = make_state(c);
    st.alpha = alpha;
    st.p0 = p0;
    st.tol = tol;
    st.max_iter = max_iter;
    st.diag_p = want_diag ? c->diag_p.p : nullptr;
    st.diag_mu = want_diag ? c->diag_mu.p : nullptr;

    // radial_fitters.py:744-752: p = 1 -> fit -> power-law guess -> fit
    HIP_TRY(fh_k2_launch_init(st, c->stream));
    int rc = solve_posterior(c, st, true, false);
    if (rc) return rc;
    HIP_TRY(fh_k2_launch_powerlaw(st, c->stream));
    rc = solve_posterior(c, st, true, true);
    if (rc) return rc;

    // radial_fitters.py:769-785, in batches; the device keeps the loop state and stops updating once converged
    int flags[FIT_NFLAGS] = {0};
    const int batch = 32;
    for (int launched = 0; launched <= max_iter + 1;) {
        for (int b = 0; b < batch; ++b, ++launched) {
            HIP_TRY(fh_k2_launch_update(st, c->stream));
            rc = solve_posterior(c, st, true, true);
            if (rc) return rc;
            if (want_diag) HIP_TRY(fh_k2_launch_record(st, c->stream));
        }
        HIP_TRY(hipMemcpyAsync(flags, c->flags.p, sizeof flags, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (flags[FIT_FLAG_DONE] || flags[FIT_FLAG_BAD_P] || flags[FIT_FLAG_NOT_SPD]) break;
    }
    if (!flags[FIT_FLAG_DONE] && !flags[FIT_FLAG_BAD_P] && !flags[FIT_FLAG_NOT_SPD]) {
        // one more update launch settles the `count <= max_iter` exit
        HIP_TRY(fh_k2_launch_update(st, c->stream));
        HIP_TRY(hipMemcpyAsync(flags, c->flags.p, sizeof flags, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    *niter = flags[FIT_FLAG_COUNT];
    HIP_TRY(hipMemcpyAsync(mu, c->mu.p, sizeof(double) * N, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(p, c->p.p, sizeof(double) * N, hipMemcpyDeviceToHost, c->stream));
    const size_t nd = (size_t)flags[FIT_FLAG_COUNT] * N;
    if (diag_p && nd) HIP_TRY(hipMemcpyAsync(diag_p, c->diag_p.p, sizeof(double) * nd, hipMemcpyDeviceToHost, c->stream));
    if (diag_mu && nd) HIP_TRY(hipMemcpyAsync(diag_mu, c->diag_mu.p, sizeof(double) * nd, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (flags[FIT_FLAG_BAD_P]) return fail(FH_ERR_BAD_P, "Bad value in power spectrum (non-positive or NaN) at iteration %d", *niter);
    if (flags[FIT_FLAG_NOT_SPD])
        return fail(FH_ERR_NOT_SPD, "Cholesky of M + S^-1 failed at iteration %d (the reference would switch to an SVD)", *niter);
    return FH_OK;
}

// q-space operands of the fit_loop kernel: A = Y^-T M Y^-1 (symmetrised, padded), b = Y^-T j.
static int prepare_qspace(fh_ctx *c, double *Aq, double *bq) {
    const int N = c->N;
    const double one = 1.0, zero = 0.0;
    // T1 = M Yinv (row-major) == column-major Yinv_buf * M_buf
    ROC_TRY(rocblas_dgemm(c->blas, rocblas_operation_none, rocblas_operation_none, N, N, N, &one, c->Yinv.p, N,
                          c->M.p, N, &zero, c->T1.p, N));
    // Araw = Yinv^T T1 (row-major) == column-major T1_buf * Yinv_buf^T
    ROC_TRY(rocblas_dgemm(c->blas, rocblas_operation_none, rocblas_operation_transpose, N, N, N, &one, c->T1.p, N,
                          c->Yinv.p, N, &zero, c->Araw.p, N));
    // b = Yinv^T j: the column-major view of the row-major Yinv buffer is Yinv^T
    ROC_TRY(rocblas_dgemv(c->blas, rocblas_operation_none, N, N, &one, c->Yinv.p, N, c->j.p, 1, &zero, bq, 1));
    HIP_TRY(fh_k2_launch_symmetrize(c->Araw.p, bq, N, c->NP, Aq, c->stream));
    return FH_OK;
}

// Cluster ("latency") mode of the fit loop (fit_loop.hip, clu::): workgroups per fit.  Five -- the first, two helpers of the
// inverse (every block column a wave of its own up to N = 383), two of the trailing update -- run a pass at N = 300 in 98 us
// against 136 on one compute unit (four: 103, three: 111); FRANK_AMD_K2_CLUSTER=1 turns the mode off, 2..8 set the size.
// Small systems (fewer than eight block rows) have nothing to hand over; the wide ones (N > 335) take three (no trailing helpers).
static int env_int(const char *name, int dflt);
static int fit_cluster_size(const fh_ctx *c) {
    const char *e = getenv("FRANK_AMD_K2_CLUSTER");  // (read at every call: tests switch it inside one process)
    int want = e ? atoi(e) : 6;  // (1 + 3 helpers of the inverse + 2 of the trailing update: 74 us per pass at N = 300; five: 77, seven: 72)
    want = want < 1 ? 1 : (want > FIT_CLUSTER_MAX ? FIT_CLUSTER_MAX : want);
    if (want <= 1 || c->NP < 128 || c->NP > fh_k2_loop_max_np()) return 1;
    // the wide instantiations: helpers of the inverse only, a helper wave takes two block columns at most (24 per helper)
    const int need = 1 + (c->NP / 16 + 23) / 24;
    if (c->NP > 336 && !e) want = need > 3 ? need : 3;
    if (want < need) want = need;
    return want;
}

static FitLoopParams make_loop_params(fh_ctx *c, int mode, double alpha, double p0, double tol, int max_iter) {
    FitLoopParams P{};
    P.N = c->N;
    P.NP = c->NP;
    P.max_iter = max_iter;
    P.mode = mode;
    P.alpha = alpha;
    P.p0 = p0;
    P.tol = tol;
    // DHT.transform(MAP) = (2 pi Rmax^2 / j_nN) Ykm mu and m = Y mu with Y = (0.5 j_nN norm) Ykm
    const double norm = 1 / (M_PI * c->dht->Qmax * c->dht->Qmax);
    P.pl_scale = ((2 * M_PI * c->dht->Rmax * c->dht->Rmax) / c->dht->j_nN) / (0.5 * c->dht->j_nN * norm);
    P.A = c->Aq.p;
    P.bq = c->bq.p;
    P.Yinv = c->Yinv.p;
    P.q = c->q.p;
    P.band_lu = c->band_lu.p;
    P.p_init = nullptr;
    P.C = c->Cq.p;
    P.W = c->Wq.p;
    P.WdT = c->WdT.p;
    P.cs = c->cs.p;
    P.mu_out = c->mu_out.p;
    P.p_out = c->p_out.p;
    P.result = c->loop_result.p;
    P.clk_out = c->loop_clocks.p;  // (NULL unless fh_ctx_loop_clocks switched the probe on)
#ifdef FIT_LOOP_TIMING
    if (!c->loop_timing.p && c->loop_timing.alloc(16 + 2048) == hipSuccess) (void)hipMemset(c->loop_timing.p, 0, (16 + 2048) * sizeof(long long));
    P.timing = c->loop_timing.p;
#endif
    return P;
}

int fh_fit_normal(fh_ctx *c, const double *M, const double *j, double alpha, double p0, double wsmooth, double tol,
                  int max_iter, double *mu, double *p, int *niter, double *diag_p, double *diag_mu) {
    if (!c || !mu || !p || !niter) return fail(FH_ERR_INVALID, "fh_fit_normal: NULL argument");
    // N > 639 does not fit the LDS-resident fit_loop kernel: the library loop (rocBLAS + rocSOLVER per iteration) serves
    if (c->use_rocsolver_loop || c->NP > fh_k2_loop_max_np())
        return fit_normal_rocsolver(c, M, j, alpha, p0, wsmooth, tol, max_iter, mu, p, niter, diag_p, diag_mu);
    if ((M == nullptr) != (j == nullptr)) return fail(FH_ERR_INVALID, "fh_fit_normal: pass both M and j or neither");
    if (!M && !c->have_device_Mj) return fail(FH_ERR_INVALID, "fh_fit_normal: no device-resident M, j (run fh_stats_finalize)");
    if (max_iter < 0) return fail(FH_ERR_INVALID, "max_iter must be >= 0");
    if (c->NP > fh_k2_loop_max_np()) return fail(FH_ERR_UNSUPPORTED, "N = %d: the fit_loop kernel covers N <= 1023", c->N);
    HIP_TRY(hipSetDevice(c->device));
    const int N = c->N;
    const size_t NN = (size_t)N * N;
    if (M) {
        HIP_TRY(hipMemcpyAsync(c->M.p, M, sizeof(double) * NN, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(c->j.p, j, sizeof(double) * N, hipMemcpyHostToDevice, c->stream));
        c->have_device_Mj = false;
    }
    std::vector<double> lu;
    smoothing_band_lu(*c->dht, wsmooth, lu);
    HIP_TRY(hipMemcpyAsync(c->band_lu.p, lu.data(), sizeof(double) * lu.size(), hipMemcpyHostToDevice, c->stream));
    const bool want_diag = diag_p || diag_mu;
    if (want_diag) {
        const size_t need = (size_t)(max_iter + 1) * N;
        if (c->diag_p.n < need) HIP_TRY(c->diag_p.alloc(need));
        if (c->diag_mu.n < need) HIP_TRY(c->diag_mu.alloc(need));
    }
    int rc = prepare_qspace(c, c->Aq.p, c->bq.p);
    if (rc) return rc;
    FitLoopParams P = make_loop_params(c, FIT_MODE_FULL, alpha, p0, tol, max_iter);
    P.diag_p = want_diag ? c->diag_p.p : nullptr;
    P.diag_mu = want_diag ? c->diag_mu.p : nullptr;
    // a single fit is what the latency of a pass decides: on a cluster of workgroups unless something else occupies the device
    P.cluster = (c->slots_busy == 0) ? fit_cluster_size(c) : 1;
    int result[2] = {0, 0};
    for (int attempt = 0; attempt < 2; ++attempt) {
        HIP_TRY(hipEventRecord(c->ev_loop0, c->stream));
        HIP_TRY(fh_k2_launch_loop(P, c->stream));
        HIP_TRY(hipEventRecord(c->ev_loop1, c->stream));
        c->loop_timed = true;
        HIP_TRY(hipMemcpyAsync(result, c->loop_result.p, sizeof result, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(mu, c->mu_out.p, sizeof(double) * N, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(p, c->p_out.p, sizeof(double) * N, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (result[1] != FIT_STATUS_CLUSTER) break;
        // the cluster did not assemble (helpers not resident in time, or not on one XCD) or broke: the same fit on one CU
        if (P.cluster <= 1) return fail(FH_ERR_HIP, "fit_loop: unexpected cluster status");
        ++c->cluster_fallbacks;
        HIP_TRY(hipMemsetAsync(c->WdT.p, 0, sizeof(double) * fh_k2_exchange_doubles(c->NP), c->stream));
        P.cluster = 1;
    }
    c->last_fit_cluster = P.cluster;
    *niter = result[0];
    const size_t nd = (size_t)result[0] * N;
    if (diag_p && nd) HIP_TRY(hipMemcpy(diag_p, c->diag_p.p, sizeof(double) * nd, hipMemcpyDeviceToHost));
    if (diag_mu && nd) HIP_TRY(hipMemcpy(diag_mu, c->diag_mu.p, sizeof(double) * nd, hipMemcpyDeviceToHost));
    if (result[1] == FIT_STATUS_BAD_P)
        return fail(FH_ERR_BAD_P, "Bad value in power spectrum (non-positive or NaN) at iteration %d", *niter);
    if (result[1] == FIT_STATUS_NOT_SPD)
        return fail(FH_ERR_NOT_SPD, "Cholesky of the posterior precision failed at iteration %d (the reference would "
                                    "switch to an SVD)", *niter);
    return FH_OK;
}

static int update_power_spectrum_rocsolver(fh_ctx *c, const double *M, const double *j, const double *p, double alpha, double p0,
                             double wsmooth, double *mu, double *p_new) {
    if (!c || !M || !j || !p) return fail(FH_ERR_INVALID, "fh_update_power_spectrum: NULL argument");
    if (c->N > FIT_MAX_N) return fail(FH_ERR_UNSUPPORTED, "N = %d > %d", c->N, FIT_MAX_N);
    HIP_TRY(hipSetDevice(c->device));
    const int N = c->N;
    for (int k = 0; k < N; ++k)
        if (!(p[k] > 0.0)) return fail(FH_ERR_BAD_P, "Bad value in power spectrum (p[%d] = %g)", k, p[k]);
    std::vector<double> lu, zero(N, 0.0);
    smoothing_band_lu(*c->dht, wsmooth, lu);
    HIP_TRY(hipMemcpyAsync(c->M.p, M, sizeof(double) * (size_t)N * N, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->j.p, j, sizeof(double) * N, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->p.p, p, sizeof(double) * N, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->p_old.p, zero.data(), sizeof(double) * N, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->band_lu.p, lu.data(), sizeof(double) * lu.size(), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemsetAsync(c->flags.p, 0, sizeof(int) * FIT_NFLAGS, c->stream));
    c->have_device_Mj = false;
    FitState st = make_state(c);
    st.alpha = alpha;
    st.p0 = p0;
    st.tol = 0.0;
    st.max_iter = 1 << 30;
    int rc = solve_posterior(c, st, true, true);
    if (rc) return rc;
    if (mu) HIP_TRY(hipMemcpyAsync(mu, c->mu.p, sizeof(double) * N, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(fh_k2_launch_update(st, c->stream));
    int flags[FIT_NFLAGS];
    HIP_TRY(hipMemcpyAsync(flags, c->flags.p, sizeof flags, hipMemcpyDeviceToHost, c->stream));
    if (p_new) HIP_TRY(hipMemcpyAsync(p_new, c->p.p, sizeof(double) * N, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (flags[FIT_FLAG_NOT_SPD]) return fail(FH_ERR_NOT_SPD, "Cholesky of M + S^-1 failed");
    return FH_OK;
}

// Launch order of the points of a sweep: ascending alpha, then ascending w_smooth (the long fits first, see fh_fit_normal_batched).
// The caller's values are not validated here -- a NaN hyper-parameter is a per-point status, as before --, so the comparison
// runs on keys that send NaN to +infinity: a strict weak ordering whatever the input (std::stable_sort on `<` of raw doubles
// with a NaN among them is undefined behaviour).
static std::vector<int> sweep_launch_order(const double *alpha, const double *wsmooth, int batch) {
    std::vector<int> order((size_t)batch);
    for (int b = 0; b < batch; ++b) order[b] = b;
    if (getenv("FRANK_AMD_SWEEP_GRID_ORDER")) return order;
    auto key = [](double x) { return std::isnan(x) ? INFINITY : x; };
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) {
        const double ax = key(alpha[x]), ay = key(alpha[y]);
        return ax != ay ? ax < ay : key(wsmooth[x]) < key(wsmooth[y]);
    });
    return order;
}

// The STAGED schedule of a sweep (round 5).  A launch of a sweep ends with its slowest fit, and the fits of a grid differ ~20 x in
// length (BASELINE configs[4]: 102 ... 2 003 passes, median 230; 15 of the 512 run to max_iter).  Which ones are long is only
// known once they run -- so every fit first runs on ONE compute unit in one launch that fills the device (the form of the
// kernel that suits a full device), but PAUSES after `cap` passes (fit_loop.hip: FIT_STATUS_PAUSED; the state of the iteration
// is p and the p before it); then the few that are left -- the long ones, by construction -- continue where they stopped: as
// many as the device holds on CLUSTERS of workgroups (a pass in half the time), the others on one compute unit each beside them.
// Every form of the kernel makes the same bits and a paused fit continues exactly: the results are those of the single launch.
// order[k] = the caller's index of the fit launched k-th.
static int fit_submit_impl(fh_ctx *c, double alpha, double p0, double wsmooth, double tol, int max_iter, int *ticket, const double *resume);
static int sweep_staged(fh_ctx *c, int batch, const std::vector<int> &order, const double *alpha, const double *p0, const double *wsmooth,
                        double tol, int max_iter, int cap, double *mu, double *p, int *niter, int *status) {
    const int N = c->N, NP = c->NP, g = fit_cluster_size(c);
    const size_t PP = (size_t)NP * NP, B = (size_t)batch, RS = 2 * (size_t)N + 1;
    const size_t G = (size_t)(batch < c->num_cu ? batch : c->num_cu);
    DevBuf<double> Cb, Wb, WdTb, csb, mub, pb, lub, alb, p0b, rsb;
    DevBuf<int> resb, counter;
    if (counter.alloc(1) != hipSuccess || Cb.alloc(G * PP) != hipSuccess || Wb.alloc(G * PP) != hipSuccess ||
        WdTb.alloc(G * NP * 16) != hipSuccess || csb.alloc(G * fh_k2_cs_doubles(NP)) != hipSuccess || mub.alloc(B * N) != hipSuccess ||
        pb.alloc(B * N) != hipSuccess || lub.alloc(B * 5 * N) != hipSuccess || alb.alloc(B) != hipSuccess || p0b.alloc(B) != hipSuccess ||
        resb.alloc(2 * B) != hipSuccess || rsb.alloc(B * RS) != hipSuccess)
        return fail(FH_ERR_NOMEM, "fh_fit_normal_batched: device allocation for %d fits failed", batch);
    std::vector<double> lu_all(B * 5 * N), lu, al_o(B), p0_o(B);
    for (int k = 0; k < batch; ++k) {
        smoothing_band_lu(*c->dht, wsmooth[order[k]], lu);
        memcpy(lu_all.data() + (size_t)k * 5 * N, lu.data(), sizeof(double) * 5 * N);
        al_o[k] = alpha[order[k]];
        p0_o[k] = p0[order[k]];
    }
    std::vector<int> res(2 * B);
    std::vector<double> mu_o(B * N), p_o(B * N);
    // one batched launch over n fits whose per-fit inputs sit in the first n entries of the host arrays; results into res / mu_o / p_o
    auto launch = [&](int n, int mode, int pass_cap, int grid, int loaded) -> int {
        HIP_TRY(hipMemsetAsync(counter.p, 0, sizeof(int), c->stream));
        HIP_TRY(hipMemcpyAsync(lub.p, lu_all.data(), sizeof(double) * (size_t)n * 5 * N, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(alb.p, al_o.data(), sizeof(double) * (size_t)n, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(p0b.p, p0_o.data(), sizeof(double) * (size_t)n, hipMemcpyHostToDevice, c->stream));
        FitLoopParams P = make_loop_params(c, mode, 0.0, 0.0, tol, max_iter);
        P.band_lu = lub.p;
        P.C = Cb.p;
        P.W = Wb.p;
        P.WdT = WdTb.p;
        P.cs = csb.p;
        P.mu_out = mub.p;
        P.p_out = pb.p;
        P.result = resb.p;
        P.batch = n;
        P.batch_alpha = alb.p;
        P.batch_p0 = p0b.p;
        P.batch_counter = counter.p;
        P.pass_cap = pass_cap;
        P.resume = mode == FIT_MODE_RESUME ? rsb.p : nullptr;
        P.loaded = loaded;
        HIP_TRY(fh_k2_launch_loop_batched(P, grid, c->stream));
        HIP_TRY(hipMemcpyAsync(res.data(), resb.p, sizeof(int) * 2 * (size_t)n, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(mu_o.data(), mub.p, sizeof(double) * (size_t)n * N, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(p_o.data(), pb.p, sizeof(double) * (size_t)n * N, hipMemcpyDeviceToHost, c->stream));
        return FH_OK;
    };
    auto finish = [&](int k, int slot) {  // fit launched k-th, its results in entry `slot` of res / mu_o / p_o
        const int b = order[k];
        memcpy(mu + (size_t)b * N, mu_o.data() + (size_t)slot * N, sizeof(double) * N);
        memcpy(p + (size_t)b * N, p_o.data() + (size_t)slot * N, sizeof(double) * N);
        niter[b] = res[2 * slot];
        if (status)
            status[b] = res[2 * slot + 1] == FIT_STATUS_BAD_P ? FH_ERR_BAD_P : res[2 * slot + 1] == FIT_STATUS_NOT_SPD ? FH_ERR_NOT_SPD : FH_OK;
    };
    const bool trace = getenv("FRANK_AMD_SWEEP_TRACE") != nullptr;  // development: stage times on stderr
    const auto t_start = std::chrono::steady_clock::now();
    auto ms_since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count(); };
    // ---- stage 1: every fit, at most `cap` passes ----
    int rc = launch(batch, FIT_MODE_FULL, cap, (int)G, batch > (int)G ? c->num_cu : 0);  // (more fits than units: the device stays full)
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    std::vector<int> paused;                 // launch-order indices of the fits that stopped at the cap
    std::vector<double> state;               // their states, RS doubles each
    for (int k = 0; k < batch; ++k) {
        if (res[2 * k + 1] != FIT_STATUS_PAUSED) {
            finish(k, k);
            continue;
        }
        paused.push_back(k);
        const size_t o = state.size();
        state.resize(o + RS);
        memcpy(&state[o], p_o.data() + (size_t)k * N, sizeof(double) * N);           // p
        memcpy(&state[o + N], mu_o.data() + (size_t)k * N, sizeof(double) * N);      // p_old (in the place of mu)
        state[o + 2 * N] = (double)res[2 * k];
    }
    const int np = (int)paused.size();
    const double t_stage1 = ms_since();
    if (np == 0) return FH_OK;
    // ---- stage 2: the first Kc of them (launch order: the longest first) on clusters, the others on one compute unit each ----
    // (workgroup ids go round the eight XCDs and the members of a cluster share one: the clusters come in eights, and an XCD's
    //  32 units must hold its clusters AND its share of the one-unit loops -- an XCD asked for more makes clusters wait for units
    //  beyond the 3 ms they are given to assemble, and a cluster that does not assemble is rerun on one unit at collection)
    int Kc = 0;
    for (int k8 = 8 * ((np + 7) / 8); k8 >= 0; k8 -= 8) {
        const int kc = k8 < np ? k8 : np, rest = np - kc;
        const int per_xcd = ((kc + 7) / 8) * g + (rest + 7) / 8;
        if (per_xcd <= c->num_cu / 8 - 4) {  // (four units of an XCD left free: the small kernels of the stage -- the q-space
                                             //  operands of every submission -- need somewhere to run; with 31 of 32 units
                                             //  spoken for, 4-6 of 40 clusters missed their 3 ms in one run of three)
            Kc = kc;
            break;
        }
    }
    Kc = env_int("FRANK_AMD_SWEEP_STAGE2_CLUSTERS", Kc) < np ? env_int("FRANK_AMD_SWEEP_STAGE2_CLUSTERS", Kc) : np;
    std::vector<int> tickets(Kc, -1);
    struct TicketGuard {
        fh_ctx *c;
        std::vector<int> &t;
        ~TicketGuard() {
            for (int &x : t)
                if (x >= 0 && c->slots[x].busy) {
                    (void)fh_fit_collect(c, x, nullptr, nullptr, nullptr);
                    x = -1;
                }
        }
    } ticket_guard{c, tickets};
    if (Kc > 0) {
        const bool had = c->have_device_Mj;
        c->have_device_Mj = true;  // (M, j are on the device: uploaded by the caller of this function or by its caller's finalisation)
        int rcs = FH_OK;
        for (int i = 0; i < Kc && rcs == FH_OK; ++i) {
            const int b = order[paused[i]];
            rcs = fit_submit_impl(c, alpha[b], p0[b], wsmooth[b], tol, max_iter, &tickets[i], &state[(size_t)i * RS]);
        }
        c->force_cluster_launch = true;
        if (rcs == FH_OK) rcs = fh_fit_flush(c);
        c->force_cluster_launch = false;
        c->have_device_Mj = had;
        if (rcs != FH_OK) return rcs;
    }
    const int n2 = np - Kc;
    if (n2 > 0) {  // the per-fit inputs of the others, compacted to the front of the host arrays
        for (int i = 0; i < n2; ++i) {
            const int k = paused[Kc + i];
            memmove(lu_all.data() + (size_t)i * 5 * N, lu_all.data() + (size_t)k * 5 * N, sizeof(double) * 5 * N);  // (i <= k)
            al_o[i] = al_o[k];
            p0_o[i] = p0_o[k];
        }
        HIP_TRY(hipMemcpyAsync(rsb.p, &state[(size_t)Kc * RS], sizeof(double) * (size_t)n2 * RS, hipMemcpyHostToDevice, c->stream));
        int free_cus = c->num_cu - Kc * g;
        if (free_cus < 1) free_cus = 1;
        rc = launch(n2, FIT_MODE_RESUME, 0, n2 < free_cus ? n2 : free_cus, 0);
        if (rc) return rc;
        HIP_TRY(hipStreamSynchronize(c->stream));
        for (int i = 0; i < n2; ++i) finish(paused[Kc + i], i);
    }
    const double t_batched2 = ms_since();
    const unsigned long long fb0 = c->cluster_fallbacks;
    for (int i = 0; i < Kc; ++i) {
        const int b = order[paused[i]];
        const int rcc = fh_fit_collect(c, tickets[i], mu + (size_t)b * N, p + (size_t)b * N, &niter[b]);
        tickets[i] = -1;
        if (rcc != FH_OK && rcc != FH_ERR_BAD_P && rcc != FH_ERR_NOT_SPD) return rcc;
        if (status) status[b] = rcc;
    }
    if (trace)
        fprintf(stderr, "[sweep_staged] %d fits, cap %d: stage 1 %.1f ms, %d paused -> %d on clusters of %d + %d on one unit; one-unit part done at "
                "%.1f ms, clusters at %.1f ms (%llu fall-backs)\n", batch, cap, t_stage1, np, Kc, g, n2, t_batched2, ms_since(),
                (unsigned long long)(c->cluster_fallbacks - fb0));
    return FH_OK;
}

int fh_fit_normal_batched(fh_ctx *c, const double *M, const double *j, int batch, const double *alpha, const double *p0,
                          const double *wsmooth, double tol, int max_iter, double *mu, double *p, int *niter,
                          int *status) {
    if (!c || !alpha || !p0 || !wsmooth || !mu || !p || !niter || batch < 1)
        return fail(FH_ERR_INVALID, "fh_fit_normal_batched: bad argument");
    if ((M == nullptr) != (j == nullptr)) return fail(FH_ERR_INVALID, "pass both M and j or neither");
    if (!M && !c->have_device_Mj) return fail(FH_ERR_INVALID, "no device-resident M, j (run fh_stats_finalize)");
    if (c->NP > fh_k2_loop_max_np()) return fail(FH_ERR_UNSUPPORTED, "N = %d: the fit_loop kernel covers N <= 1023", c->N);
    HIP_TRY(hipSetDevice(c->device));
    const int N = c->N, NP = c->NP;
    const size_t NN = (size_t)N * N, PP = (size_t)NP * NP;
    if (M) {
        HIP_TRY(hipMemcpyAsync(c->M.p, M, sizeof(double) * NN, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(c->j.p, j, sizeof(double) * N, hipMemcpyHostToDevice, c->stream));
        c->have_device_Mj = false;
    }
    int rc = prepare_qspace(c, c->Aq.p, c->bq.p);
    if (rc) return rc;
    // per-fit work buffers and parameters
    DevBuf<double> Cb, Wb, WdTb, csb, mub, pb, lub, alb, p0b;
    DevBuf<int> resb;
    const size_t B = (size_t)batch;
    // work buffers per resident workgroup (at most one per CU), outputs per fit
    const size_t G = (size_t)(batch < c->num_cu ? batch : c->num_cu);
    DevBuf<int> counter;
    if (counter.alloc(1) != hipSuccess) return fail(FH_ERR_NOMEM, "device allocation failed");
    HIP_TRY(hipMemsetAsync(counter.p, 0, sizeof(int), c->stream));
    if (Cb.alloc(G * PP) != hipSuccess || Wb.alloc(G * PP) != hipSuccess ||
        WdTb.alloc(G * NP * 16) != hipSuccess || csb.alloc(G * fh_k2_cs_doubles(NP)) != hipSuccess ||
        mub.alloc(B * N) != hipSuccess || pb.alloc(B * N) != hipSuccess || lub.alloc(B * 5 * N) != hipSuccess ||
        alb.alloc(B) != hipSuccess || p0b.alloc(B) != hipSuccess || resb.alloc(2 * B) != hipSuccess)
        return fail(FH_ERR_NOMEM, "fh_fit_normal_batched: device allocation for %d fits failed", batch);
    // The workgroups pull the fits in launch order, and the launch ends with its slowest fit: the points most likely to run
    // long go first.  The iteration count grows as alpha approaches 1 (filter.py:172: the update of p is damped by alpha - 1/2)
    // and, at equal alpha, with a weaker smoothing prior; on the 32 x 16 grid of BASELINE configs[4] the seven points that reach
    // max_iter all have alpha = 1.01 -- in grid order the last of them started 100 ms into the launch.  order[k] = the caller's
    // index of the fit launched k-th; the outputs are put back in the caller's order.
    const std::vector<int> order = sweep_launch_order(alpha, wsmooth, batch);
    // the staged schedule (sweep_staged above): sweeps of at least 64 points on an idle context, sizes the deferred kernel covers
    {
        const int gsz = fit_cluster_size(c);
        // (the cap: BASELINE configs[4], 512 fits: 640 / 800 / 900 / 1 000 / 1 200 passes -> 1 508 / 2 047 / 2 014 / 1 914 / 1 826 fits/s,
        //  the single launch with its sixteen longest on clusters 1 515-1 650; 0 turns the schedule off)
        const int cap = env_int("FRANK_AMD_SWEEP_CAP", 800);
        if (cap > 0 && gsz > 1 && batch >= 64 && c->slots_busy == 0 && c->pending_batch < 0 && !getenv("FRANK_AMD_SWEEP_NO_CLUSTERS") &&
            max_iter > cap)
            return sweep_staged(c, batch, order, alpha, p0, wsmooth, tol, max_iter, cap, mu, p, niter, status);
    }
    // ... and the first K of them -- the ones that will still be iterating when every other fit of the sweep has ended -- do not
    // join the batch at all: they are launched on CLUSTERS of workgroups (fit_loop.hip: 98 instead of 136 us per pass once the
    // device has emptied) through the fit slots, beside the batched launch of the rest on the compute units they leave free.
    int K = 0;
    {
        const int g = fit_cluster_size(c);
        if (g > 1 && batch >= 64 && c->slots_busy == 0 && c->pending_batch < 0 && !getenv("FRANK_AMD_SWEEP_NO_CLUSTERS")) {
            K = batch / 8 < 16 ? batch / 8 : 16;
            K = env_int("FRANK_AMD_SWEEP_CLUSTERS", K);  // (development: how many of the longest points go to clusters)
            if (K > batch) K = batch;
            if (K * g > c->num_cu / 2) K = c->num_cu / 2 / g;
        }
    }
    std::vector<int> tickets(K, -1);
    // whatever path leaves this function: every ticket issued and not yet collected is collected (results dropped) -- a slot left
    // busy would shrink the pool and keep later fits of this context off the clusters (slots_busy never back to 0)
    struct TicketGuard {
        fh_ctx *c;
        std::vector<int> &t;
        ~TicketGuard() {
            for (int &x : t)
                if (x >= 0 && c->slots[x].busy) {
                    (void)fh_fit_collect(c, x, nullptr, nullptr, nullptr);
                    x = -1;
                }
        }
    } ticket_guard{c, tickets};
    if (K > 0) {
        const bool had = c->have_device_Mj;
        c->have_device_Mj = true;  // (M, j are on the device: uploaded above or by the caller's finalisation)
        int rcs = FH_OK;
        for (int k = 0; k < K && rcs == FH_OK; ++k)
            rcs = fh_fit_submit(c, alpha[order[k]], p0[order[k]], wsmooth[order[k]], tol, max_iter, &tickets[k]);
        c->force_cluster_launch = true;
        if (rcs == FH_OK) rcs = fh_fit_flush(c);
        c->force_cluster_launch = false;
        c->have_device_Mj = had;
        if (rcs != FH_OK) return rcs;
    }
    const size_t BR = B - (size_t)K;  // fits of the batched launch: order[K ..]
    std::vector<double> lu_all(BR * 5 * N + 1), lu, al_o(BR + 1), p0_o(BR + 1);
    for (int k = K; k < batch; ++k) {
        smoothing_band_lu(*c->dht, wsmooth[order[k]], lu);
        memcpy(lu_all.data() + (size_t)(k - K) * 5 * N, lu.data(), sizeof(double) * 5 * N);
        al_o[k - K] = alpha[order[k]];
        p0_o[k - K] = p0[order[k]];
    }
    if (BR > 0) {
        HIP_TRY(hipMemcpyAsync(lub.p, lu_all.data(), sizeof(double) * BR * 5 * N, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(alb.p, al_o.data(), sizeof(double) * BR, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(p0b.p, p0_o.data(), sizeof(double) * BR, hipMemcpyHostToDevice, c->stream));
    }
    FitLoopParams P = make_loop_params(c, FIT_MODE_FULL, 0.0, 0.0, tol, max_iter);
    P.band_lu = lub.p;
    P.C = Cb.p;
    P.W = Wb.p;
    P.WdT = WdTb.p;
    P.cs = csb.p;
    P.mu_out = mub.p;
    P.p_out = pb.p;
    P.result = resb.p;
    P.batch = (int)BR;
    P.batch_alpha = alb.p;
    P.batch_p0 = p0b.p;
    P.batch_counter = counter.p;
    P.loaded = K * fit_cluster_size(c);  // (compute units the clusters hold beside this launch)
    {
        // (workgroups of the batched launch: one per fit, at most the compute units the clusters leave free)
        int free_cus = c->num_cu - K * fit_cluster_size(c);
        if (free_cus < 1) free_cus = 1;
        const int grid = (int)(BR < (size_t)free_cus ? BR : (size_t)free_cus);
        if (BR > 0) HIP_TRY(fh_k2_launch_loop_batched(P, grid < (int)G ? grid : (int)G, c->stream));
    }
    std::vector<int> res(2 * B);
    std::vector<double> mu_o(B * N), p_o(B * N);
    if (BR > 0) {
        HIP_TRY(hipMemcpyAsync(res.data(), resb.p, sizeof(int) * 2 * BR, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(mu_o.data(), mub.p, sizeof(double) * BR * N, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(p_o.data(), pb.p, sizeof(double) * BR * N, hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int k = K; k < batch; ++k) {
        const int b = order[k], kb = k - K;
        memcpy(mu + (size_t)b * N, mu_o.data() + (size_t)kb * N, sizeof(double) * N);
        memcpy(p + (size_t)b * N, p_o.data() + (size_t)kb * N, sizeof(double) * N);
        niter[b] = res[2 * kb];
        if (status)
            status[b] = res[2 * kb + 1] == FIT_STATUS_BAD_P ? FH_ERR_BAD_P
                        : res[2 * kb + 1] == FIT_STATUS_NOT_SPD ? FH_ERR_NOT_SPD : FH_OK;
    }
    for (int k = 0; k < K; ++k) {  // the fits that ran on clusters
        const int b = order[k];
        const int rcc = fh_fit_collect(c, tickets[k], mu + (size_t)b * N, p + (size_t)b * N, &niter[b]);
        tickets[k] = -1;  // (collected, whatever it returned)
        if (rcc != FH_OK && rcc != FH_ERR_BAD_P && rcc != FH_ERR_NOT_SPD) return rcc;
        if (status) status[b] = rcc;
    }
    return FH_OK;
}

// Defaults from a sweep of (streams, fits per launch, slots) at the headline size, fits/s at steady state: 3/32/128 825,
// 3/43/172 858, 3/64/192 931, 3/64/240 929, 4/32/160 920, 4/48/240 905, 5/32/192 919, 2/120/240 1024, **4/64/240 966-1001**; the sixteen launches of
// sixteen on sixteen streams this replaces: 711-740.
static int fit_launch_streams() {  // streams the launches are dealt to, idle ones first (FRANK_AMD_FIT_STREAMS, 1 .. 8)
    int n = 6;
    if (const char *e = getenv("FRANK_AMD_FIT_STREAMS")) n = atoi(e);
    return n < 1 ? 1 : (n > kLaunchStreamsMax ? kLaunchStreamsMax : n);
}
static int fit_batch_size() {  // fit loops per launch (FRANK_AMD_FIT_BATCH = 1 (launch at once) .. 128)
    int b = 64;
    if (const char *e = getenv("FRANK_AMD_FIT_BATCH")) b = atoi(e);
    return b < 1 ? 1 : (b > kFitBatchMax ? kFitBatchMax : b);
}
static int fit_slots_wanted() {  // fit loops in flight: every one holds a compute unit for the ~0.1 s of its iteration
    int n = 240;  // (kFitSlots = 512 is the capacity)
    if (const char *e = getenv("FRANK_AMD_FIT_SLOTS")) n = atoi(e);
    return n < 1 ? 1 : (n > kFitSlots ? kFitSlots : n);
}
int fh_fit_slots(void) {  // fits that may be outstanding: bounded by the slots and by the launches in flight
    // (a launch is free again when ALL its fits are collected: with first-in first-out collection one launch may be
    //  partly collected)
    const int by_launch = (kFitBatches - 1) * fit_batch_size() + 1, want = fit_slots_wanted();
    return by_launch < want ? by_launch : want;
}

// launch the batch that is collecting submissions (no-op if there is none)
static int flush_pending_batch(fh_ctx *c) {
    if (c->pending_batch < 0) return FH_OK;
    FitBatch &b = c->batches[c->pending_batch];
    c->pending_batch = -1;
    if (b.n == 0) {
        b.active = false;
        return FH_OK;
    }
    {   // a stream whose last launch has ended, if there is one (a pipeline that is filling up sends launches of 1, 2, 4, ..
        // fits: queued behind an earlier launch on its stream such a launch would start a whole fit late); else the next in turn
        int pick = -1;
        for (int i = 0; i < c->n_launch_streams && pick < 0; ++i) {
            const int j = (int)((c->launches + (unsigned long long)i) % (unsigned long long)c->n_launch_streams);
            if (!c->stream_last_done[j] || hipEventQuery(c->stream_last_done[j]) == hipSuccess) pick = j;
        }
        (void)hipGetLastError();  // (hipErrorNotReady of the queries)
        if (pick < 0) pick = (int)(c->launches % (unsigned long long)c->n_launch_streams);
        ++c->launches;
        b.stream = c->launch_streams[pick];
        c->stream_last_done[pick] = b.done;
    }
    // few fits outstanding: every fit of this launch on a cluster of workgroups (the latency of a pass is what a shallow
    // pipeline waits for).  Passes at N = 300 with n fits at once (tools/k2_concurrency.py): on clusters of five 95 us (8 fits), 96
    // (16), 101 (20), 105 (32); on one CU each 138-146.  (The first version of the mode -- agent-scope invalidates that wrote the
    // L2 back, band tiles stored and reloaded every step -- moved so many bytes that twenty clusters ran no faster than twenty
    // single loops; with device-scope loads and the workers' rows in registers they do.)  So: clusters while at most
    // FRANK_AMD_K2_CLUSTER_FITS (32: 160 of the 256 compute units) fits are outstanding.
    {
        static const int most = env_int("FRANK_AMD_K2_CLUSTER_FITS", 32);
        const int g = fit_cluster_size(c);
        b.cluster = (g > 1 && (c->slots_busy <= most || c->force_cluster_launch)) ? g : 1;  // (slots_busy counts the fits of this launch too)
    }
    HIP_TRY(hipEventRecord(b.ready, c->stream));  // the operands of its fits were prepared on the context's stream
    HIP_TRY(hipStreamWaitEvent(b.stream, b.ready, 0));
    FitLoopParams P = make_loop_params(c, b.mode, b.alpha, b.p0, b.tol, b.max_iter);
    const FitSlot &s0 = c->slots[0];
    P.A = s0.Aq.p;
    P.bq = s0.bq.p;
    P.band_lu = s0.band_lu.p;
    P.C = s0.Cq.p;
    P.W = s0.Wq.p;
    P.WdT = s0.WdT.p;
    P.cs = s0.cs.p;
    P.mu_out = s0.mu_out.p;
    P.p_out = s0.p_out.p;
    P.result = s0.result.p;
    P.slot_stride = c->slot_stride;
    for (int i = 0; i < FIT_MAX_BATCH / 4; ++i) P.slot_words[i] = 0;
    for (int i = 0; i < b.n; ++i) P.slot_words[i >> 2] |= (unsigned long long)b.slots[i] << (16 * (i & 3));
    P.out_host = c->slot_out_host;
    P.result_host = c->slot_result_host;
    P.cluster = b.cluster;
    P.loaded = env_int("FRANK_AMD_FIT_LOADED", 0);  // (development: what launch_loop takes for the loops resident beside this launch's)
    if (b.cluster > 1) {  // the fits of consecutive cluster launches go round the XCDs
        P.cluster_xcd0 = c->next_xcd & 7;
        c->next_xcd = (c->next_xcd + b.n) & 7;
    }
    HIP_TRY(fh_k2_launch_loop_slots(P, b.n, b.stream));
    HIP_TRY(hipEventRecord(b.done, b.stream));
    b.launched = true;
    return FH_OK;
}

int fh_fit_flush(fh_ctx *c) {
    if (!c) return fail(FH_ERR_INVALID, "fh_fit_flush: NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    return flush_pending_batch(c);
}

static int fit_submit_impl(fh_ctx *c, double alpha, double p0, double wsmooth, double tol, int max_iter, int *ticket, const double *resume);
int fh_fit_submit(fh_ctx *c, double alpha, double p0, double wsmooth, double tol, int max_iter, int *ticket) {
    return fit_submit_impl(c, alpha, p0, wsmooth, tol, max_iter, ticket, nullptr);
}
// resume != NULL: the fit continues from a paused state, [p (N), p_old (N), passes made] (kernels.h: FIT_MODE_RESUME)
static int fit_submit_impl(fh_ctx *c, double alpha, double p0, double wsmooth, double tol, int max_iter, int *ticket, const double *resume) {
    if (!c || !ticket) return fail(FH_ERR_INVALID, "fh_fit_submit: NULL argument");
    if (!c->have_device_Mj) return fail(FH_ERR_INVALID, "fh_fit_submit: no device-resident M, j (run fh_stats_finalize)");
    if (c->NP > fh_k2_loop_max_np()) return fail(FH_ERR_UNSUPPORTED, "N = %d: the fit_loop kernel covers N <= 1023", c->N);
    if (max_iter < 0) return fail(FH_ERR_INVALID, "max_iter must be >= 0");
    HIP_TRY(hipSetDevice(c->device));
    if (!c->n_slots) c->n_slots = fit_slots_wanted();
    int si = -1;
    for (int i = 0; i < c->n_slots; ++i)
        if (!c->slots[i].busy) {
            si = i;
            break;
        }
    if (si < 0) return fail(FH_ERR_INVALID, "fh_fit_submit: all %d fit slots are outstanding; collect one first", c->n_slots);
    const int N = c->N;
    const size_t PP = (size_t)c->NP * c->NP;
    if (!c->slot_pool.p) {
        // all slots at once, carved from ONE allocation: a hipMalloc per buffer costs ~0.7 ms of host time, and paying
        // 11 of them whenever a fresh slot is first used put an 8 ms hole after every binning pass of a pipeline
        const size_t per_slot = 3 * PP + (size_t)c->NP * 16 + fh_k2_cs_doubles(c->NP) + 3 * (size_t)N + 7 * (size_t)N + 4;
        HIP_TRY(c->slot_pool.alloc(per_slot * (size_t)c->n_slots));
        HIP_TRY(c->slot_results.alloc(2 * (size_t)c->n_slots));
        HIP_TRY(hipMemsetAsync(c->slot_pool.p, 0, sizeof(double) * per_slot * (size_t)c->n_slots, c->stream));
        c->slot_stride = per_slot;
        for (int i = 0; i < c->n_slots; ++i) {
            FitSlot &t = c->slots[i];
            double *b = c->slot_pool.p + per_slot * i;
            t.Aq.adopt(b, PP); b += PP;
            t.Cq.adopt(b, PP); b += PP;
            t.Wq.adopt(b, PP); b += PP;
            t.WdT.adopt(b, (size_t)c->NP * 16); b += (size_t)c->NP * 16;
            t.cs.adopt(b, fh_k2_cs_doubles(c->NP)); b += fh_k2_cs_doubles(c->NP);
            t.bq.adopt(b, N); b += N;
            t.mu_out.adopt(b, N); b += N;
            t.p_out.adopt(b, N); b += N;
            t.band_lu.adopt(b, 7 * (size_t)N + 4);  // + alpha, p0 of the fit (read by the slot launch) + the state of a paused fit
            t.result.adopt(c->slot_results.p + 2 * i, 2);
        }
        c->n_launch_streams = fit_launch_streams();
        for (int i = 0; i < c->n_launch_streams; ++i) {
            // FRANK_AMD_FIT_RESERVE_CUS = B (development): the fit loops keep off the first B compute units, the binning stream
            // stays free to use every unit -- a floor under the binning pass of a deep pipeline instead of a partition
            const int reserve = env_int("FRANK_AMD_FIT_RESERVE_CUS", 0);
            if (c->bin_cus > 0 || (reserve >= 8 && reserve <= c->num_cu - 8)) {  // fh_ctx_set_cu_partition: the fit loops keep to the compute units the binning pass leaves alone
                uint32_t mask[8];
                cu_mask(c->bin_cus > 0 ? c->bin_cus : reserve, c->num_cu, mask);
                HIP_TRY(hipExtStreamCreateWithCUMask(&c->launch_streams[i], 8, mask));
            } else {
                HIP_TRY(hipStreamCreateWithFlags(&c->launch_streams[i], hipStreamNonBlocking));
            }
        }
        for (auto &bt : c->batches) {
            HIP_TRY(hipEventCreateWithFlags(&bt.ready, hipEventDisableTiming | hipEventReleaseToDevice));  // (same device: no system-scope write-back)
            HIP_TRY(hipEventCreateWithFlags(&bt.done, hipEventDisableTiming));  // (system-scope release: the host reads the mirrors)
        }
        HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&c->slot_out_host), sizeof(double) * 2 * (size_t)N * (size_t)c->n_slots, hipHostMallocDefault));
        HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&c->slot_result_host), sizeof(int) * 2 * (size_t)c->n_slots, hipHostMallocDefault));
        c->fit_batch = fit_batch_size();
    }
    // a launch carries ONE (tol, max_iter); alpha, p0 and w_smooth are per fit (they travel with the slot's band LU)
    if (c->slots_busy == 0) c->burst_next = 1;  // an empty pipeline: the first launches are small (1, 2, 4, .. fits)
    if (c->pending_batch >= 0) {
        const FitBatch &pb = c->batches[c->pending_batch];
        const int mode = resume ? FIT_MODE_RESUME : FIT_MODE_FULL;
        if (pb.tol != tol || pb.max_iter != max_iter || pb.mode != mode) {
            int rc = flush_pending_batch(c);
            if (rc) return rc;
        }
    }
    if (c->pending_batch < 0) {
        int bi = -1;
        for (int i = 0; i < kFitBatches; ++i)
            if (!c->batches[i].active) {
                bi = i;
                break;
            }
        if (bi < 0) return fail(FH_ERR_INVALID, "fh_fit_submit: all %d launches are outstanding; collect first", kFitBatches);
        FitBatch &nb = c->batches[bi];
        nb.active = true;
        nb.launched = false;
        nb.n = nb.outstanding = 0;
        nb.alpha = alpha;
        nb.p0 = p0;
        nb.tol = tol;
        nb.max_iter = max_iter;
        nb.mode = resume ? FIT_MODE_RESUME : FIT_MODE_FULL;
        c->pending_batch = bi;
    }
    FitSlot &s = c->slots[si];
    // (the factors of T + I depend on the hyper-parameters only: a slot that already holds them -- every slot of a pipeline
    //  over one set of hyper-parameters, once it has been round -- skips the copy, one kernel boundary of the step less)
    if (!(s.lu_valid && s.lu_key[0] == wsmooth && s.lu_key[1] == alpha && s.lu_key[2] == p0)) {
        s.lu_valid = false;
        smoothing_band_lu(*c->dht, wsmooth, s.lu_host);  // the slot owns the host copy: no wait for the copy here
        s.lu_host.resize(5 * (size_t)N);
        s.lu_host.push_back(alpha);
        s.lu_host.push_back(p0);
        HIP_TRY(hipMemcpyAsync(s.band_lu.p, s.lu_host.data(), sizeof(double) * s.lu_host.size(), hipMemcpyHostToDevice, c->stream));
        s.lu_key[0] = wsmooth;
        s.lu_key[1] = alpha;
        s.lu_key[2] = p0;
        s.lu_valid = true;
    }
    if (resume) {
        s.resume_host.assign(resume, resume + 2 * (size_t)N + 1);
        HIP_TRY(hipMemcpyAsync(s.band_lu.p + 5 * (size_t)N + 2, s.resume_host.data(), sizeof(double) * s.resume_host.size(),
                               hipMemcpyHostToDevice, c->stream));
    }
    int rc = prepare_qspace(c, s.Aq.p, s.bq.p);  // on the context's stream, after the finalize that produced M, j
    if (rc) return rc;
    FitBatch &b = c->batches[c->pending_batch];
    b.slots[b.n++] = (unsigned short)si;
    ++b.outstanding;
    s.batch = c->pending_batch;
    s.busy = true;
    ++c->slots_busy;
    *ticket = si;
    // A pipeline that is filling up sends its first launches small (1, 2, 4 fits) so that the first fits start at once -- but
    // only while fewer than three launches are in flight: the command processor runs about four queues at a time, a fifth
    // launch waits for a whole fit loop to end (measured: 20 fits in launches of 1, 2, 4, 8, 5 took 229 ms against 116 in one),
    // and the binning stream needs its own.
    // OFF by default (FRANK_AMD_FIT_EARLY=1 turns it on): the first fits come back a launch earlier, but the fits of the small
    // launches run on clusters and the later ones, on one CU each, beside them: 20 fits 128 ms against 115 in one launch, and the
    // driver's 20-step region 178 fits/s against 200 -- that region ends with the LAST fit's iteration whenever it is launched.
    static const int early = env_int("FRANK_AMD_FIT_EARLY", 0);
    int in_flight = 0;
    for (const FitBatch &o : c->batches)
        if (o.active && o.launched && hipEventQuery(o.done) != hipSuccess) ++in_flight;
    (void)hipGetLastError();
    const bool small_ok = early && in_flight < 3 && c->burst_next < c->fit_batch;
    const int trigger = small_ok ? c->burst_next : c->fit_batch;
    if (b.n >= trigger) {
        c->burst_next = 2 * trigger;
        return flush_pending_batch(c);
    }
    return FH_OK;
}

int fh_fit_collect(fh_ctx *c, int ticket, double *mu, double *p, int *niter) {
    if (!c || ticket < 0 || ticket >= c->n_slots || !c->slots[ticket].busy)
        return fail(FH_ERR_INVALID, "fh_fit_collect: bad ticket %d", ticket);
    HIP_TRY(hipSetDevice(c->device));
    FitSlot &s = c->slots[ticket];
    FitBatch &b = c->batches[s.batch];
    if (!b.launched) {  // its launch is still collecting submissions: send it now
        int rc = flush_pending_batch(c);
        if (rc) return rc;
    }
    const int N = c->N;
    HIP_TRY(hipEventSynchronize(b.done));  // this fit's launch (later launches on the same stream are not waited for)
    if (c->slot_result_host[2 * ticket + 1] == FIT_STATUS_CLUSTER) {
        // its cluster did not assemble (or broke): the same fit on one CU, now; the control words of the slot back to zero
        ++c->cluster_fallbacks;
        HIP_TRY(hipMemsetAsync(s.WdT.p, 0, sizeof(double) * fh_k2_exchange_doubles(c->NP), b.stream));
        FitLoopParams P = make_loop_params(c, b.mode, b.alpha, b.p0, b.tol, b.max_iter);
        const FitSlot &s0 = c->slots[0];
        P.A = s0.Aq.p;
        P.bq = s0.bq.p;
        P.band_lu = s0.band_lu.p;
        P.C = s0.Cq.p;
        P.W = s0.Wq.p;
        P.WdT = s0.WdT.p;
        P.cs = s0.cs.p;
        P.mu_out = s0.mu_out.p;
        P.p_out = s0.p_out.p;
        P.result = s0.result.p;
        P.slot_stride = c->slot_stride;
        for (int i = 0; i < FIT_MAX_BATCH / 4; ++i) P.slot_words[i] = 0;
        P.slot_words[0] = (unsigned long long)ticket;
        P.out_host = c->slot_out_host;
        P.result_host = c->slot_result_host;
        HIP_TRY(fh_k2_launch_loop_slots(P, 1, b.stream));
        HIP_TRY(hipStreamSynchronize(b.stream));
    }
    const int result[2] = {c->slot_result_host[2 * ticket], c->slot_result_host[2 * ticket + 1]};
    const double *out = c->slot_out_host + (size_t)ticket * 2 * N;
    if (mu) memcpy(mu, out, sizeof(double) * N);
    if (p) memcpy(p, out + N, sizeof(double) * N);
    s.busy = false;
    s.batch = -1;
    --c->slots_busy;
    if (--b.outstanding == 0) b.active = false;
    if (niter) *niter = result[0];
    if (result[1] == FIT_STATUS_BAD_P) return fail(FH_ERR_BAD_P, "Bad value in power spectrum (non-positive or NaN)");
    if (result[1] == FIT_STATUS_NOT_SPD) return fail(FH_ERR_NOT_SPD, "Cholesky of the posterior precision failed");
    return FH_OK;
}

// The packed statistics of the last binning pass to / from the host: what a reduction over ranks that does not go through
// RCCL needs (frank_amd.distributed.HostComm: two processes on ONE device, which RCCL refuses; any torch.distributed backend).
int fh_stats_get_packed(fh_ctx *c, double *sum_stats, int64_t n, double *minmax) {
    double *dsum = nullptr, *dmm = nullptr;
    int64_t len = 0;
    const int rc = fh_stats_device(c, &dsum, &len, &dmm);
    if (rc) return rc;
    if (!sum_stats || !minmax || n != len) return fail(FH_ERR_INVALID, "fh_stats_get_packed: the packed statistics hold %lld doubles", (long long)len);
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemcpyAsync(sum_stats, dsum, sizeof(double) * (size_t)len, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(minmax, dmm, sizeof(double) * 2, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return FH_OK;
}
int fh_stats_set_packed(fh_ctx *c, const double *sum_stats, int64_t n, const double *minmax) {
    double *dsum = nullptr, *dmm = nullptr;
    int64_t len = 0;
    const int rc = fh_stats_device(c, &dsum, &len, &dmm);
    if (rc) return rc;
    if (!sum_stats || !minmax || n != len) return fail(FH_ERR_INVALID, "fh_stats_set_packed: the packed statistics hold %lld doubles", (long long)len);
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemcpyAsync(dsum, sum_stats, sizeof(double) * (size_t)len, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(dmm, minmax, sizeof(double) * 2, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->have_device_Mj = false;
    return FH_OK;
}

int fh_stats_upload(fh_ctx *c, const double *M, const double *j) {
    if (!c || !M || !j) return fail(FH_ERR_INVALID, "fh_stats_upload: NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    const int N = c->N;
    // (pageable host memory: the copies have left the caller's arrays when the calls return)
    HIP_TRY(hipMemcpyAsync(c->M.p, M, sizeof(double) * (size_t)N * N, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->j.p, j, sizeof(double) * N, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->have_device_Mj = true;
    return FH_OK;
}

// Posterior extras of a sweep, batched on the device (evidence.hip): see include/frank_hip.h
int fh_sweep_evidence(fh_ctx *c, const double *M, const double *j, double H0, int batch, const double *p, const double *mu,
                      const double *alpha, const double *p0, const double *wsmooth, double *sol_log_likelihood, double *log_prior,
                      double *log_evidence, double *pscov_diag) {
    if (!c || batch < 1 || !p || !mu || !alpha || !p0 || !wsmooth) return fail(FH_ERR_INVALID, "fh_sweep_evidence: bad argument");
    if ((M == nullptr) != (j == nullptr)) return fail(FH_ERR_INVALID, "pass both M and j or neither");
    if (!M && !c->have_device_Mj) return fail(FH_ERR_INVALID, "no device-resident M, j (run fh_stats_finalize)");
    if (c->N > FIT_MAX_N) return fail(FH_ERR_UNSUPPORTED, "N = %d > %d", c->N, FIT_MAX_N);
    HIP_TRY(hipSetDevice(c->device));
    const int N = c->N;
    const size_t NN = (size_t)N * N;
    for (size_t i = 0; i < (size_t)batch * N; ++i)
        if (!(p[i] > 0.0)) return fail(FH_ERR_BAD_P, "Bad value in power spectrum (point %d)", (int)(i / N));
    SyncOnExit guard{c->stream};
    if (M) {
        HIP_TRY(hipMemcpyAsync(c->M.p, M, sizeof(double) * NN, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(c->j.p, j, sizeof(double) * N, hipMemcpyHostToDevice, c->stream));
        c->have_device_Mj = false;
    }
    std::vector<double> jh(N);
    HIP_TRY(hipMemcpyAsync(jh.data(), c->j.p, sizeof(double) * N, hipMemcpyDeviceToHost, c->stream));
    int rc = prepare_qspace(c, c->Aq.p, c->bq.p);  // Araw = Y^-T M Y^-1 (row-major)
    if (rc) return rc;
    std::vector<double> band;
    smoothing_bands(*c->dht, band);
    const int CH = batch < 128 ? batch : 128;
    DevBuf<double> Cb, Hb, pb, mub, mqb, p0b, wsb, bandb, ldC, ldH, dg;
    DevBuf<int> info;
    if (Cb.alloc((size_t)CH * NN) != hipSuccess || Hb.alloc((size_t)CH * NN) != hipSuccess || pb.alloc((size_t)CH * N) != hipSuccess ||
        mub.alloc((size_t)CH * N) != hipSuccess || mqb.alloc((size_t)CH * N) != hipSuccess || p0b.alloc(CH) != hipSuccess ||
        wsb.alloc(CH) != hipSuccess || bandb.alloc(band.size()) != hipSuccess || ldC.alloc(CH) != hipSuccess ||
        ldH.alloc(CH) != hipSuccess || dg.alloc((size_t)CH * N) != hipSuccess || info.alloc(2 * (size_t)CH) != hipSuccess)
        return fail(FH_ERR_NOMEM, "fh_sweep_evidence: device allocation failed");
    HIP_TRY(hipMemcpyAsync(bandb.p, band.data(), sizeof(double) * band.size(), hipMemcpyHostToDevice, c->stream));
    std::vector<double> hC(CH), hH(CH);
    std::vector<int> hinfo(2 * (size_t)CH);
    const double one = 1.0, zero = 0.0;
    for (int first = 0; first < batch; first += CH) {
        const int n = batch - first < CH ? batch - first : CH;
        HIP_TRY(hipMemcpyAsync(pb.p, p + (size_t)first * N, sizeof(double) * (size_t)n * N, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(mub.p, mu + (size_t)first * N, sizeof(double) * (size_t)n * N, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(p0b.p, p0 + first, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(wsb.p, wsmooth + first, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
        // Dqq = C^-1, C = A + diag(1/p): Cholesky (log det C), inverse
        HIP_TRY(fh_evidence_launch_build_c(c->Araw.p, pb.p, N, n, Cb.p, c->stream));
        ROC_TRY(rocsolver_dpotrf_strided_batched(c->blas, rocblas_fill_lower, N, Cb.p, N, (rocblas_stride)NN, info.p, n));
        HIP_TRY(fh_evidence_launch_logdet(Cb.p, N, n, ldC.p, c->stream));
        ROC_TRY(rocsolver_dpotri_strided_batched(c->blas, rocblas_fill_lower, N, Cb.p, N, (rocblas_stride)NN, info.p + CH, n));
        // mq = Y mu for every point: the row-major Y buffer is Y^T in rocBLAS's column-major reading
        ROC_TRY(rocblas_dgemm(c->blas, rocblas_operation_transpose, rocblas_operation_none, N, n, N, &one, c->Y.p, N, mub.p, N, &zero,
                              mqb.p, N));
        HIP_TRY(fh_evidence_launch_hessian(Cb.p, mqb.p, pb.p, p0b.p, wsb.p, bandb.p, N, n, Hb.p, c->stream));
        HIP_TRY(hipMemcpyAsync(hinfo.data(), info.p, sizeof(int) * (size_t)CH, hipMemcpyDeviceToHost, c->stream));
        ROC_TRY(rocsolver_dpotrf_strided_batched(c->blas, rocblas_fill_lower, N, Hb.p, N, (rocblas_stride)NN, info.p, n));
        HIP_TRY(fh_evidence_launch_logdet(Hb.p, N, n, ldH.p, c->stream));
        HIP_TRY(hipMemcpyAsync(hinfo.data() + CH, info.p, sizeof(int) * (size_t)CH, hipMemcpyDeviceToHost, c->stream));
        if (pscov_diag) {
            ROC_TRY(rocsolver_dpotri_strided_batched(c->blas, rocblas_fill_lower, N, Hb.p, N, (rocblas_stride)NN, info.p + CH, n));
            HIP_TRY(fh_evidence_launch_diag(Hb.p, N, n, dg.p, c->stream));
            HIP_TRY(hipMemcpyAsync(pscov_diag + (size_t)first * N, dg.p, sizeof(double) * (size_t)n * N, hipMemcpyDeviceToHost, c->stream));
        }
        HIP_TRY(hipMemcpyAsync(hC.data(), ldC.p, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(hH.data(), ldH.p, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        for (int b = 0; b < n; ++b) {
            const double *pp = p + (size_t)(first + b) * N, *mm = mu + (size_t)(first + b) * N;
            // GaussianModel.log_likelihood (statistical_models.py:836-841): 1/2 j.mu + 1/2 log det(D S^-1) + H0
            double jm = 0.0, slp = 0.0;
            for (int i = 0; i < N; ++i) {
                jm += jh[i] * mm[i];
                slp += log(pp[i]);
            }
            const bool okC = hinfo[b] == 0, okH = hinfo[CH + b] == 0;
            const double sll = okC ? 0.5 * jm + 0.5 * (-slp - hC[b]) + H0 : NAN;
            // CriticalFilter.log_prior (filter.py:253-261)
            double lp = 0.0, quad = 0.0;
            for (int i = 0; i < N; ++i) {
                const double xi = p0[first + b] / pp[i];
                lp -= xi + (alpha[first + b] - 1.0) * log(xi);
                double ti = 0.0;
                for (int d = -2; d <= 2; ++d)
                    if (i + d >= 0 && i + d < N) ti += band[(size_t)(d + 2) * N + i] * log(pp[i + d]);
                quad += log(pp[i]) * ti;
            }
            lp -= 0.5 * wsmooth[first + b] * quad;
            if (sol_log_likelihood) sol_log_likelihood[first + b] = sll;
            if (log_prior) log_prior[first + b] = lp;
            // radial_fitters.py:963-965: log P(p, V) - 1/2 log det(Hessian / 2 pi)
            if (log_evidence) log_evidence[first + b] = (okC && okH) ? lp + sll - 0.5 * (hH[b] - N * log(2.0 * M_PI)) : NAN;
        }
    }
    return FH_OK;
}

// The clock the fit loops ran at.  on != 0 switches the probe on (every fit loop of this context then adds its shader-clock
// cycles, its ticks of the constant 100 MHz wall clock and its passes to three device counters: two clock reads and three
// atomics per FIT); out3 (may be NULL) receives the sums since the last call and resets them.  mean clock = 100 MHz x
// out3[0] / out3[1]; mean pass = out3[1] / 100 / out3[2] us.  A measurement aid: with 240 loops resident the question is
// whether a pass is slower in CYCLES (memory system) or in time only (the device's clock under an fp64 matrix load).
int fh_ctx_loop_clocks(fh_ctx *c, int on, int64_t *out3) {
    if (!c) return fail(FH_ERR_INVALID, "fh_ctx_loop_clocks: NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    if (c->loop_clocks.p && out3) {
        unsigned long long h[3];
        HIP_TRY(hipMemcpy(h, c->loop_clocks.p, sizeof h, hipMemcpyDeviceToHost));
        for (int i = 0; i < 3; ++i) out3[i] = (int64_t)h[i];
        HIP_TRY(hipMemset(c->loop_clocks.p, 0, sizeof h));
    } else if (out3) {
        out3[0] = out3[1] = out3[2] = 0;
    }
    if (on && !c->loop_clocks.p) {
        if (c->loop_clocks.alloc(3) != hipSuccess) return fail(FH_ERR_NOMEM, "device allocation failed");
        HIP_TRY(hipMemset(c->loop_clocks.p, 0, 3 * sizeof(unsigned long long)));
    }
    if (!on && c->loop_clocks.p) c->loop_clocks.release();
    return FH_OK;
}

int fh_fit_cluster_info(fh_ctx *c, int *workgroups, int64_t *fallbacks) {
    if (!c) return fail(FH_ERR_INVALID, "fh_fit_cluster_info: NULL argument");
    if (workgroups) *workgroups = c->last_fit_cluster;
    if (fallbacks) *fallbacks = (int64_t)c->cluster_fallbacks;
    return FH_OK;
}

int fh_update_power_spectrum(fh_ctx *c, const double *M, const double *j, const double *p, double alpha, double p0,
                             double wsmooth, double *mu, double *p_new) {
    if (!c || !M || !j || !p) return fail(FH_ERR_INVALID, "fh_update_power_spectrum: NULL argument");
    if (c->use_rocsolver_loop || c->NP > fh_k2_loop_max_np())
        return update_power_spectrum_rocsolver(c, M, j, p, alpha, p0, wsmooth, mu, p_new);
    if (c->NP > fh_k2_loop_max_np()) return fail(FH_ERR_UNSUPPORTED, "N = %d: the fit_loop kernel covers N <= 1023", c->N);
    HIP_TRY(hipSetDevice(c->device));
    const int N = c->N;
    for (int k = 0; k < N; ++k)
        if (!(p[k] > 0.0)) return fail(FH_ERR_BAD_P, "Bad value in power spectrum (p[%d] = %g)", k, p[k]);
    std::vector<double> lu;
    smoothing_band_lu(*c->dht, wsmooth, lu);
    HIP_TRY(hipMemcpyAsync(c->M.p, M, sizeof(double) * (size_t)N * N, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->j.p, j, sizeof(double) * N, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->p_init.p, p, sizeof(double) * N, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->band_lu.p, lu.data(), sizeof(double) * lu.size(), hipMemcpyHostToDevice, c->stream));
    c->have_device_Mj = false;
    int rc = prepare_qspace(c, c->Aq.p, c->bq.p);
    if (rc) return rc;
    // posterior mean for the given p
    FitLoopParams P = make_loop_params(c, FIT_MODE_SOLVE, alpha, p0, 0.0, 1 << 30);
    P.p_init = c->p_init.p;
    if (mu) {
        HIP_TRY(fh_k2_launch_loop(P, c->stream));
        HIP_TRY(hipMemcpyAsync(mu, c->mu_out.p, sizeof(double) * N, hipMemcpyDeviceToHost, c->stream));
    }
    P.mode = FIT_MODE_STEP;
    HIP_TRY(fh_k2_launch_loop(P, c->stream));
    int result[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(result, c->loop_result.p, sizeof result, hipMemcpyDeviceToHost, c->stream));
    if (p_new) HIP_TRY(hipMemcpyAsync(p_new, c->p_out.p, sizeof(double) * N, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (result[1] == FIT_STATUS_NOT_SPD) return fail(FH_ERR_NOT_SPD, "Cholesky of the posterior precision failed");
    if (result[1] == FIT_STATUS_BAD_P) return fail(FH_ERR_BAD_P, "Bad value in power spectrum after the update");
    return FH_OK;
}

// ---- method='LogNormal' ------------------------------------------------------------------------------------------
static int ln_np(int N) { return 16 * ((N + 15) / 16); }

static int ln_prepare(fh_ctx *c, const double *M, const double *j, LogNormalParams &P) {
    const int N = c->N;
    const size_t NN = (size_t)N * N;
    if (N > 320) return fail(FH_ERR_UNSUPPORTED, "N = %d: the LogNormal kernel covers N <= 320", N);
    if ((M == nullptr) != (j == nullptr)) return fail(FH_ERR_INVALID, "pass both M and j or neither");
    if (!M && !c->have_device_Mj) return fail(FH_ERR_INVALID, "no device-resident M, j (run fh_stats_finalize)");
    HIP_TRY(hipSetDevice(c->device));
    if (M) {
        HIP_TRY(hipMemcpyAsync(c->M.p, M, sizeof(double) * NN, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(c->j.p, j, sizeof(double) * N, hipMemcpyHostToDevice, c->stream));
        c->have_device_Mj = false;
    }
    if (!c->ln_Sinv.p) {
        HIP_TRY(c->ln_Sinv.alloc(NN));
        HIP_TRY(c->ln_H.alloc(NN));
        HIP_TRY(c->ln_LU.alloc(fh_ln_lu_doubles(N, ln_np(N))));
        HIP_TRY(c->ln_Hinv.alloc(NN));
        HIP_TRY(c->ln_s.alloc(N));
        HIP_TRY(c->ln_p.alloc(N));
        HIP_TRY(c->ln_pin.alloc(N));
        HIP_TRY(c->ln_guess.alloc(N));
        HIP_TRY(c->ln_result.alloc(2));
        HIP_TRY(c->ln_stats.alloc(17));
    }
    P = LogNormalParams{};
    P.N = N;
    P.NP = ln_np(N);
    P.fresh_products = c->ln_fresh_products ? 1 : 0;
    {
        const char *e = getenv("FRANK_AMD_LN_PIVOTED");
        P.no_cholesky = (e && e[0] == '1') ? 1 : 0;
    }
    P.max_step = 100000;  // minimizer.py:190
    P.max_hev = 1000;
    P.newton_tol = 1e-7;  // statistical_models.py:1141
    const double norm = 1 / (M_PI * c->dht->Qmax * c->dht->Qmax);
    P.pl_scale = ((2 * M_PI * c->dht->Rmax * c->dht->Rmax) / c->dht->j_nN) / (0.5 * c->dht->j_nN * norm);
    P.M = c->M.p;
    P.j = c->j.p;
    P.Y = c->Y.p;
    P.q = c->q.p;
    P.band_lu = c->band_lu.p;
    P.Sinv = c->ln_Sinv.p;
    P.H = c->ln_H.p;
    P.LU = c->ln_LU.p;
    P.Hinv = c->ln_Hinv.p;
    P.s_out = c->ln_s.p;
    P.p_out = c->ln_p.p;
    P.result = c->ln_result.p;
    P.stats = c->ln_stats.p;
    return FH_OK;
}

static int ln_finish(fh_ctx *c, double *s, double *p, double *Dinv, int64_t *stats, int result[2]) {
    const int N = c->N;
    long long st[9];
    HIP_TRY(hipMemcpyAsync(result, c->ln_result.p, sizeof(int) * 2, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(st, c->ln_stats.p, sizeof st, hipMemcpyDeviceToHost, c->stream));
    if (s) HIP_TRY(hipMemcpyAsync(s, c->ln_s.p, sizeof(double) * N, hipMemcpyDeviceToHost, c->stream));
    if (p) HIP_TRY(hipMemcpyAsync(p, c->ln_p.p, sizeof(double) * N, hipMemcpyDeviceToHost, c->stream));
    if (Dinv) HIP_TRY(hipMemcpyAsync(Dinv, c->ln_H.p, sizeof(double) * (size_t)N * N, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (stats)
        for (int k = 0; k < 9; ++k) stats[k] = st[k];
#ifdef LN_TIMING
    {
        long long cyc[8];
        HIP_TRY(hipMemcpy(cyc, c->ln_stats.p + 9, sizeof cyc, hipMemcpyDeviceToHost));
        fprintf(stderr, "[ln timing, Mcycles] eval %.1f  lu %.1f (pivoted LU: panel %.1f, fallbacks %.6f M, rest %.1f)  solve %.1f  hess %.1f  newton total %.1f\n",
                cyc[0] / 1e6, cyc[1] / 1e6, cyc[5] / 1e6, cyc[6] / 1e6, cyc[7] / 1e6, cyc[2] / 1e6, cyc[3] / 1e6, cyc[4] / 1e6);
    }
#endif
    if (result[1] == LN_STATUS_BAD_P) return fail(FH_ERR_BAD_P, "Bad value in power spectrum (non-positive or NaN)");
    if (result[1] == LN_STATUS_SLOPE) return fail(FH_ERR_NUMERIC, "Round off in slope calculation (LineSearch)");
    if (result[1] == LN_STATUS_CLUSTER)
        return fail(FH_ERR_HIP, "the helper workgroups of the LogNormal cluster did not answer within 2 s (FRANK_AMD_LN_CLUSTER=1 "
                    "runs the fit on one workgroup)");
    return FH_OK;
}

// ---- method='LogNormal' for 320 < N <= 1023: MinimizeNewton / LineSearch on the host, everything else on the device ----------
// (lognormal_wide.hip; minimizer.py:70-283, statistical_models.py:1064-1160.  The persistent kernel ends at N = 320.)
struct LnWide {
    fh_ctx *c;
    LnWideParams P{};
    double *dir_nj = nullptr, *pdir = nullptr;  // -jac (steepest descent), the limited step
    double *d0 = nullptr, *res = nullptr;       // the unrefined Newton direction, its residual
    double reduction = NAN;                     // LineSearch.reduction (None until the first success)
    bool use_inverse = false;
    bool linear = true;                         // S^-1 (x + lam p) = S^-1 x + lam S^-1 p along a search ('linear'); false: multiplied out
    long long nfev = 0, nhess = 0, nstep = 0;
    double scal[8];

    int setup(double s0) {
        const int N = c->N;
        const size_t NN = (size_t)N * N;
        if (!c->lnw_Sinv.p) {
            HIP_TRY(c->lnw_Sinv.alloc(NN));
            HIP_TRY(c->lnw_H.alloc(NN));
            HIP_TRY(c->lnw_Hinv.alloc(NN));
            HIP_TRY(c->lnw_Hc.alloc(NN));
            HIP_TRY(c->lnw_vec.alloc(14 * (size_t)N));
            HIP_TRY(c->lnw_scal.alloc(8));
            HIP_TRY(c->lnw_ipiv.alloc(N));
        }
        FitState st = make_state(c);
        P.N = N;
        P.s0 = s0;
        P.transform_norm = st.transform_norm;
        P.M = c->M.p;
        P.j = c->j.p;
        P.Y = c->Y.p;
        P.Ykm = c->Ykm.p;
        P.q = c->q.p;
        P.mu = c->mu.p;
        P.Sinv = c->lnw_Sinv.p;
        P.W = c->W.p;
        P.p = c->p.p;
        P.p_old = c->p_old.p;
        P.flags = c->flags.p;
        double *v = c->lnw_vec.p;
        P.x = v, P.xn = v + N, P.I = v + 2 * N, P.t1 = v + 3 * N, P.t2 = v + 4 * N, P.fr = v + 5 * N, P.jx = v + 6 * N, P.dx = v + 7 * N;
        dir_nj = v + 8 * N;
        pdir = v + 9 * N;
        P.Sx = v + 10 * N;
        P.Sp = v + 11 * N;
        d0 = v + 12 * N;
        res = v + 13 * N;
        linear = !c->ln_fresh_products;
        use_inverse = env_int("FRANK_AMD_LNW_INVERSE", 1) != 0;  // (0: rocSOLVER's getrs at every step, ~4x slower)
        P.scal = c->lnw_scal.p;
        return FH_OK;
    }
    int read_scal() {
        HIP_TRY(hipMemcpyAsync(scal, P.scal, sizeof scal, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        return FH_OK;
    }
    // H(x + lam dir) (dir = NULL: H(x)); the trial point and its products stay in xn, I, t1, t2
    int fun(const double *dir, double lam, double *f, bool *same, int mode = 0) {
        HIP_TRY(fh_lnw_launch_eval(P, P.x, dir, lam, mode, c->stream));
        int rc = read_scal();
        if (rc) return rc;
        *f = scal[0];
        if (same) *same = scal[1] != 0.0;
        return FH_OK;
    }
    int accept() {  // x <- xn (and its S^-1 x)
        HIP_TRY(hipMemcpyAsync(P.x, P.xn, sizeof(double) * c->N, hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(P.Sx, P.t1, sizeof(double) * c->N, hipMemcpyDeviceToDevice, c->stream));
        return FH_OK;
    }
    // LineSearch.__call__(func, jac, x0, p, f0, root=False) with reduce_step = limit_step (minimizer.py:70-187).
    // failed: 0 / 1; returns FH_ERR_NUMERIC for "Round off in slope calculation" (the reference raises ValueError there).
    // have_first: limit_step(dir) and the trial at lam = 1 were submitted with the step and are in `scal` already
    int line_search(const double *dir, double *f0, int *failed, bool have_first = false) {
        const double armijo = 1e-4, l_min = 0.1;
        const double cost = *f0;
        int rc;
        if (!have_first) {
            HIP_TRY(fh_lnw_launch_limit_step(P, P.x, dir, pdir, c->stream));
            rc = read_scal();
            if (rc) return rc;
        }
        const double delta_f = scal[3];
        if (delta_f > 0) return FH_ERR_NUMERIC;
        double lam = 1.0, cost_save = 0, lam_save = 0;
        bool first_trial = true;
        for (;;) {
            double cost_new;
            bool same;
            if (have_first) {
                cost_new = scal[0];
                same = scal[1] != 0.0;
                have_first = false;
            } else {
                rc = fun(pdir, lam, &cost_new, &same, linear ? (first_trial ? 1 : 2) : 0);
                if (rc) return rc;
            }
            first_trial = false;
            if (same) {  // (the reference tests x_new == x0 before it evaluates: no evaluation counted)
                *failed = 1;
                return FH_OK;
            }
            ++nfev;
            if (cost_new <= (cost + armijo * lam * delta_f)) {
                reduction = lam;
                rc = accept();
                if (rc) return rc;
                *f0 = cost_new;
                *failed = 0;
                return FH_OK;
            }
            double lam_new;
            if (lam == 1.0) {
                lam_new = -0.5 * delta_f / (cost_new - cost - delta_f);
            } else {
                const double r1 = (cost_new - cost - lam * delta_f) / (lam * lam);
                const double r2 = (cost_save - cost - lam_save * delta_f) / (lam_save * lam_save);
                const double a = (r1 - r2) / (lam - lam_save);
                const double b = (lam * r2 - lam_save * r1) / (lam - lam_save);
                if (a == 0) {
                    lam_new = -0.5 * delta_f / b;
                } else {
                    const double d = b * b - 3 * a * delta_f;
                    if (d < 0) lam_new = 0.5 * lam;
                    else if (b <= 0) lam_new = (-b + sqrt(d)) / (3 * a);
                    else lam_new = -1 * delta_f / (b + sqrt(d));
                    lam_new = (lam_new < 0.5 * lam) ? lam_new : 0.5 * lam;  // min(0.5 lam, lam_new)
                }
            }
            if (lam_new != lam_new) lam_new = l_min * lam;
            lam_save = lam;
            cost_save = cost_new;
            lam = (l_min * lam > lam_new) ? l_min * lam : lam_new;  // max(lam_new, l_min lam)
        }
    }
    // MinimizeNewton(fun, jac, hess, x, LineSearch(reduce_step=limit_step), tol) (minimizer.py:190-283); x in P.x.
    // status: 0 converged, 1 no improvement, 2 max steps, 3 max Hessians, 4 slope round-off
    int minimize(double tol, long long max_step, long long max_hev, int *status) {
        const int N = c->N;
        bool need_hess = true;
        nfev = 1, nhess = 0, nstep = 0;
        reduction = NAN;
        double fx;
        int rc = fun(nullptr, 0.0, &fx, nullptr);
        if (rc) return rc;
        // One submission and one read per step on the common path (a frozen Hessian, the first trial accepted): the Jacobian of x --
        // which also carries the convergence measure of the step BEFORE --, the solve, limit_step and the trial at lam = 1 go to
        // the device together; a step that turns out to follow convergence is discarded with its evaluation.  A step that needs a
        // new Hessian reads the measure first (the factorisation must not be counted if the minimiser has already stopped).
        for (nstep = 0; nstep < max_step; ++nstep) {
            // (xn, I, t1, t2 hold the products of x here: the evaluation in front of the loop, or the accepted trial of a step)
            HIP_TRY(fh_lnw_launch_jac(P, c->stream));  // jx, dx = -jx, scal[2] = max |jac| |x|
            if (need_hess) {
                if (nstep > 0) {
                    rc = read_scal();
                    if (rc) return rc;
                    if (scal[2] < tol * (fabs(fx) > 1 ? fabs(fx) : 1)) {
                        *status = 0;
                        --nstep;  // (the step that converged)
                        return FH_OK;
                    }
                }
                if (nhess == max_hev) {
                    *status = 3;
                    return FH_OK;
                }
                HIP_TRY(fh_lnw_launch_hess(P, c->lnw_H.p, c->stream));
                if (use_inverse) HIP_TRY(hipMemcpyAsync(c->lnw_Hc.p, c->lnw_H.p, sizeof(double) * (size_t)N * N, hipMemcpyDeviceToDevice, c->stream));
                ROC_TRY(rocsolver_dgetrf(c->blas, N, N, c->lnw_H.p, N, c->lnw_ipiv.p, c->info.p));  // (symmetric: either major)
                if (use_inverse) {  // lu_solve for hundreds of steps on one factorisation: the inverse once (N columns)
                    HIP_TRY(fh_lnw_launch_identity(c->lnw_Hinv.p, N, c->stream));
                    ROC_TRY(rocsolver_dgetrs(c->blas, rocblas_operation_none, N, N, c->lnw_H.p, N, c->lnw_ipiv.p, c->lnw_Hinv.p, N));
                }
                ++nhess;
            }
            if (use_inverse) {  // dx = H^-1 (-jac) with one step of refinement (P.dx holds -jac)
                HIP_TRY(fh_lnw_launch_matvec(N, c->lnw_Hinv.p, P.dx, 1.0, nullptr, d0, c->stream));
                HIP_TRY(fh_lnw_launch_matvec(N, c->lnw_Hc.p, d0, -1.0, P.dx, res, c->stream));
                HIP_TRY(fh_lnw_launch_matvec(N, c->lnw_Hinv.p, res, 1.0, d0, P.dx, c->stream));
            } else {
                ROC_TRY(rocsolver_dgetrs(c->blas, rocblas_operation_none, N, 1, c->lnw_H.p, N, c->lnw_ipiv.p, P.dx, N));  // lu_solve
            }
            HIP_TRY(fh_lnw_launch_limit_step(P, P.x, P.dx, pdir, c->stream));  // scal[3] = jac . p, scal[4] = jac . dx
            HIP_TRY(fh_lnw_launch_eval(P, P.x, pdir, 1.0, linear ? 1 : 0, c->stream));  // the first trial, speculatively
            rc = read_scal();
            if (rc) return rc;
            if (!need_hess && nstep > 0 && scal[2] < tol * (fabs(fx) > 1 ? fabs(fx) : 1)) {
                *status = 0;
                --nstep;
                return FH_OK;
            }
            int failed = 1;
            if (scal[4] < 0) {
                rc = line_search(P.dx, &fx, &failed, true);
                if (rc == FH_ERR_NUMERIC) {
                    *status = 4;
                    return FH_OK;
                }
                if (rc) return rc;
            }
            if (failed) {
                // steepest descent (minimizer.py:236-244).  x is where it was but the trials have replaced its products:
                // evaluate x again (not one of the reference's evaluations), then jx and dx = -jx
                double fx_again;
                rc = fun(nullptr, 0.0, &fx_again, nullptr);
                if (rc) return rc;
                HIP_TRY(fh_lnw_launch_jac(P, c->stream));
                HIP_TRY(hipMemcpyAsync(dir_nj, P.dx, sizeof(double) * N, hipMemcpyDeviceToDevice, c->stream));
                int failed_descent = 1;
                rc = line_search(dir_nj, &fx, &failed_descent);
                if (rc == FH_ERR_NUMERIC) {
                    *status = 4;
                    return FH_OK;
                }
                if (rc) return rc;
                if (failed_descent) {  // minimizer.py:246-262: ten ever smaller steps along the limited steepest descent
                    HIP_TRY(fh_lnw_launch_limit_step(P, P.x, dir_nj, pdir, c->stream));
                    double scale = 1.0, fn = fx;
                    bool improved = false;
                    for (int it = 0; it < 10; ++it) {
                        rc = fun(pdir, scale, &fn, nullptr);
                        if (rc) return rc;
                        ++nfev;
                        if (fn < fx) {
                            improved = true;
                            break;
                        }
                        scale *= 0.0625;
                    }
                    if (!improved) {
                        *status = 1;
                        return FH_OK;
                    }
                    fx = fn;
                    rc = accept();
                    if (rc) return rc;
                } else {
                    // (the accepted trial's products are those of the new x)
                }
            }
            need_hess = failed || (reduction != 1.0);
            if (failed) {  // the slow paths may have left another point's products behind: those of the new x again
                double fx_again;
                rc = fun(nullptr, 0.0, &fx_again, nullptr);
                if (rc) return rc;
            }
        }
        // the measure of the last step
        HIP_TRY(fh_lnw_launch_jac(P, c->stream));
        rc = read_scal();
        if (rc) return rc;
        if (scal[2] < tol * (fabs(fx) > 1 ? fabs(fx) : 1)) {
            *status = 0;
            nstep = max_step - 1;
            return FH_OK;
        }
        *status = 2;
        nstep = max_step - 1;  // (python: the loop variable after exhaustion)
        return FH_OK;
    }
    // LogNormalMAPModel(DHT, M, j, p, guess, s0): p in c->p, the guess in P.x; MAP -> P.x, hess(MAP) -> c->lnw_H.
    // totals: [0] solves, [1] steps, [2] evaluations, [3] Hessians, [4 + status] exits
    int map(long long totals[9]) {
        const int N = c->N;
        const double one = 1.0, zero = 0.0;
        HIP_TRY(fh_lnw_launch_scale(P, c->stream));
        // S^-1 = Y^T diag(1/p) Y: column-major views of the row-major buffers are the transposes; S^-1 is symmetric
        ROC_TRY(rocblas_dgemm(c->blas, rocblas_operation_none, rocblas_operation_transpose, N, N, N, &one, c->Y.p, N, c->W.p, N,
                              &zero, c->lnw_Sinv.p, N));
        int status = 2;
        int rc = minimize(1e-7, 100000, 1000, &status);  // statistical_models.py:1141, minimizer.py:190
        if (rc) return rc;
        ++totals[0];
        totals[1] += nstep;
        totals[2] += nfev;
        totals[3] += nhess;
        ++totals[4 + status];
        if (status == 4) return fail(FH_ERR_NUMERIC, "Round off in slope calculation (LineSearch)");
        double f;
        rc = fun(nullptr, 0.0, &f, nullptr);
        if (rc) return rc;
        HIP_TRY(fh_lnw_launch_hess(P, c->lnw_H.p, c->stream));  // Dinv = hess(s_MAP), statistical_models.py:1147
        return FH_OK;
    }
};

static int ln_wide_ready(fh_ctx *c, const double *M, const double *j) {
    const int N = c->N;
    if (N > FIT_MAX_N - 1) return fail(FH_ERR_UNSUPPORTED, "N = %d > %d", N, FIT_MAX_N - 1);
    if ((M == nullptr) != (j == nullptr)) return fail(FH_ERR_INVALID, "pass both M and j or neither");
    if (!M && !c->have_device_Mj) return fail(FH_ERR_INVALID, "no device-resident M, j (run fh_stats_finalize)");
    HIP_TRY(hipSetDevice(c->device));
    if (M) {
        HIP_TRY(hipMemcpyAsync(c->M.p, M, sizeof(double) * (size_t)N * N, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(c->j.p, j, sizeof(double) * N, hipMemcpyHostToDevice, c->stream));
        c->have_device_Mj = false;
    }
    return FH_OK;
}

// CriticalFilter.update_power_spectrum(fit) for a posterior (map in c->mu, precision in c->D) on the library loop's kernels:
// Cholesky of the precision, Tr2 from the triangular solve of Y^T, fit_update_kernel (filter.py:154-177)
static int ln_wide_factor_for_update(fh_ctx *c) {
    const int N = c->N;
    const double one = 1.0;
    HIP_TRY(hipMemcpyAsync(c->Z.p, c->Y.p, sizeof(double) * (size_t)N * N, hipMemcpyDeviceToDevice, c->stream));
    ROC_TRY(rocsolver_dpotrf(c->blas, rocblas_fill_lower, N, c->D.p, N, c->info.p));
    ROC_TRY(rocblas_dtrsm(c->blas, rocblas_side_left, rocblas_fill_lower, rocblas_operation_none, rocblas_diagonal_non_unit, N, N,
                          &one, c->D.p, N, c->Z.p, N));
    return FH_OK;
}

static int lognormal_model_wide(fh_ctx *c, const double *M, const double *j, const double *p, const double *guess, double s0,
                                double *s_map, double *Dinv, int64_t *stats) {
    int rc = ln_wide_ready(c, M, j);
    if (rc) return rc;
    const int N = c->N;
    LnWide w{c};
    rc = w.setup(s0);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(c->p.p, p, sizeof(double) * N, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(w.P.x, guess, sizeof(double) * N, hipMemcpyHostToDevice, c->stream));
    long long totals[9] = {0};
    rc = w.map(totals);
    if (stats)
        for (int k = 0; k < 9; ++k) stats[k] = totals[k];
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(s_map, w.P.x, sizeof(double) * N, hipMemcpyDeviceToHost, c->stream));
    if (Dinv) HIP_TRY(hipMemcpyAsync(Dinv, c->lnw_H.p, sizeof(double) * (size_t)N * N, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return FH_OK;
}

static int fit_lognormal_wide(fh_ctx *c, const double *M, const double *j, double alpha, double p0, double wsmooth, double tol,
                              int max_iter, double I_scale, double *s_map, double *p, int *niter, double *Dinv, int64_t *stats,
                              double *diag_p, double *diag_s) {
    int rc = ln_wide_ready(c, M, j);
    if (rc) return rc;
    const int N = c->N;
    const size_t NN = (size_t)N * N;
    std::vector<double> lu;
    smoothing_band_lu(*c->dht, wsmooth, lu);
    HIP_TRY(hipMemcpyAsync(c->band_lu.p, lu.data(), sizeof(double) * lu.size(), hipMemcpyHostToDevice, c->stream));
    const bool want_diag = diag_p != nullptr;
    if (want_diag) {
        const size_t need = (size_t)(max_iter + 1) * N;
        if (c->diag_p.n < need) HIP_TRY(c->diag_p.alloc(need));
        if (c->lnw_diag_s.n < need) HIP_TRY(c->lnw_diag_s.alloc(need));
    }
    FitState st = make_state(c);
    st.alpha = alpha;
    st.p0 = p0;
    st.tol = tol;
    st.max_iter = max_iter;
    st.diag_p = want_diag ? c->diag_p.p : nullptr;
    st.diag_mu = nullptr;
    // radial_fitters.py:744-752: p = 1 -> Normal fit -> power-law guess -> Normal fit (the library loop's kernels)
    HIP_TRY(fh_k2_launch_init(st, c->stream));
    rc = solve_posterior(c, st, true, false);
    if (rc) return rc;
    HIP_TRY(fh_k2_launch_powerlaw(st, c->stream));
    rc = solve_posterior(c, st, true, false);
    if (rc) return rc;
    int flags[FIT_NFLAGS] = {0}, info = 0;
    HIP_TRY(hipMemcpyAsync(flags, c->flags.p, sizeof flags, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(&info, c->info.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (flags[FIT_FLAG_NOT_SPD] || info != 0)
        return fail(FH_ERR_NOT_SPD, "Cholesky of a seed fit failed (the reference would switch to an SVD)");
    LnWide w{c};
    rc = w.setup(log(I_scale));  // radial_fitters.py:712
    if (rc) return rc;
    HIP_TRY(fh_lnw_launch_seed(w.P, c->stream));  // :756-768
    long long totals[9] = {0};
    rc = w.map(totals);
    int count = 0;
    while (rc == FH_OK) {  // `while not converged and count <= max_iter` (:769-785); the update kernel holds the condition
        HIP_TRY(hipMemcpyAsync(c->D.p, c->lnw_H.p, sizeof(double) * NN, hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(c->mu.p, w.P.x, sizeof(double) * N, hipMemcpyDeviceToDevice, c->stream));
        rc = ln_wide_factor_for_update(c);
        if (rc) break;
        HIP_TRY(fh_k2_launch_update(st, c->stream));
        HIP_TRY(hipMemcpyAsync(flags, c->flags.p, sizeof flags, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        count = flags[FIT_FLAG_COUNT];
        if (flags[FIT_FLAG_NOT_SPD]) {
            rc = fail(FH_ERR_NOT_SPD, "Cholesky of the Hessian at the MAP failed at iteration %d (the reference would switch to an SVD)", count);
            break;
        }
        if (flags[FIT_FLAG_DONE]) break;
        rc = w.map(totals);
        if (rc == FH_OK && want_diag)
            HIP_TRY(hipMemcpyAsync(c->lnw_diag_s.p + (size_t)(count - 1) * N, w.P.x, sizeof(double) * N, hipMemcpyDeviceToDevice, c->stream));
    }
    if (stats)
        for (int k = 0; k < 9; ++k) stats[k] = totals[k];
    *niter = count;
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(flags, c->flags.p, sizeof flags, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(s_map, w.P.x, sizeof(double) * N, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(p, c->p.p, sizeof(double) * N, hipMemcpyDeviceToHost, c->stream));
    if (Dinv) HIP_TRY(hipMemcpyAsync(Dinv, c->lnw_H.p, sizeof(double) * NN, hipMemcpyDeviceToHost, c->stream));
    const size_t nd = (size_t)count * N;
    if (want_diag && nd) {
        HIP_TRY(hipMemcpyAsync(diag_p, c->diag_p.p, sizeof(double) * nd, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(diag_s, c->lnw_diag_s.p, sizeof(double) * nd, hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (flags[FIT_FLAG_BAD_P]) return fail(FH_ERR_BAD_P, "Bad value in power spectrum (non-positive or NaN)");
    return FH_OK;
}

int fh_lognormal_model(fh_ctx *c, const double *M, const double *j, const double *p, const double *guess, double s0,
                       double *s_map, double *Dinv, int64_t *stats) {
    if (!c || !p || !guess || !s_map) return fail(FH_ERR_INVALID, "fh_lognormal_model: NULL argument");
    for (int k = 0; k < c->N; ++k)
        if (!(p[k] > 0.0)) return fail(FH_ERR_BAD_P, "Bad value in power spectrum (p[%d] = %g)", k, p[k]);
    if (c->N > 320) return lognormal_model_wide(c, M, j, p, guess, s0, s_map, Dinv, stats);  // (the host-driven route)
    LogNormalParams P;
    int rc = ln_prepare(c, M, j, P);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(c->ln_pin.p, p, sizeof(double) * c->N, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->ln_guess.p, guess, sizeof(double) * c->N, hipMemcpyHostToDevice, c->stream));
    P.mode = LN_MODE_MAP;
    P.s0 = s0;
    P.p_in = c->ln_pin.p;
    P.guess = c->ln_guess.p;
    HIP_TRY(fh_ln_launch(P, 1, c->stream));
    int result[2];
    return ln_finish(c, s_map, nullptr, Dinv, stats, result);
}

int fh_fit_lognormal(fh_ctx *c, const double *M, const double *j, double alpha, double p0, double wsmooth, double tol,
                     int max_iter, double I_scale, double *s_map, double *p, int *niter, double *Dinv, int64_t *stats,
                     double *diag_p, double *diag_s) {
    if (!c || !s_map || !p || !niter) return fail(FH_ERR_INVALID, "fh_fit_lognormal: NULL argument");
    if (max_iter < 0) return fail(FH_ERR_INVALID, "max_iter must be >= 0");
    if (!(I_scale > 0)) return fail(FH_ERR_INVALID, "I_scale must be positive");
    if ((diag_p == nullptr) != (diag_s == nullptr)) return fail(FH_ERR_INVALID, "pass both diag_p and diag_s or neither");
    if (c->N > 320)  // (beyond the persistent kernel: the host-driven route, lognormal_wide.hip)
        return fit_lognormal_wide(c, M, j, alpha, p0, wsmooth, tol, max_iter, I_scale, s_map, p, niter, Dinv, stats, diag_p, diag_s);
    LogNormalParams P;
    int rc = ln_prepare(c, M, j, P);
    if (rc) return rc;
    const int N = c->N;
    std::vector<double> lu;
    smoothing_band_lu(*c->dht, wsmooth, lu);
    HIP_TRY(hipMemcpyAsync(c->band_lu.p, lu.data(), sizeof(double) * lu.size(), hipMemcpyHostToDevice, c->stream));
    // radial_fitters.py:744-752: the two Normal seed fits (p = 1, then the power-law guess); max_iter = -1 stops the
    // fit_loop kernel after them
    rc = prepare_qspace(c, c->Aq.p, c->bq.p);
    if (rc) return rc;
    FitLoopParams L = make_loop_params(c, FIT_MODE_FULL, alpha, p0, tol, -1);
    HIP_TRY(fh_k2_launch_loop(L, c->stream));
    int seed[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(seed, c->loop_result.p, sizeof seed, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (seed[1] == FIT_STATUS_NOT_SPD) return fail(FH_ERR_NOT_SPD, "Cholesky of the seed fit failed");
    if (seed[1] == FIT_STATUS_BAD_P) return fail(FH_ERR_BAD_P, "Bad value in the seed power spectrum");
    if (diag_p) {
        const size_t need = (size_t)(max_iter + 1) * N;
        if (c->ln_diag_p.n < need) HIP_TRY(c->ln_diag_p.alloc(need));
        if (c->ln_diag_s.n < need) HIP_TRY(c->ln_diag_s.alloc(need));
        P.diag_p = c->ln_diag_p.p;
        P.diag_s = c->ln_diag_s.p;
    }
    P.mode = LN_MODE_FIT;
    P.max_iter = max_iter;
    P.alpha = alpha;
    P.p0 = p0;
    P.tol = tol;
    P.s0 = log(I_scale);  // radial_fitters.py:712
    P.guess = c->mu_out.p;
    {   // a cluster of workgroups for the parallel pieces of a pass (lognormal.hip): FRANK_AMD_LN_CLUSTER workgroups (default 8
        // -- one XCD's worth of workgroup ids 0, 8, .., 56 -- from N = 160 on, where S^-1 and the Tr2 solve are worth a
        // hand-over; 1 = off.  Full size: 0.66 s alone, 0.495 with four, 0.469 with eight)
        int cl = env_int("FRANK_AMD_LN_CLUSTER", N >= 160 ? 8 : 1);
        cl = cl < 1 ? 1 : (cl > 8 ? 8 : cl);
        if (cl > 1) {
            const size_t nv = (size_t)2 * N + P.NP;
            if (!c->ln_ctl.p) HIP_TRY(c->ln_ctl.alloc(8));
            if (c->ln_cluster_vecs.n < nv) HIP_TRY(c->ln_cluster_vecs.alloc(nv));
            HIP_TRY(hipMemsetAsync(c->ln_ctl.p, 0, 8 * sizeof(int), c->stream));
            P.cluster = cl;
            P.ctl = c->ln_ctl.p;
            P.rk_g = c->ln_cluster_vecs.p;
            P.tr2_g = c->ln_cluster_vecs.p + N;
            P.dvec_g = c->ln_cluster_vecs.p + 2 * N;
        }
    }
    HIP_TRY(fh_ln_launch(P, 1, c->stream));
    int result[2];
    rc = ln_finish(c, s_map, p, Dinv, stats, result);
    *niter = result[0];
    if (rc) return rc;
    const size_t nd = (size_t)result[0] * N;
    if (diag_p && nd) {
        HIP_TRY(hipMemcpy(diag_p, c->ln_diag_p.p, sizeof(double) * nd, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(diag_s, c->ln_diag_s.p, sizeof(double) * nd, hipMemcpyDeviceToHost));
    }
    return FH_OK;
}

int fh_posterior_update(fh_ctx *c, const double *map, const double *Dinv, const double *p, double alpha, double p0,
                        double wsmooth, double *p_new) {
    if (!c || !map || !Dinv || !p || !p_new) return fail(FH_ERR_INVALID, "fh_posterior_update: NULL argument");
    const int N = c->N;
    for (int k = 0; k < N; ++k)
        if (!(p[k] > 0.0)) return fail(FH_ERR_BAD_P, "Bad value in power spectrum (p[%d] = %g)", k, p[k]);
    if (N > 320) {  // beyond the persistent kernel: the library loop's kernels (Cholesky of the precision; filter.py:154-177)
        HIP_TRY(hipSetDevice(c->device));
        std::vector<double> luw;
        smoothing_band_lu(*c->dht, wsmooth, luw);
        HIP_TRY(hipMemcpyAsync(c->band_lu.p, luw.data(), sizeof(double) * luw.size(), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(c->p.p, p, sizeof(double) * N, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemsetAsync(c->p_old.p, 0, sizeof(double) * N, c->stream));  // (|p - 0| <= tol p fails: the kernel updates)
        HIP_TRY(hipMemsetAsync(c->flags.p, 0, sizeof(int) * FIT_NFLAGS, c->stream));
        HIP_TRY(hipMemcpyAsync(c->mu.p, map, sizeof(double) * N, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(c->D.p, Dinv, sizeof(double) * (size_t)N * N, hipMemcpyHostToDevice, c->stream));
        int rcw = ln_wide_factor_for_update(c);
        if (rcw) return rcw;
        FitState st = make_state(c);
        st.alpha = alpha;
        st.p0 = p0;
        st.tol = 0.0;
        st.max_iter = 1 << 30;
        HIP_TRY(fh_k2_launch_update(st, c->stream));
        int flags[FIT_NFLAGS];
        HIP_TRY(hipMemcpyAsync(flags, c->flags.p, sizeof flags, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(p_new, c->p.p, sizeof(double) * N, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (flags[FIT_FLAG_NOT_SPD])
            return fail(FH_ERR_NOT_SPD, "Cholesky of the posterior precision failed (the reference would switch to an SVD)");
        return FH_OK;
    }
    LogNormalParams P;
    const bool keep = c->have_device_Mj;
    c->have_device_Mj = true;  // M, j are not touched by this mode
    int rc = ln_prepare(c, nullptr, nullptr, P);
    c->have_device_Mj = keep;
    if (rc) return rc;
    std::vector<double> lu;
    smoothing_band_lu(*c->dht, wsmooth, lu);
    HIP_TRY(hipMemcpyAsync(c->band_lu.p, lu.data(), sizeof(double) * lu.size(), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->ln_pin.p, p, sizeof(double) * N, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->ln_guess.p, map, sizeof(double) * N, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->ln_H.p, Dinv, sizeof(double) * (size_t)N * N, hipMemcpyHostToDevice, c->stream));
    P.mode = LN_MODE_UPDATE;
    P.alpha = alpha;
    P.p0 = p0;
    P.p_in = c->ln_pin.p;
    P.guess = c->ln_guess.p;
    HIP_TRY(fh_ln_launch(P, 1, c->stream));
    int result[2];
    return ln_finish(c, nullptr, p_new, nullptr, nullptr, result);
}

int fh_fit_lognormal_batched(fh_ctx *c, const double *M, const double *j, int batch, const double *alpha,
                             const double *p0, const double *wsmooth, double tol, int max_iter, double I_scale,
                             double *s_map, double *p, int *niter, int *status, int64_t *stats) {
    if (!c || !alpha || !p0 || !wsmooth || !s_map || !p || !niter || batch < 1)
        return fail(FH_ERR_INVALID, "fh_fit_lognormal_batched: bad argument");
    if (max_iter < 0) return fail(FH_ERR_INVALID, "max_iter must be >= 0");
    if (!(I_scale > 0)) return fail(FH_ERR_INVALID, "I_scale must be positive");
    if (c->N > 320) {  // beyond the persistent kernel: the host-driven route, one point after the other
        for (int b = 0; b < batch; ++b) {
            const int rcb = fit_lognormal_wide(c, b == 0 ? M : nullptr, b == 0 ? j : nullptr, alpha[b], p0[b], wsmooth[b], tol, max_iter,
                                               I_scale, s_map + (size_t)b * c->N, p + (size_t)b * c->N, niter + b, nullptr,
                                               stats ? stats + 9 * (size_t)b : nullptr, nullptr, nullptr);
            if (b == 0 && M) c->have_device_Mj = true;  // (uploaded by the first point)
            if (status) status[b] = rcb == FH_ERR_BAD_P || rcb == FH_ERR_NUMERIC ? rcb : FH_OK;
            if (rcb != FH_OK && rcb != FH_ERR_BAD_P && rcb != FH_ERR_NUMERIC) {
                if (M) c->have_device_Mj = false;
                return rcb;
            }
        }
        if (M) c->have_device_Mj = false;
        return FH_OK;
    }
    LogNormalParams P;
    int rc = ln_prepare(c, M, j, P);
    if (rc) return rc;
    const int N = c->N;
    const size_t NN = (size_t)N * N, B = (size_t)batch;
    const size_t G = (size_t)(batch < c->num_cu ? batch : c->num_cu);
    // the seed fits do not depend on the hyper-parameters (radial_fitters.py:744-752): once for the whole sweep
    rc = prepare_qspace(c, c->Aq.p, c->bq.p);
    if (rc) return rc;
    FitLoopParams L = make_loop_params(c, FIT_MODE_FULL, 1.05, 1e-15, tol, -1);
    HIP_TRY(fh_k2_launch_loop(L, c->stream));
    int seed[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(seed, c->loop_result.p, sizeof seed, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (seed[1] == FIT_STATUS_NOT_SPD) return fail(FH_ERR_NOT_SPD, "Cholesky of the seed fit failed");
    if (seed[1] == FIT_STATUS_BAD_P) return fail(FH_ERR_BAD_P, "Bad value in the seed power spectrum");
    DevBuf<double> Sb, LUb, Hib, Hb, sb, pb, lub, alb, p0b;
    DevBuf<int> resb, counter;
    DevBuf<long long> stb;
    if (Sb.alloc(G * NN) != hipSuccess || LUb.alloc(G * fh_ln_lu_doubles(N, ln_np(N))) != hipSuccess || Hib.alloc(G * NN) != hipSuccess ||
        Hb.alloc(B * NN) != hipSuccess || sb.alloc(B * N) != hipSuccess || pb.alloc(B * N) != hipSuccess ||
        lub.alloc(B * 5 * N) != hipSuccess || alb.alloc(B) != hipSuccess || p0b.alloc(B) != hipSuccess ||
        resb.alloc(2 * B) != hipSuccess || counter.alloc(1) != hipSuccess || stb.alloc(17 * B) != hipSuccess)
        return fail(FH_ERR_NOMEM, "fh_fit_lognormal_batched: device allocation for %d fits failed", batch);
    HIP_TRY(hipMemsetAsync(counter.p, 0, sizeof(int), c->stream));
    // The workgroups pull the fits in launch order and the launch ends with its slowest fit: as in fh_fit_normal_batched the
    // points most likely to run to max_iter -- alpha next to 1 (filter.py:172), then the weaker smoothing prior -- go first.
    // order[k] = the caller's index of the fit launched k-th; the outputs are put back in the caller's order.
    const std::vector<int> order = sweep_launch_order(alpha, wsmooth, batch);
    std::vector<double> lu_all(B * 5 * N), lu, al_o(B), p0_o(B);
    for (int k = 0; k < batch; ++k) {
        smoothing_band_lu(*c->dht, wsmooth[order[k]], lu);
        memcpy(lu_all.data() + (size_t)k * 5 * N, lu.data(), sizeof(double) * 5 * N);
        al_o[k] = alpha[order[k]];
        p0_o[k] = p0[order[k]];
    }
    HIP_TRY(hipMemcpyAsync(lub.p, lu_all.data(), sizeof(double) * lu_all.size(), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(alb.p, al_o.data(), sizeof(double) * B, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(p0b.p, p0_o.data(), sizeof(double) * B, hipMemcpyHostToDevice, c->stream));
    P.mode = LN_MODE_FIT;
    P.max_iter = max_iter;
    P.tol = tol;
    P.s0 = log(I_scale);
    P.guess = c->mu_out.p;
    P.band_lu = lub.p;
    P.Sinv = Sb.p;
    P.LU = LUb.p;
    P.Hinv = Hib.p;
    P.H = Hb.p;
    P.s_out = sb.p;
    P.p_out = pb.p;
    P.result = resb.p;
    P.stats = stb.p;
    P.batch = batch;
    P.batch_counter = counter.p;
    P.batch_alpha = alb.p;
    P.batch_p0 = p0b.p;
    HIP_TRY(fh_ln_launch(P, (int)G, c->stream));
    std::vector<int> res(2 * B);
    std::vector<long long> st(17 * B);
    HIP_TRY(hipMemcpyAsync(res.data(), resb.p, sizeof(int) * 2 * B, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(st.data(), stb.p, sizeof(long long) * 17 * B, hipMemcpyDeviceToHost, c->stream));
    for (int k = 0; k < batch; ++k) {  // launch order -> the caller's
        HIP_TRY(hipMemcpyAsync(s_map + (size_t)order[k] * N, sb.p + (size_t)k * N, sizeof(double) * N, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(p + (size_t)order[k] * N, pb.p + (size_t)k * N, sizeof(double) * N, hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int k = 0; k < batch; ++k) {
        const int b = order[k];
        niter[b] = res[2 * k];
        if (status)
            status[b] = res[2 * k + 1] == LN_STATUS_BAD_P ? FH_ERR_BAD_P
                        : res[2 * k + 1] == LN_STATUS_SLOPE ? FH_ERR_NUMERIC : FH_OK;
        if (stats)
            for (int q = 0; q < 9; ++q) stats[9 * b + q] = st[17 * (size_t)k + q];
    }
    return FH_OK;
}

// ---- utilities.UVDataBinner -------------------------------------------------------------------------------------
struct fh_uvbin {
    int device = 0, num_cu = 0;
    int64_t n = 0;
    int nbins = 0, is_complex = 0;
    double bin_width = 0;
    DevBuf<double> uv, Vre, Vim, w;
    std::vector<double> b_uv, b_Vre, b_Vim, b_w, e_re, e_im;
    std::vector<int64_t> b_n;
    float kernel_ms = 0;  // max + sum + error passes of the constructor (HIP events)
};

// sums of w * qty over the bins for device-resident rows (bin_quantities, utilities.py:300-366)
static int uvbin_sums(fh_uvbin *h, const double *d_uv, const double *d_w, const double *const qty[4], int nq, int64_t n,
                      bool count, std::vector<double> &sums, std::vector<int64_t> *counts) {
    const int nb = h->nbins;
    DevBuf<double> ds;
    DevBuf<unsigned long long> dc;
    if (ds.alloc((size_t)nq * nb) != hipSuccess || dc.alloc((size_t)nb) != hipSuccess)
        return fail(FH_ERR_NOMEM, "uvbin: device allocation failed");
    HIP_TRY(hipMemset(ds.p, 0, sizeof(double) * (size_t)nq * nb));
    HIP_TRY(hipMemset(dc.p, 0, sizeof(unsigned long long) * (size_t)nb));
    UvBinParams p{};
    p.uv = d_uv;
    p.w = d_w;
    for (int q = 0; q < 4; ++q) p.qty[q] = q < nq ? qty[q] : nullptr;
    p.nq = nq;
    p.count = count ? 1 : 0;
    p.n = n;
    p.bin_width = h->bin_width;
    p.norm = 1 / h->bin_width;
    p.nbins = nb;
    p.sums = ds.p;
    p.counts = dc.p;
    DevBuf<double> scratch;
    if ((size_t)(nq + 1) * nb * sizeof(double) <= 120 * 1024 &&
        scratch.alloc(fh_uvbin_scratch_doubles(nq + 1, nb, n, h->num_cu)) == hipSuccess)
        p.scratch = scratch.p;
    hipEvent_t s0 = nullptr, s1 = nullptr;
    HIP_TRY(hipEventCreate(&s0));
    HIP_TRY(hipEventCreate(&s1));
    HIP_TRY(hipEventRecord(s0, nullptr));
    HIP_TRY(fh_uvbin_launch_sum(p, h->num_cu, nullptr));
    HIP_TRY(hipEventRecord(s1, nullptr));
    HIP_TRY(hipEventSynchronize(s1));
    float sms = 0;
    HIP_TRY(hipEventElapsedTime(&sms, s0, s1));
    (void)hipEventDestroy(s0);
    (void)hipEventDestroy(s1);
    h->kernel_ms = sms;
    sums.resize((size_t)nq * nb);
    HIP_TRY(hipMemcpy(sums.data(), ds.p, sizeof(double) * sums.size(), hipMemcpyDeviceToHost));
    if (counts) {
        std::vector<unsigned long long> cc((size_t)nb);
        HIP_TRY(hipMemcpy(cc.data(), dc.p, sizeof(unsigned long long) * cc.size(), hipMemcpyDeviceToHost));
        counts->assign(cc.begin(), cc.end());
    }
    return FH_OK;
}

// ---- geometry fits: the residual functions of geometry.py:404-763 on the resident table ---------------------------------
static int residual_scratch(const fh_vis *vis, size_t doubles, double **partial, double **sumsq) {
    const size_t nparts = (size_t)fh_residual_max_blocks() * fh_residual_sums_max();
    const size_t need = doubles + nparts + 64;
    if (vis->resid.n < need) HIP_TRY(vis->resid.alloc(need));
    *partial = vis->resid.p + doubles;
    *sumsq = *partial + nparts;  // (room for the widest row of sums)
    return FH_OK;
}

// The model visibilities of a residual pass through the bucket tables, when the binning pass of exactly these rows under
// exactly this geometry came before (the geometry fits: bin, solve, residuals) -- its baseline range, and with it the tables,
// are then in place.  Anything else (another table or range, the debris model, the first kernel generation) keeps the N Bessel
// evaluations per row.  I_dev: the profile on the device.
static int residual_through_tables(fh_ctx *c, const fh_vis *vis, VisResidualParams &P, const double *I_dev) {
    const double gkey[6] = {P.b.dRA, P.b.dDec, P.b.cos_t, P.b.sin_t, P.b.cos_i, P.b.sin_i};
    const bool known = c->v2 && !c->debris && !vis->use_mult && c->range_valid && c->range_vis == vis->serial &&
                       c->range_first == P.b.first && c->range_count == P.b.count &&
                       memcmp(gkey, c->range_geom, sizeof gkey) == 0 && !getenv("FRANK_AMD_RESIDUAL_DIRECT");
    if (!known) return FH_OK;
    const double smax = c->prepass_qmax_all * P.b.inv_Qmax;
    const int nb = (int)(smax / c->k1_delta) + 2;
    if (!(smax == smax) || nb > 16000 || nb > c->k1_nb_built) return FH_OK;
    if (c->predict_coef.n < (size_t)c->k1_nb_built * FH_K1_TERMS) HIP_TRY(c->predict_coef.alloc((size_t)c->k1_nb_built * FH_K1_TERMS));
    HIP_TRY(fh_k1v2_launch_predict_coef(c->k1_table.p, c->XS, c->N, nb, c->pref_fwd.p, I_dev, P.scale, c->predict_coef.p, c->stream));
    P.coef = c->predict_coef.p;
    P.nb = nb;
    return FH_OK;
}

int fh_vis_residuals(fh_ctx *c, const fh_geometry *g, int vis_model, const fh_vis *vis, int64_t first, int64_t count,
                     const double *I, double *out, double *sumsq) {
    if (!c || !g || !vis || !I) return fail(FH_ERR_INVALID, "fh_vis_residuals: NULL argument");
    if (first < 0 || count < 0 || first + count > vis->n) return fail(FH_ERR_INVALID, "fh_vis_residuals: bad range");
    if (vis->device != c->device) return fail(FH_ERR_INVALID, "visibility table lives on another device");
    if (vis_model != FH_VIS_OPT_THICK && vis_model != FH_VIS_OPT_THIN && vis_model != FH_VIS_DEBRIS)
        return fail(FH_ERR_INVALID, "vis_model must be one of ['opt_thick', 'opt_thin', 'debris']");
    if ((vis_model == FH_VIS_DEBRIS) != c->debris)
        return fail(FH_ERR_INVALID, "vis_model 'debris' goes with fh_ctx_set_scale_height (and only with it)");
    if (count == 0) {
        if (sumsq) *sumsq = 0.0;
        return FH_OK;
    }
    HIP_TRY(hipSetDevice(c->device));
    const int N = c->N;
    VisResidualParams P{};
    table_columns(P.b, vis, first, count);
    P.b.dRA = g->dRA_arcsec * (2. * M_PI / kRadToArcsec);
    P.b.dDec = g->dDec_arcsec * (2. * M_PI / kRadToArcsec);
    const double inc = g->inc_deg * kDegToRad, PA = g->PA_deg * kDegToRad;
    P.b.cos_t = cos(PA);
    P.b.sin_t = sin(PA);
    P.b.cos_i = cos(inc);
    P.b.sin_i = sin(inc);
    P.b.N = N;
    P.b.inv_Qmax = 1. / c->dht->Qmax;
    P.b.zeros = c->zeros.p;
    P.b.j0_table = c->j0_table.p;
    P.b.H2 = c->debris ? c->debris_H2.p : nullptr;
    P.pref = c->pref_fwd.p;
    P.scale = vis_model == FH_VIS_OPT_THICK ? cos(inc) : 1.0;
    double *d_sumsq = nullptr;
    int rc = residual_scratch(vis, out ? 2 * (size_t)count : 0, &P.partial, &d_sumsq);
    if (rc) return rc;
    P.out = out ? vis->resid.p : nullptr;
    if (c->scratch_I.n < (size_t)N + 1) HIP_TRY(c->scratch_I.alloc((size_t)N + 1));
    HIP_TRY(hipMemcpyAsync(c->scratch_I.p, I, sizeof(double) * N, hipMemcpyHostToDevice, c->stream));
    P.I = c->scratch_I.p;
    P.delta = c->k1_delta > 0 ? c->k1_delta : 1.0;
    rc = residual_through_tables(c, vis, P, P.I);
    if (rc) return rc;
    HIP_TRY(fh_launch_vis_residual(P, d_sumsq, c->stream));
    if (out) HIP_TRY(hipMemcpyAsync(out, vis->resid.p, sizeof(double) * 2 * (size_t)count, hipMemcpyDeviceToHost, c->stream));
    double ss = 0.0;
    HIP_TRY(hipMemcpyAsync(&ss, d_sumsq, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (sumsq) *sumsq = ss;
    return FH_OK;
}

int fh_gauss_residuals(const fh_vis *vis, const double *params, int fit_inc_pa, int fit_phase, double *fun, double *jac,
                       double *sumsq) {
    if (!vis || !params) return fail(FH_ERR_INVALID, "fh_gauss_residuals: NULL argument");
    const int64_t n = vis->n;
    if (n == 0) {
        if (sumsq) *sumsq = 0.0;
        return FH_OK;
    }
    HIP_TRY(hipSetDevice(vis->device));
    GaussResidualParams P{};
    table_columns(P.b, vis, 0, n);
    P.fac = 2. * M_PI / kRadToArcsec;
    P.rad_to_arcsec = kRadToArcsec;
    P.b.cos_i = cos(params[0]);
    P.b.sin_i = sin(params[0]);
    P.b.cos_t = cos(params[1]);
    P.b.sin_t = sin(params[1]);
    P.b.dRA = params[2] * P.fac;
    P.b.dDec = params[3] * P.fac;
    P.norm = params[4];
    P.scal = params[5];
    P.fit_inc_pa = fit_inc_pa;
    P.fit_phase = fit_phase;
    const size_t nf = fun ? 2 * (size_t)n : 0, nj = jac ? 12 * (size_t)n : 0;
    double *d_sumsq = nullptr;
    int rc = residual_scratch(vis, nf + nj, &P.partial, &d_sumsq);
    if (rc) return rc;
    P.fun = fun ? vis->resid.p : nullptr;
    P.jac = jac ? vis->resid.p + nf : nullptr;
    hipStream_t st = nullptr;  // (the table has no context: the null stream, synchronous copies)
    HIP_TRY(fh_launch_gauss_residual(P, d_sumsq, st));
    if (fun) HIP_TRY(hipMemcpy(fun, P.fun, sizeof(double) * nf, hipMemcpyDeviceToHost));
    if (jac) HIP_TRY(hipMemcpy(jac, P.jac, sizeof(double) * nj, hipMemcpyDeviceToHost));
    double ss = 0.0;
    HIP_TRY(hipMemcpy(&ss, d_sumsq, sizeof(double), hipMemcpyDeviceToHost));
    if (sumsq) *sumsq = ss;
    return FH_OK;
}


// FrankRadialFit.predict(u, v) (radial_fitters.py:56-98) in one pass on the device: deproject, H(q) I, scale, re-phase -- the
// residual kernel without data.  u, v: n host doubles; Vre, Vim: n host doubles each.
int fh_predict_sky(fh_ctx *c, const fh_geometry *g, int vis_model, const double *u, const double *v, int64_t n, const double *I,
                   double *Vre, double *Vim) {
    if (!c || !g || !I || n < 0 || (n > 0 && (!u || !v || !Vre || !Vim))) return fail(FH_ERR_INVALID, "fh_predict_sky: bad argument");
    if (vis_model != FH_VIS_OPT_THICK && vis_model != FH_VIS_OPT_THIN && vis_model != FH_VIS_DEBRIS)
        return fail(FH_ERR_INVALID, "vis_model must be one of ['opt_thick', 'opt_thin', 'debris']");
    if ((vis_model == FH_VIS_DEBRIS) != c->debris)
        return fail(FH_ERR_INVALID, "vis_model 'debris' goes with fh_ctx_set_scale_height (and only with it)");
    if (n == 0) return FH_OK;
    HIP_TRY(hipSetDevice(c->device));
    const int N = c->N;
    const size_t nn = (size_t)n;
    const size_t nparts = (size_t)fh_residual_max_blocks();
    if (c->scratch_q.n < 2 * nn) HIP_TRY(c->scratch_q.alloc(2 * nn));
    const size_t nscal = (size_t)c->deproject_blocks * 4;  // (a range pass's per-workgroup scalars, kept apart from the binning pass's)
    if (c->scratch_out.n < 2 * nn + nparts + 1 + nscal) HIP_TRY(c->scratch_out.alloc(2 * nn + nparts + 1 + nscal));
    if (c->scratch_I.n < (size_t)N + 1) HIP_TRY(c->scratch_I.alloc((size_t)N + 1));
    HIP_TRY(hipMemcpyAsync(c->scratch_q.p, u, sizeof(double) * nn, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->scratch_q.p + nn, v, sizeof(double) * nn, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->scratch_I.p, I, sizeof(double) * N, hipMemcpyHostToDevice, c->stream));
    VisResidualParams P{};
    P.b.u = c->scratch_q.p;
    P.b.v = c->scratch_q.p + nn;
    P.b.first = 0;
    P.b.count = n;
    P.b.dRA = g->dRA_arcsec * (2. * M_PI / kRadToArcsec);
    P.b.dDec = g->dDec_arcsec * (2. * M_PI / kRadToArcsec);
    const double inc = g->inc_deg * kDegToRad, PA = g->PA_deg * kDegToRad;
    P.b.cos_t = cos(PA);
    P.b.sin_t = sin(PA);
    P.b.cos_i = cos(inc);
    P.b.sin_i = sin(inc);
    P.b.N = N;
    P.b.inv_Qmax = 1. / c->dht->Qmax;
    P.b.zeros = c->zeros.p;
    P.b.j0_table = c->j0_table.p;
    P.b.H2 = c->debris ? c->debris_H2.p : nullptr;
    P.pref = c->pref_fwd.p;
    P.I = c->scratch_I.p;
    P.scale = vis_model == FH_VIS_OPT_THICK ? cos(inc) : 1.0;
    P.delta = 1.0;
    P.predict_only = 1;
    P.out = c->scratch_out.p;
    P.partial = c->scratch_out.p + 2 * nn;
    if (c->v2 && !c->debris && n >= 65536 && !getenv("FRANK_AMD_RESIDUAL_DIRECT")) {
        // large calls: one look at (u, v) for the longest deprojected baseline, then the model visibility of a row is the
        // degree-11 polynomial of its bucket (the binning pass's tables contracted with the profile) instead of N Bessel
        // evaluations -- at N = 300 those are a third of the call
        PrepassParams R{};
        R.bin = P.b;
        R.unroll = 2;
        R.partial_scalars = c->scratch_out.p + 2 * nn + nparts + 1;
        fh_prepass_geometry(0, c->num_cu, &R.wpb, &R.blocks);
        if ((size_t)R.blocks * 4 <= nscal) {
            HIP_TRY(fh_prepass_launch_range(R, c->stream));
            std::vector<double> scal((size_t)R.blocks * 4);
            HIP_TRY(hipMemcpyAsync(scal.data(), R.partial_scalars, sizeof(double) * scal.size(), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            double qmax_all = 0.0;
            for (int b = 0; b < R.blocks; ++b)
                if (scal[(size_t)b * 4 + 3] > qmax_all) qmax_all = scal[(size_t)b * 4 + 3];
            const double smax = qmax_all * P.b.inv_Qmax;
            if (smax == smax && smax / c->k1_delta < 15000.0) {
                const int nb = (int)(smax / c->k1_delta) + 2;
                const int rc = k1v2_ensure_table(c, nb);
                if (rc) return rc;
                if (c->predict_coef.n < (size_t)c->k1_nb_built * FH_K1_TERMS)
                    HIP_TRY(c->predict_coef.alloc((size_t)c->k1_nb_built * FH_K1_TERMS));
                HIP_TRY(fh_k1v2_launch_predict_coef(c->k1_table.p, c->XS, N, nb, c->pref_fwd.p, P.I, P.scale, c->predict_coef.p,
                                                    c->stream));
                P.coef = c->predict_coef.p;
                P.nb = nb;
                P.delta = c->k1_delta;
            }
        }
    }
    HIP_TRY(fh_launch_vis_residual(P, P.partial + nparts, c->stream));
    HIP_TRY(hipMemcpyAsync(Vre, P.out, sizeof(double) * nn, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(Vim, P.out + nn, sizeof(double) * nn, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return FH_OK;
}

// ---- the same fits on the normal equations: residual vectors stay on the device, only J^T J and J^T r come back ----------
int fh_vis_residuals_slot(fh_ctx *c, const fh_geometry *g, int vis_model, const fh_vis *vis, const double *I, int slot,
                          double *sumsq) {
    if (!c || !g || !vis) return fail(FH_ERR_INVALID, "fh_vis_residuals_slot: NULL argument");
    if (slot < 0 || slot >= FH_RESIDUAL_SLOTS) return fail(FH_ERR_INVALID, "fh_vis_residuals_slot: slot %d of %d", slot, FH_RESIDUAL_SLOTS);
    if (vis->device != c->device) return fail(FH_ERR_INVALID, "visibility table lives on another device");
    if (vis_model != FH_VIS_OPT_THICK && vis_model != FH_VIS_OPT_THIN && vis_model != FH_VIS_DEBRIS)
        return fail(FH_ERR_INVALID, "vis_model must be one of ['opt_thick', 'opt_thin', 'debris']");
    if ((vis_model == FH_VIS_DEBRIS) != c->debris)
        return fail(FH_ERR_INVALID, "vis_model 'debris' goes with fh_ctx_set_scale_height (and only with it)");
    const int64_t n = vis->n;
    if (n == 0) return fail(FH_ERR_INVALID, "fh_vis_residuals_slot: empty table");
    HIP_TRY(hipSetDevice(c->device));
    const size_t len = 2 * (size_t)n;
    // the buffer grows with the highest slot asked for (a fit uses 2 + its free parameters, not FH_RESIDUAL_SLOTS: 128 B per
    // visibility were 1.3 GB at 1e7 rows); the vectors already there move with it
    if (vis->slots.n < len * (size_t)(slot + 1)) {
        DevBuf<double> grown;
        HIP_TRY(grown.alloc(len * (size_t)(slot + 1)));
        if (vis->slots.n) HIP_TRY(hipMemcpyAsync(grown.p, vis->slots.p, sizeof(double) * vis->slots.n, hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        std::swap(vis->slots.p, grown.p);
        std::swap(vis->slots.n, grown.n);
        std::swap(vis->slots.owned, grown.owned);
        grown.release();
    }
    const int N = c->N;
    VisResidualParams P{};
    table_columns(P.b, vis, 0, n);
    P.b.dRA = g->dRA_arcsec * (2. * M_PI / kRadToArcsec);
    P.b.dDec = g->dDec_arcsec * (2. * M_PI / kRadToArcsec);
    const double inc = g->inc_deg * kDegToRad, PA = g->PA_deg * kDegToRad;
    P.b.cos_t = cos(PA);
    P.b.sin_t = sin(PA);
    P.b.cos_i = cos(inc);
    P.b.sin_i = sin(inc);
    P.b.N = N;
    P.b.inv_Qmax = 1. / c->dht->Qmax;
    P.b.zeros = c->zeros.p;
    P.b.j0_table = c->j0_table.p;
    P.b.H2 = c->debris ? c->debris_H2.p : nullptr;
    P.pref = c->pref_fwd.p;
    P.scale = vis_model == FH_VIS_OPT_THICK ? cos(inc) : 1.0;
    double *d_sumsq = nullptr;
    int rc = residual_scratch(vis, 0, &P.partial, &d_sumsq);
    if (rc) return rc;
    P.out = vis->slots.p + len * slot;
    if (I) {
        if (c->scratch_I.n < (size_t)N + 1) HIP_TRY(c->scratch_I.alloc((size_t)N + 1));
        HIP_TRY(hipMemcpyAsync(c->scratch_I.p, I, sizeof(double) * N, hipMemcpyHostToDevice, c->stream));
        P.I = c->scratch_I.p;
    } else {
        if (!c->have_device_mu)
            return fail(FH_ERR_INVALID, "fh_vis_residuals_slot: I = NULL, but no solve of this context has left a profile on the device");
        P.I = c->mu.p;  // the profile the last solve of this context left on the device (fh_gaussian_model, fh_fit_*)
    }
    P.delta = c->k1_delta > 0 ? c->k1_delta : 1.0;
    rc = residual_through_tables(c, vis, P, P.I);
    if (rc) return rc;
    HIP_TRY(fh_launch_vis_residual(P, d_sumsq, c->stream));
    double ss = 0.0;
    HIP_TRY(hipMemcpyAsync(&ss, d_sumsq, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (sumsq) *sumsq = ss;
    return FH_OK;
}

int fh_residual_normal_equations(fh_ctx *c, const fh_vis *vis, int base_slot, int ncol, const int *col_slots, const double *h,
                                 double *JtJ, double *Jtr) {
    if (!c || !vis || !col_slots || !h || !JtJ || !Jtr || ncol < 1 || ncol > 4)
        return fail(FH_ERR_INVALID, "fh_residual_normal_equations: bad argument");
    const size_t len = 2 * (size_t)vis->n;
    int top = base_slot;
    for (int k = 0; k < ncol; ++k) top = col_slots[k] > top ? col_slots[k] : top;
    if (len == 0 || top < 0 || vis->slots.n < len * (size_t)(top + 1))
        return fail(FH_ERR_INVALID, "fh_residual_normal_equations: no residual vectors on the device (fh_vis_residuals_slot)");
    HIP_TRY(hipSetDevice(c->device));
    FdNormalParams P{};
    if (base_slot < 0 || base_slot >= FH_RESIDUAL_SLOTS) return fail(FH_ERR_INVALID, "bad slot");
    P.base = vis->slots.p + len * base_slot;
    for (int k = 0; k < 4; ++k) {
        const int sl = k < ncol ? col_slots[k] : base_slot;
        if (sl < 0 || sl >= FH_RESIDUAL_SLOTS) return fail(FH_ERR_INVALID, "bad slot");
        if (k < ncol && !(h[k] != 0.0)) return fail(FH_ERR_INVALID, "fh_residual_normal_equations: zero step");
        P.col[k] = vis->slots.p + len * sl;
        P.inv_h[k] = k < ncol ? 1.0 / h[k] : 0.0;
    }
    P.ncol = ncol;
    P.len = (int64_t)len;
    double *d_out = nullptr;
    int rc = residual_scratch(vis, 0, &P.partial, &d_out);
    if (rc) return rc;
    HIP_TRY(fh_launch_fd_normal(P, d_out, c->stream));
    double out[14];
    HIP_TRY(hipMemcpyAsync(out, d_out, sizeof(out), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    int s = 0;
    for (int k = 0; k < 4; ++k)
        for (int l = k; l < 4; ++l, ++s)
            if (l < ncol) JtJ[k * ncol + l] = JtJ[l * ncol + k] = out[s];
    for (int k = 0; k < ncol; ++k) Jtr[k] = out[10 + k];
    return FH_OK;
}

int fh_gauss_normal_equations(const fh_vis *vis, const double *params, int fit_inc_pa, int fit_phase, double *JtJ, double *Jtr,
                              double *sumsq) {
    if (!vis || !params || !JtJ || !Jtr) return fail(FH_ERR_INVALID, "fh_gauss_normal_equations: NULL argument");
    const int64_t n = vis->n;
    if (n == 0) return fail(FH_ERR_INVALID, "fh_gauss_normal_equations: empty table");
    HIP_TRY(hipSetDevice(vis->device));
    GaussResidualParams P{};
    table_columns(P.b, vis, 0, n);
    P.fac = 2. * M_PI / kRadToArcsec;
    P.rad_to_arcsec = kRadToArcsec;
    P.b.cos_i = cos(params[0]);
    P.b.sin_i = sin(params[0]);
    P.b.cos_t = cos(params[1]);
    P.b.sin_t = sin(params[1]);
    P.b.dRA = params[2] * P.fac;
    P.b.dDec = params[3] * P.fac;
    P.norm = params[4];
    P.scal = params[5];
    P.fit_inc_pa = fit_inc_pa;
    P.fit_phase = fit_phase;
    double *d_out = nullptr;
    int rc = residual_scratch(vis, 0, &P.partial, &d_out);
    if (rc) return rc;
    HIP_TRY(fh_launch_gauss_normal(P, d_out, nullptr));
    double out[28];
    HIP_TRY(hipMemcpy(out, d_out, sizeof(out), hipMemcpyDeviceToHost));
    int s = 0;
    for (int k = 0; k < 6; ++k)
        for (int l = k; l < 6; ++l, ++s) JtJ[k * 6 + l] = JtJ[l * 6 + k] = out[s];
    for (int k = 0; k < 6; ++k) Jtr[k] = out[21 + k];
    if (sumsq) *sumsq = out[27];
    return FH_OK;
}


int fh_uvbin_create(int device, const double *uv, const double *Vre, const double *Vim, const double *w, int64_t n,
                    double bin_width, fh_uvbin **out) {
    if (!out || !uv || !Vre || !w || n < 1 || !(bin_width > 0)) return fail(FH_ERR_INVALID, "fh_uvbin_create: bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(FH_ERR_HIP, "no HIP device available: frank_amd has no CPU fallback for device work");
    HIP_TRY(hipSetDevice(device));
    std::unique_ptr<fh_uvbin> h(new fh_uvbin());
    h->device = device;
    h->n = n;
    h->bin_width = bin_width;
    h->is_complex = Vim ? 1 : 0;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    h->num_cu = prop.multiProcessorCount;
    const size_t nn = (size_t)n, bytes = sizeof(double) * nn;
    if (h->uv.alloc(nn) != hipSuccess || h->Vre.alloc(nn) != hipSuccess || (Vim && h->Vim.alloc(nn) != hipSuccess) ||
        h->w.alloc(nn) != hipSuccess)
        return fail(FH_ERR_NOMEM, "fh_uvbin_create: hipMalloc failed for %lld rows", (long long)n);
    HIP_TRY(hipMemcpy(h->uv.p, uv, bytes, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->Vre.p, Vre, bytes, hipMemcpyHostToDevice));
    if (Vim) HIP_TRY(hipMemcpy(h->Vim.p, Vim, bytes, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->w.p, w, bytes, hipMemcpyHostToDevice));
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr;
    HIP_TRY(hipEventCreate(&ev0));
    HIP_TRY(hipEventCreate(&ev1));
    HIP_TRY(hipEventCreate(&ev2));
    HIP_TRY(hipEventCreate(&ev3));
    struct EvGuard {
        hipEvent_t *e[4];
        ~EvGuard() { for (auto p : e) if (*p) (void)hipEventDestroy(*p); }
    } guard{{&ev0, &ev1, &ev2, &ev3}};
    // nbins = ceil(uv.max() / bin_width), +1 if rounding left the maximum outside (utilities.py:204-208)
    DevBuf<unsigned long long> mx;
    if (mx.alloc(2) != hipSuccess) return fail(FH_ERR_NOMEM, "fh_uvbin_create: hipMalloc failed");
    HIP_TRY(hipMemset(mx.p, 0, 2 * sizeof(unsigned long long)));
    HIP_TRY(hipEventRecord(ev0, nullptr));
    HIP_TRY(fh_uvbin_launch_max(h->uv.p, n, mx.p, h->num_cu, nullptr));
    HIP_TRY(hipEventRecord(ev1, nullptr));
    unsigned long long mxh[2];
    HIP_TRY(hipMemcpy(mxh, mx.p, sizeof mxh, hipMemcpyDeviceToHost));
    if (mxh[1]) return fail(FH_ERR_INVALID, "fh_uvbin_create: baselines must be non-negative and finite");
    double uvmax;
    memcpy(&uvmax, &mxh[0], sizeof uvmax);
    double nbf = ceil(uvmax / bin_width);
    if (nbf * bin_width < uvmax) nbf += 1;
    if (!(nbf >= 1) || nbf > 1e8) return fail(FH_ERR_INVALID, "fh_uvbin_create: %g bins of width %g", nbf, bin_width);
    const int nb = h->nbins = (int)nbf;
    // weighted sums of uv, 1, Re V, Im V + counts, then the means (utilities.py:214-223)
    const double *qty[4] = {h->uv.p, nullptr, h->Vre.p, h->Vim.p};
    std::vector<double> sums;
    int rc = uvbin_sums(h.get(), h->uv.p, h->w.p, qty, Vim ? 4 : 3, n, true, sums, &h->b_n);
    if (rc) return rc;
    h->b_uv.assign(sums.begin(), sums.begin() + nb);
    h->b_w.assign(sums.begin() + nb, sums.begin() + 2 * nb);
    h->b_Vre.assign(sums.begin() + 2 * nb, sums.begin() + 3 * nb);
    h->b_Vim.assign((size_t)nb, 0.0);
    if (Vim) h->b_Vim.assign(sums.begin() + 3 * nb, sums.begin() + 4 * nb);
    for (int b = 0; b < nb; ++b)
        if (h->b_n[b] > 0) {
            h->b_uv[b] /= h->b_w[b];
            if (Vim) {  // complex / real as NumPy does it: both parts divided
                h->b_Vre[b] /= h->b_w[b];
                h->b_Vim[b] /= h->b_w[b];
            } else {
                h->b_Vre[b] /= h->b_w[b];
            }
        }
    // error of the mean (utilities.py:236-263)
    DevBuf<double> mre, mim, es;
    if (mre.alloc((size_t)nb) != hipSuccess || mim.alloc((size_t)nb) != hipSuccess || es.alloc(2 * (size_t)nb) != hipSuccess)
        return fail(FH_ERR_NOMEM, "fh_uvbin_create: hipMalloc failed");
    HIP_TRY(hipMemcpy(mre.p, h->b_Vre.data(), sizeof(double) * nb, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(mim.p, h->b_Vim.data(), sizeof(double) * nb, hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(es.p, 0, sizeof(double) * 2 * nb));
    UvBinParams p{};
    p.uv = h->uv.p;
    p.w = h->w.p;
    p.qty[0] = h->Vre.p;
    p.qty[1] = Vim ? h->Vim.p : nullptr;
    p.n = n;
    p.bin_width = bin_width;
    p.norm = 1 / bin_width;
    p.nbins = nb;
    p.mu_re = mre.p;
    p.mu_im = mim.p;
    p.sums = es.p;
    DevBuf<double> escratch;
    if ((size_t)2 * nb * sizeof(double) <= 120 * 1024 &&
        escratch.alloc(fh_uvbin_scratch_doubles(2, nb, n, h->num_cu)) == hipSuccess)
        p.scratch = escratch.p;
    HIP_TRY(hipEventRecord(ev2, nullptr));
    HIP_TRY(fh_uvbin_launch_err(p, h->num_cu, nullptr));
    HIP_TRY(hipEventRecord(ev3, nullptr));
    std::vector<double> e(2 * (size_t)nb);
    HIP_TRY(hipMemcpy(e.data(), es.p, sizeof(double) * e.size(), hipMemcpyDeviceToHost));
    h->e_re.assign((size_t)nb, NAN);
    h->e_im.assign((size_t)nb, 0.0);
    for (int b = 0; b < nb; ++b)
        if (h->b_n[b] > 1) {
            const double den = h->b_w[b] * h->b_w[b] * (1 - 1 / (double)h->b_n[b]);
            h->e_re[b] = sqrt(e[b] / den);
            if (Vim) h->e_im[b] = sqrt(e[(size_t)nb + b] / den);
        }
    {
        float a = 0, b2 = 0;
        HIP_TRY(hipEventSynchronize(ev3));
        HIP_TRY(hipEventElapsedTime(&a, ev0, ev1));
        HIP_TRY(hipEventElapsedTime(&b2, ev2, ev3));
        h->kernel_ms = a + b2 + h->kernel_ms;  // + the sum pass, timed inside uvbin_sums
    }
    // bins with one row: utilities.py:256-261 assigns to `.real` of a fancy-indexed copy, which leaves np.nan
    // (nan+0j for complex V) in place -- kept, so that results match the reference
    *out = h.release();
    return FH_OK;
}

void fh_uvbin_destroy(fh_uvbin *h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    delete h;
}

int fh_uvbin_nbins(const fh_uvbin *h) { return h ? h->nbins : 0; }
float fh_uvbin_kernel_ms(const fh_uvbin *h) { return h ? h->kernel_ms : 0.0f; }

int fh_uvbin_get(const fh_uvbin *h, double *uv, double *Vre, double *Vim, double *w, int64_t *count, double *err_re,
                 double *err_im) {
    if (!h) return fail(FH_ERR_INVALID, "fh_uvbin_get: NULL handle");
    const size_t nb = (size_t)h->nbins;
    if (uv) memcpy(uv, h->b_uv.data(), sizeof(double) * nb);
    if (Vre) memcpy(Vre, h->b_Vre.data(), sizeof(double) * nb);
    if (Vim) memcpy(Vim, h->b_Vim.data(), sizeof(double) * nb);
    if (w) memcpy(w, h->b_w.data(), sizeof(double) * nb);
    if (count) memcpy(count, h->b_n.data(), sizeof(int64_t) * nb);
    if (err_re) memcpy(err_re, h->e_re.data(), sizeof(double) * nb);
    if (err_im) memcpy(err_im, h->e_im.data(), sizeof(double) * nb);
    return FH_OK;
}

int fh_uvbin_determine(fh_uvbin *h, const double *uv, int64_t n, int32_t *idx) {
    if (!h || (n > 0 && (!uv || !idx)) || n < 0) return fail(FH_ERR_INVALID, "fh_uvbin_determine: bad argument");
    if (n == 0) return FH_OK;
    // the reference indexes bins[idx] before rejecting: baselines at or past (nbins + 1) * bin_width raise IndexError
    for (int64_t i = 0; i < n; ++i)
        if (!(uv[i] >= 0) || floor(uv[i] * (1 / h->bin_width)) > h->nbins)
            return fail(FH_ERR_INVALID, "index %lld is out of bounds: baseline %g beyond the bin edges", (long long)i, uv[i]);
    HIP_TRY(hipSetDevice(h->device));
    DevBuf<double> d;
    DevBuf<int> o;
    if (d.alloc((size_t)n) != hipSuccess || o.alloc((size_t)n) != hipSuccess) return fail(FH_ERR_NOMEM, "hipMalloc failed");
    HIP_TRY(hipMemcpy(d.p, uv, sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
    HIP_TRY(fh_uvbin_launch_lookup(d.p, n, h->bin_width, h->nbins, o.p, h->num_cu, nullptr));
    HIP_TRY(hipMemcpy(idx, o.p, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost));
    return FH_OK;
}

int fh_uvbin_quantities(fh_uvbin *h, const double *uv, const double *w, const double *qre, const double *qim, int64_t n,
                        double *out_re, double *out_im, int64_t *counts) {
    if (!h || !uv || !w || !qre || !out_re || n < 0 || ((qim == nullptr) != (out_im == nullptr)))
        return fail(FH_ERR_INVALID, "fh_uvbin_quantities: bad argument");
    HIP_TRY(hipSetDevice(h->device));
    const size_t nn = (size_t)(n > 0 ? n : 1), bytes = sizeof(double) * (size_t)n;
    DevBuf<double> duv, dw, dre, dim;
    if (duv.alloc(nn) != hipSuccess || dw.alloc(nn) != hipSuccess || dre.alloc(nn) != hipSuccess ||
        (qim && dim.alloc(nn) != hipSuccess))
        return fail(FH_ERR_NOMEM, "hipMalloc failed");
    if (n > 0) {
        for (int64_t i = 0; i < n; ++i)
            if (!(uv[i] >= 0) || floor(uv[i] * (1 / h->bin_width)) > h->nbins)
                return fail(FH_ERR_INVALID, "index out of bounds: baseline %g beyond the bin edges", uv[i]);
        HIP_TRY(hipMemcpy(duv.p, uv, bytes, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(dw.p, w, bytes, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(dre.p, qre, bytes, hipMemcpyHostToDevice));
        if (qim) HIP_TRY(hipMemcpy(dim.p, qim, bytes, hipMemcpyHostToDevice));
    }
    const double *qty[4] = {dre.p, qim ? dim.p : nullptr, nullptr, nullptr};
    std::vector<double> sums;
    std::vector<int64_t> cc;
    int rc = uvbin_sums(h, duv.p, dw.p, qty, qim ? 2 : 1, n, counts != nullptr, sums, counts ? &cc : nullptr);
    if (rc) return rc;
    memcpy(out_re, sums.data(), sizeof(double) * (size_t)h->nbins);
    if (qim) memcpy(out_im, sums.data() + h->nbins, sizeof(double) * (size_t)h->nbins);
    if (counts) memcpy(counts, cc.data(), sizeof(int64_t) * (size_t)h->nbins);
    return FH_OK;
}

#ifdef FIT_LOOP_TIMING
// debug builds: cycles per phase of the fit_loop kernel accumulated since the context was created
int fh_debug_loop_timing(fh_ctx *c, long long *out16) {
    if (!c || !c->loop_timing.p) return FH_ERR_INVALID;
    HIP_TRY(hipMemcpy(out16, c->loop_timing.p, 16 * sizeof(long long), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemset(c->loop_timing.p, 0, 16 * sizeof(long long)));
    return FH_OK;
}
// per-wave time stamps of one pass (8 waves x 20 steps x 6 stamps), FIT_LOOP_TIMING builds
int fh_debug_loop_trace(fh_ctx *c, long long *out2048) {  // [wave][step < 20][6 stamps], up to 16 waves
    if (!c || !c->loop_timing.p) return FH_ERR_INVALID;
    HIP_TRY(hipMemcpy(out2048, c->loop_timing.p + 16, 2048 * sizeof(long long), hipMemcpyDeviceToHost));
    return FH_OK;
}
#endif

// ---- RCCL -------------------------------------------------------------------------------------------------------------
namespace {
struct RcclApi {
    void *lib = nullptr;
    int (*get_unique_id)(void *) = nullptr;
    int (*allreduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*destroy)(void *) = nullptr;
    const char *(*errstr)(int) = nullptr;
};
struct UniqueId {
    char internal[128];
};
RcclApi g_rccl;
int load_rccl() {
    if (g_rccl.lib) return FH_OK;
    // The RCCL that sits BESIDE the HIP runtime this process runs on.  A process can hold two ROCm stacks (this library
    // loaded first, on /opt/rocm; then `import torch`, which brings its bundled librccl / libhsa-runtime64): a bare
    // dlopen("librccl.so.1") then returns the bundled RCCL, which opens the bundled -- never initialised -- HSA runtime and
    // ncclCommInitRank fails with "no ROCm-capable device is detected".
    void *lib = nullptr;
    Dl_info hip_rt{};
    if (dladdr(reinterpret_cast<void *>(&hipGetDeviceCount), &hip_rt) && hip_rt.dli_fname) {
        std::string dir(hip_rt.dli_fname);
        const size_t slash = dir.rfind('/');
        if (slash != std::string::npos) {
            dir.resize(slash);
            lib = dlopen((dir + "/librccl.so.1").c_str(), RTLD_NOW | RTLD_LOCAL);
            if (!lib) lib = dlopen((dir + "/librccl.so").c_str(), RTLD_NOW | RTLD_LOCAL);
        }
    }
    if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) return fail(FH_ERR_HIP, "cannot load librccl: %s", dlerror());
    g_rccl.get_unique_id = (int (*)(void *))dlsym(lib, "ncclGetUniqueId");
    g_rccl.allreduce = (int (*)(const void *, void *, size_t, int, int, void *, hipStream_t))dlsym(lib, "ncclAllReduce");
    g_rccl.destroy = (int (*)(void *))dlsym(lib, "ncclCommDestroy");
    g_rccl.errstr = (const char *(*)(int))dlsym(lib, "ncclGetErrorString");
    if (!g_rccl.get_unique_id || !dlsym(lib, "ncclCommInitRank") || !g_rccl.allreduce || !g_rccl.destroy)
        return fail(FH_ERR_HIP, "librccl lacks a required symbol");
    g_rccl.lib = lib;
    return FH_OK;
}
}  // namespace

int fh_comm_unique_id(char id[128]) {
    int rc = load_rccl();
    if (rc) return rc;
    int s = g_rccl.get_unique_id(id);
    if (s != 0) return fail(FH_ERR_HIP, "ncclGetUniqueId: %s", g_rccl.errstr ? g_rccl.errstr(s) : "?");
    return FH_OK;
}

int fh_comm_create(const char id[128], int rank, int world, int device, fh_comm **out) {
    if (!id || !out || world < 1 || rank < 0 || rank >= world) return fail(FH_ERR_INVALID, "fh_comm_create: bad argument");
    int rc = load_rccl();
    if (rc) return rc;
    HIP_TRY(hipSetDevice(device));
    fh_comm *cm = new fh_comm();
    cm->rank = rank;
    cm->world = world;
    cm->device = device;
    if (hipEventCreate(&cm->ev0) != hipSuccess || hipEventCreate(&cm->ev1) != hipSuccess) {
        delete cm;
        return fail(FH_ERR_HIP, "fh_comm_create: hipEventCreate failed");
    }
    UniqueId uid;
    memcpy(uid.internal, id, 128);
    // ncclResult_t ncclCommInitRank(ncclComm_t*, int nranks, ncclUniqueId commId, int rank): the id travels by value
    typedef int (*init_fn)(void **, int, UniqueId, int);
    init_fn init = (init_fn)dlsym(g_rccl.lib, "ncclCommInitRank");
    int s = init(&cm->comm, world, uid, rank);
    if (s != 0) {
        delete cm;
        return fail(FH_ERR_HIP, "ncclCommInitRank: %s", g_rccl.errstr ? g_rccl.errstr(s) : "?");
    }
    *out = cm;
    return FH_OK;
}

void fh_comm_destroy(fh_comm *cm) {
    if (!cm) return;
    (void)hipSetDevice(cm->device);
    if (cm->comm && g_rccl.destroy) g_rccl.destroy(cm->comm);
    if (cm->ev0) (void)hipEventDestroy(cm->ev0);
    if (cm->ev1) (void)hipEventDestroy(cm->ev1);
    delete cm;
}

int fh_comm_allreduce_stats(fh_comm *cm, fh_ctx *c) {
    if (!cm || !c || !c->stats_sum.p) return fail(FH_ERR_INVALID, "fh_comm_allreduce_stats: bad argument");
    {
        const int rcs = settle_reset(c);
        if (rcs) return rcs;
    }
    if (c->device != cm->device) return fail(FH_ERR_INVALID, "fh_comm_allreduce_stats: context and communicator live on different devices");
    HIP_TRY(hipSetDevice(c->device));
    enum { kFloat64 = 8, kSum = 0, kMax = 2 };  // ncclDataType_t / ncclRedOp_t values (rccl.h)
    // the buffer fh_stats_finalize reads: the packed tile triangle, or the dense (N+1)^2 Gram of the rows + dgemm path
    // (N > 303: it lives in stats_sum; debris model at N <= 303: in wide_G) -- always with its two trailing scalars
    double *buf = use_wide(c) ? dense_gram(c) : c->stats_sum.p;
    const size_t len = use_wide(c) ? dense_tail(c) + 2 : c->stats_sum.n;
    HIP_TRY(hipEventRecord(cm->ev0, c->stream));
    int s = g_rccl.allreduce(buf, buf, len, kFloat64, kSum, cm->comm, c->stream);
    if (s == 0) s = g_rccl.allreduce(c->stats_minmax.p, c->stats_minmax.p, 2, kFloat64, kMax, cm->comm, c->stream);
    if (s != 0) return fail(FH_ERR_HIP, "ncclAllReduce: %s", g_rccl.errstr ? g_rccl.errstr(s) : "?");
    HIP_TRY(hipEventRecord(cm->ev1, c->stream));
    cm->timed = true;
    c->have_device_Mj = false;
    return FH_OK;
}

int fh_comm_last_allreduce_ms(fh_comm *cm, float *ms) {
    if (!cm || !ms) return fail(FH_ERR_INVALID, "fh_comm_last_allreduce_ms: NULL argument");
    if (!cm->timed) return fail(FH_ERR_INVALID, "no all-reduce recorded yet");
    HIP_TRY(hipSetDevice(cm->device));
    HIP_TRY(hipEventSynchronize(cm->ev1));
    HIP_TRY(hipEventElapsedTime(ms, cm->ev0, cm->ev1));
    return FH_OK;
}

int fh_comm_size(const fh_comm *cm) { return cm ? cm->world : 0; }

}  // extern "C"
