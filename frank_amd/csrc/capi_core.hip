// capi_core.hip -- see capi_internal.h for the map of the C-ABI files.
#include <dirent.h>
#include <unistd.h>

#include <mutex>

#include "capi_internal.h"

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
std::atomic<unsigned long long> g_vis_serial{1};

// ---- the allocation cache of the visibility tables' columns (DevBuf::alloc_pooled) ------------------------------------------
static std::mutex g_pool_mutex;
static std::vector<PoolEntry> g_pool;
static size_t g_pool_held = 0;
void *pool_take(size_t bytes, int device) {
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
void pool_put(void *p, size_t bytes, int device) {
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
void pool_clear() {
    std::lock_guard<std::mutex> lk(g_pool_mutex);
    for (const PoolEntry &e : g_pool) (void)hipFree(e.p);
    g_pool.clear();
    g_pool_held = 0;
}

static std::mutex g_reader_mutex;
static std::vector<std::pair<int, hipStream_t>> g_reader_streams;
void reader_stream_add(int device, hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_reader_mutex);
    g_reader_streams.emplace_back(device, s);
}
void reader_stream_remove(hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_reader_mutex);
    for (size_t i = 0; i < g_reader_streams.size(); ++i)
        if (g_reader_streams[i].second == s) {
            g_reader_streams.erase(g_reader_streams.begin() + (long)i);
            return;
        }
}
void reader_streams_sync(int device) {
    std::lock_guard<std::mutex> lk(g_reader_mutex);  // (held across the waits: a stream in the list is not destroyed meanwhile)
    for (const auto &e : g_reader_streams)
        if (e.first == device) (void)hipStreamSynchronize(e.second);
    (void)hipStreamSynchronize(nullptr);  // (the split of a complex column, the synchronous entry points' kernels)
}

// HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and kernels whose streams share a queue serialise: the
// launches of a pipeline of fits (up to six streams beside the binning stream) want at least eight.  The variable is read when the
// ROCm runtime comes up (the first HIP call of the process).  Round 6: neither loading the library nor importing the Python package
// touches the process any more; the FIRST entry point of this library that is about to make a HIP call (fh_ctx_create,
// fh_vis_upload*, fh_device_count) looks whether the runtime is up already -- /dev/kfd among the process's descriptors -- and, if
// it is not, exports GPU_MAX_HW_QUEUES=24 unless the variable is set.  What the runtime latched is then known: the variable's value
// if it came up under this library's eyes (or was set by the user), the default 4 if somebody else (torch imported first, say)
// brought it up with the variable unset -- and that, not the environment of the moment, is what a context's warning is keyed on.
static thread_local std::string g_warn;
constexpr int kHwQueuesWanted = 8;
static int hw_queues_env() {
    const char *e = getenv("GPU_MAX_HW_QUEUES");
    return e ? atoi(e) : 4;  // (the runtime's default)
}
static bool rocm_runtime_up() {
    DIR *d = opendir("/proc/self/fd");
    if (!d) return false;
    bool up = false;
    char path[64], target[256];
    while (const dirent *e = readdir(d)) {
        if (e->d_name[0] == '.') continue;
        snprintf(path, sizeof path, "/proc/self/fd/%s", e->d_name);
        const ssize_t n = readlink(path, target, sizeof target - 1);
        if (n <= 0) continue;
        target[n] = 0;
        if (!strcmp(target, "/dev/kfd")) {
            up = true;
            break;
        }
    }
    closedir(d);
    return up;
}
static int g_hw_queues_latched = 0;  // what the runtime read when it came up, as far as this library can tell (0: not settled yet)
void settle_hw_queues() {
    static std::once_flag once;
    std::call_once(once, [] {
        const bool up = rocm_runtime_up();
        const bool had = getenv("GPU_MAX_HW_QUEUES") != nullptr;
        if (!up) setenv("GPU_MAX_HW_QUEUES", "24", 0);  // never overrides a value the user chose
        g_hw_queues_latched = (up && !had) ? 4 : hw_queues_env();
    });
}

extern "C" {


int fh_init(void) {
    settle_hw_queues();
    return g_hw_queues_latched;
}
const char *fh_last_warning(void) { return g_warn.c_str(); }

const char *fh_last_error(void) { return g_err.c_str(); }
// (the build stamp ties a profile under profiles/ to the binary it was taken from: tools/profile_r04.sh records it, bench.py
//  prints the loaded library's beside the profile's)
extern "C" const char *fh_build_stamp(void);  // version_stamp.cpp: compiled again whenever any object of the library changes
const char *fh_version(void) { return fh_build_stamp(); }

int fh_device_count(int *count) {
    settle_hw_queues();
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
void load_k1_env(fh_ctx *c);  // (the FRANK_AMD_K1_* switches, read once per context)
int fh_ctx_create(const fh_dht *dht, int device, fh_ctx **out) {
    settle_hw_queues();
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
    reader_stream_add(c->device, c->stream);
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
    HIP_TRY(hipEventCreateWithFlags(&c->ev_rng0, hipEventDisableSystemFence));
    HIP_TRY(hipEventCreateWithFlags(&c->ev_rng1, hipEventDisableSystemFence));
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
                if (const char *e = FH_DEV_STR("FRANK_AMD_K1_SPLIT")) g0 = atoi(e);  // development: workgroups of part 0
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
    if (g_hw_queues_latched < kHwQueuesWanted) {
        char buf[400];
        snprintf(buf, sizeof buf, "the ROCm runtime of this process came up with %d hardware queues (GPU_MAX_HW_QUEUES): pipelined fits "
                 "(fh_fit_submit) put their launches on up to six streams beside the binning stream; with fewer than %d queues HIP lets "
                 "streams share a queue and their kernels serialise.  Export GPU_MAX_HW_QUEUES=24 -- or call fh_init() -- before the "
                 "first HIP call of the process (e.g. before importing torch).", g_hw_queues_latched, kHwQueuesWanted);
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
    if (c->pf_stream) {
        (void)hipStreamSynchronize(c->pf_stream);
        reader_stream_remove(c->pf_stream);
        (void)hipStreamDestroy(c->pf_stream);
    }
    if (c->ev_rng0) (void)hipEventDestroy(c->ev_rng0);
    if (c->ev_rng1) (void)hipEventDestroy(c->ev_rng1);
    for (auto &lp : c->pf) {
        fh_ctx::LookAhead &la = *lp;
        if (la.event) (void)hipEventDestroy(la.event);
        if (la.ev0) (void)hipEventDestroy(la.ev0);
        if (la.ev1) (void)hipEventDestroy(la.ev1);
        if (la.consumed) (void)hipEventDestroy(la.consumed);
        if (la.host) (void)hipHostFree(la.host);
    }
    if (c->ev_loop0) (void)hipEventDestroy(c->ev_loop0);
    if (c->ev_loop1) (void)hipEventDestroy(c->ev_loop1);
    if (c->stream) {
        reader_stream_remove(c->stream);
        (void)hipStreamDestroy(c->stream);
    }
    delete c;
}
int fh_ctx_synchronize(fh_ctx *c) {
    if (!c) return fail(FH_ERR_INVALID, "ctx is NULL");
    HIP_TRY(hipStreamSynchronize(c->stream));
    return FH_OK;
}
void *fh_ctx_stream(fh_ctx *c) { return c ? (void *)c->stream : nullptr; }

// ---- visibility tables -------------------------------------------------------------------------------------------
int fh_vis_upload(int device, const double *u, const double *v, const double *Vre, const double *Vim, const double *w,
                  int64_t n_w, int64_t n, fh_vis **out) {
    settle_hw_queues();
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
    settle_hw_queues();
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
    // (every way out, the error returns included, waits for the device before tmp -- and, through the unique_ptr declared before it,
    //  the table's columns -- go back to the pool: a buffer released while the split kernel of the null stream may still touch it
    //  would be handed to the next upload)
    DevBuf<double> tmp;
    struct ReleaseAfterSync {  // tmp's memory returns to the pool only behind a device synchronisation
        DevBuf<double> &b;
        ~ReleaseAfterSync() {
            (void)hipStreamSynchronize(nullptr);  // (tmp is written by a null-stream copy and read by the null-stream split kernel only)
            b.release();
        }
    } tmp_guard{tmp};
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
        HIP_TRY(hipStreamSynchronize(nullptr));  // (the split runs on the null stream; tmp goes away with this scope)
    }
    *out = t.release();
    return FH_OK;
}

int fh_vis_upload_f32(int device, const float *u, const float *v, const float *Vre, const float *Vim, const float *w,
                      int64_t n_w, int64_t n, fh_vis **out) {
    settle_hw_queues();
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
    // (hipFree waited for the device; the columns now go back to a cache and may be handed out again at once: the kernels that
    //  still read them must have ended -- those run on the contexts' binning and look-ahead streams and on the null stream)
    reader_streams_sync(vis->device);
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


}  // extern "C"
