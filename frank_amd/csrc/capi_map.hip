// capi_map.hip -- see capi_internal.h for the map of the C-ABI files.
#include "capi_internal.h"

extern "C" {

// ---- a3/a7: H(q), predict ------------------------------------------------------------------------------------
int stage_q(fh_ctx *c, const double *q, int64_t n) {
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

int k1v2_ensure_table(fh_ctx *c, int nb_needed);

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

// ---- K1 ----------------------------------------------------------------------------------------------------------
// the rows-to-memory + dgemm path is taken for N > 303 always and for the debris model at any N
bool use_wide(const fh_ctx *c) {
    return c->wide || (c->debris && !c->v2) || (c->v2 && !c->rows_ok && (c->debris || !c->k1_moments));
}
double *dense_gram(fh_ctx *c) { return c->wide ? c->stats_sum.p : c->wide_G.p; }  // (N+1)^2 + 2 scalars
size_t dense_tail(const fh_ctx *c) { return ((size_t)c->N + 1) * ((size_t)c->N + 1); }

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

int k1v2_ensure_table(fh_ctx *c, int nb_needed) {
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
    // Built on the device (round 5, j0_buckets_device.hip): the host seeds every 16th bucket in long double, one thread per
    // (chain, column) marches the Taylor expansion in double-double arithmetic -- 35 ms of host work and a 55 MB upload for a table
    // that reaches Q_max at N = 300 become ~3 ms.  FRANK_AMD_K1_TABLES=host keeps the long-double construction (and the
    // single-precision arithmetic keeps it too: its tables are rounded on the host).
    static const bool host_tables = [] { const char *e = FH_DEV_STR("FRANK_AMD_K1_TABLES"); return e && !strcmp(e, "host"); }();
    if (!host_tables && !c->arith32) {
        const int have = c->k1_nb_built;
        const int chains = fh_k1_seed_chains(have, nb_new);
        std::vector<int> seed_buckets((size_t)chains);
        for (int i = 0; i < chains; ++i) seed_buckets[i] = fh_k1_seed_bucket(have, i);
        std::vector<double> seeds((size_t)chains * c->N * 4);
        if (fh_k1_bucket_seeds(c->dht->zeros.data(), c->N, seed_buckets.data(), chains, seeds.data()) != 0)
            return fail(FH_ERR_INVALID, "fh_k1_bucket_seeds failed");
        DevBuf<double> grown, dseeds;
        if (grown.alloc((size_t)nb_new * per) != hipSuccess || dseeds.alloc(seeds.size()) != hipSuccess)
            return fail(FH_ERR_NOMEM, "hipMalloc of the bucket tables failed");
        HIP_TRY(hipStreamSynchronize(c->stream));  // nothing in flight may still read the old device table
        HIP_TRY(hipMemcpyAsync(dseeds.p, seeds.data(), sizeof(double) * seeds.size(), hipMemcpyHostToDevice, c->stream));
        if (have) HIP_TRY(hipMemcpyAsync(grown.p, c->k1_table.p, sizeof(double) * (size_t)have * per, hipMemcpyDeviceToDevice, c->stream));
        // (columns k >= N of a bucket's rows are zero: the kernels read XS of them)
        HIP_TRY(hipMemsetAsync(grown.p + (size_t)have * per, 0, sizeof(double) * (size_t)(nb_new - have) * per, c->stream));
        HIP_TRY(fh_k1_bucket_table_device(c->zeros.p, c->N, c->XS, have, nb_new, c->k1_delta, dseeds.p, grown.p + (size_t)have * per, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));  // (seeds and dseeds go out of scope)
        c->k1_nb_built = 0;
        c->k1_table.release();
        c->k1_table.p = grown.p;
        c->k1_table.n = grown.n;
        c->k1_table.owned = true;
        grown.p = nullptr;  // (ownership moved)
        grown.n = 0;
        c->k1_nb_built = nb_new;
        return FH_OK;
    }
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

// The context's device-resident Taylor tables of the first nb buckets (built if need be), layout [bucket][12][N]: tests compare
// the tables built on the device with the long-double construction of the host (fh_dht_bucket_tables).
int fh_ctx_bucket_tables(fh_ctx *c, int nb, double *table) {
    if (!c || nb < 1 || !table) return fail(FH_ERR_INVALID, "fh_ctx_bucket_tables: bad argument");
    if (!c->v2) return fail(FH_ERR_UNSUPPORTED, "fh_ctx_bucket_tables: this context does not use the bucket tables");
    HIP_TRY(hipSetDevice(c->device));
    const int rc = k1v2_ensure_table(c, nb);
    if (rc) return rc;
    HIP_TRY(hipMemcpy2D(table, sizeof(double) * c->N, c->k1_table.p, sizeof(double) * c->XS, sizeof(double) * c->N,
                        (size_t)nb * FH_K1_TERMS, hipMemcpyDeviceToHost));
    return FH_OK;
}

// The two fills of fh_bin_reset.  On a device whose other compute units run fit loops every kernel boundary of the binning
// stream costs ~40 us (the L2 write-backs between dependent kernels find the caches full of the loops' dirty tiles): a step of
// the pipeline was sixteen kernels, two of them these fills.  fh_bin_reset only notes that the sums are to start from
// zero; the last kernel of the moments path (vr_finish_kernel) then stores instead of adding; every other reader or writer
// of the sums calls settle_reset() first.
int settle_reset(fh_ctx *c) {
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
int running_fit_loops(fh_ctx *c) {
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
// ONE LOOK (round 6).  The range pass and the histogram pass of a table the context has not binned last both read (u, v) and form
// every row's baseline; the histogram only needs the bucket COUNT to be an upper bound -- s = q / Qmax <= q_N / Qmax for every table
// that passes the range check, and the histograms of a larger count are the same numbers with zeros behind them --, so one launch of
// uv_hist_kernel<true> with that bound gives the range and the histograms: 16 B per row and ~35 us less per pass.  Where the bound
// keeps the launch geometry of the sort (eight waves per workgroup: <= 2 048 buckets, N <~ 320); a table that reaches beyond the
// bound (range check off) takes the two looks of rounds 3-5.
static int one_look_buckets(const fh_ctx *c) {
    const double cap = c->dht->q[c->N - 1] / c->dht->Qmax * (1.0 + 1e-9) / c->k1_delta + 3.0;
    return cap <= 2048.0 ? (int)cap : 0;
}

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
void load_k1_env(fh_ctx *c) {
    fh_ctx::K1Env e;
    e.unroll = FH_DEV_INT("FRANK_AMD_K1_UNROLL", 2) == 2 ? 2 : 1;
    e.seg = FH_DEV_INT("FRANK_AMD_K1_SEG", 4096);
    e.wpb = FH_DEV_INT("FRANK_AMD_K1_WPB", 0);
    e.blocks = FH_DEV_INT("FRANK_AMD_K1_BLOCKS", 0);
    e.vrwaves = FH_DEV_INT("FRANK_AMD_K1_VRWAVES", 8);
    e.vrsplit = FH_DEV_INT("FRANK_AMD_K1_VRSPLIT", 8);
    e.vrblocks = FH_DEV_INT("FRANK_AMD_K1_VRBLOCKS", 0);
    e.no_range_cache = getenv("FRANK_AMD_NO_RANGE_CACHE") != nullptr;
    e.safe_trig = getenv("FRANK_AMD_K1_SAFE_TRIG") != nullptr;
    e.no_hist_cache = getenv("FRANK_AMD_K1_NO_HIST_CACHE") != nullptr;
    const char *vr = FH_DEV_STR("FRANK_AMD_K1_VR");
    e.vr_slabs = vr && !strcmp(vr, "slabs");
    e.dynamic = FH_DEV_SET("FRANK_AMD_K1_DYNAMIC");
    e.fused = env_int("FRANK_AMD_K1_FUSED", 0);
    if (const char *r = FH_DEV_STR("FRANK_AMD_K1_RESERVE_MULT")) e.reserve_mult = atof(r);
    c->k1env = e;
}
int fh_ctx_reload_env(fh_ctx *c) {
    if (!c) return fail(FH_ERR_INVALID, "fh_ctx_reload_env: NULL argument");
    load_k1_env(c);
    c->throughput_context = false;  // (the pipeline's memory of having been full: capi_fit.hip)
    return FH_OK;
}
static int bin_visibilities_v4(fh_ctx *c, BinParams &p, int64_t count, unsigned long long vis_serial,
                               unsigned long long mult_gen) {
    PrepassParams P{};
    P.bin = p;
    P.partial_scalars = c->partial_scalars.p;
    const fh_ctx::K1Env &E = c->k1env;
    P.unroll = E.unroll;
    const double gkey[6] = {p.dRA, p.dDec, p.cos_t, p.sin_t, p.cos_i, p.sin_i};
    const bool known = c->range_valid && c->range_vis == vis_serial && c->range_mult_gen == mult_gen && c->range_first == p.first &&
                       c->range_count == count && memcmp(gkey, c->range_geom, sizeof gkey) == 0 && !c->no_range_cache &&
                       !E.no_range_cache;
    // qmax_all: over every row of the range whatever its multiplicity -- the sort is sized from it, because rows drawn
    // zero times are still sorted (with weight 0) and must land in a bucket of their own argument
    double qmax = 0.0, qmin = INFINITY, qmax_all = 0.0;
    c->rng_timed = false;
    // one look (see one_look_buckets): the histograms of these rows for look_cap >= nb buckets, from the look-ahead or from this call
    int *look_hist = nullptr;
    int look_cap = 0, look_stride = 0, look_wpb = 0, look_blocks = 0;
    const double *look_scalars = nullptr;
    hipEvent_t look_event = nullptr;
    fh_ctx::LookAhead *look_la = nullptr;
    if (known) {  // same rows, same geometry: the range is the one read back last time, no host round trip
        qmin = c->prepass_qmin;
        qmax = c->prepass_qmax;
        qmax_all = c->prepass_qmax_all;
    } else {  // one look at (u, v): 16 B per visibility and the one host round trip of the pass (64 KB of per-workgroup scalars)
        const double *sc = nullptr;
        int rblocks = 0;
        fh_ctx::LookAhead *hit = nullptr;
        for (auto &lp : c->pf)
            if (lp->valid && lp->vis == vis_serial && lp->mult_gen == mult_gen && lp->first == p.first && lp->count == count &&
                memcmp(gkey, lp->geom, sizeof gkey) == 0)
                hit = lp.get();
        if (hit) {
            // ... taken ahead of time on the look-ahead stream (fh_bin_prefetch_range): wait for THAT pass only
            HIP_TRY(hipEventSynchronize(hit->event));
            sc = hit->host;
            rblocks = hit->blocks;
            hit->valid = false;
            c->rng_la = hit;
            c->rng_timed = true;
            if (hit->has_hist && hit->unroll == P.unroll) {
                look_hist = hit->hist.p;
                look_cap = hit->nb_cap;
                look_stride = hit->hist_stride;
                look_wpb = hit->wpb;
                look_blocks = hit->blocks;
                look_scalars = hit->dev.p;
                look_event = hit->event;
                look_la = hit;
            }
        } else {
            fh_prepass_geometry(0, c->num_cu, &P.wpb, &P.blocks);
            if (P.blocks > c->deproject_blocks) P.blocks = c->deproject_blocks;
            rblocks = P.blocks;
            const int cap = E.wpb ? 0 : one_look_buckets(c);  // (development geometries of P1 / P2: two looks)
            HIP_TRY(hipEventRecord(c->ev_rng0, c->stream));
            if (cap > 0 && !E.no_hist_cache) {  // one look: range + histograms (the context's own buffer)
                const int stride = (P.blocks + 255) & ~255;
                if (c->k1_hist.n < (size_t)stride * cap) {
                    if (c->k1_hist.alloc((size_t)stride * cap + 1024) != hipSuccess) return fail(FH_ERR_NOMEM, "hipMalloc (histograms) failed");
                    HIP_TRY(hipMemsetAsync(c->k1_hist.p, 0, sizeof(int) * c->k1_hist.n, c->stream));
                }
                PrepassParams Q = P;
                Q.nb = cap;
                Q.inv_delta = 1.0 / c->k1_delta;
                Q.delta = c->k1_delta;
                Q.hist = c->k1_hist.p;
                Q.hist_stride = stride;
                Q.hist_zeroed = 1;
                HIP_TRY(fh_prepass_launch_zero(c->k1_hist.p, (size_t)stride * cap, c->stream));
                HIP_TRY(fh_prepass_launch_look(Q, c->stream));
                look_hist = c->k1_hist.p;
                look_cap = cap;
                look_stride = stride;
                look_wpb = P.wpb;
                look_blocks = P.blocks;
            } else {
                HIP_TRY(fh_prepass_launch_range(P, c->stream));
            }
            HIP_TRY(hipEventRecord(c->ev_rng1, c->stream));
            c->rng_la = nullptr;
            c->rng_timed = true;
            c->k1_scalars_host.resize((size_t)rblocks * 4);
            HIP_TRY(hipMemcpyAsync(c->k1_scalars_host.data(), c->partial_scalars.p, sizeof(double) * (size_t)rblocks * 4,
                                   hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            sc = c->k1_scalars_host.data();
        }
        for (int b = 0; b < rblocks; ++b) {
            const double mn = sc[(size_t)b * 4 + 1], m = sc[(size_t)b * 4 + 2], ma = sc[(size_t)b * 4 + 3];
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
    // the histograms of a look serve if they cover the buckets and were counted by the workgroups that will scatter
    if (look_hist && (nb > look_cap || look_wpb != P.wpb || look_blocks != P.blocks)) look_hist = nullptr;
    // workspaces (grow on demand)
    const size_t nrows = (size_t)count + 16 * (size_t)nb + 16, max_pc = (size_t)fh_prepass_max_pieces(count, nb, seg);
    const size_t md = (size_t)fh_prepass_moment_doubles();
    if (c->k1_rows.n < nrows * 3 && c->k1_rows.alloc(nrows * 3 + 1024) != hipSuccess) return fail(FH_ERR_NOMEM, "hipMalloc (sorted rows) failed");
    P.hist_stride = (P.blocks + 255) & ~255;
    if (look_hist && look_stride != P.hist_stride) look_hist = nullptr;
    if (!look_hist && c->k1_hist.n < (size_t)P.hist_stride * nb) {
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
    P.hist = look_hist ? look_hist : c->k1_hist.p;
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
    // the fused form (bin_fused.hip, opt-in): one pass over the table, the buckets' moments accumulated in LDS -- when every bucket
    // has an accumulator slot there; not for multiplicities or fp32 tables (they keep the sorted path)
    const int fused_slots = fh_fused_max_slots(nb);
    const bool fused = E.fused > 0 && !p.mult && !p.u32 && fused_slots >= nb;
    const int fused_G = c->num_cu > 0 ? c->num_cu : 256;
    if (fused) {
        const size_t need = (size_t)nb * fused_G * md;
        if (c->k1_partial.n < need && c->k1_partial.alloc(need + 4096) != hipSuccess) return fail(FH_ERR_NOMEM, "hipMalloc (moments) failed");
        P.partial = c->k1_partial.p;
        if (c->k1_slot_tab.n < (size_t)nb + 1 && c->k1_slot_tab.alloc((size_t)nb + 257) != hipSuccess) return fail(FH_ERR_NOMEM, "hipMalloc failed");
        P.fused = 1;
    }
    const bool reuse = known && c->hist_valid && c->hist_nb == nb && c->hist_blocks == P.blocks && c->hist_wpb == P.wpb &&
                       c->hist_unroll == P.unroll && c->hist_seg == seg && c->hist_fused == fused && !E.no_hist_cache;
    c->hist_valid = false;
    // (the events of fh_bin_last_prepass_ms start HERE: behind the range -- cached, looked ahead, or waited for above; its kernel has
    //  events of its own, fh_bin_last_range_ms)
    if (look_hist && look_event) {
        // the look-ahead's histograms and per-workgroup ranges (vr_finish folds those) were written on its stream
        HIP_TRY(hipStreamWaitEvent(c->stream, look_event, 0));
        HIP_TRY(hipMemcpyAsync(c->partial_scalars.p, look_scalars, sizeof(double) * (size_t)P.blocks * 4, hipMemcpyDeviceToDevice, c->stream));
    }
    HIP_TRY(hipEventRecord(c->ev_pre0, c->stream));
    const int hist_mode = reuse ? 1 : (look_hist ? 2 : 0);  // 1: histograms, scan and layout stand; 2: the histograms exist (one look)
    if (fused) {
        if (hist_mode == 0) HIP_TRY(fh_prepass_launch_hist(P, c->stream));
        if (hist_mode == 2) HIP_TRY(fh_prepass_launch_scan(P, c->stream));
        if (hist_mode != 1) HIP_TRY(fh_fused_launch_layout(P, fused_G, fused_slots, c->k1_slot_tab.p, c->k1_slot_tab.p + nb, c->stream));
        HIP_TRY(fh_fused_launch(P, c->k1_slot_tab.p, fused_slots, fused_G, P.blocks, c->stream));
        HIP_TRY(fh_prepass_launch_factor(P, c->stream));
    } else {
        HIP_TRY(fh_prepass_launch(P, c->stream, hist_mode));
    }
    if (look_hist && look_la) {
        // the pre-pass kernels above read the look-ahead's histograms when the binning stream gets to them, which may be long after
        // this call returns (a pipeline queues passes ahead): the next look into that slot waits for this event
        HIP_TRY(hipEventRecord(look_la->consumed, c->stream));
        look_la->consumed_pending = true;
    }
    // (the histograms the context keeps between passes are its own: those of a look-ahead belong to the look-ahead)
    c->hist_valid = !(look_hist && look_event);
    c->hist_fused = fused;
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
void table_columns(BinParams &p, const fh_vis *vis, int64_t first, int64_t count) {
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

// columns, row range, geometry and DHT constants of a pass, as every kernel that streams the table takes them
static void bin_params(fh_ctx *c, const fh_geometry *g, const fh_vis *vis, int64_t first, int64_t count, BinParams &p) {
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
}

int fh_bin_prefetch_range(fh_ctx *c, const fh_geometry *g, const fh_vis *vis, int64_t first, int64_t count) {
    if (!c || !g || !vis) return fail(FH_ERR_INVALID, "fh_bin_prefetch_range: NULL argument");
    if (first < 0 || count < 0 || first + count > vis->n) return fail(FH_ERR_INVALID, "fh_bin_prefetch_range: bad range");
    if (vis->device != c->device) return fail(FH_ERR_INVALID, "visibility table lives on another device");
    if (!(c->v2 && !use_wide(c) && c->k1_moments && !c->debris && !c->arith32) || count == 0) return FH_OK;  // (only the moments path sizes a sort from the range)
    HIP_TRY(hipSetDevice(c->device));
    PrepassParams P{};
    bin_params(c, g, vis, first, count, P.bin);
    fh_prepass_geometry(0, c->num_cu, &P.wpb, &P.blocks);
    P.unroll = c->k1env.unroll;
    if (!c->pf_stream) {
        HIP_TRY(hipStreamCreateWithFlags(&c->pf_stream, hipStreamNonBlocking));
        reader_stream_add(c->device, c->pf_stream);
    }
    // a free look-ahead: not waiting to be taken, and the pass that took its histograms has run; else a new one; else the oldest
    fh_ctx::LookAhead *pick = nullptr, *oldest = nullptr;
    for (auto &lp : c->pf) {
        if (!oldest || lp->seq < oldest->seq) oldest = lp.get();
        if (pick || lp->valid) continue;
        if (lp->consumed_pending && hipEventQuery(lp->consumed) != hipSuccess) continue;
        lp->consumed_pending = false;
        pick = lp.get();
    }
    if (!pick && (int)c->pf.size() < fh_ctx::kMaxLookAheads) {
        c->pf.emplace_back(new fh_ctx::LookAhead());
        pick = c->pf.back().get();
    }
    if (!pick) pick = oldest;  // (its look, if nobody took it, is dropped; its consumer, if still queued, is waited for below)
    fh_ctx::LookAhead &la = *pick;
    la.seq = ++c->pf_seq;
    if (!la.event) {
        HIP_TRY(hipEventCreateWithFlags(&la.event, hipEventDisableSystemFence));
        HIP_TRY(hipEventCreateWithFlags(&la.ev0, hipEventDisableSystemFence));
        HIP_TRY(hipEventCreateWithFlags(&la.ev1, hipEventDisableSystemFence));
        HIP_TRY(hipEventCreateWithFlags(&la.consumed, hipEventDisableSystemFence));
    }
    if (la.consumed_pending) {  // (a queued pass still reads this slot's histograms)
        HIP_TRY(hipStreamWaitEvent(c->pf_stream, la.consumed, 0));
        la.consumed_pending = false;
    }
    if (la.blocks < P.blocks) {
        if (la.host) (void)hipHostFree(la.host);
        la.host = nullptr;
        HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&la.host), sizeof(double) * (size_t)P.blocks * 4, hipHostMallocDefault));
        HIP_TRY(la.dev.alloc((size_t)P.blocks * 4));
    }
    la.valid = false;
    la.blocks = P.blocks;
    P.partial_scalars = la.dev.p;
    la.has_hist = false;
    const int cap = (c->k1env.wpb || c->k1env.no_hist_cache) ? 0 : one_look_buckets(c);
    if (cap > 0) {  // one look: the histograms of these rows beside the range (see one_look_buckets)
        if (P.blocks > c->deproject_blocks) P.blocks = c->deproject_blocks;
        la.blocks = P.blocks;
        const int stride = (P.blocks + 255) & ~255;
        if (la.hist.n < (size_t)stride * cap) HIP_TRY(la.hist.alloc((size_t)stride * cap + 1024));
        P.nb = cap;
        P.inv_delta = 1.0 / c->k1_delta;
        P.delta = c->k1_delta;
        P.hist = la.hist.p;
        P.hist_stride = stride;
        la.nb_cap = cap;
        la.hist_stride = stride;
        la.wpb = P.wpb;
        la.unroll = P.unroll;
    }
    HIP_TRY(hipEventRecord(la.ev0, c->pf_stream));
    if (cap > 0) {
        P.hist_zeroed = 1;
        HIP_TRY(fh_prepass_launch_zero(la.hist.p, (size_t)la.hist_stride * cap, c->pf_stream));
        HIP_TRY(fh_prepass_launch_look(P, c->pf_stream));
        la.has_hist = true;
    } else {
        HIP_TRY(fh_prepass_launch_range(P, c->pf_stream));
    }
    HIP_TRY(hipEventRecord(la.ev1, c->pf_stream));
    HIP_TRY(hipMemcpyAsync(la.host, la.dev.p, sizeof(double) * (size_t)P.blocks * 4, hipMemcpyDeviceToHost, c->pf_stream));
    HIP_TRY(hipEventRecord(la.event, c->pf_stream));
    const double gkey[6] = {P.bin.dRA, P.bin.dDec, P.bin.cos_t, P.bin.sin_t, P.bin.cos_i, P.bin.sin_i};
    memcpy(la.geom, gkey, sizeof gkey);
    la.vis = vis->serial;
    la.mult_gen = vis->use_mult ? vis->mult_gen : 0;
    la.first = first;
    la.count = count;
    la.valid = true;
    return FH_OK;
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
    bin_params(c, g, vis, first, count, p);
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
            if (FH_DEV_SET("FRANK_AMD_WIDE_SYRK")) {
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

int fh_bin_last_range_ms(fh_ctx *c, float *ms) {
    if (!c || !ms) return fail(FH_ERR_INVALID, "fh_bin_last_range_ms: NULL argument");
    *ms = 0.0f;
    if (!c->rng_timed) return FH_OK;  // (the last pass took its range from the cache: no kernel)
    if (c->rng_la) {
        HIP_TRY(hipEventSynchronize(c->rng_la->ev1));
        HIP_TRY(hipEventElapsedTime(ms, c->rng_la->ev0, c->rng_la->ev1));
    } else {
        HIP_TRY(hipEventSynchronize(c->ev_rng1));
        HIP_TRY(hipEventElapsedTime(ms, c->ev_rng0, c->ev_rng1));
    }
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


}  // extern "C"
