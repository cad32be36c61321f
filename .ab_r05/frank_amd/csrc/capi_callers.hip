// capi_callers.hip -- see capi_internal.h for the map of the C-ABI files.
#include "capi_internal.h"

extern "C" {

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


}  // extern "C"
