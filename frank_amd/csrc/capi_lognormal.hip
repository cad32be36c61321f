// capi_lognormal.hip -- see capi_internal.h for the map of the C-ABI files.
#include <chrono>

#include "capi_internal.h"

extern "C" {

// ---- method='LogNormal' ------------------------------------------------------------------------------------------
static int ln_np(int N) { return 16 * ((N + 15) / 16); }

static int ln_prepare(fh_ctx *c, const double *M, const double *j, LogNormalParams &P) {
    const int N = c->N;
    const size_t NN = (size_t)N * N;
    if (N > 640) return fail(FH_ERR_UNSUPPORTED, "N = %d: the LogNormal kernel covers N <= 640", N);
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
        const char *d = getenv("FRANK_AMD_LN_CLUSTER_CHOL");
        P.dist_cholesky = (d && d[0] == '0') ? 0 : 1;
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

// 320 < N <= 640: the persistent kernel in its WIDE form (round 6; vectors in global memory, one panel) unless
// FRANK_AMD_LN_WIDE=host keeps the host-driven route of round 4 (lognormal_wide.hip) -- which also takes over when the kernel's
// Cholesky fails (it has rocSOLVER's pivoted LU) and for N >= 640
static bool ln_kernel_covers(const fh_ctx *c) {
    if (c->N <= 320) return true;
    const char *e = getenv("FRANK_AMD_LN_WIDE"), *pv = getenv("FRANK_AMD_LN_PIVOTED");
    return c->N <= 640 && !(e && !strcmp(e, "host")) && !(pv && pv[0] == '1');
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
    if (result[1] == LN_STATUS_NOT_SPD) return fail(FH_ERR_NOT_SPD, "Cholesky of a Hessian failed (N > 320: no pivoted LU in the kernel)");
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
        use_inverse = FH_DEV_INT("FRANK_AMD_LNW_INVERSE", 1) != 0;  // (0: rocSOLVER's getrs at every step, ~4x slower)
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
    if (!ln_kernel_covers(c)) return lognormal_model_wide(c, M, j, p, guess, s0, s_map, Dinv, stats);  // (the host-driven route)
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
    rc = ln_finish(c, s_map, nullptr, Dinv, stats, result);
    if (rc == FH_ERR_NOT_SPD && c->N > 320) return lognormal_model_wide(c, nullptr, nullptr, p, guess, s0, s_map, Dinv, stats);
    return rc;
}

int fh_fit_lognormal(fh_ctx *c, const double *M, const double *j, double alpha, double p0, double wsmooth, double tol,
                     int max_iter, double I_scale, double *s_map, double *p, int *niter, double *Dinv, int64_t *stats,
                     double *diag_p, double *diag_s) {
    if (!c || !s_map || !p || !niter) return fail(FH_ERR_INVALID, "fh_fit_lognormal: NULL argument");
    if (max_iter < 0) return fail(FH_ERR_INVALID, "max_iter must be >= 0");
    if (!(I_scale > 0)) return fail(FH_ERR_INVALID, "I_scale must be positive");
    if ((diag_p == nullptr) != (diag_s == nullptr)) return fail(FH_ERR_INVALID, "pass both diag_p and diag_s or neither");
    if (!ln_kernel_covers(c))  // (beyond the persistent kernel: the host-driven route, lognormal_wide.hip)
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
            if (!c->ln_ctl.p) HIP_TRY(c->ln_ctl.alloc(LN_CTL_WORDS));
            if (c->ln_cluster_vecs.n < nv) HIP_TRY(c->ln_cluster_vecs.alloc(nv));
            HIP_TRY(hipMemsetAsync(c->ln_ctl.p, 0, LN_CTL_WORDS * sizeof(int), c->stream));
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
    if (rc == FH_ERR_NOT_SPD && N > 320)  // (the kernel has no pivoted LU beyond N = 320: the route that has rocSOLVER's starts over)
        return fit_lognormal_wide(c, nullptr, nullptr, alpha, p0, wsmooth, tol, max_iter, I_scale, s_map, p, niter, Dinv, stats, diag_p, diag_s);
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
    if (!ln_kernel_covers(c)) {  // beyond the persistent kernel: the host-driven route, one point after the other
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
    // The staged schedule (round 6; the Normal sweeps have had it since round 5: capi_fit.hip, sweep_staged).  A launch ends with its
    // slowest fit, and which points run long is only known once they run (the bench's 64-point grid: eleven of them to max_iter,
    // 2 001 passes of 1.75 ms on one compute unit each, scattered over the grid).  So every fit first runs on one compute unit,
    // but PAUSES -- behind an update of p; its state is (s, p, the p before, the count) -- once every fit has been handed out and
    // only as many are still running as the second stage has clusters for; those continue where they stopped on clusters of eight
    // workgroups of one XCD each (the form of a single fit: S^-1, Tr2, the Hessian builds and the evaluations shared), up to 32
    // clusters in one launch.  From N = 160 on (below, a cluster does not pay); FRANK_AMD_LN_CLUSTER=1 keeps the single launch.
    const int cl2 = env_int("FRANK_AMD_LN_CLUSTER", N >= 160 ? 8 : 1) > 1 && !P.no_cholesky ? 8 : 1;
    const int max_groups = c->num_cu / cl2 > 0 ? c->num_cu / cl2 : 1;
    const bool staged = cl2 > 1 && N > 112 && batch >= 8;
    DevBuf<int> done, ctl1;
    DevBuf<double> vecs1;
    // ... and a batch that leaves compute units idle -- 64 points on 256 units -- runs its FIRST stage on clusters too: the largest of
    // 8, 4, 2 workgroups per fit that keeps every fit resident at once
    int cl1 = 1;
    if (staged)
        for (int cc = 8; cc >= 2; cc >>= 1)
            if ((long long)batch * cc <= (long long)c->num_cu) {
                cl1 = cc;
                break;
            }
    cl1 = FH_DEV_INT("FRANK_AMD_LN_STAGE1_CLUSTER", cl1);
    if (cl1 > 1) {
        const size_t vstride1 = (size_t)2 * N + P.NP;
        if (ctl1.alloc((size_t)LN_CTL_WORDS * G) != hipSuccess || vecs1.alloc(G * vstride1) != hipSuccess)
            return fail(FH_ERR_NOMEM, "device allocation failed");
        HIP_TRY(hipMemsetAsync(ctl1.p, 0, sizeof(int) * LN_CTL_WORDS * G, c->stream));
        P.cluster = cl1;
        P.groups = (int)G;
        P.ctl = ctl1.p;
        P.group_vec_stride = (int)vstride1;
        P.rk_g = vecs1.p;
        P.tr2_g = vecs1.p + N;
        P.dvec_g = vecs1.p + 2 * N;
    }
    if (staged) {
        if (done.alloc(1) != hipSuccess) return fail(FH_ERR_NOMEM, "device allocation failed");
        HIP_TRY(hipMemsetAsync(done.p, 0, sizeof(int), c->stream));
        // (how many: the bench's 64 points, ten of them to max_iter -- 8, 10, 12 left: 3.64 s; 16: 3.78; 24: 3.99; 32: 4.15; the single
        //  launch 4.44: the passes of a point that does not converge are heavy in Newton work, where a cluster gains 1.5 x, not the
        //  2.2 x of a converging fit's passes -- so only the true stragglers move: a sixth of the batch)
        P.pause_when_left = batch / 6 > 2 ? batch / 6 : 2;
        if (P.pause_when_left > max_groups) P.pause_when_left = max_groups;
        P.pause_when_left = FH_DEV_INT("FRANK_AMD_LN_STAGE_LEFT", P.pause_when_left);  // (development: tools/ln_batched64.py sweeps it)
        P.done_counter = done.p;
    }
    const bool trace = FH_DEV_SET("FRANK_AMD_SWEEP_TRACE");  // development: stage times on stderr
    const auto t_stage = std::chrono::steady_clock::now();
    HIP_TRY(fh_ln_launch(P, (int)G, c->stream));
    std::vector<int> res(2 * B);
    std::vector<long long> st(17 * B);
    HIP_TRY(hipMemcpyAsync(res.data(), resb.p, sizeof(int) * 2 * B, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(st.data(), stb.p, sizeof(long long) * 17 * B, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    std::vector<int> paused;
    if (trace) {
        int np = 0, passes = 0;
        for (int k = 0; k < batch; ++k)
            if (res[2 * k + 1] == LN_STATUS_PAUSED) {
                ++np;
                passes += res[2 * k];
            }
        fprintf(stderr, "[ln sweep] stage 1: %.3f s, %d of %d fits paused (mean count %d)\n",
                std::chrono::duration<double>(std::chrono::steady_clock::now() - t_stage).count(), np, batch, np ? passes / np : 0);
    }
    for (int k = 0; k < batch; ++k)
        if (res[2 * k + 1] == LN_STATUS_PAUSED) paused.push_back(k);
    if (!paused.empty()) {
        // second stage: the paused fits, in the order of the first, on clusters
        const size_t K = paused.size();
        const int groups = (int)(K < (size_t)max_groups ? K : (size_t)max_groups);
        const size_t vstride = (size_t)2 * N + P.NP;
        DevBuf<double> rs, sb2, pb2, Hb2, lub2, alb2, p0b2, vecs;
        DevBuf<int> resb2, ctl2;
        DevBuf<long long> stb2;
        if (rs.alloc(K * (3 * (size_t)N + 1)) != hipSuccess || sb2.alloc(K * N) != hipSuccess || pb2.alloc(K * N) != hipSuccess ||
            Hb2.alloc(K * NN) != hipSuccess || lub2.alloc(K * 5 * N) != hipSuccess || alb2.alloc(K) != hipSuccess ||
            p0b2.alloc(K) != hipSuccess || vecs.alloc((size_t)groups * vstride) != hipSuccess || resb2.alloc(2 * K) != hipSuccess ||
            ctl2.alloc((size_t)LN_CTL_WORDS * groups) != hipSuccess || stb2.alloc(17 * K) != hipSuccess)
            return fail(FH_ERR_NOMEM, "fh_fit_lognormal_batched: device allocation for the second stage failed");
        std::vector<double> al2(K), p02(K), lu2(K * 5 * N), cnt(K);
        for (size_t q = 0; q < K; ++q) {
            const int k = paused[q];
            double *dst = rs.p + q * (3 * (size_t)N + 1);
            HIP_TRY(hipMemcpyAsync(dst, sb.p + (size_t)k * N, sizeof(double) * N, hipMemcpyDeviceToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(dst + N, pb.p + (size_t)k * N, sizeof(double) * N, hipMemcpyDeviceToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(dst + 2 * N, Hb.p + (size_t)k * NN, sizeof(double) * N, hipMemcpyDeviceToDevice, c->stream));
            cnt[q] = (double)res[2 * k];
            HIP_TRY(hipMemcpyAsync(dst + 3 * N, &cnt[q], sizeof(double), hipMemcpyHostToDevice, c->stream));
            al2[q] = al_o[k];
            p02[q] = p0_o[k];
            memcpy(lu2.data() + q * 5 * N, lu_all.data() + (size_t)k * 5 * N, sizeof(double) * 5 * N);
        }
        HIP_TRY(hipMemcpyAsync(lub2.p, lu2.data(), sizeof(double) * lu2.size(), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(alb2.p, al2.data(), sizeof(double) * K, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(p0b2.p, p02.data(), sizeof(double) * K, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemsetAsync(counter.p, 0, sizeof(int), c->stream));
        HIP_TRY(hipMemsetAsync(ctl2.p, 0, sizeof(int) * LN_CTL_WORDS * (size_t)groups, c->stream));
        LogNormalParams P2 = P;
        P2.batch = (int)K;
        P2.band_lu = lub2.p;
        P2.batch_alpha = alb2.p;
        P2.batch_p0 = p0b2.p;
        P2.H = Hb2.p;
        P2.s_out = sb2.p;
        P2.p_out = pb2.p;
        P2.result = resb2.p;
        P2.stats = stb2.p;
        P2.resume = rs.p;
        P2.pause_when_left = 0;
        P2.done_counter = nullptr;
        P2.cluster = cl2;
        P2.groups = groups;
        P2.ctl = ctl2.p;
        P2.group_vec_stride = (int)vstride;
        P2.rk_g = vecs.p;
        P2.tr2_g = vecs.p + N;
        P2.dvec_g = vecs.p + 2 * N;
        HIP_TRY(fh_ln_launch(P2, groups, c->stream));
        std::vector<int> res2(2 * K);
        std::vector<long long> st2(17 * K);
        HIP_TRY(hipMemcpyAsync(res2.data(), resb2.p, sizeof(int) * 2 * K, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(st2.data(), stb2.p, sizeof(long long) * 17 * K, hipMemcpyDeviceToHost, c->stream));
        for (size_t q = 0; q < K; ++q) {  // the second stage's results in the places of the first's
            const int k = paused[q];
            HIP_TRY(hipMemcpyAsync(sb.p + (size_t)k * N, sb2.p + q * N, sizeof(double) * N, hipMemcpyDeviceToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(pb.p + (size_t)k * N, pb2.p + q * N, sizeof(double) * N, hipMemcpyDeviceToDevice, c->stream));
        }
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (trace)
            fprintf(stderr, "[ln sweep] both stages: %.3f s, %zu fits on %d clusters of %d\n",
                    std::chrono::duration<double>(std::chrono::steady_clock::now() - t_stage).count(), K, groups, cl2);
        for (size_t q = 0; q < K; ++q) {
            const int k = paused[q];
            res[2 * k] = res2[2 * q];
            res[2 * k + 1] = res2[2 * q + 1];
            for (int e = 0; e < 9; ++e) st[17 * (size_t)k + e] += st2[17 * q + e];
        }
    }
    for (int k = 0; k < batch; ++k) {  // launch order -> the caller's
        HIP_TRY(hipMemcpyAsync(s_map + (size_t)order[k] * N, sb.p + (size_t)k * N, sizeof(double) * N, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(p + (size_t)order[k] * N, pb.p + (size_t)k * N, sizeof(double) * N, hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int k = 0; k < batch; ++k)
        if (res[2 * k + 1] == LN_STATUS_NOT_SPD)
            return fail(FH_ERR_NOT_SPD, "fh_fit_lognormal_batched: Cholesky of a Hessian failed in fit %d (N > 320: FRANK_AMD_LN_WIDE=host "
                        "takes the route with the pivoted LU)", order[k]);
    for (int k = 0; k < batch; ++k)
        if (res[2 * k + 1] == LN_STATUS_CLUSTER || res[2 * k + 1] == LN_STATUS_PAUSED)  // (never a result: fail loudly, as a single fit does)
            return fail(FH_ERR_HIP, "fh_fit_lognormal_batched: a cluster of workgroups stopped answering in the middle of fit %d "
                        "(FRANK_AMD_LN_CLUSTER=1 runs every fit on one workgroup, one launch)", order[k]);
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


}  // extern "C"
