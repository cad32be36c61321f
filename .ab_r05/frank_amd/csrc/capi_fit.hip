// capi_fit.hip -- see capi_internal.h for the map of the C-ABI files.
#include "capi_internal.h"

extern "C" {

// ---- K2 -----------------------------------------------------------------------------------------------------------
FitState make_state(fh_ctx *c) {
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
int solve_posterior(fh_ctx *c, const FitState &st, bool with_prior, bool want_tr2) {
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
int svd_pinv_solve_device(fh_ctx *c, double *A_dev, double *B_dev, int nrhs, bool last_axis_scaling) {
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

void smoothing_band_lu(const fh_dht &d, double weights, std::vector<double> &out) {
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
int prepare_qspace(fh_ctx *c, double *Aq, double *bq) {
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
int fit_cluster_size(const fh_ctx *c) {
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

FitLoopParams make_loop_params(fh_ctx *c, int mode, double alpha, double p0, double tol, int max_iter) {
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
std::vector<int> sweep_launch_order(const double *alpha, const double *wsmooth, int batch) {
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
    if (counter.alloc(2) != hipSuccess || Cb.alloc(G * PP) != hipSuccess || Wb.alloc(G * PP) != hipSuccess ||
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
        HIP_TRY(hipMemsetAsync(counter.p, 0, 2 * sizeof(int), c->stream));
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
        P.pass_cap = pass_cap > 0 ? pass_cap : 0;
        if (pass_cap < 0) {  // adaptive: the last -pass_cap fits still running pause together
            P.pause_when_left = -pass_cap;
            P.done_counter = counter.p + 1;
        }
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
    // ---- stage 1: every fit; cap > 0: at most `cap` passes; cap == 0: until the fits still running are few enough for the
    //      clusters of stage 2 (four clusters per XCD: 32 on this device) -- they then pause together, wherever they are ----
    const int left = env_int("FRANK_AMD_SWEEP_LEFT", 4 * 8);
    int rc = launch(batch, FIT_MODE_FULL, cap > 0 ? cap : -left, (int)G, batch > (int)G ? c->num_cu : 0);  // (more fits than units: the device stays full)
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
        c->qspace_shared = true;   // (... and their q-space operands in the context's buffers: fh_fit_normal_batched prepared them)
        int rcs = FH_OK;
        for (int i = 0; i < Kc && rcs == FH_OK; ++i) {
            const int b = order[paused[i]];
            rcs = fit_submit_impl(c, alpha[b], p0[b], wsmooth[b], tol, max_iter, &tickets[i], &state[(size_t)i * RS]);
        }
        c->qspace_shared = false;
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
        // (FRANK_AMD_SWEEP_CAP: 0, the default: the fits pause when only as many are still running as the clusters of the second
        //  stage hold; n > 0: every fit pauses after n passes -- the best constant depends on the grid: 800 for one draw of
        //  BASELINE configs[4] (2 047 fits/s; 1 000: 1 914), 1 000 for another (1 963; 800: 1 592), the single launch with its
        //  sixteen longest points on clusters 1 515-1 650 on both; -1: that single launch)
        const int cap = env_int("FRANK_AMD_SWEEP_CAP", 0);
        if (cap >= 0 && gsz > 1 && batch >= 64 && c->slots_busy == 0 && c->pending_batch < 0 && !getenv("FRANK_AMD_SWEEP_NO_CLUSTERS") &&
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
    // (the fits in flight beside this launch: from ~128 resident loops on the register-resident form of the loop is the faster one,
    //  fit_loop.hip.  The slots that are out stand for the loops that run -- counting the loops exactly means a query per launch in
    //  flight, and with the threshold on that count a pipeline at steady state, ~140 running, went back and forth between the
    //  forms: 1 369 against 1 430 fits/s.  A context whose pipeline has once held 128 fits is a throughput context from then on:
    //  when it fills again after a drain its first launches do not go back to the form that works in memory, whose loops then hold
    //  their compute units for the next 0.13 s beside everything that follows -- 1 380-1 440 against 1 490-1 520 fits/s over 2 s
    //  windows.  A run of 20 or 100 fits that drains at once -- bench.py's timed region -- never gets there and keeps the form that
    //  is faster alone.)
    if (c->slots_busy >= 128) c->throughput_context = true;
    P.loaded = c->throughput_context ? (c->slots_busy > 128 ? c->slots_busy : 128) : 0;
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
            // (compute units reserved for the binning stream -- the fit loops keeping off the first B units while binning may use
            //  all -- were measured in round 5 and taken out: flat to B = 64, worse beyond; the hard partition below is worse still)
            if (c->bin_cus > 0) {  // fh_ctx_set_cu_partition: the fit loops keep to the compute units the binning pass leaves alone
                uint32_t mask[8];
                cu_mask(c->bin_cus, c->num_cu, mask);
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
    int rc = FH_OK;
    if (c->qspace_shared) {  // (a sweep: every fit has the M, j whose q-space operands the context already holds)
        HIP_TRY(hipMemcpyAsync(s.Aq.p, c->Aq.p, sizeof(double) * PP, hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(s.bq.p, c->bq.p, sizeof(double) * N, hipMemcpyDeviceToDevice, c->stream));
    } else {
        rc = prepare_qspace(c, s.Aq.p, s.bq.p);  // on the context's stream, after the finalize that produced M, j
    }
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


}  // extern "C"
