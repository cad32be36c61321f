/*
 * frank_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C, single-threaded, IEEE-fp64 CPU restatement of the visibility
 * fitting hot path of discsim/frank v1.2.3 (reference tree mounted at
 * /root/reference in the build container).  It exists so the HIP path can be
 * checked on a GPU box where the (Python) reference cannot travel.
 *
 *   * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 *     load this file's shared object.  Nothing under frank_amd/ imports it.
 *   * Parity is PINNED: tests/test_oracle_golden.py checks every function here
 *     against tests/golden/\*.npz, which tools/make_golden.py produced by
 *     importing the reference itself (numpy 2.2.6 / scipy 1.15.3) in the build
 *     container, and against the literal vectors of the reference's own
 *     data-free tests (frank/tests.py:37-130, 704-717).
 *
 * Third-party arithmetic the reference delegates to, restated here from the
 * published algorithms (reference call sites in brackets):
 *   - scipy.special.j0 / j1  = Cephes 2.8 j0.c / j1.c (Moshier)   [hankel.py:23,59-60]
 *   - scipy.special.jn_zeros = specfun JYZO: Newton on J0 from spacing guesses [hankel.py:72]
 *   - scipy.linalg.cho_factor / cho_solve = LAPACK dpotrf('U') / dpotrs      [statistical_models.py:742-745,778]
 *   - scipy.sparse.linalg.spsolve on the pentadiagonal (T + I) = banded LU   [filter.py:175]
 *   - scipy.linalg.lu_factor / lu_solve = LAPACK dgetrf / dgetrs             [minimizer.py:238]
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off: no FMA contraction, so
 * every product and sum rounds exactly as the NumPy expression it restates).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define FO_OK 0
#define FO_ERR_QRANGE -2      /* statistical_models.py:526-535 ValueError          */
#define FO_ERR_BAD_P -3       /* statistical_models.py:688-698 ValueError          */
#define FO_ERR_NOT_SPD -4     /* scipy.linalg.LinAlgError -> SVD fallback taken    */
#define FO_ERR_NOMEM -5

static const double FO_PI = 3.141592653589793238462643383279502884;
/* frank/constants.py:23-25 */
static double fo_rad_to_arcsec(void) { return 3600.0 * 180.0 / FO_PI; }
static double fo_deg_to_rad(void) { return FO_PI / 180.0; }

/* ------------------------------------------------------------------------ */
/* Cephes j0 / j1 (what scipy.special.j0/j1 call)                            */
/* ------------------------------------------------------------------------ */
static double polevl(double x, const double *c, int n) {
    double a = c[0];
    for (int i = 1; i <= n; i++) a = a * x + c[i];
    return a;
}
static double p1evl(double x, const double *c, int n) {
    double a = x + c[0];
    for (int i = 1; i < n; i++) a = a * x + c[i];
    return a;
}
static const double J0_PP[7] = {7.96936729297347051624E-4, 8.28352392107440799803E-2, 1.23953371646414299388E0,
                                5.44725003058768775090E0,  8.74716500199817011941E0,  5.30324038235394892183E0,
                                9.99999999999999997821E-1};
static const double J0_PQ[7] = {9.24408810558863637013E-4, 8.56288474354474431428E-2, 1.25352743901058953537E0,
                                5.47097740330417105182E0,  8.76190883237069594232E0,  5.30605288235394617618E0,
                                1.00000000000000000218E0};
static const double J0_QP[8] = {-1.13663838898469149931E-2, -1.28252718670509318512E0, -1.95539544257735972385E1,
                                -9.32060152123768231369E1,  -1.77681167980488050595E2, -1.47077505154951170175E2,
                                -5.14105326766599330220E1,  -6.05014350600728481186E0};
static const double J0_QQ[7] = {6.43178256118178023184E1, 8.56430025976980587198E2, 3.88240183605401609683E3,
                                7.24046774195652478189E3, 5.93072701187316984827E3, 2.06209331660327847417E3,
                                2.42005740240291393179E2};
static const double J0_DR1 = 5.78318596294678452118E0, J0_DR2 = 3.04712623436620863991E1;
static const double J0_RP[4] = {-4.79443220978201773821E9, 1.95617491946556577543E12, -2.49248344360967716204E14,
                                9.70862251047306323952E15};
static const double J0_RQ[8] = {4.99563147152651017219E2,  1.73785401676374683123E5,  4.84409658339962045305E7,
                                1.11855537045356834862E10, 2.11277520115489217587E12, 3.10518229857422583814E14,
                                3.18121955943204943306E16, 1.71086294081043136091E18};

double fo_j0(double x) {
    double w, z, p, q, xn;
    if (x < 0) x = -x;
    if (x <= 5.0) {
        z = x * x;
        if (x < 1.0e-5) return 1.0 - z / 4.0;
        p = (z - J0_DR1) * (z - J0_DR2);
        p = p * polevl(z, J0_RP, 3) / p1evl(z, J0_RQ, 8);
        return p;
    }
    w = 5.0 / x;
    q = 25.0 / (x * x);
    p = polevl(q, J0_PP, 6) / polevl(q, J0_PQ, 6);
    q = polevl(q, J0_QP, 7) / p1evl(q, J0_QQ, 7);
    xn = x - 7.85398163397448309616E-1;
    p = p * cos(xn) - w * q * sin(xn);
    return p * 7.9788456080286535587989E-1 / sqrt(x);
}

static const double J1_RP[4] = {-8.99971225705559398224E8, 4.52228297998194034323E11, -7.27494245221818276015E13,
                                3.68295732863852883286E15};
static const double J1_RQ[8] = {6.20836478118054335476E2,  2.56987256757748830383E5,  8.35146791431949253037E7,
                                2.21511595479792499675E10, 4.74914122079991414898E12, 7.84369607876235854894E14,
                                8.95222336184627338078E16, 5.32278620332680085395E18};
static const double J1_PP[7] = {7.62125616208173112003E-4, 7.31397056940917570436E-2, 1.12719608129684925192E0,
                                5.11207951146807644818E0,  8.42404590141772420927E0,  5.21451598682361504063E0,
                                1.00000000000000000254E0};
static const double J1_PQ[7] = {5.71323128072548699714E-4, 6.88455908754495404082E-2, 1.10514232634061696926E0,
                                5.07386386128601488557E0,  8.39985554327604159757E0,  5.20982848682361821619E0,
                                9.99999999999999997461E-1};
static const double J1_QP[8] = {5.10862594750176621635E-2, 4.98213872951233449420E0, 7.58238284132545283818E1,
                                3.66779609360150777800E2,  7.10856304998926107277E2, 5.97489612400613639965E2,
                                2.11688757100572135698E2,  2.52070205858023719784E1};
static const double J1_QQ[7] = {7.42373277035675149943E1, 1.05644886038262816351E3, 4.98641058337653607651E3,
                                9.56231892404756170795E3, 7.99704160447350683650E3, 2.82619278517639096600E3,
                                3.36093607810698293419E2};
static const double J1_Z1 = 1.46819706421238932572E1, J1_Z2 = 4.92184563216946036703E1;

double fo_j1(double x) {
    double w, z, p, q, xn;
    if (x < 0) return -fo_j1(-x);
    if (x <= 5.0) {
        z = x * x;
        w = polevl(z, J1_RP, 3) / p1evl(z, J1_RQ, 8);
        w = w * x * (z - J1_Z1) * (z - J1_Z2);
        return w;
    }
    w = 5.0 / x;
    z = w * w;
    p = polevl(z, J1_PP, 6) / polevl(z, J1_PQ, 6);
    q = polevl(z, J1_QP, 7) / p1evl(z, J1_QQ, 7);
    xn = x - 2.35619449019234492885;
    p = p * cos(xn) - w * q * sin(xn);
    return p * 7.9788456080286535587989E-1 / sqrt(x);
}

void fo_j0_array(const double *x, int64_t n, double *out) {
    for (int64_t i = 0; i < n; i++) out[i] = fo_j0(x[i]);
}

/* ------------------------------------------------------------------------ */
/* scipy.special.jn_zeros(0, nt)  [hankel.py:72]                             */
/* specfun JYZO for n = 0: first guess 2.82141, Newton x -= J0/J0' with      */
/* J0' = -J1 until |dx| <= 1e-11, next guess = last zero + 3.1416 + 0.0972/L */
/* (JYZO evaluates J0, J0' by backward recurrence; Cephes is used here, the  */
/* converged root agrees to < 1 ulp -- pinned by tests/golden/dht_N*.npz).   */
/* ------------------------------------------------------------------------ */
void fo_jn_zeros0(int nt, double *zeros) {
    double x = 2.82141;
    for (int l = 0; l < nt; l++) {
        for (int it = 0; it < 100; it++) {
            double x0 = x;
            double f = fo_j0(x), fp = -fo_j1(x);
            x = x - f / fp;
            if (fabs(x - x0) <= 1.0e-11) break;
        }
        zeros[l] = x;
        x = x + 3.1416 + 0.0972 / (double)(l + 1);
    }
}

/* ------------------------------------------------------------------------ */
/* DiscreteHankelTransform.__init__  [hankel.py:55-93], nu = 0               */
/*   Ykm is N*N row-major (C order), Ykm[k*N+m] = Ykm[k,m].                  */
/* ------------------------------------------------------------------------ */
void fo_dht_setup(double Rmax, int N, double *r, double *q, double *j_nk, double *j_nN_out, double *Qmax_out,
                  double *Ykm, double *scale_factor) {
    double *z = (double *)malloc(sizeof(double) * (size_t)(N + 1));
    fo_jn_zeros0(N + 1, z);
    double j_nN = z[N];
    double Qmax = j_nN / (2 * FO_PI * Rmax); /* hankel.py:75 */
    for (int k = 0; k < N; k++) {
        j_nk[k] = z[k];
        r[k] = Rmax * (z[k] / j_nN); /* hankel.py:77 */
        q[k] = Qmax * (z[k] / j_nN); /* hankel.py:78 */
    }
    /* hankel.py:84-87: Jnk[k,m] = J1(j_m);  meshgrid(j_nk, j_nk/j_nN) -> prod[k,m] = j_nk[m] * (j_nk[k]/j_nN) */
    for (int k = 0; k < N; k++) {
        double jk_over = z[k] / j_nN;
        for (int m = 0; m < N; m++) {
            double J1m = fo_j1(z[m]);
            double pre = 2 / (j_nN * J1m * J1m);
            Ykm[(size_t)k * N + m] = pre * fo_j0(z[m] * jk_over);
        }
    }
    for (int k = 0; k < N; k++) { /* hankel.py:89 */
        double J1k = fo_j1(z[k]);
        scale_factor[k] = 1 / (J1k * J1k);
    }
    *j_nN_out = j_nN;
    *Qmax_out = Qmax;
    free(z);
}

/* DHT.coefficients(q=None, 'forward')  [hankel.py:187-199] : Y = 0.5*j_nN*norm*Ykm */
void fo_dht_coefficients_self(int N, double j_nN, double Qmax, const double *Ykm, double *Y) {
    double norm = 1 / (FO_PI * Qmax * Qmax);
    double f = 0.5 * j_nN * norm; /* (0.5 * j_nN) * norm, left to right */
    for (size_t i = 0; i < (size_t)N * N; i++) Y[i] = f * Ykm[i];
}

/* DHT.coefficients(q, 'forward')  [hankel.py:187-204]: H[i,k] = (norm*sf_k) * j0((k*q_i) * j_nk), k = 1/Qmax */
void fo_dht_coefficients(int N, double Qmax, const double *j_nk, const double *scale_factor, const double *qs,
                         int64_t n, double *H) {
    double norm = 1 / (FO_PI * Qmax * Qmax);
    double kq = 1. / Qmax;
    for (int64_t i = 0; i < n; i++) {
        double s = kq * qs[i];
        for (int k = 0; k < N; k++) H[(size_t)i * N + k] = (norm * scale_factor[k]) * fo_j0(s * j_nk[k]);
    }
}

/* DHT.transform(f, q=None, 'forward')  [hankel.py:151-165] */
void fo_dht_transform_forward(int N, double Rmax, double j_nN, const double *Ykm, const double *f, double *out) {
    double norm = (2 * FO_PI * Rmax * Rmax) / j_nN;
    for (int k = 0; k < N; k++) {
        double a = 0;
        for (int m = 0; m < N; m++) a += Ykm[(size_t)k * N + m] * f[m];
        out[k] = norm * a;
    }
}

/* ------------------------------------------------------------------------ */
/* geometry: apply_phase_shift(inverse=True) + deproject                     */
/* [geometry.py:69-79, 111-131, 202-236]                                     */
/* NumPy complex division = Smith's algorithm (numpy loops.c.src, CDOUBLE    */
/* divide); only Re(V') is consumed downstream (statistical_models.py:172).  */
/* ------------------------------------------------------------------------ */
void fo_apply_correction(int64_t n, const double *u, const double *v, const double *Vre, const double *Vim,
                         double inc_deg, double PA_deg, double dRA_arcsec, double dDec_arcsec, double *up,
                         double *vp, double *wp, double *Vpre, double *Vpim) {
    double dRA = dRA_arcsec * (2. * FO_PI / fo_rad_to_arcsec()); /* geometry.py:69: dRA *= 2*pi/rad_to_arcsec */
    double dDec = dDec_arcsec * (2. * FO_PI / fo_rad_to_arcsec());
    double inc = inc_deg * fo_deg_to_rad();
    double PA = PA_deg * fo_deg_to_rad();
    double cos_t = cos(PA), sin_t = sin(PA);
    double cos_i = cos(inc), sin_i = sin(inc);
    for (int64_t i = 0; i < n; i++) {
        double phi = u[i] * dRA + v[i] * dDec;
        double c = cos(phi), s = sin(phi);
        double ar = Vre[i], ai = Vim ? Vim[i] : 0.0;
        double outr, outi;
        if (fabs(c) >= fabs(s)) {
            double rat = s / c, scl = 1.0 / (c + s * rat);
            outr = (ar + ai * rat) * scl;
            outi = (ai - ar * rat) * scl;
        } else {
            double rat = c / s, scl = 1.0 / (s + c * rat);
            outr = (ar * rat + ai) * scl;
            outi = (ai * rat - ar) * scl;
        }
        if (Vpre) Vpre[i] = outr;
        if (Vpim) Vpim[i] = outi;
        double upp = u[i] * cos_t - v[i] * sin_t; /* geometry.py:121 */
        double vpp = u[i] * sin_t + v[i] * cos_t; /* geometry.py:122 */
        if (wp) wp[i] = upp * sin_i;              /* geometry.py:128 */
        up[i] = upp * cos_i;                      /* geometry.py:129 */
        vp[i] = vpp;
    }
}

/* ------------------------------------------------------------------------ */
/* VisibilityMapping.map_visibilities  [statistical_models.py:109-237]       */
/* single channel; vis_model: 0 = opt_thick (scale = cos inc), 1 = opt_thin  */
/* weights may be a scalar (n_w == 1) as at statistical_models.py:173.       */
/* Returns FO_ERR_QRANGE when q_k[-1] < max(q) and check_qbounds != 0.       */
/* ------------------------------------------------------------------------ */
/* vis_model 2 = 'debris' (optically thin, geometrically thick): scale[i,k] = exp(-kz_i^2 * H2[k]) with kz the third    */
/* deprojected coordinate and H2 = 0.5 (2 pi H(r_k) / rad_to_arcsec)^2  [:101-102, :494-496]; H2 is ignored otherwise. */
int fo_map_visibilities_ex(int N, double Rmax, double inc_deg, double PA_deg, double dRA, double dDec, int vis_model,
                           int check_qbounds, int64_t block_size, int64_t n, const double *u, const double *v,
                           const double *Vre, const double *Vim, const double *w, int64_t n_w, const double *H2,
                           double *M, double *j, double *H0_out, double *qmin_out, double *qmax_out) {
    double *r = malloc(sizeof(double) * N), *qk = malloc(sizeof(double) * N), *jnk = malloc(sizeof(double) * N);
    double *Ykm = malloc(sizeof(double) * (size_t)N * N), *sf = malloc(sizeof(double) * N);
    double j_nN, Qmax;
    fo_dht_setup(Rmax, N, r, qk, jnk, &j_nN, &Qmax, Ykm, sf);
    free(Ykm);

    double *up = malloc(sizeof(double) * n), *vp = malloc(sizeof(double) * n), *Vp = malloc(sizeof(double) * n);
    double *q = malloc(sizeof(double) * n), *kz = malloc(sizeof(double) * n);
    if (!up || !vp || !Vp || !q || !kz) return FO_ERR_NOMEM;
    fo_apply_correction(n, u, v, Vre, Vim, inc_deg, PA_deg, dRA, dDec, up, vp, kz, Vp, NULL);
    double qmin = INFINITY, qmax = -INFINITY;
    for (int64_t i = 0; i < n; i++) {
        q[i] = hypot(up[i], vp[i]); /* statistical_models.py:166 */
        if (q[i] < qmin) qmin = q[i];
        if (q[i] > qmax) qmax = q[i];
    }
    if (qmin_out) *qmin_out = qmin;
    if (qmax_out) *qmax_out = qmax;
    int rc = FO_OK;
    if (check_qbounds && n > 0 && qk[N - 1] < qmax) rc = FO_ERR_QRANGE; /* statistical_models.py:526 */

    if (rc == FO_OK) {
        double scale = (vis_model == 0) ? cos(inc_deg * fo_deg_to_rad()) : 1.0; /* :486-493 */
        int64_t Nstep = (int64_t)((double)block_size / (double)N + 1);          /* :193 */
        double *X = malloc(sizeof(double) * (size_t)Nstep * N);
        double *wXT = malloc(sizeof(double) * (size_t)Nstep * N);
        double *Mc = malloc(sizeof(double) * (size_t)N * N), *jc = malloc(sizeof(double) * N);
        memset(M, 0, sizeof(double) * (size_t)N * N);
        memset(j, 0, sizeof(double) * N);
        double norm = 1 / (FO_PI * Qmax * Qmax), kq = 1. / Qmax;
        for (int64_t start = 0; start < n; start += Nstep) {
            int64_t m = (start + Nstep <= n) ? Nstep : n - start;
            for (int64_t i = 0; i < m; i++) { /* :206 -> :483-509 -> hankel.py:201-202 */
                double s = kq * q[start + i];
                double wi = (n_w == 1) ? w[0] : w[start + i];
                wi = 1.0 * wi; /* np.ones_like(V) * weights */
                double kz2 = kz[start + i] * kz[start + i];
                for (int k = 0; k < N; k++) {
                    if (vis_model == 2) scale = exp(-(kz2 * H2[k])); /* np.exp(-np.outer(ks*ks, H2)), :496 */
                    double h = ((norm * sf[k]) * fo_j0(s * jnk[k])) * scale;
                    X[(size_t)i * N + k] = h;
                    wXT[(size_t)i * N + k] = h * wi; /* :208 */
                }
            }
            memset(Mc, 0, sizeof(double) * (size_t)N * N);
            memset(jc, 0, sizeof(double) * N);
            for (int64_t i = 0; i < m; i++) { /* :210-211, np.dot(wXT, X) and np.dot(wXT, Vs) */
                const double *xi = X + (size_t)i * N, *wxi = wXT + (size_t)i * N;
                double Vi = Vp[start + i];
                for (int k = 0; k < N; k++) {
                    double a = wxi[k];
                    double *row = Mc + (size_t)k * N;
                    for (int l = 0; l < N; l++) row[l] += a * xi[l];
                    jc[k] += a * Vi;
                }
            }
            for (size_t e = 0; e < (size_t)N * N; e++) M[e] += Mc[e];
            for (int k = 0; k < N; k++) j[k] += jc[k];
        }
        free(X); free(wXT); free(Mc); free(jc);
        /* :218  H0 = 0.5*sum(log(w/(2 pi)) - V*w*V) */
        double acc = 0;
        for (int64_t i = 0; i < n; i++) {
            double wi = (n_w == 1) ? w[0] : w[i];
            acc += log(wi / (2 * FO_PI)) - Vp[i] * wi * Vp[i];
        }
        *H0_out = 0.5 * acc;
    }
    free(r); free(qk); free(jnk); free(sf); free(up); free(vp); free(Vp); free(q); free(kz);
    return rc;
}

int fo_map_visibilities(int N, double Rmax, double inc_deg, double PA_deg, double dRA, double dDec, int vis_model,
                        int check_qbounds, int64_t block_size, int64_t n, const double *u, const double *v,
                        const double *Vre, const double *Vim, const double *w, int64_t n_w, double *M, double *j,
                        double *H0_out, double *qmin_out, double *qmax_out) {
    return fo_map_visibilities_ex(N, Rmax, inc_deg, PA_deg, dRA, dDec, vis_model, check_qbounds, block_size, n, u, v, Vre,
                                  Vim, w, n_w, NULL, M, j, H0_out, qmin_out, qmax_out);
}

/* ------------------------------------------------------------------------ */
/* LAPACK-style dense helpers (row-major, upper Cholesky as cho_factor does) */
/* ------------------------------------------------------------------------ */
/* A = U^T U, U upper, stored in the upper triangle of A (row-major). Returns 0 or k+1 of the failing pivot. */
int fo_cholesky_upper(int n, double *A) {
    for (int k = 0; k < n; k++) {
        double d = A[(size_t)k * n + k];
        for (int s = 0; s < k; s++) d -= A[(size_t)s * n + k] * A[(size_t)s * n + k];
        if (!(d > 0.0)) return k + 1;
        d = sqrt(d);
        A[(size_t)k * n + k] = d;
        for (int c = k + 1; c < n; c++) {
            double t = A[(size_t)k * n + c];
            for (int s = 0; s < k; s++) t -= A[(size_t)s * n + k] * A[(size_t)s * n + c];
            A[(size_t)k * n + c] = t / d;
        }
    }
    return 0;
}
/* Solve U^T U x = b for nrhs right-hand sides; B is n x nrhs row-major, overwritten. */
void fo_cho_solve_upper(int n, const double *U, double *B, int nrhs) {
    for (int i = 0; i < n; i++) { /* U^T y = b */
        double *bi = B + (size_t)i * nrhs;
        for (int s = 0; s < i; s++) {
            double usi = U[(size_t)s * n + i];
            const double *bs = B + (size_t)s * nrhs;
            for (int c = 0; c < nrhs; c++) bi[c] -= usi * bs[c];
        }
        double d = U[(size_t)i * n + i];
        for (int c = 0; c < nrhs; c++) bi[c] /= d;
    }
    for (int i = n - 1; i >= 0; i--) { /* U x = y */
        double *bi = B + (size_t)i * nrhs;
        for (int s = i + 1; s < n; s++) {
            double uis = U[(size_t)i * n + s];
            const double *bs = B + (size_t)s * nrhs;
            for (int c = 0; c < nrhs; c++) bi[c] -= uis * bs[c];
        }
        double d = U[(size_t)i * n + i];
        for (int c = 0; c < nrhs; c++) bi[c] /= d;
    }
}

/* One-sided Jacobi SVD, A = U diag(s) V^T with the singular values in DESCENDING order (LAPACK's convention, which
 * the broadcasting quirk below makes observable).  Us = U diag(s) (n*n), V (n*n), s2[k] = s_k^2. */
static void fo_svd_jacobi(int n, const double *A, double *Us, double *V, double *s2) {
    double *W = malloc(sizeof(double) * (size_t)n * n), *Vw = malloc(sizeof(double) * (size_t)n * n);
    memcpy(W, A, sizeof(double) * (size_t)n * n);
    for (int i = 0; i < n; i++)
        for (int k = 0; k < n; k++) Vw[(size_t)i * n + k] = (i == k);
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0;
        for (int p = 0; p < n - 1; p++)
            for (int q = p + 1; q < n; q++) {
                double al = 0, be = 0, ga = 0;
                for (int i = 0; i < n; i++) {
                    double a = W[(size_t)i * n + p], c = W[(size_t)i * n + q];
                    al += a * a; be += c * c; ga += a * c;
                }
                if (ga == 0 || fabs(ga) <= 1e-17 * sqrt(al * be)) continue;
                off += fabs(ga) / sqrt(al * be);
                double zeta = (be - al) / (2 * ga);
                double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1 + zeta * zeta));
                double c = 1 / sqrt(1 + t * t), sn = c * t;
                for (int i = 0; i < n; i++) {
                    double a = W[(size_t)i * n + p], d = W[(size_t)i * n + q];
                    W[(size_t)i * n + p] = c * a - sn * d; W[(size_t)i * n + q] = sn * a + c * d;
                    a = Vw[(size_t)i * n + p]; d = Vw[(size_t)i * n + q];
                    Vw[(size_t)i * n + p] = c * a - sn * d; Vw[(size_t)i * n + q] = sn * a + c * d;
                }
            }
        if (off < 1e-15) break;
    }
    int *ord = malloc(sizeof(int) * n);
    double *raw = malloc(sizeof(double) * n);
    for (int k = 0; k < n; k++) {
        double t = 0;
        for (int i = 0; i < n; i++) t += W[(size_t)i * n + k] * W[(size_t)i * n + k];
        raw[k] = t;
        ord[k] = k;
    }
    for (int a = 1; a < n; a++) { /* insertion sort, descending, stable */
        int o = ord[a], b = a - 1;
        while (b >= 0 && raw[ord[b]] < raw[o]) { ord[b + 1] = ord[b]; b--; }
        ord[b + 1] = o;
    }
    for (int k = 0; k < n; k++) {
        s2[k] = raw[ord[k]];
        for (int i = 0; i < n; i++) {
            Us[(size_t)i * n + k] = W[(size_t)i * n + ord[k]];
            V[(size_t)i * n + k] = Vw[(size_t)i * n + ord[k]];
        }
    }
    free(W); free(Vw); free(ord); free(raw);
}

/* The reference's route when cho_factor raises LinAlgError [statistical_models.py:747-755, 779-781]:
 *   U, s, V = svd(Dinv);  s1 = where(s > 0, 1/s, 0);  x = V^T ((U^T b) * s1).
 * B: n*nrhs row-major, overwritten.  `*` is NumPy's element-wise product: s1 broadcasts over the LAST axis.  For a
 * vector b that is diag(s1) (U^T b), the pseudo-inverse solve.  For the N x N right-hand side of
 * CriticalFilter.update_power_spectrum (fit.Dsolve(Ykm.T), filter.py:168) it multiplies COLUMN c of U^T b by s1[c]
 * -- not row k by s1[k] -- and that is what the reference's loop then iterates on; `as_reference` != 0 reproduces it
 * (nrhs must be n), 0 gives the pseudo-inverse solve for every column. */
void fo_svd_solve(int n, const double *A, double *B, int nrhs, int as_reference) {
    size_t nn = (size_t)n * n;
    double *Us = malloc(sizeof(double) * nn), *V = malloc(sizeof(double) * nn), *s2 = malloc(sizeof(double) * n);
    double *T = malloc(sizeof(double) * (size_t)n * nrhs);
    fo_svd_jacobi(n, A, Us, V, s2);
    for (int k = 0; k < n; k++) { /* T = U^T B, U[:,k] = Us[:,k] / s_k */
        double sk = sqrt(s2[k]);
        for (int c = 0; c < nrhs; c++) {
            double a = 0;
            for (int i = 0; i < n; i++) a += Us[(size_t)i * n + k] * B[(size_t)i * nrhs + c];
            T[(size_t)k * nrhs + c] = sk > 0 ? a / sk : 0.0;
        }
    }
    for (int k = 0; k < n; k++)
        for (int c = 0; c < nrhs; c++) {
            double sv = sqrt(s2[(as_reference && nrhs > 1) ? c : k]);
            T[(size_t)k * nrhs + c] *= sv > 0 ? 1.0 / sv : 0.0;
        }
    for (int i = 0; i < n; i++) /* x = V T  (scipy's V is V^T here) */
        for (int c = 0; c < nrhs; c++) {
            double a = 0;
            for (int k = 0; k < n; k++) a += V[(size_t)i * n + k] * T[(size_t)k * nrhs + c];
            B[(size_t)i * nrhs + c] = a;
        }
    free(Us); free(V); free(s2); free(T);
}

static void fo_svd_pinv_solve(int n, const double *A, const double *b, double *x) {
    memcpy(x, b, sizeof(double) * n);
    fo_svd_solve(n, A, x, 1, 0);
}

/* ------------------------------------------------------------------------ */
/* GaussianModel.__init__ + _fit  [statistical_models.py:650-760], Nfields=1 */
/*   Sinv = einsum('ji,lj,jk->lik', Y, 1/p, Y)  (:700-701);  Dinv = M + Sinv */
/*   chol (upper) is returned in `chol` (N*N, upper triangle valid); p may   */
/*   be NULL (no prior, as FourierBesselFitter._fit, radial_fitters.py:576). */
/* Returns FO_OK, FO_ERR_BAD_P, or FO_ERR_NOT_SPD (mu then from the SVD      */
/* pseudo-inverse; `chol` then holds Dinv itself, for the SVD-route Dsolve). */
/* ------------------------------------------------------------------------ */
int fo_gaussian_model(int N, const double *Y, const double *M, const double *j, const double *p, double *mu,
                      double *chol, double *Sinv_out) {
    size_t NN = (size_t)N * N;
    double *Dinv = chol;
    if (p) {
        for (int k = 0; k < N; k++)
            if (!(p[k] > 0.0)) return FO_ERR_BAD_P; /* :689 (catches NaN too) */
        double *pinv = malloc(sizeof(double) * N);
        for (int k = 0; k < N; k++) pinv[k] = 1 / p[k];
        double *S = Sinv_out ? Sinv_out : malloc(sizeof(double) * NN);
        memset(S, 0, sizeof(double) * NN);
        for (int jj = 0; jj < N; jj++) { /* sum over j of Y[j,i] * pinv[j] * Y[j,k] */
            const double *yj = Y + (size_t)jj * N;
            for (int i = 0; i < N; i++) {
                double a = yj[i] * pinv[jj];
                double *row = S + (size_t)i * N;
                for (int k = 0; k < N; k++) row[k] += a * yj[k];
            }
        }
        for (size_t e = 0; e < NN; e++) Dinv[e] = M[e] + S[e]; /* :739 */
        if (!Sinv_out) free(S);
        free(pinv);
    } else {
        memcpy(Dinv, M, sizeof(double) * NN);
    }
    double *keep = malloc(sizeof(double) * NN);
    memcpy(keep, Dinv, sizeof(double) * NN);
    int info = fo_cholesky_upper(N, Dinv);
    if (info == 0) {
        memcpy(mu, j, sizeof(double) * N);
        fo_cho_solve_upper(N, Dinv, mu, 1);
        free(keep);
        return FO_OK;
    }
    fo_svd_pinv_solve(N, keep, j, mu);
    memcpy(chol, keep, sizeof(double) * NN); /* the SVD route works on Dinv itself: hand it back */
    free(keep);
    return FO_ERR_NOT_SPD;
}

/* ------------------------------------------------------------------------ */
/* spectral_smoothing_matrix  [filter.py:23-62] as 5 bands:                  */
/*   T[i, i+d] = band[(d+2)*N + i], d = -2..2 (entries outside are 0).       */
/* ------------------------------------------------------------------------ */
void fo_smoothing_matrix(int N, const double *q, double weights, double *band) {
    double *lq = malloc(sizeof(double) * N), *dc = calloc(N, sizeof(double)), *de = calloc(N, sizeof(double));
    double *D0 = calloc(N, sizeof(double)), *D1 = calloc(N, sizeof(double)), *D2 = calloc(N, sizeof(double));
    for (int i = 0; i < N; i++) lq[i] = log(q[i]);
    for (int i = 0; i + 2 < N; i++) dc[i] = (lq[i + 2] - lq[i]) / 2; /* dc[i] pairs with row i+1 */
    for (int i = 0; i + 1 < N; i++) de[i] = lq[i + 1] - lq[i];
    /* Delta rows 1..N-2:  sub D0[i] = Delta[i,i-1], diag D1[i], super D2[i] = Delta[i,i+1] */
    for (int i = 1; i + 1 < N; i++) {
        D0[i] = 1 / (dc[i - 1] * de[i - 1]);
        D1[i] = -(1 / de[i] + 1 / de[i - 1]) / dc[i - 1];
        D2[i] = 1 / (dc[i - 1] * de[i]);
    }
    memset(band, 0, sizeof(double) * 5 * (size_t)N);
    /* T = Delta^T (dce Delta):  T[a,b] = sum_i Delta[i,a] * (dce[i] * Delta[i,b]) */
    for (int i = 1; i + 1 < N; i++) {
        double dce = dc[i - 1];
        int cols[3] = {i - 1, i, i + 1};
        double vals[3] = {D0[i], D1[i], D2[i]};
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) {
                int d = cols[b] - cols[a];
                band[(size_t)(d + 2) * N + cols[a]] += vals[a] * (dce * vals[b]);
            }
    }
    for (size_t e = 0; e < 5 * (size_t)N; e++) band[e] *= weights;
    free(lq); free(dc); free(de); free(D0); free(D1); free(D2);
}

/* Solve (T + I) x = rhs for pentadiagonal T (banded LU without pivoting; T + I is SPD).  [filter.py:157,175] */
void fo_penta_solve(int N, const double *band, const double *rhs, double *x) {
    /* dense-in-band working copy: A[i][d+2] */
    double *A = malloc(sizeof(double) * 5 * (size_t)N);
    for (int i = 0; i < N; i++)
        for (int d = -2; d <= 2; d++) {
            double t = band[(size_t)(d + 2) * N + i];
            if (d == 0) t += 1.0;
            A[(size_t)i * 5 + (d + 2)] = t;
        }
    for (int i = 0; i < N; i++) x[i] = rhs[i];
    for (int k = 0; k < N; k++) {
        double piv = A[(size_t)k * 5 + 2];
        for (int i = k + 1; i <= k + 2 && i < N; i++) {
            int dk = k - i; /* -1 or -2 */
            double f = A[(size_t)i * 5 + (dk + 2)] / piv;
            if (f == 0) continue;
            for (int c = k + 1; c <= k + 2 && c < N; c++) {
                int di = c - i, dkk = c - k;
                A[(size_t)i * 5 + (di + 2)] -= f * A[(size_t)k * 5 + (dkk + 2)];
            }
            A[(size_t)i * 5 + (dk + 2)] = 0;
            x[i] -= f * x[k];
        }
    }
    for (int i = N - 1; i >= 0; i--) {
        double t = x[i];
        for (int c = i + 1; c <= i + 2 && c < N; c++) t -= A[(size_t)i * 5 + (c - i + 2)] * x[c];
        x[i] = t / A[(size_t)i * 5 + 2];
    }
    free(A);
}

/* ------------------------------------------------------------------------ */
/* CriticalFilter.update_power_spectrum  [filter.py:154-177]                 */
/*   needs the posterior of the current fit: mu, upper Cholesky of Dinv.     */
/* ------------------------------------------------------------------------ */
void fo_update_power_spectrum_ex(int N, const double *Y, const double *band, double alpha, double p0, const double *p,
                                 const double *mu, const double *chol, int svd_route, double *p_new) {
    size_t NN = (size_t)N * N;
    double *Tr1 = malloc(sizeof(double) * N), *Tr2 = malloc(sizeof(double) * N);
    double *Z = malloc(sizeof(double) * NN), *rhs = malloc(sizeof(double) * N), *tau = malloc(sizeof(double) * N);
    for (int i = 0; i < N; i++) { /* :162 */
        double a = 0;
        for (int k = 0; k < N; k++) a += Y[(size_t)i * N + k] * mu[k];
        Tr1[i] = a * a;
    }
    for (int i = 0; i < N; i++) /* Z = Y^T */
        for (int k = 0; k < N; k++) Z[(size_t)k * N + i] = Y[(size_t)i * N + k];
    if (svd_route) fo_svd_solve(N, chol, Z, N, 1); /* Dsolve(Ykm.T) of a fit whose Cholesky failed: chol = Dinv */
    else fo_cho_solve_upper(N, chol, Z, N);      /* Dsolve(Ykm.T), :168 */
    for (int i = 0; i < N; i++) {
        double a = 0;
        for (int k = 0; k < N; k++) a += Y[(size_t)i * N + k] * Z[(size_t)k * N + i];
        Tr2[i] = a;
    }
    const double rho = 1.0;
    for (int i = 0; i < N; i++) { /* :172-175 */
        double beta = (p0 + 0.5 * (Tr1[i] + Tr2[i])) / p[i] - (alpha - 1.0 + 0.5 * rho);
        rhs[i] = beta + log(p[i]);
    }
    fo_penta_solve(N, band, rhs, tau);
    for (int i = 0; i < N; i++) p_new[i] = exp(tau[i]); /* :177 */
    free(Tr1); free(Tr2); free(Z); free(rhs); free(tau);
}

void fo_update_power_spectrum(int N, const double *Y, const double *band, double alpha, double p0, const double *p,
                              const double *mu, const double *chol, double *p_new) {
    fo_update_power_spectrum_ex(N, Y, band, alpha, p0, p, mu, chol, 0, p_new);
}

/* CriticalFilter.check_convergence  [filter.py:179-181] */
int fo_check_convergence(int N, const double *p_new, const double *p_old, double tol) {
    for (int i = 0; i < N; i++)
        if (!(fabs(p_new[i] - p_old[i]) <= tol * p_new[i])) return 0;
    return 1;
}

/* ------------------------------------------------------------------------ */
/* FrankFitter._fit, method='Normal'  [radial_fitters.py:737-832]            */
/*   diag_p / diag_mu: optional (max_iter+1)*N buffers receiving pI and MAP  */
/*   of every loop pass (store_iteration_diagnostics, :781-783).             */
/*   Returns FO_OK, FO_ERR_BAD_P; *niter = `count` at loop exit; the caller  */
/*   applies the convergence_failure policy (count < max_iter is success).   */
/* ------------------------------------------------------------------------ */
int fo_frank_fit_normal(int N, double Rmax, const double *M, const double *j, double alpha, double p0,
                        double wsmooth, double tol, int max_iter, double *mu_out, double *p_out, int *niter,
                        double *diag_p, double *diag_mu, int *n_svd_fallbacks) {
    size_t NN = (size_t)N * N;
    double *r = malloc(sizeof(double) * N), *q = malloc(sizeof(double) * N), *jnk = malloc(sizeof(double) * N);
    double *Ykm = malloc(sizeof(double) * NN), *sf = malloc(sizeof(double) * N), *Y = malloc(sizeof(double) * NN);
    double *band = malloc(sizeof(double) * 5 * N), *chol = malloc(sizeof(double) * NN);
    double *pI = malloc(sizeof(double) * N), *pold = malloc(sizeof(double) * N), *mu = malloc(sizeof(double) * N);
    double *tmp = malloc(sizeof(double) * N);
    double j_nN, Qmax;
    int rc, nsvd = 0;
    fo_dht_setup(Rmax, N, r, q, jnk, &j_nN, &Qmax, Ykm, sf);
    fo_dht_coefficients_self(N, j_nN, Qmax, Ykm, Y);
    fo_smoothing_matrix(N, q, wsmooth, band);

    for (int k = 0; k < N; k++) pI[k] = 1.0; /* :744 */
    rc = fo_gaussian_model(N, Y, M, j, pI, mu, chol, NULL); /* :747 */
    if (rc == FO_ERR_NOT_SPD) { nsvd++; rc = FO_OK; }
    if (rc != FO_OK) goto done;
    fo_dht_transform_forward(N, Rmax, j_nN, Ykm, mu, tmp); /* :749 */
    double pmax = -INFINITY;
    for (int k = 0; k < N; k++) { double t = tmp[k] * tmp[k]; if (t > pmax) pmax = t; }
    for (int k = 0; k < N; k++) pI[k] = pmax * pow(q[k] / q[0], -2.0); /* :750 */
    rc = fo_gaussian_model(N, Y, M, j, pI, mu, chol, NULL); /* :752 */
    int svd_now = 0; /* the current fit went through the SVD (its Dsolve does too, :779-781) */
    if (rc == FO_ERR_NOT_SPD) { nsvd++; svd_now = 1; rc = FO_OK; }
    if (rc != FO_OK) goto done;

    int count = 0;
    for (int k = 0; k < N; k++) pold[k] = 0.0; /* :768 pi_old = 0 */
    while (!fo_check_convergence(N, pI, pold, tol) && count <= max_iter) { /* :769-770 */
        memcpy(pold, pI, sizeof(double) * N);
        fo_update_power_spectrum_ex(N, Y, band, alpha, p0, pold, mu, chol, svd_now, pI); /* :777 */
        rc = fo_gaussian_model(N, Y, M, j, pI, mu, chol, NULL);                            /* :779 */
        svd_now = 0;
        if (rc == FO_ERR_NOT_SPD) { nsvd++; svd_now = 1; rc = FO_OK; }
        if (rc != FO_OK) goto done;
        if (diag_p) memcpy(diag_p + (size_t)count * N, pI, sizeof(double) * N);
        if (diag_mu) memcpy(diag_mu + (size_t)count * N, mu, sizeof(double) * N);
        count++;
    }
    *niter = count;
    memcpy(mu_out, mu, sizeof(double) * N);
    memcpy(p_out, pI, sizeof(double) * N);
done:
    if (n_svd_fallbacks) *n_svd_fallbacks = nsvd;
    free(r); free(q); free(jnk); free(Ykm); free(sf); free(Y); free(band); free(chol);
    free(pI); free(pold); free(mu); free(tmp);
    return rc;
}
