// Host-side DHT set-up: Bessel zeros, collocation points, Ykm, scale factors.
//
// Reference: DiscreteHankelTransform.__init__ (hankel.py:55-93), which calls
// scipy.special.jn_zeros / j0 / j1.  Here the N+1 zeros of J0 and J1 at those zeros come
// from an 80-bit Miller backward recurrence + Newton (correctly rounded to fp64 in practice,
// hence within 1 ulp of SciPy's, tests/test_host_dht.py), and the N*N kernel matrix uses
// the same fp64 J0 as the GPU kernels (bessel.h).
#include "dht_host.h"

#include <cmath>

#include "../../include/frank_hip.h"
#include "bessel.h"
#include "j0_zeros_table.h"

namespace {

const long double PI_L = 3.14159265358979323846264338327950288L;

// J0(x), J1(x) by Miller's backward recurrence, normalised with 1 = J0 + 2 sum J_2k.
void bessel_j01_miller(long double x, long double *j0, long double *j1) {
    if (x == 0.0L) {
        *j0 = 1.0L;
        *j1 = 0.0L;
        return;
    }
    int M = (int)(x + 40.0L + 12.0L * cbrtl(x));
    M += (M & 1);  // even
    long double jp1 = 0.0L, jn = 1e-300L, sum = 0.0L;
    const long double tox = 2.0L / x;
    for (int n = M; n >= 1; --n) {
        // J_{n-1} = (2n/x) J_n - J_{n+1}
        long double jm1 = (long double)n * tox * jn - jp1;
        jp1 = jn;
        jn = jm1;
        if (((n - 1) & 1) == 0 && n - 1 > 0) sum += jn;  // even orders 2, 4, ...
        if (fabsl(jn) > 1e3000L) {
            jn *= 1e-3000L;
            jp1 *= 1e-3000L;
            sum *= 1e-3000L;
        }
    }
    // now jn = J0 (unnormalised), jp1 = J1
    long double norm = jn + 2.0L * sum;
    *j0 = jn / norm;
    *j1 = jp1 / norm;
}

}  // namespace

int fh_dht_build(double Rmax_rad, int N, fh_dht *out) {
    if (N < 1 || !(Rmax_rad > 0)) return FH_ERR_INVALID;
    fh_dht &d = *out;
    d.N = N;
    d.nu = 0;
    d.Rmax = Rmax_rad;
    d.zeros.resize(N + 1);
    std::vector<long double> J1z(N + 1);
    for (int k = 1; k <= N + 1; ++k) {
        // McMahon start, then Newton on J0 with J0' = -J1
        long double b = (k - 0.25L) * PI_L;
        long double x = b + 1.0L / (8.0L * b) - 124.0L / (3.0L * 512.0L * b * b * b);
        for (int it = 0; it < 8; ++it) {
            long double f, g;
            bessel_j01_miller(x, &f, &g);
            long double dx = f / g;  // x_new = x - J0/J0' = x + J0/J1
            x += dx;
            if (fabsl(dx) < 1e-19L * x) break;
        }
        // the reference's grid is built from SciPy's zeros (0.74 ulp from the true ones): use exactly those values where
        // the table has them, so that r_k, q_k, Qmax are bit-identical with hankel.py:72-78; the Newton root only beyond
        d.zeros[k - 1] = k <= FH_J0_ZEROS_TABLE ? FH_J0_ZEROS[k - 1] : (double)x;
        long double f, g;
        bessel_j01_miller((long double)d.zeros[k - 1], &f, &g);  // J1 at the fp64 zero, as j1(j_nk) does
        J1z[k - 1] = g;
    }
    const double j_nN = d.zeros[N];
    d.j_nN = j_nN;
    d.Qmax = j_nN / (2 * M_PI * Rmax_rad);  // hankel.py:75
    d.r.resize(N);
    d.q.resize(N);
    d.scale_factor.resize(N);
    for (int k = 0; k < N; ++k) {
        d.r[k] = Rmax_rad * (d.zeros[k] / j_nN);  // hankel.py:77
        d.q[k] = d.Qmax * (d.zeros[k] / j_nN);    // hankel.py:78
        d.scale_factor[k] = (double)(1.0L / (J1z[k] * J1z[k]));  // hankel.py:89
    }
    d.Ykm.resize((size_t)N * N);
    std::vector<double> tab(FH_J0_TABLE_DOUBLES);
    fh_j0_fill_table(tab.data());
    for (int k = 0; k < N; ++k) {
        const double jk_over = d.zeros[k] / j_nN;
        for (int m = 0; m < N; ++m) {
            // hankel.py:86-87: (2 / (j_nN * J1(j_m)^2)) * J0(j_m * (j_k / j_nN))
            const double pre = (double)(2.0L / ((long double)j_nN * J1z[m] * J1z[m]));
            d.Ykm[(size_t)k * N + m] = pre * fh_j0(d.zeros[m] * jk_over, (const double *)tab.data());
        }
    }
    return FH_OK;
}

void fh_dht_self_coefficients(const fh_dht &d, double *Y) {
    const double norm = 1 / (M_PI * d.Qmax * d.Qmax);
    const double f = 0.5 * d.j_nN * norm;
    for (size_t i = 0; i < (size_t)d.N * d.N; ++i) Y[i] = f * d.Ykm[i];
}
