// Taylor tables of J0 about bucket centres, in x87 long double (see j0_buckets.h).
//
// a_0 = J0(x0), a_1 = -J1(x0); Bessel's equation (x0 + t) y'' + y' + (x0 + t) y = 0 gives for the Taylor
// coefficients about x0
//     a_{n+2} = -[(n+1)^2 a_{n+1} + x0 a_n + a_{n-1}] / (x0 (n+2)(n+1)).
// Rounding errors excite the Y0-like solution, whose coefficients decay like x0^-n; they are evaluated at
// |t| <= h_k <= x0 (bucket 0: h_k = x0, later buckets h_k = x0 / (2b+1)), so their contribution stays O(n eps_80bit).
#include "j0_buckets.h"

#include <cmath>
#include <cstring>
#include <thread>
#include <vector>

double fh_k1_bucket_width(const double *zeros, int N) { return 2.0 * FH_K1_HALFWIDTH / zeros[N - 1]; }

namespace {
void fill_range(const double *zeros, int N, int XS, double Delta, int b0, int bl, int bh, double *out) {
    for (int b = bl; b < bh; ++b) {
        double *tb = out + (size_t)(b - b0) * FH_K1_TERMS * XS;
        const long double s0 = (long double)fh_k1_bucket_centre(b, Delta);
        for (int k = 0; k < N; ++k) {
            const long double jk = (long double)zeros[k];
            const long double x0 = s0 * jk;
            const long double hk = jk * (long double)Delta * 0.5L;
            long double a[FH_K1_TERMS + 1];
            a[0] = j0l(x0);
            a[1] = -j1l(x0);
            long double am1 = 0.0L;
            for (int n = 0; n + 2 < FH_K1_TERMS; ++n) {
                a[n + 2] = -((long double)((n + 1) * (n + 1)) * a[n + 1] + x0 * a[n] + am1) /
                           (x0 * (long double)((n + 2) * (n + 1)));
                am1 = a[n];
            }
            long double p = 1.0L;
            for (int n = 0; n < FH_K1_TERMS; ++n) {
                tb[(size_t)n * XS + k] = (double)(a[n] * p);
                p *= hk;
            }
        }
    }
}
}  // namespace

// Seeds of the device-side construction (j0_buckets_device.hip): J0 and J0' = -J1 at the centres of the `chains` buckets
// listed in `buckets`, for every column, in long double, as double-double pairs: out[(c * N + k) * 4] = y hi, y lo, y' hi, y' lo.
int fh_k1_bucket_seeds(const double *zeros, int N, const int *buckets, int chains, double *out) {
    if (!zeros || !out || !buckets || N < 1 || chains < 0) return -1;
    const double Delta = fh_k1_bucket_width(zeros, N);
    auto fill = [&](int c0, int c1) {
        for (int c = c0; c < c1; ++c) {
            const long double s0 = (long double)fh_k1_bucket_centre(buckets[c], Delta);
            for (int k = 0; k < N; ++k) {
                const long double x0 = s0 * (long double)zeros[k];
                const long double y = j0l(x0), yp = -j1l(x0);
                double *o = out + ((size_t)c * N + k) * 4;
                o[0] = (double)y;
                o[1] = (double)(y - (long double)o[0]);
                o[2] = (double)yp;
                o[3] = (double)(yp - (long double)o[2]);
            }
        }
    };
    unsigned hw = std::thread::hardware_concurrency();
    int nt = (int)(hw ? hw : 1);
    if (nt > 16) nt = 16;
    if ((long long)chains * N < 20000) nt = 1;
    if (nt > chains) nt = chains;
    if (nt <= 1) {
        fill(0, chains);
        return 0;
    }
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t) th.emplace_back(fill, (int)((long long)chains * t / nt), (int)((long long)chains * (t + 1) / nt));
    for (auto &x : th) x.join();
    return 0;
}

int fh_k1_bucket_table(const double *zeros, int N, int XS, int b0, int b1, double *out) {
    if (!zeros || !out || N < 1 || XS < N || b0 < 0 || b1 < b0) return -1;
    const int nb = b1 - b0;
    if (nb == 0) return 0;
    memset(out, 0, sizeof(double) * (size_t)nb * FH_K1_TERMS * XS);
    const double Delta = fh_k1_bucket_width(zeros, N);
    unsigned hw = std::thread::hardware_concurrency();
    int nt = (int)(hw ? hw : 1);
    if (nt > 32) nt = 32;
    if ((long long)nb * N < 20000) nt = 1;  // not worth the thread start-up
    if (nt > nb) nt = nb;
    if (nt <= 1) {
        fill_range(zeros, N, XS, Delta, b0, b0, b1, out);
        return 0;
    }
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t) {
        const int lo = b0 + (int)((long long)nb * t / nt), hi = b0 + (int)((long long)nb * (t + 1) / nt);
        th.emplace_back(fill_range, zeros, N, XS, Delta, b0, lo, hi, out);
    }
    for (auto &x : th) x.join();
    return 0;
}
