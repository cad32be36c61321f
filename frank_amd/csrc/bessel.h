// fp64 Bessel J0 for gfx950 (and the host DHT set-up): table-driven, division-free, one branch.
//
// Replaces scipy.special.j0 (Cephes) at hankel.py:59,201-202.  Absolute error <= 1.2e-16 over
// [0, 1e3] (tools/gen_j0_tables.py measures it against 50-digit mpmath; Cephes: 4e-16 .. 1.3e-15).
//
//   x <  8 : degree-12 Taylor polynomial about the centre of one of 16 half-unit intervals; the
//            per-lane coefficient row comes from `tab` (LDS on the device).
//   x >= 8 : J0 = rsqrt(x) A(w) cos(x - pi/4 + B(w)/x), w = 1/x^2; A, B degree-12 polynomials in
//            u = 128 w - 1 (coefficients read pairwise from `tab`, i.e. LDS, to spare scalar registers), a three-term Cody-Waite reduction modulo pi (exact under FMA) and ONE
//            cosine polynomial on [-pi/2, pi/2].  The -pi/4 is subtracted after the reduction, so
//            the phase keeps full precision at x ~ 1e3 where Cephes' `x - PIO4` already rounds.
#pragma once
#include <math.h>

#include "j0_tables.h"

#if defined(__HIP__)
#define FH_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define FH_HD static inline
#endif

FH_HD double fh_rsqrt(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    // v_rsq_f64 is good to ~2^-26; one cubically convergent step reaches < 1 ulp.
    double y = __builtin_amdgcn_rsq(x);
    double e = fma(-(x * y), y, 1.0);
    double p = fma(0.375, e, 0.5);
    return fma(y * e, p, y);
#else
    return 1.0 / sqrt(x);
#endif
}

// Horner with compile-time coefficients (they become scalar registers on the device).
template <int N>
FH_HD double fh_horner(const double (&c)[N], double x) {
    double a = c[N - 1];
#pragma unroll
    for (int k = N - 2; k >= 0; --k) a = fma(a, x, c[k]);
    return a;
}

// Large-argument branch, x >= FH_J0_XSPLIT.  `ab` = FH_J0_AB ({A_k, B_k} pairs, highest power first).
template <typename TabPtr>
FH_HD double fh_j0_large(double x, TabPtr ab) {
    const double y = fh_rsqrt(x);
    const double r = y * y;
    const double w = r * r;
    const double u = fma(w, FH_J0_USCALE, -1.0);
    double a = ab[0], b = ab[1];
#pragma unroll
    for (int k = 1; k <= FH_J0_ADEG; ++k) {
        a = fma(a, u, ab[2 * k]);
        b = fma(b, u, ab[2 * k + 1]);
    }
    const double m = rint(fma(x, FH_INV_PI, -0.25));
    double ph = fma(-m, FH_PI1, x);
    ph = fma(-m, FH_PI2, ph);
    ph = fma(-m, FH_PI3, ph);
    ph = (ph - FH_PIO4_HI) - FH_PIO4_LO;
    ph = fma(r, b, ph);
    const double z = ph * ph;
    const double c = fh_horner<FH_J0_CDEG + 1>(FH_J0_C, z);
    // (-1)^m without an integer conversion (valid for every finite m)
    const double odd = fabs(fma(-2.0, rint(0.5 * m), m));
    const double s = fma(-2.0, odd, 1.0);
    return (a * y) * (c * s);
}

// Small-argument branch, 0 <= x < FH_J0_XSPLIT.  `tab` = FH_J0_TAYLOR (any address space).
template <typename TabPtr>
FH_HD double fh_j0_small(double x, TabPtr tab) {
    int idx = (int)(x * FH_J0_INV_WIDTH);
    idx = idx > FH_J0_NI - 1 ? FH_J0_NI - 1 : idx;
    const double t = x - ((double)idx + 0.5) * FH_J0_WIDTH;
    TabPtr row = tab + idx * FH_J0_TSTRIDE;
    double a = row[FH_J0_TDEG];
#pragma unroll
    for (int k = FH_J0_TDEG - 1; k >= 0; --k) a = fma(a, t, row[k]);
    return a;
}

// `tab` points at FH_J0_TABLE_DOUBLES doubles: FH_J0_TAYLOR followed by FH_J0_AB.
template <typename TabPtr>
FH_HD double fh_j0(double x, TabPtr tab) {
    if (x < FH_J0_XSPLIT) return fh_j0_small(x, tab);
    return fh_j0_large(x, tab + FH_J0_NI * FH_J0_TSTRIDE);
}

// Host copy of the combined table (Taylor rows, then {A,B} pairs).
static inline void fh_j0_fill_table(double *dst) {
    for (int i = 0; i < FH_J0_NI * FH_J0_TSTRIDE; ++i) dst[i] = FH_J0_TAYLOR[i];
    for (int i = 0; i < FH_J0_NAB; ++i) dst[FH_J0_NI * FH_J0_TSTRIDE + i] = FH_J0_AB[i];
}
