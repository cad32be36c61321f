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

// Degree-12 polynomial sum_{k=0}^{12} c(k) x^k.  Estrin's scheme (default): 15 DP ops instead of Horner's 12, but a
// dependency depth of 5 instead of 12.  FH_J0_HORNER selects Horner: fewer operations and far fewer live registers
// (coefficients are consumed as they arrive), for kernels that hide the chain latency with more resident waves.
#ifdef FH_J0_HORNER
#define FH_ESTRIN12(res, c, x)                                  \
    do {                                                        \
        double h_ = c(12);                                      \
        h_ = fma(h_, (x), c(11));                               \
        h_ = fma(h_, (x), c(10));                               \
        h_ = fma(h_, (x), c(9));                                \
        h_ = fma(h_, (x), c(8));                                \
        h_ = fma(h_, (x), c(7));                                \
        h_ = fma(h_, (x), c(6));                                \
        h_ = fma(h_, (x), c(5));                                \
        h_ = fma(h_, (x), c(4));                                \
        h_ = fma(h_, (x), c(3));                                \
        h_ = fma(h_, (x), c(2));                                \
        h_ = fma(h_, (x), c(1));                                \
        (res) = fma(h_, (x), c(0));                             \
    } while (0)
#define FH_ESTRIN11(res, c, x)                                  \
    do {                                                        \
        double h_ = c(11);                                      \
        h_ = fma(h_, (x), c(10));                               \
        h_ = fma(h_, (x), c(9));                                \
        h_ = fma(h_, (x), c(8));                                \
        h_ = fma(h_, (x), c(7));                                \
        h_ = fma(h_, (x), c(6));                                \
        h_ = fma(h_, (x), c(5));                                \
        h_ = fma(h_, (x), c(4));                                \
        h_ = fma(h_, (x), c(3));                                \
        h_ = fma(h_, (x), c(2));                                \
        h_ = fma(h_, (x), c(1));                                \
        (res) = fma(h_, (x), c(0));                             \
    } while (0)
#else
#define FH_ESTRIN12(res, c, x)                                                              \
    do {                                                                                    \
        const double x2_ = (x) * (x), x4_ = x2_ * x2_, x8_ = x4_ * x4_;                     \
        const double p0_ = fma(c(1), (x), c(0)), p1_ = fma(c(3), (x), c(2));                \
        const double p2_ = fma(c(5), (x), c(4)), p3_ = fma(c(7), (x), c(6));                \
        const double p4_ = fma(c(9), (x), c(8)), p5_ = fma(c(11), (x), c(10));              \
        const double q0_ = fma(p1_, x2_, p0_), q1_ = fma(p3_, x2_, p2_), q2_ = fma(p5_, x2_, p4_); \
        const double r0_ = fma(q1_, x4_, q0_), r1_ = fma(c(12), x4_, q2_);                  \
        (res) = fma(r1_, x8_, r0_);                                                         \
    } while (0)
// Degree-11 variant (12 coefficients)
#define FH_ESTRIN11(res, c, x)                                                              \
    do {                                                                                    \
        const double x2_ = (x) * (x), x4_ = x2_ * x2_, x8_ = x4_ * x4_;                     \
        const double p0_ = fma(c(1), (x), c(0)), p1_ = fma(c(3), (x), c(2));                \
        const double p2_ = fma(c(5), (x), c(4)), p3_ = fma(c(7), (x), c(6));                \
        const double p4_ = fma(c(9), (x), c(8)), p5_ = fma(c(11), (x), c(10));              \
        const double q0_ = fma(p1_, x2_, p0_), q1_ = fma(p3_, x2_, p2_), q2_ = fma(p5_, x2_, p4_); \
        const double r0_ = fma(q1_, x4_, q0_);                                              \
        (res) = fma(q2_, x8_, r0_);                                                         \
    } while (0)
#endif

// Large-argument branch, x >= FH_J0_XSPLIT.  `ab` = FH_J0_AB ({A_k, B_k} pairs, highest power first) followed by
// the cosine coefficients FH_J0_CD.
template <typename TabPtr>
FH_HD double fh_j0_large(double x, TabPtr ab) {
    const double y = fh_rsqrt(x);
    const double r = y * y;
    const double w = r * r;
    const double u = fma(w, FH_J0_USCALE, -1.0);
    static_assert(FH_J0_ADEG == 12 && FH_J0_BDEG == 12 && FH_J0_CDEG == 11, "Estrin schemes are written for these degrees");
    double a, b;  // ab[2k] = A_{12-k}, ab[2k+1] = B_{12-k}
#define FH_CA(k) ab[2 * (12 - (k))]
#define FH_CB(k) ab[2 * (12 - (k)) + 1]
    FH_ESTRIN12(a, FH_CA, u);
    FH_ESTRIN12(b, FH_CB, u);
#undef FH_CA
#undef FH_CB
    const double m = rint(fma(x, FH_INV_PI, -0.25));
    double ph = fma(-m, FH_PI1, x);
    ph = fma(-m, FH_PI2, ph);
    ph = fma(-m, FH_PI3, ph);
    ph = (ph - FH_PIO4_HI) - FH_PIO4_LO;
    ph = fma(r, b, ph);
    const double z = ph * ph;
    double c;  // ab[NAB + k] = C_{11-k}
#define FH_CC(k) ab[FH_J0_NAB + 11 - (k)]
    FH_ESTRIN11(c, FH_CC, z);
#undef FH_CC
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
    static_assert(FH_J0_TDEG == 12, "Estrin scheme is written for degree 12");
    double a;
#define FH_CT(k) row[k]
    FH_ESTRIN12(a, FH_CT, t);
#undef FH_CT
    return a;
}

// `tab` points at FH_J0_TABLE_DOUBLES doubles: FH_J0_TAYLOR, FH_J0_AB, FH_J0_CD.
template <typename TabPtr>
FH_HD double fh_j0(double x, TabPtr tab) {
    if (x < FH_J0_XSPLIT) return fh_j0_small(x, tab);
    return fh_j0_large(x, tab + FH_J0_NI * FH_J0_TSTRIDE);
}

// Host copy of the combined table (Taylor rows, {A,B} pairs, cosine coefficients).
static inline void fh_j0_fill_table(double *dst) {
    for (int i = 0; i < FH_J0_NI * FH_J0_TSTRIDE; ++i) dst[i] = FH_J0_TAYLOR[i];
    for (int i = 0; i < FH_J0_NAB; ++i) dst[FH_J0_NI * FH_J0_TSTRIDE + i] = FH_J0_AB[i];
    for (int i = 0; i < FH_J0_NC; ++i) dst[FH_J0_NI * FH_J0_TSTRIDE + FH_J0_NAB + i] = FH_J0_CD[i];
}
