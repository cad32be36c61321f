// The Taylor tables of the J0 buckets (j0_buckets.h) built ON THE DEVICE (round 5).
//
// j0_buckets.cpp fills table[b][n][k] = a_n(s0_b j_k) (j_k Delta / 2)^n in x87 long double on the host -- 35 ms and a 55 MB upload
// for the 2 356 buckets of a table that reaches Q_max at N = 300, i.e. most of a whole fit's time at the first sight of every new
// (N, baseline range), which is how frank is normally used (N chosen so that q[-1] just clears the data,
// statistical_models.py:512-535).  Here the host only seeds every kSeedStride-th bucket -- J0(x0), -J1(x0) in long double, handed
// over as double-double pairs -- and one thread per (chain of kSeedStride buckets, column k) does the rest in DOUBLE-DOUBLE
// arithmetic (two doubles per value, ~104 bits: more than the host's 64):
//   * at a bucket centre x0 with y = J0(x0), y' = J0'(x0) known: the Taylor coefficients about x0 from Bessel's equation,
//         a_{n+2} = -[(n+1)^2 a_{n+1} + x0 a_n + a_{n-1}] / (x0 (n+2)(n+1))            (j0_buckets.cpp, the same recurrence)
//     the first twelve, scaled by h^n, h = j_k Delta / 2, are the bucket's table entries (rounded to double);
//   * y and y' at the NEXT centre (the next ROUNDED centre: its distance is formed in double-double) by Horner over 24 of them.
// Rounding excites the Y0-like solution, whose coefficients grow like eps x0^-n.  The SEEDS carry the host's eps (1e-19): from
// bucket b the march evaluates at t = 2h = 2 x0 / (2b + 1), so that part is eps (2 / (2b + 1))^n -- harmless from bucket 1 on
// (<= (2/3)^n), but 2^24 eps = 1e-12 out of bucket 0, whose centre is as far from the singularity at 0 as its neighbour's is from
// it.  So bucket 0 is a chain of its own: the host seeds buckets 0 and 1 (first_len = 1 when the range starts at bucket 0).
// tests/test_gpu_parity.py::test_bucket_tables_built_on_the_device holds the result to <= 1 ulp of the host's tables.
#include <hip/hip_runtime.h>

#include "j0_buckets.h"
#include "kernels.h"

#pragma STDC FP_CONTRACT OFF

namespace {

constexpr int kTermsDD = 24;

struct dd {
    double hi, lo;
};
__device__ __forceinline__ dd two_sum(double a, double b) {
    const double s = a + b, bb = s - a;
    return {s, (a - (s - bb)) + (b - bb)};
}
__device__ __forceinline__ dd quick_two_sum(double a, double b) {
    const double s = a + b;
    return {s, b - (s - a)};
}
__device__ __forceinline__ dd two_prod(double a, double b) {
    const double p = a * b;
    return {p, __builtin_fma(a, b, -p)};
}
__device__ __forceinline__ dd dd_add(dd a, dd b) {
    dd s = two_sum(a.hi, b.hi);
    const dd t = two_sum(a.lo, b.lo);
    s.lo += t.hi;
    s = quick_two_sum(s.hi, s.lo);
    s.lo += t.lo;
    return quick_two_sum(s.hi, s.lo);
}
__device__ __forceinline__ dd dd_neg(dd a) { return {-a.hi, -a.lo}; }
__device__ __forceinline__ dd dd_mul(dd a, dd b) {
    dd p = two_prod(a.hi, b.hi);
    p.lo += a.hi * b.lo + a.lo * b.hi;
    return quick_two_sum(p.hi, p.lo);
}
__device__ __forceinline__ dd dd_mul_d(dd a, double b) {
    dd p = two_prod(a.hi, b);
    p.lo += a.lo * b;
    return quick_two_sum(p.hi, p.lo);
}
__device__ __forceinline__ dd dd_div(dd a, dd b) {
    const double q1 = a.hi / b.hi;
    dd r = dd_add(a, dd_neg(dd_mul_d(b, q1)));
    const double q2 = r.hi / b.hi;
    r = dd_add(r, dd_neg(dd_mul_d(b, q2)));
    const double q3 = r.hi / b.hi;
    const dd q = quick_two_sum(q1, q2);
    return dd_add(q, dd{q3, 0.0});
}

// seeds[(c * N + k) * 4 + {0, 1, 2, 3}] = y hi, y lo, y' hi, y' lo at the centre of the first bucket of chain c, column k; chain 0
// covers first_len buckets from b0, chain c >= 1 the `stride` buckets from b0 + first_len + (c - 1) stride
__global__ __launch_bounds__(256) void bucket_table_kernel(const double *zeros, int N, int XS, int b0, int b1, int first_len, int stride,
                                                           double Delta, const double *seeds, double *table) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y;
    if (k >= N) return;
    const double jk = zeros[k];
    const double *sd = seeds + ((size_t)c * N + k) * 4;
    dd y = {sd[0], sd[1]}, yp = {sd[2], sd[3]};
    const dd h = dd_mul_d(two_prod(jk, Delta), 0.5);  // j_k Delta / 2 (the halving is exact)
    const int bfirst = c == 0 ? b0 : b0 + first_len + (c - 1) * stride, blen = c == 0 ? first_len : stride;
    for (int b = bfirst; b < bfirst + blen && b < b1; ++b) {
        const double s0 = ((double)b + 0.5) * Delta;  // fh_k1_bucket_centre: the same fp64 expression as the host
        const dd x0 = two_prod(s0, jk);
        dd a[kTermsDD];
        a[0] = y;
        a[1] = yp;
        dd am1 = {0.0, 0.0};
        for (int n = 0; n + 2 < kTermsDD; ++n) {
            dd num = dd_add(dd_add(dd_mul_d(a[n + 1], (double)((n + 1) * (n + 1))), dd_mul(x0, a[n])), am1);
            a[n + 2] = dd_neg(dd_div(num, dd_mul_d(x0, (double)((n + 2) * (n + 1)))));
            am1 = a[n];
        }
        double *tb = table + (size_t)(b - b0) * FH_K1_TERMS * XS + k;
        dd p = {1.0, 0.0};
        for (int n = 0; n < FH_K1_TERMS; ++n) {
            const dd e = dd_mul(a[n], p);
            tb[(size_t)n * XS] = e.hi + e.lo;
            p = dd_mul(p, h);
        }
        // y, y' at the next centre.  The centres are ROUNDED fp64 expressions ((b + 1/2) Delta, as the host and the kernels
        // that use the tables form them): the step is the difference of the two products, not 2 h -- which is off by the
        // rounding of a centre, 1e-17 of x0, and through y' by as much in y
        const double s1 = ((double)(b + 1) + 0.5) * Delta;
        const dd t = dd_add(two_prod(s1, jk), dd_neg(x0));
        dd sy = a[kTermsDD - 1], syp = dd_mul_d(a[kTermsDD - 1], (double)(kTermsDD - 1));
        for (int n = kTermsDD - 2; n >= 0; --n) {
            sy = dd_add(dd_mul(sy, t), a[n]);
            if (n >= 1) syp = dd_add(dd_mul(syp, t), dd_mul_d(a[n], (double)n));
        }
        y = sy;
        yp = syp;
    }
}

}  // namespace

int fh_k1_seed_stride() { return 16; }
// chains of the range [b0, b1) and the first bucket of chain c (the layout of the seeds): bucket 0 is a chain of its own
int fh_k1_seed_first_len(int b0) { return b0 == 0 ? 1 : fh_k1_seed_stride(); }
int fh_k1_seed_chains(int b0, int b1) {
    if (b1 <= b0) return 0;
    const int fl = fh_k1_seed_first_len(b0), st = fh_k1_seed_stride();
    return b1 - b0 <= fl ? 1 : 1 + (b1 - b0 - fl + st - 1) / st;
}
int fh_k1_seed_bucket(int b0, int c) { return c == 0 ? b0 : b0 + fh_k1_seed_first_len(b0) + (c - 1) * fh_k1_seed_stride(); }

hipError_t fh_k1_bucket_table_device(const double *zeros_dev, int N, int XS, int b0, int b1, double Delta, const double *seeds_dev,
                                     double *table_dev, hipStream_t stream) {
    if (b1 <= b0) return hipSuccess;
    hipLaunchKernelGGL(bucket_table_kernel, dim3((N + 255) / 256, fh_k1_seed_chains(b0, b1)), dim3(256), 0, stream, zeros_dev, N, XS, b0,
                       b1, fh_k1_seed_first_len(b0), fh_k1_seed_stride(), Delta, seeds_dev, table_dev);
    return hipGetLastError();
}
