// Host side of bin_gram v2: Taylor tables of J0 about bucket centres (DESIGN.md "K1").
//
// Replaces the per-element evaluation of scipy.special.j0 in DHT.coefficients(q) (hankel.py:201-202) inside
// map_visibilities (statistical_models.py:200-214): visibilities are bucketed by s = q/Qmax into buckets of width
// Delta = 2 h / j_N; inside bucket b
//     J0((s0_b + d) j_k) = sum_{n < 12} a_n(s0_b j_k) (d j_k)^n,      |d j_k| <= h = 0.25   (truncation 1.2e-16)
// so a 16-visibility x 16-column tile of the design block is a (16 x 12) x (12 x 16) matrix product for the MFMA pipe.
// Table entry [b][n][k] = a_n(s0_b j_k) (j_k Delta / 2)^n, to be multiplied by tau^n, tau = d / (Delta / 2) in [-1, 1].
#pragma once
#include <cstddef>

constexpr int FH_K1_TERMS = 12;        // Taylor terms = 3 MFMA k-steps of 4
constexpr double FH_K1_HALFWIDTH = 0.25;  // h: half bucket width in x = s j_N

// bucket width in s for the N zeros `zeros` (uses the largest, zeros[N-1])
double fh_k1_bucket_width(const double *zeros, int N);
// centre of bucket b, the same fp64 expression on host and device
inline double fh_k1_bucket_centre(int b, double Delta) { return ((double)b + 0.5) * Delta; }
// Fill out[(b - b0) * 12 * XS + n * XS + k] for buckets b0 <= b < b1; columns k >= N are zero.  Long-double arithmetic,
// multi-threaded.  Returns 0, or -1 on a bad argument.
int fh_k1_bucket_table(const double *zeros, int N, int XS, int b0, int b1, double *out);
// seeds for the construction on the device (j0_buckets_device.hip): see j0_buckets.cpp
int fh_k1_bucket_seeds(const double *zeros, int N, const int *buckets, int chains, double *out);
