// (T + I) tau = rhs of CriticalFilter.update_power_spectrum (filter.py:171-177) with the host-factorised pentadiagonal bands, as
// two wave scans.  Shared by the fit loop (fit_loop.hip: bands and tables in LDS) and the LogNormal kernel (lognormal.hip: in
// its global scratch, L2) so that both produce the same tau from the same right-hand side.
// band: six arrays of NP entries -- f1, f2 (lower factors), d0, u1, u2 (pivots and upper bands), 1 / d0 -- padded beyond N with
// unit pivots and zero bands; Q: 2 x 6 x 4 x 64 doubles (scan_tables, once per fit); kScanB: rows per lane, 64 kScanB >= NP.
#pragma once
#include <hip/hip_runtime.h>

namespace bandscan {

constexpr int kTableDoubles = 2 * 6 * 4 * 64;

// ---- (T + I) tau = rhs by a wave scan ------------------------------------------------------------------------------------
// The two substitutions of the pentadiagonal LU are second-order linear recurrences, x_i = r_i - a1_i x_{i-1} - a2_i x_{i-2}:
// with the state s_i = (x_i, x_{i-1}) they are affine maps s_i = M_i s_{i-1} + (r_i, 0), M_i = [[-a1_i, -a2_i], [1, 0]], and
// affine maps compose associatively.  Lane l owns a block of kScanB consecutive rows: it folds its block from the zero state
// (the block's offset vector v_l), a Kogge-Stone scan over the 64 lanes turns the v_l into the states at the block ends, and
// every lane replays its block from the state its left neighbour ended in.  The matrix parts of the maps do not depend on
// the right-hand side: the six per-level matrices of every lane are formed once per fit (scan_tables) -- a pass only moves
// vectors: 2 x 5 recurrence steps and 6 x (2 shuffles + 4 fmas) per lane and direction instead of 2 N dependent steps of
// one thread (16 us per pass, 50 us before its LDS traffic was trimmed).  dir 0: forward, coefficients f1, f2; dir 1: backward
// over reversed rows, coefficients u1 / d, u2 / d, right-hand side r / d.
__device__ __forceinline__ double shfl_up_f64(double v, int off) { return __shfl_up(v, off); }
__device__ __forceinline__ void scan_coef(const double *band, int NP, int dir, int row, double &a1, double &a2, double &rs) {
    // row: position along the direction of the recurrence; rows past NP - 1 are the identity recurrence x = 0
    if (row >= NP) {
        a1 = a2 = 0.0;
        rs = 0.0;
        return;
    }
    if (dir == 0) {
        a1 = band[row];
        a2 = band[NP + row];
        rs = 1.0;
    } else {
        const int i = NP - 1 - row;
        const double rd = band[5 * NP + i];
        a1 = band[3 * NP + i] * rd;
        a2 = band[4 * NP + i] * rd;
        rs = rd;
    }
}
template <int kScanB>
__device__ __forceinline__ void scan_tables(const double *band, int NP, double *Q, int lane) {
    for (int dir = 0; dir < 2; ++dir) {
        double A00 = 1.0, A01 = 0.0, A10 = 0.0, A11 = 1.0;  // product of the block's M_i, latest on the left
#pragma unroll
        for (int t = 0; t < kScanB; ++t) {
            double a1, a2, rs;
            scan_coef(band, NP, dir, lane * kScanB + t, a1, a2, rs);
            const double n00 = fma(-a1, A00, -a2 * A10), n01 = fma(-a1, A01, -a2 * A11);  // M A
            A10 = A00;
            A11 = A01;
            A00 = n00;
            A01 = n01;
        }
#pragma unroll
        for (int d = 0; d < 6; ++d) {
            double *q = Q + ((dir * 6 + d) * 4) * 64 + lane;
            q[0] = A00;
            q[64] = A01;
            q[128] = A10;
            q[192] = A11;
            const int off = 1 << d;
            const double L00 = shfl_up_f64(A00, off), L01 = shfl_up_f64(A01, off), L10 = shfl_up_f64(A10, off),
                         L11 = shfl_up_f64(A11, off);
            if (lane >= off) {  // A <- A * A_left
                const double n00 = fma(A00, L00, A01 * L10), n01 = fma(A00, L01, A01 * L11);
                const double n10 = fma(A10, L00, A11 * L10), n11 = fma(A10, L01, A11 * L11);
                A00 = n00;
                A01 = n01;
                A10 = n10;
                A11 = n11;
            }
        }
    }
}
// one direction of the solve, by wave 0 (all 64 lanes); rhs: NP entries in LDS, solved in place
template <int kScanB>
__device__ __forceinline__ void scan_solve(const double *band, int NP, const double *Q, double *rhs, int dir, int lane) {
    double a1[kScanB], a2[kScanB], r[kScanB];
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int t = 0; t < kScanB; ++t) {
        const int row = lane * kScanB + t;
        double rs;
        scan_coef(band, NP, dir, row, a1[t], a2[t], rs);
        r[t] = row < NP ? rhs[dir == 0 ? row : NP - 1 - row] * rs : 0.0;
        double x = fma(-a2[t], s1, r[t]);
        x = fma(-a1[t], s0, x);
        s1 = s0;
        s0 = x;
    }
#pragma unroll
    for (int d = 0; d < 6; ++d) {
        const int off = 1 << d;
        const double *q = Q + ((dir * 6 + d) * 4) * 64 + lane;
        const double l0 = shfl_up_f64(s0, off), l1 = shfl_up_f64(s1, off);
        if (lane >= off) {
            const double n0 = fma(q[0], l0, fma(q[64], l1, s0)), n1 = fma(q[128], l0, fma(q[192], l1, s1));
            s0 = n0;
            s1 = n1;
        }
    }
    double p0 = shfl_up_f64(s0, 1), p1 = shfl_up_f64(s1, 1);  // the state the left neighbour's block ends in
    if (lane == 0) p0 = p1 = 0.0;
#pragma unroll
    for (int t = 0; t < kScanB; ++t) {
        const int row = lane * kScanB + t;
        double x = fma(-a2[t], p1, r[t]);
        x = fma(-a1[t], p0, x);
        p1 = p0;
        p0 = x;
        if (row < NP) rhs[dir == 0 ? row : NP - 1 - row] = x;
    }
}

}  // namespace bandscan
