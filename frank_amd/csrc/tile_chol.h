// 16x16 tile primitives of the blocked fp64 Cholesky on gfx950 (v_mfma_f64_16x16x4_f64), shared by the K2 fit loop
// (fit_loop.hip) and the LogNormal kernel's Hessian factorisation (lognormal.hip).
//
// Reference: scipy.linalg.cho_factor as called by GaussianModel._fit (statistical_models.py:742) and
// LogNormalMAPModel._fit (statistical_models.py:1147-1149).
#pragma once
#include <hip/hip_runtime.h>

typedef double v4f64 __attribute__((ext_vector_type(4)));

namespace tilechol {

constexpr int PS = 17;  // LDS stride of the 16-wide panel rows (doubles)

__device__ __forceinline__ double bcast(double v, int lane) {  // wave-uniform broadcast of lane `lane`'s value
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// v_rsq_f64 + one cubically convergent correction: < 1 ulp, no division, no sqrt call on the serial path
__device__ __forceinline__ double rsqrt_f64(double x) {
    double y = __builtin_amdgcn_rsq(x);
    const double e = fma(-(x * y), y, 1.0);
    return fma(y * e, fma(0.375, e, 0.5), y);
}

// ---- Cholesky of a 16x16 diagonal tile by ONE wave: every 16-lane group holds the tile, lane&15 = row -----------
// `tile` points at element [0][0] (row stride `ts`; global A/C or the LDS copy made by the look-ahead);
// `pdiag` (or NULL) is added to the diagonal (first touch of A).  force_c >= 0: pivot of that local column is taken
// as 1 (the augmented row that carries b, see solve_posterior).  Writes L to Cout (global) and dl (+ 1/diag in col 16).
__device__ __forceinline__ bool factor_diag_tile(const double *tile, int ts, const double *pdiag, int force_c,
                                                 double *Cout, int ld, double *dl, int lane) {
    double t[16];
    const int r = lane & 15;
    const double *sp = tile + (size_t)r * ts;
#pragma unroll
    for (int c = 0; c < 16; ++c) t[c] = (c <= r) ? sp[c] : 0.0;
    if (pdiag) {
        const double pv = pdiag[r];
#pragma unroll
        for (int c = 0; c < 16; ++c)
            if (c == r) t[c] += pv;
    }
    bool ok = true;
    double dinv_mine = 0.0;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const double d2 = (c == force_c) ? 1.0 : bcast(t[c], c);
        ok = ok && (d2 > 0.0);
        const double dinv = rsqrt_f64(d2);
        t[c] = (r == c) ? d2 * dinv : t[c] * dinv;  // rows r > c: L[r][c]; (rows < c hold zeros)
        if (r == c) dinv_mine = dinv;
        if (c < 15) {
#pragma unroll
            for (int c2 = c + 1; c2 < 16; ++c2) {
                const double s = bcast(t[c], c2);  // L[c2][c]
                t[c2] = fma(-t[c], s, t[c2]);      // only rows r >= c2 matter
            }
        }
    }
    if (lane < 16) {
        double *dst = Cout + (size_t)r * ld;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const double v = c <= r ? t[c] : 0.0;
            dl[r * PS + c] = v;
            if (c <= r) dst[c] = v;
        }
        dl[r * PS + 16] = dinv_mine;
    }
    return ok;
}

// ---- inverse of the diagonal tile just factored (its L is in dl): W_kk = L_kk^-1 -> dli (LDS), W, WdT ----------------
// One wave; lane c holds column c of the inverse; L is read from dl with wave-uniform (broadcast) addresses.
__device__ __forceinline__ void invert_factored_tile(const double *dl, double *dli, double *W, double *WdT, int ld,
                                                     int k, int lane, double *cs_kk, int rows_valid) {
    const int c = lane & 15;
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): dl was written by this wave
    __builtin_amdgcn_wave_barrier();
    // right-looking forward substitution: once x_s is known every later row is updated independently, so the
    // dependency depth is 16 (not 120 as with row-by-row dot products)
    double x[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) x[r] = (r == c) ? 1.0 : 0.0;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        x[s] = x[s] * dl[s * PS + 16];  // * 1/L[s][s]
#pragma unroll
        for (int r = s + 1; r < 16; ++r) x[r] = fma(-dl[r * PS + s], x[s], x[r]);  // L[r][s] * X[s][c]
    }
    if (lane < 16) {
        double *wt = WdT ? WdT + (size_t)k * 256 + c * 16 : nullptr;  // WdT[k][c][r] = W_kk[r][c]
        double ssq = 0.0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const double v = (r >= c) ? x[r] : 0.0;
            dli[r * PS + c] = v;
            if (W) W[(size_t)(16 * k + r) * ld + 16 * k + c] = v;
            if (wt) wt[r] = v;
            if (r < rows_valid) ssq = fma(v, v, ssq);
        }
        if (cs_kk) cs_kk[c] = ssq;  // column sums of squares of the diagonal tile (rows of the real system only)
    }
}

// Fragment loaders (v_mfma_f64_16x16x4_f64: lane = (cl = lane & 15, rg = lane >> 4), k-step s covers k = 4s + rg).
// "Row form": element [k][cl] of a row-major tile -> 4 rows x 16 contiguous doubles per k-step (coalesced).
struct Frag {
    double v[4];
};
__device__ __forceinline__ Frag load_rows(const double *tile, int ld, int cl, int rg) {
    Frag f;
    const double *p = tile + (size_t)rg * ld + cl;
#pragma unroll
    for (int s = 0; s < 4; ++s) f.v[s] = p[(size_t)(4 * s) * ld];
    return f;
}
__device__ __forceinline__ v4f64 mfma4(const Frag &a, const Frag &b, v4f64 acc, bool neg) {
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(neg ? -a.v[s] : a.v[s], b.v[s], acc, 0, 0, 0);
    return acc;
}
// store a C/D-layout tile at block (I, J) and its transpose at block (J, I)
__device__ __forceinline__ void store_tile(double *Mx, int ld, int I, int J, const v4f64 &t, int cl, int rg, bool mirror) {
    double *p = Mx + (size_t)(16 * I + rg) * ld + 16 * J + cl;
#pragma unroll
    for (int r = 0; r < 4; ++r) p[(size_t)(4 * r) * ld] = t[r];
    if (mirror) {
        double *q = Mx + (size_t)(16 * J + cl) * ld + 16 * I + rg;
#pragma unroll
        for (int r = 0; r < 4; ++r) q[4 * r] = t[r];
    }
}

}  // namespace tilechol
