// 16x16 tile primitives of the blocked fp64 Cholesky on gfx950 (v_mfma_f64_16x16x4_f64), shared by the K2 fit loop
// (fit_loop.hip) and the LogNormal kernel's Hessian factorisation (lognormal.hip).
//
// Reference: scipy.linalg.cho_factor as called by GaussianModel._fit (statistical_models.py:742) and
// LogNormalMAPModel._fit (statistical_models.py:1147-1149).
#pragma once
#include <hip/hip_runtime.h>

#include <utility>

typedef double v4f64 __attribute__((ext_vector_type(4)));
typedef double v2f64 __attribute__((ext_vector_type(2)));

namespace tilechol {

// Pointers into HBM / L2 as GLOBAL address-space pointers: kernel-argument structs that are copied and offset lose the
// address space in the optimiser's eyes and every access becomes a flat_load / flat_store (which also ties up the LDS
// counter, so that "wait for my LDS read" waits for memory too).
typedef __attribute__((address_space(1))) double gdouble;
__device__ __forceinline__ gdouble *as_global(double *p) { return (gdouble *)p; }
__device__ __forceinline__ const gdouble *as_global(const double *p) { return (const gdouble *)p; }

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

// ---- factor AND invert a diagonal tile held in the MFMA accumulator layout, without leaving the registers ---------------
// T (in/out): the symmetric positive definite tile, register r of lane (rg = lane >> 4, cl = lane & 15) = element
// (row rg + 4 r, column cl); on return its lower triangle holds L (T = L L^T), zeros above.  X (out) = L^-1, same layout.
// Right-looking, one column per step, both recurrences in the same 16 steps:
//   * L[i][c] for a lane's own rows comes from its own 16-lane DPP row (row_newbcast:c, two full-rate movs per double);
//   * L[cl][c] = T[c][cl] dinv (symmetry) and the finished row c of X cross DPP rows: one ds_bpermute pair each;
//   * the pivot is read with v_readlane (two SGPRs, transient).
// No LDS storage, no per-lane register arrays of 16, no SGPR arrays: the first version (row per lane, v_readlane
// broadcasts, inverse from an LDS copy) cost 4 + 4 us per tile, most of it spilled SGPRs and serialised LDS reads.
template <int C>
__device__ __forceinline__ double dpp_row_bcast_c(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x150 + C, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x150 + C, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
// One column step.  Registers above the pivot's register (r < R) are finished; registers below it (r > R) take the updates
// unconditionally; only register R itself needs lane-dependent selects (its four rows straddle the pivot row).  The rank-1
// update of T is applied to EVERY entry of the unfinished registers: entries left of / above the pivot become garbage
// that no later step reads (later steps read row C' and column C' of the trailing square only).
template <int C>
__device__ __forceinline__ void chol_inv_step(v4f64 &T, v4f64 &Lo, v4f64 &X, int rg, int cl, int force_c, bool &ok) {
    constexpr int R = C >> 2, Q = C & 3;  // element (C, j) lives in register R of lane (Q, j)
    double d = bcast(T[R], Q * 16 + C);
    d = (C == force_c) ? 1.0 : d;
    ok = ok && (d > 0.0);
    const double dinv = rsqrt_f64(d);
    const double b = __shfl(T[R], Q * 16 + cl) * dinv;   // L[cl][C] for cl >= C (garbage for cl < C: unread)
    const double xs = __shfl(X[R], Q * 16 + cl) * dinv;  // row C of X, final
    const bool colC = cl == C;
#pragma unroll
    for (int r = R; r < 4; ++r) {
        const double a = dpp_row_bcast_c<C>(T[r]) * dinv;  // L[row][C] for row >= C (row == C: sqrt(d))
        T[r] = fma(-a, b, T[r]);
        if (r > R) {
            Lo[r] = colC ? a : Lo[r];
            X[r] = fma(-a, xs, X[r]);
        } else {  // the pivot's own register: rows rg + 4 R, pivot row at rg == Q
            Lo[r] = (colC && rg >= Q) ? a : Lo[r];
            const double xu = fma(-a, xs, X[r]);
            double x = (rg > Q) ? xu : X[r];
            x = (rg == Q) ? xs : x;
            X[r] = x;
        }
    }
}
template <int... Cs>
__device__ __forceinline__ void chol_inv_steps(v4f64 &T, v4f64 &Lo, v4f64 &X, int rg, int cl, int force_c, bool &ok,
                                               std::integer_sequence<int, Cs...>) {
    (chol_inv_step<Cs>(T, Lo, X, rg, cl, force_c, ok), ...);
}
// T: in = the tile, out = its factor L (lower triangle, zeros above).
__device__ __forceinline__ bool chol_inv_tile_acc(v4f64 &T, v4f64 &X, int lane, int force_c) {
    int rg = lane >> 4, cl = lane & 15;
    asm volatile("" : "+v"(rg), "+v"(cl));  // (keeps the lane-index selects out of the enclosing loops' prologue)
    v4f64 Lo = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int r = 0; r < 4; ++r) X[r] = (rg + 4 * r == cl) ? 1.0 : 0.0;
    bool ok = true;
    chol_inv_steps(T, Lo, X, rg, cl, force_c, ok, std::make_integer_sequence<int, 16>{});
    T = Lo;
    return ok;
}

// Outputs of a factored + inverted diagonal tile (both in the accumulator layout): L into the row-major matrix block
// `Cblk` (leading dimension ld; NULL: skip), X = L^-1 into the LDS panel-solve operand dli (stride PS), into the row-major
// block `Wblk` and, transposed, into WdT_k (256 doubles: WdT_k[c][r] = X[r][c]); cs[c] = sum over the first rows_valid rows
// of X[r][c]^2.  Any of Wblk, WdT_k, cs may be NULL.
__device__ __forceinline__ void store_factored_tile(const v4f64 &L, const v4f64 &X, double *Cblk, int ld, double *dli,
                                                    double *Wblk, double *WdT_k, double *cs, int rows_valid, int lane) {
    const int rg = lane >> 4, cl = lane & 15;
    double ssq = 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = rg + 4 * r;
        if (Cblk) Cblk[(size_t)row * ld + cl] = L[r];
        dli[row * PS + cl] = X[r];
        if (Wblk) Wblk[(size_t)row * ld + cl] = X[r];
        if (WdT_k) WdT_k[cl * 16 + row] = X[r];
        if (row < rows_valid) ssq = fma(X[r], X[r], ssq);
    }
    if (cs) {
        ssq += __shfl_xor(ssq, 16);
        ssq += __shfl_xor(ssq, 32);
        if (rg == 0) cs[cl] = ssq;
    }
}

// Fragment loaders (v_mfma_f64_16x16x4_f64: lane = (cl = lane & 15, rg = lane >> 4), k-step s covers k = 4s + rg).
// "Row form": element [k][cl] of a row-major tile -> 4 rows x 16 contiguous doubles per k-step (coalesced).
// t / d from a precomputed rd = RN(1 / d): quotient estimate, exact remainder (fma), one correction -- three dependent
// operations instead of the ~12 of the compiler's division (v_div_scale, v_rcp, two Newton steps, v_div_fmas, v_div_fixup).
// The result is the correctly rounded quotient (Markstein) except for operands whose quotient falls within a hair of a
// rounding boundary; serial recurrences (the pentadiagonal back-substitution) pay the division on every step.
__device__ __forceinline__ double div_rn(double t, double d, double rd) {
    const double q = t * rd;
    const double r = __builtin_fma(-q, d, t);
    return __builtin_fma(r, rd, q);
}

// uniform base + 32-bit byte offset: the global_load / global_store "saddr" form, one 32-bit add per address
__device__ __forceinline__ double ld_off(const gdouble *base, unsigned byte_off) {
    return *reinterpret_cast<const gdouble *>(reinterpret_cast<const __attribute__((address_space(1))) char *>(base) + byte_off);
}
__device__ __forceinline__ void st_off(gdouble *base, unsigned byte_off, double v) {
    *reinterpret_cast<gdouble *>(reinterpret_cast<__attribute__((address_space(1))) char *>(base) + byte_off) = v;
}

// PACKED tiles: the work matrices of the fit loop are stored tile by tile (2 KB each, tile (I, J) at (I nb + J) * 256 doubles) in
// the register layout of the matrix instructions: a lane's four values (accumulator registers 0..3 = fragments of k-steps 0..3:
// element (row 4 q + rg, column cl)) sit in two 16-byte pairs, [q >> 1][lane][q & 1].  A tile is then TWO fully contiguous
// 1 KB accesses of 16 bytes per lane instead of four of 8 bytes over four 128-byte rows of a row-major matrix -- the vector
// memory pipe of the one CU a fit runs on is what its tile products wait for.
typedef double gv2f64 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v4f64 ld_pk(const gdouble *base, unsigned tile_byte_off, int lane) {
    const auto *p = reinterpret_cast<const __attribute__((address_space(1))) gv2f64 *>(
        reinterpret_cast<const __attribute__((address_space(1))) char *>(base) + tile_byte_off + (unsigned)lane * 16u);
    const gv2f64 lo = p[0], hi = p[64];
    return v4f64{lo[0], lo[1], hi[0], hi[1]};
}
__device__ __forceinline__ void st_pk(gdouble *base, unsigned tile_byte_off, int lane, const v4f64 &v) {
    auto *p = reinterpret_cast<__attribute__((address_space(1))) gv2f64 *>(
        reinterpret_cast<__attribute__((address_space(1))) char *>(base) + tile_byte_off + (unsigned)lane * 16u);
    p[0] = gv2f64{v[0], v[1]};
    p[64] = gv2f64{v[2], v[3]};
}

// a pointer the compiler can keep in scalar registers (a select between two kernel arguments otherwise ends up in VGPRs and
// every access pays a 64-bit vector add)
template <typename T>
__device__ __forceinline__ T *uniform_ptr(T *p) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<T *>(((unsigned long long)hi << 32) | lo);
}

struct Frag {
    double v[4];
};
template <typename Ptr>
__device__ __forceinline__ Frag load_rows(Ptr tile, int ld, int cl, int rg) {
    Frag f;
    const auto p = tile + (size_t)rg * ld + cl;
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
template <typename Ptr>
__device__ __forceinline__ void store_tile(Ptr Mx, int ld, int I, int J, const v4f64 &t, int cl, int rg, bool mirror) {
    auto p = Mx + (size_t)(16 * I + rg) * ld + 16 * J + cl;
#pragma unroll
    for (int r = 0; r < 4; ++r) p[(size_t)(4 * r) * ld] = t[r];
    if (mirror) {
        auto q = Mx + (size_t)(16 * J + cl) * ld + 16 * I + rg;
#pragma unroll
        for (int r = 0; r < 4; ++r) q[4 * r] = t[r];
    }
}

}  // namespace tilechol
