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

// ---- the same, transposed outputs (round 4): 20 instead of 48 instructions per column, hand-scheduled --------------------------
// A lone wave issues one instruction per ~4 cycles whatever its kind, in order, and a dependent one ~10 cycles after its
// producer; the step above costs ~48 instructions per column (4.1 k cycles per tile; a pass of the fit loop is a chain of 19
// tiles).  This form keeps Z = X^T = L^-T instead of X:
//   X[i][:] -= L[i][C] X[C][:] / sqrt(d)   reads, in Z's layout, Z's own column C inside a 16-lane DPP row (row_newbcast) and
//   the SAME per-lane multiplier b = T[C][cl] / d the update of T uses -- one cross-row broadcast per column instead of two,
//   no lane-dependent selects of the pivot's register (the mask is per COLUMN index cl), and each update is one
//   v_fmac_f64_dpp (the only 64-bit operation that takes a DPP operand): five per column (registers 0..R of Z, R..3 of T).
// Only 1 / d is needed per column (v_rcp_f64 + one cubic correction folded into b); the pivots stay on T's diagonal and row C
// of X is scaled at the end (Z *= rsqrt(d_cl), one vector rsqrt per tile); positivity is read off the pivots.
// Rows of T travel to the four DPP rows TWO columns ahead (the LDS round trip of ds_bpermute is ~120 cycles, a column ~100):
// row C + 2 is read BEFORE column C's update and receives the two updates it misses as one fma each (w - w[C] b: the
// broadcast row is the same in every DPP row, so its own column C is a row_newbcast) -- bit-identical to what the updates make
// of the row in T.  The chain per column is then  b -> row C + 1 -> its pivot (DPP) -> v_rcp_f64 -> e = 1 - d r -> b, and the
// whole column is ONE asm block in issue order (the compiler sank the next reciprocal behind the updates, and a
// DPP operand needs two wait states behind its VALU write, which inline asm is not scanned for): every dependent pair has
// independent instructions between it, two updates of column C - 1 (registers no row broadcast reads) fill the gaps of
// column C.  Masks are arithmetic (clamp(cl - C)): 64-bit selects need sub-registers, which inline asm cannot name.
// The block ENDS on v_rcp_f64 + s_nop 0: a VALU read of a transcendental's result needs one wait state, the hazard recogniser does
// not look into inline asm, and the FORCE instantiation's selects (compiler-generated, directly behind the block) read r0 -- the
// s_nop closes the hazard inside the block whatever the compiler schedules next (one cycle on a ~100-cycle column).
// T is updated as T[i][j] -= T[i][C] (T[C][j] / d): symmetric to rounding only (both triangles are kept: row C feeds the
// multiplier, column C the DPP operand); the factor's backward error is unchanged.
// Out: Z = L^-T and, if WANT_L, T = L^T, both in the accumulator layout (register r of lane (rg, cl) = element (rg + 4 r, cl)),
// which makes register q of Z the k-step-q A fragment of X = L^-1 (element [cl][4 q + rg]) and the B fragment of X^T.
template <int Q>
__device__ __forceinline__ double xrow(double v, int addr4) {  // the value of lane (Q, cl) to the lanes (*, cl); addr4 = 4 cl
    return __hiloint2double(__builtin_amdgcn_ds_bpermute(addr4 + 64 * Q, __double2hiint(v)),
                            __builtin_amdgcn_ds_bpermute(addr4 + 64 * Q, __double2loint(v)));
}
struct ZState {
    double t[4], z[4];
    double bun;     // row C of T as column C's update finds it (unnormalised), in every DPP row
    double u;       // row C + 1, read before column C - 1's update (that update still missing)
    double d, r0;   // pivot of column C, v_rcp_f64 of it (2^-24)
    double bp;      // b of column C - 1 (two of its updates are still to come)
    double clm;     // cl - C + 1
};
#define FH_DPPC(c) " row_newbcast:%c[" #c "] row_mask:0xf bank_mask:0xf\n"
#define FH_UPD(reg, b, c) "v_fmac_f64_dpp %[" #reg "], -%[" #reg "], %[" #b "]" FH_DPPC(c)
#define FH_Z_OPERANDS                                                                                                           \
    [z0] "+v"(S.z[0]), [z1] "+v"(S.z[1]), [z2] "+v"(S.z[2]), [z3] "+v"(S.z[3]), [t0] "+v"(S.t[0]), [t1] "+v"(S.t[1]),           \
        [t2] "+v"(S.t[2]), [t3] "+v"(S.t[3]), [u] "+v"(S.u), [d] "+v"(S.d), [r0] "+v"(S.r0), [clm] "+v"(S.clm), [b] "=&v"(b),   \
        [x] "=&v"(x), [e] "=&v"(e), [ope] "=&v"(ope)
#define FH_Z_MINI FH_UPD(u, bp, cm)  /* row C + 1: column C - 1's update */
#define FH_Z_NEXT(f0, f1, f2)                                                                                \
    FH_UPD(u, b, cc) /* row C + 1 as this column leaves it */ FH_UPD(f0, b, cc) FH_UPD(f1, b, cc)           \
        "v_mov_b64_dpp %[d], %[u]" FH_DPPC(c1) FH_UPD(f2, b, cc) "v_rcp_f64 %[r0], %[d]\ns_nop 0\n"
template <int C, bool FORCE, bool WANT_L>
__device__ __forceinline__ void chol_z_step(ZState &S, double (&lt)[4], int rg, int cl, int addr4, int force_c) {
    constexpr int R = C >> 2, Q = C & 3, R2 = (C + 2) >> 2, Q2 = (C + 2) & 3;
    double wn = 0.0;
    if constexpr (C + 2 <= 15) wn = xrow<Q2>(S.t[R2], addr4);  // row C + 2, two updates short (consumed in the NEXT block)
    if constexpr (WANT_L) lt[R] = (rg == Q && cl >= C) ? S.bun : lt[R];
    double b, x, e, ope;
    // postponed pair of a column: (t3, z0) while R = 0, (z0, z1) after that -- never a register a row broadcast still reads
    if constexpr (C == 0) {
        asm volatile("v_add_f64 %[clm], %[clm], -1.0\n"
                     "v_max_f64 %[x], %[clm], %[clm] clamp\n"
                     "v_mul_f64 %[x], %[bun], %[x]\n"
                     "v_fma_f64 %[e], -%[d], %[r0], 1.0\n"
                     "v_mul_f64 %[x], %[x], %[r0]\n"
                     "v_add_f64 %[ope], %[e], 1.0\n"
                     "v_mul_f64 %[b], %[x], %[e]\n"
                     "v_fma_f64 %[b], %[b], %[ope], %[x]\n" FH_Z_NEXT(t0, t1, t2)
                     : FH_Z_OPERANDS
                     : [bun] "v"(S.bun), [bp] "v"(S.bp), [cc] "i"(C), [c1] "i"(C + 1), [cm] "i"(0));
    } else if constexpr (C < 15) {
#define FH_Z_COL(p0, p1, f0, f1, f2, TAIL)                                                                              \
    asm volatile("v_add_f64 %[clm], %[clm], -1.0\n"                                                                     \
                 "v_max_f64 %[x], %[clm], %[clm] clamp\n" FH_Z_MINI "v_mul_f64 %[x], %[bun], %[x]\n"                     \
                 "v_fma_f64 %[e], -%[d], %[r0], 1.0\n"                                                                  \
                 "v_mul_f64 %[x], %[x], %[r0]\n"                                                                        \
                 "v_add_f64 %[ope], %[e], 1.0\n"                                                                        \
                 "v_mul_f64 %[b], %[x], %[e]\n" FH_UPD(p0, bp, cm) "v_fma_f64 %[b], %[b], %[ope], %[x]\n"                \
                     FH_UPD(p1, bp, cm) TAIL                                                                            \
                 : FH_Z_OPERANDS                                                                                        \
                 : [bun] "v"(S.bun), [bp] "v"(S.bp), [cc] "i"(C), [c1] "i"(C + 1), [cm] "i"(C - 1))
        if constexpr (C <= 3)
            FH_Z_COL(t3, z0, t0, t1, t2, FH_Z_NEXT(t0, t1, t2));
        else if constexpr (C == 4)  // (column 3 left t3 and z0)
            FH_Z_COL(t3, z0, t1, t2, t3, FH_Z_NEXT(t1, t2, t3));
        else if constexpr (C <= 7)
            FH_Z_COL(z0, z1, t1, t2, t3, FH_Z_NEXT(t1, t2, t3));
        else if constexpr (C <= 11)
            FH_Z_COL(z0, z1, t2, t3, z2, FH_Z_NEXT(t2, t3, z2));
        else if constexpr (C <= 13)
            FH_Z_COL(z0, z1, t3, z2, z3, FH_Z_NEXT(t3, z2, z3));
        else  // column 14: no pivot needed any more (it stays on T's diagonal)
            FH_Z_COL(z0, z1, t3, z2, z3, FH_UPD(u, b, cc) FH_UPD(t3, b, cc) FH_UPD(z2, b, cc) FH_UPD(z3, b, cc));
#undef FH_Z_COL
    } else {  // what column 14 left
        asm volatile(FH_UPD(z0, bp, cm) FH_UPD(z1, bp, cm) : [z0] "+v"(S.z[0]), [z1] "+v"(S.z[1]) : [bp] "v"(S.bp), [cm] "i"(14));
    }
    if constexpr (C < 15) {
        S.bun = S.u;
        S.u = wn;
        S.bp = b;
        if constexpr (FORCE) {
            const bool f = C + 1 == force_c;
            S.d = f ? 1.0 : S.d;
            S.r0 = f ? 1.0 : S.r0;
        }
    }
}
#undef FH_Z_MINI
#undef FH_Z_NEXT
#undef FH_Z_OPERANDS
#undef FH_UPD
#undef FH_DPPC
template <bool FORCE, bool WANT_L, int... Cs>
__device__ __forceinline__ void chol_z_steps(ZState &S, double (&lt)[4], int rg, int cl, int addr4, int force_c,
                                             std::integer_sequence<int, Cs...>) {
    (chol_z_step<Cs, FORCE, WANT_L>(S, lt, rg, cl, addr4, force_c), ...);
}
template <bool FORCE, bool WANT_L>
__device__ __forceinline__ bool chol_inv_tile_z(v4f64 &T, v4f64 &Z, int lane, int force_c) {
    int rg = lane >> 4, cl = lane & 15;
    asm volatile("" : "+v"(rg), "+v"(cl));
    const int addr4 = cl * 4;
    ZState S;
    double lt[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int r = 0; r < 4; ++r) S.t[r] = T[r], S.z[r] = (rg + 4 * r == cl) ? 1.0 : 0.0;
    S.bun = xrow<0>(S.t[0], addr4);
    S.u = xrow<1>(S.t[0], addr4);
    S.d = dpp_row_bcast_c<0>(S.bun);
    if constexpr (FORCE) S.d = (0 == force_c) ? 1.0 : S.d;
    S.r0 = __builtin_amdgcn_rcp(S.d);
    S.bp = 0.0;
    S.clm = (double)(cl + 1);
    chol_z_steps<FORCE, WANT_L>(S, lt, rg, cl, addr4, force_c, std::make_integer_sequence<int, 16>{});
    // the pivots are still on T's diagonal (lane cl of a row of T is masked from column cl on): lane (cl & 3, cl) of register
    // cl >> 2 -> every lane of column cl
    const int rsel = cl >> 2;
    double dvec = rsel == 0 ? S.t[0] : (rsel == 1 ? S.t[1] : (rsel == 2 ? S.t[2] : S.t[3]));
    dvec = __shfl(dvec, (cl & 3) * 16 + cl);
    if constexpr (FORCE) dvec = (cl == force_c) ? 1.0 : dvec;
    const double dc = rsqrt_f64(dvec);  // 1 / sqrt(d_cl): scales column cl of Z (row cl of X) ...
#pragma unroll
    for (int r = 0; r < 4; ++r) Z[r] = S.z[r] * dc;
    if constexpr (WANT_L) {  // ... and, moved to the lanes' ROW indices, the rows of L^T
#pragma unroll
        for (int r = 0; r < 4; ++r) T[r] = lt[r] * __shfl(dc, rg + 4 * r);
    }
    return __builtin_amdgcn_ballot_w64(!(dvec > 0.0 && dvec < __builtin_inf())) == 0;
}

// Z = L^-T (accumulator layout) -> X = L^-1 row-major in the LDS panel-solve operand dli (stride PS), and X back in the
// accumulator layout (a wave's DS operations execute in order: its own reads behind its own writes need no barrier).
__device__ __forceinline__ v4f64 store_inverse_z(const v4f64 &Z, double *dli, int lane) {
    const int rg = lane >> 4, cl = lane & 15;
#pragma unroll
    for (int q = 0; q < 4; ++q) dli[cl * PS + rg + 4 * q] = Z[q];  // Z[rg + 4 q][cl] = X[cl][rg + 4 q]
    v4f64 X;
#pragma unroll
    for (int r = 0; r < 4; ++r) X[r] = dli[(rg + 4 * r) * PS + cl];
    return X;
}
// cs[c] = sum over the first rows_valid rows of X[r][c]^2 (X in the accumulator layout)
__device__ __forceinline__ void store_col_ssq(const v4f64 &X, double *cs, int rows_valid, int lane) {
    const int rg = lane >> 4, cl = lane & 15;
    double ssq = 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (rg + 4 * r < rows_valid) ssq = fma(X[r], X[r], ssq);
    ssq += __shfl_xor(ssq, 16);
    ssq += __shfl_xor(ssq, 32);
    if (rg == 0) cs[cl] = ssq;
}
// factor + invert a diagonal tile: X = L^-1 into dli and (accumulator layout) x; T is consumed.  force_c >= 0 (wave-uniform): the
// pivot of that column is taken as 1 (the augmented row of the fit loop).
__device__ __forceinline__ bool factor_invert_tile(v4f64 &T, v4f64 &x, double *dli, int lane, int force_c) {
    v4f64 z;
    bool ok;
    if (force_c >= 0)
        ok = chol_inv_tile_z<true, false>(T, z, lane, force_c);
    else
        ok = chol_inv_tile_z<false, false>(T, z, lane, -1);
    x = store_inverse_z(z, dli, lane);
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
