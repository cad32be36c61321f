// lognormal.hip -- method='LogNormal': the MAP of the log-brightness by Newton's method with back-tracking, and the
// power-spectrum iteration around it, one persistent workgroup per fit.
//
//   LogNormalMAPModel._fit        frank/statistical_models.py:1064-1160   (H, jac, hess, limit_step)
//   MinimizeNewton / LineSearch   frank/minimizer.py:45-283
//   FrankFitter._fit (LogNormal)  frank/radial_fitters.py:754-785
//   CriticalFilter.update_power_spectrum  frank/filter.py:154-177
//
// The algorithm is a serial chain of small dense operations (N <= 320): ~1e5 Newton steps, ~1e6 function evaluations
// and ~1e4 LU factorisations per fit, each depending on the last.  Nothing here is MFMA-bound: for small N the cost is
// latency, from N ~ 100 it is what one CU can stream from L2 (three N x N matrices per Newton step at ~13 B/clk).  The
// whole chain lives in ONE workgroup: state in LDS, the LU factors in LDS when they fit (N <= 112) or in L2 otherwise,
// every reduction in a fixed order so that all lanes take the same branch, no host round trip until the fit is done.
// A factorisation that keeps being re-used is turned into the explicit inverse (minimize_newton).  Throughput comes
// from running independent fits (sweeps, bootstraps) on the other 255 CUs: the batched launch pulls fits from a queue.
// The hot loop has to stay inside the 64 KB instruction cache (one evaluation site, rolled substitution chains).
//
// Differences from the reference that do not change the mathematics: scipy's lu_factor (LAPACK getrf) is an
// unblocked partial-pivoting LU here; the posterior covariance D = hess(s_MAP)^-1 is applied through that LU
// instead of a Cholesky factor (the reference falls back to an SVD inverse when the Cholesky fails,
// statistical_models.py:1150-1158 -- the same matrix); jac(x) re-uses the products of the accepted fun(x).
#include <hip/hip_runtime.h>
#include <rocprim/warp/warp_reduce.hpp>

#include "band_scan.h"
#include "kernels.h"
#include "tile_chol.h"

#pragma clang fp contract(off)

typedef double v2f64 __attribute__((ext_vector_type(2)));
typedef int v4i32 __attribute__((ext_vector_type(4)));

namespace {

#ifdef LN_TIMING
__shared__ long long ln_cyc[12];  // 0 eval, 1 factorisations, 2 solves, 3 Hessians, 4 Newton, 5-7 pivoted LU, 8 S^-1, 9 Tr2, 10 Tr1, 11 bands + exp
#define LTIC() const long long _t0 = clock64()
#define LTOC(k) do { if (threadIdx.x == 0) ln_cyc[k] += clock64() - _t0; } while (0)
__shared__ long long ln_ev[4];  // inside an evaluation: until the command is out, own items, waiting for the helpers, combining
__shared__ long long ln_ch[8];  // inside the Cholesky: chain until its flag, chain's global outputs, chain at the barrier; worker wave 1: trailing
                                // tiles, column tiles until the wait, waiting for the flag, panel, at the barrier
#define CHT(k) do { if (lane == 0) { const long long n_ = clock64(); ln_ch[k] += n_ - _tc; _tc = n_; } } while (0)
#define EVT(k) do { if (threadIdx.x == 0) { const long long n_ = clock64(); ln_ev[k] += n_ - _te; _te = n_; } } while (0)
#else
#define LTIC() do {} while (0)
#define LTOC(k) do {} while (0)
#define CHT(k) do {} while (0)
#endif

constexpr int LT = 512;        // threads per workgroup
constexpr int LNW = LT / 64;   // waves
// The thread index through an opaque move: every routine of this (one, very large) kernel forms its per-thread addresses where
// it uses them.  Taken from threadIdx.x directly they are all invariants of the outer loops: hoisted to the top of the kernel,
// kept alive across every phase and spilled (262 registers), with reloads inside the row loops.
__device__ __forceinline__ int ln_tid() {
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
}
#ifndef LN_HVB
#define LN_HVB 16
#endif
constexpr int HVB = LN_HVB;  // columns per load batch of the inverse product
constexpr int LS_MAX_TRIALS = 2000;  // back-tracking guard: lam shrinks >= 10x per trial, x + lam p == x long before

struct LnS {
    double *x, *xn, *I, *In, *Sx, *Sxn, *MI, *MIn, *jx, *dx, *pd, *jv, *p, *pold, *rhs, *tr2, *col, *rowk, *rdiag, *red;
    double *part;  // 2 * LT: per-chunk partial products of ln_eval
    double *wsol;  // LNW * N: per-wave solve vectors
    int *perm;
    double *lu;    // N*N column-major: LDS or global
    double *pan;   // global-LU kernels: LDS panel of the blocked factorisation (N * LU_NB)
    double *stage;  // WIDE: 64 x 64 doubles of LDS: the diagonal block of the factors the next substitution chain runs on (block_solve_wide)
    double *cpan, *bak;  // WIDE: the Cholesky's one panel ALIASES the eighteen vectors behind rdiag (they wait in `bak`, global, meanwhile)
    double *chol;  // LDS of the tiled Cholesky (cholesky_as_lu): N <= 320 the solve vectors' space and the LU panel's behind it; WIDE: its own
    int lu_nb;
    int redsel;
    int row, c0, c1, slot;  // ln_eval work split: this thread sums columns [c0, c1) of output `row` into pbuf[slot]
    int nch;
    // pair mode (N even, 256 < N: one chunk of N threads would leave 40 % of the workgroup idle and read 8 B per lane): a
    // thread owns rows `row`, `row + 1` (one 16-byte load per column and matrix) and one of nch column chunks; the
    // partials then need 2 * nch * N doubles and live in the solve vectors' space (idle during the products)
    int pair, pstride;
    double *pbuf;
    // cluster (see "cluster: a few workgroups on one fit" below): workgroups sharing this fit's parallel pieces, whether they
    // sit on one XCD, the flag word of the hand-overs
    int cluster;
    int seq;  // commands dispatched so far (cluster_dispatch)
    int chol_epoch;  // distributed factorisations so far (cholesky_as_lu / chol_helper count them separately, in step)
    bool same_xcd;
    int *s_cl;
};
enum { LN_CMD_NONE = 0, LN_CMD_SINV = 1, LN_CMD_TR2 = 2, LN_CMD_EXIT = 3, LN_CMD_HESS = 4, LN_CMD_EVAL = 5, LN_CMD_CHOL = 6, LN_CMD_WITH_S = 16 };
__device__ __forceinline__ void cluster_dispatch(const LogNormalParams &P, int cmd, bool same_xcd, int &seq);
__device__ __forceinline__ bool cluster_wait(const LogNormalParams &P, int *s_flag, bool same_xcd);

// Wave reductions through DPP row operations (rocprim), result broadcast to every lane.
template <class T, class Op>
__device__ __forceinline__ T wave_reduce(T v, Op op) {
    using WR = rocprim::warp_reduce<T, 64, true>;
    typename WR::storage_type st;
    T out;
    WR().reduce(v, out, st, op);
    return out;
}
struct FMax {  // operands never NaN
    __device__ __forceinline__ double operator()(double a, double b) const { return __builtin_fmax(a, b); }
};
__device__ __forceinline__ double wave_sum(double v) { return wave_reduce(v, rocprim::plus<double>()); }

// Block reductions: every thread returns the same bits (wave partials combined in wave order by every thread).
// Two partial buffers used alternately: a thread can only reach the next-but-one reduction after every thread has
// passed the barrier of the next one, i.e. has finished reading this one -- no trailing barrier needed.
__device__ __forceinline__ double *red_buf(LnS &S) {
    S.redsel ^= 1;
    return S.red + 32 * S.redsel;
}

__device__ __forceinline__ double block_sum(LnS &S, double v) {
    double *red = red_buf(S);
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = 0.0;
#pragma unroll
    for (int w = 0; w < LNW; ++w) r += red[w];
    return r;
}

__device__ __forceinline__ void block_sum3(LnS &S, double &a, double &b, double &c) {
    double *red = red_buf(S);
    a = wave_sum(a);
    b = wave_sum(b);
    c = wave_sum(c);
    if ((threadIdx.x & 63) == 0) {
        const int w = threadIdx.x >> 6;
        red[3 * w] = a;
        red[3 * w + 1] = b;
        red[3 * w + 2] = c;
    }
    __syncthreads();
    double ra = 0.0, rb = 0.0, rc = 0.0;
#pragma unroll
    for (int w = 0; w < LNW; ++w) {
        ra += red[3 * w];
        rb += red[3 * w + 1];
        rc += red[3 * w + 2];
    }
    a = ra;
    b = rb;
    c = rc;
}

template <bool IS_MAX>
__device__ __forceinline__ double block_minmax(LnS &S, double v) {
    double *red = red_buf(S);
    v = IS_MAX ? wave_reduce(v, rocprim::maximum<double>()) : wave_reduce(v, rocprim::minimum<double>());
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = red[0];
#pragma unroll
    for (int w = 1; w < LNW; ++w) r = IS_MAX ? fmax(r, red[w]) : fmin(r, red[w]);
    return r;
}

// min of a and sum of b in one pass (step limiter and slope of the line search)
__device__ __forceinline__ void block_min_sum(LnS &S, double &a, double &b) {
    double *red = red_buf(S);
    a = wave_reduce(a, rocprim::minimum<double>());
    b = wave_sum(b);
    if ((threadIdx.x & 63) == 0) {
        red[2 * (threadIdx.x >> 6)] = a;
        red[2 * (threadIdx.x >> 6) + 1] = b;
    }
    __syncthreads();
    double ra = red[0], rb = red[1];
#pragma unroll
    for (int w = 1; w < LNW; ++w) {
        ra = fmin(ra, red[2 * w]);
        rb += red[2 * w + 1];
    }
    a = ra;
    b = rb;
}

// H(s) = 1/2 s^T S^-1 s + 1/2 I^T M I - j^T I,  I = exp(s + s0)   (statistical_models.py:1075-1085)
// Iv = exp(xv + s0) is the caller's; leaves S^-1 s and M I of the point in Sxv, MIv (gradient and Hessian re-use them).
// The two matrix-vector products are split by column chunks over all threads: thread (row, chunk) walks down a
// COLUMN of the symmetric matrices, so a wave reads consecutive addresses and nothing crosses lanes.
//
// One CU pulls ~60 GB/s out of L2, so an evaluation costs what its matrix bytes cost, and the back-tracking line search
// evaluates ~15 points x + lam p per Newton step at full size.  S^-1 is linear: the first trial of a search multiplies
// S^-1 with the DIRECTION (sv = p, sdst = S^-1 p) and every trial forms S^-1 (x + lam p) = S^-1 x + lam S^-1 p from the
// cached S^-1 x (along = true); later trials (sv = NULL) read M only.  M exp(.) is recomputed for every point.
//   sv    vector multiplied with S^-1 (NULL: none), its product goes to sdst
//   along S^-1 xv = S.Sx + lam * S.col   (S.col: S^-1 p of this search; the unblocked LU's scratch, idle here)
// The work of the two products is cut into ITEMS -- a row unit (two rows for even N: one 16-byte load per column and matrix) x a
// chunk of 16 columns -- whose partial sums go to a scratch in global memory, [matrix][chunk][row]; a row's sum is the sum of
// its chunks in order.  One decomposition whoever computes the items: a workgroup alone takes them all (thread t: items t,
// t + 512, ...), a cluster deals them round its workgroups -- the same bits (test_lognormal_cluster_equals_single_workgroup).
// Before, a thread walked a third of a column (100 matrix entries, a dozen L2 round trips one behind the other) and one CU
// pulled both matrices alone: 30 k cycles per evaluation at N = 300, a sixth of a default-mode fit, half of a reference-mode one.
constexpr int EVW = 16;  // columns per item
// scratch behind the factors (fh_ln_lu_doubles): bands and scan tables of the pentadiagonal solve, the two vectors and the
// partial sums of the evaluations
__device__ __forceinline__ double *ln_band_scratch(const LogNormalParams &P) {
    return P.LU + (size_t)P.N * P.N + 2 * (size_t)P.NP * P.NP + 16 * (size_t)P.NP;
}
__device__ __forceinline__ double *ln_eval_vecs(const LogNormalParams &P) { return ln_band_scratch(P) + 6 * P.NP + bandscan::kTableDoubles; }
__device__ __forceinline__ double *ln_eval_parts(const LogNormalParams &P) { return ln_eval_vecs(P) + 2 * P.NP; }
// items gthread, gthread + gstride, ...; svv: the vector S^-1 multiplies (NULL: that product is not wanted), Ivv: I -- both
// in LDS (the multipliers are read there, at a uniform address, as the sums are formed).  The 16 columns of an item are loaded
// in two batches of eight per matrix (sixteen at once held 190 registers, and the kernel spilled elsewhere for it).
__device__ __forceinline__ void ln_eval_items(const LogNormalParams &P, const double *svv, const double *Ivv, int gthread, int gstride) {
    const int N = P.N, NP = P.NP, nch = (N + EVW - 1) / EVW;
    constexpr int EVH = EVW / 2;
    double *part_s = ln_eval_parts(P), *part_m = part_s + (size_t)nch * NP;
    if ((N & 1) == 0) {
        const int RU = N >> 1, N2 = N >> 1, items = RU * nch;
        for (int id = gthread; id < items; id += gstride) {
            const int ch = id / RU, ru = id - ch * RU, c0 = ch * EVW;
            const v2f64 *mc = reinterpret_cast<const v2f64 *>(__builtin_assume_aligned(P.M + (size_t)c0 * N + 2 * ru, 16));
            const v2f64 *sc = reinterpret_cast<const v2f64 *>(__builtin_assume_aligned(P.Sinv + (size_t)c0 * N + 2 * ru, 16));
            v2f64 a = {0.0, 0.0}, b = {0.0, 0.0};
#pragma unroll 1
            for (int h = 0; h < EVW; h += EVH) {
                v2f64 vm[EVH], vs[EVH];
#pragma unroll
                for (int u = 0; u < EVH; ++u) {  // (every load issued: columns past N are clamped and multiplied by zero)
                    const int cu = min(c0 + h + u, N - 1) - c0;
                    vm[u] = mc[cu * N2];
                    if (svv) vs[u] = sc[cu * N2];
                }
#pragma unroll
                for (int u = 0; u < EVH; ++u) {
                    const int c = c0 + h + u;
                    const double xi = (c < N) ? Ivv[min(c, N - 1)] : 0.0;
                    b[0] = fma(vm[u][0], xi, b[0]);
                    b[1] = fma(vm[u][1], xi, b[1]);
                    if (svv) {
                        const double xs = (c < N) ? svv[min(c, N - 1)] : 0.0;
                        a[0] = fma(vs[u][0], xs, a[0]);
                        a[1] = fma(vs[u][1], xs, a[1]);
                    }
                }
            }
            if (svv) *reinterpret_cast<v2f64 *>(__builtin_assume_aligned(part_s + (size_t)ch * NP + 2 * ru, 16)) = a;
            *reinterpret_cast<v2f64 *>(__builtin_assume_aligned(part_m + (size_t)ch * NP + 2 * ru, 16)) = b;
        }
    } else {
        const int items = N * nch;
        for (int id = gthread; id < items; id += gstride) {
            const int ch = id / N, r = id - ch * N, c0 = ch * EVW;
            const double *mc = P.M + (size_t)c0 * N + r, *sc = P.Sinv + (size_t)c0 * N + r;
            double a = 0.0, b = 0.0;
#pragma unroll 1
            for (int h = 0; h < EVW; h += EVH) {
                double vm[EVH], vs[EVH];
#pragma unroll
                for (int u = 0; u < EVH; ++u) {
                    const int cu = min(c0 + h + u, N - 1) - c0;
                    vm[u] = mc[cu * N];
                    if (svv) vs[u] = sc[cu * N];
                }
#pragma unroll
                for (int u = 0; u < EVH; ++u) {
                    const int c = c0 + h + u;
                    b = fma(vm[u], (c < N) ? Ivv[min(c, N - 1)] : 0.0, b);
                    if (svv) a = fma(vs[u], (c < N) ? svv[min(c, N - 1)] : 0.0, a);
                }
            }
            if (svv) part_s[(size_t)ch * NP + r] = a;
            part_m[(size_t)ch * NP + r] = b;
        }
    }
}
// WIDE (320 < N <= 640, round 6): the vectors live in global memory (L2), a thread may own two rows, a row has up to 40 chunks
template <bool WIDE>
__device__ __forceinline__ double ln_eval(const LogNormalParams &P, LnS &S, const double *xv, double *Iv, double *Sxv, double *MIv,
                                          const double *sv, double *sdst, bool along, double lam) {
    const int N = P.N, NP = P.NP, tid = ln_tid(), nch = (N + EVW - 1) / EVW;
    LTIC();
#ifdef LN_TIMING
    long long _te = clock64();
#else
#define EVT(k) do { } while (0)
#endif
    bool shared = false;
    if (S.cluster > 1) {  // the helpers read the two vectors from global memory
        double *vecs = ln_eval_vecs(P);
        for (int i = tid; i < N; i += LT) {
            if (sv) vecs[i] = sv[i];
            vecs[NP + i] = Iv[i];
        }
        cluster_dispatch(P, LN_CMD_EVAL | (sv ? LN_CMD_WITH_S : 0), S.same_xcd, S.seq);
        EVT(0);
        ln_eval_items(P, sv, Iv, tid, S.cluster * LT);
        EVT(1);
        shared = cluster_wait(P, S.s_cl, S.same_xcd);
        EVT(2);
        if (!shared) {  // the helpers did not answer: disband, and take every item here
            if (tid == 0) __hip_atomic_store(&P.ctl[4], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            S.cluster = 1;
        }
    }
    if (!shared) ln_eval_items(P, sv, Iv, tid, LT);
    __syncthreads();
    double A = 0.0, B = 0.0, C = 0.0;
    if constexpr (WIDE) {
        const double *part_s = ln_eval_parts(P), *part_m = part_s + (size_t)nch * NP;
        constexpr int CHB = 10;  // chunks per batch of loads (clamped, never predicated); the sums run over the chunks in order
        for (int r = tid; r < N; r += LT) {
            double a = 0.0, b = 0.0;
            for (int c0 = 0; c0 < nch; c0 += CHB) {
                double pm[CHB], ps[CHB];
#pragma unroll
                for (int ch = 0; ch < CHB; ++ch) {
                    pm[ch] = part_m[(size_t)min(c0 + ch, nch - 1) * NP + r];
                    if (sv) ps[ch] = part_s[(size_t)min(c0 + ch, nch - 1) * NP + r];
                }
#pragma unroll
                for (int ch = 0; ch < CHB; ++ch)
                    if (c0 + ch < nch) {
                        b += pm[ch];
                        if (sv) a += ps[ch];
                    }
            }
            if (sv) sdst[r] = a;
            if (along) {
                a = fma(lam, S.col[r], S.Sx[r]);
                Sxv[r] = a;
            }
            MIv[r] = b;
            A += xv[r] * a;
            B += Iv[r] * b;
            C += Iv[r] * S.jv[r];
        }
    } else
    if (tid < N) {
        const double *part_s = ln_eval_parts(P), *part_m = part_s + (size_t)nch * NP;
        // (the chunks of a row are loaded as one batch -- clamped, never predicated -- and added in order: one L2 round trip,
        //  not one per chunk)
        constexpr int MAXCH = 20;  // N <= 320
        double a = 0.0, b = 0.0, pm[MAXCH];
#pragma unroll
        for (int ch = 0; ch < MAXCH; ++ch) pm[ch] = part_m[(size_t)min(ch, nch - 1) * NP + tid];
        if (sv) {
            double ps[MAXCH];
#pragma unroll
            for (int ch = 0; ch < MAXCH; ++ch) ps[ch] = part_s[(size_t)min(ch, nch - 1) * NP + tid];
#pragma unroll
            for (int ch = 0; ch < MAXCH; ++ch)
                if (ch < nch) a += ps[ch];
            sdst[tid] = a;
        }
#pragma unroll
        for (int ch = 0; ch < MAXCH; ++ch)
            if (ch < nch) b += pm[ch];
        if (along) {
            a = fma(lam, S.col[tid], S.Sx[tid]);
            Sxv[tid] = a;
        }
        MIv[tid] = b;
        A = xv[tid] * a;
        B = Iv[tid] * b;
        C = Iv[tid] * S.jv[tid];
    }
    block_sum3(S, A, B, C);
    double f = 0.5 * A;
    f += 0.5 * B;
    f -= C;
    EVT(3);
    LTOC(0);
    return f;
}

// jac(s) = S^-1 s + (I (M I) - I j)   (statistical_models.py:1087-1098), from the cached products of S.x
__device__ __forceinline__ double ln_grad(const LnS &S, int i) {
    return S.Sx[i] + (S.I[i] * S.MI[i] - S.I[i] * S.jv[i]);
}

// hess(s) = I_a M_ab I_b + delta_ab (I_a (M I)_a - I_a j_a) + S^-1_ab   (statistical_models.py:1100-1122), at S.x.
// M and S^-1 are symmetric bit for bit, the product I_a M_ab I_b is not (two roundings, in an order): out[b * ld + a] is
// (I_a M_ab) I_b -- the column-major Hessian the factorisations work on -- or, with outer_first, (I_b M_ba) I_a: the same
// bits as element [a][b] of the former, i.e. its row-major image (the Dinv handed to the host), written along rows too.
// ld = P.NP: the padded copy for the tiled Cholesky (its padding rows and columns hold the identity: written once per
// kernel, the factorisation leaves them as they are).
// (rows part, part + nparts, ... of the wave-by-wave deal: the cluster's workgroups build disjoint rows, bit for bit what one
//  workgroup builds; Iv, MIv, jv: I, M I and j -- LDS for the workgroup that runs the fit, global copies for its helpers)
__device__ __forceinline__ void build_hess_rows(const LogNormalParams &P, const double *Iv, const double *MIv, const double *jv,
                                                double *out, int ld, bool outer_first, int part, int nparts) {
    const int N = P.N, tid = ln_tid();
    if ((N & 1) == 0) {
        // one wave per row, 16 bytes per lane and matrix, the loads of TWO rows issued before the first store: a single CU
        // streams from L2 at the rate its requests in flight allow (8-byte loads interleaved with stores: 131 us per
        // Hessian at N = 300; this: ~40 us, which is what 2.2 MB cost one CU)
        constexpr int HB = 3;  // 64 * HB pairs per pass: one pass for N <= 384
        const int lane = tid & 63, N2 = N >> 1;
        const int rstride = LNW * nparts;
        for (int b0 = part * LNW + __builtin_amdgcn_readfirstlane(tid >> 6); b0 < N; b0 += 2 * rstride) {
            for (int a0 = 0; a0 < N2; a0 += 64 * HB) {
                v2f64 vm[2][HB], vs[2][HB];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int b = b0 + h * rstride;
                    if (b < N) {
                        const v2f64 *mb = reinterpret_cast<const v2f64 *>(__builtin_assume_aligned(P.M + b * N, 16));
                        const v2f64 *sb = reinterpret_cast<const v2f64 *>(__builtin_assume_aligned(P.Sinv + b * N, 16));
#pragma unroll
                        for (int u = 0; u < HB; ++u) {
                            const int idx = a0 + 64 * u + lane;
                            if (idx < N2) {
                                vm[h][u] = mb[idx];
                                vs[h][u] = sb[idx];
                            }
                        }
                    }
                }
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int b = b0 + h * rstride;
                    if (b < N) {
                        const double Ib = Iv[b];
#pragma unroll
                        for (int u = 0; u < HB; ++u) {
                            const int idx = a0 + 64 * u + lane;
                            if (idx < N2) {
                                v2f64 v;
#pragma unroll
                                for (int e = 0; e < 2; ++e) {
                                    const int a = 2 * idx + e;
                                    double t = outer_first ? Ib * vm[h][u][e] * Iv[a] : Iv[a] * vm[h][u][e] * Ib;
                                    if (a == b) t += Iv[a] * MIv[a] - Iv[a] * jv[a];
                                    t += vs[h][u][e];
                                    v[e] = t;
                                }
                                *reinterpret_cast<v2f64 *>(__builtin_assume_aligned(out + (size_t)b * ld + 2 * idx, 16)) = v;
                            }
                        }
                    }
                }
            }
        }
    } else {
        for (int b = part * (LT / 32) + (tid >> 5); b < N; b += (LT / 32) * nparts) {
            const double Ib = Iv[b];
            const double *mb = P.M + b * N, *sb = P.Sinv + b * N;
            for (int a = tid & 31; a < N; a += 32) {
                double v = outer_first ? Ib * mb[a] * Iv[a] : Iv[a] * mb[a] * Ib;
                if (a == b) v += Iv[a] * MIv[a] - Iv[a] * jv[a];
                v += sb[a];
                out[(size_t)b * ld + a] = v;
            }
        }
    }
}
// The copy of the Hessian that the tiled Cholesky factors (cholesky_as_lu), in PACKED 16 x 16 tiles (tile_chol.h: tile (I, J) at
// (I nb + J) * 256 doubles, two contiguous 1 KB halves in the register layout of the matrix instructions): the lower tiles
// I >= J and the tiles (0, I) of the first block row -- what the factorisation reads before it has written it.  Element (row b,
// column a) is (I_a M_ab) I_b + ..., the bits of build_hess_rows(.., outer_first = false); rows and columns past N hold the
// identity.  A wave builds whole tiles: tiles gw, gw + nwaves, .. of one enumeration (the cluster deals them round its
// workgroups: the same bits whoever builds a tile).  Round 5 kept this copy row-major: a tile of the factorisation was then four
// 8-byte loads per lane over four 128-byte rows, ~3 000 cycles per trailing tile against ~600 for a packed one (LABNOTES).
__device__ __forceinline__ void build_hess_tiles(const LogNormalParams &P, const double *Iv, const double *MIv, const double *jv,
                                                 double *Cp, int part, int nparts) {
    using namespace tilechol;
    const int N = P.N, nb = P.NP / 16;
    const int tid = ln_tid(), lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cl = lane & 15, rg = lane >> 4;
    const int nlow = nb * (nb + 1) / 2, ntiles = nlow + nb - 1;
    gdouble *Cg = as_global(Cp);
    const gdouble *Mg = as_global(P.M), *Sg = as_global(P.Sinv);
    for (int t = part * LNW + wave; t < ntiles; t += nparts * LNW) {
        int I, J;
        if (t < nlow) {
            I = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
            while ((I + 1) * (I + 2) / 2 <= t) ++I;
            while (I * (I + 1) / 2 > t) --I;
            J = t - I * (I + 1) / 2;
        } else {
            I = 0;
            J = t - nlow + 1;
        }
        const int a = 16 * J + cl, ac = min(a, N - 1);
        double vm[4], vs[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {  // (every load issued: rows and columns past N are clamped and replaced below)
            const int bc = min(16 * I + rg + 4 * q, N - 1);
            vm[q] = Mg[(size_t)bc * N + ac];
            vs[q] = Sg[(size_t)bc * N + ac];
        }
        const double Ia = Iv[ac];
        v4f64 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int b = 16 * I + rg + 4 * q;
            double v = Ia * vm[q] * Iv[min(b, N - 1)];
            if (a == b) v += Ia * MIv[ac] - Ia * jv[ac];
            v += vs[q];
            o[q] = (a < N && b < N) ? v : (a == b ? 1.0 : 0.0);
        }
        st_pk(Cg, (unsigned)((I * nb + J) * 2048), lane, o);
    }
}
// element (i, j) of the padded matrix in the packed layout (doubles)
__device__ __forceinline__ size_t pk_elem(int i, int j, int nb) {
    const int r = i & 15, c = j & 15, q = r >> 2, ln = (r & 3) * 16 + c;
    return ((size_t)((i >> 4) * nb + (j >> 4))) * 256 + (size_t)((q >> 1) * 128 + ln * 2 + (q & 1));
}
__device__ __forceinline__ void build_hess(const LogNormalParams &P, LnS &S, double *out, int ld, bool outer_first) {
    LTIC();
    build_hess_rows(P, S.I, S.MI, S.jv, out, ld, outer_first, 0, 1);
    __syncthreads();
    LTOC(3);
}
// The packed copy the tiled Cholesky factors, built by the whole cluster (single fits): I and M I go to the helpers through the
// two global vectors that S^-1 and Tr2 use at other times; a cluster that does not answer is disbanded and the copy rebuilt here.
__device__ __forceinline__ void build_hess_padded(const LogNormalParams &P, LnS &S, double *Cp) {
    LTIC();
    if (S.cluster <= 1) {
        build_hess_tiles(P, S.I, S.MI, S.jv, Cp, 0, 1);
        __syncthreads();
        LTOC(3);
        return;
    }
    for (int i = ln_tid(); i < P.N; i += LT) {
        P.rk_g[i] = S.I[i];
        P.tr2_g[i] = S.MI[i];
    }
    cluster_dispatch(P, LN_CMD_HESS, S.same_xcd, S.seq);
    build_hess_tiles(P, S.I, S.MI, S.jv, Cp, 0, S.cluster);
    if (!cluster_wait(P, S.s_cl, S.same_xcd)) {
        if (ln_tid() == 0) __hip_atomic_store(&P.ctl[4], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // disbanded
        S.cluster = 1;
        build_hess_tiles(P, S.I, S.MI, S.jv, Cp, 0, 1);
        __syncthreads();
    }
    LTOC(3);
}

// ---- Cholesky first ------------------------------------------------------------------------------------------------
// scipy's lu_factor (minimizer.py:238) costs the blocked LU below 3.3 M cycles at N = 300, two thirds of it in the 300
// pivot-search steps of the panels.  Near the MAP the Hessian diag(I) M diag(I) + diag(.) + S^-1 is positive definite
// (it is what LogNormalMAPModel._fit hands to cho_factor, statistical_models.py:1147-1149), so it is first factored as
// H = L L^T with the 16 x 16 MFMA tile primitives of the fit loop (tile_chol.h), no pivot search at all, and rewritten
// as the unit-lower / upper pair the solves expect: H = (L D^-1)(D L^T), D = diag(L), identity permutation -- the LU
// factorisation WITHOUT pivoting, which is what partial pivoting would also choose for a diagonally dominant matrix and
// equally backward stable for a positive definite one.  A non-positive pivot leaves S.lu untouched and the pivoted LU
// runs as before.  Cp: the padded copy in packed tiles (build_hess_tiles); its trailing tiles are updated in place.
// Round 6: the copy is in PACKED tiles (build_hess_tiles), and a step has ONE barrier, as the fit loop's factorisation
// (fit_loop.hip: solve_posterior): wave 0 -- the chain -- updates, factors and inverts diagonal tile k + 1 while the other seven
// update the tiles right of column k + 1; a worker then takes its tiles of column k + 1, waits for the chain's flag and turns
// them into panel k + 1 straight from the registers -- two panels in LDS in turn -- and into the two triangles of the factors.
// Same products, same operands and the same order into every tile as the routine of rounds 3-5 (two barriers per step, panel
// from the mirror tiles in memory): the same bits (tests: the LogNormal fits land on the Newton counts they had).
// LDS: the solve vectors' space and the LU panel behind it (fh_ln_chol_doubles): two panels, two L_kk^-1, diag(L) and its
// reciprocals, flags, the tile table.
// ---- the trailing tiles on the cluster's other units (round 6) -------------------------------------------------------------------
// One compute unit moves ~25 B per cycle to and from the L2, and the right-looking update streams every trailing tile in and out
// at every step: 4.6 MB = 184 k of a factorisation's 366 k cycles at N = 300 (43 MB at N = 640).  With a cluster, the tiles (I, J),
// J >= 2, live in the REGISTERS of the helpers' waves instead (tile e of the column-major list on wave e mod W: three tiles per wave
// at N = 300 with seven helpers): the first workgroup parks every panel tile L_Ik^T it forms in the dead upper tile (k, I) of the
// copy and raises a progress word; a helper wave takes the two panel tiles of each of its tiles from the L2 (device-scope loads),
// updates in registers, and hands a tile back -- store + a counter per block column -- after panel J - 3; the first workgroup keeps
// a BAND of two block columns: at step k it applies panel k to column k + 1 (and factors it, as before) and to column k + 2, which
// it took back from the helpers at this step and keeps in the copy until the next -- so a hand-over has two steps' time and is
// off the chain's path (with a band of one the chain waited ~5 k cycles per step for its column: 286 k cycles per factorisation;
// the hand-over is six L2 round trips).
// Same products, same operands, same order into every tile: the same bits.  Hand-overs as in the fit loop's cluster form
// (fit_loop.hip, clu::): the writer waits for its stores (s_waitcnt vmcnt(0)) before an agent-scope atomic, the reader loads with
// sc1 (served by the XCD's L2, never by its own L1) -- the cluster sits on one XCD (ctl[6]); no invalidate.  Progress word and
// column counters are cumulative over the factorisations of a launch (both sides count them: `epoch`), every wait is bounded.
constexpr int kCholHelperTiles = 16;  // tiles a helper wave holds at most (128 of its registers)
__device__ __forceinline__ int ctl_load(const int *p);
__device__ __forceinline__ void ctl_store(int *p, int v);
__device__ __forceinline__ v4f64 ld_pk_dev(const tilechol::gdouble *base, unsigned tile_byte_off, int lane) {
    v2f64 lo, hi;
    const unsigned a = tile_byte_off + (unsigned)lane * 16u;
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %2, %3 sc1\n\tglobal_load_dwordx4 %1, %2, %3 offset:1024 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(lo), "=&v"(hi)
                 : "v"(a), "s"(base)
                 : "memory");
    return v4f64{lo[0], lo[1], hi[0], hi[1]};
}
// two tiles with one wait
__device__ __forceinline__ void ld_pk_dev2(const tilechol::gdouble *base, unsigned oa, unsigned ob, int lane, v4f64 &ta, v4f64 &tb) {
    v2f64 alo, ahi, blo, bhi;
    const unsigned a = oa + (unsigned)lane * 16u, b = ob + (unsigned)lane * 16u;
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %4, %6 sc1\n\tglobal_load_dwordx4 %1, %4, %6 offset:1024 sc1\n\t"
                 "global_load_dwordx4 %2, %5, %6 sc1\n\tglobal_load_dwordx4 %3, %5, %6 offset:1024 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(alo), "=&v"(ahi), "=&v"(blo), "=&v"(bhi)
                 : "v"(a), "v"(b), "s"(base)
                 : "memory");
    ta = v4f64{alo[0], alo[1], ahi[0], ahi[1]};
    tb = v4f64{blo[0], blo[1], bhi[0], bhi[1]};
}
// four tiles with one wait (a helper wave's two panel tiles for two of its tiles)
__device__ __forceinline__ void ld_pk_dev4(const tilechol::gdouble *base, unsigned o0, unsigned o1, unsigned o2, unsigned o3, int lane,
                                           v4f64 &t0, v4f64 &t1, v4f64 &t2, v4f64 &t3) {
    v2f64 l0, h0, l1, h1, l2, h2, l3, h3;
    const unsigned ln = (unsigned)lane * 16u, a0 = o0 + ln, a1 = o1 + ln, a2 = o2 + ln, a3 = o3 + ln;
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %8, %12 sc1\n\tglobal_load_dwordx4 %1, %8, %12 offset:1024 sc1\n\t"
                 "global_load_dwordx4 %2, %9, %12 sc1\n\tglobal_load_dwordx4 %3, %9, %12 offset:1024 sc1\n\t"
                 "global_load_dwordx4 %4, %10, %12 sc1\n\tglobal_load_dwordx4 %5, %10, %12 offset:1024 sc1\n\t"
                 "global_load_dwordx4 %6, %11, %12 sc1\n\tglobal_load_dwordx4 %7, %11, %12 offset:1024 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(l0), "=&v"(h0), "=&v"(l1), "=&v"(h1), "=&v"(l2), "=&v"(h2), "=&v"(l3), "=&v"(h3)
                 : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "s"(base)
                 : "memory");
    t0 = v4f64{l0[0], l0[1], h0[0], h0[1]};
    t1 = v4f64{l1[0], l1[1], h1[0], h1[1]};
    t2 = v4f64{l2[0], l2[1], h2[0], h2[1]};
    t3 = v4f64{l3[0], l3[1], h3[0], h3[1]};
}
__device__ __forceinline__ bool chol_dist_ok(const LogNormalParams &P, int cluster, bool same_xcd) {
    const int nb = P.NP / 16;
    return P.dist_cholesky && cluster > 1 && same_xcd && nb >= 5 && (nb - 3) * (nb - 2) / 2 <= kCholHelperTiles * (cluster - 1) * LNW;
}
// a helper workgroup's part of one factorisation (command LN_CMD_CHOL); epoch: distributed factorisations so far, this one included
__device__ __forceinline__ void chol_helper(const LogNormalParams &P, int member, int epoch) {
    using namespace tilechol;
    const int nb = P.NP / 16, tid = ln_tid(), lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = P.cluster - 1, W = H * LNW, w = wave * H + (member - 1);  // (consecutive tiles of a column on different units)
    const int nT = (nb - 3) * (nb - 2) / 2;  // the tiles (I, J), 3 <= J <= I
    gdouble *Cg = as_global(uniform_ptr(P.LU + P.N * P.N));
    int *const prog = P.ctl + LN_CTL_PROG, *const col = P.ctl + LN_CTL_COL;
    v4f64 T[kCholHelperTiles];
    int ij[kCholHelperTiles];  // I << 8 | J, -1: no tile (any more)
    {
        int J = 3, cum = 0;  // cum: tiles of the columns before J
#pragma unroll
        for (int s = 0; s < kCholHelperTiles; ++s) {
            const int e = w + s * W;
            ij[s] = -1;
            T[s] = v4f64{0.0, 0.0, 0.0, 0.0};
            if (e < nT) {
                while (e - cum >= nb - J) {
                    cum += nb - J;
                    ++J;
                }
                const int I = J + (e - cum);
                ij[s] = (I << 8) | J;
                T[s] = ld_pk(Cg, (unsigned)((I * nb + J) * 2048), lane);
            }
        }
    }
    for (int p = 0; p + 3 < nb; ++p) {
        {   // panel p is in the L2 when the progress word says so
            const long long t0 = wall_clock64();
            int v;
            while ((v = ctl_load(prog)) < epoch * 64 + p + 1) {
                __builtin_amdgcn_s_sleep(1);
                if (wall_clock64() - t0 > 5000000ll) return;  // (50 ms: the first workgroup has given up or gone; nothing hangs)
            }
            if ((v & 63) == 63 && (v >> 6) == epoch) return;  // abandoned (a pivot that is not positive)
        }
        bool any = false;
        auto apply = [&](int s, const v4f64 &dI, const v4f64 &dJ) __attribute__((always_inline)) {
            const int I = ij[s] >> 8, J = ij[s] & 255;
#pragma unroll
            for (int q = 0; q < 4; ++q) T[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(-dI[q], dJ[q], T[s], 0, 0, 0);
            if (J == p + 3) {  // the column the first workgroup takes into its band next: back to the copy, counted
                st_pk(Cg, (unsigned)((I * nb + J) * 2048), lane, T[s]);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) __hip_atomic_fetch_add(col + J, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ij[s] = -1;
            }
        };
#pragma unroll
        for (int s = 0; s < kCholHelperTiles; s += 2) {  // two tiles per wait: their four panel tiles in flight together
            const bool l0 = ij[s] >= 0, l1 = ij[s + 1] >= 0;
            if (l0 || l1) {
                any = true;
                const int e0 = l0 ? ij[s] : ij[s + 1], e1 = l1 ? ij[s + 1] : ij[s];
                v4f64 dI0, dJ0, dI1, dJ1;
                ld_pk_dev4(Cg, (unsigned)((p * nb + (e0 >> 8)) * 2048), (unsigned)((p * nb + (e0 & 255)) * 2048),
                           (unsigned)((p * nb + (e1 >> 8)) * 2048), (unsigned)((p * nb + (e1 & 255)) * 2048), lane, dI0, dJ0, dI1, dJ1);
                if (l0) apply(s, dI0, dJ0);
                if (l1) apply(s + 1, dI1, dJ1);
            }
        }
        if (!any) break;
    }
}

constexpr int kCholMaxTiles = 171;  // tiles right of column 1 at nb = 20
__host__ __device__ constexpr int fh_ln_chol_doubles(int NP) {
    return 2 * NP * tilechol::PS + 2 * 16 * tilechol::PS + 2 * NP + 4 + (kCholMaxTiles + 1) / 2;
}
// WIDE (320 < N <= 640): ONE panel -- two of 640 x 17 doubles do not fit --; the column tiles wait in the dead upper tiles of the
// copy until everybody has read panel k (a second barrier per step), the tile table has 741 entries
constexpr int kCholMaxTilesWide = 741;  // nb = 40
__host__ __device__ constexpr int fh_ln_chol_small_wide(int NP) {  // L_kk^-1 (two), diag(L) and its reciprocals, flags, the tile table
    return 2 * 16 * tilechol::PS + 2 * NP + 4 + (kCholMaxTilesWide + 1) / 2 + 1;
}
template <bool WIDE>
__device__ __forceinline__ double *chol_dvec(const LnS &S, int NP) { return S.chol + (WIDE ? 0 : 2 * NP * tilechol::PS) + 2 * 16 * tilechol::PS; }
template <bool WIDE>
__device__ __forceinline__ bool cholesky_as_lu(const LogNormalParams &P, LnS &S, double *Cp, double *Xd = nullptr) {
    using namespace tilechol;
    const int N = P.N, NP = P.NP, nb = NP / 16;
    const int tid = ln_tid(), lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (scalar: loop control on the SALU)
    const int cl = lane & 15, rg = lane >> 4;
    double *pan0 = WIDE ? S.cpan : S.chol, *dli0 = WIDE ? S.chol : pan0 + 2 * NP * PS;
    double *dvec = chol_dvec<WIDE>(S, NP), *rdv = dvec + NP;  // diag(L), 1 / diag(L)
    int *flag = reinterpret_cast<int *>(rdv + NP);      // [0] a pivot that is not positive, [1] last diagonal tile whose inverse is in LDS
    int *tab = flag + 8;                                // tile e = (i - 1) i / 2 + (j - 1), 1 <= j <= i, relative to block (k+1, k+1): i << 8 | j
    auto pan_of = [&](int k) { return pan0 + (WIDE ? (size_t)0 : (size_t)(k & 1) * NP * PS); };
    auto dli_of = [&](int k) { return dli0 + (k & 1) * 16 * PS; };
    gdouble *Cg = as_global(uniform_ptr(Cp));
    gdouble *lu = as_global(S.lu);       // the factors go straight to their final place: column-major N x N, unit-lower L D^-1
                                         // below the diagonal, D L^T on and above it (both scalings need the diagonal of block
                                         // column k only, known when its tile is factored)
    LTIC();
    // the trailing tiles on the helpers (chol_helper): decided per factorisation (a cluster may have been disbanded since the last)
    const bool dist = chol_dist_ok(P, S.cluster, S.same_xcd);
    const int epoch = dist ? ++S.chol_epoch : 0;
    int *const prog = P.ctl + LN_CTL_PROG;
    if (dist) cluster_dispatch(P, LN_CMD_CHOL, S.same_xcd, S.seq);  // (the copy is complete: its builders were waited for)
    if constexpr (WIDE) {  // the panel takes the place of the vectors: 18 N doubles out (and back in at the end), 2 x 92 KB against
                           // the factorisation's 43 MB of tiles
        for (int i = tid; i < 18 * N; i += LT) S.bak[i] = S.cpan[i];
        __syncthreads();
    }
    auto leave = [&](bool ok) {
        if (dist) {
            if (!ok) {  // the helpers stop waiting for panels
                __syncthreads();
                if (tid == 0) ctl_store(prog, epoch * 64 + 63);
            }
            if (!cluster_wait(P, S.s_cl, S.same_xcd)) {
                if (tid == 0) __hip_atomic_store(&P.ctl[4], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // disbanded
                S.cluster = 1;
            }
        }
        if constexpr (WIDE) {
            __syncthreads();
            for (int i = tid; i < 18 * N; i += LT) S.cpan[i] = S.bak[i];
            __syncthreads();
        }
        return ok;
    };
    if (tid == 0) flag[0] = flag[1] = 0;
    for (int e = tid; e < (nb - 1) * (nb - 2) / 2; e += LT) {
        int i = (int)((sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
        while ((i + 1) * (i + 2) / 2 <= e) ++i;
        while (i * (i + 1) / 2 > e) --i;
        tab[e] = ((i + 1) << 8) | (e - i * (i + 1) / 2 + 1);
    }
    // factor + invert diagonal tile k (held in `t`, accumulator layout): what the other waves wait for goes to LDS first
    // (diag_factor: L_kk^-1, diag(L), its reciprocals; the chain raises its flag behind it), its part of the factors to memory
    // afterwards (diag_outputs)
    auto diag_factor = [&](int k, v4f64 &t, v4f64 &x) {
        double *dli = dli_of(k);
        const bool ok = chol_inv_tile_acc(t, x, lane, -1);
        if (!ok && lane == 0) flag[0] = 1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            dli[(rg + 4 * r) * PS + cl] = x[r];
            if (rg + 4 * r == cl) {
                dvec[16 * k + cl] = t[r];
                rdv[16 * k + cl] = 1.0 / t[r];
            }
        }
    };
    auto diag_outputs = [&](int k, const v4f64 &t, const v4f64 &x) {
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        const double dc = dvec[16 * k + cl], rdc = rdv[16 * k + cl];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (Xd) Xd[k * 256 + (rg + 4 * r) * 16 + cl] = x[r];  // L_kk^-1, row-major (the Tr2 solve's A operand)
            const int i = 16 * k + rg + 4 * r, j = 16 * k + cl;  // element (i, j) of L, i >= j holds data
            if (i < N && j < N && i >= j) {
                if (i > j) {
                    lu[(size_t)j * N + i] = t[r] * rdc;  // L_ij / L_jj
                    lu[(size_t)i * N + j] = dc * t[r];   // U_ji = L_jj L_ij
                } else {
                    lu[(size_t)i * N + i] = dc * dc;
                    S.rdiag[i] = rdc * rdc;
                    S.perm[i] = i;
                }
            }
        }
    };
    // a panel tile: D = L_kk^-1 T^T = L_Ik^T from T^T (accumulator layout) -> LDS panel `pan` (unscaled, for the trailing updates)
    // and, scaled, into the factors.  Element (a, b) of D is L[16 I + b][16 k + a]: its store runs along a column of the
    // unit-lower part; the upper part wants the transposed tile, which is the same product with the operands exchanged (the
    // accumulator registers of T^T are the A fragments of T, the A fragments of X the B fragments of X^T).
    // (pan == nullptr, WIDE: the one panel is still being read -- D waits in the dead upper tile (k, I) of the copy and goes to
    //  the panel behind the step's first barrier, panel_from_park)
    auto panel_tile = [&](int k, int I, const Frag &fx, const Frag &ft, double *pan) {
        const v4f64 z = {0.0, 0.0, 0.0, 0.0};
        const v4f64 d = mfma4(fx, ft, z, false);
        const v4f64 dt = mfma4(ft, fx, z, false);
        const double dck = dvec[16 * k + cl];
        if (!pan || dist) st_pk(Cg, (unsigned)((k * nb + I) * 2048), lane, d);  // (dist: where the helpers take the panel from)
        double *pr = (pan ? pan : pan0) + (size_t)((I - k - 1) * 16 + cl) * PS + rg;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (pan) pr[4 * r] = d[r];
            const int a = 16 * k + rg + 4 * r, b = 16 * I + cl;
            if (a < N && b < N) lu[(size_t)a * N + b] = d[r] * rdv[a];  // (L D^-1)[b][a]
            const int bt = 16 * I + rg + 4 * r, at = 16 * k + cl;
            if (at < N && bt < N) lu[(size_t)bt * N + at] = dck * dt[r];  // (D L^T)[at][bt]
        }
    };
    if (wave == 0) {
        v4f64 t = ld_pk(Cg, 0u, lane), x;
        diag_factor(0, t, x);
        diag_outputs(0, t, x);
    }
    __syncthreads();
    if (flag[0]) return leave(false);
    {   // panel 0 from memory: the tiles (0, I) of the first block row ARE (C_I0)^T
        Frag fx;
#pragma unroll
        for (int q = 0; q < 4; ++q) fx.v[q] = dli_of(0)[cl * PS + 4 * q + rg];
        if constexpr (WIDE) {
            for (int I = 1 + wave; I < nb; I += LNW) {  // (once per factorisation: tile by tile)
                const v4f64 t = ld_pk(Cg, (unsigned)(I * 2048), lane);
                Frag fb1;
#pragma unroll
                for (int q = 0; q < 4; ++q) fb1.v[q] = t[q];
                panel_tile(0, I, fx, fb1, pan_of(0));
            }
        } else {
        constexpr int kPanelMax = 3;  // ceil((NP / 16 - 1) / LNW) for NP <= 400
        Frag fb[kPanelMax];
#pragma unroll
        for (int u = 0; u < kPanelMax; ++u) {
            const int I = 1 + wave + u * LNW;
            if (I < nb) {
                const v4f64 t = ld_pk(Cg, (unsigned)(I * 2048), lane);
#pragma unroll
                for (int q = 0; q < 4; ++q) fb[u].v[q] = t[q];
            }
        }
#pragma unroll
        for (int u = 0; u < kPanelMax; ++u) {
            const int I = 1 + wave + u * LNW;
            if (I < nb) panel_tile(0, I, fx, fb[u], pan_of(0));
        }
        }
    }
    auto publish = [&](int panels) {  // every wave's stores have left the unit before the word is raised behind the barrier
        if (tid == 0) ctl_store(prog, epoch * 64 + panels);
    };
    if (dist) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (dist) publish(1);
    // block column J >= 3 comes back from the helpers (updated through panel J - 3): its counter, then device-scope loads
    auto column_in = [&](int J) {
        int *const cw = P.ctl + LN_CTL_COL + J;
        const int want = epoch * (nb - J);
        const long long t0 = wall_clock64();
        while (ctl_load(cw) < want) {
            __builtin_amdgcn_s_sleep(1);
            if (wall_clock64() - t0 > 2000000ll) {  // (20 ms: the factorisation fails, the fit goes on without the distributed form)
                if (lane == 0) flag[0] = 1;
                break;
            }
        }
    };
    // distributed, N <= 320: a worker has at most six tiles per step, and the chain is what the step waits for -- wave 4, which shares
    // the chain's SIMD (a vector instruction issued while the mate's matrix instructions run waits for them), sits the steps out
    const bool idle_mate = dist && !WIDE;
    double ident[4];  // B fragments of the 16 x 16 identity: four matrix instructions against it transpose a tile
#pragma unroll
    for (int q = 0; q < 4; ++q) ident[q] = (4 * q + rg == cl) ? 1.0 : 0.0;
    for (int k = 0; k + 1 < nb; ++k) {
        const int m = nb - k - 1;
        const int cntA = __builtin_amdgcn_readfirstlane(m * (m - 1) / 2);  // tiles with k + 1 < J <= I
        const int ncol = m - 1;                                             // tiles (I, k + 1), I > k + 1
        const double *pan_cur = pan_of(k);
        const unsigned base = (unsigned)((k + 1) * (nb + 1) * 2048);        // tile (k+1, k+1)
        auto update = [&](int i, int j, v4f64 a) {  // a -= L_{k+1+i,k} L_{k+1+j,k}^T
            const double *pa1 = pan_cur + (size_t)(i * 16 + cl) * PS + rg;
            const double *pb1 = pan_cur + (size_t)(j * 16 + cl) * PS + rg;
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) a = __builtin_amdgcn_mfma_f64_16x16x4f64(-pa1[4 * s2], pb1[4 * s2], a, 0, 0, 0);
            return a;
        };
#ifdef LN_TIMING
        long long _tc = clock64();
#endif
        if (wave == 0) {
            // the chain: diagonal tile k + 1 updated, factored and inverted while the other waves update the rest
            // (distributed, two panels: the diagonal tile was left in LDS by the worker that gave it panel k - 1 -- the spare tile
            //  behind the nb - 1 tiles of this step's panel buffer -- instead of going through the L2: ~2 k cycles of the chain's path)
            v4f64 t0;
            if (!WIDE && dist && k >= 1) {
                const v2f64 *dsl = reinterpret_cast<const v2f64 *>(pan_of(k) + (size_t)(nb - 1) * 16 * PS);
                const v2f64 lo = dsl[2 * lane], hi = dsl[2 * lane + 1];
                t0 = v4f64{lo[0], lo[1], hi[0], hi[1]};
            } else {
                t0 = ld_pk(Cg, base, lane);
            }
            v4f64 t = update(0, 0, t0), x;
            diag_factor(k + 1, t, x);
            __hip_atomic_store(&flag[1], k + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            CHT(0);
            diag_outputs(k + 1, t, x);
            CHT(1);
        } else if (!(idle_mate && wave == 4)) {
            const int NWK = idle_mate ? LNW - 2 : LNW - 1;
            const int widx = idle_mate && wave > 4 ? wave - 2 : wave - 1;
            if (dist && k + 2 < nb) {
                // ---- the band's second column: tiles (I, k + 2), I >= k + 2, take panel k here and wait in the copy for the next step
                //      (from step 1 on they come back from the helpers, updated through panel k - 1: counter, device-scope loads)
                const bool back = k >= 1;
                const int n2 = nb - k - 2;
                if (back && widx < n2) column_in(k + 2);
                auto band_out = [&](int c2, unsigned co, const v4f64 &u2) {
                    if (!WIDE && c2 == 0) {  // the next step's diagonal tile: to the chain through LDS (the other panel buffer's spare tile)
                        v2f64 *dsl = reinterpret_cast<v2f64 *>(pan_of(k + 1) + (size_t)(nb - 1) * 16 * PS);
                        dsl[2 * lane] = v2f64{u2[0], u2[1]};
                        dsl[2 * lane + 1] = v2f64{u2[2], u2[3]};
                    } else {
                        st_pk(Cg, co, lane, u2);
                    }
                };
                auto off2 = [&](int c2) { return (unsigned)(((k + 2 + c2) * nb + (k + 2)) * 2048); };
                int c2 = widx;
                for (; c2 + NWK < n2; c2 += 2 * NWK) {  // two tiles per wait
                    v4f64 ta, tb;
                    if (back) {
                        ld_pk_dev2(Cg, off2(c2), off2(c2 + NWK), lane, ta, tb);
                    } else {
                        ta = ld_pk(Cg, off2(c2), lane);
                        tb = ld_pk(Cg, off2(c2 + NWK), lane);
                    }
                    band_out(c2, off2(c2), update(c2 + 1, 1, ta));
                    band_out(c2 + NWK, off2(c2 + NWK), update(c2 + NWK + 1, 1, tb));
                }
                if (c2 < n2) band_out(c2, off2(c2), update(c2 + 1, 1, back ? ld_pk_dev(Cg, off2(c2), lane) : ld_pk(Cg, off2(c2), lane)));
            }
            // ---- the tiles right of column k + 1: every NWK-th tile of the table, two in flight in two named register sets ----
            auto off_of = [&](int t) { return base + (unsigned)(((t >> 8) * nb + (t & 255)) * 2048); };
            auto finish = [&](int t, v4f64 a) { st_pk(Cg, off_of(t), lane, update(t >> 8, t & 255, a)); };
            int e = widx;
            if (!dist && e < cntA) {
                int ta = tab[e], tb = 0;
                v4f64 a = ld_pk(Cg, off_of(ta), lane), b = a;
                bool hb = e + NWK < cntA;
                if (hb) {
                    tb = tab[e + NWK];
                    b = ld_pk(Cg, off_of(tb), lane);
                }
                e += 2 * NWK;
                for (;;) {
                    const int tc = ta;
                    const v4f64 c = a;
                    const bool ha = e < cntA;
                    if (ha) {
                        ta = tab[e];
                        a = ld_pk(Cg, off_of(ta), lane);
                    }
                    finish(tc, c);
                    if (!hb) break;
                    const int td = tb;
                    const v4f64 d = b;
                    hb = e + NWK < cntA;
                    if (hb) {
                        tb = tab[e + NWK];
                        b = ld_pk(Cg, off_of(tb), lane);
                    }
                    finish(td, d);
                    if (!ha) break;
                    e += 2 * NWK;
                }
            }
#ifdef LN_TIMING
            if (wave == 1) CHT(3);
#endif
            // ---- column k + 1 (the round-robin deal goes on where the table stopped): update, then panel k + 1 from the registers ----
            int cfirst = widx - cntA % NWK;
            if (cfirst < 0) cfirst += NWK;
            if constexpr (WIDE) {
                // up to six tiles per wave: one at a time; D is parked in the copy's dead upper tile until the panel is free
                bool waited = false;
                Frag fx;
                for (int c = cfirst; c < ncol; c += NWK) {
                    const v4f64 t = update(c + 1, 0, ld_pk(Cg, base + (unsigned)((c + 1) * nb * 2048), lane));
                    v4f64 tt = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int q = 0; q < 4; ++q) tt = __builtin_amdgcn_mfma_f64_16x16x4f64(t[q], ident[q], tt, 0, 0, 0);
                    Frag ft1;
#pragma unroll
                    for (int q = 0; q < 4; ++q) ft1.v[q] = tt[q];
                    if (!waited) {
                        int spins = 0;
                        while (__hip_atomic_load(&flag[1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < k + 1) {
                            __builtin_amdgcn_s_sleep(1);
                            if (++spins > (1 << 22)) {
                                if (lane == 0) flag[0] = 1;
                                break;
                            }
                        }
#pragma unroll
                        for (int q = 0; q < 4; ++q) fx.v[q] = dli_of(k + 1)[cl * PS + 4 * q + rg];
                        waited = true;
                    }
                    panel_tile(k + 1, k + 2 + c, fx, ft1, nullptr);
                }
            } else {
            constexpr int kColMax = 3;  // ceil(18 / 7)
            v4f64 tc[kColMax];
#pragma unroll
            for (int u = 0; u < kColMax; ++u) {
                const int c = cfirst + u * NWK;
                if (c < ncol) tc[u] = ld_pk(Cg, base + (unsigned)((c + 1) * nb * 2048), lane);
            }
            Frag ft[kColMax];
#pragma unroll
            for (int u = 0; u < kColMax; ++u) {
                const int c = cfirst + u * NWK;
                if (c < ncol) {
                    const v4f64 t = update(c + 1, 0, tc[u]);  // T = C_{I,k+1} - L_Ik L_{k+1,k}^T
                    v4f64 tt = {0.0, 0.0, 0.0, 0.0};          // T^T: what the mirror tile held in rounds 3-5
#pragma unroll
                    for (int q = 0; q < 4; ++q) tt = __builtin_amdgcn_mfma_f64_16x16x4f64(t[q], ident[q], tt, 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < 4; ++q) ft[u].v[q] = tt[q];
                }
            }
#ifdef LN_TIMING
            if (wave == 1) CHT(4);
#endif
            if (cfirst < ncol) {
                int spins = 0;  // L_{k+1,k+1}^-1 comes from the chain
                while (__hip_atomic_load(&flag[1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < k + 1) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1 << 22)) {  // (a stuck flag must not hang the device)
                        if (lane == 0) flag[0] = 1;
                        break;
                    }
                }
#ifdef LN_TIMING
                if (wave == 1) CHT(5);
#endif
                Frag fx;
#pragma unroll
                for (int q = 0; q < 4; ++q) fx.v[q] = dli_of(k + 1)[cl * PS + 4 * q + rg];
#pragma unroll
                for (int u = 0; u < kColMax; ++u) {
                    const int c = cfirst + u * NWK;
                    if (c < ncol) panel_tile(k + 1, k + 2 + c, fx, ft[u], pan_of(k + 1));
                }
#ifdef LN_TIMING
                if (wave == 1) CHT(6);
#endif
            }
            }
        }
        if (dist) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the parked tiles of panel k + 1)
        if constexpr (WIDE) {
            __syncthreads();  // everybody has read panel k: the parked tiles of column k + 1 become panel k + 1
            if (dist) publish(k + 2);
            if (wave > 0) {
                int cfirst = (wave - 1) - cntA % (LNW - 1);
                if (cfirst < 0) cfirst += LNW - 1;
                for (int c = cfirst; c < ncol; c += LNW - 1) {
                    const v4f64 d = ld_pk(Cg, (unsigned)(((k + 1) * nb + k + 2 + c) * 2048), lane);
                    double *pr = pan0 + (size_t)(c * 16 + cl) * PS + rg;
#pragma unroll
                    for (int r = 0; r < 4; ++r) pr[4 * r] = d[r];
                }
            }
        }
        __syncthreads();
        if (!WIDE && dist) publish(k + 2);
#ifdef LN_TIMING
        if (wave == 0) CHT(2);
        if (wave == 1) CHT(7);
#endif
        if (flag[0]) return leave(false);
    }
    LTOC(1);
    return leave(true);
}

// Partial-pivoting LU in place (column-major, unit lower); perm[i] = source row of row i, rdiag[i] = 1 / U_ii.
// Two barriers per column: every wave finds the pivot of column k for itself (same data, same reduction, same
// answer -- no broadcast, no barrier), then
//   phase A: rows k <-> piv are exchanged OUTSIDE column k, the new row k and the scaled column k are staged in LDS
//            vectors (column k itself is only read, so a wave still searching it sees the old values);
//   phase B: the scaled column is written back and the trailing block gets its rank-1 update.
// A zero pivot leaves the column unscaled (LAPACK getf2 does the same and reports it).
__device__ __forceinline__ void lu_factor(LnS &S, int N, double *A) {
    const int tid = ln_tid(), lane = tid & 63;
    LTIC();
    for (int i = tid; i < N; i += LT) S.perm[i] = i;
    __syncthreads();
    for (int k = 0; k < N; ++k) {
        const double *ck = A + k * N;
        double best = -1.0;
        int bi = 0x7fffffff;
        for (int i = k + lane; i < N; i += 64) {  // first maximum of |A[k:, k]|
            const double v = fabs(ck[i]);
            if (v > best) {
                best = v;
                bi = i;
            }
        }
        const double top = wave_reduce(best, FMax());  // (best is never NaN: a NaN entry fails `v > best`)
        // the row of the maximum: one lane in all but exceptional cases (ties go to the first row)
        const unsigned long long tie = __ballot(best == top);
        int piv;
        if ((tie & (tie - 1)) == 0) piv = __builtin_amdgcn_readlane(bi, __ffsll((long long)tie) - 1);
        else piv = wave_reduce(best == top ? bi : 0x7fffffff, rocprim::minimum<int>());
        if (piv >= N) piv = k;  // all-NaN column: keep the diagonal
        const double pv = ck[piv], dkk = ck[k];
        // phase A
        for (int j = tid; j < N; j += LT) {
            if (j == k) continue;
            const double a = A[j * N + k], b = A[j * N + piv];
            if (piv != k) {
                A[j * N + k] = b;
                A[j * N + piv] = a;
            }
            if (j > k) S.rowk[j] = b;
        }
        for (int i = k + 1 + tid; i < N; i += LT) {
            const double v = (i == piv) ? dkk : ck[i];
            S.col[i] = (pv != 0.0) ? v / pv : v;
        }
        if (tid == LT - 1) {
            S.rdiag[k] = 1.0 / pv;
            if (piv != k) {
                const int t = S.perm[k];
                S.perm[k] = S.perm[piv];
                S.perm[piv] = t;
            }
        }
        __syncthreads();
        // phase B
        for (int i = k + tid; i < N; i += LT) A[k * N + i] = (i == k) ? pv : S.col[i];
        if (pv != 0.0)
            for (int j = k + 1 + (tid >> 5); j < N; j += LT / 32) {
                const double uj = S.rowk[j];
                double *cj = A + j * N;
                for (int i = k + 1 + (tid & 31); i < N; i += 32) cj[i] = fma(-S.col[i], uj, cj[i]);
            }
        __syncthreads();
    }
    LTOC(1);
}

// Blocked right-looking LU with partial pivoting for factors that live in global memory (N > 112), panels of 32
// columns.  The unblocked sweep above rewrites the whole trailing matrix once per COLUMN (144 MB per factorisation at
// N = 300); here it is rewritten once per PANEL, on the matrix pipe.
//   panel   one thread per row keeps its 32 panel entries in REGISTERS for all 32 column steps.  Rows never move: a
//           thread tracks the position its row would have after LAPACK's swaps (`lpos`); the pivot of a step is the
//           largest |a_j| among the rows not yet chosen, first position on ties.  One barrier per column: every wave
//           finds its best row (DPP max + ballot) and publishes that row before the barrier, afterwards every thread
//           picks the winner among the wave maxima and eliminates with 31 register fmas whose destination is the next
//           slot (the frame shifts one column per step, so the step loop is rolled with static register indices).
//           Measured per column step at N = 300: 7 k cycles (the LDS-panel version with two barriers: 6.5 k; a fully
//           unrolled 32-step version: 7.5 k) -- the step is a chain of ~15 dependent LDS round trips and DPP stages.
//   swaps   the rows that changed position (<= 64) form a (source, destination) list; a wave moves one outside column
//           with a single gather / scatter pair, several columns in flight.
//   U12     = L11^-1 A12 per block of 16 columns on MFMAs: the two 16 x 16 diagonal blocks of L11 are inverted by 32
//           threads in registers, and U_top = A^-1 T0, U_bot = C^-1 (T1 - B U_top) chains through the accumulator
//           layout (the C/D layout of a 16 x 16 tile is the B-operand layout of its four k-steps).
//   trailing  A22 -= L21 U12, 16 x 16 tiles shared evenly among the waves, L21 fragments from the LDS panel, U12 tiles
//           loaded once per block column, the next tile's load in flight during the 8 MFMAs of this one.
// Same pivots and the same elimination arithmetic as the unblocked algorithm inside a panel; U12 goes through the
// explicit block inverses (differences at round-off level).
constexpr int LU_NB = 32;
__device__ __forceinline__ void lu_factor_blocked(LnS &S, int N, double *A) {
    const int tid = ln_tid(), lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cl = lane & 15, rg = lane >> 4;
    double *pan = S.pan;
    double *linv = S.part;                               // two 16 x 16 inverse blocks, k-major: [blk][k][i]
    double *pred = S.part + 512;                         // 2 x LNW wave maxima
    int *pidx = reinterpret_cast<int *>(S.part + 528);   // 2 x LNW positions of the wave maxima
    int *mv = reinterpret_cast<int *>(S.part + 540);     // [0]: count, [1..64]: src | dst << 16 of the moved rows
    // candidate pivot rows, 2 (steps) x LNW (waves) x 32, in the solve vectors' space: 16-byte aligned for ds_read_b128
    v2f64 *rowv = reinterpret_cast<v2f64 *>(__builtin_assume_aligned(S.wsol, 16));
    LTIC();
    for (int i = tid; i < N; i += LT) S.perm[i] = i;
    __syncthreads();
    for (int k0 = 0; k0 < N; k0 += LU_NB) {
        const int nb = min(LU_NB, N - k0), m = N - k0;
#ifdef LN_TIMING
        long long _tp = clock64();
#endif
        // ---- panel ----------------------------------------------------------------------------------------------
        // a[c] is the row's entry in panel column j + c: the frame shifts by one column per step (the shift is the
        // destination of the elimination fma), so the step loop stays ROLLED with static register indices -- unrolled
        // it is ~60 KB of straight-line code and runs at instruction-fetch speed (7.5 k cycles per column, measured).
        const bool mine = tid < m;
        double a[LU_NB];
        {
            const double *src = A + (size_t)k0 * N + k0 + (mine ? tid : 0);
#pragma unroll
            for (int c = 0; c < LU_NB; ++c) a[c] = (c < nb) ? src[(size_t)c * N] : 0.0;
        }
        int lpos = tid;
        bool live = mine;  // not yet chosen as a pivot row
        if (tid == 0) mv[0] = 0;
        // The step is bound by the CU's LDS instruction rate (measured: 16-byte broadcast reads cost their full 8
        // cycles each), so waves that hold no rows of this panel only keep the barrier count (the last one also does the
        // bookkeeping), and only the part of a row that is still inside the panel is published / read.
        const bool wactive = wv * 64 < m;
        if (!wactive && lane == 0) {
            pred[wv] = pred[LNW + wv] = -1.0;
            pidx[wv] = pidx[LNW + wv] = 0x7fffffff;
        }
        // winner among the wave maxima: all partials in flight at once, branch-free selection
        auto combine = [&](int j, double &top, int &piv, int &pw) {
            v2f64 tv2[LNW / 2];
            v4i32 ti4[LNW / 4];
            const v2f64 *prv = reinterpret_cast<const v2f64 *>(__builtin_assume_aligned(pred + (j & 1) * LNW, 16));
            const v4i32 *piv4 = reinterpret_cast<const v4i32 *>(__builtin_assume_aligned(pidx + (j & 1) * LNW, 16));
#pragma unroll
            for (int w = 0; w < LNW / 2; ++w) tv2[w] = prv[w];
#pragma unroll
            for (int w = 0; w < LNW / 4; ++w) ti4[w] = piv4[w];
            top = -1.0;
            piv = 0x7fffffff;
            pw = -1;
#pragma unroll
            for (int w = 0; w < LNW; ++w) {
                const double tv = tv2[w >> 1][w & 1];
                const int ti = ti4[w >> 2][w & 3];
                const bool better = tv > top || (tv == top && ti < piv);
                top = better ? tv : top;
                piv = better ? ti : piv;
                pw = better ? w : pw;
            }
            if (pw < 0) {  // all-NaN column: keep the diagonal; nothing is eliminated (every candidate entry is NaN)
                piv = j;
                pw = 0;
            }
        };
        if (!wactive) {
            for (int j = 0; j < nb; ++j) {
                __syncthreads();
                if (wv == LNW - 1) {  // bookkeeping off the critical path (wave 7 holds no rows for N <= 448)
                    double top;
                    int piv, pw;
                    combine(j, top, piv, pw);
                    const double pv = reinterpret_cast<const double *>(rowv + (j & 1) * (LNW * LU_NB / 2) + pw * (LU_NB / 2))[0];
                    if (tid == LT - 1) {
                        S.rdiag[k0 + j] = 1.0 / pv;
                        if (piv != j) {
                            const int t = S.perm[k0 + j];
                            S.perm[k0 + j] = S.perm[k0 + piv];
                            S.perm[k0 + piv] = t;
                        }
                    }
                }
            }
        } else {
            for (int j = 0; j < nb; ++j) {
                // ONE barrier per column: every wave's best row is published before the barrier (candidate rows double
                // buffered), afterwards every thread picks the winner among the wave maxima and eliminates with its row.
                const int rem = nb - j;  // columns of the frame still inside the panel
                double v = fabs(a[0]);
                const bool ok = live && v == v;  // a NaN never wins
                if (!ok) v = -1.0;
                const double wtop = wave_reduce(v, FMax());
                // position of the maximum: one lane in all but exceptional cases (ties go to the first position)
                const unsigned long long tie = __ballot(ok && v == wtop);
                int wpos = 0x7fffffff;
                if (tie != 0) {
                    if ((tie & (tie - 1)) == 0) wpos = __builtin_amdgcn_readlane(lpos, __ffsll((long long)tie) - 1);
                    else wpos = wave_reduce((ok && v == wtop) ? lpos : 0x7fffffff, rocprim::minimum<int>());
                }
                v2f64 *rows = rowv + (j & 1) * (LNW * LU_NB / 2);
                if (ok && lpos == wpos) {  // this wave's candidate (at most one lane)
                    pred[(j & 1) * LNW + wv] = wtop;
                    pidx[(j & 1) * LNW + wv] = wpos;
                    v2f64 *rw = rows + wv * (LU_NB / 2);
#pragma unroll
                    for (int c = 0; c < LU_NB; c += 2)
                        if (c < rem) rw[c >> 1] = v2f64{a[c], a[c + 1]};
                } else if (lane == 0 && tie == 0) {
                    pred[(j & 1) * LNW + wv] = -1.0;
                    pidx[(j & 1) * LNW + wv] = wpos;
                }
                __syncthreads();
                double top;
                int piv, pw;
                combine(j, top, piv, pw);
                const bool nopiv = top < 0.0;
                const v2f64 *rw = rows + pw * (LU_NB / 2);
                v2f64 r2 = rw[0];
                const double pv = nopiv ? a[0] : r2[0];
                if (mine) {
                    if (lpos == piv) {
                        lpos = j;
                        live = false;
                    } else if (lpos == j) {
                        lpos = piv;
                    }
                }
                if (tid == LT - 1) {  // (only when the last wave holds rows)
                    S.rdiag[k0 + j] = 1.0 / pv;
                    if (piv != j) {
                        const int t = S.perm[k0 + j];
                        S.perm[k0 + j] = S.perm[k0 + piv];
                        S.perm[k0 + piv] = t;
                    }
                }
                // the entry of column j: the multiplier for a live row, the U entry for a row chosen now or earlier (its
                // frame keeps shifting with l = 0)
                double l = 0.0;
                if (live && !nopiv) {
                    l = (pv != 0.0) ? a[0] / pv : a[0];
                    pan[j * m + tid] = l;
                    if (pv == 0.0) l = 0.0;  // zero pivot: column left unscaled, no elimination (getf2 does the same)
                } else if (mine) {
                    pan[j * m + tid] = a[0];
                }
                a[0] = fma(-l, r2[1], a[1]);
#pragma unroll
                for (int c = 2; c < LU_NB; c += 2) {  // (unconditional: the 15 reads stay in flight together; slots at
                                                      // and beyond rem hold stale values that never move back inside)
                    r2 = rw[c >> 1];
                    a[c - 1] = fma(-l, r2[0], a[c]);
                    a[c] = fma(-l, r2[1], a[c + 1 < LU_NB ? c + 1 : c]);
                }
            }
        }
        // (every row now sits in the LDS panel at its ORIGINAL position: pivot rows wrote themselves, the others their
        // multipliers step by step)  moved rows to their final positions, then the panel back to the factor
        const bool moved = mine && lpos != tid;
        if (moved) mv[1 + atomicAdd(&mv[0], 1)] = tid | (lpos << 16);
        __syncthreads();
        if (moved) {
#pragma unroll
            for (int c = 0; c < LU_NB; ++c) a[c] = pan[min(c, nb - 1) * m + tid];
        }
        __syncthreads();
        if (moved) {
#pragma unroll
            for (int c = 0; c < LU_NB; ++c)
                if (c < nb) pan[c * m + lpos] = a[c];
        }
        __syncthreads();
        if (mine) {
            double *dst = A + (size_t)k0 * N + k0 + tid;
#pragma unroll 8
            for (int c = 0; c < nb; ++c) dst[(size_t)c * N] = pan[c * m + tid];
        }
        __syncthreads();
#ifdef LN_TIMING
        if (tid == 0) { const long long n_ = clock64(); ln_cyc[5] += n_ - _tp; _tp = n_; }
#endif
        // ---- row moves in the columns outside the panel; wave 0 first inverts the diagonal blocks of L11 ----------
        const bool trailing = k0 + LU_NB < N;
        if (trailing && tid < 32) {
            const int blk = tid >> 4, jj = tid & 15;
            const double *L = pan + (16 * blk) * m + 16 * blk;  // L[i][k] = L[k * m + i]
            double x[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) x[i] = (i == jj) ? 1.0 : 0.0;
#pragma unroll
            for (int i = 1; i < 16; ++i)
#pragma unroll
                for (int k = 0; k < i; ++k) x[i] = fma(-L[k * m + i], x[k], x[i]);
#pragma unroll
            for (int i = 0; i < 16; ++i) linv[blk * 256 + jj * 16 + i] = x[i];
        }
        {
            const int cnt = mv[0];
            if (cnt > 0) {
                const int pr = lane < cnt ? mv[1 + lane] : 0;
                const int so = pr & 0xffff, dd = pr >> 16;
                const bool on = lane < cnt;
                const int nout = N - nb;  // outside columns, in order: 0 .. k0-1, k0+nb .. N-1
                for (int ci = wv; ci < nout; ci += 4 * LNW) {
                    double vv[4];
                    double *cp[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int c1 = min(ci + u * LNW, nout - 1);
                        cp[u] = A + (size_t)(c1 < k0 ? c1 : c1 + nb) * N + k0;
                        vv[u] = on ? cp[u][so] : 0.0;
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (on && ci + u * LNW < nout) cp[u][dd] = vv[u];
                }
            }
        }
        __syncthreads();
#ifdef LN_TIMING
        if (tid == 0) { const long long n_ = clock64(); ln_cyc[7] += n_ - _tp; _tp = n_; }  // (slot 6 counts the fallbacks)
#endif
        if (trailing) {
            const int c0 = k0 + LU_NB, mr = m - LU_NB;     // first trailing column; trailing rows (= trailing columns)
            const int nblk = (mr + 15) >> 4;                // block rows = block columns
            // U12: one wave per block of 16 columns
            for (int cb = wv; cb < nblk; cb += LNW) {
                const int cc = c0 + 16 * cb + cl;
                const bool cok = cc < N;
                double *cp = A + (size_t)min(cc, N - 1) * N + k0 + rg;
                v4f64 t0, t1;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    t0[q] = cp[4 * q];
                    t1[q] = cp[16 + 4 * q];
                }
                v4f64 u0 = {0.0, 0.0, 0.0, 0.0}, u1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4)
                    u0 = __builtin_amdgcn_mfma_f64_16x16x4f64(linv[(4 * s4 + rg) * 16 + cl], t0[s4], u0, 0, 0, 0);
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4)
                    t1 = __builtin_amdgcn_mfma_f64_16x16x4f64(-pan[(4 * s4 + rg) * m + 16 + cl], u0[s4], t1, 0, 0, 0);
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4)
                    u1 = __builtin_amdgcn_mfma_f64_16x16x4f64(linv[256 + (4 * s4 + rg) * 16 + cl], t1[s4], u1, 0, 0, 0);
                if (cok) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        cp[4 * q] = u0[q];
                        cp[16 + 4 * q] = u1[q];
                    }
                }
            }
            __syncthreads();
            // trailing update: tiles in block-column-major order, an equal share per wave
            const int T = nblk * nblk, per = (T + LNW - 1) / LNW;
            const int tb = wv * per, te = min(T, tb + per);
            auto tile_ptr = [&](int t) {
                const int cb = t / nblk, rb = t - cb * nblk;
                return A + (size_t)min(c0 + 16 * cb + cl, N - 1) * N + k0 + LU_NB + 16 * rb + rg;
            };
            auto tile_load = [&](int t) {
                const int cb = t / nblk, rb = t - cb * nblk;
                const double *tp = tile_ptr(t);
                v4f64 r;
#pragma unroll
                for (int q = 0; q < 4; ++q) r[q] = (16 * rb + rg + 4 * q < mr) ? tp[4 * q] : 0.0;
                return r;
            };
            if (tb < te) {
                int curb = -1;
                v4f64 u0 = {0.0, 0.0, 0.0, 0.0}, u1 = {0.0, 0.0, 0.0, 0.0};
                v4f64 acc = tile_load(tb);
                for (int t = tb; t < te; ++t) {
                    const int cb = t / nblk, rb = t - cb * nblk;
                    if (cb != curb) {
                        const double *up = A + (size_t)min(c0 + 16 * cb + cl, N - 1) * N + k0 + rg;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            u0[q] = up[4 * q];
                            u1[q] = up[16 + 4 * q];
                        }
                        curb = cb;
                    }
                    v4f64 nxt = acc;
                    if (t + 1 < te) nxt = tile_load(t + 1);
                    const double *lp = pan + rg * m + min(LU_NB + 16 * rb + cl, m - 1);  // -L21[row cl][k = 4 s + rg]
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4)
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-lp[(4 * s4) * m], u0[s4], acc, 0, 0, 0);
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4)
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-lp[(16 + 4 * s4) * m], u1[s4], acc, 0, 0, 0);
                    if (c0 + 16 * cb + cl < N) {
                        double *tp = tile_ptr(t);
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (16 * rb + rg + 4 * q < mr) tp[4 * q] = acc[q];
                    }
                    acc = nxt;
                }
            }
            __syncthreads();
        }
#ifdef LN_TIMING
        if (tid == 0) { const long long n_ = clock64(); ln_cyc[7] += n_ - _tp; _tp = n_; }
#endif
    }
    LTOC(1);
}

__device__ __forceinline__ double lane_bcast(double v, int l) {  // l uniform: v_readlane, no LDS round trip
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ void wave_sync() {  // LDS traffic between the lanes of ONE wave: order it, no s_barrier
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// One wave solves (P L U) x = sign * b into xs (LDS, N doubles owned by the wave).  64 rows at a time, lane = row:
// the part of the row left of (right of, for U) the diagonal block is a dot product with the already known x, read
// down the columns of the column-major factors (coalesced).  Inside the 64 x 64 diagonal block the substitution
// chain runs on v_readlane broadcasts, eight columns per trip of a ROLLED loop (the kernel has to stay inside the
// instruction cache), the next eight columns in flight while the chain of the current eight executes.
// b == nullptr: the right-hand side is the unit vector e_unit.
__device__ __forceinline__ void wave_solve(const LnS &S, int N, const double *A, const double *b, int unit, double sign,
                                           double *xs) {
    const int lane = ln_tid() & 63;
    for (int jj = lane; jj < N; jj += 64) xs[jj] = b ? sign * b[S.perm[jj]] : (S.perm[jj] == unit ? sign : 0.0);
    wave_sync();
    for (int k0 = 0; k0 < N; k0 += 64) {  // L y = P b, unit lower
        const int r = k0 + lane, rc = min(r, N - 1);
        const int ng = (min(64, N - k0) + 7) >> 3;
        double acc = (r < N) ? xs[r] : 0.0;
#pragma unroll 8
        for (int c = 0; c < k0; ++c) acc = fma(-A[c * N + rc], xs[c], acc);
        double nxt[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) nxt[u] = A[min(k0 + u, N - 1) * N + rc];
#pragma unroll 1
        for (int g = 0; g < ng; ++g) {
            double cur[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) cur[u] = nxt[u];
            if (g + 1 < ng) {
#pragma unroll
                for (int u = 0; u < 8; ++u) nxt[u] = A[min(k0 + 8 * g + 8 + u, N - 1) * N + rc];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int c = 8 * g + u;  // rows below the diagonal of column k0 + c (columns past N see x = 0)
                acc = fma(-((lane > c && r < N) ? cur[u] : 0.0), lane_bcast(acc, c), acc);
            }
        }
        if (r < N) xs[r] = acc;
        wave_sync();
    }
    for (int k0 = 64 * ((N - 1) / 64); k0 >= 0; k0 -= 64) {  // U x = y
        const int r = k0 + lane, rc = min(r, N - 1);
        const int ng = (min(64, N - k0) + 7) >> 3;
        double acc = (r < N) ? xs[r] : 0.0;
#pragma unroll 8
        for (int c = k0 + 64; c < N; ++c) acc = fma(-A[c * N + rc], xs[c], acc);
        const double rd = (r < N) ? S.rdiag[r] : 0.0;
        double nxt[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) nxt[u] = A[min(k0 + 8 * (ng - 1) + u, N - 1) * N + rc];
#pragma unroll 1
        for (int g = ng - 1; g >= 0; --g) {
            double cur[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) cur[u] = nxt[u];
            if (g > 0) {
#pragma unroll
                for (int u = 0; u < 8; ++u) nxt[u] = A[(k0 + 8 * g - 8 + u) * N + rc];
            }
#pragma unroll
            for (int u = 7; u >= 0; --u) {
                const int c = 8 * g + u;  // lanes past the last row carry rd = 0: their "pivot" broadcasts 0
                const double xi = lane_bcast(acc * rd, c);
                acc = (lane == c) ? xi : acc;
                acc = fma(-((lane < c) ? cur[u] : 0.0), xi, acc);
            }
        }
        if (r < N) xs[r] = acc;
        wave_sync();
    }
}

// The same solve for ONE right-hand side with the whole workgroup: the part of a 64-row block that lies left of (right
// of, for U) its diagonal block is a 64 x k0 matrix-vector product -- 90 % of the loads of a solve at N = 300 -- and is
// split by column ranges over the 8 waves (partials through LDS); wave 0 then runs the substitution chain of the diagonal
// block as wave_solve does.  wave_solve leaves seven waves idle for ~130 us per Newton step.  xs: LDS, N doubles.
__device__ __forceinline__ void block_solve(LnS &S, int N, const double *A, const double *b, double sign, double *xs) {
    const int tid = ln_tid(), lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    double *part = S.part;  // LNW x 64 partial sums
    for (int jj = tid; jj < N; jj += LT) xs[jj] = sign * b[S.perm[jj]];
    __syncthreads();
    for (int k0 = 0; k0 < N; k0 += 64) {  // L y = P b, unit lower
        const int r = k0 + lane, rc = min(r, N - 1);
        {
            const int c0 = wv * k0 / LNW, c1 = (wv + 1) * k0 / LNW;
            double acc = 0.0;
            int c = c0;
            for (; c + 8 <= c1; c += 8) {
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = A[(c + u) * N + rc];
#pragma unroll
                for (int u = 0; u < 8; ++u) acc = fma(v[u], xs[c + u], acc);
            }
            for (; c < c1; ++c) acc = fma(A[c * N + rc], xs[c], acc);
            part[wv * 64 + lane] = acc;
        }
        __syncthreads();
        if (wv == 0) {
            const int ng = (min(64, N - k0) + 7) >> 3;
            double acc = (r < N) ? xs[r] : 0.0;
#pragma unroll
            for (int w = 0; w < LNW; ++w) acc -= part[w * 64 + lane];
            double nxt[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) nxt[u] = A[min(k0 + u, N - 1) * N + rc];
#pragma unroll 1
            for (int g = 0; g < ng; ++g) {
                double cur[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) cur[u] = nxt[u];
                if (g + 1 < ng) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) nxt[u] = A[min(k0 + 8 * g + 8 + u, N - 1) * N + rc];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int c = 8 * g + u;
                    acc = fma(-((lane > c && r < N) ? cur[u] : 0.0), lane_bcast(acc, c), acc);
                }
            }
            if (r < N) xs[r] = acc;
        }
        __syncthreads();
    }
    for (int k0 = 64 * ((N - 1) / 64); k0 >= 0; k0 -= 64) {  // U x = y
        const int r = k0 + lane, rc = min(r, N - 1);
        {
            const int lo = min(N, k0 + 64), span = N - lo;
            const int c0 = lo + wv * span / LNW, c1 = lo + (wv + 1) * span / LNW;
            double acc = 0.0;
            int c = c0;
            for (; c + 8 <= c1; c += 8) {
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = A[(c + u) * N + rc];
#pragma unroll
                for (int u = 0; u < 8; ++u) acc = fma(v[u], xs[c + u], acc);
            }
            for (; c < c1; ++c) acc = fma(A[c * N + rc], xs[c], acc);
            part[wv * 64 + lane] = acc;
        }
        __syncthreads();
        if (wv == 0) {
            const int ng = (min(64, N - k0) + 7) >> 3;
            double acc = (r < N) ? xs[r] : 0.0;
#pragma unroll
            for (int w = 0; w < LNW; ++w) acc -= part[w * 64 + lane];
            const double rd = (r < N) ? S.rdiag[r] : 0.0;
            double nxt[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) nxt[u] = A[min(k0 + 8 * (ng - 1) + u, N - 1) * N + rc];
#pragma unroll 1
            for (int g = ng - 1; g >= 0; --g) {
                double cur[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) cur[u] = nxt[u];
                if (g > 0) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) nxt[u] = A[(k0 + 8 * g - 8 + u) * N + rc];
                }
#pragma unroll
                for (int u = 7; u >= 0; --u) {
                    const int c = 8 * g + u;
                    const double xi = lane_bcast(acc * rd, c);
                    acc = (lane == c) ? xi : acc;
                    acc = fma(-((lane < c) ? cur[u] : 0.0), xi, acc);
                }
            }
            if (r < N) xs[r] = acc;
        }
        __syncthreads();
    }
}

// block_solve for the WIDE form (320 < N <= 640: ten blocks of 64 rows, 3.3 MB of factors per direction, from beyond the L2).
// Two changes to the schedule, none to the substitution: (i) the product of a block's rows with the x of all blocks but the one just
// solved does not wait for that one -- waves 1..7 form it WHILE wave 0 runs the chain of the block before, and only the last 64
// columns (eight per wave) follow behind the chain; (ii) sixteen columns in flight per wave instead of eight (one compute unit
// takes ~25 B per cycle from the L2 side only with ~60 KB outstanding).  part: two sets of LNW x 64 partial sums.
__device__ __forceinline__ double wide_cols(const double *A, int N, int rc, const double *xs, int c, int c1, double acc) {
    const double *a = A + (size_t)c * N + rc;
    for (; c + 16 <= c1; c += 16, a += 16 * (size_t)N) {
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = a[u * (size_t)N];
#pragma unroll
        for (int u = 0; u < 16; ++u) acc = fma(v[u], xs[c + u], acc);
    }
    for (; c + 4 <= c1; c += 4, a += 4 * (size_t)N) {
        double v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = a[u * (size_t)N];
#pragma unroll
        for (int u = 0; u < 4; ++u) acc = fma(v[u], xs[c + u], acc);
    }
    for (; c < c1; ++c, a += N) acc = fma(*a, xs[c], acc);
    return acc;
}

__device__ __forceinline__ void block_solve_wide(LnS &S, int N, const double *A, const double *b, double sign, double *xs) {
    const int tid = ln_tid(), lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    double *part = S.part;
#ifdef LN_TIMING
    const long long _tw0 = clock64();
    long long _tchain = 0;
#define SVT(stmt) do { const long long a_ = clock64(); stmt; _tchain += clock64() - a_; } while (0)
#else
#define SVT(stmt) stmt
#endif
    for (int jj = tid; jj < N; jj += LT) xs[jj] = sign * b[S.perm[jj]];
    __syncthreads();
    const int nblk = (N + 63) >> 6;
    // The chains of the WIDE form run on a diagonal block that is already in LDS: at these sizes the factors stream from beyond the L2
    // (~2-3 k cycles), and the eight-columns-ahead prefetch of block_solve stalled at every group -- 12.8 k cycles per block instead of
    // 3.8 k (timing build, N = 640); with the lane's whole row loaded first the chain still paid one such latency, 7.7 k.  So the block
    // of the NEXT chain is staged (stage_block: eight columns per wave, beside the loads of the step's last 64 columns) while nothing
    // reads the buffer -- a chain copies its row into registers first thing.  Same operations in the same order: the same bits.
    auto stage_block = [&](int kb) {  // every wave; barrier before the chain that uses it
        const int k0 = kb << 6, rc = min(k0 + lane, N - 1);
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = A[min(k0 + 8 * wv + u, N - 1) * N + rc];
#pragma unroll
        for (int u = 0; u < 8; ++u) S.stage[(8 * wv + u) * 64 + lane] = v[u];
    };
    auto chain_lower = [&](int kb) {  // wave 0: the 64 x 64 unit-lower diagonal block
        const int k0 = kb << 6, r = k0 + lane, rc = min(r, N - 1);
        const double *pp = part + (kb & 1) * (LNW * 64);
        const int ng = (min(64, N - k0) + 7) >> 3;
        double acc = (r < N) ? xs[r] : 0.0;
#pragma unroll
        for (int w = 0; w < LNW; ++w) acc -= pp[w * 64 + lane];
#pragma unroll
        for (int h = 0; h < 2; ++h) {  // (the row in two halves of 32 registers: LDS reads, no latency to hide -- and 64 more live
                                       //  registers here once made the compiler's spill code of this kernel fault on the device)
            double dv[32];
#pragma unroll
            for (int c = 0; c < 32; ++c) dv[c] = S.stage[(32 * h + c) * 64 + lane];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (4 * h + g < ng) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int c = 32 * h + 8 * g + u;
                        acc = fma(-((lane > c && r < N) ? dv[8 * g + u] : 0.0), lane_bcast(acc, c), acc);
                    }
                }
            }
        }
        if (r < N) xs[r] = acc;
    };
    auto chain_upper = [&](int kb, int t) {  // wave 0: the upper diagonal block with its reciprocal pivots
        const int k0 = kb << 6, r = k0 + lane, rc = min(r, N - 1);
        const double *pp = part + (t & 1) * (LNW * 64);
        const int ng = (min(64, N - k0) + 7) >> 3;
        double acc = (r < N) ? xs[r] : 0.0;
#pragma unroll
        for (int w = 0; w < LNW; ++w) acc -= pp[w * 64 + lane];
        const double rd = (r < N) ? S.rdiag[r] : 0.0;
#pragma unroll
        for (int h = 1; h >= 0; --h) {
            double dv[32];
#pragma unroll
            for (int c = 0; c < 32; ++c) dv[c] = S.stage[(32 * h + c) * 64 + lane];
#pragma unroll
            for (int g = 3; g >= 0; --g) {
                if (4 * h + g < ng) {
#pragma unroll
                    for (int u = 7; u >= 0; --u) {
                        const int c = 32 * h + 8 * g + u;
                        const double xi = lane_bcast(acc * rd, c);
                        acc = (lane == c) ? xi : acc;
                        acc = fma(-((lane < c) ? dv[8 * g + u] : 0.0), xi, acc);
                    }
                }
            }
        }
        if (r < N) xs[r] = acc;
    };
    stage_block(0);
    __syncthreads();
    for (int kb = 0; kb < nblk; ++kb) {  // L y = P b, unit lower
        const int k0 = kb << 6, rc = min(k0 + lane, N - 1);
        const int early = max(0, k0 - 64);
        double acc = 0.0;
        if (wv == 0) {
            if (kb > 0) SVT(chain_lower(kb - 1));
        } else {
            acc = wide_cols(A, N, rc, xs, (wv - 1) * early / (LNW - 1), wv * early / (LNW - 1), 0.0);
        }
        __syncthreads();
        {
            const int late = k0 - early;
            if (kb > 0) stage_block(kb);  // (chain(kb - 1) has its block in registers since before the barrier above; block 0 is in already)
            acc = wide_cols(A, N, rc, xs, early + wv * late / LNW, early + (wv + 1) * late / LNW, acc);
            part[(kb & 1) * (LNW * 64) + wv * 64 + lane] = acc;
        }
        __syncthreads();
    }
    if (wv == 0) SVT(chain_lower(nblk - 1));
    __syncthreads();
    stage_block(nblk - 1);
    __syncthreads();
    for (int t = 0; t < nblk; ++t) {  // U x = y
        const int kb = nblk - 1 - t, k0 = kb << 6, rc = min(k0 + lane, N - 1);
        const int lo = min(N, k0 + 64), mid = min(N, lo + 64);  // [lo, mid): the block just solved; [mid, N): the ones before it
        double acc = 0.0;
        if (wv == 0) {
            if (t > 0) SVT(chain_upper(kb + 1, t - 1));
        } else {
            const int span = N - mid;
            acc = wide_cols(A, N, rc, xs, mid + (wv - 1) * span / (LNW - 1), mid + wv * span / (LNW - 1), 0.0);
        }
        __syncthreads();
        {
            const int late = mid - lo;
            if (t > 0) stage_block(kb);
            acc = wide_cols(A, N, rc, xs, lo + wv * late / LNW, lo + (wv + 1) * late / LNW, acc);
            part[(t & 1) * (LNW * 64) + wv * 64 + lane] = acc;
        }
        __syncthreads();
    }
    if (wv == 0) SVT(chain_upper(0, nblk - 1));
    __syncthreads();
#ifdef LN_TIMING
    if (tid == 0) {  // (timing build: the chains' share of the directions in the slots of the pivoted LU, which this form does not have)
        ln_cyc[5] += _tchain;
        ln_cyc[7] += clock64() - _tw0;
    }
#endif
#undef SVT
}

__device__ __forceinline__ void accept_trial(LnS &S, int N) {
    for (int i = ln_tid(); i < N; i += LT) {
        S.x[i] = S.xn[i];
        S.I[i] = S.In[i];
        S.Sx[i] = S.Sxn[i];
        S.MI[i] = S.MIn[i];
    }
    __syncthreads();
}

// One back-tracking pass along `dir` from S.x -- the single place in the Newton loop where the objective is
// evaluated (one copy of the code in the instruction cache):
//   fallback == false: LineSearch.__call__(fun, jac, x, dir, fx, root=False)  (minimizer.py:70-187) with
//                      reduce_step = limit_step (statistical_models.py:1126-1130); acceptance by the Armijo rule,
//                      quadratic / cubic model for the next step length;
//   fallback == true : the last resort of MinimizeNewton (minimizer.py:256-267): ten trials with the step divided by
//                      16 each time, accepted as soon as the objective decreases.
// amin = min |x / dir| and slope = jac . dir come from the caller's reduction.
// returns 0 accepted (x, fx updated; reduction too for the line search), 1 failed, -1 "Round off in slope calculation".
template <bool WIDE>
__device__ __forceinline__ int backtrack(const LogNormalParams &P, LnS &S, const double *dir, double amin, double slope,
                                         bool fallback, double &fx, int &nfev, double &reduction) {
    const int N = P.N, tid = ln_tid();
    const double armijo = 1e-4, l_min = 0.1;
    const double cost = fx;
    double alpha = 1.1 * amin;
    if (1.0 < alpha) alpha = 1.0;
    double delta_f = slope;  // alpha == 1: p = dir bit for bit, and jac . p is the caller's sum
    const double *p = dir;
    if (alpha != 1.0) {
        double df = 0.0;
        for (int i = tid; i < N; i += LT) {
            const double pi = alpha * dir[i];
            S.pd[i] = pi;
            df += S.jx[i] * pi;
        }
        delta_f = block_sum(S, df);
        p = S.pd;
    }
    if (!fallback && delta_f > 0) return -1;
    double lam = 1.0, cost_save = 0.0, lam_save = 0.0;
    const int max_trials = fallback ? 10 : LS_MAX_TRIALS;
    for (int trial = 0; trial < max_trials; ++trial) {
        int moved = 0;
        for (int i = tid; i < N; i += LT) {
            const double xn = S.x[i] + lam * p[i];
            S.xn[i] = xn;
            S.In[i] = exp(xn + P.s0);
            moved |= (xn != S.x[i]);
        }
        moved = __syncthreads_or(moved);
        if (!fallback && !moved) return 1;
        // (fresh_products: S^-1 x_n multiplied out for every trial point, the reference's arithmetic)
        const bool fresh = P.fresh_products != 0;
        const double *sv = fresh ? S.xn : (trial == 0 ? p : nullptr);
        const double cost_new = ln_eval<WIDE>(P, S, S.xn, S.In, S.Sxn, S.MIn, sv, fresh ? S.Sxn : S.col, !fresh, lam);
        ++nfev;
        if (fallback ? (cost_new < cost) : (cost_new <= (cost + armijo * lam * delta_f))) {
            if (!fallback) reduction = lam;
            accept_trial(S, N);
            fx = cost_new;
            return 0;
        }
        if (fallback) {
            lam *= 0.0625;  // dx *= 2**-4: exact, so x + lam p carries the reference's bits
            continue;
        }
        double lam_new;
        if (lam == 1.0) {
            lam_new = -0.5 * delta_f / (cost_new - cost - delta_f);
        } else {
            const double r1 = (cost_new - cost - lam * delta_f) / (lam * lam);
            const double r2 = (cost_save - cost - lam_save * delta_f) / (lam_save * lam_save);
            const double a = (r1 - r2) / (lam - lam_save);
            const double b = (lam * r2 - lam_save * r1) / (lam - lam_save);
            if (a == 0) {
                lam_new = -0.5 * delta_f / b;
            } else {
                const double d = b * b - 3 * a * delta_f;
                if (d < 0) lam_new = 0.5 * lam;
                else if (b <= 0) lam_new = (-b + sqrt(d)) / (3 * a);
                else lam_new = -1 * delta_f / (b + sqrt(d));
                lam_new = (lam_new < 0.5 * lam) ? lam_new : 0.5 * lam;
            }
        }
        if (lam_new != lam_new) lam_new = l_min * lam;
        lam_save = lam;
        cost_save = cost_new;
        lam = (l_min * lam > lam_new) ? l_min * lam : lam_new;
    }
    return 1;
}

struct NewtonExit {
    int status, nstep, nfev, nhess;
};

// MinimizeNewton(H, jac, hess, S.x, LineSearch(reduce_step=limit_step), tol=1e-7)  (minimizer.py:190-283)
template <bool WIDE>
__device__ __forceinline__ NewtonExit minimize_newton(const LogNormalParams &P, LnS &S) {
    const int N = P.N, tid = ln_tid();
    bool need_hess = true;
    int nfev = 1, nhess = 0;
    double reduction = NAN;  // LineSearch.reduction starts as None
    int reuse = 0;
    bool have_inv = false;
    const int inv_after = max(2, N >> 3);
    for (int i = tid; i < N; i += LT) S.I[i] = exp(S.x[i] + P.s0);
    __syncthreads();
    double fx = ln_eval<WIDE>(P, S, S.x, S.I, S.Sx, S.MI, S.x, S.Sx, false, 0.0);
    for (int i = tid; i < N; i += LT) S.jx[i] = ln_grad(S, i);
    __syncthreads();
    for (int nstep = 0; nstep < P.max_step; ++nstep) {
        if (need_hess) {
            if (nhess == P.max_hev) return {3, nstep, nfev, nhess};
            double *Cp = S.lu_nb > 0 ? P.LU + N * N : nullptr;
            if (S.lu_nb > 0) {
                if (!P.no_cholesky) build_hess_padded(P, S, Cp);
                if constexpr (WIDE) {  // (no pivoted LU beyond N = 320: its panel keeps a row per thread; the host takes the other route)
                    if (!cholesky_as_lu<true>(P, S, Cp)) return {5, nstep, nfev, nhess};
                } else
                if (P.no_cholesky || !cholesky_as_lu<false>(P, S, Cp)) {  // not positive definite: the attempt has written into S.lu
#ifdef LN_TIMING
                    if (tid == 0) ln_cyc[6] += 1;
#endif
                    build_hess(P, S, S.lu, N, false);
                    lu_factor_blocked(S, N, S.lu);
                }
            } else {
                build_hess(P, S, S.lu, N, false);
                lu_factor(S, N, S.lu);
            }
            ++nhess;
            reuse = 0;
            have_inv = false;
        }
        // S.jx = jac(x) (from the end of the previous step); dx = -hess^-1 jac.  A factorisation that keeps being
        // re-used (reduction == 1: the Hessian is frozen, minimizer.py:273) is turned into the explicit inverse once
        // it has served N/8 solves -- N wave-solves shared by the 8 waves cost as much as that -- and every later
        // step is a matrix-vector product over all threads instead of a substitution chain in one wave.
        LTIC();
        if (!WIDE && !have_inv && reuse >= inv_after) {  // (WIDE: the row / chunk split of the product below assumes N <= 512 threads' worth)
            for (int r = __builtin_amdgcn_readfirstlane(tid >> 6); r < N; r += LNW) {
                double *xs = S.wsol + (tid >> 6) * N;
                wave_solve(S, N, S.lu, nullptr, r, 1.0, xs);
                for (int i = tid & 63; i < N; i += 64) P.Hinv[r * N + i] = xs[i];  // column r of hess^-1
            }
            __syncthreads();
            have_inv = true;
        }
        if (have_inv) {
            if (S.row >= 0 && S.pair) {
                const int N2 = N >> 1;
                const v2f64 *hc = reinterpret_cast<const v2f64 *>(__builtin_assume_aligned(P.Hinv + (S.c0 * N + S.row), 16));
                v2f64 a = {0.0, 0.0};
                int c = S.c0;
                for (; c + HVB <= S.c1; c += HVB, hc += HVB * N2) {
                    v2f64 vh[HVB];
#pragma unroll
                    for (int u = 0; u < HVB; ++u) vh[u] = hc[u * N2];
#pragma unroll
                    for (int u = 0; u < HVB; ++u) {
                        a[0] = fma(vh[u][0], S.jx[c + u], a[0]);
                        a[1] = fma(vh[u][1], S.jx[c + u], a[1]);
                    }
                }
                for (; c < S.c1; ++c, hc += N2) {
                    const v2f64 vh = *hc;
                    a[0] = fma(vh[0], S.jx[c], a[0]);
                    a[1] = fma(vh[1], S.jx[c], a[1]);
                }
                S.pbuf[S.slot] = a[0];
                S.pbuf[S.slot + 1] = a[1];
            } else if (S.row >= 0) {
                const double *hc = P.Hinv + (S.c0 * N + S.row);
                double a = 0.0;
                int c = S.c0;
                // the loads of 16 columns are issued as one batch: what a single CU can pull from L2 is set by the
                // number of requests in flight, and the compiler does not always keep an unrolled loop's loads together
                for (; c + HVB <= S.c1; c += HVB, hc += HVB * N) {
                    double vh[HVB];
#pragma unroll
                    for (int u = 0; u < HVB; ++u) vh[u] = hc[u * N];
#pragma unroll
                    for (int u = 0; u < HVB; ++u) a = fma(vh[u], S.jx[c + u], a);
                }
                for (; c < S.c1; ++c, hc += N) a = fma(*hc, S.jx[c], a);
                S.pbuf[S.slot] = a;
            }
            __syncthreads();
            if (tid < N) {
                double a = 0.0;
                for (int ch = 0; ch < S.nch; ++ch) a += S.pbuf[ch * N + tid];
                S.dx[tid] = -a;
            }
        } else if (S.lu_nb > 0) {
            if constexpr (WIDE)
                block_solve_wide(S, N, S.lu, S.jx, -1.0, S.dx);
            else
                block_solve(S, N, S.lu, S.jx, -1.0, S.dx);  // factors in L2: every wave streams its share of them
        } else if (tid < 64) {
            wave_solve(S, N, S.lu, S.jx, -1, -1.0, S.dx);
        }
        ++reuse;
        __syncthreads();
        LTOC(2);
        double am = INFINITY, d = 0.0;
        for (int i = tid; i < N; i += LT) {
            am = fmin(am, fabs(S.x[i] / S.dx[i]));
            d += S.jx[i] * S.dx[i];
        }
        block_min_sum(S, am, d);
        // attempt 0: Newton direction (only if it descends); 1: steepest descent; 2: descent, shrinking by 16
        // (minimizer.py:243-271).  `failed` is the outcome of attempt 0, as in the reference.
        int failed = 1, res = 1;
        for (int attempt = (d < 0) ? 0 : 1; attempt < 3; ++attempt) {
            if (attempt == 1) {
                am = INFINITY;
                d = 0.0;
                for (int i = tid; i < N; i += LT) {
                    const double gi = -S.jx[i];
                    S.dx[i] = gi;
                    am = fmin(am, fabs(S.x[i] / gi));
                    d += S.jx[i] * gi;
                }
                block_min_sum(S, am, d);
            }
            res = backtrack<WIDE>(P, S, S.dx, am, d, attempt == 2, fx, nfev, reduction);
            if (res < 0) return {4, nstep, nfev, nhess};
            if (attempt == 0) failed = res;
            if (res == 0) break;
        }
        if (res != 0) return {1, nstep, nfev, nhess};  // neither direction improves the solution
        need_hess = failed || (reduction != 1.0);
        double g = -INFINITY;
        for (int i = tid; i < N; i += LT) {
            const double gi = ln_grad(S, i);
            S.jx[i] = gi;
            g = fmax(g, fabs(gi) * fabs(S.x[i]));
        }
        g = block_minmax<true>(S, g);
        if (g < P.newton_tol * fmax(fabs(fx), 1.0)) return {0, nstep, nfev, nhess};
    }
    return {2, P.max_step - 1, nfev, nhess};
}

// S^-1 = Y^T diag(1/p) Y  (statistical_models.py:1061) on the matrix cores: tile (I, J) = sum_k (Y_k,I / p_k)^T Y_k,J, the
// A fragment of k-step s being rows 4s..4s+3 of Y's block column I scaled by 1/p, the B fragment the same rows of block
// column J.  A wave owns a 2 x 2 group of tiles on or below the block diagonal (four fragment loads feed four MFMAs);
// the upper triangle is the mirror image, so S^-1 is symmetric bit for bit: the transposed tile comes from four more
// MFMAs against the identity (the accumulator registers of a tile ARE the A fragments of its transpose) and both are
// stored along rows.  rk: LDS, 1 / p.  (The scalar loop this replaces took 6 ms per power-spectrum iteration at N = 300
// -- two strided 8-byte loads per multiply-add on one CU; this takes ~0.2 ms.)
__device__ __forceinline__ void build_sinv(const LogNormalParams &P, const double *rk, int part = 0, int nparts = 1) {
    using namespace tilechol;
    const int N = P.N, tid = ln_tid(), lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (scalar: loop control on the SALU)
    const int cl = lane & 15, rg = lane >> 4;
    const int nbt = (N + 15) >> 4, G = (nbt + 1) >> 1, ngroups = G * (G + 1) / 2;
    const int ksteps = (N + 3) >> 2;
    const gdouble *Y = as_global(P.Y);
    gdouble *out = as_global(P.Sinv);
    double ident[4];  // B fragments of the 16 x 16 identity
#pragma unroll
    for (int q = 0; q < 4; ++q) ident[q] = (4 * q + rg == cl) ? 1.0 : 0.0;
    for (int g = part * LNW + wave; g < ngroups; g += nparts * LNW) {  // (a cluster deals the tile groups to its workgroups)
        int gi = (int)((sqrtf(8.0f * (float)g + 1.0f) - 1.0f) * 0.5f);
        while ((gi + 1) * (gi + 2) / 2 <= g) ++gi;
        while (gi * (gi + 1) / 2 > g) --gi;
        const int gj = g - gi * (gi + 1) / 2;
        const int I0 = 2 * gi, J0 = 2 * gj;
        // columns past N are clamped: they only reach tile elements that are never stored
        const int ca0 = min(16 * I0 + cl, N - 1), ca1 = min(16 * I0 + 16 + cl, N - 1);
        const int cb0 = min(16 * J0 + cl, N - 1), cb1 = min(16 * J0 + 16 + cl, N - 1);
        v4f64 d00 = {0.0, 0.0, 0.0, 0.0}, d01 = d00, d10 = d00, d11 = d00;
        // k-steps in batches of KU through two named register sets: the loads of the next batch are in flight while the
        // sixteen MFMAs of the current one issue (every load unconditional, rows clamped: the waits stay exact counts)
        constexpr int KU = 4;
        struct Frags {
            double fa0[KU], fa1[KU], fb0[KU], fb1[KU], r[KU];
        };
        auto load_frags = [&](Frags &f, int s0) {
#pragma unroll
            for (int u = 0; u < KU; ++u) {
                const int kc = min(4 * (s0 + u) + rg, N - 1);
                const gdouble *row = Y + (size_t)kc * N;
                f.fa0[u] = row[ca0];
                f.fa1[u] = row[ca1];
                f.fb0[u] = row[cb0];
                f.fb1[u] = row[cb1];
                f.r[u] = rk[kc];
            }
        };
        auto mul_frags = [&](const Frags &f, int s0) {
#pragma unroll
            for (int u = 0; u < KU; ++u) {
                const int k = 4 * (s0 + u) + rg;
                const double r = (k < N) ? f.r[u] : 0.0;  // rows past N (and k-steps past the last one) contribute nothing
                const double a0 = f.fa0[u] * r, a1 = f.fa1[u] * r;
                d00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, f.fb0[u], d00, 0, 0, 0);
                d01 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, f.fb1[u], d01, 0, 0, 0);
                d10 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, f.fb0[u], d10, 0, 0, 0);
                d11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, f.fb1[u], d11, 0, 0, 0);
            }
        };
        Frags A, B;
        load_frags(A, 0);
        for (int s0 = 0; s0 < ksteps; s0 += 2 * KU) {
            load_frags(B, s0 + KU);
            mul_frags(A, s0);
            load_frags(A, s0 + 2 * KU);
            mul_frags(B, s0 + KU);  // (k-steps past the last one: r = 0, the products add exact zeros)
        }
        auto put = [&](int I, int J, const v4f64 &d) {
            if (I >= nbt || J > I) return;
            v4f64 t = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int q = 0; q < 4; ++q) t = __builtin_amdgcn_mfma_f64_16x16x4f64(d[q], ident[q], t, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int lr = rg + 4 * q;  // element (lr, cl) of the tile: d = D[lr][cl], t = D[cl][lr]
                const int a = 16 * I + lr, b = 16 * J + cl;
                if (I == J) {
                    if (a < N && b < N) out[(size_t)a * N + b] = (lr >= cl) ? d[q] : t[q];
                } else {
                    if (a < N && b < N) out[(size_t)a * N + b] = d[q];
                    const int at = 16 * J + lr, bt = 16 * I + cl;  // element (lr, cl) of the mirror tile (J, I)
                    if (at < N && bt < N) out[(size_t)at * N + bt] = t[q];
                }
            }
        };
        put(I0, J0, d00);
        put(I0, J0 + 1, d01);
        put(I0 + 1, J0, d10);
        put(I0 + 1, J0 + 1, d11);
    }
}

// Tr2_r = y_r^T Dinv^-1 y_r for every row y_r of Y (filter.py:168-170) from the Cholesky factors of Dinv = L L^T:
// Tr2_r = |L^-1 y_r|^2, a triangular solve with N right-hand sides on the matrix cores.  The right-hand sides are taken 16
// at a time (block column c = rows 16c.. of Y); a wave owns up to three block columns and walks down the block rows:
//     Z_I = L_II^-1 (Y^T_I - sum_{J<I} L'_IJ W_J),   W_J = D_J Z_J,   L' = L D^-1 (the unit-lower factor in S.lu)
// The accumulator registers of a tile are the B fragments of the same tile, so W_J goes to a lane-private scratch (Wsc)
// and comes back as an operand without any reshuffling; L_II^-1 was kept by the factorisation (Xd); the transposed tile
// of Y comes from four MFMAs against the identity.  No barriers: the block columns are independent.  The substitution
// by waves this replaces (wave_solve per row) took 6.5 ms per power-spectrum iteration at N = 300.
// (lu: the factors in global memory; dvec: diag(L) -- LDS for the workgroup that factored, a global copy for the helpers of a
//  cluster; part / nparts: the block columns are dealt to the nparts x LNW waves of the cluster)
// NC: block columns per wave; JB: products per column whose operands are loaded as one batch.  A lone column per wave (the
// cluster) is a chain of nb (nb + 1) / 2 products: with the operands of one product in flight it waited for an L2 round trip per
// product (285 k cycles per pass on four workgroups against 523 k on one); the sums run over J in the same order either way.
template <int NC, int JB>
__device__ __forceinline__ void tr2_solve_t(const LogNormalParams &P, const double *lu_p, const double *dvec, const double *Xd,
                                            double *Wsc, double *tr2, int part, int nparts, int cbase = 0) {
    using namespace tilechol;
    const int N = P.N, NP = P.NP, nb = NP / 16;
    const int tid = ln_tid(), lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (scalar: loop control on the SALU)
    const int cl = lane & 15, rg = lane >> 4;
    const int cw = cbase + wave * nparts + part, cstride = nparts * LNW;  // this wave's first block column (dealt round the workgroups:
                                                                  // each gets as many busy waves as the others), the stride to its next
    const gdouble *lu = as_global(lu_p), *Y = as_global(P.Y), *X = as_global(Xd);
    gdouble *W = as_global(Wsc);
    if (NC == 1 && cw >= nb) return;  // (no barriers in here: the block columns are independent)
    double ident[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) ident[q] = (4 * q + rg == cl) ? 1.0 : 0.0;
    v4f64 ss[NC];
#pragma unroll
    for (int u = 0; u < NC; ++u) ss[u] = v4f64{0.0, 0.0, 0.0, 0.0};
    // the tiles of Y one block row ahead (Y was last read at the start of the pass: these loads miss the L2, ~5 k cycles, and
    // sat at the top of every row of a chain that is nb rows long)
    double ty[NC][4];
    auto load_y = [&](int I) {
#pragma unroll
        for (int u = 0; u < NC; ++u) {
            const int c = cw + u * cstride;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = 16 * c + rg + 4 * q, col = 16 * I + cl;
                ty[u][q] = (c < nb && row < N && col < N) ? Y[(size_t)row * N + col] : 0.0;
            }
        }
    };
    load_y(0);
    for (int I = 0; I < nb; ++I) {
        v4f64 acc[NC];
#pragma unroll
        for (int u = 0; u < NC; ++u) {
            acc[u] = v4f64{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(ty[u][q], ident[q], acc[u], 0, 0, 0);
        }
        load_y(min(I + 1, nb - 1));
        const int lrow = 16 * I + cl;
        const bool lvalid = lrow < N;
        const gdouble *la = lu + min(lrow, N - 1);
        double fx[4], dI[4];  // (the operands of the row's last product: in flight while the sum over J runs)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            fx[q] = X[I * 256 + cl * 16 + 4 * q + rg];
            dI[q] = dvec[16 * I + rg + 4 * q];
        }
        if constexpr (JB == 0) {
#pragma unroll 2
            for (int J = 0; J < I; ++J) {
                double fa[4], fb[NC][4];
#pragma unroll
                for (int q = 0; q < 4; ++q) fa[q] = la[(size_t)(16 * J + 4 * q + rg) * N];
#pragma unroll
                for (int u = 0; u < NC; ++u) {
                    const int c = cw + u * cstride;
                    if (c < nb) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) fb[u][q] = W[((size_t)(c * nb + J) * 4 + q) * 64 + lane];
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) fa[q] = lvalid ? -fa[q] : 0.0;
#pragma unroll
                for (int u = 0; u < NC; ++u) {
                    const int c = cw + u * cstride;
                    if (c < nb) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[q], fb[u][q], acc[u], 0, 0, 0);
                    }
                }
            }
        } else {
            // A column per wave: a chain of nb (nb + 1) / 2 products.  The operands of JB products are one batch; two named
            // register sets, the next batch in flight while the current one multiplies; every load is issued (clamped to
            // the last product of the row), never predicated, so that the waits stay exact counts.
            static_assert(JB == 0 || NC == 1, "batches: one column per wave");
            constexpr int JBB = JB > 0 ? JB : 1;
            struct OperandSet {
                double fa[JBB][4], fb[JBB][4];
            };
            const int jlast = max(I - 1, 0);
            auto load_set = [&](OperandSet &o, int J0) {
#pragma unroll
                for (int jb = 0; jb < JBB; ++jb) {
                    const int Jc = min(J0 + jb, jlast);
#pragma unroll
                    for (int q = 0; q < 4; ++q) o.fa[jb][q] = la[(size_t)(16 * Jc + 4 * q + rg) * N];
#pragma unroll
                    for (int q = 0; q < 4; ++q) o.fb[jb][q] = W[((size_t)(cw * nb + Jc) * 4 + q) * 64 + lane];
                }
            };
            auto mul_set = [&](const OperandSet &o, int J0) {
#pragma unroll
                for (int jb = 0; jb < JBB; ++jb) {
                    if (J0 + jb < I) {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(lvalid ? -o.fa[jb][q] : 0.0, o.fb[jb][q], acc[0], 0, 0, 0);
                    }
                }
            };
            OperandSet A, B;
            load_set(A, 0);
            for (int J0 = 0; J0 < I; J0 += 2 * JBB) {
                load_set(B, J0 + JBB);
                mul_set(A, J0);
                load_set(A, J0 + 2 * JBB);
                mul_set(B, J0 + JBB);
            }
        }
#pragma unroll
        for (int u = 0; u < NC; ++u) {
            const int c = cw + u * cstride;
            if (c < nb) {
                v4f64 z = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int q = 0; q < 4; ++q) z = __builtin_amdgcn_mfma_f64_16x16x4f64(fx[q], acc[u][q], z, 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    ss[u][q] = fma(z[q], z[q], ss[u][q]);
                    W[((size_t)(c * nb + I) * 4 + q) * 64 + lane] = z[q] * dI[q];
                }
            }
        }
    }
#pragma unroll
    for (int u = 0; u < NC; ++u) {
        const int c = cw + u * cstride;
        double t = (ss[u][0] + ss[u][1]) + (ss[u][2] + ss[u][3]);
        t += __shfl_xor(t, 16);
        t += __shfl_xor(t, 32);
        if (c < nb && rg == 0 && 16 * c + cl < N) tr2[16 * c + cl] = t;
    }
}

template <bool WIDE>
__device__ __forceinline__ void tr2_solve(const LogNormalParams &P, const double *lu_p, const double *dvec, const double *Xd,
                                          double *Wsc, double *tr2, int part = 0, int nparts = 1) {
    if (P.NP / 16 <= nparts * LNW) tr2_solve_t<1, 2>(P, lu_p, dvec, Xd, Wsc, tr2, part, nparts);  // a column per wave at most
    else if constexpr (!WIDE) tr2_solve_t<3, 0>(P, lu_p, dvec, Xd, Wsc, tr2, part, nparts);       // up to three: nb <= 24
    else  // 320 < N <= 639 on few workgroups: up to three per wave and round, round after round
        for (int cb = 0; cb < P.NP / 16; cb += 3 * nparts * LNW) tr2_solve_t<3, 0>(P, lu_p, dvec, Xd, Wsc, tr2, part, nparts, cb);
}

// ---- cluster: a few workgroups on one fit ---------------------------------------------------------------------------------
// One CU computes the two per-pass pieces that are plain parallel work -- S^-1 = Y^T diag(1/p) Y (N^3 multiply-adds, 0.2 ms at
// N = 300) and the Tr2 triangular solve with N right-hand sides (0.25 ms) -- at its matrix-pipe rate; together they are a fifth
// of a default-mode fit.  A cluster is `cluster` workgroups of ONE launch: workgroup ids 0, 8, 16, ... (ids go round the eight
// XCDs, so these share an L2; the ids between them return at once).  The first runs the fit; the others wait for commands:
//   ctl[0] sequence number << 8 | command (one word: one load tells a helper everything), ctl[2] helpers done with it,
//   ctl[3] helpers that have started, ctl[4] disbanded, ctl[5] the XCDs the workgroups sit on (bit mask), ctl[6] all on one.
// Hand-over = a release fence by thread 0 behind a barrier + an agent-scope atomic; the receiver's acquire fence invalidates its
// L1.  Every wait is bounded (wall clock): a cluster whose helpers do not all show up within 200 us is disbanded and the first
// workgroup runs alone; a helper that hears nothing for 20 s leaves; a command that the helpers do not finish within 2 s ends
// the fit with LN_STATUS_CLUSTER.  Nothing can hang.
__device__ __forceinline__ int ctl_load(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void ctl_store(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xf; }  // HW_REG_XCC_ID
// Release of everything this workgroup has stored (called by thread 0 behind a barrier).  An agent-scope release writes the
// XCD's L2 back (tens of microseconds with megabytes of dirty Hessian in it); when every workgroup of the cluster sits on the
// SAME XCD -- checked at start-up from the hardware register, ctl[6] -- the L2 is shared and it is enough that the stores have
// left this CU (the L1 is write-through: the barrier's wait for their acknowledgement) before the flag, an agent-scope atomic
// executed in that L2, is raised.
__device__ __forceinline__ void cluster_release(bool same_xcd) {
    if (same_xcd) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    else __threadfence();
}
// Acquire before reading what another workgroup wrote: invalidates this CU's L1 (and, across XCDs, what the L2 holds of it)
__device__ __forceinline__ void cluster_acquire(bool same_xcd) {
    if (same_xcd) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    else __threadfence();
}
// first workgroup, all threads: publish a command (everything written so far becomes visible to the helpers)
// (the command travels IN the sequence word -- sequence number << 8 | command | LN_CMD_WITH_S --: a helper that sees the word
//  change has everything in that one load; command, flag and XCD bit in words of their own were three more L2 round trips per
//  hand-over, one behind the other)
__device__ __forceinline__ void cluster_dispatch(const LogNormalParams &P, int cmd, bool same_xcd, int &seq) {
    __syncthreads();
    ++seq;
    if (ln_tid() == 0) {
        ctl_store(&P.ctl[2], 0);
        cluster_release(same_xcd);
        ctl_store(&P.ctl[0], (int)(((unsigned)seq << 8) | (unsigned)cmd));  // (wraps after 2^24 commands: still a new word every time)
    }
}
// first workgroup, all threads: wait for the helpers; false on a timeout
__device__ __forceinline__ bool cluster_wait(const LogNormalParams &P, int *s_flag, bool same_xcd) {
    __syncthreads();
    if (ln_tid() == 0) {
        const long long t0 = wall_clock64();
        int ok = 1;
        while (ctl_load(&P.ctl[2]) < P.cluster - 1) {
            __builtin_amdgcn_s_sleep(2);
            if (wall_clock64() - t0 > 200000000ll) {  // 2 s at 100 MHz
                ok = 0;
                break;
            }
        }
        cluster_acquire(same_xcd);
        *s_flag = ok;
    }
    __syncthreads();
    return *s_flag != 0;
}
// a helper workgroup (member 1 .. cluster - 1): serve commands until told to leave
template <bool WIDE>
__device__ __forceinline__ void cluster_helper(const LogNormalParams &P, int member, int *s_cmd) {
    const int tid = ln_tid();
    if (tid == 0) {
        __hip_atomic_fetch_or(&P.ctl[5], 1 << xcc_id(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&P.ctl[3], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    int seen = 0, same = -1, chol_epoch = 0;
    for (;;) {
        if (tid == 0) {
            const long long t0 = wall_clock64();
            int cmd = LN_CMD_EXIT;
            for (;;) {
                if (ctl_load(&P.ctl[4])) break;            // disbanded
                const int word = ctl_load(&P.ctl[0]);
                if (word != seen) {
                    seen = word;
                    cmd = word & 0xff;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
                if (wall_clock64() - t0 > 2000000000ll) break;  // 20 s of silence
            }
            if (same < 0) same = ctl_load(&P.ctl[6]) != 0;  // (settled before the first command)
            cluster_acquire(same != 0);  // (what the first workgroup wrote before the command)
            s_cmd[0] = cmd;
            s_cmd[1] = same;
        }
        __syncthreads();
        const bool with_s = (s_cmd[0] & LN_CMD_WITH_S) != 0;
        const int cmd = s_cmd[0] & 0xf;
        const bool same_xcd = s_cmd[1] != 0;
        __syncthreads();
        if (cmd == LN_CMD_EXIT || cmd == LN_CMD_NONE) return;
        if (cmd == LN_CMD_SINV) {
            build_sinv(P, P.rk_g, member, P.cluster);
        } else if (cmd == LN_CMD_EVAL) {
            extern __shared__ __attribute__((aligned(16))) double smem[];  // (a helper's LDS is otherwise unused)
            const double *vecs = ln_eval_vecs(P);
            for (int i = tid; i < P.N; i += LT) {
                if (with_s) smem[i] = vecs[i];
                smem[P.NP + i] = vecs[P.NP + i];
            }
            __syncthreads();
            ln_eval_items(P, with_s ? smem : nullptr, smem + P.NP, member * LT + tid, P.cluster * LT);
        } else if (cmd == LN_CMD_CHOL) {
            chol_helper(P, member, ++chol_epoch);
        } else if (cmd == LN_CMD_HESS) {
            build_hess_tiles(P, P.rk_g, P.tr2_g, P.j, P.LU + P.N * P.N, member, P.cluster);
        } else if (cmd == LN_CMD_TR2) {
            double *const Cp = P.LU + P.N * P.N, *const Wsc = Cp + P.NP * P.NP, *const Xd = Wsc + P.NP * P.NP;
            if (P.cluster > 2) tr2_solve<WIDE>(P, P.LU, P.dvec_g, Xd, Wsc, P.tr2_g, member - 1, P.cluster - 1);  // (the helpers alone)
            else tr2_solve<WIDE>(P, P.LU, P.dvec_g, Xd, Wsc, P.tr2_g, member, P.cluster);
        }
        __syncthreads();
        if (tid == 0) {
            cluster_release(same_xcd);
            __hip_atomic_fetch_add(&P.ctl[2], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// LDS_LU: the LU factors live in LDS (N <= 112), else in global memory (L2)
// WIDE: 320 < N <= 640 (round 6; LDS_LU false): the twenty vectors and the solve vectors in global memory, LDS for the reductions
// and the one panel of the tiled Cholesky only; no pivoted LU (a Cholesky that fails ends the fit with LN_STATUS_NOT_SPD: the host
// takes the host-driven route, lognormal_wide.hip), no explicit inverse of a re-used Hessian
template <bool LDS_LU, bool WIDE = false>
__global__ __launch_bounds__(LT) void lognormal_kernel(LogNormalParams P) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ int s_fit;
    __shared__ int s_cl[2];  // cluster: command / flag hand-over inside the workgroup
    int cluster = 1;         // workgroups that share this fit's parallel pieces (1: this one alone)
    bool same_xcd = false;
    int slot = blockIdx.x;  // which set of work buffers (Sinv, LU, Hinv) this workgroup's fits use
    if (!LDS_LU && P.cluster > 1) {
        // workgroup ids go round the eight XCDs: id b sits on XCD b & 7 as the (b >> 3)-th of the launch there; the members of
        // group g are `cluster` consecutive ones of XCD g & 7 (a single fit: group 0 = ids 0, 8, 16, ..)
        const int x = blockIdx.x & 7, i = blockIdx.x >> 3;
        const int group = (i / P.cluster) * 8 + x, member = i % P.cluster;
        if (group >= (P.groups > 0 ? P.groups : 1)) return;
        slot = group;
        P.ctl += LN_CTL_WORDS * group;
        P.rk_g += (size_t)group * P.group_vec_stride;
        P.tr2_g += (size_t)group * P.group_vec_stride;
        P.dvec_g += (size_t)group * P.group_vec_stride;
        if (P.batch) {  // (the helpers work on the group's buffers, whatever fit its first workgroup is on)
            P.Sinv += (size_t)slot * P.N * P.N;
            P.LU += (size_t)slot * fh_ln_lu_doubles(P.N, P.NP);
            P.Hinv += (size_t)slot * P.N * P.N;
        }
        if (member > 0) {
            cluster_helper<WIDE>(P, member, s_cl);
            return;
        }
        // the helpers have 200 us to show up; otherwise this workgroup runs alone and they leave when they see the flag
        if (threadIdx.x == 0) {
            const long long t0 = wall_clock64();
            int ok = 1;
            while (ctl_load(&P.ctl[3]) < P.cluster - 1) {
                __builtin_amdgcn_s_sleep(2);
                if (wall_clock64() - t0 > 20000) {
                    ok = 0;
                    ctl_store(&P.ctl[4], 1);
                    break;
                }
            }
            const int xccs = ctl_load(&P.ctl[5]) | (1 << xcc_id());
            const int one = ok && (xccs & (xccs - 1)) == 0;
            ctl_store(&P.ctl[6], one);
            s_cl[0] = ok;
            s_cl[1] = one;
        }
        __syncthreads();
        cluster = s_cl[0] ? P.cluster : 1;
        same_xcd = s_cl[1] != 0;
        __syncthreads();
    }
    const LogNormalParams P0 = P;
    int cl_seq = 0;  // commands dispatched to this group's helpers so far (the words must never repeat: carried from fit to fit)
    int cl_epoch = 0;  // ... and the distributed factorisations (the helpers count them too)
    // batched launch: the workgroups pull fit indices from a counter; work buffers belong to the workgroup, outputs to
    // the fit (per-fit alpha, p0, band_lu)
    for (;;) {
    P = P0;
    if (P.batch) {
        if (threadIdx.x == 0) s_fit = atomicAdd(P.batch_counter, 1);
        __syncthreads();
        const int f = s_fit;
        __syncthreads();
        if (f >= P.batch) {
            if (cluster > 1) cluster_dispatch(P, LN_CMD_EXIT, same_xcd, cl_seq);  // (the helpers of a group leave with its last fit)
            return;
        }
        const int NN = P.N * P.N;
        P.alpha = P.batch_alpha[f];
        P.p0 = P.batch_p0[f];
        P.band_lu += (size_t)f * 5 * P.N;
        if (!(P.cluster > 1)) {  // (a group's buffers were selected above, once)
            P.Sinv += (size_t)slot * NN;
            P.LU += (size_t)slot * fh_ln_lu_doubles(P.N, P.NP);
            P.Hinv += (size_t)slot * NN;
        }
        if (P.resume) P.resume += (size_t)f * (3 * P.N + 1);
        P.H += (size_t)f * NN;
        P.s_out += (size_t)f * P.N;
        P.p_out += (size_t)f * P.N;
        P.result += 2 * f;
        P.stats += 17 * f;
    }
    const int N = P.N, tid = ln_tid(), w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    LnS S;
    {
        double *b = smem;
        S.red = b;  // 2 x 32 reduction partials + the pivot value
        b += 72;
        S.part = b;  // (fixed-size array first: part starts on a 16-byte boundary for every N)
        b += 2 * LT;
        double *const gscratch = ln_eval_parts(P) + 2 * (size_t)((N + EVW - 1) / EVW) * P.NP;  // (WIDE: 29 N + 64 doubles behind the partial sums)
        if constexpr (WIDE) {
            // LDS: the Cholesky's small arrays, rdiag and the permutation (written during a factorisation), then the other eighteen
            // vectors -- with the Cholesky's one panel over them; the solve vectors and the vectors' backup in global memory
            S.chol = b;
            b += fh_ln_chol_small_wide(P.NP) + (fh_ln_chol_small_wide(P.NP) & 1);
            S.rdiag = b;
            b += N;
            S.perm = reinterpret_cast<int *>(b);
            b += N;
            S.cpan = b;
            S.bak = gscratch + LNW * N;
            {   // behind the larger of the eighteen vectors and the panel that lies over them (fh_ln_smem_bytes)
                const size_t v18 = 18 * (size_t)N, pnl = (size_t)P.NP * tilechol::PS;
                S.stage = b + ((v18 > pnl ? v18 : pnl) + 1) / 2 * 2;
            }
        }
        double **vecs[] = {&S.x,  &S.xn, &S.I, &S.In,   &S.Sx,  &S.Sxn, &S.MI,  &S.MIn,  &S.jx,   &S.dx,
                           &S.pd, &S.jv, &S.p, &S.pold, &S.rhs, &S.tr2, &S.col, &S.rowk, &S.rdiag};
        for (auto v : vecs) {
            if (WIDE && v == &S.rdiag) continue;
            *v = b;
            b += N;
        }
        if constexpr (!WIDE) {
            S.perm = reinterpret_cast<int *>(b);
            b += N;  // 2N ints
        }
        S.wsol = WIDE ? gscratch : b;  // one solve vector per wave (72 + 2 LT + 20 N doubles in front: a 16-byte boundary for every N); the tiled
        b += LNW * N;  // Cholesky works in this space AND the panel's behind it (cholesky_as_lu, fh_ln_chol_doubles)
        if constexpr (!WIDE) S.chol = S.wsol;
        S.cluster = cluster;
        S.seq = cl_seq;
        S.chol_epoch = cl_epoch;
        S.same_xcd = same_xcd;
        S.s_cl = s_cl;
        S.lu = LDS_LU ? b : P.LU;
        S.lu_nb = LDS_LU ? 0 : P.lu_nb;  // blocked factorisation only for factors in global memory
        S.pan = b;                       // (global-LU kernels: the panel follows the int arrays)
    }
    // bands of the factorised T + I with the reciprocal pivots behind them, and the tables of the wave scans that solve with
    // them (band_scan.h), in this workgroup's scratch: formed once per fit, read by wave 0 once per pass
    double *const band_g = ln_band_scratch(P), *const scan_g = band_g + 6 * P.NP;
    if (P.band_lu) {
        for (int i = tid; i < 5 * N; i += LT) {
            const int bnd = i / N, c = i - bnd * N;
            double v = P.band_lu[i];
            if (c == 0 && bnd < 2) v = 0.0;  // no sub-diagonal entries in row 0
            band_g[i] = v;
            if (bnd == 2) band_g[5 * N + c] = 1.0 / v;
        }
        __syncthreads();
        if (tid < 64) bandscan::scan_tables<WIDE ? 10 : 6>(band_g, N, scan_g, tid);  // (6 rows per lane: N <= 320 < 384; ten: <= 640)
    }
    S.redsel = 0;
    S.nch = min(LT / N, N);
    S.row = -1;
    S.c0 = S.c1 = S.slot = 0;
    S.pair = (S.nch == 1 && (N & 1) == 0 && 2 * (LT / (N / 2)) * N <= LNW * N) ? 1 : 0;
    S.pbuf = S.part;
    S.pstride = LT;
    if (S.pair) {
        const int N2 = N / 2;
        S.nch = LT / N2;  // 3 at N = 300
        S.pbuf = S.wsol;
        S.pstride = S.nch * N;
        if (tid < S.nch * N2) {
            const int ch = tid / N2;
            S.row = 2 * (tid - ch * N2);
            S.c0 = ch * N / S.nch;
            S.c1 = (ch + 1) * N / S.nch;
            S.slot = ch * N + S.row;
        }
    } else if (tid < S.nch * N) {
        const int ch = tid / N;
        S.row = tid - ch * N;
        S.c0 = ch * N / S.nch;
        S.c1 = (ch + 1) * N / S.nch;
        S.slot = tid;
    }
    __shared__ long long s_tot[9];
    for (int i = tid; i < N; i += LT) {
        S.jv[i] = P.j[i];
        S.pold[i] = 0.0;  // radial_fitters.py:768
    }
    if (tid < 9) s_tot[tid] = 0;
#ifdef LN_TIMING
    if (tid < 12) ln_cyc[tid] = 0;
    if (tid < 4) ln_ev[tid] = 0;
    if (tid < 8) ln_ch[tid] = 0;
#endif
    __syncthreads();

    if (P.mode == LN_MODE_FIT) {
        // radial_fitters.py:756-761: s = log(max(MAP, 1e-3 MAP.max())) - s_scale; pI = max(transform(s)^2) (q/q0)^-4
        double mx = -INFINITY;
        for (int i = tid; i < N; i += LT) mx = fmax(mx, P.guess[i]);
        mx = block_minmax<true>(S, mx);
        for (int i = tid; i < N; i += LT) S.x[i] = log(fmax(P.guess[i], 1e-3 * mx)) - P.s0;
        __syncthreads();
        double best = -INFINITY;
        for (int r = w; r < N; r += LNW) {
            const double *yr = P.Y + r * N;
            double a = 0.0;
            for (int c = lane; c < N; c += 64) a = fma(yr[c], S.x[c], a);
            a = wave_sum(a) * P.pl_scale;
            best = fmax(best, a * a);
        }
        best = block_minmax<true>(S, best);
        for (int i = tid; i < N; i += LT) S.p[i] = best * pow(P.q[i] / P.q[0], -4.0);
    } else {
        for (int i = tid; i < N; i += LT) {
            S.x[i] = P.guess[i];
            S.p[i] = P.p_in[i];
        }
    }
    const bool resumed = P.mode == LN_MODE_FIT && P.resume != nullptr;
    if (resumed) {  // a paused fit: it stopped behind an update of p
        __syncthreads();
        for (int i = tid; i < N; i += LT) {
            S.x[i] = P.resume[i];
            S.p[i] = P.resume[N + i];
            S.pold[i] = P.resume[2 * N + i];
        }
    }
    __syncthreads();

    int status = LN_STATUS_OK, count = resumed ? (int)P.resume[3 * N] : 0;
    bool in_pass = resumed;
    for (;;) {
        if (P.mode != LN_MODE_UPDATE) {
            // ---- LogNormalMAPModel(DHT, M, j, p, guess=s, s0)  (statistical_models.py:1012-1160) ----
            int badp = 0;
            for (int i = tid; i < N; i += LT) badp |= !(S.p[i] > 0.0);  // :1049
            if (__syncthreads_or(badp)) {
                status = LN_STATUS_BAD_P;
                break;
            }
            for (int i = tid; i < N; i += LT) S.rhs[i] = 1 / S.p[i];
            __syncthreads();
#ifdef LN_TIMING
            const long long _ts = clock64();
#endif
            if (S.cluster > 1) {
                for (int i = tid; i < N; i += LT) P.rk_g[i] = S.rhs[i];
                cluster_dispatch(P, LN_CMD_SINV, same_xcd, S.seq);
                build_sinv(P, S.rhs, 0, S.cluster);
                if (!cluster_wait(P, s_cl, same_xcd)) {
                    status = LN_STATUS_CLUSTER;
                    break;
                }
            } else {
                build_sinv(P, S.rhs);
            }
            __syncthreads();
#ifdef LN_TIMING
            if (tid == 0) ln_cyc[8] += clock64() - _ts;
            const long long _tn = clock64();
#endif
            const NewtonExit ex = minimize_newton<WIDE>(P, S);
#ifdef LN_TIMING
            if (tid == 0) ln_cyc[4] += clock64() - _tn;
#endif
            if (tid == 0) {
                s_tot[0] += 1;
                s_tot[1] += ex.nstep;
                s_tot[2] += ex.nfev;
                s_tot[3] += ex.nhess;
                s_tot[4 + min(max(ex.status, 0), 4)] += 1;
            }
            if (ex.status == 4) {
                status = LN_STATUS_SLOPE;
                break;
            }
            if (ex.status == 5) {  // (WIDE: a Hessian the tiled Cholesky could not factor)
                status = LN_STATUS_NOT_SPD;
                break;
            }
            // Dinv = hess(s_MAP) (:1147), row-major for the host
            // (a whole fit hands out the Dinv of its last pass only: built behind the loop; 72 k cycles of every pass)
            if (P.mode != LN_MODE_FIT) build_hess(P, S, P.H, N, true);
        }
        if (P.mode == LN_MODE_MAP) break;
        // Factors of Dinv for Tr2 (filter.py:168-170 solves with the model's cho_factor / SVD fallback,
        // statistical_models.py:1147-1158): tiled Cholesky first -- Dinv is positive definite at a MAP -- and the
        // pivoted LU when a pivot is not positive.
        // (P.H of a caller-supplied posterior is row-major Dinv; the Hessian of this kernel is rebuilt in the layout
        //  each factorisation wants)
        bool chol = false;
        double *const Cp = P.LU + N * N, *const Wsc = Cp + P.NP * P.NP, *const Xd = Wsc + P.NP * P.NP;
        if (S.lu_nb > 0 && !P.no_cholesky) {
            if (P.mode != LN_MODE_UPDATE) {
                build_hess_padded(P, S, Cp);
            } else {  // (a caller's posterior: every element of the padded matrix, packed)
                const int nbk = P.NP / 16;
                for (int b = tid >> 5; b < P.NP; b += LT / 32)
                    for (int a = tid & 31; a < P.NP; a += 32)
                        Cp[pk_elem(b, a, nbk)] = (a < N && b < N) ? P.H[a * N + b] : (a == b ? 1.0 : 0.0);
                __syncthreads();
            }
            chol = cholesky_as_lu<WIDE>(P, S, Cp, Xd);
            if (WIDE && !chol) {
                status = LN_STATUS_NOT_SPD;
                break;
            }
        }
        if (!chol) {
            if (P.mode != LN_MODE_UPDATE) {
                build_hess(P, S, S.lu, N, false);
            } else {
                for (int b = tid >> 5; b < N; b += LT / 32)
                    for (int a = tid & 31; a < N; a += 32) S.lu[b * N + a] = P.H[a * N + b];
                __syncthreads();
            }
            if (S.lu_nb > 0) lu_factor_blocked(S, N, S.lu);
            else lu_factor(S, N, S.lu);
        }
        if (in_pass) {  // radial_fitters.py:781-785
            if (P.diag_p)
                for (int i = tid; i < N; i += LT) {
                    P.diag_p[count * N + i] = S.p[i];
                    P.diag_s[count * N + i] = S.x[i];
                }
            ++count;
        }
        int bad = 0;
        for (int i = tid; i < N; i += LT) bad |= !(fabs(S.p[i] - S.pold[i]) <= P.tol * S.p[i]);  // filter.py:181
        bad = __syncthreads_or(bad);
        if (P.mode == LN_MODE_FIT && (!bad || count > P.max_iter)) break;  // radial_fitters.py:769-770
        // ---- CriticalFilter.update_power_spectrum(fit)  (filter.py:154-177) ----
#ifdef LN_TIMING
        long long _tu = clock64();
#endif
        // (Y s)_r into S.rhs: a wave takes rows w, w + 8, ...; the loads of four rows are issued together (a row at a time the
        // wave waited for one L2 round trip per row), the sums are formed in the same order
        auto tr1_sums = [&]() {
            constexpr int RB = WIDE ? 2 : 4, CB = WIDE ? 10 : 5;  // N <= 320: five column chunks of 64; WIDE: ten
            double xs[CB];
#pragma unroll
            for (int cc = 0; cc < CB; ++cc) xs[cc] = (lane + 64 * cc < N) ? S.x[lane + 64 * cc] : 0.0;
            for (int r0 = w; r0 < N; r0 += RB * LNW) {
                double y[RB][CB];
#pragma unroll
                for (int k = 0; k < RB; ++k) {
                    const int r = min(r0 + k * LNW, N - 1);
                    const double *yr = P.Y + r * N;
#pragma unroll
                    for (int cc = 0; cc < CB; ++cc) y[k][cc] = yr[min(lane + 64 * cc, N - 1)];
                }
#pragma unroll
                for (int k = 0; k < RB; ++k) {
                    const int r = r0 + k * LNW;
                    double a = 0.0;
#pragma unroll
                    for (int cc = 0; cc < CB; ++cc)
                        if (lane + 64 * cc < N) a = fma(y[k][cc], xs[cc], a);
                    a = wave_sum(a);
                    if (lane == 0 && r < N) S.rhs[r] = a;
                }
            }
        };
        const bool tr1_early = chol && S.cluster > 2;  // formed while the helpers solve for Tr2
        if (chol) {
            const double *dvec = chol_dvec<WIDE>(S, P.NP);  // diag(L), left in LDS by cholesky_as_lu
            if (S.cluster > 1) {
                for (int i = tid; i < P.NP; i += LT) P.dvec_g[i] = dvec[i];
                cluster_dispatch(P, LN_CMD_TR2, same_xcd, S.seq);
                if (S.cluster > 2) tr1_sums();  // (three helpers and more take the Tr2 columns among themselves)
                else tr2_solve<WIDE>(P, S.lu, dvec, Xd, Wsc, P.tr2_g, 0, S.cluster);
                if (!cluster_wait(P, s_cl, same_xcd)) {
                    status = LN_STATUS_CLUSTER;
                    break;
                }
                for (int i = tid; i < N; i += LT) S.tr2[i] = P.tr2_g[i];
            } else {
                tr2_solve<WIDE>(P, S.lu, dvec, Xd, Wsc, S.tr2);
            }
            __syncthreads();
        }
#ifdef LN_TIMING
        if (tid == 0) ln_cyc[9] += clock64() - _tu;
        _tu = clock64();
#endif
        // Tr1_r = (Y s)_r^2
        if (chol) {
            if (!tr1_early) tr1_sums();
            __syncthreads();
            // (the division and the logarithm of a row by all threads at once: on lane 0 of the row's wave, row after row,
            //  they were most of this phase -- 150 k cycles per pass)
            for (int r = tid; r < N; r += LT) {
                const double a = S.rhs[r], pi = S.p[r];
                const double beta = (P.p0 + 0.5 * (a * a + S.tr2[r])) / pi - (P.alpha - 1.0 + 0.5 * 1.0);
                S.rhs[r] = beta + log(pi);
            }
        }
        for (int r = w; r < N && !chol; r += LNW) {
            const double *yr = P.Y + r * N;
            double a = 0.0;  // Tr1_r = (Y s)_r^2
            for (int c = lane; c < N; c += 64) a = fma(yr[c], S.x[c], a);
            a = wave_sum(a);
            double t2;       // Tr2_r = y_r . D y_r,  D = Dinv^-1
            if (chol) {
                t2 = S.tr2[r];
            } else {
                double *xs = S.wsol + w * N;
                wave_solve(S, N, S.lu, yr, -1, 1.0, xs);
                t2 = 0.0;
                for (int c = lane; c < N; c += 64) t2 = fma(yr[c], xs[c], t2);
                t2 = wave_sum(t2);
            }
            if (lane == 0) {
                const double pi = S.p[r];
                const double beta = (P.p0 + 0.5 * (a * a + t2)) / pi - (P.alpha - 1.0 + 0.5 * 1.0);
                S.rhs[r] = beta + log(pi);
            }
        }
        __syncthreads();
        for (int i = tid; i < N; i += LT) S.pold[i] = S.p[i];
#ifdef LN_TIMING
        if (tid == 0) ln_cyc[10] += clock64() - _tu;
        _tu = clock64();
#endif
        if (tid < 64) {  // (T + I) tau = beta + log p with the host-factorised bands: both substitutions as wave scans of wave 0
                         // (one thread took 2 N dependent steps with its operands in L2: 70 us of a 1.2 ms pass at N = 300)
            int t = tid;
            asm volatile("" : "+v"(t));
            bandscan::scan_solve<WIDE ? 10 : 6>(band_g, N, scan_g, S.rhs, 0, t);
            bandscan::scan_solve<WIDE ? 10 : 6>(band_g, N, scan_g, S.rhs, 1, t);
        }
        __syncthreads();
        for (int i = tid; i < N; i += LT) S.p[i] = exp(S.rhs[i]);  // filter.py:177
        __syncthreads();
#ifdef LN_TIMING
        if (tid == 0) ln_cyc[11] += clock64() - _tu;
#endif
        if (P.mode == LN_MODE_UPDATE) break;
        in_pass = true;
        // a staged sweep: pause here -- (s, p, p_old, count) is the whole state of the iteration -- once every fit of the batch has
        // been handed out and only a few have not ended: those continue on clusters of workgroups (capi_lognormal.hip)
        if (P.pause_when_left > 0 && P.batch && (count & 15) == 0) {
            if (tid == 0)
                s_fit = __hip_atomic_load(P.batch_counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= P.batch &&
                        P.batch - __hip_atomic_load(P.done_counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= P.pause_when_left;
            __syncthreads();
            const int stop = s_fit;
            __syncthreads();
            if (stop) {
                status = LN_STATUS_PAUSED;
                break;
            }
        }
    }
    if (P.mode == LN_MODE_FIT && status == LN_STATUS_OK) build_hess(P, S, P.H, N, true);  // Dinv = hess(s_MAP) (:1147), row-major
    if (status == LN_STATUS_PAUSED)
        for (int i = tid; i < N; i += LT) P.H[i] = S.pold[i];  // (the power spectrum before the last update travels in the fit's H)

    for (int i = tid; i < N; i += LT) {
        P.s_out[i] = S.x[i];
        P.p_out[i] = S.p[i];
    }
    if (tid == 0) {
        P.result[0] = count;
        P.result[1] = status;
        for (int k = 0; k < 9; ++k) P.stats[k] = s_tot[k];
#ifdef LN_TIMING
        for (int k = 0; k < 8; ++k) P.stats[9 + k] = ln_cyc[k];
        printf("[ln timing, Mcycles] S^-1 %.1f  Tr2 %.1f  Tr1 %.1f  bands + exp %.1f\n", ln_cyc[8] / 1e6, ln_cyc[9] / 1e6, ln_cyc[10] / 1e6,
               ln_cyc[11] / 1e6);
        printf("[ln timing, Mcycles] inside the Cholesky: chain to its flag %.1f, its outputs %.1f, at the barrier %.1f; worker 1: trailing tiles %.1f, column tiles %.1f, waiting for the flag %.1f, panel %.1f, at the barrier %.1f\n",
               ln_ch[0] / 1e6, ln_ch[1] / 1e6, ln_ch[2] / 1e6, ln_ch[3] / 1e6, ln_ch[4] / 1e6, ln_ch[5] / 1e6, ln_ch[6] / 1e6, ln_ch[7] / 1e6);
        printf("[ln timing, Mcycles] inside the evaluations: command out %.1f  own items %.1f  waiting for the helpers %.1f  combining %.1f\n",
               ln_ev[0] / 1e6, ln_ev[1] / 1e6, ln_ev[2] / 1e6, ln_ev[3] / 1e6);
#endif
    }
    if (tid == 0 && P.done_counter && status != LN_STATUS_PAUSED) atomicAdd(P.done_counter, 1);
    if (!P.batch) {
        if (S.cluster > 1) cluster_dispatch(P, LN_CMD_EXIT, same_xcd, S.seq);
        return;
    }
    cluster = S.cluster;  // (a cluster that was disbanded stays so)
    cl_seq = S.seq;
    cl_epoch = S.chol_epoch;
    __syncthreads();
    }  // next fit of the batch
}

}  // namespace

// lu_nb: panel width of the blocked LU (0: factors in LDS, unblocked); N = 320 with a 32-column panel takes 159 KB
size_t fh_ln_smem_bytes(int N, int *lu_in_lds, int *lu_nb) {
    if (N > 320) {  // WIDE kernel: the reductions' partials and the Cholesky's one panel
        if (lu_in_lds) *lu_in_lds = 0;
        if (lu_nb) *lu_nb = LU_NB;
        const int NPw = (N + 15) / 16 * 16;
        const size_t small = (size_t)fh_ln_chol_small_wide(NPw) + (fh_ln_chol_small_wide(NPw) & 1);
        const size_t vec18 = 18 * (size_t)N, panel = (size_t)NPw * tilechol::PS;
        return sizeof(double) * (72 + 2 * LT + small + 2 * (size_t)N + ((vec18 > panel ? vec18 : panel) + 1) / 2 * 2 + 64 * 64);  // (+ the staged diagonal block)
    }
    size_t doubles = (19 + LNW) * N + 72 + 2 * LT + N;
    const int fits = (N <= 112);
    if (lu_in_lds) *lu_in_lds = fits;
    int nb = 0;
    if (fits) {
        doubles += N * N;
    } else {
        nb = LU_NB;
        doubles += (size_t)N * nb;
        // the tiled Cholesky works in the solve vectors' space and the panel's behind it
        const size_t chol = (size_t)fh_ln_chol_doubles((N + 15) / 16 * 16), have = (size_t)(LNW + nb) * N;
        if (chol > have) doubles += chol - have;
    }
    if (lu_nb) *lu_nb = nb;
    return doubles * sizeof(double);
}

hipError_t fh_ln_launch(const LogNormalParams &P0, int nblocks, hipStream_t s) {
    LogNormalParams P = P0;
    const size_t smem = fh_ln_smem_bytes(P.N, &P.lu_in_lds, &P.lu_nb);
    using Kernel = void (*)(LogNormalParams);
    Kernel fn = P.lu_in_lds ? lognormal_kernel<true, false> : (P.N > 320 ? lognormal_kernel<false, true> : lognormal_kernel<false, false>);
    if (P.N > 640) return hipErrorInvalidValue;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)smem);
    if (e != hipSuccess) return e;
    // clusters: group g of a launch = `cluster` workgroup ids of XCD g % 8 (factors in global memory only); a single fit is one
    // group, a staged sweep's second stage up to 256 / cluster of them (P.groups)
    int grid = nblocks;
    if (P.cluster > 1 && !P.lu_in_lds && (!P.batch || P.groups > 0)) {
        if (P.groups < 1) P.groups = 1;
        grid = P.groups == 1 ? 8 * (P.cluster - 1) + 1 : 8 * P.cluster * ((P.groups + 7) / 8);
    } else {
        P.cluster = 1;
        P.groups = 0;
    }
    fn<<<grid, LT, smem, s>>>(P);
    return hipGetLastError();
}
