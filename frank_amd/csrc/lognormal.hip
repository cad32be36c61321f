// lognormal.hip -- method='LogNormal': the MAP of the log-brightness by Newton's method with back-tracking, and the
// power-spectrum iteration around it, one persistent workgroup per fit.
//
//   LogNormalMAPModel._fit        frank/statistical_models.py:1064-1160   (H, jac, hess, limit_step)
//   MinimizeNewton / LineSearch   frank/minimizer.py:45-283
//   FrankFitter._fit (LogNormal)  frank/radial_fitters.py:754-785
//   CriticalFilter.update_power_spectrum  frank/filter.py:154-177
//
// The algorithm is a serial chain of small dense operations (N <= 320): ~1e5 Newton steps, ~1e6 function evaluations
// and ~1e4 LU factorisations per fit, each depending on the last.  Nothing here is bandwidth- or MFMA-bound; the
// cost is latency, so the whole chain lives in ONE workgroup: state in LDS, the LU factors in LDS when they fit
// (N <= 112) or in L2 otherwise, every reduction in a fixed order so that all lanes take the same branch, no host
// round trip until the fit is done.  Throughput comes from running independent fits (sweeps, bootstraps) on the
// other 255 CUs, not from splitting one fit.
//
// Differences from the reference that do not change the mathematics: scipy's lu_factor (LAPACK getrf) is an
// unblocked partial-pivoting LU here; the posterior covariance D = hess(s_MAP)^-1 is applied through that LU
// instead of a Cholesky factor (the reference falls back to an SVD inverse when the Cholesky fails,
// statistical_models.py:1150-1158 -- the same matrix); jac(x) re-uses the products of the accepted fun(x).
#include <hip/hip_runtime.h>

#include "kernels.h"

#pragma clang fp contract(off)

namespace {

constexpr int LT = 512;        // threads per workgroup
constexpr int LNW = LT / 64;   // waves
constexpr int LN_TMAX = 5;     // 64-row register slabs of a solve vector (N <= 320)
constexpr int LS_MAX_TRIALS = 2000;  // back-tracking guard: lam shrinks >= 10x per trial, x + lam p == x long before

struct LnS {
    double *x, *xn, *I, *In, *Sx, *Sxn, *MI, *MIn, *jx, *dx, *pd, *jv, *p, *pold, *rhs, *tr2, *col, *rowk, *rdiag, *red;
    int *perm, *ipiv;
    double *lu;  // N*N column-major: LDS or global
};

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// Block reductions: every thread returns the same bits (partials combined in wave order by every thread).
__device__ double block_sum(LnS &S, double v) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) S.red[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = 0.0;
#pragma unroll
    for (int w = 0; w < LNW; ++w) r += S.red[w];
    __syncthreads();
    return r;
}

__device__ void block_sum3(LnS &S, double &a, double &b, double &c) {
    a = wave_sum(a);
    b = wave_sum(b);
    c = wave_sum(c);
    if ((threadIdx.x & 63) == 0) {
        const int w = threadIdx.x >> 6;
        S.red[3 * w] = a;
        S.red[3 * w + 1] = b;
        S.red[3 * w + 2] = c;
    }
    __syncthreads();
    double ra = 0.0, rb = 0.0, rc = 0.0;
#pragma unroll
    for (int w = 0; w < LNW; ++w) {
        ra += S.red[3 * w];
        rb += S.red[3 * w + 1];
        rc += S.red[3 * w + 2];
    }
    __syncthreads();
    a = ra;
    b = rb;
    c = rc;
}

template <bool IS_MAX>
__device__ double block_minmax(LnS &S, double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const double o = __shfl_xor(v, off);
        v = IS_MAX ? fmax(v, o) : fmin(v, o);
    }
    if ((threadIdx.x & 63) == 0) S.red[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = S.red[0];
#pragma unroll
    for (int w = 1; w < LNW; ++w) r = IS_MAX ? fmax(r, S.red[w]) : fmin(r, S.red[w]);
    __syncthreads();
    return r;
}

// H(s) = 1/2 s^T S^-1 s + 1/2 I^T M I - j^T I,  I = exp(s + s0)   (statistical_models.py:1075-1085)
// leaves I, S^-1 s and M I of the point in Iv, Sxv, MIv (the gradient and the Hessian diagonal re-use them).
__device__ double ln_eval(const LogNormalParams &P, LnS &S, const double *xv, double *Iv, double *Sxv, double *MIv) {
    const int N = P.N, tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    for (int i = tid; i < N; i += LT) Iv[i] = exp(xv[i] + P.s0);
    __syncthreads();
    for (int r = w; r < N; r += LNW) {
        const double *sr = P.Sinv + (size_t)r * N, *mr = P.M + (size_t)r * N;
        double a = 0.0, b = 0.0;
        for (int c = l; c < N; c += 64) {
            a = fma(sr[c], xv[c], a);
            b = fma(mr[c], Iv[c], b);
        }
        a = wave_sum(a);
        b = wave_sum(b);
        if (l == 0) {
            Sxv[r] = a;
            MIv[r] = b;
        }
    }
    __syncthreads();
    double A = 0.0, B = 0.0, C = 0.0;
    for (int i = tid; i < N; i += LT) {
        A += xv[i] * Sxv[i];
        B += Iv[i] * MIv[i];
        C += Iv[i] * S.jv[i];
    }
    block_sum3(S, A, B, C);
    double f = 0.5 * A;
    f += 0.5 * B;
    f -= C;
    return f;
}

// jac(s) = S^-1 s + (I (M I) - I j)   (statistical_models.py:1087-1098), from the cached products of S.x
__device__ __forceinline__ double ln_grad(const LnS &S, int i) {
    return S.Sx[i] + (S.I[i] * S.MI[i] - S.I[i] * S.jv[i]);
}

// hess(s) = I_a M_ab I_b + delta_ab (I_a (M I)_a - I_a j_a) + S^-1_ab   (statistical_models.py:1100-1122), at S.x,
// written column-major into A (and into `copy` when given).  M is exactly symmetric, S^-1 to round-off.
__device__ void build_hess(const LogNormalParams &P, LnS &S, double *A, double *copy) {
    const int N = P.N, tid = threadIdx.x;
    for (int b = tid >> 5; b < N; b += LT / 32) {
        const double Ib = S.I[b];
        const double *mb = P.M + (size_t)b * N, *sb = P.Sinv + (size_t)b * N;
        for (int a = tid & 31; a < N; a += 32) {
            double v = S.I[a] * mb[a] * Ib;
            if (a == b) v += S.I[a] * S.MI[a] - S.I[a] * S.jv[a];
            v += sb[a];
            A[(size_t)b * N + a] = v;
            if (copy) copy[(size_t)a * N + b] = v;  // row-major H_ab
        }
    }
    __syncthreads();
}

// Partial-pivoting LU in place (column-major, unit lower), three barriers per column; perm[i] = source row of row i,
// rdiag[i] = 1 / U_ii.  A zero pivot leaves the column unscaled (LAPACK getf2 does the same and reports it).
__device__ void lu_factor(LnS &S, int N, double *A) {
    const int tid = threadIdx.x;
    for (int k = 0; k < N; ++k) {
        if (tid < 64) {  // first maximum of |A[k:, k]|
            double best = -1.0;
            int bi = k;
            for (int i = k + tid; i < N; i += 64) {
                const double v = fabs(A[(size_t)k * N + i]);
                if (v > best) {
                    best = v;
                    bi = i;
                }
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                const double ob = __shfl_xor(best, off);
                const int oi = __shfl_xor(bi, off);
                if (ob > best || (ob == best && oi < bi)) {
                    best = ob;
                    bi = oi;
                }
            }
            if (tid == 0) {
                S.ipiv[k] = bi;
                S.red[0] = A[(size_t)k * N + bi];
            }
        }
        __syncthreads();
        const int piv = S.ipiv[k];
        const double pv = S.red[0];
        // swap rows k <-> piv outside column k, stage the new row k and the scaled column k
        for (int j = tid; j < N; j += LT) {
            if (j == k) continue;
            const double a = A[(size_t)j * N + k], b = A[(size_t)j * N + piv];
            if (piv != k) {
                A[(size_t)j * N + k] = b;
                A[(size_t)j * N + piv] = a;
            }
            if (j > k) S.rowk[j] = b;
        }
        for (int i = k + 1 + tid; i < N; i += LT) {
            const double v = (i == piv) ? A[(size_t)k * N + k] : A[(size_t)k * N + i];
            const double lv = (pv != 0.0) ? v / pv : v;
            A[(size_t)k * N + i] = lv;
            S.col[i] = (pv != 0.0) ? lv : 0.0;
            if (i == piv) A[(size_t)k * N + k] = pv;
        }
        if (tid == 0) S.rdiag[k] = 1.0 / pv;
        __syncthreads();
        // trailing update A[i, j] -= l_i u_j
        for (int j = k + 1 + (tid >> 5); j < N; j += LT / 32) {
            const double uj = S.rowk[j];
            double *cj = A + (size_t)j * N;
            for (int i = k + 1 + (tid & 31); i < N; i += 32) cj[i] = fma(-S.col[i], uj, cj[i]);
        }
        __syncthreads();
    }
    if (tid == 0) {
        for (int i = 0; i < N; ++i) S.perm[i] = i;
        for (int k = 0; k < N; ++k) {
            const int pk = S.ipiv[k];
            if (pk != k) {
                const int t = S.perm[k];
                S.perm[k] = S.perm[pk];
                S.perm[pk] = t;
            }
        }
    }
    __syncthreads();
}

// One wave solves (P L U) x = sign * b; lane l keeps x[l], x[64 + l], ... in registers.
__device__ void lu_solve_regs(const LnS &S, int N, const double *A, const double *b, double sign, double (&x)[LN_TMAX]) {
    const int lane = threadIdx.x & 63;
    const int T = (N + 63) >> 6;
#pragma unroll
    for (int t = 0; t < LN_TMAX; ++t) {
        const int jj = 64 * t + lane;
        x[t] = (jj < N) ? sign * b[S.perm[jj]] : 0.0;
    }
#pragma unroll
    for (int t = 0; t < LN_TMAX; ++t) {  // L y = b (unit lower), column by column
        if (t >= T) break;
        const int lim = min(64, N - 64 * t);
        for (int l = 0; l < lim; ++l) {
            const int i = 64 * t + l;
            const double xi = __shfl(x[t], l);
            const double *ci = A + (size_t)i * N;
#pragma unroll
            for (int t2 = t; t2 < LN_TMAX; ++t2) {
                const int jj = 64 * t2 + lane;
                if (t2 < T && jj > i && jj < N) x[t2] = fma(-ci[jj], xi, x[t2]);
            }
        }
    }
#pragma unroll
    for (int t = LN_TMAX - 1; t >= 0; --t) {  // U x = y
        if (t >= T) continue;
        const int lim = min(64, N - 64 * t);
        for (int l = lim - 1; l >= 0; --l) {
            const int i = 64 * t + l;
            const double xi = __shfl(x[t], l) * S.rdiag[i];
            if (lane == l) x[t] = xi;
            const double *ci = A + (size_t)i * N;
#pragma unroll
            for (int t2 = 0; t2 <= t; ++t2) {
                const int jj = 64 * t2 + lane;
                if (jj < i) x[t2] = fma(-ci[jj], xi, x[t2]);
            }
        }
    }
}

// limit_step (statistical_models.py:1126-1130) into S.pd, and delta_f = jac . p (minimizer.py:131-134)
__device__ double limited_step(LnS &S, int N, const double *dir) {
    const int tid = threadIdx.x;
    double am = INFINITY;
    for (int i = tid; i < N; i += LT) am = fmin(am, fabs(S.x[i] / dir[i]));
    am = block_minmax<false>(S, am);
    double alpha = 1.1 * am;
    if (1.0 < alpha) alpha = 1.0;
    double df = 0.0;
    for (int i = tid; i < N; i += LT) {
        const double pi = alpha * dir[i];
        S.pd[i] = pi;
        df += S.jx[i] * pi;
    }
    return block_sum(S, df);
}

__device__ void accept_trial(LnS &S, int N) {
    for (int i = threadIdx.x; i < N; i += LT) {
        S.x[i] = S.xn[i];
        S.I[i] = S.In[i];
        S.Sx[i] = S.Sxn[i];
        S.MI[i] = S.MIn[i];
    }
    __syncthreads();
}

// LineSearch.__call__(fun, jac, x, dir, fx, root=False)  (minimizer.py:70-187).
// returns 0 accepted (x, fx, reduction updated), 1 failed, -1 "Round off in slope calculation".
__device__ int line_search(const LogNormalParams &P, LnS &S, const double *dir, double &fx, int &nfev, double &reduction) {
    const int N = P.N, tid = threadIdx.x;
    const double armijo = 1e-4, l_min = 0.1;
    const double cost = fx;
    const double delta_f = limited_step(S, N, dir);
    if (delta_f > 0) return -1;
    double lam = 1.0, cost_save = 0.0, lam_save = 0.0;
    for (int trial = 0; trial < LS_MAX_TRIALS; ++trial) {
        int moved = 0;
        for (int i = tid; i < N; i += LT) {
            const double xn = S.x[i] + lam * S.pd[i];
            S.xn[i] = xn;
            moved |= (xn != S.x[i]);
        }
        if (!__syncthreads_or(moved)) return 1;
        const double cost_new = ln_eval(P, S, S.xn, S.In, S.Sxn, S.MIn);
        ++nfev;
        if (cost_new <= (cost + armijo * lam * delta_f)) {
            reduction = lam;
            accept_trial(S, N);
            fx = cost_new;
            return 0;
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
__device__ NewtonExit minimize_newton(const LogNormalParams &P, LnS &S) {
    const int N = P.N, tid = threadIdx.x;
    bool need_hess = true;
    int nfev = 1, nhess = 0;
    double reduction = NAN;  // LineSearch.reduction starts as None
    double fx = ln_eval(P, S, S.x, S.I, S.Sx, S.MI);
    for (int nstep = 0; nstep < P.max_step; ++nstep) {
        if (need_hess) {
            if (nhess == P.max_hev) return {3, nstep, nfev, nhess};
            build_hess(P, S, S.lu, nullptr);
            lu_factor(S, N, S.lu);
            ++nhess;
        }
        for (int i = tid; i < N; i += LT) S.jx[i] = ln_grad(S, i);
        __syncthreads();
        if (tid < 64) {
            double xr[LN_TMAX];
            lu_solve_regs(S, N, S.lu, S.jx, -1.0, xr);
#pragma unroll
            for (int t = 0; t < LN_TMAX; ++t)
                if (64 * t + tid < N) S.dx[64 * t + tid] = xr[t];
        }
        __syncthreads();
        double d = 0.0;
        for (int i = tid; i < N; i += LT) d += S.jx[i] * S.dx[i];
        d = block_sum(S, d);
        int failed;
        if (d < 0) {
            failed = line_search(P, S, S.dx, fx, nfev, reduction);
            if (failed < 0) return {4, nstep, nfev, nhess};
        } else {
            failed = 1;
        }
        if (failed) {  // gradient descent when Newton's direction does not improve (minimizer.py:249-271)
            for (int i = tid; i < N; i += LT) S.dx[i] = -S.jx[i];
            __syncthreads();
            const int failed_descent = line_search(P, S, S.dx, fx, nfev, reduction);
            if (failed_descent < 0) return {4, nstep, nfev, nhess};
            if (failed_descent) {
                (void)limited_step(S, N, S.dx);
                bool improved = false;
                double fn = fx;
                for (int it = 0; it < 10; ++it) {
                    for (int i = tid; i < N; i += LT) S.xn[i] = S.x[i] + S.pd[i];
                    __syncthreads();
                    fn = ln_eval(P, S, S.xn, S.In, S.Sxn, S.MIn);
                    ++nfev;
                    if (fn < fx) {
                        improved = true;
                        break;
                    }
                    for (int i = tid; i < N; i += LT) S.pd[i] *= 0.0625;
                    __syncthreads();
                }
                if (!improved) return {1, nstep, nfev, nhess};
                fx = fn;
                accept_trial(S, N);
            }
        }
        need_hess = failed || (reduction != 1.0);
        double g = -INFINITY;
        for (int i = tid; i < N; i += LT) g = fmax(g, fabs(ln_grad(S, i)) * fabs(S.x[i]));
        g = block_minmax<true>(S, g);
        if (g < P.newton_tol * fmax(fabs(fx), 1.0)) return {0, nstep, nfev, nhess};
    }
    return {2, P.max_step - 1, nfev, nhess};
}

__global__ __launch_bounds__(LT) void lognormal_kernel(LogNormalParams P) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int N = P.N, tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    LnS S;
    {
        double *b = smem;
        double **vecs[] = {&S.x,  &S.xn, &S.I, &S.In,   &S.Sx,  &S.Sxn, &S.MI,  &S.MIn,  &S.jx,   &S.dx,
                           &S.pd, &S.jv, &S.p, &S.pold, &S.rhs, &S.tr2, &S.col, &S.rowk, &S.rdiag};
        for (auto v : vecs) {
            *v = b;
            b += N;
        }
        S.red = b;
        b += 64;
        S.perm = reinterpret_cast<int *>(b);
        S.ipiv = S.perm + N;
        b += N;  // 2N ints
        S.lu = P.lu_in_lds ? b : P.LU;
    }
    __shared__ long long s_tot[9];
    for (int i = tid; i < N; i += LT) {
        S.jv[i] = P.j[i];
        S.pold[i] = 0.0;  // radial_fitters.py:768
    }
    if (tid < 9) s_tot[tid] = 0;
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
            const double *yr = P.Y + (size_t)r * N;
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
    __syncthreads();

    int status = LN_STATUS_OK, count = 0;
    bool in_pass = false;
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
            for (int a = tid >> 5; a < N; a += LT / 32)  // S^-1 = Y^T diag(1/p) Y  (:1061)
                for (int b = tid & 31; b < N; b += 32) {
                    double acc = 0.0;
                    for (int k = 0; k < N; ++k) acc += (P.Y[(size_t)k * N + a] * S.rhs[k]) * P.Y[(size_t)k * N + b];
                    P.Sinv[(size_t)a * N + b] = acc;
                }
            __syncthreads();
            const NewtonExit ex = minimize_newton(P, S);
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
            // Dinv = hess(s_MAP) (:1147); its LU stands in for cho_factor / the SVD fallback
            build_hess(P, S, S.lu, P.H);
        } else {
            // a caller-supplied posterior: s_MAP in S.x, Dinv (row-major) in P.H
            for (int b = tid >> 5; b < N; b += LT / 32)
                for (int a = tid & 31; a < N; a += 32) S.lu[(size_t)b * N + a] = P.H[(size_t)a * N + b];
            __syncthreads();
        }
        lu_factor(S, N, S.lu);
        if (P.mode == LN_MODE_MAP) break;
        if (in_pass) {  // radial_fitters.py:781-785
            if (P.diag_p)
                for (int i = tid; i < N; i += LT) {
                    P.diag_p[(size_t)count * N + i] = S.p[i];
                    P.diag_s[(size_t)count * N + i] = S.x[i];
                }
            ++count;
        }
        int bad = 0;
        for (int i = tid; i < N; i += LT) bad |= !(fabs(S.p[i] - S.pold[i]) <= P.tol * S.p[i]);  // filter.py:181
        bad = __syncthreads_or(bad);
        if (P.mode == LN_MODE_FIT && (!bad || count > P.max_iter)) break;  // radial_fitters.py:769-770
        // ---- CriticalFilter.update_power_spectrum(fit)  (filter.py:154-177) ----
        for (int r = w; r < N; r += LNW) {
            const double *yr = P.Y + (size_t)r * N;
            double a = 0.0;  // Tr1_r = (Y s)_r^2
            for (int c = lane; c < N; c += 64) a = fma(yr[c], S.x[c], a);
            a = wave_sum(a);
            double xr[LN_TMAX];  // Tr2_r = y_r . D y_r,  D = Dinv^-1
            lu_solve_regs(S, N, S.lu, yr, 1.0, xr);
            double t2 = 0.0;
#pragma unroll
            for (int t = 0; t < LN_TMAX; ++t)
                if (64 * t + lane < N) t2 = fma(yr[64 * t + lane], xr[t], t2);
            t2 = wave_sum(t2);
            if (lane == 0) {
                const double pi = S.p[r];
                const double beta = (P.p0 + 0.5 * (a * a + t2)) / pi - (P.alpha - 1.0 + 0.5 * 1.0);
                S.rhs[r] = beta + log(pi);
            }
        }
        __syncthreads();
        for (int i = tid; i < N; i += LT) S.pold[i] = S.p[i];
        if (tid == 0) {  // (T + I) tau = beta + log p with the host-factorised bands
            const double *f1 = P.band_lu, *f2 = f1 + N, *d0 = f2 + N, *u1 = d0 + N, *u2 = u1 + N;
            double x1 = S.rhs[0], x2 = 0.0;
            for (int i = 1; i < N; ++i) {
                double xi = S.rhs[i];
                xi = fma(-f2[i], x2, xi);
                xi = fma(-f1[i], x1, xi);
                S.rhs[i] = xi;
                x2 = x1;
                x1 = xi;
            }
            double y1 = 0.0, y2 = 0.0;
            for (int i = N - 1; i >= 0; --i) {
                double t = S.rhs[i];
                t = fma(-u1[i], y1, t);
                t = fma(-u2[i], y2, t);
                t = t / d0[i];
                S.rhs[i] = t;
                y2 = y1;
                y1 = t;
            }
        }
        __syncthreads();
        for (int i = tid; i < N; i += LT) S.p[i] = exp(S.rhs[i]);  // filter.py:177
        __syncthreads();
        if (P.mode == LN_MODE_UPDATE) break;
        in_pass = true;
    }

    for (int i = tid; i < N; i += LT) {
        P.s_out[i] = S.x[i];
        P.p_out[i] = S.p[i];
    }
    if (tid == 0) {
        P.result[0] = count;
        P.result[1] = status;
        for (int k = 0; k < 9; ++k) P.stats[k] = s_tot[k];
    }
}

}  // namespace

size_t fh_ln_smem_bytes(int N, int *lu_in_lds) {
    size_t doubles = 19 * (size_t)N + 64 + (size_t)N;
    const int fits = (N <= 112);
    if (lu_in_lds) *lu_in_lds = fits;
    if (fits) doubles += (size_t)N * N;
    return doubles * sizeof(double);
}

hipError_t fh_ln_launch(const LogNormalParams &P0, int nblocks, hipStream_t s) {
    LogNormalParams P = P0;
    const size_t smem = fh_ln_smem_bytes(P.N, &P.lu_in_lds);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(lognormal_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return e;
    lognormal_kernel<<<nblocks, LT, smem, s>>>(P);
    return hipGetLastError();
}
