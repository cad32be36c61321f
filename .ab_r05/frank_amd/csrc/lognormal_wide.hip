// method='LogNormal' beyond the persistent kernel's basis size (320 < N <= 1023): the pieces of LogNormalMAPModel's Newton
// minimisation as plain device kernels, driven from the host (capi_lognormal.hip: LnWide) with rocSOLVER's LU for the Newton system.
//
// Reference: LogNormalMAPModel._fit (statistical_models.py:1064-1160): H(s), jac(s), hess(s) (:1075-1122), limit_step (:1126-
// 1130); MinimizeNewton and LineSearch (minimizer.py:70-283) -- the control flow of those two lives on the host, every array
// and every O(N^2) operation here.  The persistent kernel (lognormal.hip) keeps twenty vectors, the solve vectors and an LU panel
// in LDS and ends at N = 320; this route trades its speed (a host round trip per function evaluation, ~60 us) for the full range
// of the binning pass and of the fit loop.  All reductions are single-workgroup trees in a fixed order: run-to-run identical.
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace {

constexpr int kT = 1024;  // threads of the single-workgroup kernels

__device__ __forceinline__ double block_sum(double v, double *scratch) {  // every thread returns the same bits
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    __syncthreads();
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    double r = 0.0;
    for (int w = 0; w < nw; ++w) r += scratch[w];
    return r;
}
// np.min / np.max propagate NaN
__device__ __forceinline__ double nan_min(double a, double b) { return (a != a || b != b) ? NAN : fmin(a, b); }
__device__ __forceinline__ double nan_max(double a, double b) { return (a != a || b != b) ? NAN : fmax(a, b); }
template <class Op>
__device__ __forceinline__ double block_reduce(double v, double *scratch, Op op) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = op(v, __shfl_xor(v, off));
    __syncthreads();
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    double r = scratch[0];
    for (int w = 1; w < nw; ++w) r = op(r, scratch[w]);
    return r;
}

// The log-space seed (radial_fitters.py:756-763): s = log(max(mu, 1e-3 max(mu))) - s0, p = max(DHT.transform(s)^2) (q / q0)^-4;
// pi_old = 0 and count = 0 (:766-768).  One workgroup.
__global__ void __launch_bounds__(kT) lnw_seed_kernel(LnWideParams P) {
    __shared__ double scratch[kT / 64];
    __shared__ double sv[1024];
    const int N = P.N;
    double mx = -INFINITY;
    for (int k = threadIdx.x; k < N; k += kT) mx = fmax(mx, P.mu[k]);
    mx = block_reduce(mx, scratch, [](double a, double b) { return fmax(a, b); });
    for (int k = threadIdx.x; k < N; k += kT) {
        const double s = log(fmax(P.mu[k], 1e-3 * mx)) - P.s0;
        sv[k] = s;
        P.x[k] = s;
    }
    __syncthreads();
    double best = -INFINITY;
    for (int k = threadIdx.x; k < N; k += kT) {
        const double *row = P.Ykm + (size_t)k * N;
        double a = 0.0;
        for (int m = 0; m < N; ++m) a = fma(row[m], sv[m], a);
        const double t = P.transform_norm * a;
        best = fmax(best, t * t);
    }
    const double pmax = block_reduce(best, scratch, [](double a, double b) { return fmax(a, b); });
    for (int k = threadIdx.x; k < N; k += kT) {
        P.p[k] = pmax * pow(P.q[k] / P.q[0], -4.0);
        P.p_old[k] = 0.0;
    }
    if (threadIdx.x == 0) {
        P.flags[FIT_FLAG_DONE] = 0;
        P.flags[FIT_FLAG_COUNT] = 0;
        P.flags[FIT_FLAG_BAD_P] = 0;
        P.flags[FIT_FLAG_NOT_SPD] = 0;
    }
}

// W[j][i] = Y[j][i] / p[j] (the einsum of statistical_models.py:1061 is then Y^T W); flags p <= 0 or NaN
__global__ void lnw_scale_kernel(LnWideParams P) {
    const int N = P.N;
    const size_t NN = (size_t)N * N;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < NN; e += (size_t)gridDim.x * blockDim.x) {
        const double pj = P.p[e / N];
        if (!(pj > 0.0)) P.flags[FIT_FLAG_BAD_P] = 1;
        P.W[e] = P.Y[e] * (1 / pj);
    }
}

// One function evaluation's O(N^2) part at xn = x + lam * dir (dir may be NULL: xn = x): I = exp(xn + s0), t1 = S^-1 xn,
// t2 = M I, and per row the summand of H(s) = 0.5 s.S^-1 s + 0.5 I.M I - I.j (statistical_models.py:1075-1085).
// Eight rows per workgroup, a wave per row; every workgroup forms xn and I for itself in LDS.
// mode 0: S^-1 xn multiplied out (the reference's arithmetic; at dir = NULL it also refreshes Sx = S^-1 x);
// mode 1: the first trial of a search: Sp = S^-1 dir multiplied out and kept, t1 = Sx + lam Sp;  mode 2: later trials of the
// same search: t1 = Sx + lam Sp, M only -- S^-1 is linear (the persistent kernel's default line search does the same: a trial
// point's S^-1 x carries no fresh rounding of its own, the searches accept at the first trial and the Hessian stays frozen)
__global__ void __launch_bounds__(512) lnw_eval_kernel(LnWideParams P, const double *x, const double *dir, double lam, int mode) {
    __shared__ double xs[1024], Is[1024];
    __shared__ int s_diff;
    const int N = P.N, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_diff = 0;
    __syncthreads();
    int diff = 0;
    for (int k = tid; k < N; k += 512) {
        const double x0 = x[k], v = dir ? x0 + lam * dir[k] : x0;
        diff |= v != x0;
        xs[k] = mode == 1 ? dir[k] : v;  // (the vector S^-1 multiplies)
        Is[k] = exp(v + P.s0);
    }
    if (diff) s_diff = 1;
    __syncthreads();
    const int i = blockIdx.x * 8 + wave;
    if (i < N) {
        const double *sr = P.Sinv + (size_t)i * N, *mr = P.M + (size_t)i * N;
        double a = 0.0, b = 0.0;
        if (mode != 2) {
            for (int k = lane; k < N; k += 64) {
                a = fma(sr[k], xs[k], a);
                b = fma(mr[k], Is[k], b);
            }
        } else {
            for (int k = lane; k < N; k += 64) b = fma(mr[k], Is[k], b);
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            a += __shfl_xor(a, off);
            b += __shfl_xor(b, off);
        }
        if (lane == 0) {
            const double x0 = x[i], xi = dir ? x0 + lam * dir[i] : x0, Ii = Is[i];
            double t1 = a;
            if (mode == 1) {
                P.Sp[i] = a;
                t1 = P.Sx[i] + lam * a;
            } else if (mode == 2) {
                t1 = P.Sx[i] + lam * P.Sp[i];
            } else if (!dir) {
                P.Sx[i] = a;
            }
            P.xn[i] = xi;
            P.I[i] = Ii;
            P.t1[i] = t1;
            P.t2[i] = b;
            P.fr[i] = 0.5 * (xi * t1) + 0.5 * (Ii * b) - Ii * P.j[i];
        }
    }
    if (blockIdx.x == 0 && tid == 0) P.scal[1] = s_diff ? 0.0 : 1.0;  // x + lam dir == x in every component (minimizer.py:128)
}
__global__ void __launch_bounds__(kT) lnw_sum_kernel(LnWideParams P) {
    __shared__ double scratch[kT / 64];
    double v = 0.0;
    for (int k = threadIdx.x; k < P.N; k += kT) v += P.fr[k];
    v = block_sum(v, scratch);
    if (threadIdx.x == 0) P.scal[0] = v;
}

// jac = S^-1 s + (I (M I) - I j) at the point of the last evaluation (:1087-1098); dx = -jac (the Newton right-hand side);
// scal[2] = max |jac| |x| (the convergence measure, minimizer.py:272).
__global__ void __launch_bounds__(kT) lnw_jac_kernel(LnWideParams P) {
    __shared__ double scratch[kT / 64];
    double g = -INFINITY;
    for (int k = threadIdx.x; k < P.N; k += kT) {
        const double Ii = P.I[k];
        const double jk = P.t1[k] + (Ii * P.t2[k] - Ii * P.j[k]);
        P.jx[k] = jk;
        P.dx[k] = -jk;
        g = nan_max(g, fabs(jk) * fabs(P.xn[k]));
    }
    g = block_reduce(g, scratch, [](double a, double b) { return nan_max(a, b); });
    if (threadIdx.x == 0) P.scal[2] = g;
}

// hess = diag(I) M diag(I) + diag(I (M I) - I j) + S^-1 at the point of the last evaluation (:1100-1122, full Hessian)
__global__ void lnw_hess_kernel(LnWideParams P, double *H) {
    const int N = P.N;
    const size_t NN = (size_t)N * N;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < NN; e += (size_t)gridDim.x * blockDim.x) {
        const int a = (int)(e / N), b = (int)(e - (size_t)a * N);
        const double Ia = P.I[a];
        double v = Ia * P.M[e] * P.I[b];
        if (a == b) v += Ia * P.t2[a] - Ia * P.j[a];
        H[e] = v + P.Sinv[e];
    }
}

// limit_step (:1126-1130) and the slope along it (minimizer.py:98-101): p = min(1.1 min |x / dir|, 1) dir, scal[3] = jac . p,
// scal[4] = jac . dir
__global__ void __launch_bounds__(kT) lnw_limit_step_kernel(LnWideParams P, const double *x, const double *dir, double *p) {
    __shared__ double scratch[kT / 64];
    double amin = INFINITY;
    for (int k = threadIdx.x; k < P.N; k += kT) amin = nan_min(amin, fabs(x[k] / dir[k]));
    amin = block_reduce(amin, scratch, [](double a, double b) { return nan_min(a, b); });
    double alpha = 1.1 * amin;
    if (1.0 < alpha) alpha = 1.0;  // min(alpha, 1): a NaN stays
    double d0 = 0.0, d1 = 0.0;
    for (int k = threadIdx.x; k < P.N; k += kT) {
        const double pk = alpha * dir[k];
        p[k] = pk;
        d0 += P.jx[k] * pk;
        d1 += P.jx[k] * dir[k];
    }
    d0 = block_sum(d0, scratch);
    d1 = block_sum(d1, scratch);
    if (threadIdx.x == 0) {
        P.scal[3] = d0;
        P.scal[4] = d1;
    }
}

// y = alpha A x + z (z may be NULL), a wave per row.  The Newton direction of a step comes from the EXPLICIT inverse of the factored
// Hessian -- a Hessian serves hundreds of steps; rocSOLVER's two triangular solves of one column cost ~250 us per step, three of
// these products ~25 -- with one step of iterative refinement against the unfactored Hessian (d0 = H^-1 b, r = b - H d0,
// dx = d0 + H^-1 r), which gives the solve the residual of an LU solve back.
__global__ void __launch_bounds__(512) lnw_matvec_kernel(int N, const double *A, const double *x, double alpha, const double *z,
                                                        double *y) {
    __shared__ double xs[1024];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int k = tid; k < N; k += 512) xs[k] = x[k];
    __syncthreads();
    const int i = blockIdx.x * 8 + wave;
    if (i >= N) return;
    const double *ar = A + (size_t)i * N;
    double a = 0.0;
    for (int k = lane; k < N; k += 64) a = fma(ar[k], xs[k], a);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) a += __shfl_xor(a, off);
    if (lane == 0) y[i] = alpha * a + (z ? z[i] : 0.0);
}
__global__ void lnw_identity_kernel(double *A, int N) {
    const size_t NN = (size_t)N * N;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < NN; e += (size_t)gridDim.x * blockDim.x)
        A[e] = (e / N == e % N) ? 1.0 : 0.0;
}

}  // namespace

hipError_t fh_lnw_launch_matvec(int N, const double *A, const double *x, double alpha, const double *z, double *y, hipStream_t s) {
    hipLaunchKernelGGL(lnw_matvec_kernel, dim3((N + 7) / 8), dim3(512), 0, s, N, A, x, alpha, z, y);
    return hipGetLastError();
}
hipError_t fh_lnw_launch_identity(double *A, int N, hipStream_t s) {
    hipLaunchKernelGGL(lnw_identity_kernel, dim3(256), dim3(256), 0, s, A, N);
    return hipGetLastError();
}
hipError_t fh_lnw_launch_seed(const LnWideParams &P, hipStream_t s) {
    hipLaunchKernelGGL(lnw_seed_kernel, dim3(1), dim3(kT), 0, s, P);
    return hipGetLastError();
}
hipError_t fh_lnw_launch_scale(const LnWideParams &P, hipStream_t s) {
    hipLaunchKernelGGL(lnw_scale_kernel, dim3(256), dim3(256), 0, s, P);
    return hipGetLastError();
}
hipError_t fh_lnw_launch_eval(const LnWideParams &P, const double *x, const double *dir, double lam, int mode, hipStream_t s) {
    hipLaunchKernelGGL(lnw_eval_kernel, dim3((P.N + 7) / 8), dim3(512), 0, s, P, x, dir, lam, mode);
    hipLaunchKernelGGL(lnw_sum_kernel, dim3(1), dim3(kT), 0, s, P);
    return hipGetLastError();
}
hipError_t fh_lnw_launch_jac(const LnWideParams &P, hipStream_t s) {
    hipLaunchKernelGGL(lnw_jac_kernel, dim3(1), dim3(kT), 0, s, P);
    return hipGetLastError();
}
hipError_t fh_lnw_launch_hess(const LnWideParams &P, double *H, hipStream_t s) {
    hipLaunchKernelGGL(lnw_hess_kernel, dim3(256), dim3(256), 0, s, P, H);
    return hipGetLastError();
}
hipError_t fh_lnw_launch_limit_step(const LnWideParams &P, const double *x, const double *dir, double *p, hipStream_t s) {
    hipLaunchKernelGGL(lnw_limit_step_kernel, dim3(1), dim3(kT), 0, s, P, x, dir, p);
    return hipGetLastError();
}
