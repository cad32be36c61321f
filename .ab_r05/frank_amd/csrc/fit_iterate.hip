// K2 `fit_iterate`: device side of the power-spectrum iteration (small kernels around rocBLAS / rocSOLVER).
//
// One pass of FrankFitter._fit's loop (radial_fitters.py:769-785) is, on the device:
//   fit_update_kernel   check_convergence(pI, pi_old) (filter.py:179-181); if not done: pi_old = pI,
//                       pI = update_power_spectrum(fit) (filter.py:154-177), count += 1
//   fit_prep_kernel     validate p (statistical_models.py:688-698); W = diag(1/p) Y; D = M
//   rocblas_dgemm       D += W^T Y          = M + S^-1           (statistical_models.py:700-701, 739)
//   rocsolver_dpotrf    D = U^T U                                (statistical_models.py:742)
//   rocsolver_dpotrs    mu = D^-1 j                              (statistical_models.py:745)
//   rocblas_dtrsm       Z = U^-T Y^T  -> Tr2_i = |Z[:, i]|^2     (filter.py:168, one triangular solve instead of two)
// All matrices are row-major N x N fp64 buffers; rocBLAS sees them as their column-major transposes.
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace {

constexpr int kFitThreads = 1024;

__device__ __forceinline__ double block_reduce_max(double v, double *scratch) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmax(v, __shfl_down(v, off));
    __syncthreads();
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    double r = scratch[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) r = fmax(r, scratch[w]);
    __syncthreads();
    return r;
}

// W[j][i] = Y[j][i] / p[j];  D = M;  flags p <= 0 or NaN (statistical_models.py:689).
__global__ void fit_prep_kernel(FitState st) {
    const int N = st.N;
    const size_t NN = (size_t)N * N;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < NN; e += (size_t)gridDim.x * blockDim.x) {
        const int jrow = (int)(e / N);
        const double pj = st.p[jrow];
        if (!(pj > 0.0)) st.flags[FIT_FLAG_BAD_P] = 1;
        st.W[e] = st.Y[e] * (1 / pj);  // einsum('ji,lj,jk->lik', Y, 1/p, Y): (Y[j,i] * (1/p)[j]) * Y[j,k]
        st.D[e] = st.M[e];
        st.Z[e] = st.Y[e];             // right-hand side of the triangular solve for Tr2
    }
    if (blockIdx.x == 0 && threadIdx.x < N) st.mu[threadIdx.x] = st.j[threadIdx.x];
    if (blockIdx.x == 0 && N > (int)blockDim.x)
        for (int k = threadIdx.x + blockDim.x; k < N; k += blockDim.x) st.mu[k] = st.j[k];
}

// p = 1 (radial_fitters.py:744)
__global__ void fit_init_kernel(FitState st) {
    for (int k = threadIdx.x; k < st.N; k += blockDim.x) {
        st.p[k] = 1.0;
        st.p_old[k] = 0.0;  // pi_old = 0, radial_fitters.py:768
    }
    if (threadIdx.x == 0) {
        st.flags[FIT_FLAG_DONE] = 0;
        st.flags[FIT_FLAG_COUNT] = 0;
        st.flags[FIT_FLAG_BAD_P] = 0;
        st.flags[FIT_FLAG_NOT_SPD] = 0;
        st.flags[FIT_FLAG_INFO] = 0;
    }
}

// pI = max(DHT.transform(MAP)^2) * (q/q[0])^-2   (radial_fitters.py:749-750, hankel.py:151-165)
__global__ void fit_powerlaw_kernel(FitState st) {
    __shared__ double scratch[kFitThreads / 64];
    const int N = st.N;
    if (st.info[0] != 0) st.flags[FIT_FLAG_NOT_SPD] = 1;
    double best = -INFINITY;
    for (int k = threadIdx.x; k < N; k += blockDim.x) {
        const double *row = st.Ykm + (size_t)k * N;
        double a = 0.0;
        for (int m = 0; m < N; ++m) a = fma(row[m], st.mu[m], a);
        const double t = st.transform_norm * a;
        best = fmax(best, t * t);
    }
    const double pmax = block_reduce_max(best, scratch);
    for (int k = threadIdx.x; k < N; k += blockDim.x) st.p[k] = pmax * pow(st.q[k] / st.q[0], -2.0);
}

// One loop pass up to (not including) the solve; see the file header.
__global__ void fit_update_kernel(FitState st) {
    __shared__ double rhs[FIT_MAX_N];
    __shared__ int s_flag;
    const int N = st.N;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    if (tid == 0) {
        if (st.info[0] != 0) st.flags[FIT_FLAG_NOT_SPD] = 1;  // potrf of the previous solve
        s_flag = st.flags[FIT_FLAG_DONE] | st.flags[FIT_FLAG_BAD_P] | st.flags[FIT_FLAG_NOT_SPD];
    }
    __syncthreads();
    if (s_flag) return;
    // check_convergence(pI, pi_old): all(|pI - pi_old| <= tol * pI)   (filter.py:181)
    int bad = 0;
    for (int k = tid; k < N; k += blockDim.x) bad |= !(fabs(st.p[k] - st.p_old[k]) <= st.tol * st.p[k]);
    bad = __syncthreads_or(bad);
    const int count = st.flags[FIT_FLAG_COUNT];
    if (!bad || count > st.max_iter) {  // loop condition radial_fitters.py:769-770
        if (tid == 0) st.flags[FIT_FLAG_DONE] = 1;
        return;
    }
    // Tr1 = (Y mu)^2, Tr2_i = sum_r Z[i][r]^2 (Z row-major holds (U^-T Y^T)^T row by row)   (filter.py:162-168)
    for (int i = wave; i < N; i += nwaves) {
        const double *yr = st.Y + (size_t)i * N, *zr = st.Z + (size_t)i * N;
        double a = 0.0, b = 0.0;
        for (int k = lane; k < N; k += 64) {
            a = fma(yr[k], st.mu[k], a);
            b = fma(zr[k], zr[k], b);
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            a += __shfl_down(a, off);
            b += __shfl_down(b, off);
        }
        if (lane == 0) {
            const double Tr1 = a * a, Tr2 = b;
            const double pi = st.p[i];
            const double beta = (st.p0 + 0.5 * (Tr1 + Tr2)) / pi - (st.alpha - 1.0 + 0.5 * 1.0);  // filter.py:172-173
            rhs[i] = beta + log(pi);
        }
    }
    __syncthreads();
    // tau = (T + I)^-1 rhs with the banded LU factors prepared on the host (filter.py:175)
    if (tid == 0) {
        const double *f1 = st.band_lu, *f2 = st.band_lu + N, *d0 = st.band_lu + 2 * N, *u1 = st.band_lu + 3 * N,
                     *u2 = st.band_lu + 4 * N;
        for (int i = 1; i < N; ++i) {
            // same elimination order as a row-by-row banded LU: row i-2's multiple first, then row i-1's
            if (i >= 2) rhs[i] -= f2[i] * rhs[i - 2];
            rhs[i] -= f1[i] * rhs[i - 1];
        }
        for (int i = N - 1; i >= 0; --i) {
            double t = rhs[i];
            if (i + 1 < N) t -= u1[i] * rhs[i + 1];
            if (i + 2 < N) t -= u2[i] * rhs[i + 2];
            rhs[i] = t / d0[i];
        }
    }
    __syncthreads();
    for (int k = tid; k < N; k += blockDim.x) {
        st.p_old[k] = st.p[k];
        const double pn = exp(rhs[k]);  // filter.py:177
        st.p[k] = pn;
        if (st.diag_p) st.diag_p[(size_t)count * N + k] = pn;
    }
    if (tid == 0) st.flags[FIT_FLAG_COUNT] = count + 1;
}

// After the solve of a loop pass: record MAP for the diagnostics (radial_fitters.py:783).
__global__ void fit_record_kernel(FitState st) {
    if (st.flags[FIT_FLAG_DONE] | st.flags[FIT_FLAG_BAD_P] | st.flags[FIT_FLAG_NOT_SPD]) return;
    const int count = st.flags[FIT_FLAG_COUNT];
    for (int k = threadIdx.x; k < st.N; k += blockDim.x) st.diag_mu[(size_t)(count - 1) * st.N + k] = st.mu[k];
}

}  // namespace

hipError_t fh_k2_launch_init(const FitState &st, hipStream_t s) {
    hipLaunchKernelGGL(fit_init_kernel, dim3(1), dim3(256), 0, s, st);
    return hipGetLastError();
}
hipError_t fh_k2_launch_prep(const FitState &st, hipStream_t s) {
    hipLaunchKernelGGL(fit_prep_kernel, dim3(128), dim3(256), 0, s, st);
    return hipGetLastError();
}
hipError_t fh_k2_launch_powerlaw(const FitState &st, hipStream_t s) {
    hipLaunchKernelGGL(fit_powerlaw_kernel, dim3(1), dim3(kFitThreads), 0, s, st);
    return hipGetLastError();
}
hipError_t fh_k2_launch_update(const FitState &st, hipStream_t s) {
    hipLaunchKernelGGL(fit_update_kernel, dim3(1), dim3(kFitThreads), 0, s, st);
    return hipGetLastError();
}
hipError_t fh_k2_launch_record(const FitState &st, hipStream_t s) {
    hipLaunchKernelGGL(fit_record_kernel, dim3(1), dim3(256), 0, s, st);
    return hipGetLastError();
}

// s1 = where(s > 0, 1 / s, 0): the cut of the SVD pseudo-inverse (statistical_models.py:751, :1155)
__global__ void pinv_scale_kernel(const double *s, int n, double *s1) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) s1[i] = s[i] > 0 ? 1.0 / s[i] : 0.0;
}

hipError_t fh_k2_launch_pinv_scale(const double *s, int n, double *s1, hipStream_t st) {
    hipLaunchKernelGGL(pinv_scale_kernel, dim3((n + 255) / 256), dim3(256), 0, st, s, n, s1);
    return hipGetLastError();
}

