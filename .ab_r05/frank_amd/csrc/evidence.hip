// Posterior extras of a whole sweep on the device: for every (alpha, w_smooth) point of a sweep over one mapping the marginal
// likelihood, the curvature of log P(p, V) in tau = log p at the maximum (the inverse covariance of the power spectrum) and its
// determinant -- what FrankFitter.log_likelihood / log_evidence_laplace / MAP_spectrum_covariance compute one point at a time:
//   GaussianModel.log_likelihood        statistical_models.py:790-856   1/2 j^T mu + 1/2 log det(D S^-1) + H0
//   CriticalFilter.covariance_MAP       filter.py:184-227               the Hessian below
//   FrankFitter.log_evidence_laplace    radial_fitters.py:951-967       log P(p_MAP, V) - 1/2 log det(Hessian / 2 pi)
// The reference forms Y D Y^T and the Hessian with dense NumPy products and calls slogdet: O(N^3) on the host per point, 512 times
// for a sweep that is to be ranked by evidence (fit.py:534-548).  In the basis where the prior is diagonal (fit_loop.hip) the
// posterior covariance of m = Y mu is C^-1, C = A + diag(1/p), A = Y^-T M Y^-1 -- so Dqq = Y D Y^T = C^-1 needs no product with Y
// at all, det(D S^-1) = det(diag(1/p)) / det(C), and a point costs two Cholesky factorisations and two inversions of N x N
// matrices, batched over the points (rocSOLVER potrf / potri, strided batched); these kernels only fill and read the batches.
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace {

// C_b = (Araw + Araw^T) / 2 + diag(1 / p_b)  (row-major == column-major: symmetric)
__global__ void evidence_build_c_kernel(const double *Araw, const double *p, int N, int batch, double *C) {
    const size_t NN = (size_t)N * N, total = NN * (size_t)batch;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(e / NN);
        const size_t r = e - (size_t)b * NN;
        const int i = (int)(r / N), j = (int)(r - (size_t)i * N);
        double v = 0.5 * (Araw[(size_t)i * N + j] + Araw[(size_t)j * N + i]);
        if (i == j) v += 1.0 / p[(size_t)b * N + i];
        C[e] = v;
    }
}
// out[b] = 2 sum_i log L_ii of the factors a batched potrf left in place (one wave per matrix)
__global__ void evidence_logdet_kernel(const double *L, int N, int batch, double *out) {
    const int b = blockIdx.x, lane = threadIdx.x;
    double s = 0.0;
    for (int i = lane; i < N; i += 64) s += log(L[(size_t)b * N * N + (size_t)i * N + i]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_down(s, off);
    if (lane == 0) out[b] = 2.0 * s;
}
// The Hessian of -log P(p, V) in tau = log p at the maximum (filter.py:213-217), from Dqq = C^-1 (valid triangle: row <= column
// of the row-major view, what potri(lower) leaves in a column-major matrix) and mq = Y mu:
//   H_ij = delta_ij (p0 / p_i + (mq_i^2 + Dqq_ii) / (2 p_i)) + w T_ij - (2 mq_i mq_j + Dqq_ij) Dqq_ij / (2 p_i p_j)
__global__ void evidence_hessian_kernel(const double *Dqq, const double *mq, const double *p, const double *p0, const double *ws,
                                        const double *Tband /* [5][N]: T_unit[i][i + d - 2] */, int N, int batch, double *H) {
    const size_t NN = (size_t)N * N, total = NN * (size_t)batch;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(e / NN);
        const size_t r = e - (size_t)b * NN;
        const int i = (int)(r / N), j = (int)(r - (size_t)i * N);
        const double *Db = Dqq + (size_t)b * NN, *pb = p + (size_t)b * N, *mb = mq + (size_t)b * N;
        const int lo = i < j ? i : j, hi = i < j ? j : i;
        const double dij = Db[(size_t)lo * N + hi];
        double h = -0.5 * (1.0 / pb[i]) * (1.0 / pb[j]) * (2.0 * (mb[i] * mb[j]) + dij) * dij;
        const int d = j - i;
        if (d >= -2 && d <= 2) h += ws[b] * Tband[(size_t)(d + 2) * N + i];
        if (i == j) h += p0[b] / pb[i] + 0.5 * (mb[i] * mb[i] + Db[(size_t)i * N + i]) / pb[i];
        H[e] = h;
    }
}
__global__ void evidence_diag_kernel(const double *A, int N, int batch, double *out) {
    const int total = N * batch;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int b = e / N, i = e - b * N;
        out[e] = A[(size_t)b * N * N + (size_t)i * N + i];
    }
}

// complex visibilities as NumPy holds them (re, im interleaved) -> the two columns of the table
__global__ void split_complex_kernel(const double *vc, int64_t n, double *re, double *im) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double2 z = reinterpret_cast<const double2 *>(vc)[i];
        re[i] = z.x;
        im[i] = z.y;
    }
}

}  // namespace

hipError_t fh_launch_split_complex(const double *vc, int64_t n, double *re, double *im, hipStream_t s) {
    hipLaunchKernelGGL(split_complex_kernel, dim3(2048), dim3(256), 0, s, vc, n, re, im);
    return hipGetLastError();
}
hipError_t fh_evidence_launch_build_c(const double *Araw, const double *p, int N, int batch, double *C, hipStream_t s) {
    hipLaunchKernelGGL(evidence_build_c_kernel, dim3(1024), dim3(256), 0, s, Araw, p, N, batch, C);
    return hipGetLastError();
}
hipError_t fh_evidence_launch_logdet(const double *L, int N, int batch, double *out, hipStream_t s) {
    hipLaunchKernelGGL(evidence_logdet_kernel, dim3(batch), dim3(64), 0, s, L, N, batch, out);
    return hipGetLastError();
}
hipError_t fh_evidence_launch_hessian(const double *Dqq, const double *mq, const double *p, const double *p0, const double *ws,
                                      const double *Tband, int N, int batch, double *H, hipStream_t s) {
    hipLaunchKernelGGL(evidence_hessian_kernel, dim3(1024), dim3(256), 0, s, Dqq, mq, p, p0, ws, Tband, N, batch, H);
    return hipGetLastError();
}
hipError_t fh_evidence_launch_diag(const double *A, int N, int batch, double *out, hipStream_t s) {
    hipLaunchKernelGGL(evidence_diag_kernel, dim3(256), dim3(256), 0, s, A, N, batch, out);
    return hipGetLastError();
}
