// Kernels of the geometry fits (geometry.py:404-763), the callers that run the binning pass inside an optimiser's residual
// function: both stream the resident visibility table once per evaluation and write 16 bytes per row (HBM-bound: 40 B in,
// 16 B out -- the Bessel sums of a small basis hide under the loads).
//
//   vis_residual_kernel   sqrt(w) (V_model - V) of a brightness profile under a trial geometry: the last two lines of
//                         FitGeometryFourierBessel._residual (geometry.py:681-683), i.e. FrankRadialFit.predict
//                         (radial_fitters.py:56-98: deproject, H(q) I, scale, re-phase) minus the data.
//   gauss_residual_kernel residual and 6-column Jacobian of the Gaussian in the uv-plane (_fit_geometry_gaussian,
//                         geometry.py:535-585).
//
// One thread per row; zeros, prefactors and the profile in LDS beside the J0 tables.  Sums of squares leave as one partial per
// workgroup and are added in order by one workgroup (run-to-run identical).
#include <hip/hip_runtime.h>

#include "bessel.h"
#include "deproject.h"
#include "j0_buckets.h"
#include "kernels.h"

namespace {

constexpr int kThreads = 256;

__device__ __forceinline__ double block_sum(double x, double *red) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) x += __shfl_down(x, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = x;
    __syncthreads();
    double r = 0.0;
    for (int w = 0; w < kThreads / 64; ++w) r += red[w];
    __syncthreads();
    return r;
}

// TABLES: the model visibility of a row is a degree-11 polynomial in the offset of s = q / Qmax inside its bucket (the tables
// the binning pass of the same rows has built, contracted with the profile: 12 numbers per bucket) instead of N Bessel
// evaluations -- the pass is then bound by the 40 B read and 16 B written per row.
template <bool TABLES>
__global__ __launch_bounds__(kThreads) void vis_residual_kernel(VisResidualParams P) {
    extern __shared__ double lds[];
    double *tab = lds, *zk = tab + FH_J0_TABLE_DOUBLES, *ck = zk + P.b.N, *Ik = ck + P.b.N, *h2 = Ik + P.b.N;
    __shared__ double red[kThreads / 64];
    const int N = P.b.N;
    if (!TABLES) {
        for (int i = threadIdx.x; i < FH_J0_TABLE_DOUBLES; i += kThreads) tab[i] = P.b.j0_table[i];
        for (int k = threadIdx.x; k < N; k += kThreads) {
            zk[k] = P.b.zeros[k];
            // H[i, k] = (pref_k J0) * scale, V = H . I  (hankel.py:201-202, statistical_models.py:486-496, :326-328)
            ck[k] = P.pref[k];
            Ik[k] = P.I[k];
            if (P.b.H2) h2[k] = P.b.H2[k];
        }
        __syncthreads();
    }
    const double inv_delta = 1.0 / P.delta, inv_half = 2.0 * inv_delta;
    double ss = 0.0;
    const int64_t stride = (int64_t)gridDim.x * kThreads;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < P.b.count; i += stride) {
        VisRow r;
        if (P.predict_only) {
            fh_load_uv(P.b, P.b.first + i, r.u, r.v);
            r.Vre = r.Vim = 0.0;
            r.w = 1.0;
        } else {
            r = fh_load_row(P.b, P.b.first + i);
        }
        const double s = fh_deproject_q(P.b, r.u, r.v) * P.b.inv_Qmax;
        double a = 0.0;
        if (TABLES) {
            const int b = fh_bucket_of(s, inv_delta, P.nb);
            const double tau = fh_bucket_tau(s, b, P.delta, inv_half);
            const double *c = P.coef + (size_t)b * FH_K1_TERMS;
            a = c[FH_K1_TERMS - 1];
#pragma unroll
            for (int m = FH_K1_TERMS - 2; m >= 0; --m) a = fma(a, tau, c[m]);
        } else if (P.b.H2) {
            const double kz2 = fh_deproject_kz2(P.b, r.u, r.v);
            for (int k = 0; k < N; ++k) {
                const double h = (ck[k] * fh_j0(s * zk[k], tab)) * exp(-kz2 * h2[k]);
                a = fma(h, Ik[k], a);
            }
        } else {
            for (int k = 0; k < N; ++k) {
                const double h = (ck[k] * fh_j0(s * zk[k], tab)) * P.scale;
                a = fma(h, Ik[k], a);
            }
        }
        // undo_correction (geometry.py:238-268): the model visibility is real in the source frame; the phase centre offset
        // turns it by exp(+i phi), phi = u dRA + v dDec on the sky-plane baselines
        double sn, cs;
        {
#pragma clang fp contract(off)
            const double phi = r.u * P.b.dRA + r.v * P.b.dDec;
            sincos(phi, &sn, &cs);
        }
        const double sw = sqrt(r.w);
        const double er = sw * (a * cs - r.Vre), ei = sw * (a * sn - r.Vim);
        if (P.out) {
            P.out[i] = er;
            P.out[P.b.count + i] = ei;
        }
        ss = fma(er, er, ss);
        ss = fma(ei, ei, ss);
    }
    ss = block_sum(ss, red);
    if (threadIdx.x == 0) P.partial[blockIdx.x] = ss;
}

__global__ __launch_bounds__(kThreads) void fold_partials_kernel(const double *partial, int n, double *out) {
    __shared__ double red[kThreads / 64];
    double a = 0.0;
    for (int i = threadIdx.x; i < n; i += kThreads) a += partial[i];
    a = block_sum(a, red);
    if (threadIdx.x == 0) out[0] = a;
}

// f = sqrt(w) (norm exp(-(u'^2 + v'^2) / 2) - V'), u' = (u cos PA - v sin PA) cos inc / (scal rad_to_arcsec),
// v' = (u sin PA + v cos PA) / (scal rad_to_arcsec), V' = V exp(-i phi)  (geometry.py:535-551); the Jacobian columns of
// :553-585 in the order (inc, PA, dRA, dDec, norm, scal), row-major [2n][6] as least_squares takes it.  (The PA column is the
// reference's, which is half the derivative -- the "/ 2" of geometry.py:572; kept, so that the optimiser takes the reference's
// steps: a scaled column moves the path, not the point where J^T f = 0.)
__global__ __launch_bounds__(kThreads) void gauss_residual_kernel(GaussResidualParams P) {
    __shared__ double red[kThreads / 64];
    double ss = 0.0;
    const int64_t n = P.b.count, stride = (int64_t)gridDim.x * kThreads;
    const double c_t = P.b.cos_t, s_t = P.b.sin_t, c_i = P.b.cos_i, s_i = P.b.sin_i;
    const double sr = P.scal * P.rad_to_arcsec, inv_sr2 = 1.0 / (sr * sr);
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) {
        const VisRow r = fh_load_row(P.b, P.b.first + i);
        const double sw = sqrt(r.w);
        double sn, cs;  // (a given phase centre is applied like a trial one; only its Jacobian columns are zero)
        sincos(r.u * P.b.dRA + r.v * P.b.dDec, &sn, &cs);
        // V' = V (cos phi - i sin phi)
        const double Vr = r.Vre * cs + r.Vim * sn, Vi = r.Vim * cs - r.Vre * sn;
        const double up = r.u * c_t - r.v * s_t, vp = r.u * s_t + r.v * c_t;
        const double uv = up * up * c_i * c_i + vp * vp;
        const double G = exp(-0.5 * uv * inv_sr2);
        const double fr = sw * (P.norm * G - Vr), fi = -sw * Vi;
        if (P.fun) {
            P.fun[i] = fr;
            P.fun[n + i] = fi;
        }
        ss = fma(fr, fr, ss);
        ss = fma(fi, fi, ss);
        if (P.jac) {
            double *jr = P.jac + 6 * i, *ji = P.jac + 6 * (n + i);
            const double wG = sw * G, nrm = P.norm * inv_sr2;
            // d/d(dRA), d/d(dDec) of -sqrt(w) V exp(-i phi): +i sqrt(w) V' fac (u | v)   (fac is inside dRA, dDec's units)
            const double dr = P.fit_phase ? -sw * Vi * P.fac : 0.0, di = P.fit_phase ? sw * Vr * P.fac : 0.0;
            jr[0] = P.fit_inc_pa ? nrm * wG * up * up * c_i * s_i : 0.0;
            jr[1] = P.fit_inc_pa ? nrm * wG * up * vp * (c_i * c_i - 1.0) / 2.0 : 0.0;
            jr[2] = dr * r.u;
            jr[3] = dr * r.v;
            jr[4] = wG;
            jr[5] = nrm * wG * uv / P.scal;
            ji[0] = 0.0;
            ji[1] = 0.0;
            ji[2] = di * r.u;
            ji[3] = di * r.v;
            ji[4] = 0.0;
            ji[5] = 0.0;
        }
    }
    ss = block_sum(ss, red);
    if (threadIdx.x == 0) P.partial[blockIdx.x] = ss;
}


// ---- Levenberg-Marquardt on the normal equations: J^T J, J^T r by streaming kernels, nothing of size n leaves the device ----
// K running sums per thread -> one row of K per workgroup (wave shuffles, then the four waves through LDS)
template <int K>
__device__ __forceinline__ void block_sums_store(double (&acc)[K], double *row) {
    __shared__ double redk[K][kThreads / 64];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        double x = acc[k];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) x += __shfl_down(x, off);
        if ((threadIdx.x & 63) == 0) redk[k][threadIdx.x >> 6] = x;
    }
    __syncthreads();
    if (threadIdx.x < K) {
        double r = 0.0;
        for (int w = 0; w < kThreads / 64; ++w) r += redk[threadIdx.x][w];
        row[threadIdx.x] = r;
    }
}

// column k of the [nblocks][K] partial sums, one workgroup per column (a fixed tree: run-to-run identical)
__global__ __launch_bounds__(kThreads) void fold_rows_kernel(const double *partial, int nblocks, int K, double *out) {
    __shared__ double red[kThreads / 64];
    const int k = blockIdx.x;
    double a = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += kThreads) a += partial[(size_t)b * K + k];
    a = block_sum(a, red);
    if (threadIdx.x == 0) out[k] = a;
}

// forward-difference Jacobian columns d_k = (r_k - r_0) / h_k of up to four parameters from residual vectors kept on the
// device (MINPACK's fdjac2); sums: J^T J (upper triangle, row by row: 10), J^T r_0 (4)
constexpr int kFdSums = 14;
__global__ __launch_bounds__(kThreads) void fd_normal_kernel(FdNormalParams P) {
    double acc[kFdSums];
#pragma unroll
    for (int k = 0; k < kFdSums; ++k) acc[k] = 0.0;
    const int64_t stride = (int64_t)gridDim.x * kThreads;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < P.len; i += stride) {
        const double r0 = P.base[i];
        double d[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) d[k] = k < P.ncol ? (P.col[k][i] - r0) * P.inv_h[k] : 0.0;
        int s = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int l = k; l < 4; ++l) acc[s] = fma(d[k], d[l], acc[s]), ++s;
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[10 + k] = fma(d[k], r0, acc[10 + k]);
    }
    block_sums_store<kFdSums>(acc, P.partial + (size_t)blockIdx.x * kFdSums);
}

// the Gaussian's residual and analytic Jacobian row by row (gauss_residual_kernel's arithmetic), summed into J^T J (upper
// triangle, 21), J^T f (6) and f^T f (1)
constexpr int kGaussSums = 28;
__global__ __launch_bounds__(kThreads) void gauss_normal_kernel(GaussResidualParams P) {
    double acc[kGaussSums];
#pragma unroll
    for (int k = 0; k < kGaussSums; ++k) acc[k] = 0.0;
    const int64_t n = P.b.count, stride = (int64_t)gridDim.x * kThreads;
    const double c_t = P.b.cos_t, s_t = P.b.sin_t, c_i = P.b.cos_i, s_i = P.b.sin_i;
    const double sr = P.scal * P.rad_to_arcsec, inv_sr2 = 1.0 / (sr * sr);
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) {
        const VisRow r = fh_load_row(P.b, P.b.first + i);
        const double sw = sqrt(r.w);
        double sn, cs;
        sincos(r.u * P.b.dRA + r.v * P.b.dDec, &sn, &cs);
        const double Vr = r.Vre * cs + r.Vim * sn, Vi = r.Vim * cs - r.Vre * sn;
        const double up = r.u * c_t - r.v * s_t, vp = r.u * s_t + r.v * c_t;
        const double uv = up * up * c_i * c_i + vp * vp;
        const double G = exp(-0.5 * uv * inv_sr2);
        const double f[2] = {sw * (P.norm * G - Vr), -sw * Vi};
        const double wG = sw * G, nrm = P.norm * inv_sr2;
        const double dr = P.fit_phase ? -sw * Vi * P.fac : 0.0, di = P.fit_phase ? sw * Vr * P.fac : 0.0;
        const double J[2][6] = {{P.fit_inc_pa ? nrm * wG * up * up * c_i * s_i : 0.0,
                                 P.fit_inc_pa ? nrm * wG * up * vp * (c_i * c_i - 1.0) / 2.0 : 0.0, dr * r.u, dr * r.v, wG,
                                 nrm * wG * uv / P.scal},
                                {0.0, 0.0, di * r.u, di * r.v, 0.0, 0.0}};
#pragma unroll
        for (int part = 0; part < 2; ++part) {
            int s = 0;
#pragma unroll
            for (int k = 0; k < 6; ++k)
#pragma unroll
                for (int l = k; l < 6; ++l) acc[s] = fma(J[part][k], J[part][l], acc[s]), ++s;
#pragma unroll
            for (int k = 0; k < 6; ++k) acc[21 + k] = fma(J[part][k], f[part], acc[21 + k]);
            acc[27] = fma(f[part], f[part], acc[27]);
        }
    }
    block_sums_store<kGaussSums>(acc, P.partial + (size_t)blockIdx.x * kGaussSums);
}

int grid_for(int64_t n, int max_blocks) {
    int64_t g = (n + kThreads - 1) / kThreads;
    if (g > max_blocks) g = max_blocks;
    return g < 1 ? 1 : (int)g;
}

}  // namespace

int fh_residual_max_blocks() { return 2048; }  // 8 workgroups per compute unit

hipError_t fh_launch_vis_residual(const VisResidualParams &P, double *sumsq, hipStream_t stream) {
    const int grid = grid_for(P.b.count, fh_residual_max_blocks());
    const size_t lds = sizeof(double) * (FH_J0_TABLE_DOUBLES + 4 * (size_t)P.b.N);
    if (P.coef)
        hipLaunchKernelGGL(vis_residual_kernel<true>, dim3(grid), dim3(kThreads), 0, stream, P);
    else
        hipLaunchKernelGGL(vis_residual_kernel<false>, dim3(grid), dim3(kThreads), lds, stream, P);
    hipLaunchKernelGGL(fold_partials_kernel, dim3(1), dim3(kThreads), 0, stream, P.partial, grid, sumsq);
    return hipGetLastError();
}

hipError_t fh_launch_gauss_residual(const GaussResidualParams &P, double *sumsq, hipStream_t stream) {
    const int grid = grid_for(P.b.count, fh_residual_max_blocks());
    hipLaunchKernelGGL(gauss_residual_kernel, dim3(grid), dim3(kThreads), 0, stream, P);
    hipLaunchKernelGGL(fold_partials_kernel, dim3(1), dim3(kThreads), 0, stream, P.partial, grid, sumsq);
    return hipGetLastError();
}

int fh_residual_sums_max() { return kGaussSums; }

hipError_t fh_launch_fd_normal(const FdNormalParams &P, double *out14, hipStream_t stream) {
    const int grid = grid_for(P.len, fh_residual_max_blocks());
    hipLaunchKernelGGL(fd_normal_kernel, dim3(grid), dim3(kThreads), 0, stream, P);
    hipLaunchKernelGGL(fold_rows_kernel, dim3(kFdSums), dim3(kThreads), 0, stream, P.partial, grid, kFdSums, out14);
    return hipGetLastError();
}

hipError_t fh_launch_gauss_normal(const GaussResidualParams &P, double *out28, hipStream_t stream) {
    const int grid = grid_for(P.b.count, fh_residual_max_blocks());
    hipLaunchKernelGGL(gauss_normal_kernel, dim3(grid), dim3(kThreads), 0, stream, P);
    hipLaunchKernelGGL(fold_rows_kernel, dim3(kGaussSums), dim3(kThreads), 0, stream, P.partial, grid, kGaussSums, out28);
    return hipGetLastError();
}
