// uvbin.hip -- utilities.UVDataBinner (frank/utilities.py:180-400): weighted means of the visibilities in
// equal-width baseline bins, the number of rows per bin and the error of the mean.
//
// HBM-bound streaming work: 32 B per row and pass (uv, Re V, Im V, w), a handful of flops, and a scatter into a
// histogram of a few hundred to a few thousand bins.  Each workgroup keeps a private histogram in LDS (hardware
// ds_add_f64 / ds_add_u64), flushes it once with global fp64 atomics; the bin INDEX of a row is integer work and
// follows the reference's floor / edge fix-ups exactly (bit-exact indices and counts; the fp64 sums differ from
// NumPy's sequential bincount by summation order only).
//
//   uvbin_index        bin_quantities   :329-341  (accumulation index, never -1)
//   uvbin_lookup       determine_uv_bin :271-298  (-1 past the last edge)
//   uvbin_max_kernel   __init__ :204     uv.max()
//   uvbin_sum_kernel   bin_quantities   :300-366  sums of w*q for up to four quantities + counts
//   uvbin_err_kernel   __init__ :239-246 sums of w^2 (V - mean[bin])^2
#include <hip/hip_runtime.h>

#include "kernels.h"

#pragma clang fp contract(off)

namespace {

constexpr int UT = 256;

__device__ __forceinline__ double edge(int k, double bin_width) { return (double)k * bin_width; }  // arange * width

__device__ __forceinline__ int uvbin_index(double uv, double norm, double bin_width, int nbins) {
    int idx = (int)floor(uv * norm);
    if (uv < edge(idx, bin_width)) idx -= 1;
    if (idx == nbins) idx -= 1;
    if (uv >= edge(idx + 1, bin_width) && idx + 1 != nbins) idx += 1;
    return idx;
}

__device__ __forceinline__ int uvbin_lookup(double uv, double norm, double bin_width, int nbins) {
    int idx = (int)floor(uv * norm);
    if (uv < edge(idx, bin_width)) idx -= 1;
    if (uv == edge(nbins, bin_width)) idx -= 1;
    if (idx >= nbins) return -1;
    if (uv >= edge(idx + 1, bin_width) && idx + 1 < nbins) idx += 1;
    return idx;
}

__global__ __launch_bounds__(UT) void uvbin_max_kernel(const double *uv, int64_t n, unsigned long long *out) {
    // baselines are non-negative: the IEEE bit pattern orders like the value
    double m = 0.0;
    int bad = 0;
    for (int64_t i = (int64_t)blockIdx.x * UT + threadIdx.x; i < n; i += (int64_t)gridDim.x * UT) {
        const double x = uv[i];
        bad |= !(x >= 0.0);
        m = fmax(m, x);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmax(m, __shfl_xor(m, off));
    __shared__ double wmax[UT / 64];
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {  // one contended atomic per workgroup, not per wave
#pragma unroll
        for (int w = 1; w < UT / 64; ++w) m = fmax(m, wmax[w]);
        atomicMax(out, (unsigned long long)__double_as_longlong(m));
    }
    if (bad) atomicOr(out + 1, 1ull);
}

// sums[q * nbins + b] += w * qty_q for the given quantities, counts[b] += 1
__global__ __launch_bounds__(1024) void uvbin_sum_kernel(UvBinParams p) {
    extern __shared__ __attribute__((aligned(16))) double hist[];
    const int nb = p.nbins, nq = p.nq;
    unsigned long long *hcnt = reinterpret_cast<unsigned long long *>(hist + (size_t)nq * nb);
    if (p.use_lds) {
        for (int i = threadIdx.x; i < (nq + 1) * nb; i += blockDim.x) hist[i] = 0.0;  // +0.0 and 0ull share the bit pattern
        __syncthreads();
    }
    double *acc = p.use_lds ? hist : p.sums;
    unsigned long long *cnt = p.use_lds ? hcnt : p.counts;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < p.n; i += (int64_t)gridDim.x * blockDim.x) {
        const double x = p.uv[i], w = p.w[i];
        const int b = uvbin_index(x, p.norm, p.bin_width, nb);
        if (b < 0 || b >= nb) continue;  // NaN / negative baselines: reported by the max kernel
        if (p.count) atomicAdd(cnt + b, 1ull);
#pragma unroll 4
        for (int q = 0; q < nq; ++q) {
            const double *src = p.qty[q];
            unsafeAtomicAdd(acc + (size_t)q * nb + b, w * (src ? src[i] : 1.0));
        }
    }
    if (p.use_lds) {
        // private histogram -> this workgroup's slab; uvbin_fold_kernel adds the slabs in block order (no contended
        // global atomics, and the cross-workgroup part of the sum is deterministic)
        __syncthreads();
        unsigned long long *slab = reinterpret_cast<unsigned long long *>(p.scratch) + (size_t)blockIdx.x * (nq + 1) * nb;
        const unsigned long long *raw = reinterpret_cast<const unsigned long long *>(hist);
        for (int i = threadIdx.x; i < (nq + 1) * nb; i += blockDim.x) slab[i] = raw[i];  // sums and counts alike, bit for bit
    }
}

// sums[i] += the slabs of the workgroups [y * per_group, ...) (quantities), counts likewise (the last nbins entries of
// a slab); gridDim.y groups of slabs so that the 8-deep unrolled reads of a thread are the only serial part
constexpr int kFoldGroups = 32;
__global__ __launch_bounds__(UT) void uvbin_fold_kernel(const double *scratch, int blocks, int nq, int nbins, int has_counts,
                                                        double *sums, unsigned long long *counts) {
    const int per = (nq + has_counts) * nbins;
    const int i = blockIdx.x * UT + threadIdx.x;
    if (i >= per) return;
    const int per_group = (blocks + kFoldGroups - 1) / kFoldGroups;
    const int b0 = blockIdx.y * per_group, b1 = min(blocks, b0 + per_group);
    if (b0 >= b1) return;
    if (i < nq * nbins) {
        double a = 0.0;
#pragma unroll 8
        for (int b = b0; b < b1; ++b) a += scratch[(size_t)b * per + i];
        unsafeAtomicAdd(sums + i, a);
    } else {
        unsigned long long c = 0;
        const unsigned long long *sc = reinterpret_cast<const unsigned long long *>(scratch);
#pragma unroll 8
        for (int b = b0; b < b1; ++b) c += sc[(size_t)b * per + i];
        atomicAdd(counts + (i - nq * nbins), c);
    }
}

// sums[b] += w^2 (Re V - mu_re[bin])^2, sums[nbins + b] += w^2 (Im V - mu_im[bin])^2   (utilities.py:239-246)
__global__ __launch_bounds__(1024) void uvbin_err_kernel(UvBinParams p) {
    extern __shared__ __attribute__((aligned(16))) double hist[];
    const int nb = p.nbins, nq = p.qty[1] ? 2 : 1;
    if (p.use_lds) {
        for (int i = threadIdx.x; i < nq * nb; i += blockDim.x) hist[i] = 0.0;
        __syncthreads();
    }
    double *acc = p.use_lds ? hist : p.sums;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < p.n; i += (int64_t)gridDim.x * blockDim.x) {
        const double x = p.uv[i], w = p.w[i];
        const int b = uvbin_index(x, p.norm, p.bin_width, nb);
        const int l = uvbin_lookup(x, p.norm, p.bin_width, nb);
        if (b < 0 || b >= nb || l < 0) continue;
        const double w2 = w * w;
        const double dr = p.qty[0][i] - p.mu_re[l];
        unsafeAtomicAdd(acc + b, w2 * (dr * dr));
        if (nq == 2) {
            const double di = p.qty[1][i] - p.mu_im[l];
            unsafeAtomicAdd(acc + nb + b, w2 * (di * di));
        }
    }
    if (p.use_lds) {
        __syncthreads();
        double *slab = p.scratch + (size_t)blockIdx.x * nq * nb;
        for (int i = threadIdx.x; i < nq * nb; i += blockDim.x) slab[i] = hist[i];
    }
}

__global__ __launch_bounds__(UT) void uvbin_lookup_kernel(const double *uv, int64_t n, double norm, double bin_width,
                                                          int nbins, int *out) {
    for (int64_t i = (int64_t)blockIdx.x * UT + threadIdx.x; i < n; i += (int64_t)gridDim.x * UT)
        out[i] = uvbin_lookup(uv[i], norm, bin_width, nbins);
}

int grid_for(int64_t n, int num_cu) {
    const int64_t want = (n + UT - 1) / UT;
    const int64_t cap = (int64_t)num_cu * 4;
    return (int)(want < 1 ? 1 : (want < cap ? want : cap));
}

}  // namespace

hipError_t fh_uvbin_launch_max(const double *uv, int64_t n, unsigned long long *out2, int num_cu, hipStream_t s) {
    uvbin_max_kernel<<<grid_for(n, num_cu), UT, 0, s>>>(uv, n, out2);
    return hipGetLastError();
}

static size_t uvbin_lds_bytes(int nq_plus, int nbins) { return (size_t)nq_plus * nbins * sizeof(double); }

// workgroups of a streaming pass: enough to fill the chip, few enough that the slabs stay small
static int uvbin_blocks(int64_t n, int num_cu, size_t slab_doubles) {
    int g = grid_for(n, num_cu);
    const size_t budget = (size_t)2 << 20;  // 2 M doubles (16 MB) of slabs at most: they are written and read back
    while (g > num_cu && (size_t)g * slab_doubles > budget) g /= 2;
    return g;
}

size_t fh_uvbin_scratch_doubles(int nq_plus, int nbins, int64_t n, int num_cu) {
    return (size_t)uvbin_blocks(n, num_cu, (size_t)nq_plus * nbins) * nq_plus * nbins;
}

hipError_t fh_uvbin_launch_sum(const UvBinParams &p0, int num_cu, hipStream_t s) {
    UvBinParams p = p0;
    const size_t lds = uvbin_lds_bytes(p.nq + 1, p.nbins);
    p.use_lds = lds <= 120 * 1024 && p.scratch != nullptr;
    hipError_t e = hipSuccess;
    if (p.use_lds)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(uvbin_sum_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds);
    if (e != hipSuccess) return e;
    // a histogram above 40 KB leaves room for one workgroup per CU: make it a big one (16 waves hide the latency)
    const int threads = (p.use_lds && lds > 40 * 1024) ? 1024 : UT;
    int g = p.use_lds ? uvbin_blocks(p.n, num_cu, (size_t)(p.nq + 1) * p.nbins) : grid_for(p.n, num_cu);
    if (threads == 1024 && g > num_cu) g = num_cu;
    uvbin_sum_kernel<<<g, threads, p.use_lds ? lds : 0, s>>>(p);
    if (p.use_lds)
        uvbin_fold_kernel<<<dim3(((p.nq + 1) * p.nbins + UT - 1) / UT, kFoldGroups), UT, 0, s>>>(p.scratch, g, p.nq, p.nbins, 1,
                                                                                                  p.sums, p.counts);
    return hipGetLastError();
}

hipError_t fh_uvbin_launch_err(const UvBinParams &p0, int num_cu, hipStream_t s) {
    UvBinParams p = p0;
    const int nq = p.qty[1] ? 2 : 1;
    const size_t lds = uvbin_lds_bytes(nq, p.nbins);
    p.use_lds = lds <= 120 * 1024 && p.scratch != nullptr;
    hipError_t e = hipSuccess;
    if (p.use_lds)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(uvbin_err_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds);
    if (e != hipSuccess) return e;
    const int threads = (p.use_lds && lds > 40 * 1024) ? 1024 : UT;
    int g = p.use_lds ? uvbin_blocks(p.n, num_cu, (size_t)nq * p.nbins) : grid_for(p.n, num_cu);
    if (threads == 1024 && g > num_cu) g = num_cu;
    uvbin_err_kernel<<<g, threads, p.use_lds ? lds : 0, s>>>(p);
    if (p.use_lds)
        uvbin_fold_kernel<<<dim3((nq * p.nbins + UT - 1) / UT, kFoldGroups), UT, 0, s>>>(p.scratch, g, nq, p.nbins, 0, p.sums,
                                                                                         nullptr);
    return hipGetLastError();
}

hipError_t fh_uvbin_launch_lookup(const double *uv, int64_t n, double bin_width, int nbins, int *out, int num_cu,
                                  hipStream_t s) {
    uvbin_lookup_kernel<<<grid_for(n, num_cu), UT, 0, s>>>(uv, n, 1 / bin_width, bin_width, nbins, out);
    return hipGetLastError();
}
