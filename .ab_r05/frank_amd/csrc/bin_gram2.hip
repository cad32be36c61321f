// K1 `bin_gram` v2: the Bessel design block is GENERATED ON THE MATRIX PIPE, then the weighted Gram as before.
//
// Replaces the chunk loop of VisibilityMapping.map_visibilities (statistical_models.py:165-218) with
// DHT.coefficients (hankel.py:201-202) fused in; the deprojection pre-pass (geometry.py:69-79, 111-131) is
// deproject_kernel of bin_gram.hip.
//
// v1 evaluated J0(s_i j_k) with ~55 fp64 VALU instructions per 64 entries; on MI355X fp64 VALU and fp64 MFMA share
// the DP units, and the J0 temporaries forced the 190-tile triangle into two workgroup specialisations that each
// re-evaluated most columns (1.9x the J0 work): matrix pipe 46 % busy, 26.6 ms per 1e7 visibilities at N = 300.
//
// v2 (j0_buckets.h): visibilities are sorted into buckets of s = q/Qmax of width Delta = 2h/j_N, h = 1/4.  Inside
// bucket b, with tau = (s - s0_b)/(Delta/2) in [-1, 1],
//     sqrt(w) J0(s j_k) = sum_{n<12} [sqrt(w) tau^n] * C_b[n][k],     C_b[n][k] = a_n(s0_b j_k) (j_k Delta/2)^n,
// so a 16-visibility x 16-column tile of the design block is P (16 x 12) times C_b (12 x 16): THREE
// v_mfma_f64_16x16x4_f64.  Register r of the result (C/D layout: column = lane & 15, row = (lane >> 4) + 4 r) is
// exactly the A/B operand fragment of Gram k-step r (rows 4r .. 4r+3), so the tile goes to LDS in the layout the
// consumers read, no transposition.  19 column blocks x 3 MFMAs per 16 visibilities against 190 x 4 for the Gram
// (+7.5 %); the J0 VALU work is gone, and with it the register pressure: ONE workgroup specialisation holds all
// 190 tiles (8 waves x 24 tiles x 8 accumulator registers, two waves per SIMD) and nothing is evaluated twice.
//
//   bucket_hist / bucket_scan / bucket_starts / bucket_scatter   stable (deterministic) counting sort of the K1a
//                          output by bucket; bucket starts aligned to 16 rows (padding rows have sqrt(w) = 0);
//                          32 B per sorted row: tau, sqrt(w), sqrt(w) Re V', -
//   bin_gram2_kernel       per 16-row chunk: generate the chunk's tiles (each wave 2-3 column blocks) into the other
//                          LDS buffer while the Gram MFMAs of the current chunk run; one barrier per chunk.
//   bucket_compress        (v3) the rows of a bucket enter the Gram only through 12 x 12 moments: with P_i = sqrt(w_i) [1, tau_i,
//                          .., tau_i^11], sum_i X_i^T X_i = C_b^T (sum_i P_i^T P_i) C_b = C_b^T H_b C_b, H_b[n][m] = sum_i w_i
//                          tau_i^(n+m).  The data column rides along: the 13 x 13 matrix [[H, nu], [nu^T, eta]], nu_n = sum_i
//                          w_i V_i tau_i^n, eta = sum_i w_i V_i^2, is the Gram matrix of (1, tau, .., tau^11, V) -- positive
//                          semi-definite -- and its Cholesky factor R (R^T R = H_aug up to a backward error of a few ulp of
//                          its entries, whatever its condition) gives 13 VIRTUAL rows per bucket that the Gram kernel
//                          cannot tell from visibilities: row r has P = R[r][0..11] and the data column R[r][12].  One
//                          16-row chunk per bucket instead of one per 16 visibilities: 1e7 visibilities in ~1.6e4
//                          buckets are 39x fewer chunks.  Buckets of <= 16 rows keep their rows.
//   N > 303                the triangle is cut into row-aligned PARTS of <= 192 tiles; a part's workgroups generate
//                          only the column blocks its tiles touch (cheap now), so the fused path reaches N = 511.
#include <hip/hip_runtime.h>

#include <utility>

#include "j0_buckets.h"
#include "k1v2_tiles.h"
#include "kernels.h"

typedef double v4f64 __attribute__((ext_vector_type(4)));

namespace {

#ifndef K1_STAGGER
#define K1_STAGGER 1
#endif
constexpr int kWaves = 8;
constexpr int kThreads = 64 * kWaves;
constexpr int kCap = 24;              // tiles per wave (kTiles* tables)
constexpr int kRows = 16;             // rows per chunk = one generated tile = 4 Gram k-steps
constexpr int kTerms = FH_K1_TERMS;   // 12
constexpr int kRun = 48;              // chunks per dynamic hand-out

constexpr int xstride(int NBT) {  // LDS row stride in doubles, == 16 (mod 32): conflict-free fragment reads / writes
    return (NBT * 16) % 32 == 16 ? NBT * 16 : NBT * 16 + 16;
}
constexpr int ntiles(int NBT) { return NBT * (NBT + 1) / 2; }
// NBT = 32 (N <= 511): table (12 rows) + two buffers of design rows would need 186 KB of LDS; with ONE buffer the chunk
// is generated behind the Gram k-steps of its predecessor instead of beside them (two barriers per chunk)
constexpr bool double_buffered(int NBT) { return NBT <= 24; }
constexpr int nparts(int NBT) { return NBT <= 19 ? 1 : (NBT == 24 ? 2 : 3); }
constexpr int wave_tile(int NBT, int P, int W, int T) {
    return NBT == 4    ? kTiles4[0][W][T]
           : NBT == 8  ? kTiles8[0][W][T]
           : NBT == 13 ? kTiles13[0][W][T]
           : NBT == 19 ? kTiles19[0][W][T]
           : NBT == 24 ? kTiles24[P < 2 ? P : 0][W][T]
                       : kTiles32[P < 3 ? P : 0][W][T];
}
constexpr int tile_I(int NBT, int t) {
    int I = 0;
    while (t >= NBT - I) {
        t -= NBT - I;
        ++I;
    }
    return I;
}
constexpr int tile_J(int NBT, int t) {
    int I = 0;
    while (t >= NBT - I) {
        t -= NBT - I;
        ++I;
    }
    return I + t;
}
constexpr int wave_ntiles(int NBT, int P, int W) {
    int n = 0;
    while (n < kCap && wave_tile(NBT, P, W, n) >= 0) ++n;
    return n;
}
// first / one-past-last tile (row-major triangle numbering) of a part: parts are contiguous, row-aligned
constexpr int part_tile0(int NBT, int P) {
    int m = 1 << 30;
    for (int W = 0; W < kWaves; ++W)
        for (int T = 0; T < kCap; ++T) {
            const int t = wave_tile(NBT, P, W, T);
            if (t >= 0 && t < m) m = t;
        }
    return m;
}
constexpr int part_tile1(int NBT, int P) {
    int m = -1;
    for (int W = 0; W < kWaves; ++W)
        for (int T = 0; T < kCap; ++T) {
            const int t = wave_tile(NBT, P, W, T);
            if (t > m) m = t;
        }
    return m + 1;
}
constexpr int part_block0(int NBT, int P) { return tile_I(NBT, part_tile0(NBT, P)); }  // first column block needed
constexpr bool wave_needs(int NBT, int P, int W, int B) {
    for (int T = 0; T < kCap; ++T) {
        const int t = wave_tile(NBT, P, W, T);
        if (t >= 0 && (tile_I(NBT, t) == B || tile_J(NBT, t) == B)) return true;
    }
    return false;
}

// ---- counting sort by bucket ------------------------------------------------------------------------------------
__device__ __forceinline__ int bucket_of(double s, double inv_delta, int nb) {
    int b = (int)(s * inv_delta);  // s >= 0
    return b < nb - 1 ? b : nb - 1;
}

// hist[block][bucket]: rows i = block * 256 + tid + k * grid * 256 (the SAME mapping as bucket_scatter_kernel)
__global__ __launch_bounds__(256) void bucket_hist_kernel(const double *s, int64_t n, double inv_delta, int nb, int *hist) {
    extern __shared__ int lh[];
    for (int b = threadIdx.x; b < nb; b += 256) lh[b] = 0;
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        atomicAdd(&lh[bucket_of(s[i], inv_delta, nb)], 1);
    __syncthreads();
    int *row = hist + (size_t)blockIdx.x * nb;
    for (int b = threadIdx.x; b < nb; b += 256) row[b] = lh[b];
}

// per bucket: exclusive prefix over the blocks (in place) and the bucket's total.  A workgroup takes 64 buckets (one per
// lane: the rows of `hist` are read 256 bytes at a time) and its 16 waves 16 contiguous groups of blocks: sum of the group,
// the sums of the groups before it through LDS, then the group's prefixes.  (One thread per bucket over all 512 blocks --
// 8 000 threads on the whole device -- took 60 us of the 0.7 ms binning pass.)
__global__ __launch_bounds__(1024) void bucket_scan_blocks_kernel(int *hist, int nblocks, int nb, int *totals) {
    __shared__ int part[16][64];
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int b = blockIdx.x * 64 + lane;
    const int per = (nblocks + 15) / 16, k0 = min(nblocks, g * per), k1 = min(nblocks, k0 + per);
    int sum = 0;
    if (b < nb) {
#pragma unroll 8
        for (int k = k0; k < k1; ++k) sum += hist[(size_t)k * nb + b];
    }
    part[g][lane] = sum;
    __syncthreads();
    int run = 0;
    for (int h = 0; h < g; ++h) run += part[h][lane];
    if (b >= nb) return;
#pragma unroll 8
    for (int k = k0; k < k1; ++k) {
        const int v = hist[(size_t)k * nb + b];
        hist[(size_t)k * nb + b] = run;
        run += v;
    }
    if (g == 15) totals[b] = run;
}

struct Row32 {
    double tau, sw, swV, pad;
};

// starts[b] = first sorted row of bucket b (a multiple of 16), starts[nb] = padded length; info[0] = chunks;
// the padding rows behind each bucket are zeroed.  One workgroup of 1024 threads.
__global__ __launch_bounds__(1024) void bucket_starts_kernel(const int *totals, int nb, int *starts, int *info, Row32 *rows) {
    __shared__ int part[1024];
    const int t = threadIdx.x;
    const int per = (nb + 1023) / 1024;
    const int b0 = t * per, b1 = min(nb, b0 + per);
    int sum = 0;
    for (int b = b0; b < b1; ++b) sum += (totals[b] + 15) & ~15;
    part[t] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {  // Hillis-Steele inclusive scan
        const int v = t >= off ? part[t - off] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = part[t] - sum;
    for (int b = b0; b < b1; ++b) {
        starts[b] = run;
        const int tot = totals[b], pad = (tot + 15) & ~15;
        for (int r = tot; r < pad; ++r) rows[(size_t)run + r] = Row32{0.0, 0.0, 0.0, 0.0};
        run += pad;
    }
    if (t == 1023) {
        starts[nb] = part[1023];
        info[0] = part[1023] / kRows;
    }
}

// bucket of every 16-row chunk (binary search in starts; empty buckets have equal starts)
__global__ void chunk_bucket_kernel(const int *starts, int nb, const int *info, int *chunk_bucket) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= info[0]) return;
    const int row = c * kRows;
    int lo = 0, hi = nb;  // largest b with starts[b] <= row and starts[b + 1] > row
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (starts[mid] <= row) lo = mid;
        else hi = mid;
    }
    chunk_bucket[c] = lo;
}

// Stable scatter: a row's position is bucket start + rows of that bucket in earlier blocks (hist, scanned) + rows of
// that bucket earlier in this block, counted in (iteration, wave, lane) order -- no atomics, the same position in
// every run.  Per 64 rows the lanes of equal bucket find each other with one ballot per bucket-index bit.
__global__ __launch_bounds__(256) void bucket_scatter_kernel(const double *s, const double *sw, const double *swV,
                                                             const double *k2, int64_t n, double inv_delta, double delta,
                                                             int nb, int nbits, const int *hist, const int *starts,
                                                             Row32 *rows) {
    extern __shared__ int cnt[];
    const int *row = hist + (size_t)blockIdx.x * nb;
    for (int b = threadIdx.x; b < nb; b += 256) cnt[b] = starts[b] + row[b];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double inv_half = 2.0 * inv_delta;
    const int64_t stride = (int64_t)gridDim.x * 256;
    const int64_t iters = (n + stride - 1) / stride;
    for (int64_t k = 0; k < iters; ++k) {
        const int64_t i = k * stride + (int64_t)blockIdx.x * 256 + threadIdx.x;
        const bool active = i < n;
        const double si = active ? s[i] : 0.0;
        const int b = bucket_of(si, inv_delta, nb);
        unsigned long long peers = __ballot(active);
        for (int bit = 0; bit < nbits; ++bit) {
            const bool one = (b >> bit) & 1;
            const unsigned long long m = __ballot(one);
            peers &= one ? m : ~m;
        }
        const int rank = __popcll(peers & ((1ull << lane) - 1ull));
        const int leader_lane = __ffsll((long long)peers) - 1;
        int base = 0;
        for (int w = 0; w < 4; ++w) {
            if (wave == w && active && rank == 0) {
                base = cnt[b];
                cnt[b] = base + __popcll(peers);
            }
            __syncthreads();
        }
        base = __shfl(base, leader_lane < 0 ? 0 : leader_lane);
        if (active) {
            double tau;
            {
#pragma clang fp contract(off)
                tau = (si - ((double)b + 0.5) * delta) * inv_half;  // fh_k1_bucket_centre, the table's expansion point
            }
            rows[(size_t)base + rank] = Row32{tau, sw[i], swV[i], k2 ? k2[i] : 0.0};  // (debris model: kz^2 of the row)
        }
    }
}

// ---- bucket compression (v3) ----------------------------------------------------------------------------------
// cidx[b] = number of non-empty buckets before b; info[1] = their total.  One workgroup of 1024 threads.
__global__ __launch_bounds__(1024) void bucket_compact_kernel(const int *totals, int nb, int *cidx, int *info) {
    __shared__ int part[1024];
    const int t = threadIdx.x;
    const int per = (nb + 1023) / 1024;
    const int b0 = t * per, b1 = min(nb, b0 + per);
    int sum = 0;
    for (int b = b0; b < b1; ++b) sum += totals[b] > 0;
    part[t] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int v = t >= off ? part[t - off] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = part[t] - sum;
    for (int b = b0; b < b1; ++b) {
        cidx[b] = run;
        run += totals[b] > 0;
    }
    if (t == 1023) info[1] = part[1023];
}

constexpr int kMom = 2 * kTerms - 1;  // moments 0 .. 22 of tau
constexpr int kMomAll = kMom + kTerms + 1;  // + nu_0 .. nu_11 + eta
// Partial moments: wave (b, part) sums the rows of the part-th slice of bucket b (slices of whole 16-row chunks; fixed
// order: lane l takes rows l, l + 64, .. of the slice, then a butterfly across the lanes) into partial[b][part][36].
// The buckets are very unequal (a (u, v) distribution piles up at short baselines): one wave per bucket took 10 ms.
__global__ __launch_bounds__(256) void bucket_moments_kernel(const Row32 *rows, const int *starts, const int *totals, int nb,
                                                             int parts, double *partial) {
    const int lane = threadIdx.x & 63;
    const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int b = wid / parts, part = wid - b * parts;
    if (b >= nb) return;
    const int tot = totals[b];
    if (tot <= 16) return;  // such a bucket keeps its rows (bucket_factor_kernel)
    const int chunks = (tot + 15) >> 4, per = (chunks + parts - 1) / parts;
    const int r0 = min(tot, part * per * 16), r1 = min(tot, (part + 1) * per * 16);
    const Row32 *rb = rows + starts[b];
    double mu[kMom], nu[kTerms], eta = 0.0;
#pragma unroll
    for (int m = 0; m < kMom; ++m) mu[m] = 0.0;
#pragma unroll
    for (int n = 0; n < kTerms; ++n) nu[n] = 0.0;
    for (int i = r0 + lane; i < r1; i += 64) {
        const double *rp = reinterpret_cast<const double *>(rb + i);
        Row32 r;
        r.tau = __builtin_nontemporal_load(rp);
        r.sw = __builtin_nontemporal_load(rp + 1);
        r.swV = __builtin_nontemporal_load(rp + 2);
        const double w = r.sw * r.sw, wv = r.sw * r.swV;
        double pw = 1.0;
#pragma unroll
        for (int m = 0; m < kMom; ++m) {
            mu[m] = fma(w, pw, mu[m]);
            if (m < kTerms) nu[m] = fma(wv, pw, nu[m]);
            pw *= r.tau;
        }
        eta = fma(r.swV, r.swV, eta);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
#pragma unroll
        for (int m = 0; m < kMom; ++m) mu[m] += __shfl_xor(mu[m], off);
#pragma unroll
        for (int n = 0; n < kTerms; ++n) nu[n] += __shfl_xor(nu[n], off);
        eta += __shfl_xor(eta, off);
    }
    if (lane == 0) {
        double *o = partial + ((size_t)b * parts + part) * kMomAll;
#pragma unroll
        for (int m = 0; m < kMom; ++m) o[m] = mu[m];
#pragma unroll
        for (int n = 0; n < kTerms; ++n) o[kMom + n] = nu[n];
        o[kMom + kTerms] = eta;
    }
}

// One wave per bucket.  <= 16 rows: the P rows of the visibilities themselves.  More: the partial moments are added in the
// order of the parts, then the Cholesky factor of the augmented moment matrix with lane c holding column c (right-looking,
// 13 steps of one broadcast, one square root and <= 12 fmas per lane).  A pivot that is not positive beyond the round-off
// of its own formation ends the factorisation of that row: its contribution is below that round-off (for a positive
// semi-definite matrix the rest of the row is bounded by the pivot).
__global__ __launch_bounds__(256) void bucket_factor_kernel(const Row32 *rows, const int *starts, const int *totals,
                                                            const int *cidx, int nb, int parts, const double *partial,
                                                            double *vrows, int *vbucket) {
    __shared__ double mom[4][kMomAll];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x * 4 + wave;
    if (b >= nb) return;
    const int tot = totals[b];
    if (tot == 0) return;
    const int c = cidx[b];
    if (lane == 0) vbucket[c] = b;
    double *out = vrows + (size_t)c * 256;
    if (tot <= 16) {
        if (lane < 16) {
            const Row32 r = rows[starts[b] + lane];  // (rows past the bucket's last one are zero rows)
            double pw = r.sw;
#pragma unroll
            for (int n = 0; n < kTerms; ++n) {
                out[lane * 16 + n] = pw;
                pw *= r.tau;
            }
            out[lane * 16 + 12] = r.swV;
            out[lane * 16 + 13] = out[lane * 16 + 14] = out[lane * 16 + 15] = 0.0;
        }
        return;
    }
    if (lane < kMomAll) {
        const double *pp = partial + (size_t)b * parts * kMomAll + lane;
        double v = 0.0;
        for (int k = 0; k < parts; ++k) v += pp[(size_t)k * kMomAll];
        mom[wave][lane] = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // column cc of the augmented matrix: H_aug[i][cc], i = 0 .. 12
    constexpr int NA = kTerms + 1;
    const int cc = lane < NA ? lane : NA - 1;
    double col[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int idx = (cc < kTerms) ? (i < kTerms ? i + cc : kMom + cc) : (i < kTerms ? kMom + i : kMom + kTerms);
        col[i] = mom[wave][idx];
    }
#pragma unroll
    for (int r = 0; r < NA; ++r) {
        const double h0 = mom[wave][r < kTerms ? 2 * r : kMom + kTerms];  // the diagonal entry before any update
        const double piv = __shfl(col[r], r);
        const bool ok = piv > 1.5e-14 * h0;
        const double inv = ok ? 1.0 / sqrt(piv) : 0.0;
        double Rrc = (cc >= r) ? col[r] * inv : 0.0;
        if (r < NA - 1 && cc == NA - 1) {  // (the data column never takes more than is left of its own diagonal entry:
                                            //  bucket_factor2_kernel, bin_prepass.hip)
            const double lim = sqrt(fmax(col[NA - 1], 0.0));
            Rrc = fmin(fmax(Rrc, -lim), lim);
        }
        if (lane < NA) out[r * 16 + lane] = Rrc;
#pragma unroll
        for (int i = r + 1; i < NA; ++i) {
            const double Rri = __shfl(Rrc, i);
            col[i] = fma(-Rri, Rrc, col[i]);
        }
    }
    // columns 13 .. 15 of rows 0 .. 12, and rows 13 .. 15
    for (int e = lane; e < 256; e += 64) {
        const int r = e >> 4, c2 = e & 15;
        if (r >= NA || c2 >= NA) out[e] = 0.0;
    }
}

// ---- K1b v2 ---------------------------------------------------------------------------------------------------
// Element type T of the design block and the Gram MFMAs.
//   double  v_mfma_f64_16x16x4_f64, 64 cycles per SIMD; the result stays in registers for the whole stream.
//   float   v_mfma_f32_16x16x4_f32, 32 cycles per SIMD (BASELINE configs[2], "fp32"): tau, the bucket centre and hence
//           the argument of J0 are still formed in fp64 (bucket sort), only the Taylor tail sum_n C[n] tau^n, the
//           design rows and the tile products are single precision; the single-precision accumulators are added into
//           the workgroup's fp64 slab every kFlush rows (block accumulation), so the sums over millions of rows are fp64.
template <typename T>
struct Mx;
template <>
struct Mx<double> {
    typedef double v4 __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ v4 mfma(double a, double b, v4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    // C/D layout: column = lane & 15, row = (lane >> 4) + 4 reg
    static __device__ __forceinline__ int row_of(int kk, int reg) { return kk + 4 * reg; }
};
template <>
struct Mx<float> {
    typedef float v4 __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ v4 mfma(float a, float b, v4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    // C/D layout: column = lane & 15, row = 4 (lane >> 4) + reg.  Register r as an A/B operand of a Gram k-step then
    // pairs rows {r, 4 + r, 8 + r, 12 + r}: any partition of the 16 rows into k-steps gives the same Gram.
    static __device__ __forceinline__ int row_of(int kk, int reg) { return 4 * kk + reg; }
};
constexpr int kFlush = 64;  // fp32: chunks (of 16 rows) between two additions into the fp64 slab

template <typename T, int NBT, int P, int W, int Tt>
__device__ __forceinline__ void mfma_one(typename Mx<T>::v4 &acc, const T (&f)[NBT]) {
    constexpr int tt = wave_tile(NBT, P, W, Tt);
    constexpr int I = tile_I(NBT, tt), J = tile_J(NBT, tt);
    // A[i][k] = Xt[k][16I+i] and B[k][j] = Xt[k][16J+j] share one fragment layout: lane -> (k = lane>>4, i|j = lane&15)
    acc = Mx<T>::mfma(f[I], f[J], acc);
}
template <typename T, int NBT, int P, int W, int TPW, int... Ts>
__device__ __forceinline__ void mfma_all(typename Mx<T>::v4 (&acc)[TPW > 0 ? TPW : 1], const T (&f)[NBT],
                                         std::integer_sequence<int, Ts...>) {
    (mfma_one<T, NBT, P, W, Ts>(acc[Ts], f), ...);
}
template <typename T, int NBT, int P, int W, int B>
__device__ __forceinline__ void load_frag(T (&f)[NBT], const T *xb) {
    if constexpr (wave_needs(NBT, P, W, B)) f[B] = xb[B * 16];
}
template <typename T, int NBT, int P, int W, int... Bs>
__device__ __forceinline__ void load_frags(T (&f)[NBT], const T *xb, std::integer_sequence<int, Bs...>) {
    (load_frag<T, NBT, P, W, Bs>(f, xb), ...);
}

// DEB: vis_model = 'debris' (statistical_models.py:494-496): every generated entry is further scaled by
// exp(-kz_i^2 H2[k]) (kz^2 travels in the fourth slot of the sorted row, H2 sits in LDS, zero beyond column N - 1)
template <typename T, int NBT, int P, int W, bool DEB, bool VR>
__device__ __forceinline__ void wave_main(const Bin2Params &p, double *smem, int part_block, int part_nblocks) {
    typedef typename Mx<T>::v4 v4;
    constexpr bool kF32 = sizeof(T) == 4;
    constexpr int XS = xstride(NBT);
    constexpr int T0 = part_tile0(NBT, P), T1 = part_tile1(NBT, P);
    constexpr int NTP = T1 - T0;
    constexpr int TPW = wave_ntiles(NBT, P, W);
    constexpr int TPWA = TPW > 0 ? TPW : 1;
    constexpr int B0 = part_block0(NBT, P);            // first column block this part needs
    constexpr int NGEN = (NBT - B0 - W + kWaves - 1) / kWaves > 0 ? (NBT - B0 - W + kWaves - 1) / kWaves : 0;  // blocks B0+W+8g

    T *Ctab = reinterpret_cast<T *>(smem);  // [kTerms][XS]
    constexpr bool kDB = double_buffered(NBT);
    T *X = Ctab + kTerms * XS;              // [2][kRows][XS] (one buffer when !kDB)
    int *sq = reinterpret_cast<int *>(X + (kDB ? 2 : 1) * kRows * XS);  // [4] run queue (dynamic hand-out)
    double *H2s = reinterpret_cast<double *>(sq + 4);                   // [NBT * 16] (DEB only)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int kk = lane >> 4, ii = lane & 15;
    const int N = p.N;
    const int JN = N >> 4, jn = N & 15;     // the data column sqrt(w) Re V' lives at column N
    const int nchunks = p.info[0];

    v4 acc[TPWA];
#pragma unroll
    for (int t = 0; t < TPW; ++t) acc[t] = v4{0, 0, 0, 0};
    double *slab = p.partials[P] + (size_t)part_block * NTP * 256;
    // fp32: add the single-precision accumulators into the fp64 slab ([tile][lane][reg]: 32 contiguous bytes per lane)
    // and clear them; `first`: the slab holds nothing yet
    auto flush = [&](bool first) {
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            double *sp = slab + ((size_t)(wave_tile(NBT, P, W, t) - T0) * 64 + lane) * 4;
            double o[4] = {0.0, 0.0, 0.0, 0.0};
            if (!first) {
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = sp[r];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                sp[r] = o[r] + (double)acc[t][r];
                acc[t][r] = 0;
            }
        }
    };
    bool flushed = false;
    int since_flush = 0;

    // ---- rows of one chunk: lane (kk, ii) holds row ii ----------------------------------------------------------
    struct RowRegs {
        double tau, sw, swV, k2;
    };
    auto load_row = [&](int chunk) -> RowRegs {
        if constexpr (VR) {  // compressed rows: the A operands themselves (tau, sw, k2 carry k-steps 0, 1, 2) and the data column
            const double *rp = p.rows + ((size_t)chunk * kRows + ii) * 16;
            return RowRegs{rp[kk], rp[4 + kk], rp[12], rp[8 + kk]};
        } else {
            const double *rp = p.rows + ((size_t)chunk * kRows + ii) * 4;
            return RowRegs{rp[0], rp[1], rp[2], DEB ? rp[3] : 0.0};
        }
    };
    if constexpr (DEB) {
        for (int c = threadIdx.x; c < NBT * 16; c += kThreads) H2s[c] = c < p.N ? p.H2[c] : 0.0;
        __syncthreads();
    }
    // ---- the Taylor table of one bucket into LDS (all threads) ---------------------------------------------------
    auto load_table = [&](int bucket) {
        if constexpr (kF32) {
            const float *src = p.table32 + (size_t)bucket * kTerms * XS;
            for (int e = tid; e < kTerms * XS; e += kThreads) Ctab[e] = src[e];
        } else {
            const double *src = p.table + (size_t)bucket * kTerms * XS;
            for (int e = tid; e < kTerms * XS; e += kThreads) Ctab[e] = src[e];
        }
    };
    // ---- generate this wave's column blocks of one chunk into X[xbuf] -------------------------------------------
    // A operand P[i][n] = sqrt(w_i) tau_i^n : lane (k = kk, i = ii) of k-step t holds n = 4t + kk
    auto gen_one = [&](const RowRegs &r, int xbuf, int J, T a0, T a1, T a2) {
        const T *cb = Ctab + kk * XS + J * 16 + ii;
        v4 d = v4{0, 0, 0, 0};
        d = Mx<T>::mfma(a2, cb[8 * XS], d);  // smallest terms first
        d = Mx<T>::mfma(a1, cb[4 * XS], d);
        d = Mx<T>::mfma(a0, cb[0], d);
        if constexpr (DEB) {
            const double h2 = H2s[J * 16 + ii];
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
#pragma clang fp contract(off)
                const double kz2 = __shfl(r.k2, Mx<T>::row_of(kk, reg));
                d[reg] = (T)((double)d[reg] * exp(-(kz2 * h2)));
            }
        }
        if (J == JN) {  // column N: sqrt(w) Re V' of the row this register holds; columns beyond: table zeros
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const double v = __shfl(r.swV, Mx<T>::row_of(kk, reg));  // lane q < 16 holds row q
                if (ii == jn) d[reg] = (T)v;
            }
        }
        // register `reg` goes where Gram k-step `reg` reads its fragment: position 4 reg + kk of the chunk
        T *xw = X + ((size_t)xbuf * kRows + kk) * XS + J * 16 + ii;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) xw[reg * 4 * XS] = d[reg];
    };
    auto powers = [&](const RowRegs &r, T &a0, T &a1, T &a2) {
        if constexpr (VR) {
            a0 = (T)r.tau;
            a1 = (T)r.sw;
            a2 = (T)r.k2;
            return;
        }
        const double t2 = r.tau * r.tau;
        const double pk = kk == 0 ? 1.0 : (kk == 1 ? r.tau : (kk == 2 ? t2 : t2 * r.tau));
        const double t4 = t2 * t2;
        const double b0 = r.sw * pk, b1 = b0 * t4;
        a0 = (T)b0;
        a1 = (T)b1;
        a2 = (T)(b1 * t4);
    };
    // ---- Gram MFMAs of k-step ks (positions 4 ks .. 4 ks + 3) of X[xbuf] -------------------------------------------
    auto gram = [&](int xbuf, int ks) {
        const T *xb = X + ((size_t)xbuf * kRows + ks * 4 + kk) * XS + ii;
        T f[NBT];
        load_frags<T, NBT, P, W>(f, xb, std::make_integer_sequence<int, NBT>{});
        mfma_all<T, NBT, P, W, TPW>(acc, f, std::make_integer_sequence<int, TPW>{});
    };

    // ---- work hand-out: contiguous chunk ranges (static, bitwise reproducible) or runs from an atomic counter ----
    int *counter = p.work_counter ? p.work_counter + P : nullptr;
    constexpr int run = VR ? 4 : kRun;  // (compressed: ~1.6e4 chunks in all)
    int c0, c1;
    if (!counter) {
        const int per = (nchunks + part_nblocks - 1) / part_nblocks;
        c0 = part_block * per;
        c1 = min(nchunks, c0 + per);
    } else {
        if (tid == 0) sq[0] = atomicAdd(counter, 1);
        __syncthreads();
        c0 = sq[0] * run;
        c1 = min(nchunks, c0 + run);
    }
    int qslot = 0;
    int cur_bucket = -1;
    while (c0 < c1) {
        if (counter && tid == 0) sq[qslot ^ 1] = atomicAdd(counter, 1);  // the run after this one
        // prologue of a range: table + first chunk.  Row scalars and bucket ids are fetched TWO chunks ahead of their
        // use (a dependent global load at the top of every chunk would stall the wave for a microsecond of each five)
        RowRegs rnext = load_row(c0);
        int bnext = c0 + 1 < c1 ? p.chunk_bucket[c0 + 1] : 0;  // bucket of the chunk generated in the first iteration
        {
            const int b = p.chunk_bucket[c0];
            if (b != cur_bucket) {
                __syncthreads();
                load_table(b);
                cur_bucket = b;
            }
            __syncthreads();
            T a0, a1, a2;
            powers(rnext, a0, a1, a2);
#pragma unroll
            for (int g = 0; g < NGEN; ++g) gen_one(rnext, 0, B0 + W + kWaves * g, a0, a1, a2);
            if (c0 + 1 < c1) rnext = load_row(c0 + 1);
            __syncthreads();
        }
        int xbuf = 0;
#pragma unroll 1
        for (int c = c0; c < c1; ++c) {
            const bool more = c + 1 < c1;
            const RowRegs rgen = rnext;
            const int bgen = bnext;
            if (c + 2 < c1) {
                rnext = load_row(c + 2);
                bnext = p.chunk_bucket[c + 2];
            }
            if (more && bgen != cur_bucket) {  // uniform: nobody reads Ctab between the closing barrier of a chunk and here
                load_table(bgen);
                cur_bucket = bgen;
                __syncthreads();
            }
            T a0 = 0, a1 = 0, a2 = 0;
            if (more) powers(rgen, a0, a1, a2);
            static_assert(NGEN <= 4, "a wave generates at most four column blocks per chunk");
            if constexpr (kDB) {
                // this wave's generated blocks of chunk c+1 (matrix pipe + LDS writes) between the Gram k-steps of chunk c
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    // K1_STAGGER: the second wave of each SIMD (W >= 4) generates behind its Gram k-steps instead of in
                    // front of them, so that the two waves of a SIMD are not in the same phase of the chunk
                    const int g = (K1_STAGGER && W >= 4) ? ks - (4 - NGEN) : ks;
                    if (!(K1_STAGGER && W >= 4)) {
                        if (g >= 0 && g < NGEN && more) gen_one(rgen, xbuf ^ 1, B0 + W + kWaves * g, a0, a1, a2);
                        gram(xbuf, ks);
                    } else {
                        gram(xbuf, K1_STAGGER == 2 ? (ks + 2) & 3 : ks);
                        if (g >= 0 && g < NGEN && more) gen_one(rgen, xbuf ^ 1, B0 + W + kWaves * g, a0, a1, a2);
                    }
                }
            } else {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) gram(0, ks);
                __syncthreads();  // everybody has read chunk c: its buffer takes chunk c+1
                if (more) {
#pragma unroll
                    for (int g = 0; g < NGEN; ++g) gen_one(rgen, 0, B0 + W + kWaves * g, a0, a1, a2);
                }
            }
            if constexpr (kF32) {
                if (++since_flush == kFlush) {
                    flush(!flushed);
                    flushed = true;
                    since_flush = 0;
                }
            }
            __syncthreads();
            if constexpr (kDB) xbuf ^= 1;
        }
        if (!counter) break;
        c0 = sq[qslot ^ 1] * run;
        c1 = min(nchunks, c0 + run);
        qslot ^= 1;
    }

    // ---- write this workgroup's partial tiles ------------------------------------------------------------------------
    if constexpr (kF32) {
        flush(!flushed);  // slab[part_block][tile - T0][lane][reg], fp64
    } else {              // slab[part_block][tile - T0][reg][lane]
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            const int tl = wave_tile(NBT, P, W, t) - T0;
#pragma unroll
            for (int r = 0; r < 4; ++r) slab[(size_t)tl * 256 + r * 64 + lane] = acc[t][r];
        }
    }
}

template <typename T, int NBT, int P, bool DEB, bool VR>
__device__ __forceinline__ void part_main(const Bin2Params &p, double *smem, int part_block, int part_nblocks) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    switch (wave) {
        case 0: wave_main<T, NBT, P, 0, DEB, VR>(p, smem, part_block, part_nblocks); break;
        case 1: wave_main<T, NBT, P, 1, DEB, VR>(p, smem, part_block, part_nblocks); break;
        case 2: wave_main<T, NBT, P, 2, DEB, VR>(p, smem, part_block, part_nblocks); break;
        case 3: wave_main<T, NBT, P, 3, DEB, VR>(p, smem, part_block, part_nblocks); break;
        case 4: wave_main<T, NBT, P, 4, DEB, VR>(p, smem, part_block, part_nblocks); break;
        case 5: wave_main<T, NBT, P, 5, DEB, VR>(p, smem, part_block, part_nblocks); break;
        case 6: wave_main<T, NBT, P, 6, DEB, VR>(p, smem, part_block, part_nblocks); break;
        default: wave_main<T, NBT, P, 7, DEB, VR>(p, smem, part_block, part_nblocks); break;
    }
}

template <typename T, int NBT, bool DEB, bool VR = false>
__global__ __launch_bounds__(kThreads, 2) void bin_gram2_kernel(Bin2Params p) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int b = blockIdx.x;
    if (nparts(NBT) == 1 || b < p.part_blocks[0]) {
        part_main<T, NBT, 0, DEB, VR>(p, smem, b, p.part_blocks[0]);
    } else if (nparts(NBT) == 2 || b < p.part_blocks[0] + p.part_blocks[1]) {
        part_main<T, NBT, nparts(NBT) >= 2 ? 1 : 0, DEB, VR>(p, smem, b - p.part_blocks[0], p.part_blocks[1]);
    } else {
        part_main<T, NBT, nparts(NBT) >= 3 ? 2 : 0, DEB, VR>(p, smem, b - p.part_blocks[0] - p.part_blocks[1], p.part_blocks[2]);
    }
}

template <typename T, int NBT>
constexpr size_t bin2_smem_bytes() {  // (+ 8-byte alignment slack and the debris model's H2 row)
    return sizeof(T) * ((size_t)kTerms * xstride(NBT) + (double_buffered(NBT) ? 2 : 1) * kRows * xstride(NBT)) + 4 * sizeof(int) +
           8 + sizeof(double) * NBT * 16;
}

template <typename T, int NBT, bool DEB, bool VR = false>
hipError_t launch_bin2(const Bin2Params &p, hipStream_t stream) {
    constexpr size_t smem = bin2_smem_bytes<T, NBT>();
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&bin_gram2_kernel<T, NBT, DEB, VR>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return e;
    const int grid = p.part_blocks[0] + p.part_blocks[1] + p.part_blocks[2];
    hipLaunchKernelGGL((bin_gram2_kernel<T, NBT, DEB, VR>), dim3(grid), dim3(kThreads), smem, stream, p);
    return hipGetLastError();
}

// The fp32 kernel leaves its fp64 slabs as [tile][lane][reg] in the C/D layout of v_mfma_f32_16x16x4_f32 (row = 4 (lane >> 4)
// + reg, column = lane & 15); the reduction and finalize kernels expect [tile][reg'][lane'] in the fp64 layout (row =
// (lane' >> 4) + 4 reg').  Element (kk, ii, reg) -> row 4 kk + reg -> reg' = kk, lane' = 16 reg + ii.  In place per tile.
__global__ __launch_bounds__(256) void slab_relayout_kernel(double *slab, size_t ntile_slabs) {
    __shared__ double t[256];
    for (size_t s = blockIdx.x; s < ntile_slabs; s += gridDim.x) {
        double *sp = slab + s * 256;
        const int lane = threadIdx.x >> 2, reg = threadIdx.x & 3;
        const int kk = lane >> 4, ii = lane & 15;
        t[kk * 64 + reg * 16 + ii] = sp[threadIdx.x];  // element (lane, reg) sits at lane * 4 + reg
        __syncthreads();
        sp[threadIdx.x] = t[threadIdx.x];
        __syncthreads();
    }
}

// ---- predict_visibilities through the same tables (statistical_models.py:279-329) --------------------------------------
// V_i = sum_k H[i,k] I_k = sum_k pref_k scale I_k J0(s_i j_k); inside bucket b, J0(s j_k) = sum_n C_b[n][k] tau^n, so
//     V_i = sum_n tau_i^n c_b[n],        c_b[n] = sum_k C_b[n][k] (pref_k scale I_k)      -- 12 numbers per bucket.
// The per-bucket coefficients are one small matrix-vector product over the tables; a visibility then costs a degree-11
// polynomial instead of N Bessel evaluations: the pass is bound by reading q and writing V (16 B per visibility).
__global__ __launch_bounds__(256) void predict_bucket_coef_kernel(const double *table, int XS, int N, int nb, const double *pref,
                                                                  const double *I, double scale, double *coef) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;  // row = b * 12 + n
    if (row >= nb * kTerms) return;
    const double *t = table + (size_t)row * XS;
    double a = 0.0;
    for (int k = lane; k < N; k += 64) a = fma(t[k], (pref[k] * I[k]) * scale, a);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) a += __shfl_down(a, off);
    if (lane == 0) coef[row] = a;
}
__global__ __launch_bounds__(256) void predict_taylor_kernel(const double *q, int64_t n, double inv_Q, double inv_delta,
                                                             double delta, int nb, const double *coef, double *V) {
    const double inv_half = 2.0 * inv_delta;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double s = inv_Q * fabs(q[i]);  // J0 is even: the reference and the direct kernel take q of either sign
        const int b = bucket_of(s, inv_delta, nb);
        double tau;
        {
#pragma clang fp contract(off)
            tau = (s - ((double)b + 0.5) * delta) * inv_half;
        }
        const double *c = coef + (size_t)b * kTerms;
        double a = c[kTerms - 1];
#pragma unroll
        for (int m = kTerms - 2; m >= 0; --m) a = fma(a, tau, c[m]);
        V[i] = a;
    }
}
__global__ void max_abs_kernel(const double *q, int64_t n, double *out) {  // out[0] = max |q|, one workgroup of 1024
    __shared__ double red[16];
    double m = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 1024) m = fmax(m, fabs(q[i]));
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmax(m, __shfl_down(m, off));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w) m = fmax(m, red[w]);
        out[0] = m;
    }
}

}  // namespace

hipError_t fh_k1v2_launch_max(const double *q, int64_t n, double *out, hipStream_t stream) {
    hipLaunchKernelGGL(max_abs_kernel, dim3(1), dim3(1024), 0, stream, q, n, out);
    return hipGetLastError();
}
hipError_t fh_k1v2_launch_predict_coef(const double *table, int XS, int N, int nb, const double *pref, const double *I, double scale,
                                       double *coef, hipStream_t stream) {
    hipLaunchKernelGGL(predict_bucket_coef_kernel, dim3((nb * kTerms + 3) / 4), dim3(256), 0, stream, table, XS, N, nb, pref, I,
                       scale, coef);
    return hipGetLastError();
}
hipError_t fh_k1v2_launch_predict(const double *table, int XS, int N, int nb, const double *pref, const double *I, double scale,
                                  double *coef, const double *q, int64_t n, double inv_Q, double delta, double *V,
                                  hipStream_t stream) {
    hipLaunchKernelGGL(predict_bucket_coef_kernel, dim3((nb * kTerms + 3) / 4), dim3(256), 0, stream, table, XS, N, nb, pref, I,
                       scale, coef);
    int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(predict_taylor_kernel, dim3(grid), dim3(256), 0, stream, q, n, inv_Q, 1.0 / delta, delta, nb, coef, V);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
static int nbt_for(int N) {
    const int nb = (N + 1 + 15) / 16;
    if (nb <= 4) return 4;
    if (nb <= 8) return 8;
    if (nb <= 13) return 13;
    if (nb <= 19) return 19;
    if (nb <= 24) return 24;
    if (nb <= 32) return 32;
    return 0;  // N > 511
}
int fh_k1v2_nbt_for(int N) { return nbt_for(N); }
int fh_k1v2_xstride(int NBT) { return xstride(NBT); }
int fh_k1v2_ntiles(int NBT) { return ntiles(NBT); }
int fh_k1v2_nparts(int NBT) { return nparts(NBT); }
int fh_k1v2_part_tile0(int NBT, int P) {
    switch (NBT) {
        case 4: return part_tile0(4, 0);
        case 8: return part_tile0(8, 0);
        case 13: return part_tile0(13, 0);
        case 19: return part_tile0(19, 0);
        case 24: return P == 0 ? part_tile0(24, 0) : part_tile0(24, 1);
        case 32: return P == 0 ? part_tile0(32, 0) : (P == 1 ? part_tile0(32, 1) : part_tile0(32, 2));
    }
    return 0;
}
int fh_k1v2_part_ntiles(int NBT, int P) {
    switch (NBT) {
        case 4: return part_tile1(4, 0) - part_tile0(4, 0);
        case 8: return part_tile1(8, 0) - part_tile0(8, 0);
        case 13: return part_tile1(13, 0) - part_tile0(13, 0);
        case 19: return part_tile1(19, 0) - part_tile0(19, 0);
        case 24: return P == 0 ? part_tile1(24, 0) - part_tile0(24, 0) : part_tile1(24, 1) - part_tile0(24, 1);
        case 32:
            return P == 0 ? part_tile1(32, 0) - part_tile0(32, 0)
                          : (P == 1 ? part_tile1(32, 1) - part_tile0(32, 1) : part_tile1(32, 2) - part_tile0(32, 2));
    }
    return 0;
}
int fh_k1v2_part_block0(int NBT, int P) {
    switch (NBT) {
        case 24: return P == 0 ? part_block0(24, 0) : part_block0(24, 1);
        case 32: return P == 0 ? part_block0(32, 0) : (P == 1 ? part_block0(32, 1) : part_block0(32, 2));
    }
    return 0;
}

hipError_t fh_k1v2_launch_sort(const SortParams &sp, hipStream_t stream) {
    const size_t lds = sizeof(int) * (size_t)sp.nb;
    int nbits = 0;
    while ((1 << nbits) < sp.nb) ++nbits;
    hipLaunchKernelGGL(bucket_hist_kernel, dim3(sp.blocks), dim3(256), lds, stream, sp.s, sp.n, sp.inv_delta, sp.nb, sp.hist);
    hipLaunchKernelGGL(bucket_scan_blocks_kernel, dim3((sp.nb + 63) / 64), dim3(1024), 0, stream, sp.hist, sp.blocks, sp.nb,
                       sp.totals);
    hipLaunchKernelGGL(bucket_starts_kernel, dim3(1), dim3(1024), 0, stream, sp.totals, sp.nb, sp.starts, sp.info,
                       reinterpret_cast<Row32 *>(sp.rows));
    hipLaunchKernelGGL(bucket_scatter_kernel, dim3(sp.blocks), dim3(256), lds, stream, sp.s, sp.sw, sp.swV, sp.k2, sp.n,
                       sp.inv_delta, sp.delta, sp.nb, nbits, sp.hist, sp.starts, reinterpret_cast<Row32 *>(sp.rows));
    const int64_t max_chunks = (sp.n + (int64_t)kRows * sp.nb) / kRows + 1;
    hipLaunchKernelGGL(chunk_bucket_kernel, dim3((unsigned)((max_chunks + 255) / 256)), dim3(256), 0, stream, sp.starts,
                       sp.nb, sp.info, sp.chunk_bucket);
    return hipGetLastError();
}

int fh_k1v2_moment_doubles() { return kMomAll; }
hipError_t fh_k1v2_launch_compress(const CompressParams &cp, hipStream_t stream) {
    hipLaunchKernelGGL(bucket_compact_kernel, dim3(1), dim3(1024), 0, stream, cp.totals, cp.nb, cp.cidx, cp.info);
    const long long waves = (long long)cp.nb * cp.parts;
    hipLaunchKernelGGL(bucket_moments_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, stream,
                       reinterpret_cast<const Row32 *>(cp.rows), cp.starts, cp.totals, cp.nb, cp.parts, cp.partial);
    hipLaunchKernelGGL(bucket_factor_kernel, dim3((cp.nb + 3) / 4), dim3(256), 0, stream, reinterpret_cast<const Row32 *>(cp.rows),
                       cp.starts, cp.totals, cp.cidx, cp.nb, cp.parts, cp.partial, cp.vrows, cp.vbucket);
    return hipGetLastError();
}

hipError_t fh_k1v2_launch_bin(int NBT, const Bin2Params &p, hipStream_t stream) {
    if (p.table32 && !p.H2) {  // single-precision design block and tile products, fp64 block accumulation
        hipError_t e = hipErrorInvalidValue;
        switch (NBT) {
            case 4: e = launch_bin2<float, 4, false>(p, stream); break;
            case 8: e = launch_bin2<float, 8, false>(p, stream); break;
            case 13: e = launch_bin2<float, 13, false>(p, stream); break;
            case 19: e = launch_bin2<float, 19, false>(p, stream); break;
            case 24: e = launch_bin2<float, 24, false>(p, stream); break;
            case 32: e = launch_bin2<float, 32, false>(p, stream); break;
        }
        if (e != hipSuccess) return e;
        for (int P = 0; P < 3; ++P)
            if (p.part_blocks[P] > 0) {
                const size_t n = (size_t)p.part_blocks[P] * fh_k1v2_part_ntiles(NBT, P);
                hipLaunchKernelGGL(slab_relayout_kernel, dim3((unsigned)(n < 4096 ? n : 4096)), dim3(256), 0, stream,
                                   p.partials[P], n);
            }
        return hipGetLastError();
    }
    if (p.H2) {  // debris model: the scaled design block (fp64 only)
        switch (NBT) {
            case 4: return launch_bin2<double, 4, true>(p, stream);
            case 8: return launch_bin2<double, 8, true>(p, stream);
            case 13: return launch_bin2<double, 13, true>(p, stream);
            case 19: return launch_bin2<double, 19, true>(p, stream);
            case 24: return launch_bin2<double, 24, true>(p, stream);
            case 32: return launch_bin2<double, 32, true>(p, stream);
        }
        return hipErrorInvalidValue;
    }
    if (p.virtual_rows) {
        switch (NBT) {
            case 4: return launch_bin2<double, 4, false, true>(p, stream);
            case 8: return launch_bin2<double, 8, false, true>(p, stream);
            case 13: return launch_bin2<double, 13, false, true>(p, stream);
            case 19: return launch_bin2<double, 19, false, true>(p, stream);
            case 24: return launch_bin2<double, 24, false, true>(p, stream);
            case 32: return launch_bin2<double, 32, false, true>(p, stream);
        }
        return hipErrorInvalidValue;
    }
    switch (NBT) {
        case 4: return launch_bin2<double, 4, false>(p, stream);
        case 8: return launch_bin2<double, 8, false>(p, stream);
        case 13: return launch_bin2<double, 13, false>(p, stream);
        case 19: return launch_bin2<double, 19, false>(p, stream);
        case 24: return launch_bin2<double, 24, false>(p, stream);
        case 32: return launch_bin2<double, 32, false>(p, stream);
    }
    return hipErrorInvalidValue;
}
