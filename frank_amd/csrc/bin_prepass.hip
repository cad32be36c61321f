// K1 pre-pass of the moments path (round 3): from the visibility table to one 16 x 16 chunk of "virtual rows" per J0 bucket
// in three streaming passes and two small kernels, 1.05 GB of HBM traffic per 1e7 visibilities (round 2: 1.9 GB in eight
// kernels: deproject -> s, sqrt(w), sqrt(w) V' -> histogram -> scan -> starts -> scatter -> chunk map -> compact ->
// moments -> factor).
//
// Replaces, for the default (moments) path of VisibilityMapping.map_visibilities (statistical_models.py:109-237):
// geometry.apply_correction (geometry.py:69-79, 111-131), q = hypot(u', v') (:166), and the grouping of the rows by the
// bucket of their J0 argument (bin_gram2.hip).  The rows path (FRANK_AMD_K1=rows, the debris model, fp32 arithmetic)
// keeps deproject_kernel + the sort of bin_gram2.hip.
//
//   P1 uv_hist_kernel            reads u, v only (16 B / row): bucket of every row, one histogram per workgroup (its rows are a
//                                fixed set: tiles of 1024 rows dealt round-robin), the baseline range of the pass
//   -- bucket_scan_kernel        exclusive prefix over the workgroups per bucket; the LAST workgroup to finish lays out the
//                                sorted table (bucket starts aligned to 16 rows), numbers the non-empty buckets and counts
//                                the pieces P3 cuts the buckets into (a (u, v) distribution piles up at short baselines:
//                                buckets differ by 1e4 in size)
//   P2 deproject_scatter_kernel  reads the five columns (40 B / row), phase-centres and deprojects, and writes the row
//                                (tau, sqrt(w), sqrt(w) Re V': 24 B) at its place: bucket start + rows of the bucket in
//                                earlier workgroups + in earlier tiles, waves, lanes of this workgroup -- no atomics, two
//                                barriers per tile, the same place in every run; sum log(w / 2 pi) rides along
//   P3 piece_moments_kernel      one wave per piece (the rows of a bucket inside one 4096-row segment of the sorted table):
//                                23 moments of tau, 12 of V tau^n, sum w V^2 (16-byte loads)
//   -- bucket_factor2_kernel     per bucket: pieces added in order, Cholesky factor of the 13 x 13 moment matrix =
//                                13 virtual rows (bin_gram2.hip explains why this is exact to round-off)
//
// What a row costs: 16 + 40 + 24 + 24 = 104 B against the 40 B it holds; the sort cannot do with less than one look at
// (u, v) before it knows where a row goes.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "deproject.h"
#include "j0_buckets.h"
#include "kernels.h"

namespace {

constexpr int kTerms = FH_K1_TERMS;          // 12
constexpr int kMom = 2 * kTerms - 1;         // moments 0 .. 22 of tau
constexpr int kMomAll = kMom + kTerms + 1;   // + nu_0 .. nu_11 + eta = 36

typedef double d2 __attribute__((ext_vector_type(2)));

// ---- P1 ------------------------------------------------------------------------------------------------------------
// Workgroup k of the pass owns the tiles t = k, k + G, ... of 64 x (waves per workgroup) rows (the same in P1 and P2).
// hist[workgroup][b] = rows of bucket b among the workgroup's rows (LDS integer atomics: the counts do not depend on their order);
// partial_scalars[workgroup] = (-, qmin, qmax over rows of multiplicity > 0, qmax over all rows)
template <bool HIST>
__global__ __launch_bounds__(1024) void uv_hist_kernel(PrepassParams P) {
    extern __shared__ int lds_i[];
    __shared__ double red[3][16];
    const BinParams &p = P.bin;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    if (HIST) {
        for (int b = threadIdx.x; b < P.nb; b += blockDim.x) lds_i[b] = 0;
        __syncthreads();
    }
    double qmin = INFINITY, qmax = -INFINITY, qmax_all = -INFINITY;
    const int tile = blockDim.x * P.unroll;
    const int64_t ntiles = (p.count + tile - 1) / tile;
    for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x)
      for (int j = 0; j < P.unroll; ++j) {
        const int64_t i = t * tile + j * blockDim.x + threadIdx.x;
        if (i < p.count) {
            double u, v;
            fh_load_uv(p, p.first + i, u, v);
            const int mult = p.mult ? p.mult[p.first + i] : 1;
            const double q = fh_deproject_q_fast(p, u, v);
            qmax_all = fmax(qmax_all, q);
            if (mult > 0) {
                qmin = fmin(qmin, q);
                qmax = fmax(qmax, q);
            }
            if (HIST) atomicAdd(&lds_i[fh_bucket_of(p.inv_Qmax * q, P.inv_delta, P.nb)], 1);
        }
      }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        qmin = fmin(qmin, __shfl_down(qmin, off));
        qmax = fmax(qmax, __shfl_down(qmax, off));
        qmax_all = fmax(qmax_all, __shfl_down(qmax_all, off));
    }
    if (lane == 0) {
        red[0][wave] = qmin;
        red[1][wave] = qmax;
        red[2][wave] = qmax_all;
    }
    __syncthreads();
    if (HIST) {  // bucket-major: hist[b][workgroup]
        // (one look: the count of buckets is an upper bound, ~8 x the buckets the rows reach -- 1.45 M four-byte stores per pass at
        //  N = 300, 50 us; the buffer is cleared in front of the kernel instead, 6 MB at memory speed, and only counts are stored)
        for (int b = threadIdx.x; b < P.nb; b += blockDim.x) {
            const int v = lds_i[b];
            if (v != 0 || !P.hist_zeroed) P.hist[(size_t)b * P.hist_stride + blockIdx.x] = v;
        }
    }
    if (threadIdx.x == 0) {
        double mn = INFINITY, mx = -INFINITY, ma = -INFINITY;
        for (int w = 0; w < wpb; ++w) {
            mn = fmin(mn, red[0][w]);
            mx = fmax(mx, red[1][w]);
            ma = fmax(ma, red[2][w]);
        }
        double *ps = P.partial_scalars + (size_t)blockIdx.x * 4;
        ps[1] = mn;
        ps[2] = mx;
        ps[3] = ma;
    }
}

// ---- scan + layout -----------------------------------------------------------------------------------------------
// hist[bucket][workgroup] (bucket-major: a bucket's counts are contiguous) -> rows of the bucket in earlier workgroups (in
// place); totals[bucket].  One WAVE per bucket: a few consecutive counts per lane, a prefix inside the lane, a scan over the
// lanes.  (The first version -- workgroup-major histograms, a workgroup per 64 buckets, 16 thread groups walking down
// the rows -- was four workgroups for the 243 buckets of the bench: 24 us of dependent L2 round trips.)
// The last workgroup to arrive (ticket in info[3]) then does the serial part on the totals -- O(buckets):
//   starts[b]   first sorted row of bucket b (multiple of 16; the <= 15 padding rows behind a bucket are zeroed),
//   cidx[b]     number of non-empty buckets before b; info[1] their total,
//   piece0[b]   first slot of the bucket's partial moments: P3 cuts the SORTED TABLE into segments of seg_rows rows, bucket b
//               meets the segments starts[b] / seg_rows .. (starts[b] + tot - 1) / seg_rows, one slot each.
__device__ __forceinline__ int wave_inclusive_scan(int v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(v, off);
        if (lane >= off) v += o;
    }
    return v;
}
// exclusive prefix of `v` over the 1024 threads of the workgroup (thread order); *total = the sum.  Two barriers.
__device__ __forceinline__ int block_exclusive_scan(int v, int *wsum /* [16] */, int *total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int inc = wave_inclusive_scan(v, lane);
    __syncthreads();  // (wsum may still be read from the scan before)
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int before = 0, tot = 0;
    for (int w = 0; w < 16; ++w) {
        const int x = wsum[w];
        before += w < wave ? x : 0;
        tot += x;
    }
    *total = tot;
    return before + inc - v;
}
__global__ __launch_bounds__(1024) void bucket_scan_kernel(PrepassParams P) {
    __shared__ int wsum[16];
    __shared__ int last;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nb = P.nb, G = P.blocks;
    {
        const int b = blockIdx.x * 16 + wave;
        if (b < nb) {
            int *row = P.hist + (size_t)b * P.hist_stride;
            // lane l holds the counts of workgroups [per l, per (l + 1)): per = a multiple of 4, rows are padded to it
            const int per = P.hist_stride / 64;
            int run = 0;
            for (int k = 0; k < per; ++k) run += lane * per + k < G ? row[lane * per + k] : 0;  // (the padding holds whatever)
            const int inc = wave_inclusive_scan(run, lane);
            int pre = inc - run;
            for (int k = 0; k < per; ++k) {
                if (lane * per + k < G) {
                    const int v = row[lane * per + k];
                    row[lane * per + k] = pre;
                    pre += v;
                }
            }
            if (lane == 63) P.totals[b] = inc;
        }
    }
    // (one fence per workgroup: on this multi-die part an agent-scope release writes the XCD's L2 back -- with every one of the
    //  16 x 1024 threads fencing, the kernel took 25 us; the barrier orders the other threads' stores before thread 0's fence.
    //  The last workgroup reads the totals with agent-scope atomic loads, which do not go through its L1.)
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        last = atomicAdd(&P.info[3], 1) == (int)gridDim.x - 1;
    }
    __syncthreads();
    if (!last) return;
    const int t = threadIdx.x;
    const int per = (nb + 1023) / 1024;
    const int b0 = min(nb, t * per), b1 = min(nb, b0 + per);
    const int seg = P.seg_rows;
    // pass 1: padded rows and non-empty buckets of this thread's range
    int s_rows = 0, s_ne = 0;
    for (int b = b0; b < b1; ++b) {
        const int tot = __hip_atomic_load(&P.totals[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (other workgroups' writes)
        s_rows += (tot + 15) & ~15;
        s_ne += tot > 0;
    }
    int tot_rows, tot_ne, tot_pc;
    const int e_rows = block_exclusive_scan(s_rows, wsum, &tot_rows);
    const int e_ne = block_exclusive_scan(s_ne, wsum, &tot_ne);
    // pass 2: starts, indices, padding; the pieces of each bucket (their number needs the bucket's start)
    int r_rows = e_rows, r_ne = e_ne, s_pc = 0;
    for (int b = b0; b < b1; ++b) {
        const int tot = __hip_atomic_load(&P.totals[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int pad = (tot + 15) & ~15;
        P.starts[b] = r_rows;
        P.cidx[b] = r_ne;
        for (int r = tot; r < pad; ++r) {
            double *z = P.rows + ((size_t)r_rows + r) * 3;
            z[0] = z[1] = z[2] = 0.0;
        }
        s_pc += tot > 16 ? (r_rows + tot - 1) / seg - r_rows / seg + 1 : 0;
        r_rows += pad;
        r_ne += tot > 0;
    }
    int r_pc = block_exclusive_scan(s_pc, wsum, &tot_pc);
    r_rows = e_rows;
    for (int b = b0; b < b1; ++b) {
        const int tot = __hip_atomic_load(&P.totals[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        P.piece0[b] = r_pc;
        r_pc += tot > 16 ? (r_rows + tot - 1) / seg - r_rows / seg + 1 : 0;
        r_rows += (tot + 15) & ~15;
    }
    if (t == 1023) {
        P.starts[nb] = tot_rows;
        P.piece0[nb] = tot_pc;
        P.info[0] = tot_rows / 16;
        P.info[1] = tot_ne;
        P.info[2] = (tot_rows + seg - 1) / seg;  // segments of the sorted table
        P.info[3] = 0;                             // the ticket of the next pass
    }
}

// ---- P2 ------------------------------------------------------------------------------------------------------------
// Stable scatter without atomics.  A workgroup owns the same tiles of rows as in P1 and keeps ONE write front per bucket
// (cnt[b] in LDS, starting at bucket start + rows of the bucket in earlier workgroups): 256 fronts x the buckets are a few MB of
// partly written lines, which the L2 merges (one front per WAVE was 16 x as many: 1.8 x the bytes went to memory).  Inside a
// tile a row's place is front + rows of its bucket in earlier waves of the tile + its rank among the rows of its bucket in its
// own wave (an LDS atomic add on the wave's counter wc[wave][b], below); after one barrier every lane adds up the earlier waves'
// counters; after a second one the first row of a bucket in the tile advances the front.  The same place in every run.
// The per-row arithmetic is the fast set of deproject.h (the pass was half bound by fp64 vector instructions); the sum of
// log(w / 2 pi) (statistical_models.py:218) is the log of a running product -- mantissa and exponent kept apart, one
// logarithm per lane at the end instead of one per row, and a smaller rounding error than the sum of the logarithms.
// MULT: bootstrap multiplicities (fh_vis_set_multiplicity); F32: the table is stored in single precision; SAFE: phases beyond
// 1e5 rad may occur (the host bounds them by (|dRA| + |dDec|) qmax / cos(inc)), the library's sincos takes them; U: rows
// per lane and tile (a tile = 64 U rows per wave; wave w of the workgroup holds rows [64 U w, 64 U (w + 1)) of it).
// The loop over the full tiles is straight-line -- no lane is ever idle in it, the last, partial tile is peeled off -- so
// that the compiler counts its outstanding memory operations exactly: the next tile's loads are waited for with the current
// tile's stores still in flight.
// Rank of a row among the rows of its bucket in its wave: the value an LDS atomic add on the wave's counter of that bucket
// returns.  Lanes that meet in one instruction are served in a fixed order, and the U instructions of a tile in program
// order, so the ranks -- and with them the place of every row -- are the same in every run.
template <bool MULT, bool F32, bool SAFE, int U>
__global__ __launch_bounds__(1024) void deproject_scatter_kernel(PrepassParams P) {
    extern __shared__ int lds_i[];  // cnt[nb], wc[wpb][nb]
    __shared__ double red[16];
    const BinParams &p = P.bin;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const int nb = P.nb;
    int *cnt = lds_i, *wc = lds_i + nb;
    {
        for (int b = threadIdx.x; b < nb; b += blockDim.x) cnt[b] = P.starts[b] + P.hist[(size_t)b * P.hist_stride + blockIdx.x];
        for (int e = threadIdx.x; e < wpb * nb; e += blockDim.x) wc[e] = 0;
    }
    int *wcw = wc + wave * nb;
    const double inv_half = 2.0 * P.inv_delta;
    double sum_logw = 0.0;      // rows with multiplicities: the plain sum
    double pm = 1.0;            // mantissa of the product of the weights, in [0.5, 1)
    int pe = 0, pn = 0;         // its exponent; the number of rows in it
    double w_prev = -1.0, sw_prev = 0.0;
    const int tile = blockDim.x * U;
    const int64_t ntiles = (p.count + tile - 1) / tile, nfull = p.count / tile;
    const int64_t last = p.first + p.count - 1;
    // (columns that do not exist are read from one that does and then ignored)
    const double *colVim = p.Vim ? p.Vim : p.Vre;
    const float *colVim32 = p.Vim32 ? p.Vim32 : p.Vre32;
    const bool has_im = F32 ? p.Vim32 != nullptr : p.Vim != nullptr;
    VisRow r[U];
    int rmult[U];
    auto fetch = [&](int64_t tt) {
#pragma unroll
        for (int j = 0; j < U; ++j) {
            int64_t g = p.first + tt * tile + (wave * U + j) * 64 + lane;
            g = g < last ? g : last;
            if (p.count == 0) g = 0;  // (an empty range: a table holds at least one element)
            const int64_t gw = p.w_scalar ? 0 : g;
            if (F32) {
                r[j].u = (double)p.u32[g];
                r[j].v = (double)p.v32[g];
                r[j].Vre = (double)p.Vre32[g];
                r[j].Vim = (double)colVim32[g];
                r[j].w = (double)p.w32[gw];
            } else {
                r[j].u = __builtin_nontemporal_load(&p.u[g]);
                r[j].v = __builtin_nontemporal_load(&p.v[g]);
                r[j].Vre = __builtin_nontemporal_load(&p.Vre[g]);
                r[j].Vim = __builtin_nontemporal_load(&colVim[g]);
                r[j].w = __builtin_nontemporal_load(&p.w[gw]);
            }
            if (MULT) rmult[j] = p.mult[g];
        }
    };
    // one tile; FULL: every lane holds a row of the table
    auto process = [&](int64_t t, auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        VisRow c[U];
        double cm[U];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            c[j] = r[j];
            cm[j] = MULT ? (double)rmult[j] : 1.0;
        }
        fetch(t + gridDim.x);  // the next tile's rows are in flight during this tile's arithmetic and barriers
        double tau[U], sw[U], swV[U];
        int bk[U], rank[U];
        bool active[U];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            active[j] = FULL || t * tile + (wave * U + j) * 64 + lane < p.count;
            const double Vim = has_im ? c[j].Vim : 0.0;
            const double re = SAFE ? fh_phase_centre_re(p, c[j].u, c[j].v, c[j].Vre, Vim)
                                   : fh_phase_centre_re_fast(p, c[j].u, c[j].v, c[j].Vre, Vim);
            const double q = fh_deproject_q_fast(p, c[j].u, c[j].v);
            const double wj = MULT ? cm[j] * c[j].w : c[j].w;
            if (__all(wj == w_prev)) {  // (tables of constant weight: the square root of the row before)
                sw[j] = sw_prev;
            } else {
                sw[j] = sqrt(wj);
                w_prev = wj;
                sw_prev = sw[j];
            }
            const double s = p.inv_Qmax * q;  // k * q, hankel.py:189,202
            swV[j] = sw[j] * re;
            if (MULT) {
                if (active[j] && cm[j] > 0.0) sum_logw += cm[j] * log(c[j].w / (2 * M_PI));  // statistical_models.py:218
            } else {
                // mantissa and exponent of w by integer arithmetic; anything but a positive normal number takes frexp
                const double w1 = active[j] ? c[j].w : 1.0;
                const unsigned long long bits = __double_as_longlong(w1);
                const int ex = (int)((bits >> 52) & 0x7ff);
                double m1 = __longlong_as_double((bits & 0x800fffffffffffffull) | 0x3fe0000000000000ull);
                int e1 = ex - 1022;
                if (__builtin_expect(__any(ex == 0 || ex == 0x7ff || (long long)bits < 0), 0)) {
                    m1 = frexp(w1, &e1);
                    if (!(w1 > 0.0)) m1 = w1 == 0.0 ? 0.0 : NAN;  // log(0) = -inf, log(negative) = NaN, as the reference's sum
                    if (ex == 0x7ff && w1 > 0.0) m1 = INFINITY;   // log(+inf) = +inf (frexp returns inf: the mantissa mask below
                                                                  // would have turned the product into a finite number)
                }
                pm *= m1;  // in [0.25, 1): back to [0.5, 1)
                const unsigned long long pb = __double_as_longlong(pm);
                const int e2 = (int)((pb >> 52) & 0x7ff) - 1022;  // 0 or -1 (0, NaN: whatever, the product stays what it is)
                if (pm > 0.0 && pm < INFINITY) {
                    pm = __longlong_as_double((pb & 0x800fffffffffffffull) | 0x3fe0000000000000ull);
                    pe += e1 + e2;
                }
                pn += active[j] ? 1 : 0;
            }
            bk[j] = fh_bucket_of(s, P.inv_delta, nb);
            tau[j] = fh_bucket_tau(s, bk[j], P.delta, inv_half);
            rank[j] = 0;
            if (active[j]) rank[j] = atomicAdd(&wcw[bk[j]], 1);  // (ds_add_rtn_u32)
        }
        __syncthreads();
        int base[U], before[U], total[U];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            before[j] = total[j] = 0;
            for (int w = 0; w < wpb; ++w) {
                const int v = wc[w * nb + bk[j]];
                total[j] += v;
                before[j] += w < wave ? v : 0;
            }
            base[j] = cnt[bk[j]] + before[j];
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < U; ++j) {
            // the first row of a bucket in the tile advances the workgroup's front; the first in a wave clears the wave's counter
            // (with U > 1 the same bucket may come up again at j + 1: its rank is then > 0)
            if (active[j] && rank[j] == 0) {
                if (before[j] == 0) cnt[bk[j]] += total[j];
                wcw[bk[j]] = 0;
            }
        }
#pragma unroll
        for (int j = 0; j < U; ++j) {
            // 24-byte rows: the 16-byte aligned pair in one store, the third double in another (no branch)
            const size_t row = active[j] ? (size_t)base[j] + rank[j] : (size_t)P.dummy_row;
            double *o = P.rows + row * 3;
            const bool odd = row & 1;
            *reinterpret_cast<d2 *>(o + (odd ? 1 : 0)) = odd ? d2{sw[j], swV[j]} : d2{tau[j], sw[j]};
            o[odd ? 0 : 2] = odd ? tau[j] : swV[j];
        }
    };
    fetch(blockIdx.x);
    __syncthreads();
    int64_t t = blockIdx.x;
    for (; t < nfull; t += gridDim.x) process(t, std::true_type{});
    if (t < ntiles) process(t, std::false_type{});  // (at most one workgroup)
    if (!MULT) sum_logw = (log(pm) + (double)pe * M_LN2) - (double)pn * log(2 * M_PI);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sum_logw += __shfl_down(sum_logw, off);
    if (lane == 0) red[wave] = sum_logw;
    __syncthreads();
    if (threadIdx.x == 0) {
        double sacc = 0.0;
        for (int w = 0; w < wpb; ++w) sacc += red[w];
        P.partial_scalars[(size_t)blockIdx.x * 4] = sacc;
    }
}

// ---- P3 ------------------------------------------------------------------------------------------------------------
// One wave per PIECE = the rows of one bucket inside one segment of the sorted table (seg_rows rows; a bucket of more than
// 16 rows has piece0[b + 1] - piece0[b] of them, small buckets one).  The wave finds its bucket by bisection of piece0 (staged
// in LDS); lane l takes rows 2l, 2l + 1 of every group of 128 -- three 16-byte loads --, then a butterfly over the lanes;
// fixed order, same bits in every run.
__global__ __launch_bounds__(256) void piece_moments_kernel(PrepassParams P) {
    extern __shared__ int p0s[];  // piece0[0 .. nb]
    const int nb = P.nb;
    for (int b = threadIdx.x; b <= nb; b += 256) p0s[b] = P.piece0[b];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int pc = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pc >= p0s[nb]) return;
    int lo = 0, hi = nb;  // the bucket with piece0[b] <= pc < piece0[b + 1] (buckets without pieces have equal neighbours)
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (p0s[mid] <= pc) lo = mid;
        else hi = mid;
    }
    const int b = lo, k = pc - p0s[b];
    const int tot = P.totals[b], st = P.starts[b], seg = P.seg_rows;
    const int sg = st / seg + k;
    const int r0 = max(sg * seg, st), r1 = min((sg + 1) * seg, st + tot);  // sorted rows of the piece (r0 a multiple of 16)
    const double *rb = P.rows;
    double mu[kMom], nu[kTerms], eta = 0.0;
#pragma unroll
    for (int m = 0; m < kMom; ++m) mu[m] = 0.0;
#pragma unroll
    for (int n = 0; n < kTerms; ++n) nu[n] = 0.0;
    // A group of 128 rows is 3 KB.  Lane l wants rows 2l, 2l + 1 = bytes 48 l .. 48 l + 47, but three 16-byte loads at a stride of
    // 48 bytes make every load instruction touch all 24 lines of the group: three times the L2 requests (the kernel sat at the
    // request rate of the L2, ~1e11 per second, at 3.4 TB/s).  So the wave loads the group as three fully coalesced 1 KB
    // pieces (lane l: bytes 16 l of each), passes them through its own 3 KB of LDS and reads its 48 bytes back.
    // The next group's loads are in flight while this one's 120 multiply-adds run.
    __shared__ __attribute__((aligned(16))) d2 stage[4][192];
    d2 *lds3 = stage[threadIdx.x >> 6];
    d2 n0 = d2{0.0, 0.0}, n1 = n0, n2 = n0;
    const int r0a = r0;  // (a multiple of 16 rows: 384 bytes)
    auto issue = [&](int g0) {  // rows g0 .. g0 + 127; whole 16-byte pieces inside the sorted table (its tail is padded)
        const d2 *gp = reinterpret_cast<const d2 *>(rb + (size_t)g0 * 3) + lane;
        n0 = __builtin_nontemporal_load(gp);
        n1 = __builtin_nontemporal_load(gp + 64);
        n2 = __builtin_nontemporal_load(gp + 128);
    };
    int g0 = r0a;
    if (g0 < r1) issue(g0);
    for (; g0 < r1; g0 += 128) {
        lds3[lane] = n0;
        lds3[64 + lane] = n1;
        lds3[128 + lane] = n2;
        if (g0 + 128 < r1) issue(g0 + 128);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const d2 x0 = lds3[3 * lane], x1 = lds3[3 * lane + 1], x2 = lds3[3 * lane + 2];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int i = g0 + 2 * lane;
        const bool one = i < r1, two = i + 1 < r1;  // (what lies behind the piece may be the next bucket's rows)
        const double tau[2] = {one ? x0.x : 0.0, two ? x1.y : 0.0}, sw[2] = {one ? x0.y : 0.0, two ? x2.x : 0.0},
                     swV[2] = {one ? x1.x : 0.0, two ? x2.y : 0.0};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const double w = sw[h] * sw[h], wv = sw[h] * swV[h];
            double pw = 1.0;
#pragma unroll
            for (int m = 0; m < kMom; ++m) {
                mu[m] = fma(w, pw, mu[m]);
                if (m < kTerms) nu[m] = fma(wv, pw, nu[m]);
                pw *= tau[h];
            }
            eta = fma(swV[h], swV[h], eta);
        }
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
        double *o = P.partial + (size_t)pc * kMomAll;
#pragma unroll
        for (int m = 0; m < kMom; ++m) o[m] = mu[m];
#pragma unroll
        for (int n = 0; n < kTerms; ++n) o[kMom + n] = nu[n];
        o[kMom + kTerms] = eta;
    }
}

// One workgroup per bucket.  <= 16 rows: the P rows of the visibilities themselves.  More: the pieces' moments are added in a
// fixed order (wave g of sixteen takes pieces g, g + 16, ... eight loads at a time; then a fixed tree), then wave 0 forms the
// Cholesky factor of the augmented moment matrix with lane c holding column c (right-looking, 13 steps of one broadcast, one
// square root and <= 12 fmas per lane).  A pivot that is not positive beyond the round-off of its own formation ends the
// factorisation of that row: its contribution is below that round-off (for a positive semi-definite matrix the rest of the
// row is bounded by the pivot).
__global__ __launch_bounds__(1024) void bucket_factor2_kernel(PrepassParams P) {
    __shared__ double psum[16][kMomAll];
    __shared__ double mom[kMomAll];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x;
    const int tot = P.totals[b];
    if (tot == 0) return;
    const int c = P.cidx[b];
    if (threadIdx.x == 0) P.vbucket[c] = b;
    double *out = P.vrows + (size_t)c * 256;
    if (tot <= 16 && !P.fused) {
        if (threadIdx.x < 16) {
            const double *rp = P.rows + ((size_t)P.starts[b] + lane) * 3;  // (rows past the bucket's last one are zero rows)
            const double tau = rp[0], sw = rp[1], swV = rp[2];
            double pw = sw;
#pragma unroll
            for (int n = 0; n < kTerms; ++n) {
                out[lane * 16 + n] = pw;
                pw *= tau;
            }
            out[lane * 16 + 12] = swV;
            out[lane * 16 + 13] = out[lane * 16 + 14] = out[lane * 16 + 15] = 0.0;
        }
        return;
    }
    const int s0 = P.piece0[b], ns = P.piece0[b + 1] - s0;
    if (lane < kMomAll) {  // wave g takes pieces g, g + 16, ..: eight loads at a time
        const double *pp = P.partial + (size_t)s0 * kMomAll + lane;
        double a = 0.0;
        int k = wave;
        for (; k + 16 * 7 < ns; k += 16 * 8) {
            double x[8];
#pragma unroll
            for (int h = 0; h < 8; ++h) x[h] = pp[(size_t)(k + 16 * h) * kMomAll];
#pragma unroll
            for (int h = 0; h < 8; ++h) a += x[h];
        }
        for (; k < ns; k += 16) a += pp[(size_t)k * kMomAll];
        psum[wave][lane] = a;
    }
    __syncthreads();
    if (wave != 0) return;
    if (lane < kMomAll) {
        double t[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) t[q] = (psum[4 * q][lane] + psum[4 * q + 1][lane]) + (psum[4 * q + 2][lane] + psum[4 * q + 3][lane]);
        mom[lane] = (t[0] + t[1]) + (t[2] + t[3]);
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
        col[i] = mom[idx];
    }
#pragma unroll
    for (int r = 0; r < NA; ++r) {
        const double h0 = mom[r < kTerms ? 2 * r : kMom + kTerms];  // the diagonal entry before any update
        const double piv = __shfl(col[r], r);
        const bool ok = piv > 1.5e-14 * h0;
        const double inv = ok ? 1.0 / sqrt(piv) : 0.0;
        double Rrc = (cc >= r) ? col[r] * inv : 0.0;
        // The data column of a row whose pivot is all but cancelled: for a positive semi-definite matrix |R[r][12]|^2 cannot exceed
        // what is left of the data column's own diagonal entry; an entry beyond that is the round-off of H[r][12] divided by the
        // square root of a pivot of ~1e-14 H_rr.  Left alone it drove the last pivot negative -- dropped, i.e. taken as zero -- and
        // the bucket's sum of w V^2 came out too LARGE by what had been subtracted once too often (M and j do not notice: the
        // row's other entries are ~sqrt(pivot)): H0 off by 1.8e-3 at N = 38 and 5e-6 at N = 57 for one table, right at 507 other
        // sizes (tools/size_sweep_binning.py).  The clamp is inactive for every entry that obeys the bound: same bits elsewhere.
        if (r < NA - 1 && cc == NA - 1) {
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

// ---- Gram of the virtual rows ----------------------------------------------------------------------------------------------
// G = sum_b X_b^T X_b over the non-empty buckets, X_b = R_b C_b: the 16 virtual rows of bucket b (bucket_factor2_kernel: P in
// columns 0 .. 11, the data column in 12) times the bucket's Taylor table (12 x N).  A few hundred to a few thousand chunks
// of 16 rows: bin_gram2_kernel (built to stream 1e7 rows: every workgroup holds all 190 tiles and writes a slab of them)
// needed 85 us on 32 workgroups or 100 MB of slabs on 256.  Here a workgroup owns ONE output tile (I, J) and its waves split
// the chunks: per chunk a wave generates the two 16 x 16 blocks X_I, X_J it needs -- three matrix instructions each; register r
// of the result is the operand fragment of Gram k-step r, so nothing goes through LDS -- and adds X_I^T X_J in four more.
// No slabs: the waves' tiles are added in LDS in wave order, the `split` workgroups of a tile through `scratch` in
// vr_finish_kernel.  Same arithmetic per chunk as bin_gram2_kernel<.., VR>; the order over the chunks is fixed.
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(1024) void vr_gram_kernel(VrGramParams G) {
    __shared__ double red[16][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const int kk = lane >> 4, ii = lane & 15;
    const int t = blockIdx.x / G.split, sp = blockIdx.x - t * G.split;
    int I = 0, tt = t;
    while (tt >= G.NBT - I) {
        tt -= G.NBT - I;
        ++I;
    }
    const int J = I + tt;
    const int JN = G.N >> 4, jn = G.N & 15;  // the data column sqrt(w) Re V' lives at column N
    const int nchunks = G.info[0];
    const int stride = nwaves * G.split;
    v4d acc = v4d{0, 0, 0, 0};
    // operands of one chunk: 9 doubles per lane (+ 4 of the data column for the tiles of the last block column); the next
    // chunk's are in flight while this one's ten matrix instructions run, its bucket id one chunk further ahead
    struct Ops {
        double a0, a1, a2, i0, i1, i2, j0, j1, j2, dc[4];
    };
    auto load_ops = [&](int c, int b) {
        Ops o;
        const double *rp = G.vrows + (size_t)c * 256 + ii * 16;
        const double *cb = G.table + ((size_t)b * kTerms + kk) * G.XS + ii;
        const double *cI = cb + I * 16, *cJ = cb + J * 16;
        o.a0 = rp[kk];
        o.a1 = rp[4 + kk];
        o.a2 = rp[8 + kk];
        o.i0 = cI[0];
        o.i1 = cI[4 * G.XS];
        o.i2 = cI[8 * G.XS];
        o.j0 = cJ[0];
        o.j1 = cJ[4 * G.XS];
        o.j2 = cJ[8 * G.XS];
        if (J == JN) {
            const double *dc = G.vrows + (size_t)c * 256 + 12;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) o.dc[reg] = dc[(kk + 4 * reg) * 16];
        }
        return o;
    };
    int c = sp * nwaves + wave;
    Ops nxt{};
    int bn = 0;
    if (c < nchunks) nxt = load_ops(c, G.vbucket[c]);
    if (c + stride < nchunks) bn = G.vbucket[c + stride];
    for (; c < nchunks; c += stride) {
        const Ops o = nxt;
        if (c + stride < nchunks) nxt = load_ops(c + stride, bn);
        if (c + 2 * stride < nchunks) bn = G.vbucket[c + 2 * stride];
        v4d dI = v4d{0, 0, 0, 0}, dJ = v4d{0, 0, 0, 0};
        dI = __builtin_amdgcn_mfma_f64_16x16x4f64(o.a2, o.i2, dI, 0, 0, 0);  // smallest terms first
        dJ = __builtin_amdgcn_mfma_f64_16x16x4f64(o.a2, o.j2, dJ, 0, 0, 0);
        dI = __builtin_amdgcn_mfma_f64_16x16x4f64(o.a1, o.i1, dI, 0, 0, 0);
        dJ = __builtin_amdgcn_mfma_f64_16x16x4f64(o.a1, o.j1, dJ, 0, 0, 0);
        dI = __builtin_amdgcn_mfma_f64_16x16x4f64(o.a0, o.i0, dI, 0, 0, 0);
        dJ = __builtin_amdgcn_mfma_f64_16x16x4f64(o.a0, o.j0, dJ, 0, 0, 0);
        if (J == JN) {  // column N: the data column of the row this register holds (row kk + 4 reg); columns beyond: table zeros
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                if (ii == jn) {
                    dJ[reg] = o.dc[reg];
                    if (I == JN) dI[reg] = o.dc[reg];
                }
            }
        }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(dI[reg], dJ[reg], acc, 0, 0, 0);
    }
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) red[wave][reg * 64 + lane] = acc[reg];
    __syncthreads();
    if (threadIdx.x < 256) {
        double sum = 0.0;
        for (int w = 0; w < nwaves; ++w) sum += red[w][threadIdx.x];
        G.scratch[((size_t)sp * G.ntiles + t) * 256 + threadIdx.x] = sum;
    }
}

// stats_sum += the splits of every tile in order; workgroup 0 also folds the per-workgroup scalars of the pre-pass
// (sum log(w / 2 pi), baseline range): strided partial sums, then a fixed tree
__global__ __launch_bounds__(256) void vr_finish_kernel(VrGramParams G, double *stats_sum, double *stats_minmax) {
    const int t = threadIdx.x;
    const size_t e = (size_t)blockIdx.x * 256 + t, ne = (size_t)G.ntiles * 256;
    double sum = 0.0;
    for (int k = 0; k < G.split; ++k) sum += G.scratch[(size_t)k * ne + e];
    stats_sum[e] = (G.fresh ? 0.0 : stats_sum[e]) + sum;  // (0.0 + sum: the bits of an addition to zeroed memory)
    if (blockIdx.x == 0) {
        __shared__ double rs[256], rmn[256], rmx[256];
        double sl = 0.0, mn = INFINITY, mx = -INFINITY;
        for (int b = t; b < G.scalar_blocks; b += 256) {
            sl += G.partial_scalars[b * 4 + 0];
            mn = fmin(mn, G.partial_scalars[b * 4 + 1]);
            mx = fmax(mx, G.partial_scalars[b * 4 + 2]);
        }
        rs[t] = sl;
        rmn[t] = mn;
        rmx[t] = mx;
        __syncthreads();
        for (int h = 128; h >= 1; h >>= 1) {
            if (t < h) {
                rs[t] += rs[t + h];
                rmn[t] = fmin(rmn[t], rmn[t + h]);
                rmx[t] = fmax(rmx[t], rmx[t + h]);
            }
            __syncthreads();
        }
        if (t == 0) {
            stats_sum[ne] = (G.fresh ? 0.0 : stats_sum[ne]) + rs[0];
            if (G.fresh) stats_sum[ne + 1] = 0.0;
            // min/max are kept as (-qmin, qmax) so that one max-all-reduce serves both (a fresh pair is NaN: fmax(NaN, x) = x)
            stats_minmax[0] = G.fresh ? fmax(__builtin_nan(""), -rmn[0]) : fmax(stats_minmax[0], -rmn[0]);
            stats_minmax[1] = G.fresh ? fmax(__builtin_nan(""), rmx[0]) : fmax(stats_minmax[1], rmx[0]);
        }
    }
}

}  // namespace

int fh_prepass_moment_doubles() { return kMomAll; }

// Geometry of P1 / P2 for nb buckets on a device of num_cu compute units: waves per workgroup, workgroups.  P2 keeps
// (waves + 1) counters per bucket in LDS; workgroups of eight waves, two resident per compute unit (one's barriers and stores beside
// the other's arithmetic: 154 against 189 us per 1e7 rows for one of sixteen) while that fits 72 KB each -- 2 048 buckets --,
// fewer waves per workgroup beyond (one at 9 000 buckets and more: 128 KB at the 16 000 the sort admits).
void fh_prepass_geometry(int nb, int num_cu, int *wpb, int *blocks) {
    int w = 8;
    const size_t n = (size_t)(nb > 0 ? nb : 1);
    while (w > 1 && (size_t)(w + 1) * n * sizeof(int) > (w == 1 ? 144 : 72) * 1024) w >>= 1;
    *wpb = w;
    // (round 5: THREE workgroups of eight waves per compute unit's worth, not two -- the same 0.27 ms per 1e7 rows with the device
    //  to itself, but a pipeline's binning passes find ~110 units free beside the resident fit loops, where 512 workgroups run in
    //  three rounds the last of which fills a third of them: 1 400-1 412 against 1 368-1 378 fits/s at steady state)
    *blocks = (num_cu > 0 ? num_cu : 256) * (w == 8 ? 3 : 16 / w);
}

int64_t fh_prepass_max_pieces(int64_t count, int nb, int seg_rows) { return count / seg_rows + 2 * (int64_t)nb + 2; }

hipError_t fh_prepass_launch_range(const PrepassParams &P, hipStream_t stream) {
    PrepassParams Q = P;
    Q.nb = 0;
    hipLaunchKernelGGL(uv_hist_kernel<false>, dim3(P.blocks), dim3(64 * P.wpb), 0, stream, Q);
    return hipGetLastError();
}

template <bool MULT, bool F32, bool SAFE, int U>
static hipError_t launch_scatter_t(const PrepassParams &P, size_t lds, hipStream_t stream) {
    if (lds > 48 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&deproject_scatter_kernel<MULT, F32, SAFE, U>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((deproject_scatter_kernel<MULT, F32, SAFE, U>), dim3(P.blocks), dim3(64 * P.wpb), lds, stream, P);
    return hipGetLastError();
}
template <int U>
static hipError_t launch_scatter_u(const PrepassParams &P, size_t lds, hipStream_t stream) {
    const bool mult = P.bin.mult != nullptr, f32 = P.bin.u32 != nullptr, safe = P.safe_trig != 0;
    const int sel = (mult ? 4 : 0) | (f32 ? 2 : 0) | (safe ? 1 : 0);
    switch (sel) {
        case 0: return launch_scatter_t<false, false, false, U>(P, lds, stream);
        case 1: return launch_scatter_t<false, false, true, U>(P, lds, stream);
        case 2: return launch_scatter_t<false, true, false, U>(P, lds, stream);
        case 3: return launch_scatter_t<false, true, true, U>(P, lds, stream);
        case 4: return launch_scatter_t<true, false, false, U>(P, lds, stream);
        case 5: return launch_scatter_t<true, false, true, U>(P, lds, stream);
        case 6: return launch_scatter_t<true, true, false, U>(P, lds, stream);
        default: return launch_scatter_t<true, true, true, U>(P, lds, stream);
    }
}
static hipError_t launch_scatter(const PrepassParams &P, size_t lds, hipStream_t stream) {
    return P.unroll == 2 ? launch_scatter_u<2>(P, lds, stream) : launch_scatter_u<1>(P, lds, stream);
}

hipError_t fh_prepass_launch(const PrepassParams &P, hipStream_t stream, int skip_hist) {
    const size_t lds1 = sizeof(int) * (size_t)P.nb, lds2 = sizeof(int) * (size_t)P.nb * (P.wpb + 1);
    const size_t lds3 = sizeof(int) * ((size_t)P.nb + 1);
    hipError_t e;
    if (lds1 > 48 * 1024) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&uv_hist_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds1);
        if (e != hipSuccess) return e;
    }
    if (lds3 > 48 * 1024) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&piece_moments_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds3);
        if (e != hipSuccess) return e;
    }
    if (skip_hist == 0) hipLaunchKernelGGL(uv_hist_kernel<true>, dim3(P.blocks), dim3(64 * P.wpb), lds1, stream, P);
    if (skip_hist != 1) hipLaunchKernelGGL(bucket_scan_kernel, dim3((P.nb + 15) / 16), dim3(1024), 0, stream, P);  // (2: the histograms exist)
    e = launch_scatter(P, lds2, stream);
    if (e != hipSuccess) return e;
    const int64_t max_pieces = fh_prepass_max_pieces(P.bin.count, P.nb, P.seg_rows);
    hipLaunchKernelGGL(piece_moments_kernel, dim3((unsigned)((max_pieces + 3) / 4)), dim3(256), lds3, stream, P);
    hipLaunchKernelGGL(bucket_factor2_kernel, dim3(P.nb), dim3(1024), 0, stream, P);
    return hipGetLastError();
}

// the pieces of fh_prepass_launch the fused form (bin_fused.hip) uses: P1 + scan, and the factorisation of the buckets' moments
hipError_t fh_prepass_launch_hist(const PrepassParams &P, hipStream_t stream) {
    const size_t lds1 = sizeof(int) * (size_t)P.nb;
    if (lds1 > 48 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&uv_hist_kernel<true>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(uv_hist_kernel<true>, dim3(P.blocks), dim3(64 * P.wpb), lds1, stream, P);
    hipLaunchKernelGGL(bucket_scan_kernel, dim3((P.nb + 15) / 16), dim3(1024), 0, stream, P);
    return hipGetLastError();
}
// (a kernel of this library's own: hipMemsetAsync of a few MB held the calling thread until the stream had caught up, 0.25 ms per
//  step of a pipeline)
__global__ __launch_bounds__(256) void zero_ints_kernel(int4 *p, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) p[i] = make_int4(0, 0, 0, 0);
}
hipError_t fh_prepass_launch_zero(int *p, size_t n, hipStream_t stream) {  // n a multiple of 4, p 16-byte aligned
    hipLaunchKernelGGL(zero_ints_kernel, dim3(512), dim3(256), 0, stream, reinterpret_cast<int4 *>(p), n / 4);
    return hipGetLastError();
}
// P1 alone with a bucket count that is an upper bound (the look at (u, v) that gives the range AND the histograms, round 6)
hipError_t fh_prepass_launch_look(const PrepassParams &P, hipStream_t stream) {
    const size_t lds1 = sizeof(int) * (size_t)P.nb;
    if (lds1 > 48 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&uv_hist_kernel<true>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(uv_hist_kernel<true>, dim3(P.blocks), dim3(64 * P.wpb), lds1, stream, P);
    return hipGetLastError();
}
hipError_t fh_prepass_launch_scan(const PrepassParams &P, hipStream_t stream) {
    hipLaunchKernelGGL(bucket_scan_kernel, dim3((P.nb + 15) / 16), dim3(1024), 0, stream, P);
    return hipGetLastError();
}
hipError_t fh_prepass_launch_factor(const PrepassParams &P, hipStream_t stream) {
    hipLaunchKernelGGL(bucket_factor2_kernel, dim3(P.nb), dim3(1024), 0, stream, P);
    return hipGetLastError();
}

hipError_t fh_vr_gram_launch(const VrGramParams &G, double *stats_sum, double *stats_minmax, hipStream_t stream) {
    hipLaunchKernelGGL(vr_gram_kernel, dim3(G.ntiles * G.split), dim3(64 * G.waves), 0, stream, G);
    hipLaunchKernelGGL(vr_finish_kernel, dim3(G.ntiles), dim3(256), 0, stream, G, stats_sum, stats_minmax);
    return hipGetLastError();
}
