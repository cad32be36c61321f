// K1, the FUSED form of the moments pre-pass (round 6): the visibility table is read ONCE (40 B per row) and the 36 moment sums of
// every J0 bucket are accumulated in LDS -- no sorted table is written or read back.  For tables whose buckets fit the LDS of a
// compute unit (bucket accumulators of 37 doubles: <= ~500 slots); wider tables keep the sorted path of bin_prepass.hip.
//
// Replaces P2 (deproject_scatter_kernel) + P3 (piece_moments_kernel) of bin_prepass.hip, i.e. the same reference code:
// geometry.apply_correction (geometry.py:69-79, 111-131), q = hypot(u', v') (statistical_models.py:166) and the grouping of the
// rows of VisibilityMapping.map_visibilities (statistical_models.py:192-214) by the bucket of their J0 argument
// (hankel.py:187-204 through the Taylor tables of j0_buckets.h).
//
//   fused_layout_kernel    from the bucket totals of P1 (uv_hist_kernel + bucket_scan_kernel, kept between passes over the same
//                          rows): accumulator slots per bucket -- a bucket that many lanes of one instruction hit gets several
//                          copies (lane & mask picks one), because lanes that meet at one LDS address are served one after the
//                          other --, and piece0[b] = b G: every workgroup hands bucket_factor2_kernel one "piece" per bucket
//   fused_moments_kernel   one workgroup of sixteen waves per compute unit; per row 36 ds_add_f64 on its bucket's slot; at the end
//                          the copies of a bucket are added in slot order and written to partial[b][workgroup][36]
//
// What this form gives up: the order in which the waves of a workgroup reach a slot is not fixed, so the sums of a bucket differ
// in their last bits from run to run (the sorted path lands on the same bits in every run).  Opt-in (FRANK_AMD_K1_FUSED=1); what it
// measures is in profiles/r06_binning_fused.txt and DESIGN.md K1.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "deproject.h"
#include "j0_buckets.h"
#include "kernels.h"

namespace {

constexpr int kTerms = FH_K1_TERMS;          // 12
constexpr int kMom = 2 * kTerms - 1;         // moments 0 .. 22 of tau
constexpr int kMomAll = kMom + kTerms + 1;   // + nu_0 .. nu_11 + eta = 36
constexpr int kSlot = kMomAll + 1;           // doubles per accumulator slot (odd: consecutive slots start in different banks)
constexpr int FT = 1024;                     // threads per workgroup
constexpr int kMaxCopies = 16;

// slot_tab[b] = first slot << 8 | (copies - 1); copies a power of two <= 16.  One workgroup, thread b for bucket b.  lanes: expected lanes of one 64-wide instruction that meet in bucket b = 64 totals[b] / rows; a bucket gets the
// power of two >= scale x that, with the largest scale of 4, 2, 1, 1/2, 0 whose slots fit max_slots.
__global__ __launch_bounds__(FT) void fused_layout_kernel(PrepassParams P, int G, int max_slots, int *slot_tab, int *nslots_out) {
    __shared__ long long s_rows[FT / 64];
    __shared__ int s_sum[FT / 64];
    const int nb = P.nb, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    // (thread b holds bucket b: the fused form needs a slot per bucket in LDS, so nb <= max_slots < FT)
    const int tot_b = t < nb ? P.totals[t] : 0;
    long long r = tot_b;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) r += __shfl_down(r, off);
    if (lane == 0) s_rows[wave] = r;
    __syncthreads();
    long long rows_ll = 0;
    for (int w = 0; w < FT / 64; ++w) rows_ll += s_rows[w];
    const double rows = (double)(rows_ll > 0 ? rows_ll : 1);
    const double scales[5] = {4.0, 2.0, 1.0, 0.5, 0.0};
    auto copies_for = [&](double scale) {
        const double want = scale * 64.0 * (double)tot_b / rows;
        int c = 1;
        while (c < kMaxCopies && (double)c < want) c <<= 1;
        return t < nb ? c : 0;
    };
    int c = 0;
    for (int k = 0; k < 5; ++k) {
        c = copies_for(scales[k]);
        int mine = c;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) mine += __shfl_down(mine, off);
        __syncthreads();  // (s_sum of the round before has been read)
        if (lane == 0) s_sum[wave] = mine;
        __syncthreads();
        int tot = 0;
        for (int w = 0; w < FT / 64; ++w) tot += s_sum[w];
        if (tot <= max_slots) break;  // (the same sum in every thread: a uniform exit)
    }
    // exclusive prefix of the copies in bucket order
    int inc = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(inc, off);
        if (lane >= off) inc += o;
    }
    __syncthreads();
    if (lane == 63) s_sum[wave] = inc;
    __syncthreads();
    int before = 0, total = 0;
    for (int w = 0; w < FT / 64; ++w) {
        before += w < wave ? s_sum[w] : 0;
        total += s_sum[w];
    }
    if (t < nb) slot_tab[t] = ((before + inc - c) << 8) | (c - 1);
    if (t == 0) *nslots_out = total;
    for (int b = t; b <= nb; b += FT) P.piece0[b] = b * G;
}

// SAFE: phases beyond 1e5 rad may occur (bin_prepass.hip); U rows per lane and tile
template <bool SAFE, int U>
__global__ __launch_bounds__(FT) void fused_moments_kernel(PrepassParams P, const int *slot_tab, int max_slots, int scalar_blocks) {
    extern __shared__ __attribute__((aligned(16))) double acc[];  // max_slots x kSlot, then nb ints
    __shared__ double red[FT / 64];
    const BinParams &p = P.bin;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nb = P.nb, G = gridDim.x;
    int *tab = reinterpret_cast<int *>(acc + (size_t)max_slots * kSlot);
    for (int e = tid; e < max_slots * kSlot; e += FT) acc[e] = 0.0;
    for (int b = tid; b < nb; b += FT) tab[b] = slot_tab[b];
    const double inv_half = 2.0 * P.inv_delta;
    double pm = 1.0;            // mantissa of the product of the weights, in [0.5, 1) (bin_prepass.hip: one logarithm per lane)
    int pe = 0, pn = 0;
    const int tile = FT * U;
    const int64_t ntiles = (p.count + tile - 1) / tile;
    const int64_t last = p.first + p.count - 1;
    const double *colVim = p.Vim ? p.Vim : p.Vre;
    const bool has_im = p.Vim != nullptr;
    VisRow r[U];
    auto fetch = [&](int64_t tt) {
#pragma unroll
        for (int j = 0; j < U; ++j) {
            int64_t g = p.first + tt * tile + j * FT + tid;
            g = g < last ? g : last;
            if (p.count == 0) g = 0;
            const int64_t gw = p.w_scalar ? 0 : g;
            r[j].u = __builtin_nontemporal_load(&p.u[g]);
            r[j].v = __builtin_nontemporal_load(&p.v[g]);
            r[j].Vre = __builtin_nontemporal_load(&p.Vre[g]);
            r[j].Vim = __builtin_nontemporal_load(&colVim[g]);
            r[j].w = __builtin_nontemporal_load(&p.w[gw]);
        }
    };
    fetch(blockIdx.x);
    __syncthreads();
    for (int64_t t = blockIdx.x; t < ntiles; t += G) {
        VisRow c[U];
#pragma unroll
        for (int j = 0; j < U; ++j) c[j] = r[j];
        fetch(t + G);  // the next tile's rows are in flight during this tile's arithmetic
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const bool active = t * tile + j * FT + tid < p.count;
            const double Vim = has_im ? c[j].Vim : 0.0;
            const double re = SAFE ? fh_phase_centre_re(p, c[j].u, c[j].v, c[j].Vre, Vim)
                                   : fh_phase_centre_re_fast(p, c[j].u, c[j].v, c[j].Vre, Vim);
            const double q = fh_deproject_q_fast(p, c[j].u, c[j].v);
            const double s = p.inv_Qmax * q;
            {   // sum of log(w / 2 pi) (statistical_models.py:218) as the log of a running product, as P2 forms it
                const double w1 = active ? c[j].w : 1.0;
                const unsigned long long bits = __double_as_longlong(w1);
                const int ex = (int)((bits >> 52) & 0x7ff);
                double m1 = __longlong_as_double((bits & 0x800fffffffffffffull) | 0x3fe0000000000000ull);
                int e1 = ex - 1022;
                if (__builtin_expect(__any(ex == 0 || ex == 0x7ff || (long long)bits < 0), 0)) {
                    m1 = frexp(w1, &e1);
                    if (!(w1 > 0.0)) m1 = w1 == 0.0 ? 0.0 : NAN;
                    if (ex == 0x7ff && w1 > 0.0) m1 = INFINITY;
                }
                pm *= m1;
                const unsigned long long pb = __double_as_longlong(pm);
                const int e2 = (int)((pb >> 52) & 0x7ff) - 1022;
                if (pm > 0.0 && pm < INFINITY) {
                    pm = __longlong_as_double((pb & 0x800fffffffffffffull) | 0x3fe0000000000000ull);
                    pe += e1 + e2;
                }
                pn += active ? 1 : 0;
            }
            const int bk = fh_bucket_of(s, P.inv_delta, nb);
            const double tau = fh_bucket_tau(s, bk, P.delta, inv_half);
            if (active) {
                const int e = tab[bk];
                double *a = acc + (size_t)((e >> 8) + (lane & (e & 255))) * kSlot;
                const double w = c[j].w, wv = w * re;
                double pw = w, pv = wv;
#pragma unroll
                for (int m = 0; m < kMom; ++m) {
                    unsafeAtomicAdd(a + m, pw);
                    if (m < kTerms) {
                        unsafeAtomicAdd(a + kMom + m, pv);
                        pv *= tau;
                    }
                    pw *= tau;
                }
                unsafeAtomicAdd(a + kMom + kTerms, wv * re);
            }
        }
    }
    __syncthreads();
    // the copies of a bucket in slot order -> this workgroup's "piece" of the bucket
    for (int e = tid; e < nb * kMomAll; e += FT) {
        const int b = e / kMomAll, m = e - b * kMomAll;
        const int te = tab[b], s0 = te >> 8, nc = (te & 255) + 1;
        double sum = 0.0;
        for (int k = 0; k < nc; ++k) sum += acc[(size_t)(s0 + k) * kSlot + m];
        P.partial[((size_t)b * G + blockIdx.x) * kMomAll + m] = sum;
    }
    double sum_logw = (log(pm) + (double)pe * M_LN2) - (double)pn * log(2 * M_PI);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sum_logw += __shfl_down(sum_logw, off);
    if (lane == 0) red[wave] = sum_logw;
    __syncthreads();
    if (tid == 0) {
        double sacc = 0.0;
        for (int w = 0; w < FT / 64; ++w) sacc += red[w];
        P.partial_scalars[(size_t)blockIdx.x * 4] = sacc;
        for (int k = blockIdx.x + G; k < scalar_blocks; k += G) P.partial_scalars[(size_t)k * 4] = 0.0;  // (vr_finish adds them all)
    }
}

}  // namespace

int fh_fused_slot_doubles() { return kSlot; }

// slots that fit the LDS of one workgroup beside the table of nb ints (160 KB per compute unit, one workgroup resident)
int fh_fused_max_slots(int nb) {
    const long bytes = 160 * 1024 - 1024 - (long)sizeof(int) * nb;
    const long s = bytes / (long)(sizeof(double) * kSlot);
    return s > 0 ? (int)s : 0;
}

hipError_t fh_fused_launch_layout(const PrepassParams &P, int G, int max_slots, int *slot_tab, int *nslots_out, hipStream_t stream) {
    hipLaunchKernelGGL(fused_layout_kernel, dim3(1), dim3(FT), 0, stream, P, G, max_slots, slot_tab, nslots_out);
    return hipGetLastError();
}

hipError_t fh_fused_launch(const PrepassParams &P, const int *slot_tab, int max_slots, int G, int scalar_blocks, hipStream_t stream) {
    const size_t lds = sizeof(double) * (size_t)max_slots * kSlot + sizeof(int) * (size_t)P.nb;
    auto go = [&](auto kern) -> hipError_t {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(G), dim3(FT), lds, stream, P, slot_tab, max_slots, scalar_blocks);
        return hipGetLastError();
    };
    if (P.unroll == 2) return P.safe_trig ? go(fused_moments_kernel<true, 2>) : go(fused_moments_kernel<false, 2>);
    return P.safe_trig ? go(fused_moments_kernel<true, 1>) : go(fused_moments_kernel<false, 1>);
}
