// K1 `bin_gram`: Bessel design block + weighted Gram on gfx950, after a deprojection pre-pass.
//
// Replaces the chunk loop of VisibilityMapping.map_visibilities (statistical_models.py:165-218),
// with geometry.apply_correction (geometry.py:69-79, 111-131) and DHT.coefficients
// (hankel.py:201-202) fused in.
//
//   K1a deproject_kernel   one thread per visibility (HBM-bound streaming pass, 40 B in / 24 B out):
//                          phase-centre, deproject, q = hypot, and emit  s = q/Qmax, sqrt(w), sqrt(w) Re V';
//                          per-block partial sums of log(w/2pi) and min/max q.
//   K1b bin_gram_kernel    per visibility i form the row
//                              Xt[i,k] = sqrt(w_i) J0(s_i j_k) (k < N),  Xt[i,N] = sqrt(w_i) Re V'_i,  0 beyond,
//                          and accumulate the symmetric Gram G = Xt^T Xt with v_mfma_f64_16x16x4_f64, so that
//                              M[k,l] = a_k a_l G[k,l],  j[k] = a_k G[k,N],  sum w V'^2 = G[N,N],  a_k = norm sf_k scale.
// The DHT scaling a_k (~1e-14) is applied once afterwards in fp64 (finalize kernel): the Gram has O(1) entries.
//
// Work decomposition of K1b (DESIGN.md "K1"):
//   * one 768-thread workgroup per CU = 12 waves, three per SIMD (K1_WAVES; the first design had 8): while a wave
//     of a SIMD waits on the matrix pipe or on LDS the others issue J0 polynomial work (VALU) or their own MFMAs;
//     the waves of a SIMD produce their J0 row at different k-steps of the chunk;
//   * the upper triangle of the NBT x NBT grid of 16x16 output tiles (190 tiles at N = 300) stays in
//     accumulator registers for the whole visibility stream.  190 tiles x 8 registers do not fit one CU
//     beside the J0 temporaries, so for NBT = 19 the triangle is cut row-aligned into two PARTS (tile rows
//     0-6: 112 tiles, rows 7-18: 78 tiles); the grid is split between the parts in proportion to their tile
//     counts and every part streams ALL visibilities, evaluating only the J0 columns it needs (1.68x J0);
//   * visibilities are streamed in super-chunks of 768 and chunks of 12 rows; a chunk's rows are written to
//     LDS (double-buffered), one row per wave, and read back as MFMA fragments (the A-fragment of block I is
//     the B-fragment of block I).  The loop is specialised per wave (static accumulator registers), with the J0
//     evaluation rolled over the column groups so that a part's eight specialisations fit the I-cache;
//   * super-chunks are handed out either statically (block b takes b, b+G, ...: bitwise reproducible, the
//     default) or from an atomic counter (throughput mode, used while fit_loop kernels of earlier fits
//     occupy CUs: a workgroup that starts late simply takes fewer super-chunks);
//   * each workgroup writes its partial tiles once; a second kernel reduces the slabs in block order and a
//     third applies the DHT scaling and mirrors the triangle.
#include <hip/hip_runtime.h>

#include <utility>

// Horner form of the J0 polynomials: with three waves per SIMD hiding the chain latency, the 9 DP operations fewer per
// large-argument evaluation are worth more than Estrin's shorter dependency depth (28.5 -> 28.0 ms); max |error| vs
// 40-digit mpmath on [0, 1000]: 1.1e-16 (Estrin form: 2.2e-16).
#define FH_J0_HORNER 1
#include "bessel.h"
#include "kernels.h"
#include "deproject.h"

typedef double v4f64 __attribute__((ext_vector_type(4)));

namespace {

// K1_WAVES = 12 (default): three waves per SIMD (166 VGPRs), 24-row chunks, two J0 rows per wave and chunk: 27.3 ms.
// K1_WAVES = 8: two waves per SIMD, 16-row chunks, two J0 rows per wave and chunk: 30.2 ms (the first design).
// K1_WAVES = 16: four per SIMD would need <= 128 VGPRs: 167 spilled registers, not viable.
#ifndef K1_WAVES
#define K1_WAVES 12
#endif
constexpr int kWaves = K1_WAVES;
constexpr int kThreads = 64 * kWaves;
constexpr int kSuper = kThreads;          // visibilities per super-chunk (one LDS row of scalars per thread)
#ifndef K1_CHUNK
#define K1_CHUNK (K1_WAVES == 8 ? 16 : 2 * K1_WAVES)  // 24 rows: half the barriers of 12 (28.1 -> 27.3 ms), 158 KB of LDS
#endif
constexpr int kChunk = K1_CHUNK;  // rows per LDS buffer = kChunk / 4 MFMA k-steps
constexpr int kRowsPerWave = kChunk / kWaves;  // J0 rows a wave produces per chunk
constexpr int kChunksPerSuper = kSuper / kChunk;
static_assert(kChunk % 4 == 0 && kChunk % kWaves == 0 && kSuper % kChunk == 0, "chunk geometry");

constexpr int xstride(int NBT) {  // LDS row stride in doubles, == 16 (mod 32): conflict-free fragment reads
    return (NBT * 16) % 32 == 16 ? NBT * 16 : NBT * 16 + 16;
}
constexpr int ntiles(int NBT) { return NBT * (NBT + 1) / 2; }
// t-th tile of the upper triangle in row-major order -> (I, J)
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
constexpr int row_first_tile(int NBT, int I) { return I * NBT - I * (I - 1) / 2; }
// parts: NBT = 19 -> rows [0,7) and [7,19); otherwise one part
constexpr int nparts(int NBT) { return NBT > 13 ? 2 : 1; }
constexpr int part_row0(int NBT, int P) { return (NBT > 13 && P == 1) ? 7 : 0; }
constexpr int part_row1(int NBT, int P) { return (NBT > 13 && P == 0) ? 7 : NBT; }
constexpr int part_tile0(int NBT, int P) { return row_first_tile(NBT, part_row0(NBT, P)); }
constexpr int part_tile1(int NBT, int P) { return row_first_tile(NBT, part_row1(NBT, P)); }


// Which tiles a wave owns.  The accumulation of a tile does not depend on its owner, so any assignment gives the same
// bits; what it changes is how many distinct 16-column fragments a wave reads from LDS per k-step (the union of the row
// and column blocks of its tiles).  Round-robin over the row-major triangle: 13 fragments per wave in part 0, 9-10 in
// part 1.  For NBT = 19 with 12 waves the tiles are dealt as compact rectangles (two block rows x <= 5 block columns):
// 5-8 fragments in part 0, 3-6 in part 1 -- half the fragment reads -- with the same <= 10 tiles per wave and 28 (19-20)
// tiles per SIMD (waves W, W+4, W+8 share SIMD W % 4).  Tables: tools/gen_k1_tile_tables.py.
constexpr short kTiles19P0[12][10] = {
    {5, 6, 7, 8, 9, 23, 24, 25, 26, 27},        {42, 43, 44, 45, 46, 58, 59, 60, 61, 62},
    {75, 76, 77, 78, 79, 89, 90, 91, 92, 93},   {80, 81, 82, 83, 84, 94, 95, 96, 97, 98},
    {10, 11, 12, 13, 14, 28, 29, 30, 31, 32},   {47, 48, 49, 50, 51, 63, 64, 65, 66, 67},
    {0, 1, 2, 3, 4, 19, 20, 21, 22, -1},        {70, 71, 72, 73, 74, 85, 86, 87, 88, -1},
    {15, 16, 17, 18, 33, 34, 35, 36, -1, -1},   {99, 100, 101, 102, 103, 104, 105, 106, -1, -1},
    {37, 38, 39, 40, 41, 54, 55, 56, 57, -1},   {52, 53, 68, 69, 107, 108, 109, 110, 111, -1},
};
constexpr short kTiles19P1[12][10] = {
    {180, 181, 182, 183, 184, 185, 186, 187, 188, 189}, {158, 159, 160, 161, 165, 166, 167, 168, -1, -1},
    {112, 113, 114, 115, 124, 125, 126, -1, -1, -1},    {154, 155, 156, 157, 162, 163, 164, -1, -1, -1},
    {169, 170, 171, 175, 176, -1, -1, -1, -1, -1},      {116, 117, 118, 127, 128, 129, -1, -1, -1, -1},
    {135, 136, 137, 138, 145, 146, 147, -1, -1, -1},    {142, 143, 144, 151, 152, 153, -1, -1, -1, -1},
    {122, 123, 133, 134, -1, -1, -1, -1, -1, -1},       {119, 120, 121, 130, 131, 132, -1, -1, -1, -1},
    {139, 140, 141, 148, 149, 150, -1, -1, -1, -1},     {172, 173, 174, 177, 178, 179, -1, -1, -1, -1},
};
#ifdef K1_ROUND_ROBIN_TILES
constexpr bool kCompactTiles = false;
#else
constexpr bool kCompactTiles = (kWaves == 12);
#endif
// global index of the T-th tile of wave W in part P (-1: none)
constexpr int wave_tile(int NBT, int P, int W, int T) {
    if (kCompactTiles && NBT == 19 && nparts(NBT) == 2) return T < 10 ? (P == 0 ? kTiles19P0[W][T] : kTiles19P1[W][T]) : -1;
    const int tt = part_tile0(NBT, P) + W + T * kWaves;
    return tt < part_tile1(NBT, P) ? tt : -1;
}
constexpr int wave_ntiles(int NBT, int P, int W) {
    int n = 0;
    while (n < 64 && wave_tile(NBT, P, W, n) >= 0) ++n;
    return n;
}

// ---- K1a ----------------------------------------------------------------------------------------------------
// geometry.py:69-79 (inverse phase shift, NumPy's Smith complex division), :111-131 (deproject),
// statistical_models.py:166 (hypot).  Every product/sum rounds separately, as the NumPy expressions do.
__global__ __launch_bounds__(256) void deproject_kernel(BinParams p) {
    __shared__ double red[3 * 4], red_all[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double sum_logw = 0.0, qmin = INFINITY, qmax = -INFINITY, qmax_all = -INFINITY;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + tid; i < p.count; i += (int64_t)gridDim.x * blockDim.x) {
#pragma clang fp contract(off)
        const int64_t g = p.first + i;
        const VisRow r = fh_load_row(p, g);
        const double u = r.u, v = r.v, w = r.w;
        // multiplicity of the row in a bootstrap resample (utilities.py:632-666): c copies of a row contribute
        // c w h h^T, c w V h, c (log(w/2pi) - w V^2); rows drawn zero times drop out of min/max q as well
        const double mult = p.mult ? (double)p.mult[g] : 1.0;
        const double re = fh_phase_centre_re(p, u, v, r.Vre, r.Vim);
        const double q = fh_deproject_q(p, u, v);
        const double sw = p.mult ? sqrt(mult * w) : sqrt(w);
        if (p.prep_k2) p.prep_k2[i] = fh_deproject_kz2(p, u, v);  // debris model
        p.prep_s[i] = p.inv_Qmax * q;  // k * q, hankel.py:189,202
        p.prep_sw[i] = sw;
        p.prep_swV[i] = sw * re;
        qmax_all = fmax(qmax_all, q);
        if (mult > 0.0) {
            const double lw = log(w / (2 * M_PI));  // statistical_models.py:218
            sum_logw += p.mult ? mult * lw : lw;
            qmin = fmin(qmin, q);
            qmax = fmax(qmax, q);
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        sum_logw += __shfl_down(sum_logw, off);
        qmin = fmin(qmin, __shfl_down(qmin, off));
        qmax = fmax(qmax, __shfl_down(qmax, off));
        qmax_all = fmax(qmax_all, __shfl_down(qmax_all, off));
    }
    if (lane == 0) {
        red[wave * 3 + 0] = sum_logw;
        red[wave * 3 + 1] = qmin;
        red[wave * 3 + 2] = qmax;
        red_all[wave] = qmax_all;
    }
    __syncthreads();
    if (tid == 0) {
        double s = 0, mn = INFINITY, mx = -INFINITY, ma = -INFINITY;
        for (int w = 0; w < 4; ++w) {
            s += red[w * 3 + 0];
            mn = fmin(mn, red[w * 3 + 1]);
            mx = fmax(mx, red[w * 3 + 2]);
            ma = fmax(ma, red_all[w]);
        }
        double *ps = p.partial_scalars + (size_t)blockIdx.x * 4;
        ps[0] = s;
        ps[1] = mn;
        ps[2] = mx;
        ps[3] = ma;  // every row whatever its multiplicity: sizes the bucket sort (capi_map.hip)
    }
}

// ---- K1b ----------------------------------------------------------------------------------------------------
template <int NBT, int P, int W, int T>
__device__ __forceinline__ void mfma_one(v4f64 &acc, const double (&f)[NBT]) {
    constexpr int tt = wave_tile(NBT, P, W, T);
    constexpr int I = tile_I(NBT, tt), J = tile_J(NBT, tt);
    // A[i][k] = Xt[k][16I+i] and B[k][j] = Xt[k][16J+j] share one fragment layout: lane -> (k = lane>>4, i|j = lane&15)
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(f[I], f[J], acc, 0, 0, 0);
}
template <int NBT, int P, int W, int TPW, int... Ts>
__device__ __forceinline__ void mfma_all(v4f64 (&acc)[TPW > 0 ? TPW : 1], const double (&f)[NBT],
                                         std::integer_sequence<int, Ts...>) {
    (mfma_one<NBT, P, W, Ts>(acc[Ts], f), ...);
}

template <int NBT, int P, int W>
__device__ __forceinline__ void wave_main(const BinParams &p, double *smem, int part_block, int part_nblocks) {
    constexpr int NC = NBT * 16;
    constexpr int XS = xstride(NBT);
    constexpr int T0 = part_tile0(NBT, P), T1 = part_tile1(NBT, P);
    constexpr int NTP = T1 - T0;                          // tiles of this part
    constexpr int TPW = wave_ntiles(NBT, P, W);           // tiles of this wave (wave_tile)
    constexpr int B0 = part_row0(NBT, P);                 // first column block this part needs
    constexpr int C0 = B0 * 16;                           // first column
    constexpr int NCG = (NC - C0 + 63) / 64;              // column groups of 64 lanes
    constexpr bool kSkew = W >= 4;                        // second wave of each SIMD: half a chunk out of phase
    constexpr int kPhase = W / 4;                         // position among the waves of this SIMD (12-wave layout)
    constexpr int TPWA = TPW > 0 ? TPW : 1;               // (with 12 waves the small bases leave some waves without tiles)

    double *tab = smem;                                                      // FH_J0_TABLE_DOUBLES
    double *vs_s = tab + FH_J0_TABLE_DOUBLES;                                // [2][kSuper] each
    double *vs_sw = vs_s + 2 * kSuper;
    double *vs_swV = vs_sw + 2 * kSuper;
    double *X = vs_swV + 2 * kSuper;                                         // [2][kChunk][XS]
    double *jkl = X + 2 * kChunk * XS;                                       // zeros j_k of columns C0.. (NCG*64)
    int *scq = reinterpret_cast<int *>(jkl + NCG * 64);                      // [2] super-chunk queue

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int N = p.N;
    const int kk = lane >> 4, ii = lane & 15;
    const int nsup = (int)((p.count + kSuper - 1) / kSuper);

    for (int c = tid; c < NCG * 64; c += kThreads) jkl[c] = (C0 + c) < N ? p.zeros[C0 + c] : 0.0;  // J0(0) = 1 beyond N

    v4f64 acc[TPWA];
#pragma unroll
    for (int t = 0; t < TPW; ++t) acc[t] = v4f64{0.0, 0.0, 0.0, 0.0};

    // ---- super-chunk hand-out: static stride (reproducible) or atomic counter (throughput mode) -------------
    int *counter = p.work_counter ? p.work_counter + P : nullptr;
    auto fetch = [&](int prev) -> int {  // thread 0 only
        if (counter) return atomicAdd(counter, 1);
        return prev < 0 ? part_block : prev + part_nblocks;
    };
    if (tid == 0) {
        const int s0 = fetch(-1);
        scq[0] = s0;
        scq[1] = fetch(s0);
    }
    __syncthreads();
    int cur = scq[0], nxt = scq[1];

    // ---- scalars of one super-chunk into vs[buf] (coalesced copy of the K1a output) --------------------------
    auto load_scalars = [&](int sc, int buf) {
        const int64_t i = (int64_t)sc * kSuper + tid;
        const bool ok = sc < nsup && i < p.count;
        vs_s[buf * kSuper + tid] = ok ? p.prep_s[i] : 0.0;
        vs_sw[buf * kSuper + tid] = ok ? p.prep_sw[i] : 0.0;   // padded rows contribute exactly zero
        vs_swV[buf * kSuper + tid] = ok ? p.prep_swV[i] : 0.0;
    };
    // ---- one J0 row: this wave produces rows 2W, 2W+1 of every chunk, one 64-column group at a time -----------
    auto produce_row = [&](int sbuf, int ch, int xbuf, int rr) {
        const int row = W * kRowsPerWave + rr;
        const int vi = sbuf * kSuper + ch * kChunk + row;
        const double s = vs_s[vi], sw = vs_sw[vi], swV = vs_swV[vi];
        double *xr = X + (xbuf * kChunk + row) * XS + C0;
#pragma unroll 1
        for (int cg = 0; cg < NCG; ++cg) {
            const int lc = cg * 64 + lane;  // column - C0
            if ((NC - C0) % 64 == 0 || lc < NC - C0) {
                double x;
                {
#pragma clang fp contract(off)
                    x = s * jkl[lc];  // fl(fl(k*q) * j_k), hankel.py:202
                }
                const double val = fh_j0(x, tab);
                const int col = C0 + lc;
                // columns < N: sqrt(w) J0;  column N: sqrt(w) Re V';  beyond: 0
                xr[lc] = col < N ? val * sw : (col == N ? swV : 0.0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // ---- MFMAs of `NKS` k-steps (4 rows each) of X[xbuf] ------------------------------------------------------
    auto consume = [&](int xbuf, int ks0, auto nks_tag) {
        constexpr int NKS = decltype(nks_tag)::value;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const double *xb = X + (xbuf * kChunk + (ks0 + ks) * 4 + kk) * XS + ii;
            double f[NBT];
#pragma unroll
            for (int b = 0; b < NBT; ++b) f[b] = b >= B0 ? xb[b * 16] : 0.0;
            mfma_all<NBT, P, W, TPW>(acc, f, std::make_integer_sequence<int, TPW>{});
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto consume2 = [&](int xbuf, int ks0) { consume(xbuf, ks0, std::integral_constant<int, 2>{}); };
    auto consume1 = [&](int xbuf, int ks0) { consume(xbuf, ks0, std::integral_constant<int, 1>{}); };

    // ---- main loop --------------------------------------------------------------------------------------------
    int sbuf = 0, xbuf = 0;
    if (cur < nsup) {
        load_scalars(cur, 0);
        __syncthreads();
#pragma unroll
        for (int rr = 0; rr < kRowsPerWave; ++rr) produce_row(0, 0, 0, rr);
        __syncthreads();
    }
    int qslot = 0;
    while (cur < nsup) {
        if (tid == 0) scq[qslot] = fetch(nxt);  // the super-chunk after next; read after this iteration's barriers
        load_scalars(nxt, sbuf ^ 1);             // visible after the first barrier below
#pragma unroll 1
        for (int ch = 0; ch < kChunksPerSuper; ++ch) {
            const bool last = (ch == kChunksPerSuper - 1);
            const bool more = !last || nxt < nsup;
            const int nsb = last ? (sbuf ^ 1) : sbuf;
            const int nch = last ? 0 : ch + 1;
            // J0 rows of chunk c+1 (VALU) against the MFMAs of chunk c (matrix pipe)
            if constexpr (kWaves == 8) {
                if (!kSkew) {
                    if (more) produce_row(nsb, nch, xbuf ^ 1, 0);
                    consume2(xbuf, 0);
                    if (more) produce_row(nsb, nch, xbuf ^ 1, 1);
                    consume2(xbuf, 2);
                } else {
                    consume2(xbuf, 0);
                    if (more) produce_row(nsb, nch, xbuf ^ 1, 0);
                    consume2(xbuf, 2);
                    if (more) produce_row(nsb, nch, xbuf ^ 1, 1);
                }
            } else {  // three waves per SIMD: a wave's J0 rows slide through the k-steps according to its position
#pragma unroll
                for (int ks = 0; ks < kChunk / 4; ++ks) {
#pragma unroll
                    for (int rr = 0; rr < kRowsPerWave; ++rr)  // (rows half a chunk apart instead: no difference)
                        if (ks == kPhase * kRowsPerWave + rr && more) produce_row(nsb, nch, xbuf ^ 1, rr);
                    consume1(xbuf, ks);
                }
            }
            __syncthreads();
            xbuf ^= 1;
        }
        sbuf ^= 1;
        cur = nxt;
        nxt = scq[qslot];
        qslot ^= 1;
    }

    // ---- write this workgroup's partial tiles: slab[part_block][tile - T0][reg][lane] --------------------------
    double *slab = p.partials[P] + (size_t)part_block * NTP * 256;
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int tl = wave_tile(NBT, P, W, t) - T0;
#pragma unroll
        for (int r = 0; r < 4; ++r) slab[(size_t)tl * 256 + r * 64 + lane] = acc[t][r];
    }
}

template <int NBT, int P>
__device__ __forceinline__ void part_main(const BinParams &p, double *smem, int part_block, int part_nblocks) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#ifdef K1_ICACHE_TEST  // experiment only (wrong results): every wave runs ONE specialisation -> 1/8 of the hot code
    if (wave >= 0) {
        if (wave & 4) wave_main<NBT, P, 4>(p, smem, part_block, part_nblocks);
        else wave_main<NBT, P, 0>(p, smem, part_block, part_nblocks);
        return;
    }
#endif
    switch (wave) {
        case 0: wave_main<NBT, P, 0>(p, smem, part_block, part_nblocks); break;
        case 1: wave_main<NBT, P, 1>(p, smem, part_block, part_nblocks); break;
        case 2: wave_main<NBT, P, 2>(p, smem, part_block, part_nblocks); break;
        case 3: wave_main<NBT, P, 3>(p, smem, part_block, part_nblocks); break;
        case 4: wave_main<NBT, P, 4>(p, smem, part_block, part_nblocks); break;
        case 5: wave_main<NBT, P, 5>(p, smem, part_block, part_nblocks); break;
        case 6: wave_main<NBT, P, 6>(p, smem, part_block, part_nblocks); break;
#if K1_WAVES == 8
        default: wave_main<NBT, P, 7>(p, smem, part_block, part_nblocks); break;
#else
        case 7: wave_main<NBT, P, 7>(p, smem, part_block, part_nblocks); break;
        case 8: wave_main<NBT, P, 8>(p, smem, part_block, part_nblocks); break;
        case 9: wave_main<NBT, P, 9>(p, smem, part_block, part_nblocks); break;
        case 10: wave_main<NBT, P, 10>(p, smem, part_block, part_nblocks); break;
#if K1_WAVES == 12
        default: wave_main<NBT, P, 11>(p, smem, part_block, part_nblocks); break;
#else
        case 11: wave_main<NBT, P, 11>(p, smem, part_block, part_nblocks); break;
        case 12: wave_main<NBT, P, 12>(p, smem, part_block, part_nblocks); break;
        case 13: wave_main<NBT, P, 13>(p, smem, part_block, part_nblocks); break;
        case 14: wave_main<NBT, P, 14>(p, smem, part_block, part_nblocks); break;
        default: wave_main<NBT, P, 15>(p, smem, part_block, part_nblocks); break;
#endif
#endif
    }
}

template <int NBT>
__global__ __launch_bounds__(kThreads, kWaves / 4) void bin_gram_kernel(BinParams p) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    for (int i = threadIdx.x; i < FH_J0_TABLE_DOUBLES; i += kThreads) smem[i] = p.j0_table[i];
    __syncthreads();
    const int b = blockIdx.x;
    if (nparts(NBT) == 1 || b < p.part_blocks[0]) {
        part_main<NBT, 0>(p, smem, b, p.part_blocks[0]);
    } else {
        part_main<NBT, nparts(NBT) - 1>(p, smem, b - p.part_blocks[0], p.part_blocks[1]);
    }
}

template <int NBT>
constexpr size_t bin_smem_bytes() {
    return sizeof(double) * (FH_J0_TABLE_DOUBLES + 3 * 2 * kSuper + 2 * kChunk * xstride(NBT) + ((NBT * 16 + 63) / 64) * 64 + 2);
}

// Sum the per-workgroup slabs of every part and add into the running statistics -- in a FIXED order (bitwise
// reproducible), in two levels so that the ~50 MB of slabs are read by thousands of threads instead of 190 serial
// chains: level 1, group g of kReduceGroups sums its contiguous range of slabs in block order into scratch[g][e];
// level 2 adds the groups in order.
constexpr int kReduceGroups = 8;
__global__ void reduce_partials_level1(ReduceParams rp, double *scratch) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t ne = (int64_t)rp.ntiles * 256;
    if (e >= ne) return;
    const int g = blockIdx.y;
    int part = 0;
    for (int q = 1; q < rp.nparts; ++q)
        if (e >= (int64_t)rp.part_tile0[q] * 256) part = q;
    const int64_t pe = e - (int64_t)rp.part_tile0[part] * 256;
    const int64_t stride = (int64_t)rp.part_ntiles[part] * 256;
    const double *src = rp.partials[part];
    const int nb = rp.part_blocks[part], per = (nb + kReduceGroups - 1) / kReduceGroups;
    const int b0 = g * per, b1 = min(nb, b0 + per);
    double s = 0.0;
#pragma unroll 8
    for (int b = b0; b < b1; ++b) s += src[(size_t)b * stride + pe];
    scratch[(size_t)g * ne + e] = s;
}
__global__ void reduce_partials_kernel(ReduceParams rp, const double *scratch, double *stats_sum, double *stats_minmax) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t ne = (int64_t)rp.ntiles * 256;
    if (e < ne) {
        double s = 0.0;
#pragma unroll
        for (int g = 0; g < kReduceGroups; ++g) s += scratch[(size_t)g * ne + e];
        stats_sum[e] += s;
    }
    if (blockIdx.x == 0) {
        // the per-block scalars of deproject_kernel (up to 2048 of them): strided partial sums, then a fixed tree
        __shared__ double rs[256], rmn[256], rmx[256];
        const int t = threadIdx.x;
        double s = 0.0, mn = INFINITY, mx = -INFINITY;
        for (int b = t; b < rp.scalar_blocks; b += 256) {
            s += rp.partial_scalars[b * 4 + 0];
            mn = fmin(mn, rp.partial_scalars[b * 4 + 1]);
            mx = fmax(mx, rp.partial_scalars[b * 4 + 2]);
        }
        rs[t] = s;
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
            stats_sum[ne + 0] += rs[0];
            // min/max are kept as (-qmin, qmax) so that one max-all-reduce serves both
            stats_minmax[0] = fmax(stats_minmax[0], -rmn[0]);
            stats_minmax[1] = fmax(stats_minmax[1], rmx[0]);
        }
    }
}

// Apply a_k a_l, unpack the upper-triangle tiles to the dense symmetric M (N*N), j (N) and sum w V^2.
__global__ void finalize_stats_kernel(const double *stats_sum, int NBT, int N, const double *a, double *M, double *j,
                                      double *sumwV2) {
    const int t = blockIdx.x;  // tile
    int I = 0, tt = t;
    while (tt >= NBT - I) {
        tt -= NBT - I;
        ++I;
    }
    const int J = I + tt;
    const int lane = threadIdx.x & 63, r = threadIdx.x >> 6;
    // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
    const int row = 16 * I + (lane >> 4) + 4 * r;
    const int col = 16 * J + (lane & 15);
    const double g = stats_sum[(size_t)t * 256 + r * 64 + lane];
    if (row < N && col < N) {
        const double m = (a[row] * a[col]) * g;
        if (I != J || row <= col) {
            M[(size_t)row * N + col] = m;
            M[(size_t)col * N + row] = m;
        }
    } else if (row < N && col == N) {
        j[row] = a[row] * g;
    } else if (row == N && col == N) {
        *sumwV2 = g;
    }
}

// ---- N > 303: rows to memory + rocBLAS dsyrk (the register-resident kernel cannot hold more than 190 tiles) --------
// Xt[i, k] = sqrt(w_i) J0(s_i j_k) (k < N), Xt[i, N] = sqrt(w_i) Re V'_i, row-major with leading dimension N + 1: read
// as a column-major (N+1) x rows matrix it is the operand of G += Xc Xc^T.
// debris model (statistical_models.py:494-496): the row is further scaled by exp(-kz_i^2 H2[k]); k2 / H2 NULL otherwise.
__global__ void wide_rows_kernel(const double *prep_s, const double *prep_sw, const double *prep_swV, const double *k2,
                                 const double *H2, int64_t first, int64_t rows, int N, const double *zeros,
                                 const double *j0_table, double *X) {
    __shared__ double tab[FH_J0_TABLE_DOUBLES];
    for (int i = threadIdx.x; i < FH_J0_TABLE_DOUBLES; i += blockDim.x) tab[i] = j0_table[i];
    __syncthreads();
    const int N1 = N + 1;
    const int64_t total = rows * (int64_t)N1;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = e / N1;
        const int k = (int)(e - i * N1);
        const int64_t g = first + i;
        double val;
        if (k < N) {
            double x;
            {
#pragma clang fp contract(off)
                x = prep_s[g] * zeros[k];
            }
            val = prep_sw[g] * fh_j0(x, tab);
            if (H2) {
#pragma clang fp contract(off)
                val = val * exp(-(k2[g] * H2[k]));
            }
        } else {
            val = prep_swV[g];
        }
        X[e] = val;
    }
}

// fold the per-block scalars of deproject_kernel into the running totals (the tile reducer does this for N <= 303)
__global__ void wide_scalars_kernel(const double *partial_scalars, int blocks, double *tail, double *stats_minmax) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        double s = 0.0, mn = INFINITY, mx = -INFINITY;
        for (int b = 0; b < blocks; ++b) {
            s += partial_scalars[b * 4 + 0];
            mn = fmin(mn, partial_scalars[b * 4 + 1]);
            mx = fmax(mx, partial_scalars[b * 4 + 2]);
        }
        tail[0] += s;
        stats_minmax[0] = fmax(stats_minmax[0], -mn);
        stats_minmax[1] = fmax(stats_minmax[1], mx);
    }
}

// G is the (N+1) x (N+1) column-major Gram with the UPPER triangle valid: entry (r, c), r <= c, at G[c * (N+1) + r]
__global__ void wide_finalize_kernel(const double *G, int N, const double *a, double *M, double *j, double *sumwV2) {
    const int N1 = N + 1;
    const int64_t total = (int64_t)N1 * N1;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(e / N1), r = (int)(e - (int64_t)c * N1);
        if (r > c) continue;
        const double g = G[e];
        if (c < N) {
            const double m = (a[r] * a[c]) * g;
            M[(size_t)r * N + c] = m;
            M[(size_t)c * N + r] = m;
        } else if (r < N) {
            j[r] = a[r] * g;
        } else {
            *sumwV2 = g;
        }
    }
}

// a3/a7: H[i,k] = (norm*sf_k) * J0((kq*q_i) * j_k) * scale   (hankel.py:201-202, statistical_models.py:507)
__global__ void coefficients_kernel(const double *q, int64_t n, int N, const double *zeros, const double *pref,
                                    double inv_Q, double scale, const double *j0_table, double *H) {
    __shared__ double tab[FH_J0_TABLE_DOUBLES];
    for (int i = threadIdx.x; i < FH_J0_TABLE_DOUBLES; i += blockDim.x) tab[i] = j0_table[i];
    __syncthreads();
    const int64_t total = n * (int64_t)N;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = e / N;
        const int k = (int)(e - i * N);
        double x;
        {
#pragma clang fp contract(off)
            x = (inv_Q * q[i]) * zeros[k];
        }
        H[e] = (pref[k] * fh_j0(x, tab)) * scale;
    }
}

// predict_visibilities: V_i = sum_k H[i,k] I_k, one wave per visibility row (statistical_models.py:326-328)
__global__ void predict_kernel(const double *q, int64_t n, int N, const double *zeros, const double *pref,
                               double inv_Q, double scale, const double *I, const double *j0_table, double *V) {
    __shared__ double tab[FH_J0_TABLE_DOUBLES];
    for (int i = threadIdx.x; i < FH_J0_TABLE_DOUBLES; i += blockDim.x) tab[i] = j0_table[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t i = wave; i < n; i += nwaves) {
        const double s = inv_Q * q[i];
        double a = 0.0;
        for (int k = lane; k < N; k += 64) {
            const double h = (pref[k] * fh_j0(s * zeros[k], tab)) * scale;
            a = fma(h, I[k], a);
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) a += __shfl_down(a, off);
        if (lane == 0) V[i] = a;
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------
int fh_k1_nbt_for(int N) {
    const int nb = (N + 1 + 15) / 16;
    if (nb <= 4) return 4;
    if (nb <= 8) return 8;
    if (nb <= 13) return 13;
    if (nb <= 19) return 19;
    return 0;  // N > 303: not covered by the register-resident kernel
}
int fh_k1_ntiles(int NBT) { return ntiles(NBT); }
int fh_k1_nparts(int NBT) { return nparts(NBT); }
int fh_k1_part_tile0(int NBT, int P) { return part_tile0(NBT, P); }
int fh_k1_part_ntiles(int NBT, int P) { return part_tile1(NBT, P) - part_tile0(NBT, P); }
int fh_k1_super() { return kSuper; }

template <int NBT>
static hipError_t launch_bin(const BinParams &p, hipStream_t stream) {
    constexpr size_t smem = bin_smem_bytes<NBT>();
    {  // per launch (cheap): the attribute is per device, and contexts on several devices share this code
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&bin_gram_kernel<NBT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
    }
    const int grid = p.part_blocks[0] + (nparts(NBT) > 1 ? p.part_blocks[1] : 0);
    hipLaunchKernelGGL(bin_gram_kernel<NBT>, dim3(grid), dim3(kThreads), smem, stream, p);
    return hipGetLastError();
}

hipError_t fh_k1_launch_deproject(const BinParams &p, int blocks, hipStream_t stream) {
    hipLaunchKernelGGL(deproject_kernel, dim3(blocks), dim3(256), 0, stream, p);
    return hipGetLastError();
}

hipError_t fh_k1_launch_bin(int NBT, const BinParams &p, hipStream_t stream) {
    switch (NBT) {
        case 4: return launch_bin<4>(p, stream);
        case 8: return launch_bin<8>(p, stream);
        case 13: return launch_bin<13>(p, stream);
        case 19: return launch_bin<19>(p, stream);
    }
    return hipErrorInvalidValue;
}

hipError_t fh_k1_launch_reduce(const ReduceParams &rp, double *stats_sum, double *stats_minmax,
                               hipStream_t stream) {
    const int64_t ne = (int64_t)rp.ntiles * 256;
    hipLaunchKernelGGL(reduce_partials_level1, dim3((unsigned)((ne + 255) / 256), kReduceGroups), dim3(256), 0, stream, rp,
                       rp.scratch);
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, stream, rp, rp.scratch,
                       stats_sum, stats_minmax);
    return hipGetLastError();
}

hipError_t fh_k1_launch_finalize(const double *stats_sum, int NBT, int N, const double *a, double *M, double *j,
                                 double *sumwV2, hipStream_t stream) {
    hipLaunchKernelGGL(finalize_stats_kernel, dim3(ntiles(NBT)), dim3(256), 0, stream, stats_sum, NBT, N, a, M, j,
                       sumwV2);
    return hipGetLastError();
}

hipError_t fh_k1_launch_coefficients(const double *q, int64_t n, int N, const double *zeros, const double *pref,
                                     double inv_Q, double scale, const double *j0_table, double *H,
                                     hipStream_t stream) {
    const int64_t total = n * (int64_t)N;
    int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(coefficients_kernel, dim3(grid), dim3(256), 0, stream, q, n, N, zeros, pref, inv_Q, scale,
                       j0_table, H);
    return hipGetLastError();
}

hipError_t fh_k1_launch_predict(const double *q, int64_t n, int N, const double *zeros, const double *pref,
                                double inv_Q, double scale, const double *I, const double *j0_table, double *V,
                                hipStream_t stream) {
    int grid = (int)((n + 3) / 4 < 4096 ? (n + 3) / 4 : 4096);
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(predict_kernel, dim3(grid), dim3(256), 0, stream, q, n, N, zeros, pref, inv_Q, scale, I,
                       j0_table, V);
    return hipGetLastError();
}

hipError_t fh_k1_launch_wide_rows(const BinParams &p, int64_t first, int64_t rows, double *X, hipStream_t stream) {
    const int64_t total = rows * (int64_t)(p.N + 1);
    int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(wide_rows_kernel, dim3(grid), dim3(256), 0, stream, p.prep_s, p.prep_sw, p.prep_swV, p.prep_k2,
                       p.H2, first, rows, p.N, p.zeros, p.j0_table, X);
    return hipGetLastError();
}

hipError_t fh_k1_launch_wide_scalars(const double *partial_scalars, int blocks, double *tail, double *stats_minmax,
                                     hipStream_t stream) {
    hipLaunchKernelGGL(wide_scalars_kernel, dim3(1), dim3(64), 0, stream, partial_scalars, blocks, tail, stats_minmax);
    return hipGetLastError();
}

hipError_t fh_k1_launch_wide_finalize(const double *G, int N, const double *a, double *M, double *j, double *sumwV2,
                                      hipStream_t stream) {
    const int64_t total = (int64_t)(N + 1) * (N + 1);
    hipLaunchKernelGGL(wide_finalize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, G, N, a, M, j,
                       sumwV2);
    return hipGetLastError();
}
