// K2 `fit_loop`: the whole FrankFitter power-spectrum iteration in ONE persistent single-workgroup kernel.
//
// Reference: FrankFitter._fit (radial_fitters.py:737-832) + GaussianModel (statistical_models.py:700-760) +
// CriticalFilter.update_power_spectrum / check_convergence (filter.py:154-181).
//
// Formulation (DESIGN.md "K2").  With Y = DHT.coefficients() (constant, cond ~ 2e2) the reference needs, per
// iteration,  Dinv = M + Y^T P^-1 Y,  mu = Dinv^-1 j,  Tr1 = (Y mu)^2,  Tr2 = diag(Y Dinv^-1 Y^T).  Writing
// Dinv = Y^T (A + P^-1) Y with the constant  A = Y^-T M Y^-1,  b = Y^-T j  gives
//     C = A + diag(1/p),   m = C^-1 b = Y mu,   Tr1 = m^2,   Tr2 = diag(C^-1),
// i.e. per iteration only a diagonal update, one Cholesky C = L L^T (N^3/6 MACs) and one triangular inverse
// W = L^-1 (N^3/6) -- Tr2_i = sum_r W[r,i]^2 and m = W^T (W b) -- instead of the reference's 13 N^3/6.
// The brightness mu = Y^-1 m is formed once at the end (and per iteration only for the diagnostics).
// The iteration is sequential and tiny (1.8e7 flop), so it is latency-bound: everything runs inside one
// 768-thread workgroup (12 waves, s_barrier only), matrices stay in L2, the 16x16 tile products run on
// v_mfma_f64_16x16x4_f64.  One launch per fit, no host round trip; fits are independent, so many of these
// kernels run concurrently (one CU each) beside the bin_gram kernel of the next fit.
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "band_scan.h"
#include "tile_chol.h"
#include <cstdlib>

namespace {

using namespace tilechol;

#ifdef FIT_LOOP_TIMING
#define TSTAMP(ph) do { if (threadIdx.x == 0) { long long now_ = clock64(); P.timing[ph] += now_ - t_last; t_last = now_; } } while (0)
#else
#define TSTAMP(ph) do { } while (0)
#endif
#ifdef FIT_LOOP_TIMING
#define OSTAMP(slot) do { if (threadIdx.x == 0) { long long n3_ = clock64(); P.timing[slot] += n3_ - o_last; o_last = n3_; } } while (0)
#else
#define OSTAMP(slot) do { } while (0)
#endif
#ifdef FIT_LOOP_TIMING
// per-wave time stamps of ONE pass behind timing[16] (tools/k2_quick.py prints the timeline of the steps)
#define TRACE(slot) do { if (P.trace_on && lane == 0 && k < 20) P.timing[16 + (wave * 20 + k) * 6 + (slot)] = clock64(); } while (0)
#else
#define TRACE(slot) do { } while (0)
#endif
#ifdef FIT_LOOP_TIMING
#define FSTAMP(slot) do { if (threadIdx.x == 64 * K2_CHAIN_WAVE) { long long n2_ = clock64(); P.timing[slot] += n2_ - f_last; f_last = n2_; } } while (0)
#else
#define FSTAMP(slot) do { } while (0)
#endif
#ifdef FIT_LOOP_TIMING
#define ISTAMP(ph) do { if (threadIdx.x == 0) { long long now_ = clock64(); timing[ph] += now_ - t_last; t_last = now_; } } while (0)
#else
#define ISTAMP(ph) do { } while (0)
#endif
#ifdef FIT_LOOP_TIMING
#define WSTAMP(slot) do { if (threadIdx.x == 64) { long long n2_ = clock64(); P.timing[slot] += n2_ - w_last; w_last = n2_; } } while (0)
#else
#define WSTAMP(slot) do { } while (0)
#endif

// Twelve waves, three per SIMD (168 registers each): the worker waves wait for L2 round trips (~850 cycles per tile load against
// 4 x 64 cycles of matrix pipe per product) with the operands of two products in flight each, so what a pass needs is more of
// them.  Eight waves (256 registers): 164 us per pass at N = 300; twelve: 153; sixteen (128 registers, 105 of them spilled): 153.
// Below ~N = 100 the three agree to 1 %.  What made more than eight waves pay: the parameter struct out of scratch memory
// (see the slot index below) and the scan's addresses formed where they are used.
#ifndef FIT_LOOP_THREADS
#define FIT_LOOP_THREADS 768
#endif
constexpr int KT = FIT_LOOP_THREADS;
constexpr int NW = KT / 64;
// Trailing-update workers.  At first the waves that share wave 0's SIMD sat the trailing update out (fp64 VALU and MFMA
// share a SIMD's DP units and wave 0's serial chain was the critical path); with the rows of the inverse merged into the
// steps the workers are the longer side, so every wave but wave 0 works (K2_ALL_WORK 0 restores the old split).
#ifndef K2_ALL_WORK
#define K2_ALL_WORK 1
#endif
constexpr int NWK = K2_ALL_WORK ? NW - 1 : NW - NW / 4;
// The wave that runs the serial factor-and-invert chain (any: with twelve waves every SIMD holds three; with 11 waves, 704
// threads, wave 3 shares its SIMD with one mate instead of two -- measured, 4 % slower than twelve waves all the same).
// inverse tiles with at least this many products go through the ring of hand-issued loads (>= K2_RING_SETS)
#ifndef K2_RING_MIN
#define K2_RING_MIN 2
#endif
// register sets of that ring (2 or 3): 4 sets 97.4 ms per fit, 3 sets 94.0, 2 sets 92.2 -- what pays is the exact wait and the
// missing register copies, not the depth; every register the inverse tiles do not hold is worth more than a load in flight
#ifndef K2_RING_SETS
#define K2_RING_SETS 2
#endif
#ifndef K2_CHAIN_WAVE
#define K2_CHAIN_WAVE 3
#endif
constexpr int kChain = K2_CHAIN_WAVE;
constexpr int kMaxTiles = 210;  // tiles of the largest trailing triangle: NP / 16 - 1 = 20 block rows (NP <= 336)
// WIDE (336 < NP <= 640, N <= 639): two panels of NP x 17 doubles no longer fit the 160 KB of LDS beside the vectors and the scan
// tables.  ONE panel then: the tiles of column k + 1 are computed into registers, a second barrier of the step lets everybody
// finish reading panel k, and only then panel k + 1 overwrites it.  Two barriers per step instead of one; N <= 320 is untouched
// (its own instantiation).  Before this, 320 < N <= 512 ran the library loop: ~8 x the time per pass at N = 321.
constexpr int kMaxTilesWide = 780;  // 39 block rows
constexpr int kWideMinNP = 337;  // (up to NP = 640)
// XWIDE (640 < NP <= 1024, N <= 1023; WIDE = 2): the one panel alone takes 139 KB at NP = 1024.  The vectors of the outer loop (p,
// m, Tr2, the right-hand side, ..: 8 NP doubles) join the band factors and scan tables in global memory (L2), the tile table is
// computed instead of stored, the wave scan takes sixteen rows per lane.  Before this, N >= 640 ran the library loop (rocBLAS +
// rocSOLVER per pass, ~2.5 ms): a 2.4 x cliff at N = 640.
constexpr int kXWideMinNP = 641, kXWideMaxNP = 1024;
constexpr int kHandMaxNP = 320;  // cluster mode: the four hand-over tiles (8 KB) fit the LDS beside two panels up to here
template <int WIDE> constexpr int max_tiles() { return WIDE == 2 ? 0 : (WIDE ? kMaxTilesWide : kMaxTiles); }
template <int WIDE> constexpr int npanels() { return WIDE ? 1 : 2; }

struct Smem {
    double *pan;   // panels of the current and the next block column: 2 x NP x PS doubles
    double *dli;   // 16 x PS: its inverse (A operand of the MFMA panel solve)
    double *p, *pold, *m, *y, *tr2, *rhs, *b, *red;  // NP each
    double *scanQ; // transfer matrices of the banded recurrences at the six levels of the wave scan (per fit)
    double *band;  // 6 NP: LU factors of the pentadiagonal T + I (five bands) and the reciprocal pivots, staged once per fit
    uint4 *rec;    // tile e = (i - 1) i / 2 + (j - 1), 1 <= j <= i, of the trailing triangle right of its first column (block
                   // (k+1+i, k+1+j) at step k), built once per launch: x = packed byte offset of the MIRROR tile (j, i) relative
                   // to block (k+1, k+1) (step 0 reads the transposes from A) | bit 1: diagonal tile; y, z = byte offsets of
                   // panel row blocks j, i (A and B operand of the transposed update); w = packed offset of tile (i, j) | i
    int *flag;
    int *dcnt;     // deferred mode (CLM = 4): entries of the two tile tables with i <= row, [parity][24]
    double *hand;  // cluster mode: 2 x 2 packed tiles a worker hands to the chain wave through LDS (solve_posterior_cluster)
};

// ---- (4) row-by-row inverse: W_IJ = -W_II * sum_{K=J}^{I-1} L_IK W_KJ --------------------------------------------------
// The first design was a recursive, GEMM-shaped block inverse whose temporary T was written and read back at every
// level (~8 MB of L2 traffic per iteration, 156 us); the second computed the rows one after the other BEHIND the
// factorisation (a row's L tiles staged in LDS, the sum kept in registers and multiplied by W_II straight from the
// accumulator, 123 us).  Row I only needs rows <= I of L and rows < I of W, so it is now computed DURING step I of the
// factorisation: the trailing update shrinks quadratically with the step while the rows of the inverse grow, and the
// two together keep the waves about evenly busy (the factorisation alone left them idle for most of its second half).
// One tile W_IJ.  The A operand L_IK is read from its mirror block (K, I) of C in row form (4 rows x 128 contiguous
// bytes), like every other operand; a first version staged row I of L in LDS, transposed, once per step: the staging loads
// sat in front of the panel (their registers and the in-order vmcnt tied them to the spill reloads: ~1 us per step) and
// took 39 KB of LDS.  `fw` is the A operand W_II (rows of its transpose).  All addresses: uniform base + 32-bit offset.
__device__ __forceinline__ void inverse_tile(const gdouble *Cu, const double *dli_I, gdouble *Wu, double *cs_IJ, int I, int J,
                                             int N, int nb, int lane) {
    const int rg = lane >> 4;
    const unsigned blk = (unsigned)(nb * 2048);                 // one block row further, packed tiles
    unsigned oa = (unsigned)((J * nb + I) * 2048);              // tile (K, I) of C (the mirror, L_IK^T), K = J
    unsigned ob = (unsigned)((J * nb + J) * 2048);              // tile (K, J) of W
    v4f64 acc = {0.0, 0.0, 0.0, 0.0};
    // A ring of register sets (K2_RING_SETS, two by default: one product in flight behind the one being multiplied).  The
    // loads are issued and waited for by hand (uniform base + 32-bit offset; vmcnt counts in issue order: product p is complete
    // when at most 4 x (products issued after it) loads are outstanding): the compiler's version -- two named sets, copied at
    // the top of every trip behind a vmcnt(0) -- had one L2 round trip and 32 register moves per two products on the path of a
    // chain of up to nb - 1 of them, and the late steps of a pass wait for exactly that chain.  Same order of summation, same bits.
    struct Operands {
        v2f64 alo, ahi, blo, bhi;
    };
    const int n = I - J;  // products
    const unsigned lane_o = (unsigned)lane * 16u;
    auto issue = [&](Operands &r, int p) {
        const unsigned pa = oa + (unsigned)p * blk + lane_o, pb = ob + (unsigned)p * blk + lane_o;
        asm volatile(
            // (s_nop: a base pointer the compiler has just restored from a spill lane with v_readlane needs five wait states before
            //  a memory instruction may read it as its scalar address, and the hazard recogniser does not look into inline asm --
            //  a timing build that spilled scalar registers here faulted on exactly that)
            "s_nop 4\n\t"
            "global_load_dwordx4 %0, %4, %6\n\tglobal_load_dwordx4 %1, %4, %6 offset:1024\n\t"
            "global_load_dwordx4 %2, %5, %7\n\tglobal_load_dwordx4 %3, %5, %7 offset:1024"
            : "=&v"(r.alo), "=&v"(r.ahi), "=&v"(r.blo), "=&v"(r.bhi)
            : "v"(pa), "v"(pb), "s"(Cu), "s"(Wu)
            : "memory");
    };
    auto consume = [&](Operands &r, int p) {
        switch (min(K2_RING_SETS - 1, n - 1 - p)) {  // products issued after p
            case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            case 1: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        }
        asm volatile("" : "+v"(r.alo), "+v"(r.ahi), "+v"(r.blo), "+v"(r.bhi));  // (used behind the wait)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(r.alo[0], r.blo[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(r.alo[1], r.blo[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(r.ahi[0], r.bhi[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(r.ahi[1], r.bhi[1], acc, 0, 0, 0);
    };
    if (n <= K2_RING_MIN - 1) {  // short chains: plain loads, the compiler's waits (nothing to pipeline)
        for (int p = 0; p < n; ++p) {
            const v4f64 ta = ld_pk(Cu, oa + (unsigned)p * blk, lane), tb = ld_pk(Wu, ob + (unsigned)p * blk, lane);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ta[q], tb[q], acc, 0, 0, 0);
        }
    } else {
#if K2_RING_SETS == 2
        Operands s0, s1;
        issue(s0, 0);
        for (int p = 0;; p += 2) {
            if (p + 1 < n) issue(s1, p + 1);
            consume(s0, p);
            if (p + 1 >= n) break;
            if (p + 2 < n) issue(s0, p + 2);
            consume(s1, p + 1);
            if (p + 2 >= n) break;
        }
#else
        Operands s0, s1, s2;
        issue(s0, 0);
        issue(s1, 1);
        for (int p = 0;; p += 3) {
            if (p + 2 < n) issue(s2, p + 2);
            consume(s0, p);
            if (p + 1 >= n) break;
            if (p + 3 < n) issue(s0, p + 3);
            consume(s1, p + 1);
            if (p + 2 >= n) break;
            if (p + 4 < n) issue(s1, p + 4);
            consume(s2, p + 2);
            if (p + 3 >= n) break;
        }
#endif
    }
    // the C/D layout of acc (row = rg + 4 r, col = cl) is the B-operand layout (k = 4 s + rg, j = cl)
    Frag fs;
#pragma unroll
    for (int q = 0; q < 4; ++q) fs.v[q] = acc[q];
    Frag fw;  // A operand W_II = L_II^-1, read from LDS only now: eight registers less across the chain
    {
        const int cl = lane & 15;
#pragma unroll
        for (int q = 0; q < 4; ++q) fw.v[q] = dli_I[cl * PS + 4 * q + rg];
    }
    v4f64 w = {0.0, 0.0, 0.0, 0.0};
    w = mfma4(fw, fs, w, true);
    st_pk(Wu, (unsigned)((I * nb + J) * 2048), lane, w);
    double ssq = 0.0;  // column sums of squares of this (final) tile over the rows of the real system
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (16 * I + rg + 4 * r < N) ssq = fma(w[r], w[r], ssq);
    ssq += __shfl_xor(ssq, 16);
    ssq += __shfl_xor(ssq, 32);
    if (rg == 0) cs_IJ[lane & 15] = ssq;
}

// TWO rows of the inverse at once (round 5, deferred mode): W_{I-1,J} and W_{I,J}.  With the device full a pass is bound by the
// bytes it moves beyond the L2, and the row-by-row inverse reads every tile of W so far once per ROW: 1 330 tile loads of W per
// pass at N = 300 that nothing else shares (the tiles of L of a row are read by every column's wave and come from the L1).
// Here the sums of the two rows run side by side over K = J .. I - 2 with ONE load of W_KJ for two products; row I then takes
// its last product, L_{I,I-1} W_{I-1,J}, with W_{I-1,J} straight from the registers it was formed in (the accumulator layout is
// the B-operand layout).  The same products in the same order into each accumulator as inverse_tile: the same bits.
// Column J pairs its rows (J + 1, J + 2), (J + 3, J + 4), ..: at step I the columns with J + I even are due; a last single row
// is left to inverse_tile at the last step.  dli_a, dli_b: L^-1 of the diagonal tiles I - 1 and I (A operands, from LDS).
__device__ __forceinline__ void inverse_pair(const gdouble *Cu, const double *dli_a, const double *dli_b, gdouble *Wu, double *cs_a,
                                             double *cs_b, int I, int J, int N, int nb, int lane) {
    const int rg = lane >> 4, cl = lane & 15;
    const unsigned blk = (unsigned)(nb * 2048);
    const unsigned oa = (unsigned)((J * nb + I - 1) * 2048);  // tile (K, I - 1) of C = L_{I-1,K}^T; tile (K, I) is the next one
    const unsigned ob = (unsigned)((J * nb + J) * 2048);      // tile (K, J) of W, K = J
    const int n = I - 1 - J;                                  // shared products, K = J .. I - 2 (>= 1)
    v4f64 acc1 = {0.0, 0.0, 0.0, 0.0}, acc2 = {0.0, 0.0, 0.0, 0.0};
    struct Operands {
        v2f64 a1lo, a1hi, a2lo, a2hi, blo, bhi;
    };
    const unsigned lane_o = (unsigned)lane * 16u;
    auto issue = [&](Operands &r, int p) {
        const unsigned pa = oa + (unsigned)p * blk + lane_o, pb = ob + (unsigned)p * blk + lane_o;
        asm volatile(
            "s_nop 4\n\t"  // (see inverse_tile)
            "global_load_dwordx4 %0, %6, %8\n\tglobal_load_dwordx4 %1, %6, %8 offset:1024\n\t"
            "global_load_dwordx4 %2, %6, %8 offset:2048\n\tglobal_load_dwordx4 %3, %6, %8 offset:3072\n\t"
            "global_load_dwordx4 %4, %7, %9\n\tglobal_load_dwordx4 %5, %7, %9 offset:1024"
            : "=&v"(r.a1lo), "=&v"(r.a1hi), "=&v"(r.a2lo), "=&v"(r.a2hi), "=&v"(r.blo), "=&v"(r.bhi)
            : "v"(pa), "v"(pb), "s"(Cu), "s"(Wu)
            : "memory");
    };
    auto consume = [&](Operands &r, int p) {
        if (p + 1 < n) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");  // (one later product in flight: six loads)
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("" : "+v"(r.a1lo), "+v"(r.a1hi), "+v"(r.a2lo), "+v"(r.a2hi), "+v"(r.blo), "+v"(r.bhi));
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(r.a1lo[0], r.blo[0], acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(r.a2lo[0], r.blo[0], acc2, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(r.a1lo[1], r.blo[1], acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(r.a2lo[1], r.blo[1], acc2, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(r.a1hi[0], r.bhi[0], acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(r.a2hi[0], r.bhi[0], acc2, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(r.a1hi[1], r.bhi[1], acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(r.a2hi[1], r.bhi[1], acc2, 0, 0, 0);
    };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the counts below are of THESE loads)
    Operands s0, s1;
    issue(s0, 0);
    for (int p = 0;; p += 2) {
        if (p + 1 < n) issue(s1, p + 1);
        consume(s0, p);
        if (p + 1 >= n) break;
        if (p + 2 < n) issue(s0, p + 2);
        consume(s1, p + 1);
        if (p + 2 >= n) break;
    }
    const v4f64 tl = ld_pk(Cu, (unsigned)(((I - 1) * nb + I) * 2048), lane);  // L_{I,I-1}^T, the A operand of row I's last product
    auto finish = [&](const v4f64 &acc, const double *dli, int R, double *cs) {
        Frag fs, fw;
#pragma unroll
        for (int q = 0; q < 4; ++q) fs.v[q] = acc[q];
#pragma unroll
        for (int q = 0; q < 4; ++q) fw.v[q] = dli[cl * PS + 4 * q + rg];
        v4f64 w = {0.0, 0.0, 0.0, 0.0};
        w = mfma4(fw, fs, w, true);
        st_pk(Wu, (unsigned)((R * nb + J) * 2048), lane, w);
        double ssq = 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (16 * R + rg + 4 * r < N) ssq = fma(w[r], w[r], ssq);
        ssq += __shfl_xor(ssq, 16);
        ssq += __shfl_xor(ssq, 32);
        if (rg == 0) cs[lane & 15] = ssq;
        return w;
    };
    const v4f64 w1 = finish(acc1, dli_a, I - 1, cs_a);
#pragma unroll
    for (int q = 0; q < 4; ++q) acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(tl[q], w1[q], acc2, 0, 0, 0);
    (void)finish(acc2, dli_b, I, cs_b);
}

// ---- cluster ("latency") mode: the block columns of the inverse on helper workgroups ---------------------------------------
// One fit on one CU is a chain: factor-and-invert of a diagonal tile, the column tiles, a barrier -- 19 times per pass -- with
// the trailing update and the rows of the inverse filling the time in between; half of a pass's matrix instructions (1 311 of
// 2 622 tile products at N = 300) are the inverse W = L^-1, which nothing inside the factorisation waits for.  Column J of W
// (the tiles W_JJ, W_{J+1,J}, ..) depends on L and on ITSELF only: W_rJ = -X_rr sum_{K=J}^{r-1} L_rK W_KJ.  So in cluster mode
// `cluster - 1` helper workgroups of the same XCD (one L2) take the columns, ONE WAVE PER COLUMN, no barrier anywhere: a wave
// waits for the first workgroup's progress word (row r of L is final when column r - 1 is: value r; X_rr: value r + 1), adds up
// the products of its tile -- the same products in the same order as inverse_tile: the same bits --, multiplies by X_rr when
// that is published, keeps the column's sum of squares (Tr2) and the entry of row N (m = Y mu) and hands them over at the end of
// the pass.  The first workgroup publishes once per step: every wave waits for its stores (s_waitcnt vmcnt(0): the barrier
// itself only waits for LDS), thread 0 raises the word behind the barrier with an agent-scope store -- executed in the L2 the
// helpers share --, and a helper invalidates its L1 when it has seen the value it needs.  Workgroup ids 0, 8, 16, .. of a
// launch go to one XCD; the XCC_ID register of every member is checked at start-up and a cluster that is not on one XCD, or
// whose helpers do not show up within 3 ms, ends with FIT_STATUS_CLUSTER (the host runs the fit on one CU then).  Every wait
// is bounded by the wall clock.
namespace clu {
// control words (ints), zero between fits: PROG progress of the factorisation, DONE helper waves that have handed over their
// columns, IN helper workgroups present, XCC the XCDs the members sit on; HCOL + J: trailing tiles of block column J the helpers
// have handed back (cumulative over the passes)
enum { PROG = 0, DONE = 32, IN = 33, XCC = 34, HCOL = 64, NCTL = 64 + 64 };  // (128-byte lines: the word every helper polls, the
                                                                              //  rarely written ones, the counters of the hand-back)
constexpr int kSeq = 128;  // progress word = pass * kSeq + block columns of L that are final (<= 64)
// exchange area (the fit's WdT buffer): [0, NP) Tr2 and [NP, 2 NP) m from the helpers, [2 NP, 3 NP) 1 / p from the first
// workgroup, then the control words
constexpr int kBand = 2;    // block columns right of the panel the first workgroup updates itself (see trailing_wave)
constexpr int kTMax = 7;    // trailing tiles a helper wave keeps in registers at most (N <= 335 with two helpers' 24 waves: 153)
// Members of a cluster: 0 the first workgroup, 1 .. inv the helpers of the inverse, inv + 1 .. cluster - 1 the helpers of the
// trailing update.  A wave that held trailing tiles AND ran the load ring of an inverse column spilled both to scratch memory
// (the spill traffic counts in the ring's hand-written waits): 20 us behind the factorisation at the end of a pass.
__device__ __forceinline__ int inv_helpers(const FitLoopParams &P) { return P.cluster_inv; }
__device__ __forceinline__ int trail_helpers(const FitLoopParams &P) { return P.cluster - 1 - P.cluster_inv; }
__device__ __forceinline__ int ld(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int add(int *p, int v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xf; }  // HW_REG_XCC_ID
// The receiving side of a hand-over reads what another compute unit wrote with DEVICE-SCOPE loads (sc1: served by the L2 the
// members of a cluster share, never by this compute unit's L1) -- no invalidate.  What was tried before: the agent-scope
// invalidate (buffer_inv sc1) also writes the L2's dirty lines back (843 MB to memory per fit against 127 MB on one workgroup)
// and cost 4-15 us per pass, in two modes from run to run; the workgroup-scope one (buffer_inv sc0) leaves the L1 alone outside
// the threadgroup-split mode -- helpers then read LAST pass's tiles wherever a compute unit's share of the matrices is small
// enough to stay in its L1 (N < 128 with five workgroups, found by tools/size_sweep_cluster.py).
__device__ __forceinline__ v4f64 ld_pk_dev(const gdouble *base, unsigned tile_byte_off, int lane) {
    v2f64 lo, hi;
    const unsigned a = tile_byte_off + (unsigned)lane * 16u;
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %2, %3 sc1\n\tglobal_load_dwordx4 %1, %2, %3 offset:1024 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(lo), "=&v"(hi)
                 : "v"(a), "s"(base)
                 : "memory");
    return v4f64{lo[0], lo[1], hi[0], hi[1]};
}
// issue only: the caller waits (s_waitcnt vmcnt) and fences the registers before it reads them
__device__ __forceinline__ void ld_pk_dev_issue(const gdouble *base, unsigned tile_byte_off, int lane, v2f64 &lo, v2f64 &hi) {
    const unsigned a = tile_byte_off + (unsigned)lane * 16u;
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %2, %3 sc1\n\tglobal_load_dwordx4 %1, %2, %3 offset:1024 sc1"
                 : "=&v"(lo), "=&v"(hi)
                 : "v"(a), "s"(base)
                 : "memory");
}
// two tiles with one wait
__device__ __forceinline__ void ld_pk_dev2(const gdouble *ba, unsigned oa, const gdouble *bb, unsigned ob, int lane, v4f64 &ta, v4f64 &tb) {
    v2f64 alo, ahi, blo, bhi;
    const unsigned a = oa + (unsigned)lane * 16u, b = ob + (unsigned)lane * 16u;
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %4, %6 sc1\n\tglobal_load_dwordx4 %1, %4, %6 offset:1024 sc1\n\t"
                 "global_load_dwordx4 %2, %5, %7 sc1\n\tglobal_load_dwordx4 %3, %5, %7 offset:1024 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(alo), "=&v"(ahi), "=&v"(blo), "=&v"(bhi)
                 : "v"(a), "v"(b), "s"(ba), "s"(bb)
                 : "memory");
    ta = v4f64{alo[0], alo[1], ahi[0], ahi[1]};
    tb = v4f64{blo[0], blo[1], bhi[0], bhi[1]};
}
__device__ __forceinline__ double ld_dev(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
constexpr long long kTicksPerUs = 100;  // wall_clock64: 100 MHz
__device__ __forceinline__ int *ctl_of(const FitLoopParams &P) { return reinterpret_cast<int *>(P.WdT + 3 * (size_t)P.NP); }
// block columns the first workgroup updates itself at every step; the tiles further right belong to the helper waves until
// their column enters the band.  nb (everything) when the helper waves cannot hold the tiles (few helpers, wide systems).
__device__ __forceinline__ int band_of(const FitLoopParams &P) {
    const int nb = P.NP / 16, T = trail_helpers(P) * NW;
    const int first = 2 + kBand, cols = nb - first;
    const int ntl = cols > 0 ? cols * (cols + 1) / 2 : 0;
    return (ntl > 0 && ntl <= kTMax * T) ? kBand : nb;
}
// one wave: wait until the progress word reaches `need`; false: the fit is over (word < 0) or nothing was heard for 2 s
__device__ __forceinline__ bool wait_prog(const int *ctl, int need, int &seen) {
    if (seen >= need) return true;
    const long long t0 = wall_clock64();
    for (;;) {
        const int v = __builtin_amdgcn_readfirstlane(ld(ctl + PROG));
        if (v < 0) return false;
        if (v >= need) {
            seen = v;
            return true;
        }
        __builtin_amdgcn_s_sleep(1);
        if (wall_clock64() - t0 > 2000000 * kTicksPerUs) return false;
    }
}
// sum_{K=J}^{J+n-1} L_rK W_KJ: the chain of inverse_tile (same ring of hand-issued loads, same order of summation)
__device__ __forceinline__ v4f64 chain_sum(const gdouble *Cu, const gdouble *Wu, int r, int J, int n, int nb, int lane) {
    const unsigned blk = (unsigned)(nb * 2048);
    const unsigned oa = (unsigned)((J * nb + r) * 2048), ob = (unsigned)((J * nb + J) * 2048);
    v4f64 acc = {0.0, 0.0, 0.0, 0.0};
    struct Operands {
        v2f64 alo, ahi, blo, bhi;
    };
    const unsigned lane_o = (unsigned)lane * 16u;
    auto issue = [&](Operands &o, int p) {
        const unsigned pa = oa + (unsigned)p * blk + lane_o, pb = ob + (unsigned)p * blk + lane_o;
        asm volatile(
            // (s_nop: a base pointer the compiler has just restored from a spill lane with v_readlane needs five wait states before
            //  a memory instruction may read it as its scalar address, and the hazard recogniser does not look into inline asm --
            //  a timing build that spilled scalar registers here faulted on exactly that)
            "s_nop 4\n\t"
            "global_load_dwordx4 %0, %4, %6 sc1\n\tglobal_load_dwordx4 %1, %4, %6 offset:1024 sc1\n\t"  // (the first workgroup's tiles)
            "global_load_dwordx4 %2, %5, %7\n\tglobal_load_dwordx4 %3, %5, %7 offset:1024"  // (this wave's own)
            : "=&v"(o.alo), "=&v"(o.ahi), "=&v"(o.blo), "=&v"(o.bhi)
            : "v"(pa), "v"(pb), "s"(Cu), "s"(Wu)
            : "memory");
    };
    auto consume = [&](Operands &o, int p) {
        switch (min(3, n - 1 - p)) {  // products issued after p
            case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            case 1: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        }
        asm volatile("" : "+v"(o.alo), "+v"(o.ahi), "+v"(o.blo), "+v"(o.bhi));
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(o.alo[0], o.blo[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(o.alo[1], o.blo[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(o.ahi[0], o.bhi[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(o.ahi[1], o.bhi[1], acc, 0, 0, 0);
    };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the counts below are of THESE loads: nothing of the wave's own in flight)
    // four register sets, three products in flight behind the one being multiplied: a helper wave has registers to spare, and
    // what its column waits for in the last rows of a pass is the latency of these loads (two sets: the helpers ended ~15 us
    // behind the factorisation)
    Operands s0, s1, s2, s3;
    issue(s0, 0);
    if (1 < n) issue(s1, 1);
    if (2 < n) issue(s2, 2);
    for (int p = 0;; p += 4) {
        if (p + 3 < n) issue(s3, p + 3);
        consume(s0, p);
        if (p + 1 >= n) break;
        if (p + 4 < n) issue(s0, p + 4);
        consume(s1, p + 1);
        if (p + 2 >= n) break;
        if (p + 5 < n) issue(s1, p + 5);
        consume(s2, p + 2);
        if (p + 3 >= n) break;
        if (p + 6 < n) issue(s2, p + 6);
        consume(s3, p + 3);
        if (p + 4 >= n) break;
    }
    return acc;
}
// A wave of a helper of the TRAILING UPDATE, for every pass of the fit (hw: its number among the T such waves, counted ACROSS
// the workgroups -- wave w of helper h is number w H + h).  It owns the trailing tiles number hw, hw + T, .. of the enumeration
// (J = 2 + band .., I = J ..) of the tiles right of the band: tile (I, J) takes the updates k = 0 .. J - 2 - band here -- in
// REGISTERS, from the mirror tiles (k, J), (k, I) of C as they are published, the same products in the same order as the first
// workgroup's trailing update --, goes back through memory when its column enters the band (the upper tiles of the W buffer,
// the diagonal ones in the cs buffer) and takes its last band + 1 updates there.  An event is "column c of L is final".
// OPERANDS THROUGH LDS (round 4, second half).  Every product read its two operand tiles (c, J), (c, I) from the L2 with
// device-scope loads and waited for them: ~0.9 us per product in situ (an in-kernel timeline of one helper wave; 0.22 us for an
// undisturbed pair, tools/microbench/xcu_latency.hip -- but the same few tiles of row c are asked for by up to 24 waves at
// once and never served from the L1), 7 us for the first events of a pass, and the first workgroup waited ~1.1 us of every
// 3.4 us step for the column that enters its band.  Now the workgroup reads row c ONCE per event -- each wave one or two of
// the <= 15 tiles (c, x), x >= c + 2 + band, into a double-buffered 30 KB stage in LDS, one barrier -- and every product takes
// its operands from there.  Same products in the same order per tile: same bits.
__device__ __forceinline__ void trailing_wave(const FitLoopParams &P, int *ctl, int hw, int T, int lane, double *stage, int wv) {
    const int nb = P.NP / 16;
    const int cl = lane & 15, rg = lane >> 4;
    const gdouble *Cg = as_global(uniform_ptr(const_cast<const double *>(P.C)));
    const gdouble *Ag = as_global(uniform_ptr(P.A));
    gdouble *Wg = as_global(uniform_ptr(P.W));
    gdouble *Dg = as_global(uniform_ptr(P.cs));
    const double *xg = P.WdT;
    const int band = band_of(P);
    const int first = 2 + band, last = nb - 1 - first;  // the events: columns first .. nb - 1 leave at events 0 .. last
    int seen = 0;
    int tI[kTMax], tJ[kTMax];
#pragma unroll
    for (int i = 0; i < kTMax; ++i) {
        int t = hw + i * T, J = first;
        while (J < nb && t >= nb - J) {
            t -= nb - J;
            ++J;
        }
        tJ[i] = J < nb ? J : -1;
        tI[i] = J + t;
    }
    v4f64 tt[kTMax];
    auto first_touch = [&]() {  // tile (I, J) holds T_IJ^T, the tile (J, I) of the symmetric A (the constant of the fit: no wait)
#pragma unroll
        for (int i = 0; i < kTMax; ++i)
            if (tJ[i] >= 0) tt[i] = ld_pk(Ag, (unsigned)((tJ[i] * nb + tI[i]) * 2048), lane);
    };
    first_touch();
    if (last < 0) {  // no band (too few helper waves for the tiles: band_of): nothing to do but leave with the others
        while (wait_prog(ctl, 0x7fffffff, seen)) {
        }
        return;
    }
    const int stage_tiles = nb - first;  // (per buffer)
    volatile int *wflag = reinterpret_cast<volatile int *>(stage + 2 * (size_t)stage_tiles * 256);  // (behind the two buffers)
    for (int seq = 1;; ++seq) {
        for (int c = 0; c <= last; ++c) {
            // ONE wave of the workgroup polls the progress word (sixty helper waves polling the same line slowed every poll of
            // it, the first workgroup's included); the others hear of it at a barrier
            if (wv == 0) {
                const bool ok = wait_prog(ctl, seq * clu::kSeq + c + 1, seen);
                if (lane == 0) *wflag = ok ? 1 : 0;
            }
            __syncthreads();
            if (*wflag == 0) return;
            const int x0 = c + first;
            double *buf = stage + (size_t)(c & 1) * stage_tiles * 256;
            {   // (at most two tiles per wave: nb - x0 <= 2 NW; both loads in flight together)
                const int xa = x0 + wv, xb = xa + NW;
                if (xb < nb) {
                    v4f64 va, vb;
                    ld_pk_dev2(Cg, (unsigned)((c * nb + xa) * 2048), Cg, (unsigned)((c * nb + xb) * 2048), lane, va, vb);
                    v2f64 *qa = reinterpret_cast<v2f64 *>(buf + (size_t)(xa - x0) * 256) + lane;
                    v2f64 *qb = reinterpret_cast<v2f64 *>(buf + (size_t)(xb - x0) * 256) + lane;
                    qa[0] = v2f64{va[0], va[1]};
                    qa[64] = v2f64{va[2], va[3]};
                    qb[0] = v2f64{vb[0], vb[1]};
                    qb[64] = v2f64{vb[2], vb[3]};
                } else if (xa < nb) {
                    const v4f64 v = ld_pk_dev(Cg, (unsigned)((c * nb + xa) * 2048), lane);
                    v2f64 *q = reinterpret_cast<v2f64 *>(buf + (size_t)(xa - x0) * 256) + lane;
                    q[0] = v2f64{v[0], v[1]};
                    q[64] = v2f64{v[2], v[3]};
                }
            }
            __syncthreads();  // (one per event: a wave cannot be two events ahead, the buffers alternate)
            int handed = 0;
#pragma unroll
            for (int i = 0; i < kTMax; ++i) {
                if (tJ[i] < 0 || tJ[i] < x0) continue;
                if (c == 0 && tI[i] == tJ[i]) {  // diag(1 / p) (written by the first workgroup before its first word of the pass)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (rg + 4 * q == cl) tt[i][q] += ld_dev(xg + 2 * P.NP + 16 * tI[i] + cl);
                }
                const v2f64 *qa = reinterpret_cast<const v2f64 *>(buf + (size_t)(tJ[i] - x0) * 256) + lane;
                const v2f64 *qb = reinterpret_cast<const v2f64 *>(buf + (size_t)(tI[i] - x0) * 256) + lane;
                const v2f64 a0 = qa[0], a1 = qa[64], b0 = qb[0], b1 = qb[64];
                tt[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a0[0], b0[0], tt[i], 0, 0, 0);
                tt[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a0[1], b0[1], tt[i], 0, 0, 0);
                tt[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a1[0], b1[0], tt[i], 0, 0, 0);
                tt[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a1[1], b1[1], tt[i], 0, 0, 0);
                if (tJ[i] == x0) {  // the column enters the band at the next step: hand the tile back
                    if (tI[i] == tJ[i]) st_pk(Dg, (unsigned)(tJ[i] * 2048), lane, tt[i]);
                    else st_pk(Wg, (unsigned)((tJ[i] * nb + tI[i]) * 2048), lane, tt[i]);
                    ++handed;
                }
            }
            if (handed) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) add(ctl + HCOL + x0, handed);
            }
        }
        first_touch();  // (for the next pass, before the wait for it: its first event, or the end of the fit)
    }
}
// A wave of a helper of the INVERSE, for every pass of the fit: the block columns J = hw, hw + T, .. of W = L^-1.  At the event
// "column c of L and X_cc are final" it finishes row c (W_cJ = -X_cc acc1), adds the LAST product of row c + 1 from the registers
// it has just filled (acc1 = acc2 + L_{c+1,c} W_cJ) and then, off the critical path, adds up the products K = J .. c of row c + 2
// (acc2) -- all of them but the last, which needs W_{c+1,J}.  The sums run over K in ascending order as in inverse_tile (the same
// bits); what is left to do after the LAST event of a pass is one tile product instead of a chain of nb - 1.
template <int CMAX>
__device__ __forceinline__ void inverse_wave(const FitLoopParams &P, int *ctl, int hw, int T, int lane) {
    const int N = P.N, nb = P.NP / 16;
    const int cl = lane & 15, rg = lane >> 4;
    const gdouble *Cg = as_global(uniform_ptr(const_cast<const double *>(P.C)));
    gdouble *Wg = as_global(uniform_ptr(P.W));
    double *xg = P.WdT;
    const int aug_tile = N / 16, aug_r = N - 16 * aug_tile;
    int seen = 0;
    if (hw >= nb) {  // no column: wait for the end of the fit (the workgroup leaves together)
        while (wait_prog(ctl, 0x7fffffff, seen)) {
        }
        return;
    }
    Frag ident;  // B operand: the identity
#pragma unroll
    for (int q = 0; q < 4; ++q) ident.v[q] = (4 * q + rg == cl) ? 1.0 : 0.0;
    for (int seq = 1;; ++seq) {
        double t2[CMAX];
        v4f64 acc1[CMAX], acc2[CMAX];
#pragma unroll
        for (int i = 0; i < CMAX; ++i) {
            t2[i] = 0.0;
            acc1[i] = v4f64{0.0, 0.0, 0.0, 0.0};
            acc2[i] = v4f64{0.0, 0.0, 0.0, 0.0};
        }
        for (int c = hw; c < nb; ++c) {  // (the first event that concerns this wave: its first column)
            if (!wait_prog(ctl, seq * clu::kSeq + c + 1, seen)) return;
            v4f64 xp, la;  // X_cc = W_cc; mirror tile (c, c + 1): L_{c+1,c}
            ld_pk_dev2(Cg, (unsigned)((c * nb + c) * 2048), Cg, (unsigned)((c * nb + (c + 1 < nb ? c + 1 : c)) * 2048), lane, xp, la);
#pragma unroll
            for (int i = 0; i < CMAX; ++i) {
                const int J = hw + i * T;
                if (J > c || J >= nb) continue;
                v4f64 w;
                if (J == c) {
                    w = xp;
                } else {
                    // the A operand X_cc (element [cl][4 q + rg]) is the accumulator layout of X_cc^T: the packed tile used AS an
                    // A operand is X_cc^T, times the identity (exact)
                    v4f64 z = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int q = 0; q < 4; ++q) z = __builtin_amdgcn_mfma_f64_16x16x4f64(xp[q], ident.v[q], z, 0, 0, 0);
                    Frag fw, fs;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        fw.v[q] = z[q];
                        fs.v[q] = acc1[i][q];
                    }
                    w = v4f64{0.0, 0.0, 0.0, 0.0};
                    w = mfma4(fw, fs, w, true);
                }
                st_pk(Wg, (unsigned)((c * nb + J) * 2048), lane, w);
                if (c + 1 < nb) {  // the last product of row c + 1: the accumulator registers of W_cJ are its B fragments
                    acc1[i] = acc2[i];
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc1[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(la[q], w[q], acc1[i], 0, 0, 0);
                }
                double ssq = 0.0;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (16 * c + rg + 4 * q < N) ssq = fma(w[q], w[q], ssq);
                ssq += __shfl_xor(ssq, 16);
                ssq += __shfl_xor(ssq, 32);
                t2[i] += ssq;  // (rows in order, from 0.0: the sum solve_posterior forms from the tiles' column sums)
                if (c == aug_tile) {  // row N of W: -m
                    double mv = w[0];
                    mv = (aug_r >> 2) == 1 ? w[1] : mv;
                    mv = (aug_r >> 2) == 2 ? w[2] : mv;
                    mv = (aug_r >> 2) == 3 ? w[3] : mv;
                    if (rg == (aug_r & 3) && 16 * J + cl < N) xg[P.NP + 16 * J + cl] = -mv;
                }
            }
            if (c + 2 < nb) {  // off the critical path: the products K = J .. c of row c + 2 (columns <= c of L are final)
#pragma unroll
                for (int i = 0; i < CMAX; ++i) {
                    const int J = hw + i * T;
                    if (J > c || J >= nb) continue;
                    acc2[i] = chain_sum(Cg, Wg, c + 2, J, c - J + 1, nb, lane);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < CMAX; ++i) {
            const int J = hw + i * T;
            if (J < nb && rg == 0 && 16 * J + cl < N) xg[16 * J + cl] = t2[i];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the stores have reached the L2 before the count says so
        if (lane == 0) add(ctl + DONE, 1);
    }
}

// ---- TWO WAVES PER BLOCK COLUMN (round 4, second half; N <= 335, every wave at most one column) -----------------------------------
// The wave of block column J adds up, at event c, the products K = J .. c of row c + 2: c - J + 1 of them, each waiting for a
// device-scope tile of L.  At the end of a pass the first columns have ~17 per event against a step of ~2.7 us: they fall ~10 us
// behind the factorisation, and the first workgroup waits for them at the end of EVERY pass (in-kernel timeline: 12 of 76 us).
// Most waves of the helpers' workgroups have no column (36 waves, 19 columns).  So a column gets a SECOND wave of its workgroup:
// B adds up K = J .. r - 4 of row r as soon as the column's own wave A has stored W_{r-4,J} (two events before A needs the sum),
// hands the accumulator over through LDS, and A adds K = r - 3, r - 2 (at event r - 2: W from its registers, two tiles of L
// loaded with X_cc) and K = r - 1 (the "last product", as before).  A's work per event no longer grows with c.  The products
// enter every accumulator in the same ascending order of K: the same bits.
// LDS of the workgroup: words [0, NW) rows A has stored (pass * 64 + c), [NW, 2 NW) rows B has summed, [2 NW] quit; behind them
// two 2 KB slots per column (rows of alternating parity).
__device__ __forceinline__ void inverse_wave_paired(const FitLoopParams &P, int *ctl, int hi, int m1, int wv, int lane, double *lds) {
    const int N = P.N, nb = P.NP / 16;
    const int cl = lane & 15, rg = lane >> 4;
    const gdouble *Cg = as_global(uniform_ptr(const_cast<const double *>(P.C)));
    gdouble *Wg = as_global(uniform_ptr(P.W));
    double *xg = P.WdT;
    const int aug_tile = N / 16, aug_r = N - 16 * aug_tile;
    int *words = reinterpret_cast<int *>(lds);
    double *slots = lds + 64;  // (512 bytes of words)
    if (wv == 0 && lane < 2 * NW + 1) words[lane] = 0;
    __syncthreads();
    const int nA = (nb - m1 + hi - 1) / hi;  // waves of this workgroup with a column: w hi + m1 < nb
    int seen = 0;
    auto quit = [&]() { return __hip_atomic_load(words + 2 * NW, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0; };
    if (wv >= nA) {
        const int a = wv - nA, J = a * hi + m1;
        if (a >= nA || J + 4 >= nb) {  // nobody to help: wait for the end of the fit (the workgroup leaves together)
            while (wait_prog(ctl, 0x7fffffff, seen)) {
            }
            return;
        }
        v2f64 *slot = reinterpret_cast<v2f64 *>(slots + (size_t)a * 512);
        for (int seq = 1;; ++seq) {
            for (int r = J + 4; r < nb; ++r) {
                const int need = seq * 64 + r - 4;  // A has stored W_{r-4,J} (and seen the progress word of that row)
                int spins = 0;
                while (__hip_atomic_load(words + a, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < need) {
                    __builtin_amdgcn_s_sleep(2);
                    if ((++spins & 1023) == 0) {
                        if (quit()) return;
                        if (clu::ld(ctl + PROG) < 0) return;  // (the fit is over and A has gone without a word)
                    }
                }
                const v4f64 acc = chain_sum(Cg, Wg, r, J, r - 3 - J, nb, lane);
                v2f64 *q = slot + (size_t)(r & 1) * 128 + lane;
                q[0] = v2f64{acc[0], acc[1]};
                q[64] = v2f64{acc[2], acc[3]};
                __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the tile is in LDS
                if (lane == 0) __hip_atomic_store(words + NW + a, seq * 64 + r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
    // ---- A: the column's own wave ----
    const int J = wv * hi + m1;
    const bool hasB = nA + wv < NW && J + 4 < nb;
    const v2f64 *slot = reinterpret_cast<const v2f64 *>(slots + (size_t)wv * 512);
    Frag ident;  // B operand: the identity
#pragma unroll
    for (int q = 0; q < 4; ++q) ident.v[q] = (4 * q + rg == cl) ? 1.0 : 0.0;
    auto leave = [&]() {
        if (lane == 0) __hip_atomic_store(words + 2 * NW, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    for (int seq = 1;; ++seq) {
        double t2 = 0.0;
        v4f64 acc1 = {0.0, 0.0, 0.0, 0.0}, acc2 = acc1, wprev = acc1;
        for (int c = J; c < nb; ++c) {
            if (!wait_prog(ctl, seq * clu::kSeq + c + 1, seen)) {
                leave();
                return;
            }
            v4f64 xp, la;  // X_cc = W_cc; mirror tile (c, c + 1): L_{c+1,c}
            v4f64 l2a = {0.0, 0.0, 0.0, 0.0}, l2b = l2a;  // mirror tiles (c - 1, c + 2), (c, c + 2): L_{c+2,c-1}, L_{c+2,c}
            if (hasB && c + 2 < nb) {
                v2f64 t0, t1, t2r, t3, t4, t5, t6, t7;
                const unsigned o0 = (unsigned)((c * nb + c) * 2048) + (unsigned)lane * 16u, o1 = o0 + 2048u, o3 = o0 + 4096u;
                const unsigned o2 = (unsigned)(((c > J ? c - 1 : c) * nb + c + 2) * 2048) + (unsigned)lane * 16u;
                asm volatile("s_nop 4\n\t"
                             "global_load_dwordx4 %0, %8, %12 sc1\n\tglobal_load_dwordx4 %1, %8, %12 offset:1024 sc1\n\t"
                             "global_load_dwordx4 %2, %9, %12 sc1\n\tglobal_load_dwordx4 %3, %9, %12 offset:1024 sc1\n\t"
                             "global_load_dwordx4 %4, %10, %12 sc1\n\tglobal_load_dwordx4 %5, %10, %12 offset:1024 sc1\n\t"
                             "global_load_dwordx4 %6, %11, %12 sc1\n\tglobal_load_dwordx4 %7, %11, %12 offset:1024 sc1\n\t"
                             "s_waitcnt vmcnt(0)"
                             : "=&v"(t0), "=&v"(t1), "=&v"(t2r), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7)
                             : "v"(o0), "v"(o1), "v"(o2), "v"(o3), "s"(Cg)
                             : "memory");
                xp = v4f64{t0[0], t0[1], t1[0], t1[1]};
                la = v4f64{t2r[0], t2r[1], t3[0], t3[1]};
                l2a = v4f64{t4[0], t4[1], t5[0], t5[1]};
                l2b = v4f64{t6[0], t6[1], t7[0], t7[1]};
            } else {
                ld_pk_dev2(Cg, (unsigned)((c * nb + c) * 2048), Cg, (unsigned)((c * nb + (c + 1 < nb ? c + 1 : c)) * 2048), lane, xp, la);
            }
            v4f64 w;
            if (J == c) {
                w = xp;
            } else {
                // the A operand X_cc (element [cl][4 q + rg]) is the accumulator layout of X_cc^T: the packed tile used AS an A
                // operand is X_cc^T, times the identity (exact)
                v4f64 z = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int q = 0; q < 4; ++q) z = __builtin_amdgcn_mfma_f64_16x16x4f64(xp[q], ident.v[q], z, 0, 0, 0);
                Frag fw, fs;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    fw.v[q] = z[q];
                    fs.v[q] = acc1[q];
                }
                w = v4f64{0.0, 0.0, 0.0, 0.0};
                w = mfma4(fw, fs, w, true);
            }
            st_pk(Wg, (unsigned)((c * nb + J) * 2048), lane, w);
            if (c + 1 < nb) {  // the last product of row c + 1: the accumulator registers of W_cJ are its B fragments
                acc1 = acc2;
#pragma unroll
                for (int q = 0; q < 4; ++q) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(la[q], w[q], acc1, 0, 0, 0);
            }
            double ssq = 0.0;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (16 * c + rg + 4 * q < N) ssq = fma(w[q], w[q], ssq);
            ssq += __shfl_xor(ssq, 16);
            ssq += __shfl_xor(ssq, 32);
            t2 += ssq;  // (rows in order, from 0.0: the sum solve_posterior forms from the tiles' column sums)
            if (c == aug_tile) {  // row N of W: -m
                double mv = w[0];
                mv = (aug_r >> 2) == 1 ? w[1] : mv;
                mv = (aug_r >> 2) == 2 ? w[2] : mv;
                mv = (aug_r >> 2) == 3 ? w[3] : mv;
                if (rg == (aug_r & 3) && 16 * J + cl < N) xg[P.NP + 16 * J + cl] = -mv;
            }
            if (c + 2 < nb) {  // the products K = J .. c of row c + 2 (columns <= c of L are final)
                if (hasB) {
                    v4f64 part = {0.0, 0.0, 0.0, 0.0};
                    if (c - 2 >= J) {  // K = J .. c - 2: B's
                        const int need = seq * 64 + c + 2;
                        int spins = 0;
                        while (__hip_atomic_load(words + NW + wv, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < need) {
                            __builtin_amdgcn_s_sleep(1);
                            if (++spins > (1 << 22)) {  // (never observed.  B gone: this column is never reported done, the first
                                leave();                //  workgroup's wait for the columns runs out and the host repeats the fit
                                return;                 //  on one compute unit -- FIT_STATUS_CLUSTER -- rather than a wrong sum)
                            }
                        }
                        const v2f64 *q = slot + (size_t)((c + 2) & 1) * 128 + lane;
                        const v2f64 p0 = q[0], p1 = q[64];
                        part = v4f64{p0[0], p0[1], p1[0], p1[1]};
                    }
                    if (c - 1 >= J) {  // K = c - 1: W_{c-1,J} from last event's registers
#pragma unroll
                        for (int q = 0; q < 4; ++q) part = __builtin_amdgcn_mfma_f64_16x16x4f64(l2a[q], wprev[q], part, 0, 0, 0);
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) part = __builtin_amdgcn_mfma_f64_16x16x4f64(l2b[q], w[q], part, 0, 0, 0);  // K = c
                    acc2 = part;
                } else {
                    acc2 = chain_sum(Cg, Wg, c + 2, J, c - J + 1, nb, lane);
                }
            }
            wprev = w;
            if (hasB) {  // W_cJ is in the L2 (and this wave's L1): B may read it
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) __hip_atomic_store(words + wv, seq * 64 + c, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        if (rg == 0 && 16 * J + cl < N) xg[16 * J + cl] = t2;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the stores have reached the L2 before the count says so
        if (lane == 0) add(ctl + DONE, 1);
    }
}
}  // namespace clu

// ---- one posterior solve: C = A + diag(1/p) -> L -> W = L^-1 -> y = W b, m = W^T y, tr2 = colnorm2(W) --------------
// Storage: C strictly-upper blocks = L^T (the mirror tiles (K, I) = L_IK^T: what the inverse reads); the lower tiles of C
// hold the trailing matrix in progress, every tile TRANSPOSED (tile (I, J) holds T_IJ^T: the update swaps its operands, and
// the tiles of column k + 1 are then the B operands of D = L_{k+1,k+1}^-1 T^T as they stand -- no transposition on the
// way into the panel); W lower = L^-1.  A, C and W are stored as PACKED tiles (tile_chol.h): two contiguous 1 KB accesses.
// CLM: 0 one workgroup does everything; 1, 2: cluster mode (the rows of the inverse on the helper workgroups, clu::), with
// every wave but the chain's on the trailing update (1) or with the two waves that share the chain's SIMD sitting out (2);
// seq: number of this solve within the fit (the helpers count the passes the same way)
// CLM = 4: DEFERRED trailing update (round 5, one workgroup, N <= 319).  With the device full a pass is bound by the bytes it moves
// beyond the L2 (profiles/r05_pmc_fit_loop_loaded.json: every store leaves the L2, 85 % of the L1's read misses too; 2.4 MB read
// + 2.7 MB written per pass, 4.9 TB/s chip-wide), and 2.0 of the 2.7 MB written are the trailing tiles, loaded, updated by ONE
// panel and stored again at every step.  Here a tile right of column k + 2 is touched at every OTHER step -- the steps of its
// own parity (I + J + k even) -- and takes the two panels it then misses, k - 1 and k, in that order, from THREE panels in LDS
// (k - 1, k and the k + 1 being formed); the tiles of column k + 2 (next step's column tiles) are always brought up to date.
// Same products, same operands, same order per tile: the same bits.  Half the loads and stores of the trailing update; both
// parities work at every step, so the steps stay balanced.  The band factors and scan tables move to the W buffer (global
// memory, as the wide instantiations have them) to make room for the third panel.
template <int WIDE, int CLM>
__device__ __forceinline__ bool solve_posterior(const FitLoopParams &P, const Smem &S, int seq) {
    constexpr bool CL = CLM == 1 || CLM == 2;
    constexpr bool DF = CLM == 4 || CLM == 5;
    constexpr bool PAIR = CLM == 5;  // ... and the rows of the inverse in pairs (inverse_pair)
    static_assert(!(DF && WIDE), "the deferred update keeps three panels in LDS: N <= 319");
    constexpr int NWKc = CLM == 2 ? NW - NW / 4 : NWK;  // trailing-update workers
    const int N = P.N, NP = P.NP, nb = P.NP / 16, ld = P.NP;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (scalar: loop control on the SALU)
    const int cl = lane & 15, rg = lane >> 4;
    // W = L^-1 lives in the LOWER tiles of C: tile (k, J), J < k, of the trailing matrix is dead once column J has been
    // processed (its factor is the mirror tile (J, k)), the diagonal tile once wave 0 has it in registers; row k of W is
    // written at step k > J.  One matrix less in the working set of a pass (1.5 -> 1.1 MB: three loops per 4 MB L2).
    double *C = P.C, *W = P.C;
#ifdef FIT_LOOP_TIMING
    long long t_last = clock64();
#endif

    // 1/p (padding rows: 1) -- the diagonal of C = A + diag(1/p) is added when a tile is first read from A
    for (int i = tid; i < NP; i += KT) {
        const double rp = i < N ? 1.0 / S.p[i] : 1.0;  // row N (b) and padding: 1
        S.y[i] = rp;
        if constexpr (CL) P.WdT[2 * NP + i] = rp;  // (the helpers' diagonal tiles; in the L2 before the first word of the pass)
    }
    if (tid == 0) {
        S.flag[0] = 0;  // not positive definite
        S.flag[1] = S.flag[2] = 0;  // column counters of the inverse rows (even / odd steps)
        S.flag[3] = 0;  // index of the last diagonal tile whose inverse is in LDS
    }
    __syncthreads();
    const double *pinv = S.y;
    TSTAMP(0);

    // (2) right-looking blocked Cholesky, 16-wide panels; step 0 reads A, later steps read C.
    //   * look-ahead: wave 0 updates tile (k+1,k+1) first, factors it AND inverts it (dl, dli) while the other waves
    //     do the trailing update of step k.  fp64 VALU and fp64 MFMA share the DP units of a SIMD, so the waves
    //     that share wave 0's SIMD (4, 8, 12) sit the trailing update out;
    //   * panel: L_Ik^T = L_kk^-1 C_Ik^T on MFMAs (C_Ik^T is the mirror block (k, I)), one tile per wave.
    // Row N of the padded system carries b:  C[N, 0:N] = b^T with the pivot of that row forced to 1, so the factor's
    // row N is y^T = (L^-1 b)^T and row N of W = L^-1 is -(W^T y)^T = -m^T: both matvecs come for free.
    const int aug_tile = N / 16, aug_c = N - 16 * aug_tile;
    auto cs_ptr = [&](int I, int J) { return P.cs + ((size_t)I * nb + J) * 16; };
    auto rows_valid = [&](int I) { return min(16, max(0, N - 16 * I)); };
    if (wave == 0) {
        v4f64 t0 = ld_pk(as_global(P.A), 0u, lane), x0;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (rg + 4 * r == cl) t0[r] += pinv[cl];
        const bool ok = factor_invert_tile(t0, x0, S.dli, lane, aug_tile == 0 ? aug_c : -1);  // L_00^-1 -> dli[0]
        if (!ok && lane == 0) *S.flag = 1;
        if (!CL) store_col_ssq(x0, cs_ptr(0, 0), rows_valid(0), lane);
        st_pk(as_global(W), 0u, lane, x0);  // W_00
    }
    __syncthreads();
    TSTAMP(1);
    // Panel 0 from memory: D = L_00^-1 (C_I0)^T for I > 0; D -> block (0, I), D^T -> block (I, 0), D -> LDS panel 0.  The
    // panels of the later steps are produced INSIDE the trailing update of the step before (below): one barrier per step.
    {
        Frag fa;
#pragma unroll
        for (int q = 0; q < 4; ++q) fa.v[q] = S.dli[cl * PS + 4 * q + rg];
        constexpr int kPanelMax = WIDE == 2 ? 6 : (WIDE ? 4 : 3);  // ceil((NP / 16 - 1) / NW): 4 for NP <= 640, 6 for NP <= 1024
        Frag fb[kPanelMax];
#pragma unroll
        for (int u = 0; u < kPanelMax; ++u) {
            const int I = 1 + wave + u * NW;
            if (I < nb) {
                const v4f64 t = ld_pk(as_global(P.A), (unsigned)(I * 2048), lane);  // tile (0, I) = (A_I0)^T: A is symmetric
#pragma unroll
                for (int q = 0; q < 4; ++q) fb[u].v[q] = t[q];
            }
        }
#pragma unroll
        for (int u = 0; u < kPanelMax; ++u) {
            const int I = 1 + wave + u * NW;
            if (I < nb) {
                v4f64 d = {0.0, 0.0, 0.0, 0.0};
                d = mfma4(fa, fb[u], d, false);
                st_pk(as_global(C), (unsigned)(I * 2048), lane, d);  // tile (0, I) = L_I0^T (what the inverse reads)
                double *pr = S.pan + (size_t)((I - 1) * 16 + cl) * PS + rg;
#pragma unroll
                for (int r = 0; r < 4; ++r) pr[4 * r] = d[r];
            }
        }
    }
    int *const ctl = CL ? clu::ctl_of(P) : nullptr;
    if constexpr (CL) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the barrier waits for LDS only)
    __syncthreads();
    if constexpr (CL)
        if (tid == 0) clu::st(ctl + clu::PROG, seq * clu::kSeq + 1);  // column 0 of L and X_00 are final
    TSTAMP(2);
    const gdouble *C_inv = as_global(uniform_ptr(C));
    gdouble *W_inv = as_global(uniform_ptr(W));
    gdouble *C_u = as_global(uniform_ptr(C));
    const int band = CL ? clu::band_of(P) : nb;
    const gdouble *H_u = as_global(uniform_ptr(const_cast<const double *>(P.W)));   // cluster mode: trailing tiles from the helpers
    const gdouble *HD_u = as_global(uniform_ptr(const_cast<const double *>(P.cs)));  // (the diagonal ones)
    for (int k = 0; k < nb; ++k) {
        if (*S.flag) return false;
        const double *src = (k == 0) ? P.A : C;
        const gdouble *src_u = as_global(uniform_ptr(src));
        const int m = nb - k - 1;
        const int cntA = __builtin_amdgcn_readfirstlane(m * (m - 1) / 2);  // tiles with k + 1 < J <= I
        const int ncol = __builtin_amdgcn_readfirstlane(max(m - 1, 0));     // tiles (I, k + 1), I > k + 1
        // cluster mode: only the `band` block columns right of column k + 1 are updated here (column by column: cntB tiles);
        // the tiles of the LAST of them, from index cntH on, come back from the helper waves at this step (clu::helper_wave)
        int cntB = cntA, cntH = cntA;
        if constexpr (CL) {
            if (band < nb) {
                const int nbc = min(band, m - 1);
                cntB = 0;
                for (int jj = 1; jj <= nbc; ++jj) {
                    if (jj == band) cntH = cntB;
                    cntB += m - jj;
                }
                if (nbc < band || k == 0) cntH = cntB;
                cntB = __builtin_amdgcn_readfirstlane(cntB);
                cntH = __builtin_amdgcn_readfirstlane(cntH);
            }
        }
        double *pan_cur = S.pan + (WIDE ? (size_t)0 : (size_t)(DF ? k % 3 : (k & 1)) * NP * PS);
        double *pan_nxt = S.pan + (WIDE ? (size_t)0 : (size_t)(DF ? (k + 1) % 3 : ((k + 1) & 1)) * NP * PS);
        // deferred mode: panel k - 1 is still in LDS; the tiles of this step's parity come from one of two filtered tables
        const char *pan_p = reinterpret_cast<const char *>(S.pan + (DF ? (size_t)((k + 2) % 3) * NP * PS : (size_t)0));
        const int dpar = k & 1;
        if constexpr (DF) cntB = __builtin_amdgcn_readfirstlane(m > 0 ? S.dcnt[dpar * 24 + m - 1] : 0);
        int *ctr_cur = S.flag + 1 + (k & 1);
        if (tid == 0) S.flag[1 + ((k + 1) & 1)] = 0;  // column counter of the NEXT step's inverse row (nobody reads it now)
        const unsigned base_pk = (unsigned)((k + 1) * (nb + 1) * 2048);   // tile (k+1, k+1), packed
        const unsigned lane_p = (unsigned)((cl * PS + rg) * 8);
        const char *pan_b = reinterpret_cast<const char *>(pan_cur);
        // columns of row k of the inverse, pulled from an LDS counter, longest chain (J = 0) first (which wave computes a tile
        // does not change its bits); row k of L is final since the panel of step k - 1
        // L^-1 of the diagonal tiles in LDS: two buffers in turn -- three in deferred mode, where the rows k - 1 and k of the
        // inverse are formed together at step k (inverse_pair) while the chain writes that of tile k + 1
        auto dli_of = [&](int t) { return S.dli + (DF ? t % 3 : (t & 1)) * 16 * PS; };
        const double *dli_k = dli_of(k);  // L_kk^-1 = W_kk, left there by the chain of step k - 1
        auto inverse_one = [&]() {
            int J = 0;
            if (lane == 0) J = atomicAdd(ctr_cur, 1);
            J = __builtin_amdgcn_readfirstlane(J);
            if constexpr (PAIR) {
                // rows k - 1 and k together for the columns with J + k even (inverse_pair); the other columns wait for step
                // k + 1 -- or, at the last step, take their last row alone
                if (k < nb - 1) {
                    J = 2 * J + (k & 1);
                    if (J > k - 2) return false;
                } else {
                    if (J >= k) return false;
                    if ((J + k) & 1) {
                        inverse_tile(C_inv, dli_k, W_inv, cs_ptr(k, J), k, J, N, nb, lane);
                        return true;
                    }
                }
                inverse_pair(C_inv, dli_of(k - 1), dli_k, W_inv, cs_ptr(k - 1, J), cs_ptr(k, J), k, J, N, nb, lane);
                return true;
            }
            if (J >= k) return false;
            inverse_tile(C_inv, dli_k, W_inv, cs_ptr(k, J), k, J, N, nb, lane);
            return true;
        };
        auto inverse_columns = [&]() {
            if (k < 1) return;
            while (inverse_one()) {
            }
        };
        TRACE(0);
        if (wave == kChain) {
            if (m > 0) {  // look-ahead: tile (k+1, k+1) updated, factored and inverted while the other waves update the rest
#ifdef FIT_LOOP_TIMING
                long long f_last = clock64();
#endif
                v4f64 a, xi;
                a = ld_pk(src_u, base_pk, lane);
                if (k == 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (rg + 4 * r == cl) a[r] += pinv[16 + cl];
                }
                const double *pa1 = reinterpret_cast<const double *>(pan_b + lane_p);
#pragma unroll
                for (int s = 0; s < 4; ++s) a = __builtin_amdgcn_mfma_f64_16x16x4f64(-pa1[4 * s], pa1[4 * s], a, 0, 0, 0);
                FSTAMP(8);
                // factor and invert in the accumulator layout (DPP row broadcasts, no LDS round trip, no transposition)
                const bool ok = factor_invert_tile(a, xi, dli_of(k + 1), lane, aug_tile == k + 1 ? aug_c : -1);
                if (!ok && lane == 0) *S.flag = 1;
                FSTAMP(9);
                if (!CL) store_col_ssq(xi, cs_ptr(k + 1, k + 1), rows_valid(k + 1), lane);
                st_pk(as_global(uniform_ptr(W)), base_pk, lane, xi);  // W_{k+1,k+1}
                // L_{k+1,k+1}^-1 is in LDS: the waves holding tiles of column k + 1 may now turn them into panel k + 1
                __hip_atomic_store(&S.flag[3], k + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                FSTAMP(10);
                TRACE(5);
            }
            if constexpr (WIDE) __syncthreads();  // (everybody has read panel k: the column tiles may overwrite it)
            if constexpr (!CL) inverse_columns();
            TRACE(4);
        } else if (CLM == 2 && (wave & 3) == (kChain & 3)) {
            // cluster mode 2: the waves on the chain's SIMD leave its double-precision units to the factor-and-invert chain
            if constexpr (WIDE) __syncthreads();
        } else {
            const int widx = CLM == 2 ? wave - (wave >> 2) - ((wave & 3) > (kChain & 3) ? 1 : 0) : (wave < kChain ? wave : wave - 1);  // 0..NWKc-1
#ifdef FIT_LOOP_TIMING
            long long w_last = clock64();
#endif
            // ---- trailing update of the tiles right of column k + 1 ----
            // Per tile: 4 MFMAs (256 cycles of the matrix pipe) against, at first, ~90 other instructions, most of them 64-bit
            // address arithmetic.  Every address is now a uniform base + a 32-bit byte offset (one v_add per row), the offsets
            // of tile e relative to block (k+1, k+1) come from a table built once per launch (S.rec: the tiles of step k are
            // the first cntA entries of ONE row-wise enumeration), and THREE named register sets rotate through an unrolled
            // trip: the loads of the tile after next are issued before the stores of the current one (vmcnt counts loads and
            // stores in order -- a load issued behind a store would wait for the store's acknowledgement) without a register
            // copy.  (2 x 2 groups of tiles with four interleaved MFMA chains were measured too: no faster.)
            bool hready = false;
            auto rec_at = [&](int e) -> uint4 {
                if constexpr (DF) return S.rec[dpar * 128 + e];
                if constexpr (CL) {
                    if (band < nb) {  // column by column: column j of the band holds the rows i = j .. m - 1
                        int j = 1, i = e;
                        while (i >= m - j) {
                            i -= m - j;
                            ++j;
                        }
                        i += j;
                        return S.rec[(i - 1) * i / 2 + (j - 1)];
                    }
                }
                if constexpr (WIDE == 2) {  // no table in LDS: tile e = (i - 1) i / 2 + (j - 1), 1 <= j <= i (e is wave-uniform)
                    int i = (int)((sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
                    while ((i + 1) * (i + 2) / 2 <= e) ++i;
                    while (i * (i + 1) / 2 > e) --i;
                    const int j = e - i * (i + 1) / 2 + 1;
                    ++i;
                    return make_uint4((unsigned)((j * nb + i) * 2048) | (i == j ? 2u : 0u), (unsigned)(j * 16 * PS * 8),
                                      (unsigned)(i * 16 * PS * 8), (unsigned)((i * nb + j) * 2048) | (unsigned)i);
                } else {
                    return S.rec[e];
                }
            };
            auto ldt = [&](const uint4 &t, int e) {  // (step 0 reads A: the transpose of tile (I, J) is its tile (J, I))
                if constexpr (CL) {
                    if (e >= cntH) {  // a tile of the column that enters the band: updates 0 .. k - 1 were the helpers'
                        if (!hready) {
                            const int J = k + 1 + band, need = seq * (nb - J);
                            const long long t0 = wall_clock64();
                            while (__builtin_amdgcn_readfirstlane(clu::ld(ctl + clu::HCOL + J)) < need) {
                                __builtin_amdgcn_s_sleep(1);
                                if (wall_clock64() - t0 > 1000000 * clu::kTicksPerUs) {
                                    if (lane == 0) S.flag[0] = 2;
                                    break;
                                }
                            }
                            hready = true;
                        }
                        if (e == cntH) return clu::ld_pk_dev(HD_u, (unsigned)((k + 1 + band) * 2048), lane);  // (its first tile: the diagonal one)
                        return clu::ld_pk_dev(H_u, base_pk + (t.x & ~2047u), lane);  // tile (I, J) waits at the mirror position (J, I)
                    }
                }
                if constexpr (DF) {
                    // first touch of a tile: step 0 for its own parity and for column 2, step 1 for the other parity (bit 2 of x:
                    // a tile of column k + 2 that only takes panel k)
                    const bool fromA = k == 0 || (k == 1 && !(t.x & 4u));
                    return ld_pk(as_global(uniform_ptr(fromA ? P.A : const_cast<const double *>(C))),
                                 base_pk + ((fromA ? t.x : t.w) & ~2047u), lane);
                }
                return ld_pk(src_u, base_pk + ((k == 0 ? t.x : t.w) & ~2047u), lane);
            };
            auto upd = [&](unsigned pa, unsigned pb, v4f64 a) {
                const double *pa1 = reinterpret_cast<const double *>(pan_b + pa + lane_p);
                const double *pb1 = reinterpret_cast<const double *>(pan_b + pb + lane_p);
#pragma unroll
                for (int s = 0; s < 4; ++s) a = __builtin_amdgcn_mfma_f64_16x16x4f64(-pa1[4 * s], pb1[4 * s], a, 0, 0, 0);
                return a;
            };
            auto fin = [&](const uint4 &t, v4f64 a) {
                if (k == 0 && (t.x & 2u)) {  // first touch: add diag(1/p) on diagonal tiles
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (rg + 4 * r == cl) a[r] += pinv[16 * (1 + (t.w & 127u)) + cl];  // (w: packed offset | i)
                }
                if constexpr (DF) {
                    if (k >= 1 && !(t.x & 4u)) {  // the panel this tile sat out: k - 1 (its rows sit one block further down)
                        const double *pa0 = reinterpret_cast<const double *>(pan_p + t.y + 16 * PS * 8 + lane_p);
                        const double *pb0 = reinterpret_cast<const double *>(pan_p + t.z + 16 * PS * 8 + lane_p);
#pragma unroll
                        for (int s = 0; s < 4; ++s) a = __builtin_amdgcn_mfma_f64_16x16x4f64(-pa0[4 * s], pb0[4 * s], a, 0, 0, 0);
                    }
                }
                a = upd(t.y, t.z, a);
                st_pk(C_u, base_pk + (t.w & ~2047u), lane, a);
#ifdef K2_PROBE_STORE2
                // sensitivity probe (make probe; never shipped): every trailing tile stored a second time, into the unused W
                // buffer -- what 2 MB more of stores per pass cost a loaded device (all stores leave the L2)
                if constexpr (!CL && !WIDE && !DF) st_pk(as_global(uniform_ptr(P.W)), base_pk + (t.w & ~2047u), lane, a);
#endif
            };
            int e = widx;  // every NWKc-th tile of the enumeration
            const int cnt = (CL || DF) ? cntB : cntA;
            if (e < cnt) {
                uint4 ta = rec_at(e), tb = ta, tc = ta;
                v4f64 a = ldt(ta, e), b = a, c = a;
                if (e + NWKc < cnt) {
                    tb = rec_at(e + NWKc);
                    b = ldt(tb, e + NWKc);
                }
                for (;;) {
                    // sets in flight: a (current), b (next); c is free
                    if (e + 2 * NWKc < cnt) {
                        tc = rec_at(e + 2 * NWKc);
                        c = ldt(tc, e + 2 * NWKc);
                    }
                    fin(ta, a);
                    if (e + NWKc >= cnt) break;
                    if (e + 3 * NWKc < cnt) {
                        ta = rec_at(e + 3 * NWKc);
                        a = ldt(ta, e + 3 * NWKc);
                    }
                    fin(tb, b);
                    if (e + 2 * NWKc >= cnt) break;
                    if (e + 4 * NWKc < cnt) {
                        tb = rec_at(e + 4 * NWKc);
                        b = ldt(tb, e + 4 * NWKc);
                    }
                    fin(tc, c);
                    if (e + 3 * NWKc >= cnt) break;
                    e += 3 * NWKc;
                }
            }
            WSTAMP(11);
            TRACE(1);
            // ---- column k + 1: update, then the panel of step k + 1 straight from the registers ----
            // (the round-robin deal of the tiles goes on where the enumeration above stopped)
            int cfirst = widx - cnt % NWKc;
            if (cfirst < 0) cfirst += NWKc;
            constexpr int kColMax = WIDE == 2 ? 6 : (CLM == 2 ? 5 : 4);  // (WIDE: ncol <= 38, XWIDE: <= 62 <= kColMax x NWKc)
            v4f64 dcol[WIDE ? kColMax : 1];
            if (cfirst < ncol) {
                // L_{k+1,k+1}^-1 comes from wave 0's chain (~5 us into the step): inverse columns fill the wait
                if (k >= 1 && !CL) {
                    int spins = 0;
                    while (__hip_atomic_load(&S.flag[3], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < k + 1) {
                        if (!inverse_one()) {
                            __builtin_amdgcn_s_sleep(2);
                            if (++spins > (1 << 22)) {  // (never observed; a stuck flag must not hang the device)
                                if (lane == 0) *S.flag = 1;
                                break;
                            }
                        }
                    }
                } else {
                    int spins = 0;
                    while (__hip_atomic_load(&S.flag[3], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < k + 1) {
                        __builtin_amdgcn_s_sleep(2);
                        if (++spins > (1 << 22)) {
                            if (lane == 0) *S.flag = 1;
                            break;
                        }
                    }
                }
                TRACE(2);
                Frag fx;  // L_{k+1,k+1}^-1: as A operand X, as B operand X^T
#pragma unroll
                for (int q = 0; q < 4; ++q) fx.v[q] = dli_of(k + 1)[cl * PS + 4 * q + rg];
                if constexpr (!WIDE) {
                    for (int c = cfirst; c < ncol; c += NWKc) {
                        const int i = c + 1;  // block row I = k + 1 + i
                        v4f64 t = ld_pk(src_u, base_pk + (unsigned)(k == 0 ? i * 2048 : i * nb * 2048), lane);
                        t = upd(0u, (unsigned)(i * 16 * PS * 8), t);  // T^T = C_{I,k+1}^T - L_{k+1,k} L_Ik^T: the tile is kept transposed
                        Frag ft;  // as B operand: T^T (the accumulator registers of a matrix are its B fragments)
#pragma unroll
                        for (int q = 0; q < 4; ++q) ft.v[q] = t[q];
                        const v4f64 z4 = {0.0, 0.0, 0.0, 0.0};
                        const v4f64 d = mfma4(fx, ft, z4, false);  // D = X T^T = L_{I,k+1}^T  (what the panel from memory computes)
                        // tile (k+1, I): the block the inverse reads; L_{I,k+1} itself (block (I, k+1)) has no reader
                        st_pk(C_u, base_pk + (unsigned)(i * 2048), lane, d);
                        double *pr = pan_nxt + (size_t)((i - 1) * 16 + cl) * PS + rg;
#pragma unroll
                        for (int r = 0; r < 4; ++r) pr[4 * r] = d[r];
                    }
                } else {
                    // one panel: the tiles stay in registers until everybody has read panel k (the barrier below)
#pragma unroll
                    for (int u = 0; u < kColMax; ++u) {
                        const int c = cfirst + u * NWKc;
                        if (c < ncol) {
                            const int i = c + 1;
                            v4f64 t = ld_pk(src_u, base_pk + (unsigned)(k == 0 ? i * 2048 : i * nb * 2048), lane);
                            t = upd(0u, (unsigned)(i * 16 * PS * 8), t);
                            Frag ft;
#pragma unroll
                            for (int q = 0; q < 4; ++q) ft.v[q] = t[q];
                            const v4f64 z4 = {0.0, 0.0, 0.0, 0.0};
                            dcol[u] = mfma4(fx, ft, z4, false);
                        }
                    }
                }
            }
            if constexpr (WIDE) {
                __syncthreads();
#pragma unroll
                for (int u = 0; u < kColMax; ++u) {
                    const int c = cfirst + u * NWKc;
                    if (c < ncol) {
                        const int i = c + 1;
                        st_pk(C_u, base_pk + (unsigned)(i * 2048), lane, dcol[u]);
                        double *pr = pan_nxt + (size_t)((i - 1) * 16 + cl) * PS + rg;
#pragma unroll
                        for (int r = 0; r < 4; ++r) pr[4 * r] = dcol[u][r];
                    }
                }
            }
            TRACE(3);
            if constexpr (!CL) inverse_columns();
            TRACE(4);
            WSTAMP(12);
        }
        if constexpr (CL) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this step's tiles are in the L2 the helpers read
        __syncthreads();
        if constexpr (CL)
            if (tid == 0 && m > 0) clu::st(ctl + clu::PROG, seq * clu::kSeq + k + 2);  // columns <= k + 1 of L and X_{k+1,k+1} are final
        TSTAMP(3);
    }
    if (*S.flag) return false;
    TSTAMP(4);

    if constexpr (CL) {
        // Tr2 and m = Y mu come from the helper waves that own the block columns of W (exchange area: the fit's WdT buffer)
        const int T = clu::inv_helpers(P) * NW, nsig = T < nb ? T : nb;
        if (tid == 0) {
            const long long t0 = wall_clock64();
            while (clu::ld(ctl + clu::DONE) < seq * nsig) {
                __builtin_amdgcn_s_sleep(1);
                if (wall_clock64() - t0 > 1000000 * clu::kTicksPerUs) {  // 1 s: the helpers are gone
                    S.flag[0] = 2;
                    break;
                }
            }
        }
        __syncthreads();
        if (*S.flag) return false;
        const double *xg = P.WdT;
        for (int i = tid; i < N; i += KT) {  // (device-scope loads: the helpers' stores, not this compute unit's L1)
            S.tr2[i] = clu::ld_dev(xg + i);
            S.m[i] = clu::ld_dev(xg + NP + i);
        }
        __syncthreads();
        TSTAMP(7);
        return true;
    }

    // (5) m = -(row N of W),  tr2_i = sum over the block column of the tile column sums (fixed order)
    for (int i = tid; i < N; i += KT) {
        const int J = i >> 4, c = i & 15;
        double t2 = 0.0;
        for (int I = J; I < nb; ++I) t2 += cs_ptr(I, J)[c];
        S.tr2[i] = t2;
        {   // row N of W, packed: tile (aug_tile, J), element (N - 16 aug_tile, c)
            const int rr = N - 16 * aug_tile, q = rr >> 2, ln = (rr & 3) * 16 + c;
            S.m[i] = -W[((size_t)aug_tile * nb + J) * 256 + (q >> 1) * 128 + ln * 2 + (q & 1)];
        }
    }
    __syncthreads();
    TSTAMP(7);
    return true;
}

// ---- the posterior solve of the first workgroup of a cluster (N <= 335: two panels in LDS) -----------------------------------
// Without the rows of the inverse and with the trailing tiles right of the band on the helpers (clu::), a step of
// solve_posterior is the chain -- factor-and-invert of the diagonal tile, ~2.5 us -- FOLLOWED by the column tiles, which need
// its result, and the barrier: ~4.5 us, with the worker waves mostly waiting.  Here the chain wave runs ahead: after X_{k+1} it
// forms the one column tile its next diagonal tile needs, L_{k+2,k+1}, itself, updates tile (k+2, k+2) with it straight from
// the accumulator registers (a tile D = X T^T in the accumulator layout IS its own operand fragments) and goes on factoring,
// while the workers turn the rest of column k + 1 into panel k + 1.  What the chain wave needs from the workers' step k -- the
// panel and the band tiles they stored -- it needs only AFTER X_{k+2}; so it joins the barrier that ends their step k there.
// Every wave still executes one s_barrier per step; a step is the chain alone, ~3 us.  The products, their operands and their
// order are those of solve_posterior: the same bits.
// Progress is published to the helpers without waiting for that barrier: every wave counts itself in (LDS) when its stores of
// the step have reached the L2, the last one raises the word.
template <int CLM>
__device__ __forceinline__ bool solve_posterior_cluster(const FitLoopParams &P, const Smem &S, int seq) {
    constexpr int NWKc = CLM == 2 ? NW - NW / 4 : NW - 1;  // trailing-update workers
    const int N = P.N, NP = P.NP, nb = P.NP / 16, ld = P.NP;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cl = lane & 15, rg = lane >> 4;
    double *C = P.C;
    int *const ctl = clu::ctl_of(P);
#ifdef FIT_LOOP_TIMING
    long long t_last = clock64();
#endif
    for (int i = tid; i < NP; i += KT) {
        const double rp = i < N ? 1.0 / S.p[i] : 1.0;  // row N (b) and padding: 1
        S.y[i] = rp;
        P.WdT[2 * NP + i] = rp;  // (the helpers' diagonal tiles; in the L2 before the first word of the pass)
    }
    if (tid == 0) {
        S.flag[0] = 0;  // 1: not positive definite, 2: the helpers are gone
        S.flag[3] = 0;  // index of the last diagonal tile whose inverse is in LDS
        S.flag[4] = 0;  // waves whose stores of the steps so far are in the L2
    }
    __syncthreads();
    const double *pinv = S.y;
    TSTAMP(0);
    const int aug_tile = N / 16, aug_c = N - 16 * aug_tile;
    auto rows_valid = [&](int I) { return min(16, max(0, N - 16 * I)); };
    if (wave == 0) {
        v4f64 t0 = ld_pk(as_global(P.A), 0u, lane), x0;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (rg + 4 * r == cl) t0[r] += pinv[cl];
        const bool ok = factor_invert_tile(t0, x0, S.dli, lane, aug_tile == 0 ? aug_c : -1);  // L_00^-1 -> dli[0]
        if (!ok && lane == 0) *S.flag = 1;
        st_pk(as_global(C), 0u, lane, x0);  // X_00 (the helpers' W_00)
    }
    __syncthreads();
    TSTAMP(1);
    {   // panel 0 from memory: D = L_00^-1 (C_I0)^T for I > 0; D -> block (0, I), D -> LDS panel 0
        Frag fa;
#pragma unroll
        for (int q = 0; q < 4; ++q) fa.v[q] = S.dli[cl * PS + 4 * q + rg];
        constexpr int kPanelMax = 3;  // ceil((NP / 16 - 1) / NW)
        Frag fb[kPanelMax];
#pragma unroll
        for (int u = 0; u < kPanelMax; ++u) {
            const int I = 1 + wave + u * NW;
            if (I < nb) {
                const v4f64 t = ld_pk(as_global(P.A), (unsigned)(I * 2048), lane);  // tile (0, I) = (A_I0)^T: A is symmetric
#pragma unroll
                for (int q = 0; q < 4; ++q) fb[u].v[q] = t[q];
            }
        }
#pragma unroll
        for (int u = 0; u < kPanelMax; ++u) {
            const int I = 1 + wave + u * NW;
            if (I < nb) {
                v4f64 d = {0.0, 0.0, 0.0, 0.0};
                d = mfma4(fa, fb[u], d, false);
                st_pk(as_global(C), (unsigned)(I * 2048), lane, d);  // tile (0, I) = L_I0^T
                double *pr = S.pan + (size_t)((I - 1) * 16 + cl) * PS + rg;
#pragma unroll
                for (int r = 0; r < 4; ++r) pr[4 * r] = d[r];
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the barrier waits for LDS only)
    __syncthreads();
    if (tid == 0) clu::st(ctl + clu::PROG, seq * clu::kSeq + 1);  // column 0 of L and X_00 are final
    TSTAMP(2);
    gdouble *C_u = as_global(uniform_ptr(C));
    const gdouble *A_u = as_global(uniform_ptr(P.A));
    const int band = clu::band_of(P);
    const gdouble *H_u = as_global(uniform_ptr(const_cast<const double *>(P.W)));   // trailing tiles from the helpers
    const gdouble *HD_u = as_global(uniform_ptr(const_cast<const double *>(P.cs)));  // (the diagonal ones)
    const unsigned lane_p = (unsigned)((cl * PS + rg) * 8);
    // every wave counts itself in when its stores of step k have reached the L2; the last one publishes "columns <= k + 1 of L
    // and X_{k+1,k+1} are final" (all arrivals of step k precede the barrier that ends it, all of step k + 1 follow it)
    auto arrive = [&](int k) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) {
            const int old = atomicAdd(&S.flag[4], 1);
            if (old + 1 == (k + 1) * NW) clu::st(ctl + clu::PROG, seq * clu::kSeq + k + 2);
        }
    };
    v4f64 dg = {0.0, 0.0, 0.0, 0.0};  // the chain wave's next diagonal tile, all updates applied
    if (wave == kChain && nb > 1) {
        dg = ld_pk(A_u, (unsigned)((nb + 1) * 2048), lane);
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (rg + 4 * r == cl) dg[r] += pinv[16 + cl];
        const double *pa1 = reinterpret_cast<const double *>(reinterpret_cast<const char *>(S.pan) + lane_p);
#pragma unroll
        for (int s = 0; s < 4; ++s) dg = __builtin_amdgcn_mfma_f64_16x16x4f64(-pa1[4 * s], pa1[4 * s], dg, 0, 0, 0);
    }
    // The step loop, once per ROLE: the chain wave's and the workers' bodies share nothing but the barriers (the same number in
    // both), and in one loop the registers a worker keeps from step to step (rA, rB) were live across the chain wave's tile
    // routine too -- the kernel spilled 828 bytes per lane and every form ran at less than half speed.
    if (wave == kChain) {
    for (int k = 0; k + 1 < nb; ++k) {
        const gdouble *src_u = k == 0 ? A_u : as_global(uniform_ptr(const_cast<const double *>(C)));
        const int m = nb - k - 1;
        double *pan_cur = S.pan + (size_t)(k & 1) * NP * PS;
        double *pan_nxt = S.pan + (size_t)((k + 1) & 1) * NP * PS;
        const unsigned base_pk = (unsigned)((k + 1) * (nb + 1) * 2048);  // tile (k+1, k+1), packed
        const char *pan_b = reinterpret_cast<const char *>(pan_cur);
        auto upd = [&](unsigned pa, unsigned pb, v4f64 a) {
            const double *pa1 = reinterpret_cast<const double *>(pan_b + pa + lane_p);
            const double *pb1 = reinterpret_cast<const double *>(pan_b + pb + lane_p);
#pragma unroll
            for (int s = 0; s < 4; ++s) a = __builtin_amdgcn_mfma_f64_16x16x4f64(-pa1[4 * s], pb1[4 * s], a, 0, 0, 0);
            return a;
        };
        TRACE(0);
        {
#ifdef FIT_LOOP_TIMING
            long long f_last = clock64();
#endif
            v4f64 xi;
            const bool ok = factor_invert_tile(dg, xi, S.dli + ((k + 1) & 1) * 16 * PS, lane, aug_tile == k + 1 ? aug_c : -1);
            if (!ok && lane == 0) *S.flag = 1;
            FSTAMP(9);
            st_pk(C_u, base_pk, lane, xi);  // X_{k+1,k+1} (the helpers' W_{k+1,k+1})
            __hip_atomic_store(&S.flag[3], k + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            FSTAMP(10);
            TRACE(5);
            if (k >= 1) __syncthreads();  // the barrier that ends the workers' step k - 1: panel k and their band tiles are complete
            if (m >= 2) {
                // column tile (k+2, k+1): T^T = C^T - L_{k+1,k} L_{k+2,k}^T, D = X T^T = L_{k+2,k+1}^T -- what a worker did
                v4f64 t, nd;  // tiles (k+2, k+1) and (k+2, k+2), updates < k applied
                if (k == 0 || !S.hand) {
                    t = ld_pk(src_u, base_pk + (unsigned)(k == 0 ? 2048 : nb * 2048), lane);
                    nd = ld_pk(src_u, base_pk + (unsigned)((nb + 1) * 2048), lane);
                } else {  // left in LDS by the workers that updated them at step k - 1
                    const v2f64 *hp = reinterpret_cast<const v2f64 *>(S.hand + (((k - 1) & 1) * 2) * 256) + lane;
                    const v2f64 a0 = hp[0], a1 = hp[64], b0 = hp[128], b1 = hp[192];
                    t = v4f64{a0[0], a0[1], a1[0], a1[1]};
                    nd = v4f64{b0[0], b0[1], b1[0], b1[1]};
                }
                t = upd(0u, (unsigned)(16 * PS * 8), t);
                Frag fx, ft;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    fx.v[q] = S.dli[((k + 1) & 1) * 16 * PS + cl * PS + 4 * q + rg];
                    ft.v[q] = t[q];
                }
                const v4f64 z4 = {0.0, 0.0, 0.0, 0.0};
                const v4f64 d = mfma4(fx, ft, z4, false);
                st_pk(C_u, base_pk + 2048u, lane, d);  // tile (k+1, k+2): the block the inverse reads
                double *pr = pan_nxt + (size_t)cl * PS + rg;
#pragma unroll
                for (int r = 0; r < 4; ++r) pr[4 * r] = d[r];
                // the next diagonal tile: update k from the panel, update k + 1 from the registers of D (its rows are the
                // operand fragments of both sides)
                if (k == 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (rg + 4 * r == cl) nd[r] += pinv[32 + cl];
                }
                nd = upd((unsigned)(16 * PS * 8), (unsigned)(16 * PS * 8), nd);
#pragma unroll
                for (int s = 0; s < 4; ++s) nd = __builtin_amdgcn_mfma_f64_16x16x4f64(-d[s], d[s], nd, 0, 0, 0);
                dg = nd;
            } else {
                dg = v4f64{1.0, 1.0, 1.0, 1.0};
            }
            FSTAMP(8);
            arrive(k);
            TRACE(4);
        }
        TSTAMP(3);
    }
    } else {
    // Worker rows (band < nb): block row I >= 3 belongs to worker (I - 3) mod NWKc for as long as it has tiles in the band
    // (steps 0 .. I - 3); its tiles of columns k + 1 and k + 2 STAY IN REGISTERS from step to step (rA, rB)
    v4f64 rA[2], rB[2];
    bool colready = false;  // the column that enters the band at the next step is known to be back from the helpers
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) rA[s2] = rB[s2] = v4f64{0.0, 0.0, 0.0, 0.0};
    for (int k = 0; k + 1 < nb; ++k) {
        const gdouble *src_u = k == 0 ? A_u : as_global(uniform_ptr(const_cast<const double *>(C)));
        const int m = nb - k - 1;
        double *pan_cur = S.pan + (size_t)(k & 1) * NP * PS;
        double *pan_nxt = S.pan + (size_t)((k + 1) & 1) * NP * PS;
        const unsigned base_pk = (unsigned)((k + 1) * (nb + 1) * 2048);  // tile (k+1, k+1), packed
        const char *pan_b = reinterpret_cast<const char *>(pan_cur);
        auto upd = [&](unsigned pa, unsigned pb, v4f64 a) {
            const double *pa1 = reinterpret_cast<const double *>(pan_b + pa + lane_p);
            const double *pb1 = reinterpret_cast<const double *>(pan_b + pb + lane_p);
#pragma unroll
            for (int s = 0; s < 4; ++s) a = __builtin_amdgcn_mfma_f64_16x16x4f64(-pa1[4 * s], pb1[4 * s], a, 0, 0, 0);
            return a;
        };
        TRACE(0);
        if (CLM == 2 && (wave & 3) == (kChain & 3)) {
            // the waves on the chain's SIMD leave its double-precision units to the chain
            arrive(k);
            __syncthreads();
        } else {
            const int widx = CLM == 2 ? wave - (wave >> 2) - ((wave & 3) > (kChain & 3) ? 1 : 0) : (wave < kChain ? wave : wave - 1);  // 0..NWKc-1
#ifdef FIT_LOOP_TIMING
            long long w_last = clock64();
#endif
            // ---- the band: tiles (I, J), k + 1 < J <= k + 1 + band, J <= I, without tile (k+2, k+2) (the chain wave's) ----
            // enumeration column by column (row by row when there are no helpers of the trailing update: band = nb); the tiles
            // of the LAST band column, from index cntH on, come back from the helper waves at this step
            int cntB = m * (m - 1) / 2, cntH = cntB;
            if (band < nb) {
                const int nbc = min(band, m - 1);
                cntB = 0;
                for (int jj = 1; jj <= nbc; ++jj) {
                    if (jj == band) cntH = cntB;
                    cntB += m - jj;
                }
                if (nbc < band || k == 0) cntH = cntB;
            }
            cntB = __builtin_amdgcn_readfirstlane(cntB);
            cntH = __builtin_amdgcn_readfirstlane(cntH);
            int hc = 0;
            bool anyn = false;
            bool hready = false;
            auto rec_at = [&](int e) -> uint4 {
                if (band < nb) {  // column by column: column j of the band holds the rows i = j .. m - 1
                    int j = 1, i = e;
                    while (i >= m - j) {
                        i -= m - j;
                        ++j;
                    }
                    i += j;
                    return S.rec[(i - 1) * i / 2 + (j - 1)];
                }
                return S.rec[e];
            };
            auto ldt = [&](const uint4 &t, int e) {  // (step 0 reads A: the transpose of tile (I, J) is its tile (J, I))
                if (e >= cntH) {  // a tile of the column that enters the band: updates 0 .. k - 1 were the helpers'
                    if (!hready) {
                        const int J = k + 1 + band, need = seq * (nb - J);
                        const long long t0 = wall_clock64();
                        while (__builtin_amdgcn_readfirstlane(clu::ld(ctl + clu::HCOL + J)) < need) {
                            __builtin_amdgcn_s_sleep(1);
                            if (wall_clock64() - t0 > 1000000 * clu::kTicksPerUs) {
                                if (lane == 0) S.flag[0] = 2;
                                break;
                            }
                        }
                        hready = true;
                    }
                    if (e == cntH) return clu::ld_pk_dev(HD_u, (unsigned)((k + 1 + band) * 2048), lane);  // (its first tile: the diagonal one)
                    return clu::ld_pk_dev(H_u, base_pk + (t.x & ~2047u), lane);  // tile (I, J) waits at the mirror position (J, I)
                }
                return ld_pk(src_u, base_pk + ((k == 0 ? t.x : t.w) & ~2047u), lane);
            };
            auto fin = [&](const uint4 &t, v4f64 a) {
                if (k == 0 && (t.x & 2u)) {  // first touch: add diag(1/p) on diagonal tiles
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (rg + 4 * r == cl) a[r] += pinv[16 * (1 + (t.w & 127u)) + cl];  // (w: packed offset | i)
                }
                a = upd(t.y, t.z, a);
                st_pk(C_u, base_pk + (t.w & ~2047u), lane, a);
                // tiles (k+3, k+2) and (k+3, k+3) are what the chain wave needs right after X_{k+2}: also through LDS (a reload
                // from the L2 behind the barrier was ~0.7 us of every step's chain)
                if (S.hand && (t.w & 127u) == 2u && t.y <= (unsigned)(2 * 16 * PS * 8)) {
                    v2f64 *hp = reinterpret_cast<v2f64 *>(S.hand + ((k & 1) * 2 + (t.y == (unsigned)(16 * PS * 8) ? 0 : 1)) * 256) + lane;
                    hp[0] = v2f64{a[0], a[1]};
                    hp[64] = v2f64{a[2], a[3]};
                }
            };
            // (tile 0 of either enumeration is (k+2, k+2): skipped -- indices below are shifted by one)
            const int cnt = cntB - 1;
            const int ncol = max(m - 2, 0);  // tiles (I, k + 1), I > k + 2 (the first one is the chain wave's)
            int cfirst = widx - max(cnt, 0) % NWKc;  // (the round-robin deal goes on where the band stopped)
            if (cfirst < 0) cfirst += NWKc;
            auto wait_flag = [&]() {
                int spins = 0;
                while (__hip_atomic_load(&S.flag[3], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < k + 1) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1 << 22)) {  // (never observed; a stuck flag must not hang the device)
                        if (lane == 0) *S.flag = 1;
                        break;
                    }
                }
            };
            auto column_finish = [&](int i, v4f64 t, const Frag &fx) {  // D = X T^T = L_{I,k+1}^T -> tile (k+1, I), panel k + 1
                Frag ft;
#pragma unroll
                for (int q = 0; q < 4; ++q) ft.v[q] = t[q];
                const v4f64 z4 = {0.0, 0.0, 0.0, 0.0};
                const v4f64 d = mfma4(fx, ft, z4, false);
                // (the panel first, and its address formed HERE from the lane index: hoisted out of the step loop, cl and rg were
                //  spilled -- five scratch reloads per step in the workers' loop -- and the reload behind the tile's global stores
                //  made the wave wait for their acknowledgement, s_waitcnt vmcnt(0), before it could write the panel)
                int l2 = lane;
                asm volatile("" : "+v"(l2));
                double *pr = pan_nxt + (size_t)((i - 1) * 16 + (l2 & 15)) * PS + (l2 >> 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) pr[4 * r] = d[r];
                st_pk(C_u, base_pk + (unsigned)(i * 2048), l2, d);
            };
            if (band == 2 && band < nb) {
                // ROWS IN REGISTERS.  A tile of the band lives three steps in the first workgroup: it enters from the helpers
                // (column k + 3), is updated again as column k + 2, and is finalised as column k + 1.  Dealt tile by tile (below),
                // a different wave took it at every step -- a store and a reload through the L2 between any two of them, and
                // the early steps, with their ~40 tiles, were worker-bound at ~1.5 us per tile product.  Dealt ROW by row, the
                // same wave has the tile all three times: it stays in registers (rA: column k + 1, rB: column k + 2), the band
                // is never stored, and what a step loads is the one tile per row that enters.  Same operations per tile in the
                // same order: same bits.
                // THE COLUMN THAT ENTERS THE BAND, AHEAD OF THE STEP.  A wave's poll of the helpers' counter and its device-scope
                // load of the returned tile cost ~1.2 + ~0.4 us at the head of every step (in-kernel timeline; the helpers had
                // delivered long before) -- a third of the step.  Now the word of column k + 4 is asked for late in step k
                // (asynchronously, in front of the column tiles), read behind arrive()'s wait, and if the column is there step
                // k + 1 issues the loads of its tiles at once and waits for them behind the products of its other two tiles.
                // (Asked for in the middle of step k it was not there yet: the helpers need ~1.6 us from the end of step k - 1;
                // registers in flight across the loop's back edge are not safe from the compiler's copies.)
                v2f64 rNlo[2], rNhi[2];
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) rNlo[s2] = rNhi[s2] = v2f64{0.0, 0.0};
                if (k > 0 && colready) {  // (issue and use inside one step: no register in flight across the loop's back edge)
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        const int I = 3 + widx + s2 * NWKc;
                        if (I >= nb || k > I - 3) continue;
                        if (I == k + 3) clu::ld_pk_dev_issue(HD_u, (unsigned)((k + 3) * 2048), lane, rNlo[s2], rNhi[s2]);
                        else clu::ld_pk_dev_issue(H_u, (unsigned)(((k + 3) * nb + I) * 2048), lane, rNlo[s2], rNhi[s2]);
                    }
                }
                v4f64 rowC[2];
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const int I = 3 + widx + s2 * NWKc;
                    if (I >= nb || k > I - 3) continue;
                    const int i = I - k - 1;  // row block of the panel
                    if (k == 0) {  // first touch: the transposes from A (tile (J, I) of the symmetric A), 1 / p on the diagonal
                        rA[s2] = ld_pk(src_u, base_pk + (unsigned)(i * 2048), lane);
                        rB[s2] = ld_pk(src_u, base_pk + (unsigned)((nb + i) * 2048), lane);
                        rowC[s2] = ld_pk(src_u, base_pk + (unsigned)((2 * nb + i) * 2048), lane);
                        if (I == 3) {
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (rg + 4 * r == cl) rowC[s2][r] += pinv[16 * 3 + cl];
                        }
                    }
                    rA[s2] = upd(0u, (unsigned)(i * 16 * PS * 8), rA[s2]);
                    rB[s2] = upd((unsigned)(16 * PS * 8), (unsigned)(i * 16 * PS * 8), rB[s2]);
                }
                if (k > 0 && colready) asm volatile("s_waitcnt vmcnt(0)" : "+v"(rNlo[0]), "+v"(rNhi[0]), "+v"(rNlo[1]), "+v"(rNhi[1])::"memory");
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const int I = 3 + widx + s2 * NWKc;
                    if (I >= nb || k > I - 3) continue;
                    const int i = I - k - 1;
                    v4f64 tC = rowC[s2];
                    if (k > 0) {  // column k + 3 comes back from the helpers of the trailing update (updates 0 .. k - 1 applied)
                        if (colready) {
                            tC = v4f64{rNlo[s2][0], rNlo[s2][1], rNhi[s2][0], rNhi[s2][1]};
                        } else {
                            if (!hready) {
                                const int J = k + 3, need = seq * (nb - J);
                                const long long t0 = wall_clock64();
                                while (__builtin_amdgcn_readfirstlane(clu::ld(ctl + clu::HCOL + J)) < need) {
                                    __builtin_amdgcn_s_sleep(1);
                                    if (wall_clock64() - t0 > 1000000 * clu::kTicksPerUs) {
                                        if (lane == 0) S.flag[0] = 2;
                                        break;
                                    }
                                }
                                hready = true;
                            }
                            tC = (I == k + 3) ? clu::ld_pk_dev(HD_u, (unsigned)((k + 3) * 2048), lane)
                                              : clu::ld_pk_dev(H_u, (unsigned)(((k + 3) * nb + I) * 2048), lane);
                        }
                    }
                    tC = upd((unsigned)(2 * 16 * PS * 8), (unsigned)(i * 16 * PS * 8), tC);
                    if (k == I - 3) {  // the row's last step here: tiles (I, I - 1) and (I, I) go to the chain wave
                        if (S.hand) {
                            v2f64 *hp = reinterpret_cast<v2f64 *>(S.hand + ((k & 1) * 2) * 256) + lane;
                            hp[0] = v2f64{rB[s2][0], rB[s2][1]};
                            hp[64] = v2f64{rB[s2][2], rB[s2][3]};
                            hp[128] = v2f64{tC[0], tC[1]};
                            hp[192] = v2f64{tC[2], tC[3]};
                        } else {
                            st_pk(C_u, base_pk + (unsigned)((2 * nb + 1) * 2048), lane, rB[s2]);
                            st_pk(C_u, base_pk + (unsigned)((2 * nb + 2) * 2048), lane, tC);
                        }
                    }
                    rowC[s2] = tC;  // (rA is finalised behind the flag; rB and this tile become next step's rA, rB there)
                }
                WSTAMP(11);
                TRACE(1);
                bool any = false;
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const int I = 3 + widx + s2 * NWKc;
                    any = any || (I < nb && k <= I - 3);
                }
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const int I = 3 + widx + s2 * NWKc;
                    anyn = anyn || (I < nb && k + 1 <= I - 3);
                }
                if (any) {
                    wait_flag();
                    TRACE(2);
                    Frag fx;  // L_{k+1,k+1}^-1: as A operand X, as B operand X^T
#pragma unroll
                    for (int q = 0; q < 4; ++q) fx.v[q] = S.dli[((k + 1) & 1) * 16 * PS + cl * PS + 4 * q + rg];
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        const int I = 3 + widx + s2 * NWKc;
                        if (I >= nb || k > I - 3) continue;
                        column_finish(I - k - 1, rA[s2], fx);
                        rA[s2] = rB[s2];
                        rB[s2] = rowC[s2];
                    }
                }
                // (as late as it can be and still return under arrive()'s wait for the stores above: the helpers hand the column
                //  back ~1.5 us after the end of the step before; asked in front of the column tiles the word was often short)
                if (anyn)
                    asm volatile("s_nop 4\n\tglobal_load_dword %0, %1, %2 sc1" : "=&v"(hc) : "v"(0), "s"(ctl + clu::HCOL + k + 4) : "memory");
            } else {
                int e = widx;
                if (e < cnt) {
                    uint4 ta = rec_at(e + 1), tb = ta, tc = ta;
                    v4f64 a = ldt(ta, e + 1), b = a, c = a;
                    if (e + NWKc < cnt) {
                        tb = rec_at(e + NWKc + 1);
                        b = ldt(tb, e + NWKc + 1);
                    }
                    for (;;) {
                        if (e + 2 * NWKc < cnt) {
                            tc = rec_at(e + 2 * NWKc + 1);
                            c = ldt(tc, e + 2 * NWKc + 1);
                        }
                        fin(ta, a);
                        if (e + NWKc >= cnt) break;
                        if (e + 3 * NWKc < cnt) {
                            ta = rec_at(e + 3 * NWKc + 1);
                            a = ldt(ta, e + 3 * NWKc + 1);
                        }
                        fin(tb, b);
                        if (e + 2 * NWKc >= cnt) break;
                        if (e + 4 * NWKc < cnt) {
                            tb = rec_at(e + 4 * NWKc + 1);
                            b = ldt(tb, e + 4 * NWKc + 1);
                        }
                        fin(tc, c);
                        if (e + 3 * NWKc >= cnt) break;
                        e += 3 * NWKc;
                    }
                }
                WSTAMP(11);
                TRACE(1);
                // ---- column k + 1 without its first tile (the chain wave's): update, then the panel of step k + 1 ----
                if (cfirst < ncol) {
                    wait_flag();
                    TRACE(2);
                    Frag fx;
#pragma unroll
                    for (int q = 0; q < 4; ++q) fx.v[q] = S.dli[((k + 1) & 1) * 16 * PS + cl * PS + 4 * q + rg];
                    for (int c = cfirst; c < ncol; c += NWKc) {
                        const int i = c + 2;  // block row I = k + 1 + i
                        v4f64 t = ld_pk(src_u, base_pk + (unsigned)(k == 0 ? i * 2048 : i * nb * 2048), lane);
                        t = upd(0u, (unsigned)(i * 16 * PS * 8), t);
                        column_finish(i, t, fx);
                    }
                }
            }
            TRACE(3);
            arrive(k);
            colready = false;
            if (anyn) {  // (arrive() has waited for the word)
                asm volatile("" : "+v"(hc));
                colready = __builtin_amdgcn_readfirstlane(hc) >= seq * (nb - (k + 4));
            }
            TRACE(4);
            WSTAMP(12);
            __syncthreads();
        }
        TSTAMP(3);
    }
    }
    if (wave == kChain && nb > 1) __syncthreads();  // (the barrier that ends the workers' last step)
    __syncthreads();
    if (*S.flag) return false;
    TSTAMP(4);
    {
        // Tr2 and m = Y mu come from the helper waves that own the block columns of W (exchange area: the fit's WdT buffer)
        const int T = clu::inv_helpers(P) * NW, nsig = T < nb ? T : nb;
        if (tid == 0) {
            const long long t0 = wall_clock64();
            while (clu::ld(ctl + clu::DONE) < seq * nsig) {
                __builtin_amdgcn_s_sleep(1);
                if (wall_clock64() - t0 > 1000000 * clu::kTicksPerUs) {  // 1 s: the helpers are gone
                    S.flag[0] = 2;
                    break;
                }
            }
        }
        __syncthreads();
        if (*S.flag) return false;
        const double *xg = P.WdT;
        for (int i = tid; i < N; i += KT) {  // (device-scope loads: the helpers' stores, not this compute unit's L1)
            S.tr2[i] = clu::ld_dev(xg + i);
            S.m[i] = clu::ld_dev(xg + NP + i);
        }
        __syncthreads();
        TSTAMP(7);
    }
    return true;
}

// ---- the posterior solve, left-looking (one workgroup, N <= 335) -----------------------------------------------------------
// solve_posterior above is right-looking: step k rewrites every tile of the trailing triangle -- 1 140 loads and as many stores
// of 2 KB per pass at N = 300 -- and the rows of the inverse read both their operands from memory (2 622 loads): 10.9 MB per
// pass through the L1 of the one compute unit, and, with ~16 fit loops per XCD sharing 4 MB of L2, most of it beyond the L2
// (137 us per pass alone, 200-270 with the device full).  Here a tile is formed ONCE: at step J the tiles (I, J) of block column
// J take all their updates k < J at a stretch, T_IJ = A_IJ - sum_k L_Ik L_Jk^T, one operand -- row J of L, the same for every
// tile of the column -- from LDS, the other from memory, are multiplied by X_JJ^T when the chain wave has factored the
// diagonal tile, and stored once; row J of the inverse, W_JK = -X_JJ sum_m L_Jm W_mK, takes its L operands from the SAME LDS
// row.  Row J + 1 is staged in a second buffer while step J runs (its last tile comes straight from the wave that forms it).
// The diagonal tiles stay right-looking, in LDS: the wave that forms L_IJ updates diagonal tile I at once, so the chain wave
// finds tile (J, J) complete at the start of step J.  6.4 MB per pass instead of 10.9, a quarter of the stores.
// Every tile sees the operations of solve_posterior in the same order -- first touch (A, 1 / p on the diagonal), the updates
// k = 0, 1, .., the product with X^T; the chains of the inverse over ascending m -- so the results are the same bits.
// Work of step J: nb - 1 items (nb - 1 - J column tiles of J products, J inverse tiles of J .. 1 products), dealt longest
// first to the eleven worker waves, at most two each; a wave forms the sums of its items, waits for the chain wave's flag,
// and finishes them.  LDS: the two row buffers take the place of the two panels, the diagonal tiles that of the band factors,
// scan tables and tile table (the bands and tables move to the W buffer in global memory, as in the wide instantiation).
__device__ __forceinline__ v4f64 lds_tile(const double *t, int lane) {
    const v2f64 *hp = reinterpret_cast<const v2f64 *>(t) + lane;
    const v2f64 lo = hp[0], hi = hp[64];
    return v4f64{lo[0], lo[1], hi[0], hi[1]};
}
__device__ __forceinline__ void lds_tile_store(double *t, int lane, const v4f64 &v) {
    v2f64 *hp = reinterpret_cast<v2f64 *>(t) + lane;
    hp[0] = v2f64{v[0], v[1]};
    hp[64] = v2f64{v[2], v[3]};
}
// acc (+/-)= sum_{t < n} Ltile[t] x Btile[t]: the A operands are consecutive packed tiles of an LDS row, the B operands packed
// tiles `blk` bytes apart in memory, fetched through a ring of four register sets with the loads issued and waited for by hand
// (three products in flight behind the one being multiplied; the order of the sum is t ascending)
template <bool NEG>
__device__ __forceinline__ v4f64 ll_chain(const double *arow, const gdouble *Bu, unsigned ob, unsigned blk, int n, v4f64 acc, int lane) {
    struct Operands {
        v2f64 lo, hi;
    };
    const unsigned lane_o = (unsigned)lane * 16u;
    auto issue = [&](Operands &o, int t) {
        const unsigned pb = ob + (unsigned)t * blk + lane_o;
        asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %2, %3\n\tglobal_load_dwordx4 %1, %2, %3 offset:1024"
                     : "=&v"(o.lo), "=&v"(o.hi)
                     : "v"(pb), "s"(Bu)
                     : "memory");
    };
    auto consume = [&](Operands &o, int t) {
        const v4f64 fa = lds_tile(arow + (size_t)t * 256, lane);
        switch (min(3, n - 1 - t)) {  // products issued after t (two loads each)
            case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            case 1: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        }
        asm volatile("" : "+v"(o.lo), "+v"(o.hi));
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(NEG ? -fa[0] : fa[0], o.lo[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(NEG ? -fa[1] : fa[1], o.lo[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(NEG ? -fa[2] : fa[2], o.hi[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(NEG ? -fa[3] : fa[3], o.hi[1], acc, 0, 0, 0);
    };
    if (n <= 0) return acc;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the counts are of THESE loads)
    Operands s0, s1, s2, s3;
    issue(s0, 0);
    if (1 < n) issue(s1, 1);
    if (2 < n) issue(s2, 2);
    for (int t = 0;; t += 4) {
        if (t + 3 < n) issue(s3, t + 3);
        consume(s0, t);
        if (t + 1 >= n) break;
        if (t + 4 < n) issue(s0, t + 4);
        consume(s1, t + 1);
        if (t + 2 >= n) break;
        if (t + 5 < n) issue(s1, t + 5);
        consume(s2, t + 2);
        if (t + 3 >= n) break;
        if (t + 6 < n) issue(s2, t + 6);
        consume(s3, t + 3);
        if (t + 4 >= n) break;
    }
    return acc;
}
__device__ __forceinline__ bool solve_posterior_ll(const FitLoopParams &P, const Smem &S) {
    constexpr int NWK_ = NW - 1;
    const int N = P.N, NP = P.NP, nb = P.NP / 16, ld = P.NP;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cl = lane & 15, rg = lane >> 4;
    double *C = P.C;  // upper tiles (k, I): L_Ik^T; lower tiles (r, K): W = L^-1; diagonal tiles: X_kk
    double *rowL[2] = {S.pan, S.pan + (size_t)(nb - 1) * 256};
    double *dg = S.band;  // (LDS: solve_posterior's band / scan / tile-table region) the diagonal tiles, packed
#ifdef FIT_LOOP_TIMING
    long long t_last = clock64();
#endif
    for (int i = tid; i < NP; i += KT) S.y[i] = i < N ? 1.0 / S.p[i] : 1.0;  // row N (b) and padding: 1
    if (tid == 0) {
        S.flag[0] = 0;  // not positive definite
        S.flag[3] = 0;  // diagonal tiles factored so far
    }
    __syncthreads();
    const double *pinv = S.y;
    const gdouble *A_u = as_global(uniform_ptr(P.A));
    gdouble *C_u = as_global(uniform_ptr(C));
    const gdouble *Cr_u = as_global(uniform_ptr(const_cast<const double *>(C)));
    // diagonal tiles: A_JJ + diag(1 / p)
    for (int J = wave; J < nb; J += NW) {
        v4f64 t = ld_pk(A_u, (unsigned)(J * (nb + 1) * 2048), lane);
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (rg + 4 * r == cl) t[r] += pinv[16 * J + cl];
        lds_tile_store(dg + (size_t)J * 256, lane, t);
    }
    __syncthreads();
    TSTAMP(0);
    const int aug_tile = N / 16, aug_c = N - 16 * aug_tile;
    auto cs_ptr = [&](int I, int J) { return P.cs + ((size_t)I * nb + J) * 16; };
    auto rows_valid = [&](int I) { return min(16, max(0, N - 16 * I)); };
    for (int J = 0; J < nb; ++J) {
        const double *rowc = rowL[J & 1];
        double *rown = rowL[(J + 1) & 1];
        double *dli_J = S.dli + (J & 1) * 16 * PS;
        [[maybe_unused]] const int k = J;  // (the trace macros of the timing build index the step by this name)
        TRACE(0);
        if (wave == kChain) {
#ifdef FIT_LOOP_TIMING
            long long f_last = clock64();
#endif
            v4f64 a = lds_tile(dg + (size_t)J * 256, lane), xi;
            FSTAMP(8);
            const bool ok = factor_invert_tile(a, xi, dli_J, lane, aug_tile == J ? aug_c : -1);
            if (!ok && lane == 0) *S.flag = 1;
            FSTAMP(9);
            store_col_ssq(xi, cs_ptr(J, J), rows_valid(J), lane);
            st_pk(C_u, (unsigned)(J * (nb + 1) * 2048), lane, xi);  // W_JJ = X_JJ
            __hip_atomic_store(&S.flag[3], J + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            FSTAMP(10);
            TRACE(5);
        } else {
            const int widx = wave < kChain ? wave : wave - 1;  // 0 .. 10
#ifdef FIT_LOOP_TIMING
            long long w_last = clock64();
#endif
            // items of step J, longest first: the inverse tile (J, 0) [J products], the column tiles (I, J), I = J + 1 .. nb - 1
            // [J products each], the inverse tiles (J, K), K = 1 .. J - 1 [J - K products]
            const int ncolt = nb - 1 - J, nit = ncolt + J;
            v4f64 acc[2];
            int kind[2], idx[2];  // kind 0: none, 1: column tile (I = idx), 2: inverse tile (K = idx)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                // (dealt back and forth: wave w takes item w and item 2 NWK_ - 1 - w, a long one and a short one)
                const int it = s2 == 0 ? widx : 2 * NWK_ - 1 - widx;
                kind[s2] = 0;
                idx[s2] = 0;
                if (it < nit) {
                    if (J > 0 && it == 0) {
                        kind[s2] = 2;
                        idx[s2] = 0;
                    } else if (it - (J > 0 ? 1 : 0) < ncolt) {
                        kind[s2] = 1;
                        idx[s2] = J + 1 + it - (J > 0 ? 1 : 0);
                    } else {
                        kind[s2] = 2;
                        idx[s2] = it - ncolt;  // K = 1 .. J - 1
                    }
                }
            }
            // staging of row J + 1 for the next step: mirror tiles (k, J + 1), k < J (tile k = J comes from its column item)
            v4f64 stg[2];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int k = widx + s2 * NWK_;
                if (J + 1 < nb && k < J) stg[s2] = ld_pk(Cr_u, (unsigned)((k * nb + J + 1) * 2048), lane);
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                if (kind[s2] == 1) {  // T_IJ^T = A_JI - sum_k L_Jk L_Ik^T (kept transposed, as solve_posterior keeps it)
                    const int I = idx[s2];
                    const v4f64 a = ld_pk(A_u, (unsigned)((J * nb + I) * 2048), lane);
                    acc[s2] = ll_chain<true>(rowc, Cr_u, (unsigned)(I * 2048), (unsigned)(nb * 2048), J, a, lane);
                } else if (kind[s2] == 2) {  // sum_{m=K}^{J-1} L_Jm W_mK
                    const int K = idx[s2];
                    const v4f64 a = {0.0, 0.0, 0.0, 0.0};
                    acc[s2] = ll_chain<false>(rowc + (size_t)K * 256, Cr_u, (unsigned)((K * nb + K) * 2048), (unsigned)(nb * 2048),
                                              J - K, a, lane);
                }
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int k = widx + s2 * NWK_;
                if (J + 1 < nb && k < J) lds_tile_store(rown + (size_t)k * 256, lane, stg[s2]);
            }
            WSTAMP(11);
            TRACE(1);
            if (kind[0]) {
                int spins = 0;
                while (__hip_atomic_load(&S.flag[3], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < J + 1) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1 << 22)) {  // (never observed; a stuck flag must not hang the device)
                        if (lane == 0) *S.flag = 1;
                        break;
                    }
                }
                TRACE(2);
                Frag fx;  // X_JJ as A operand
#pragma unroll
                for (int q = 0; q < 4; ++q) fx.v[q] = dli_J[cl * PS + 4 * q + rg];
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    Frag ft;
#pragma unroll
                    for (int q = 0; q < 4; ++q) ft.v[q] = acc[s2][q];
                    const v4f64 z4 = {0.0, 0.0, 0.0, 0.0};
                    if (kind[s2] == 1) {
                        const int I = idx[s2];
                        const v4f64 d = mfma4(fx, ft, z4, false);  // D = X T^T = L_IJ^T
                        st_pk(C_u, (unsigned)((J * nb + I) * 2048), lane, d);  // mirror tile (J, I)
                        if (I == J + 1) lds_tile_store(rown + (size_t)J * 256, lane, d);
                        // diagonal tile I: update J (the rows of D are the operand fragments of both sides)
                        v4f64 nd = lds_tile(dg + (size_t)I * 256, lane);
#pragma unroll
                        for (int q = 0; q < 4; ++q) nd = __builtin_amdgcn_mfma_f64_16x16x4f64(-d[q], d[q], nd, 0, 0, 0);
                        lds_tile_store(dg + (size_t)I * 256, lane, nd);
                    } else if (kind[s2] == 2) {
                        const int K = idx[s2];
                        const v4f64 w = mfma4(fx, ft, z4, true);  // W_JK = -X_JJ sum
                        st_pk(C_u, (unsigned)((J * nb + K) * 2048), lane, w);
                        double ssq = 0.0;
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (16 * J + rg + 4 * r < N) ssq = fma(w[r], w[r], ssq);
                        ssq += __shfl_xor(ssq, 16);
                        ssq += __shfl_xor(ssq, 32);
                        if (rg == 0) cs_ptr(J, K)[lane & 15] = ssq;
                    }
                }
            }
            TRACE(3);
            TRACE(4);
            WSTAMP(12);
        }
        __syncthreads();
        if (*S.flag) return false;
        TSTAMP(3);
    }
    TSTAMP(4);
    // m = -(row N of W),  tr2_i = sum over the block column of the tile column sums (fixed order)
    for (int i = tid; i < N; i += KT) {
        const int J = i >> 4, c = i & 15;
        double t2 = 0.0;
        for (int I = J; I < nb; ++I) t2 += cs_ptr(I, J)[c];
        S.tr2[i] = t2;
        {
            const int rr = N - 16 * aug_tile, q = rr >> 2, ln = (rr & 3) * 16 + c;
            S.m[i] = -C[((size_t)aug_tile * nb + J) * 256 + (q >> 1) * 128 + ln * 2 + (q & 1)];
        }
    }
    __syncthreads();
    TSTAMP(7);
    return true;
}

// ---- the posterior solve with the matrix RESIDENT IN REGISTERS (round 5; fit_loop_rr.hip: 512 threads, CLM = 6) -------------------
// With the device full a pass of the one-workgroup solve is bound by the bytes it moves beyond the L2 (3.5-5 MB per pass at N = 300,
// profiles/r05_pmc_fit_loop_256_resident.json), and every form so far kept the working matrix in memory: the lower triangle of
// N = 300 is 190 tiles of 2 KB = 380 KB, more than twice the LDS.  But a CU has 512 KB of vector registers.  Eight waves (two
// per SIMD, 256 registers each) hold 24 tiles apiece: wave w owns whole block ROWS -- (18 - w, 4 + w) for w < 7, (11, 3, 2, 1, 0)
// for w = 7: 24 tiles each, 22 for the last -- and a tile never leaves its registers from the load of A to the end of the pass:
//   * slot (I, J) holds T_IJ^T while the factorisation runs, L_IJ^T from step J - 1 on (the accumulator layout of a matrix is the
//     B fragment of itself and the A fragment of its transpose: D = X T^T takes the slot as it stands, the inverse takes L_IK from
//     it as it stands), then the running sum S_IJ = sum_K L_IK W_KJ of the inverse, then W_IJ;
//   * what the OTHER waves need travels through LDS in the packed tile format: panel k (the tiles L_Ik^T, I > k: both operands of the
//     trailing update) and row K of W (the B operands of the inverse).  At step k the panel occupies the positions I > k of a
//     buffer of nb tiles and row k - 1 of W its positions J < k: two buffers of nb tiles (76 KB) carry both, double buffered;
//   * the inverse runs IN PLACE, one step behind the factorisation: at step k every wave adds L_{I,k-1} W_{k-1,J} to its sums
//     (rows I >= k, visited with J descending: slot (I, k - 1) still holds the L tile when the row is entered and becomes the sum
//     S_{I,k-1} = L_{I,k-1} W_{k-1,k-1}; the sums left of it take their next product); row k is then complete and its owner
//     multiplies by -X_kk and publishes it for step k + 1.  The sums run over K ascending into one accumulator, as inverse_tile's.
// Per pass the workgroup reads A (380 KB, from the L2 or beyond) and nothing else: no C, no W, no cs buffer.
// One barrier per step; the owner of row k + 1 updates, factors and inverts tile (k + 1, k + 1) FIRST (the chain every form of this
// kernel waits for), the others meet its result behind their own products.  The same products, the same operands, the same order
// per tile as solve_posterior: the same bits.  N <= 303 (nb <= 19).
#ifdef FIT_LOOP_RR
constexpr int kRRSlots = 25, kRRMaxNB = 19, kRRRows = 6, kRRSpec = 7;
// Who holds what.  Wave 7, the SPECIALIST, runs the factor-and-invert chain of every step and nothing else -- the chain is what a
// step waits for, and a wave that had its share of the step's products to do as well made the step chain + share (5 + 4 us of
// 10).  It holds no tiles: the 19 DIAGONAL tiles live in LDS (38 KB; a diagonal tile takes one product per step, formed by the
// worker that holds its row, whose operands are the row's shared operand it has in registers anyway).  The WORKERS,
// waves 0 .. 6, hold the tiles left of the diagonal by whole block rows, J = I - 1 .. 0 in consecutive slots: (18, 7), (17, 8),
// (16, 9), (15, 10), (14, 11), (13, 12) -- 25 tiles each -- and the six short rows 6 .. 1 (21 tiles) on wave 3, the wave that
// shares the specialist's SIMD (double-precision vector and matrix instructions share a SIMD's units: the chain runs at
// full speed once that neighbour has nothing left to do).
__device__ __forceinline__ void rr_rows(int w, int nb, int (&rI)[kRRRows], int (&rS)[kRRRows]) {
#pragma unroll
    for (int r = 0; r < kRRRows; ++r) rI[r] = -100, rS[r] = 0;
    if (w == 3) {
        int st = 0;
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            rI[r] = 6 - r;
            rS[r] = st;
            st += 6 - r;
        }
    } else if (w < kRRSpec) {
        const int a = w < 3 ? 18 - w : 19 - w;  // 18, 17, 16, -, 15, 14, 13
        rI[0] = a, rS[0] = 0;
        rI[1] = 25 - a, rS[1] = a;
    }
#pragma unroll
    for (int r = 0; r < kRRRows; ++r)
        if (rI[r] >= nb) rI[r] = -100;
}
// slot s of wave w: (I << 8) | J, or 0xffff (no tile)
__device__ __forceinline__ unsigned rr_slot_ij(int w, int s, int nb) {
    if (w == kRRSpec) return 0xffffu;
    int rI[kRRRows], rS[kRRRows];
    rr_rows(w, nb, rI, rS);
    unsigned e = 0xffffu;
#pragma unroll
    for (int r = 0; r < kRRRows; ++r)
        if (rI[r] > 0 && s >= rS[r] && s < rS[r] + rI[r]) e = (unsigned)((rI[r] << 8) | (rI[r] - 1 - (s - rS[r])));
    return e;
}
// the tables of all eight waves, built once per fit (thread e: one word): per wave kRRTabStride ints -- [0, 13) the slots' (I, J),
// two to a word; [16, 22) the rows' block rows; [22, 28) their first slots
constexpr int kRRTabStride = 32;
__device__ __forceinline__ void rr_tables(int *tab, int nb, int tid) {
    for (int e = tid; e < 8 * kRRTabStride; e += KT) {
        const int w = e / kRRTabStride, i = e % kRRTabStride;
        int v = 0;
        if (i < (kRRSlots + 1) / 2) {
            v = (int)(rr_slot_ij(w, 2 * i, nb) | ((2 * i + 1 < kRRSlots ? rr_slot_ij(w, 2 * i + 1, nb) : 0xffffu) << 16));
        } else if (i >= 16 && i < 16 + 2 * kRRRows) {
            int rI[kRRRows], rS[kRRRows];
            rr_rows(w, nb, rI, rS);
#pragma unroll
            for (int r = 0; r < kRRRows; ++r) {
                if (i == 16 + r) v = rI[r];
                if (i == 16 + kRRRows + r) v = rS[r];
            }
        }
        tab[e] = v;
    }
}
#define RR_KEEP_BRANCH asm volatile("" ::: "memory")  // (a branch, not 8 selects per slot: the condition is wave-uniform)
// T += A B for one k-step quadruple, IN PLACE: the accumulator is tied to its own registers (the builtin leaves the destination to
// the register allocator, which under 24 live tiles answered with copies of whole tiles behind s_nop 14).  The hazard recogniser
// does not look into inline asm: a VALU write of an operand needs a wait state before the matrix instruction reads it (s_nop 1 in
// front), and a read of T by anything but the next in-place product must come 18 wait states after the last product -- also
// the copies and spills the register allocator may place directly behind a block, which is why the block itself ends on them
// (a first version waited once per step, before its own reads: its results depended on where the allocator put its copies).
// (the operand a row shares travels as four separate 64-bit pairs: a whole-tile operand was copied in front of every product;
//  RR_NO_BLOCK_NOPS: the blocks without their wait states, for tools/check_mfma_hazard.py's self-test only)
#ifdef RR_NO_BLOCK_NOPS
#define RR_BLOCK_TAIL
#else
#define RR_BLOCK_TAIL "\n\ts_nop 15\n\ts_nop 2" /* the results settle before anything the compiler may place behind the block reads them */
#endif
#define RR_MFMA_SETTLE asm volatile("s_nop 15\n\ts_nop 3" ::: "memory")
// The operand tile of a slot, read from LDS ONE SLOT AHEAD by hand (two ds_read_b128, issued and waited for explicitly: the wait is
// "all but the two reads issued last", which the compiler's own counting cannot express across the branches of the slots).  The
// reads are issued at every slot, used or not -- no register of the double buffer is ever defined under a branch.
// The address is a row's base + a compile-time offset: within a row the slots go down the columns one by one, J = c - s, so
// the operand of slot s is at (buffer + (c - 24) tiles) + (24 - s) tiles -- one vector add per ROW (2 - 6 per pass) instead of
// one per slot.  (The buffers sit behind >= 48 KB of LDS -- the vectors, the diagonal tiles, padding for the small sizes -- so
// that a base never falls below the start of LDS.)
template <int OFF>
__device__ __forceinline__ void rr_lds_issue_imm(v2f64 &lo, v2f64 &hi, unsigned base) {
    asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4" : "=&v"(lo), "=&v"(hi) : "v"(base), "n"(OFF), "n"(OFF + 1024) : "memory");
}
template <typename F, int... Ss>
__device__ __forceinline__ void rr_for_slots(F &&f, std::integer_sequence<int, Ss...>) {
    (f(std::integral_constant<int, Ss>{}), ...);
}
constexpr int kRRBufferMinOffset = 24 * 2048;  // bytes of LDS in front of the tile buffers
#define RR_LDS_WAIT(n) asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory")
__device__ __forceinline__ void rr_mfma4_bp(v4f64 &T, const v2f64 &alo, const v2f64 &ahi, double b0, double b1, double b2, double b3) {
    asm volatile("s_nop 1\n\t"
                 "v_mfma_f64_16x16x4_f64 %0, %1, %5, %0\n\t"
                 "v_mfma_f64_16x16x4_f64 %0, %2, %6, %0\n\t"
                 "v_mfma_f64_16x16x4_f64 %0, %3, %7, %0\n\t"
                 "v_mfma_f64_16x16x4_f64 %0, %4, %8, %0" RR_BLOCK_TAIL
                 : "+v"(T)
                 : "v"(alo[0]), "v"(alo[1]), "v"(ahi[0]), "v"(ahi[1]), "v"(b0), "v"(b1), "v"(b2), "v"(b3));
}
__device__ __forceinline__ void rr_mfma4_ap(v4f64 &T, double a0, double a1, double a2, double a3, const v2f64 &blo, const v2f64 &bhi) {
    asm volatile("s_nop 1\n\t"
                 "v_mfma_f64_16x16x4_f64 %0, %1, %5, %0\n\t"
                 "v_mfma_f64_16x16x4_f64 %0, %2, %6, %0\n\t"
                 "v_mfma_f64_16x16x4_f64 %0, %3, %7, %0\n\t"
                 "v_mfma_f64_16x16x4_f64 %0, %4, %8, %0" RR_BLOCK_TAIL
                 : "+v"(T)
                 : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(blo[0]), "v"(blo[1]), "v"(bhi[0]), "v"(bhi[1]));
}

__device__ __forceinline__ bool solve_posterior_rr(const FitLoopParams &P, const Smem &S) {
    const int N = P.N, NP = P.NP, nb = P.NP / 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cl = lane & 15, rg = lane >> 4;
    for (int i = tid; i < NP; i += KT) S.y[i] = i < N ? 1.0 / S.p[i] : 1.0;  // row N (b) and padding: 1
    if (tid == 0) {
        S.flag[0] = 0;  // not positive definite
        S.flag[3] = 0;  // index of the last diagonal tile whose inverse is in LDS
        S.flag[4] = 0;  // rows of the inverse whose raw sums are in LDS
        S.flag[5] = S.flag[6] = 0;  // work items of a step that any wave may take (even / odd steps)
    }
    __syncthreads();
    const double *pinv = S.y;
    const int aug_tile = N / 16, aug_c = N - 16 * aug_tile;
    auto rows_valid = [&](int I) { return min(16, max(0, N - 16 * I)); };
    const gdouble *A_u = as_global(uniform_ptr(P.A));
    const int tile_doubles = 256;
    const bool spec = wave == kRRSpec;
#ifdef FIT_LOOP_TIMING
    const long long t_entry = clock64();
#endif

    // the slots' (I, J), two to a scalar register (16 bits each), and the rows of this wave
    // (from a table in LDS built once per fit, rr_tables: formed here they were ~3 000 scalar instructions of every pass)
    const int *const tab = S.dcnt + wave * kRRTabStride;
    unsigned ijp[(kRRSlots + 1) / 2];
#pragma unroll
    for (int h = 0; h < (kRRSlots + 1) / 2; ++h) ijp[h] = (unsigned)__builtin_amdgcn_readfirstlane(tab[h]);
#define RR_E(s) ((ijp[(s) >> 1] >> (16 * ((s) & 1))) & 0xffffu)
    // (inside the step loop the packed words go through an opaque move first: hoisted out of the loop, the 25 operand offsets and the
    //  25 row indices were 50 scalar registers, spilled to vector lanes and read back with a v_readlane per use -- and with two
    //  waves on a SIMD a VECTOR instruction issued while the other wave's matrix instructions run costs ~65 cycles
    //  (tools/microbench/tile_step_bench.hip), a scalar one nothing)
    auto slot_e = [&](int s) __attribute__((always_inline)) {
        unsigned w = ijp[s >> 1];
        asm volatile("" : "+s"(w));
        return (w >> (16 * (s & 1))) & 0xffffu;
    };
    int rI[kRRRows], rS[kRRRows];
#pragma unroll
    for (int r = 0; r < kRRRows; ++r) {
        rI[r] = __builtin_amdgcn_readfirstlane(tab[16 + r]);
        rS[r] = __builtin_amdgcn_readfirstlane(tab[16 + kRRRows + r]);
    }
    unsigned mRow = 0;  // the first slots of this wave's rows
#pragma unroll
    for (int r = 0; r < kRRRows; ++r)
        if (rI[r] > 0) mRow |= 1u << rS[r];

    v4f64 T[kRRSlots];
    // (1) the tiles from A: T_IJ^T is tile (J, I) of the symmetric A; 1/p on the diagonal before anything else touches the tile
#pragma unroll
    for (int s = 0; s < kRRSlots; ++s) {
        T[s] = v4f64{0.0, 0.0, 0.0, 0.0};
        const unsigned e = RR_E(s);
        if (e != 0xffffu) {
            const int I = e >> 8, J = e & 255;
            T[s] = ld_pk(A_u, (unsigned)((J * nb + I) * 2048), lane);
            if (I == J) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (rg + 4 * r == cl) T[s][r] += pinv[16 * I + cl];
            }
        }
    }
    // m = -(row N of W): element (N - 16 aug_tile, c) of the tiles of the last block row
    auto put_m = [&](const v4f64 &w, int J) {
        const int rr = N - 16 * aug_tile, r = rr >> 2;
        const double v = r == 0 ? w[0] : (r == 1 ? w[1] : (r == 2 ? w[2] : w[3]));
        if (rg == (rr & 3)) S.m[16 * J + cl] = -v;
    };
    double *const diag = S.hand;  // the diagonal tiles (packed, nb x 2 KB): T_II^T, then X_II = W_II from step I - 1 on
    auto diag_from_A = [&](int I) {
        v4f64 t = ld_pk(A_u, (unsigned)((I * nb + I) * 2048), lane);
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (rg + 4 * r == cl) t[r] += pinv[16 * I + cl];
        return t;
    };
    // the diagonal tiles from A into LDS: every worker those of its rows
    if (!spec) {
#pragma unroll
        for (int r = 0; r < kRRRows; ++r)
            if (rI[r] > 0) lds_tile_store(diag + (size_t)rI[r] * tile_doubles, lane, diag_from_A(rI[r]));
    }
    // tile (0, 0)
    if (spec) {
        v4f64 t0 = diag_from_A(0), x0;
        const bool ok = factor_invert_tile(t0, x0, S.dli, lane, aug_tile == 0 ? aug_c : -1);
        if (!ok && lane == 0) S.flag[0] = 1;
        store_col_ssq(x0, S.tr2, rows_valid(0), lane);
        lds_tile_store(diag, lane, x0);
    }
    __syncthreads();
    // column 0: L_I0^T = X_00 T_I0^T into the slot and into panel 0 (buffer 0, position I) -- the last slot of every row
    if (!spec) {
        Frag fx;
#pragma unroll
        for (int q = 0; q < 4; ++q) fx.v[q] = S.dli[cl * PS + 4 * q + rg];
        unsigned mC0 = 0;
#pragma unroll
        for (int r = 0; r < kRRRows; ++r)
            if (rI[r] > 0) mC0 |= 1u << (rS[r] + rI[r] - 1);
#pragma unroll
        for (int s = 0; s < kRRSlots; ++s) {
            if (mC0 & (1u << s)) {
                RR_KEEP_BRANCH;
                Frag ft;
#pragma unroll
                for (int q = 0; q < 4; ++q) ft.v[q] = T[s][q];
                const v4f64 z4 = {0.0, 0.0, 0.0, 0.0};
                T[s] = mfma4(fx, ft, z4, false);
                lds_tile_store(S.pan + (size_t)(RR_E(s) >> 8) * tile_doubles, lane, T[s]);
            }
        }
    }
    __syncthreads();

    for (int k = 0; k < nb; ++k) {
        if (S.flag[0]) return false;
        const int m = nb - k - 1;
        if (tid == KT - 1) S.flag[5 + ((k + 1) & 1)] = 0;  // (the next step's counter: nobody reads it during this one)
        double *const bufk = S.pan + (size_t)(k & 1) * nb * tile_doubles;        // panel k (positions > k), row k - 1 of W (< k)
        double *const bufn = S.pan + (size_t)((k + 1) & 1) * nb * tile_doubles;  // panel k + 1, row k of W: written at this step
        const double *const dli_k = S.dli + (k & 1) * 16 * PS;
        double *const dli_n = S.dli + ((k + 1) & 1) * 16 * PS;
        TRACE(0);
#ifdef FIT_LOOP_TIMING
        if (P.trace_on && lane == 0 && k == 0) P.timing[16 + (wave * 20 + 0) * 6 + 5] = t_entry;  // (tools/rr_trace.py: the prologue)
#endif
        if (spec) {
            // ---- the chain: tile (k + 1, k + 1) takes its last update, is factored and inverted ----
            if (m > 0) {
                v4f64 a = lds_tile(diag + (size_t)(k + 1) * tile_doubles, lane);
                const v4f64 pa = lds_tile(bufk + (size_t)(k + 1) * tile_doubles, lane);
#pragma unroll
                for (int q = 0; q < 4; ++q) a = __builtin_amdgcn_mfma_f64_16x16x4f64(-pa[q], pa[q], a, 0, 0, 0);
                v4f64 xi;
                const bool ok = factor_invert_tile(a, xi, dli_n, lane, aug_tile == k + 1 ? aug_c : -1);
                if (!ok && lane == 0) S.flag[0] = 1;
                __hip_atomic_store(&S.flag[3], k + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                TRACE(1);
                store_col_ssq(xi, S.tr2 + 16 * (k + 1), rows_valid(k + 1), lane);
                lds_tile_store(diag + (size_t)(k + 1) * tile_doubles, lane, xi);
            }
            // ---- W_kk = X_kk, the last tile of row k of W, into the buffer of the next step ----
            {
                const v4f64 wkk = lds_tile(diag + (size_t)k * tile_doubles, lane);
                lds_tile_store(bufn + (size_t)k * tile_doubles, lane, wkk);
                if (k == aug_tile) put_m(wkk, k);
            }
            TRACE(2);
            // (the slots are the workers': nothing of them survives this branch, so the chain may have their registers)
#pragma unroll
            for (int s = 0; s < kRRSlots; ++s) asm volatile("" : "=v"(T[s]));
        } else {
            // What this step asks of each slot, as bit masks (one scalar test per slot instead of four comparisons of its (I, J): a
            // lone wave issues an instruction every ~5 cycles, and 24 slots x ~30 scalar instructions was 1.5 us of every step).
            // A row I of this wave starting at slot st holds J = I - 1, .., 0:
            //   trailing update (J > k): slots st .. st + I - k - 2; the last of them is the tile of column k + 1
            //   inverse step K = k - 1 (J < k, I >= k >= 1): slots st + I - k .. st + I - 1; the first of them (J = k - 1) starts its sum
            unsigned mT = 0, mCol = 0, mInv = 0, mStart = 0, mFin = 0;
#pragma unroll
            for (int r = 0; r < kRRRows; ++r) {
                const int I = rI[r], st = rS[r];
                if (I > k + 1) {
                    mT |= ((1u << (I - k - 1)) - 1u) << st;
                    mCol |= 1u << (st + I - k - 2);
                }
                if (I >= k && k >= 1) {
                    const unsigned bits = ((1u << k) - 1u) << (st + I - k);
                    if (I == k) mFin |= bits;
                    else mInv |= bits;
                    mStart |= 1u << (st + I - k);
                }
            }
            double ro0 = 0.0, ro1 = 0.0, ro2 = 0.0, ro3 = 0.0;  // the operand a row shares: -L_Ik^T (trailing), L_{I,k-1}^T (inverse)
            // ---- row k of the inverse first: its owner hands the row over in LDS as it stands -- the sums S_kJ still one product short
            //      (positions J < k - 1 of the next step's buffer) and L_{k,k-1}^T, the tile that starts the last sum (position k of
            //      THIS step's buffer, which nothing else uses) -- and every wave takes tiles of it off the counter below: the last
            //      product S_kJ += L_{k,k-1} W_{k-1,J}, then W_kJ = -X_kk S_kJ.  (With the last products formed here, by the one
            //      wave that holds the row, the late steps were that wave's: 18 tiles, 5 us of a 6.5 us step at k = 18.)
            if (mFin != 0) {
#pragma unroll
                for (int s = 0; s < kRRSlots; ++s) {
                    if (mFin & (1u << s)) {
                        RR_KEEP_BRANCH;
                        const int J = slot_e(s) & 255;
                        if (mStart & (1u << s)) lds_tile_store(bufk + (size_t)k * tile_doubles, lane, T[s]);
                        else lds_tile_store(bufn + (size_t)J * tile_doubles, lane, T[s]);
                    }
                }
                __hip_atomic_store(&S.flag[4], k + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            // ---- one pass over the slots: the trailing update with panel k; step K = k - 1 of the inverse for the rows below k ----
            // (both kinds take ONE operand tile from position J of the step's buffer: read one slot ahead, see rr_lds_issue)
            if ((mT | mInv) != 0) {
                RR_KEEP_BRANCH;
                int curI = -1;
                const unsigned lbase = (unsigned)reinterpret_cast<unsigned long long>(bufk) + (unsigned)lane * 16u;
                auto rowbase = [&](int t) __attribute__((always_inline)) {  // the base of the row that starts at slot t (c = J_t + t)
                    return lbase + (unsigned)(((int)(slot_e(t) & 31u) + t - 24) * 2048);
                };
                unsigned curbase = lbase;
                v2f64 oLo[2], oHi[2];
                unsigned mRowK = mRow;  // (opaque per step: its 25 bit tests, hoisted out of the step loop, were 25 spilled masks)
                asm volatile("" : "+s"(mRowK));
                if (mRowK & 1u) {
                    RR_KEEP_BRANCH;
                    curbase = rowbase(0);
                }
                rr_lds_issue_imm<24 * 2048>(oLo[0], oHi[0], curbase);
                rr_for_slots([&](auto sc) __attribute__((always_inline)) {
                    constexpr int s = decltype(sc)::value;
                    if constexpr (s + 1 < kRRSlots) {
                        if (mRowK & (1u << (s + 1))) {
                            RR_KEEP_BRANCH;
                            curbase = rowbase(s + 1);
                        }
                        rr_lds_issue_imm<(24 - (s + 1)) * 2048>(oLo[(s + 1) & 1], oHi[(s + 1) & 1], curbase);
                        RR_LDS_WAIT(2);
                    } else {
                        RR_LDS_WAIT(0);
                    }
                    if (mT & (1u << s)) {  // T_IJ^T -= L_Jk L_Ik^T  (the sign on the row's operand: once per row)
                        RR_KEEP_BRANCH;
                        const int I = slot_e(s) >> 8;
                        if (I != curI) {
                            RR_KEEP_BRANCH;
                            const v4f64 t = lds_tile(bufk + (size_t)I * tile_doubles, lane);
                            ro0 = -t[0], ro1 = -t[1], ro2 = -t[2], ro3 = -t[3];
                            curI = I;
                        }
                        rr_mfma4_bp(T[s], oLo[s & 1], oHi[s & 1], ro0, ro1, ro2, ro3);
                    }
                    if (mInv & (1u << s)) {  // S_IJ += L_{I,k-1} W_{k-1,J}
                        RR_KEEP_BRANCH;
                        if (mStart & (1u << s)) {
                            RR_KEEP_BRANCH;
                            ro0 = T[s][0], ro1 = T[s][1], ro2 = T[s][2], ro3 = T[s][3];
                            asm volatile("" : "+v"(ro0), "+v"(ro1), "+v"(ro2), "+v"(ro3));  // (a copy: the slot starts its sum from zero)
                            curI = -1;
                            T[s] = v4f64{0.0, 0.0, 0.0, 0.0};
                        }
                        rr_mfma4_ap(T[s], ro0, ro1, ro2, ro3, oLo[s & 1], oHi[s & 1]);
                    }
                }, std::make_integer_sequence<int, kRRSlots>{});
            }
            RR_MFMA_SETTLE;  // (the column tiles below read their slots as operands)
            TRACE(2);
            // ---- column k + 1: L_{I,k+1}^T = X_{k+1,k+1} T_{I,k+1}^T into the slot and into panel k + 1 ----
            // (in a pass of its own: inside the pass above -- behind the tile's last product -- the step took 0.3 us longer)
            if (m > 1 && mCol != 0) {
                int spins = 0;
                while (__hip_atomic_load(&S.flag[3], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < k + 1) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1 << 22)) {  // (a stuck flag must not hang the device)
                        if (lane == 0) S.flag[0] = 1;
                        break;
                    }
                }
                TRACE(3);
                Frag fx;
#pragma unroll
                for (int q = 0; q < 4; ++q) fx.v[q] = dli_n[cl * PS + 4 * q + rg];
#pragma unroll
                for (int s = 0; s < kRRSlots; ++s) {
                    if (mCol & (1u << s)) {
                        RR_KEEP_BRANCH;
                        const int I = slot_e(s) >> 8;
                        Frag ft;
#pragma unroll
                        for (int q = 0; q < 4; ++q) ft.v[q] = T[s][q];
                        const v4f64 z4 = {0.0, 0.0, 0.0, 0.0};
                        T[s] = mfma4(fx, ft, z4, false);
                        lds_tile_store(bufn + (size_t)I * tile_doubles, lane, T[s]);
                    }
                }
            }
            TRACE(4);
        }
        // ---- work that lives in LDS and that ANY wave may take, off a counter (the specialist once its chain is done, a worker when
        //      its own slots are): the diagonal tiles I > k + 1 take their product of this step, T_II^T -= L_Ik L_Ik^T; then row k of
        //      W, W_kJ = -X_kk S_kJ in place, Tr2 and m on the way
        {
            const int nd = m > 1 ? m - 1 : 0, nitems = nd + k;
            int *const ctr = S.flag + 5 + (k & 1);
            bool sums_in = false;
            for (;;) {
                int it = 0;
                if (lane == 0) it = atomicAdd(ctr, 1);
                it = __builtin_amdgcn_readfirstlane(it);
                if (it >= nitems) break;
                if (it < nd) {
                    const int I = k + 2 + it;
                    v4f64 d = lds_tile(diag + (size_t)I * tile_doubles, lane);
                    const v4f64 pa = lds_tile(bufk + (size_t)I * tile_doubles, lane);
#pragma unroll
                    for (int q = 0; q < 4; ++q) d = __builtin_amdgcn_mfma_f64_16x16x4f64(-pa[q], pa[q], d, 0, 0, 0);
                    lds_tile_store(diag + (size_t)I * tile_doubles, lane, d);
                    continue;
                }
                const int J = it - nd;
                if (!sums_in) {
                    int spins = 0;
                    while (__hip_atomic_load(&S.flag[4], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < k + 1) {
                        __builtin_amdgcn_s_sleep(1);
                        if (++spins > (1 << 22)) {
                            if (lane == 0) S.flag[0] = 1;
                            break;
                        }
                    }
                    sums_in = true;
                }
                Frag fs, fw;
                v4f64 sv = {0.0, 0.0, 0.0, 0.0};
                if (J < k - 1) sv = lds_tile(bufn + (size_t)J * tile_doubles, lane);
                {   // the last product of the sum: L_{k,k-1} W_{k-1,J}
                    const v4f64 la = lds_tile(bufk + (size_t)k * tile_doubles, lane), bw = lds_tile(bufk + (size_t)J * tile_doubles, lane);
#pragma unroll
                    for (int q = 0; q < 4; ++q) sv = __builtin_amdgcn_mfma_f64_16x16x4f64(la[q], bw[q], sv, 0, 0, 0);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) fs.v[q] = sv[q];
#pragma unroll
                for (int q = 0; q < 4; ++q) fw.v[q] = dli_k[cl * PS + 4 * q + rg];
                v4f64 w = {0.0, 0.0, 0.0, 0.0};
                w = mfma4(fw, fs, w, true);
                lds_tile_store(bufn + (size_t)J * tile_doubles, lane, w);
                double ssq = 0.0;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (16 * k + rg + 4 * r < N) ssq = fma(w[r], w[r], ssq);
                ssq += __shfl_xor(ssq, 16);
                ssq += __shfl_xor(ssq, 32);
                if (rg == 0) S.tr2[16 * J + cl] += ssq;  // (column J's sum over the rows in ascending order: cs_JJ came first)
                if (k == aug_tile) put_m(w, J);
            }
        }
        __syncthreads();
    }
#undef RR_E
    if (S.flag[0]) return false;
    return true;
}
#undef RR_KEEP_BRANCH
#endif  // FIT_LOOP_RR

// ---- (T + I) tau = rhs by a wave scan: band_scan.h ----
template <int WIDE> constexpr int scan_rows() { return WIDE == 2 ? 16 : (WIDE ? 10 : 6); }  // rows per lane: 64 * 6 = 384, 64 * 10 = 640, 64 * 16 = 1024 >= NP
using bandscan::scan_solve;
using bandscan::scan_tables;

template <int WIDE, int CLM>
__global__ __launch_bounds__(KT) void fit_loop_kernel(FitLoopParams P) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ int s_fit;
    constexpr bool CL = CLM == 1 || CLM == 2;
    constexpr bool LL = CLM == 3;  // left-looking solve on one workgroup (solve_posterior_ll), N <= 335
    constexpr bool DF = CLM == 4 || CLM == 5;  // deferred trailing update on one workgroup (solve_posterior<0, 4 | 5>), N <= 319
    constexpr bool RR = CLM == 6;  // the matrix resident in registers (solve_posterior_rr: 512 threads, fit_loop_rr.hip), N <= 303
    static_assert(!(DF && WIDE), "the deferred update keeps three panels in LDS");
    static_assert(!(RR && WIDE), "24 tiles per wave: N <= 303");
    static_assert(!(LL && WIDE), "the left-looking solve keeps two rows of L in LDS: N <= 335");
    // cluster mode: workgroup b sits on XCD b & 7 (ids go round the XCDs) as the (b >> 3)-th of the launch there; the members of
    // a fit are `cluster` consecutive ones of ONE XCD: fit (i / cluster) * 8 + x of the launch, member i % cluster
    int launch_index = blockIdx.x, member = 0;
    if constexpr (CL) {
        const int x = (blockIdx.x - P.cluster_xcd0) & 7, i = blockIdx.x >> 3;
        launch_index = (i / P.cluster) * 8 + x;
        member = i % P.cluster;
        if (launch_index >= P.nfits) return;
    }
    const FitLoopParams P0 = P;
    __shared__ long long s_clk[2];  // (fh_ctx_loop_clocks: shader clock and wall clock at the start of a fit)
    // batched launch: the workgroups pull fit indices from a counter (fits of a sweep differ ~20x in iteration count)
    for (;;) {
    P = P0;
    if (P.batch) {
        if (threadIdx.x == 0) s_fit = atomicAdd(P.batch_counter, 1);
        __syncthreads();
        const int f = s_fit;
        __syncthreads();
        if (f >= P.batch) return;
        const size_t PP = (size_t)P.NP * P.NP;
        const size_t slot = blockIdx.x;  // work buffers belong to the workgroup, outputs to the fit
        P.alpha = P.batch_alpha[f];
        P.p0 = P.batch_p0[f];
        P.band_lu += (size_t)f * 5 * P.N;
        P.C += slot * PP;
        P.W += slot * PP;
        P.WdT += slot * (size_t)P.NP * 16;
        P.cs += slot * fh_k2_cs_doubles(P.NP);
        P.mu_out += (size_t)f * P.N;
        P.p_out += (size_t)f * P.N;
        P.result += 2 * f;
        if (P.resume) P.resume += (size_t)f * (2 * P.N + 1);
    }
    if (P.slot_stride) {
        // (constant indices only: a dynamic index into the by-value parameter struct forces the WHOLE struct into scratch
        // memory, every later P.field a scratch load and the prologue 256 bytes of scratch stores)
        const int g = launch_index >> 2;  // (uniform selects between constant indices; four 16-bit slot ids to a word)
        unsigned long long wsel = P.slot_words[0];
#define FH_SLOT_WORD(i) if (g == (i)) wsel = P.slot_words[i];
        FH_SLOT_WORD(1) FH_SLOT_WORD(2) FH_SLOT_WORD(3) FH_SLOT_WORD(4) FH_SLOT_WORD(5) FH_SLOT_WORD(6) FH_SLOT_WORD(7)
        FH_SLOT_WORD(8) FH_SLOT_WORD(9) FH_SLOT_WORD(10) FH_SLOT_WORD(11) FH_SLOT_WORD(12) FH_SLOT_WORD(13) FH_SLOT_WORD(14)
        FH_SLOT_WORD(15) FH_SLOT_WORD(16) FH_SLOT_WORD(17) FH_SLOT_WORD(18) FH_SLOT_WORD(19) FH_SLOT_WORD(20) FH_SLOT_WORD(21)
        FH_SLOT_WORD(22) FH_SLOT_WORD(23) FH_SLOT_WORD(24) FH_SLOT_WORD(25) FH_SLOT_WORD(26) FH_SLOT_WORD(27) FH_SLOT_WORD(28)
        FH_SLOT_WORD(29) FH_SLOT_WORD(30) FH_SLOT_WORD(31)
#undef FH_SLOT_WORD
        const int sl = (int)((wsel >> (16 * (launch_index & 3))) & 0xffffull);
        const size_t off = (size_t)sl * P.slot_stride;
        P.A += off;
        P.bq += off;
        P.band_lu += off;
        P.C += off;
        P.W += off;
        P.WdT += off;
        P.cs += off;
        P.mu_out += off;
        P.p_out += off;
        P.result += 2 * sl;
        if (P.out_host) {
            P.out_host += (size_t)sl * 2 * P.N;
            P.result_host += 2 * sl;
        }
        P.alpha = P.band_lu[5 * P.N];  // per-fit hyper-parameters travel behind the slot's band LU
        P.p0 = P.band_lu[5 * P.N + 1];
        P.resume = P.mode == FIT_MODE_RESUME ? P.band_lu + 5 * P.N + 2 : nullptr;  // (... and the state of a paused fit behind them)
    }
    const int N = P.N, NP = P.NP;
    const int tid = threadIdx.x;
    if (P.clk_out && tid == 0) {
        s_clk[0] = clock64();
        s_clk[1] = wall_clock64();
    }
    int *const ctl = CL ? clu::ctl_of(P) : nullptr;
    if constexpr (CL) {
        if (member > 0) {  // a helper workgroup: block columns of the inverse (clu::inverse_wave) or trailing tiles (clu::trailing_wave)
            if (P.cluster_break & 1) return;
            if (tid == 0) {
                __hip_atomic_fetch_or(ctl + clu::XCC, 1 << clu::xcc_id(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                clu::add(ctl + clu::IN, 1);
            }
            const int hwave = __builtin_amdgcn_readfirstlane(tid >> 6);
            const int hi = clu::inv_helpers(P), ht = clu::trail_helpers(P);
            if (member <= hi) {
                // (every wave at most one block column: the columns' second waves, see inverse_wave_paired)
                if (hi * NW >= NP / 16) clu::inverse_wave_paired(P, ctl, hi, member - 1, hwave, tid & 63, smem);
                else clu::inverse_wave<2>(P, ctl, hwave * hi + (member - 1), hi * NW, tid & 63);
            }
            else clu::trailing_wave(P, ctl, hwave * ht + (member - 1 - hi), ht * NW, tid & 63, smem, hwave);
            __syncthreads();
            if (tid == 0) clu::add(ctl + clu::IN, -1);
            return;
        }
    }
    Smem S;
    S.pan = smem;
    S.dli = RR ? smem : S.pan + (size_t)(DF ? 3 : npanels<WIDE>()) * NP * PS;  // (RR: the two buffers of nb packed tiles come LAST, see below)
    // (XWIDE: the vectors live behind the band factors and scan tables in global memory -- the W buffer, or the cs buffer in
    //  cluster mode --, only the panel, the inverse of the diagonal tile and the flags in LDS)
    double *const gscratch = CL ? P.cs : P.W;
    S.p = WIDE == 2 ? gscratch + 6 * NP + 2 * 6 * 4 * 64 : S.dli + (DF ? 3 : 2) * 16 * PS;
    S.pold = S.p + NP;
    S.m = S.pold + NP;
    S.y = S.m + NP;
    S.tr2 = S.y + NP;
    S.rhs = S.tr2 + NP;
    S.b = S.rhs + NP;
    S.red = S.b + NP;  // NP (block reductions use NW entries)
    if constexpr (WIDE) {
        // the band factors and the scan tables (48 KB at NP = 512) in global memory -- the workgroup's W buffer, unused since
        // W lives in the dead tiles of C --: wave 0 reads them once per pass (band_scan.h takes either address space), and
        // the one panel, the vectors and the tile table of NP = 512 fit the LDS beside nothing else
        // (cluster mode: the helpers of the inverse write W there; the cs buffer, (NP / 16)^2 x 16 doubles, is free instead)
        S.band = gscratch;
        S.scanQ = S.band + 6 * NP;
        S.rec = reinterpret_cast<uint4 *>(WIDE == 2 ? S.dli + 2 * 16 * PS : S.red + NP);
    } else if constexpr (LL) {
        // (the LDS region of the bands, scan tables and tile table holds the diagonal tiles: solve_posterior_ll)
        S.band = P.W;
        S.scanQ = S.band + 6 * NP;
        S.rec = reinterpret_cast<uint4 *>(S.red + NP + 6 * NP + 2 * 6 * 4 * 64);
    } else if constexpr (DF || RR) {
        // (three panels in LDS -- RR: two buffers of tiles and the diagonal tiles --: the band factors and scan tables in the W buffer, which the one-workgroup solve does not use --
        //  W lives in the dead tiles of C --; wave 0 reads them once per pass, as in the wide instantiations)
        S.band = P.W;
        S.scanQ = S.band + 6 * NP;
        S.rec = reinterpret_cast<uint4 *>(S.red + NP);  // two filtered tile tables of 128 entries (see below)
    } else {
        S.band = S.red + NP;
        S.scanQ = S.band + 6 * NP;  // [2 directions][6 levels][4 entries][64 lanes]
        S.rec = reinterpret_cast<uint4 *>(S.scanQ + 2 * 6 * 4 * 64);  // (16-byte aligned: every region before it is an even number of doubles)
    }
    if constexpr (RR) S.rec = reinterpret_cast<uint4 *>(S.red + NP);  // (no tile table: the flags follow the vectors)
    S.flag = reinterpret_cast<int *>(S.rec + (RR ? 0 : (DF ? 256 : max_tiles<WIDE>())));  // [0] not positive definite, [1] column counter of the inverse row
    S.hand = RR ? reinterpret_cast<double *>(S.flag + 8 + 48) : (!WIDE && !DF && NP <= kHandMaxNP) ? reinterpret_cast<double *>(S.flag + 8) : nullptr;  // (32 bytes of flags; 16-byte aligned)
    S.dcnt = S.flag + 8;  // (deferred mode: 2 x 24 ints behind the flags)
#ifdef FIT_LOOP_RR
    if constexpr (RR) {
        S.dcnt = reinterpret_cast<int *>(S.hand + (size_t)(NP / 16) * 256);  // (register-resident mode: the slot tables, behind the diagonal tiles)
        rr_tables(S.dcnt, NP / 16, tid);
        size_t off = (size_t)(reinterpret_cast<double *>(S.dcnt + 8 * kRRTabStride) - smem);  // (doubles: every region is a multiple of 16 bytes)
        if (off < (size_t)kRRBufferMinOffset / 8) off = (size_t)kRRBufferMinOffset / 8;
        S.pan = smem + off;
    }
#endif
    if constexpr (DF) {
        // Deferred mode: at a step of parity par the tiles (i, j) relative to block (k + 1, k + 1), 1 <= j <= i, that are touched
        // are those with (i + j + par) even -- they take panels k - 1 and k -- and, bit 2 of x set, the other tiles of column
        // j = 1 (next step's column tiles), which take panel k only.  Row-wise enumeration as in the full table: the tiles of
        // step k are the entries with i <= nb - k - 2, a prefix; dcnt[par][i] = entries with row <= i.  Built by two threads,
        // once per fit (~120 entries each).
        if (tid < 2) {
            const int par = tid, nbk = NP / 16;
            int n = 0;
            S.dcnt[par * 24] = 0;
            for (int i = 1; i <= 20; ++i) {
                for (int j = 1; j <= i; ++j) {
                    const bool own = ((i + j + par) & 1) == 0;
                    if (!own && j != 1) continue;
                    S.rec[par * 128 + n++] = make_uint4((unsigned)((j * nbk + i) * 2048) | (i == j ? 2u : 0u) | (own ? 0u : 4u),
                                                       (unsigned)(j * 16 * PS * 8), (unsigned)(i * 16 * PS * 8),
                                                       (unsigned)((i * nbk + j) * 2048) | (unsigned)i);
                }
                S.dcnt[par * 24 + i] = n;
            }
        }
    }
    for (int e = tid; e < ((LL || DF || RR) ? 0 : max_tiles<WIDE>()); e += KT) {  // tile e = (i - 1) i / 2 + (j - 1), 1 <= j <= i
        int i = (int)((sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
        while ((i + 1) * (i + 2) / 2 <= e) ++i;
        while (i * (i + 1) / 2 > e) --i;
        const int j = e - i * (i + 1) / 2 + 1;
        ++i;
        S.rec[e] = make_uint4((unsigned)((j * (NP / 16) + i) * 2048) | (i == j ? 2u : 0u), (unsigned)(j * 16 * PS * 8),
                              (unsigned)(i * 16 * PS * 8), (unsigned)((i * (NP / 16) + j) * 2048) | (unsigned)i);
    }
    __shared__ int s_ctl[4];  // [0] stop, [1] status, [2] cluster assembled
    if constexpr (CL) {
        // the helpers have 3 ms to show up (beside a binning pass their workgroups queue behind its thousands: 200 us, the first
        // value, made a quarter of a pipeline's clusters fall back), all on this XCD; otherwise the fit ends with
        // FIT_STATUS_CLUSTER (the host runs it on one CU) and helpers that arrive later find the word negative and leave
        constexpr long long kAssembleUs = 3000;
        if (tid == 0) {
            const long long t0 = wall_clock64();
            int ok = 1;
            while (clu::ld(ctl + clu::IN) < P.cluster - 1) {
                __builtin_amdgcn_s_sleep(2);
                if (wall_clock64() - t0 > kAssembleUs * clu::kTicksPerUs) {
                    ok = 0;
                    break;
                }
            }
            const int xccs = clu::ld(ctl + clu::XCC) | (1 << clu::xcc_id());
            if (xccs & (xccs - 1)) ok = 0;
            if (!ok) clu::st(ctl + clu::PROG, -1);
            s_ctl[2] = ok;
        }
        __syncthreads();
        if (!s_ctl[2]) {
            if (tid == 0) {
                P.result[0] = 0;
                P.result[1] = FIT_STATUS_CLUSTER;
                if (P.result_host) {
                    P.result_host[0] = 0;
                    P.result_host[1] = FIT_STATUS_CLUSTER;
                }
            }
            return;
        }
    }

    if (P.band_lu)
        for (int i = tid; i < 5 * NP; i += KT) {  // five bands of NP entries (padding: unit pivots, zero bands) + reciprocal pivots
            const int bnd = i / NP, c = i - bnd * NP;
            double v = (c < N) ? P.band_lu[bnd * N + c] : (bnd == 2 ? 1.0 : 0.0);
            if (c == 0 && bnd < 2) v = 0.0;  // no sub-diagonal entries in row 0
            S.band[i] = v;
            if (bnd == 2) S.band[5 * NP + c] = 1.0 / v;
        }
    if (P.band_lu) {
        __syncthreads();
        if (tid < 64) scan_tables<scan_rows<WIDE>()>(S.band, NP, S.scanQ, tid);
    }
    for (int i = tid; i < NP; i += KT) {
        S.b[i] = i < N ? P.bq[i] : 0.0;
        S.p[i] = i < N ? (P.p_init ? P.p_init[i] : 1.0) : 1.0;  // radial_fitters.py:744 (p = 1)
        S.pold[i] = 0.0;                                        // radial_fitters.py:768 (pi_old = 0)
        if (P.mode == FIT_MODE_RESUME && i < N) {               // a paused fit: its power spectrum and the one before
            S.p[i] = P.resume[i];
            S.pold[i] = P.resume[N + i];
        }
    }
    if (tid == 0) {
        s_ctl[0] = 0;
        s_ctl[1] = 0;
    }
    __syncthreads();
    int status = 0, count = P.mode == FIT_MODE_RESUME ? (int)P.resume[2 * N] : 0, nsolve = 0;
    // One call site for the posterior solve.  phase 0: p = 1 (radial_fitters.py:744-747); phase 1: power-law
    // guess (:749-752); phase 2: the loop of :769-785 (FIT_MODE_STEP: exactly one pass, FIT_MODE_SOLVE: none).
    int phase = (P.mode == FIT_MODE_FULL) ? 0 : 2;
    bool in_pass = P.mode == FIT_MODE_RESUME;  // (a paused fit stopped behind an update of p: its next solve ends a pass)
#ifdef FIT_LOOP_TIMING
    long long o_last = clock64();
#endif
    for (;;) {
        OSTAMP(14);
#ifdef FIT_LOOP_TIMING
        P.trace_on = (count == 5);
#endif
        bool solved;
        if constexpr (CL && !WIDE) solved = solve_posterior_cluster<CLM>(P, S, ++nsolve);
        else if constexpr (LL) {
            Smem SL = S;
            SL.band = smem + (S.red + NP - smem);  // the LDS region behind the vectors: the diagonal tiles
            ++nsolve;
            solved = solve_posterior_ll(P, SL);
        }
#ifdef FIT_LOOP_RR
        else if constexpr (RR) {
            ++nsolve;
            solved = solve_posterior_rr(P, S);
        }
#endif
        else solved = solve_posterior<WIDE, CLM>(P, S, ++nsolve);
        if (!solved) {
            status = (CL && S.flag[0] == 2) ? FIT_STATUS_CLUSTER : FIT_STATUS_NOT_SPD;
            break;
        }
        if (phase == 0) {
            // pI = max(DHT.transform(MAP)^2) * (q/q[0])^-2; transform(MAP) = pl_scale * m
            double best = -INFINITY;
            for (int i = tid; i < N; i += KT) {
                const double t = P.pl_scale * S.m[i];
                best = fmax(best, t * t);
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) best = fmax(best, __shfl_down(best, off));
            if ((tid & 63) == 0) S.red[tid >> 6] = best;
            __syncthreads();
            double pmax = S.red[0];
            for (int w = 1; w < NW; ++w) pmax = fmax(pmax, S.red[w]);
            __syncthreads();
            for (int i = tid; i < N; i += KT) S.p[i] = pmax * pow(P.q[i] / P.q[0], -2.0);
            __syncthreads();
            phase = 1;
            continue;
        }
        OSTAMP(13);
        phase = 2;
        if (in_pass) {
            if (P.diag_mu) {  // MAP of this pass: mu = Y^-1 m   (radial_fitters.py:783)
                // (the loads of four rows are issued together: a row at a time the wave waited for one L2 round trip per row,
                //  24 us of a 160 us pass with store_iteration_diagnostics on; the sums are formed in the same order)
                constexpr int RB = WIDE == 2 ? 1 : 4, CB = WIDE == 2 ? 16 : (WIDE ? 10 : 6);  // column chunks of 64: N < 336, <= 639, <= 1023
                const int ln = tid & 63;
                double ms[CB];
#pragma unroll
                for (int cc = 0; cc < CB; ++cc) ms[cc] = (ln + 64 * cc < N) ? S.m[ln + 64 * cc] : 0.0;
                for (int r0 = __builtin_amdgcn_readfirstlane(tid >> 6); r0 < N; r0 += RB * NW) {
                    double y[RB][CB];
#pragma unroll
                    for (int k = 0; k < RB; ++k) {
                        const double *yr = P.Yinv + (size_t)min(r0 + k * NW, N - 1) * N;
#pragma unroll
                        for (int cc = 0; cc < CB; ++cc) y[k][cc] = yr[min(ln + 64 * cc, N - 1)];
                    }
#pragma unroll
                    for (int k = 0; k < RB; ++k) {
                        const int r = r0 + k * NW;
                        double a = 0.0;
#pragma unroll
                        for (int cc = 0; cc < CB; ++cc)
                            if (ln + 64 * cc < N) a = fma(y[k][cc], ms[cc], a);
#pragma unroll
                        for (int off = 32; off >= 1; off >>= 1) a += __shfl_down(a, off);
                        if (ln == 0 && r < N) P.diag_mu[(size_t)count * N + r] = a;
                    }
                }
            }
            ++count;
            if (P.mode == FIT_MODE_STEP) break;
        }
        if (P.mode == FIT_MODE_SOLVE) break;
        int bad = 0;
        for (int i = tid; i < N; i += KT) bad |= !(fabs(S.p[i] - S.pold[i]) <= P.tol * S.p[i]);  // filter.py:181
        bad = __syncthreads_or(bad);
        if (!bad || count > P.max_iter) break;  // radial_fitters.py:769-770
        // beta and the right-hand side of (T + I) tau = beta + log p   (filter.py:172-175)
        for (int i = tid; i < NP; i += KT) {
            if (i < N) {
                const double pi = S.p[i], mi = S.m[i];
                const double beta = (P.p0 + 0.5 * (mi * mi + S.tr2[i])) / pi - (P.alpha - 1.0 + 0.5 * 1.0);
                S.rhs[i] = beta + log(pi);
                S.pold[i] = pi;
            } else {
                S.rhs[i] = 0.0;  // padding rows of the banded system
            }
        }
        __syncthreads();
        OSTAMP(14);
#ifndef K2_SERIAL_BAND
        if (tid < 64) {  // wave 0: both substitutions as wave scans (see scan_solve); LDS traffic of one wave only, in order
            // (the lane index goes through an opaque move: the LDS addresses of the scan are then formed HERE, a few integer
            // operations, instead of being hoisted out of the pass loop, kept alive across the factorisation and spilled)
            int t = tid;
            asm volatile("" : "+v"(t));
            scan_solve<scan_rows<WIDE>()>(S.band, NP, S.scanQ, S.rhs, 0, t);
            scan_solve<scan_rows<WIDE>()>(S.band, NP, S.scanQ, S.rhs, 1, t);
        }
#else
        if (tid == 0) {
            // Banded LU solve with the host-prepared factors (staged in LDS once per fit, padded to NP with unit pivots and
            // zero bands).  One thread, a chain of 2 N dependent steps: a step must cost its two dependent fmas (and the
            // three-operation division on the way back) and little else -- 16-byte LDS accesses for the operands and the
            // results of eight steps, no per-step predicate (the padding rows solve to 0), everything in registers.
            // (per step it was ~200 cycles, 50 us per pass: 14 scalar-width LDS instructions and an exec-mask update)
            const v2f64 *f1 = reinterpret_cast<const v2f64 *>(S.band), *f2 = f1 + NP / 2, *d0 = f2 + NP / 2, *u1 = d0 + NP / 2,
                        *u2 = u1 + NP / 2, *rd0 = u2 + NP / 2;
            v2f64 *rv = reinterpret_cast<v2f64 *>(S.rhs);
            double x1 = 0.0, x2 = 0.0;  // x_{i-1}, x_{i-2}  (f1[0] = f2[0] = 0: x_0 = rhs_0 exactly)
            for (int h = 0; h < NP / 2; h += 4) {
                v2f64 r[4], a1[4], a2[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    r[k] = rv[h + k];
                    a1[k] = f1[h + k];
                    a2[k] = f2[h + k];
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        double xi = r[k][e];
                        xi = fma(-a2[k][e], x2, xi);
                        xi = fma(-a1[k][e], x1, xi);
                        r[k][e] = xi;
                        x2 = x1;
                        x1 = xi;
                    }
                    rv[h + k] = r[k];
                }
            }
            double y1 = 0.0, y2 = 0.0;  // x_{i+1}, x_{i+2}
            for (int h = NP / 2 - 4; h >= 0; h -= 4) {
                v2f64 r[4], b1[4], b2[4], dd[4], rr[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    r[k] = rv[h + k];
                    b1[k] = u1[h + k];
                    b2[k] = u2[h + k];
                    dd[k] = d0[h + k];
                    rr[k] = rd0[h + k];
                }
#pragma unroll
                for (int k = 3; k >= 0; --k) {
#pragma unroll
                    for (int e = 1; e >= 0; --e) {
                        double t = r[k][e];
                        t = fma(-b1[k][e], y1, t);
                        t = fma(-b2[k][e], y2, t);
                        t = div_rn(t, dd[k][e], rr[k][e]);  // (the division itself: ~12 dependent operations per step)
                        r[k][e] = t;
                        y2 = y1;
                        y1 = t;
                    }
                    rv[h + k] = r[k];
                }
            }
        }
#endif
        OSTAMP(15);
        __syncthreads();
        int badp = 0;
        for (int i = tid; i < N; i += KT) {
            const double pn = exp(S.rhs[i]);  // filter.py:177
            S.p[i] = pn;
            badp |= !(pn > 0.0);              // statistical_models.py:689
            if (P.diag_p) P.diag_p[(size_t)count * N + i] = pn;
        }
        badp = __syncthreads_or(badp);
        if (badp) {
            status = FIT_STATUS_BAD_P;
            break;
        }
        in_pass = true;
        if (P.pass_cap > 0 && count >= P.pass_cap) {  // pause here: (p, p_old, count) is the whole state of the iteration
            status = FIT_STATUS_PAUSED;
            break;
        }
        if (P.pause_when_left > 0 && (count & 15) == 0) {  // ... or when this fit is one of the last few of its batch still running
            if (tid == 0)
                s_ctl[3] = __hip_atomic_load(P.batch_counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= P.batch &&
                           P.batch - __hip_atomic_load(P.done_counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= P.pause_when_left;
            __syncthreads();
            const int stop = s_ctl[3];
            __syncthreads();
            if (stop) {
                status = FIT_STATUS_PAUSED;
                break;
            }
        }
    }

    // outputs: mu = Y^-1 m, p, count, status (a paused fit: p_old in the place of mu)
    if (status == FIT_STATUS_PAUSED) {
        for (int i = tid; i < N; i += KT) {
            P.mu_out[i] = S.pold[i];
            if (P.out_host) P.out_host[i] = S.pold[i];
        }
    } else
    for (int r = __builtin_amdgcn_readfirstlane(tid >> 6); r < N; r += NW) {
        const double *yr = P.Yinv + (size_t)r * N;
        double a = 0.0;
        for (int c = tid & 63; c < N; c += 64) a = fma(yr[c], S.m[c], a);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) a += __shfl_down(a, off);
        if ((tid & 63) == 0) {
            P.mu_out[r] = a;
            if (P.out_host) P.out_host[r] = a;
        }
    }
    for (int i = tid; i < N; i += KT) {
        P.p_out[i] = S.p[i];
        if (P.out_host) P.out_host[N + i] = S.p[i];
    }
    if constexpr (CL) {
        // the helpers leave on a negative word; when the last of them has gone the control words go back to zero (the state the
        // next fit on these buffers expects).  A cluster that broke (FIT_STATUS_CLUSTER) is cleaned up by the host.
        if (tid == 0) {
            clu::st(ctl + clu::PROG, -1);
            if (status != FIT_STATUS_CLUSTER) {
                const long long t0 = wall_clock64();
                bool gone = true;
                while (clu::ld(ctl + clu::IN) > 0) {
                    __builtin_amdgcn_s_sleep(2);
                    if (wall_clock64() - t0 > 1000000 * clu::kTicksPerUs) {
                        gone = false;
                        break;
                    }
                }
                if (gone) {
                    for (int jj = 0; jj < NP / 16; ++jj) clu::st(ctl + clu::HCOL + jj, 0);
                    clu::st(ctl + clu::DONE, 0);
                    clu::st(ctl + clu::XCC, 0);
                    clu::st(ctl + clu::PROG, 0);
                } else {
                    status = FIT_STATUS_CLUSTER;
                }
            }
        }
    }
    if (tid == 0) {
        P.result[0] = count;
        P.result[1] = status;
        if (P.result_host) {
            P.result_host[0] = count;
            P.result_host[1] = status;
        }
        if (P.done_counter && status != FIT_STATUS_PAUSED) atomicAdd(P.done_counter, 1);
        if (P.clk_out) {
            atomicAdd(P.clk_out, (unsigned long long)(clock64() - s_clk[0]));
            atomicAdd(P.clk_out + 1, (unsigned long long)(wall_clock64() - s_clk[1]));
            atomicAdd(P.clk_out + 2, (unsigned long long)nsolve);
        }
    }
    if (!P.batch) return;
    __syncthreads();
    }  // next fit of the batch
}

// A <- (A + A^T)/2 on the leading N x N block, b in row/column N, zero elsewhere -- written as PACKED tiles (tile_chol.h): the
// fit loop is the only reader.
__global__ void symmetrize_pad_kernel(const double *Araw, const double *bq, int N, int NP, double *A) {
    const size_t total = (size_t)NP * NP;
    const int nb = NP / 16;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        // e enumerates the PACKED positions: tile, pair index h, lane, element of the pair
        const int tile = (int)(e >> 8), w = (int)(e & 255), h = w >> 7, ln = (w >> 1) & 63, e2 = w & 1;
        const int I = tile / nb, J = tile - I * nb;
        const int i = 16 * I + 4 * (2 * h + e2) + (ln >> 4), j = 16 * J + (ln & 15);
        double v = 0.0;
        if (i < N && j < N) v = 0.5 * (Araw[(size_t)i * N + j] + Araw[(size_t)j * N + i]);
        else if (i == N && j < N) v = bq[j];
        else if (j == N && i < N) v = bq[i];
        A[e] = v;
    }
}

}  // namespace

#ifdef FIT_LOOP_RR
// the register-resident instantiation only (fit_loop_rr.hip compiles this file with 512 threads per workgroup)
size_t fh_k2_loop_rr_smem_bytes(int NP) {
    // L_kk^-1 (two), the vectors, the flags, the nb diagonal tiles, the slot tables -- at least 48 KB --, then two buffers of nb
    // packed tiles (the band factors and scan tables: the W buffer)
    size_t front = sizeof(double) * (size_t)((NP / 16) * 256 + 2 * 16 * PS + 8 * NP) + 32 + 4 * 48 + 4 * 8 * kRRTabStride;
    if (front < (size_t)kRRBufferMinOffset) front = kRRBufferMinOffset;
    return front + sizeof(double) * (size_t)(2 * (NP / 16) * 256);
}
hipError_t fh_k2_launch_loop_rr(const FitLoopParams &P, int blocks, hipStream_t s) {
    if (P.NP / 16 > kRRMaxNB || P.cluster > 1) return hipErrorInvalidValue;
    const size_t smem = fh_k2_loop_rr_smem_bytes(P.NP);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&fit_loop_kernel<0, 6>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((fit_loop_kernel<0, 6>), dim3(blocks), dim3(KT), smem, s, P);
    return hipGetLastError();
}
#else
static bool loop_is_wide(int NP) { return NP >= kWideMinNP; }
static bool loop_is_xwide(int NP) { return NP >= kXWideMinNP; }
int fh_k2_loop_max_np() { return kXWideMaxNP; }
// deferred mode (fit_loop_kernel<0, 4>): three panels, L_kk^-1 (two), the vectors, two tile tables, flags and counts
static size_t loop_smem_bytes_deferred(int NP) {
    return sizeof(double) * (size_t)(3 * NP * PS + 3 * 16 * PS + 8 * NP) + 16 * 256 + 32 + 4 * 48;
}
constexpr int kDeferMaxNP = 320;
constexpr int kRegResidentMaxNP = 304;  // 19 block rows: 25 tiles per worker wave
constexpr int kRegResidentMinLoops = 128;  // (161 888 of the 163 840 bytes at NP = 320)
size_t fh_k2_loop_smem_bytes(int NP) {
    if (loop_is_xwide(NP)) return sizeof(double) * (size_t)(NP * PS + 2 * 16 * PS) + 64;  // the panel, L_kk^-1 (two), the flags
    const bool wide = loop_is_wide(NP);
    return sizeof(double) * (size_t)((wide ? 1 : 2) * NP * PS + 2 * 16 * PS + 7 * NP + NP + (wide ? 0 : 6 * NP + 2 * 6 * 4 * 64)) +
           16 * (size_t)(wide ? kMaxTilesWide : kMaxTiles) + 32 + ((wide || NP > kHandMaxNP) ? 0 : 4 * 2048);  // (+ the hand-over tiles of the cluster mode)
}

// Development experiment (FRANK_AMD_K2_DUMMY=<milliseconds>): a workgroup that occupies a CU exactly like the fit loop
// (threads, LDS) but only spins -- separates what co-running fit loops cost bin_gram through the CU they hold from what
// they cost through the memory system.  Results are garbage.
__global__ __launch_bounds__(KT) void fit_loop_dummy_kernel(long long cycles, int *result) {
    extern __shared__ double dsm[];
    const long long t0 = clock64();
    double acc = 0.0;
    while (clock64() - t0 < cycles) {
        acc += dsm[threadIdx.x];
        __builtin_amdgcn_s_sleep(32);
    }
    if (threadIdx.x == 0) {
        result[0] = 1 + (acc == 12345.678);
        result[1] = 0;
    }
}

// one launch of `blocks` workgroups of the instantiation that covers P.NP
static hipError_t launch_loop(const FitLoopParams &P, int blocks, hipStream_t s) {
    size_t smem = fh_k2_loop_smem_bytes(P.NP);
    if (P.NP > kXWideMaxNP) return hipErrorInvalidValue;
    if (const char *d = getenv("FRANK_AMD_K2_DUMMY")) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&fit_loop_dummy_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(fit_loop_dummy_kernel, dim3(blocks), dim3(KT), smem, s, (long long)(atof(d) * 2.4e6), P.result);
        return hipGetLastError();
    }
    // the attribute per launch (cheap): it is per device, and contexts on several devices share this code
    auto go = [&](auto kernel, const FitLoopParams &Q, int grid) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(KT), smem, s, Q);
        return hipGetLastError();
    };
    if (P.cluster > 1) {
        // cluster mode: `blocks` fits, P.cluster workgroups each, the members of a fit on one XCD (see the kernel)
        if (P.cluster > FIT_CLUSTER_MAX || P.batch) return hipErrorInvalidValue;
        FitLoopParams Q = P;
        Q.nfits = blocks;
        // helpers of the inverse / of the trailing update: two of the first kind give every block column a wave of its own up
        // to N = 383 (a wave takes two columns at most: nb <= 24 helpers' waves x 2); the others keep the trailing tiles right
        // of the band (not for the wide systems: their tiles do not fit the registers of a few waves)
        const int nbk = P.NP / 16, h = P.cluster - 1;
        int inv = h >= 4 ? h - 2 : (h >= 2 ? 2 : 1);  // (two helpers of the trailing update are enough from five workgroups on)
        if (loop_is_wide(P.NP)) inv = h;
        if (const char *ie = getenv("FRANK_AMD_K2_CLUSTER_INV")) inv = atoi(ie);  // development
        inv = inv < 1 ? 1 : (inv > h ? h : inv);
        if (nbk > 2 * inv * NW) return hipErrorInvalidValue;
        Q.cluster_inv = inv;
        if (const char *be = getenv("FRANK_AMD_K2_CLUSTER_BREAK")) Q.cluster_break = atoi(be);  // tests: the fall-back to one CU
        const int grid = 8 * P.cluster * ((blocks + 7) / 8);
        // (CLM = 2, the two waves on the chain's SIMD sitting the trailing update out, was measured at every stage of this mode:
        //  never faster -- 99.1 against 97.6 us per pass at the end -- and is not instantiated)
        if (loop_is_xwide(P.NP)) return go(&fit_loop_kernel<2, 1>, Q, grid);
        if (loop_is_wide(P.NP)) return go(&fit_loop_kernel<1, 1>, Q, grid);
        return go(&fit_loop_kernel<0, 1>, Q, grid);
    }
    if (loop_is_xwide(P.NP)) return go(&fit_loop_kernel<2, 0>, P, blocks);
    if (loop_is_wide(P.NP)) return go(&fit_loop_kernel<1, 0>, P, blocks);
    // FRANK_AMD_K2_LL=1: the left-looking solve (solve_posterior_ll).  Measured (round 4) and NOT the default: the same bits with
    // 6.4 instead of 10.9 MB through the L1 per pass, but 156-163 us per pass alone against 136 -- its work piles up in the late
    // steps (171 tile products in step 18 against a 4.4 us chain in the first ones) and the chain's LDS shuffles queue behind the
    // operand reads (5.3 us per diagonal tile) -- and with the device full it only draws level (200 loops: 227 against 233 us,
    // steady state 1 078 against 1 056 fits/s).
    const char *le = getenv("FRANK_AMD_K2_LL");
    if (le && atoi(le) != 0 && P.NP >= 64) return go(&fit_loop_kernel<0, 3>, P, blocks);  // (its tables live in the W buffer too: see below)
    // The deferred trailing update (round 5; solve_posterior, CLM = 4): the same bits with half the loads and stores of the
    // trailing update -- what a pass moves beyond the L2 is what bounds a loaded device.  FRANK_AMD_K2_DEFER=0 keeps the
    // kernel of rounds 2-4 (read at every launch: the tests compare the two inside one process).
    // The matrix resident in registers (round 5; solve_posterior_rr, fit_loop_rr.hip): the same bits, and per pass only A is read.
    // A loop ALONE takes 148 us per pass in this form against 135 -- eight waves, and a wave cannot issue its matrix instructions
    // faster than one per 64 cycles (tools/microbench/mfma_f64_bench.hip) --, level at ~64 resident loops (150 us both), and from
    // there on it is the faster one: 161 against 196 us with 256 resident.  (A run that fills the device only to drain at once pays
    // for the slower lone passes of its last fits: 100 fits in flight 689 against 727 fits/s -- hence 128 and not 64.)  So it runs where the device is loaded: this launch and
    // the loops resident beside it (P.loaded, the host's count) make kRegResidentMinLoops.  FRANK_AMD_K2_RR = 0 / 1 forces the choice
    // (read at every launch: the tests compare the forms inside one process).
    bool rr = blocks + P.loaded >= kRegResidentMinLoops;
    if (const char *re = getenv("FRANK_AMD_K2_RR")) rr = atoi(re) != 0;
    if (rr && P.NP >= 64 && P.NP <= kRegResidentMaxNP) return fh_k2_launch_loop_rr(P, blocks, s);
    const char *de = getenv("FRANK_AMD_K2_DEFER");
    // (from NP = 64 on: the band factors and scan tables of this form, 6 NP + 3 072 doubles, live in the fit's W buffer of NP^2
    //  doubles -- at NP = 48 they overran it by a third, which the suite only noticed as a memory fault when a small LogNormal
    //  fit's seed loop followed a test that had left unmapped pages behind the buffer)
    if (P.NP >= 64 && P.NP <= kDeferMaxNP && !(de && atoi(de) == 0)) {
        smem = loop_smem_bytes_deferred(P.NP);
        // ... and the rows of the inverse in pairs (CLM = 5: one load of W per two products) where the device is FULL: with 256
        // loops resident a pass takes 186 us against 217 without the pairs (and 264 for the kernel of rounds 2-4), but alone
        // 156 against 135 (133) -- half as many, twice as long chains per step leave waves idle in the late steps --, level
        // at ~190 loops.  P.loaded: fit loops that will be resident beside this launch's (the host's count); the same bits
        // either way.  FRANK_AMD_K2_PAIR = 0 / 1 forces the choice.
        bool pair = blocks + P.loaded >= 192;
        if (const char *pe = getenv("FRANK_AMD_K2_PAIR")) pair = atoi(pe) != 0;
        if (pair) return go(&fit_loop_kernel<0, 5>, P, blocks);
        return go(&fit_loop_kernel<0, 4>, P, blocks);
    }
    return go(&fit_loop_kernel<0, 0>, P, blocks);
}

hipError_t fh_k2_launch_loop_batched(const FitLoopParams &P, int batch, hipStream_t s) { return launch_loop(P, batch, s); }
hipError_t fh_k2_launch_loop(const FitLoopParams &P, hipStream_t s) { return launch_loop(P, 1, s); }
hipError_t fh_k2_launch_loop_slots(const FitLoopParams &P, int nslots, hipStream_t s) { return launch_loop(P, nslots, s); }
size_t fh_k2_exchange_doubles(int NP) { return (size_t)NP * 16; }  // >= 2 NP + the control words (clu::NCTL ints)

hipError_t fh_k2_launch_symmetrize(const double *Araw, const double *bq, int N, int NP, double *A, hipStream_t s) {
    hipLaunchKernelGGL(symmetrize_pad_kernel, dim3(128), dim3(256), 0, s, Araw, bq, N, NP, A);
    return hipGetLastError();
}
#endif  // FIT_LOOP_RR
