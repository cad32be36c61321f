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
// 1024-thread workgroup (16 waves, s_barrier only), matrices stay in L2, the 16x16 tile products run on
// v_mfma_f64_16x16x4_f64.  One launch per fit, no host round trip; fits are independent, so many of these
// kernels run concurrently (one CU each) beside the bin_gram kernel of the next fit.
#include <hip/hip_runtime.h>

#include "kernels.h"

typedef double v4f64 __attribute__((ext_vector_type(4)));

namespace {

#ifdef FIT_LOOP_TIMING
#define TSTAMP(ph) do { if (threadIdx.x == 0) { long long now_ = clock64(); P.timing[ph] += now_ - t_last; t_last = now_; } } while (0)
#else
#define TSTAMP(ph) do { } while (0)
#endif

constexpr int KT = 1024;
constexpr int NW = KT / 64;
constexpr int PS = 17;  // LDS stride of the 16-wide panel rows (doubles)

__device__ __forceinline__ double bcast(double v, int lane) {  // wave-uniform broadcast of lane `lane`'s value
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

struct Smem {
    double *pan;   // panel / block-row staging, max(NP*PS, 16*(NP+1)) doubles
    double *lw;    // NW x 16 x PS: per-wave scratch for the diagonal-tile inverses
    double *dl;    // 16 x PS: factor of the current diagonal tile (+ reciprocal diagonal in column 16)
    double *p, *pold, *m, *y, *tr2, *rhs, *b, *red;  // NP each (red: 3*NP scratch)
    int *flag;
};

// ---- Cholesky of the 16x16 diagonal tile k by ONE wave: lane r (< 16) holds row r ---------------------------------
__device__ __forceinline__ bool factor_diag_tile(double *C, int ld, int k, double *dl, int lane) {
    double t[16];
    const int r = lane & 15;
    const double *src = C + (size_t)(16 * k + r) * ld + 16 * k;
#pragma unroll
    for (int c = 0; c < 16; ++c) t[c] = (lane < 16 && c <= r) ? src[c] : 0.0;
    bool ok = true;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const double d2 = bcast(t[c], c);
        ok = ok && (d2 > 0.0);
        const double d = sqrt(d2);
        const double dinv = 1.0 / d;
        t[c] = (r == c) ? d : t[c] * dinv;  // rows r > c: L[r][c]; (rows < c hold zeros)
        if (c < 15) {
#pragma unroll
            for (int c2 = c + 1; c2 < 16; ++c2) {
                const double s = bcast(t[c], c2);  // L[c2][c]
                t[c2] = fma(-t[c], s, t[c2]);      // only rows r >= c2 matter
            }
        }
        if (r == c) dl[r * PS + 16] = dinv;
    }
    if (lane < 16) {
        double *dst = C + (size_t)(16 * k + r) * ld + 16 * k;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const double v = c <= r ? t[c] : 0.0;
            dl[r * PS + c] = v;
            if (c <= r) dst[c] = v;
        }
    }
    return ok;
}

// ---- inverse of the 16x16 lower-triangular diagonal tile I of C into W, by one wave: lane c holds column c --------
// L_II is staged in this wave's LDS scratch `lw` (16 x PS) and read back with wave-uniform (broadcast) addresses.
__device__ __forceinline__ void invert_diag_tile(const double *C, double *W, int ld, int I, int lane, double *lw) {
    const int c = lane & 15;
    if (lane < 16) {
        const double *src = C + (size_t)(16 * I + c) * ld + 16 * I;
#pragma unroll
        for (int s = 0; s < 16; ++s) lw[c * PS + s] = (s <= c) ? src[s] : 0.0;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's LDS writes are done (single-wave hand-off)
    __builtin_amdgcn_wave_barrier();
    double x[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        double a = (r == c) ? 1.0 : 0.0;
#pragma unroll
        for (int s = 0; s < r; ++s) a = fma(-lw[r * PS + s], x[s], a);  // L[r][s] * X[s][c]
        x[r] = a / lw[r * PS + r];
    }
    if (lane < 16) {
#pragma unroll
        for (int r = 0; r < 16; ++r) W[(size_t)(16 * I + r) * ld + 16 * I + c] = (r >= c) ? x[r] : 0.0;
    }
}

// ---- one posterior solve: C = A + diag(1/p) -> L -> W = L^-1 -> y = W b, m = W^T y, tr2 = colnorm2(W) --------------
__device__ bool solve_posterior(const FitLoopParams &P, const Smem &S) {
    const int N = P.N, NP = P.NP, nb = P.NP / 16, ld = P.NP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cl = lane & 15, rg = lane >> 4;
    double *C = P.C, *W = P.W;
#ifdef FIT_LOOP_TIMING
    long long t_last = clock64();
#endif

    // (1) C (lower triangle) = A + diag(1/p); padding rows/cols = identity
    for (int i = wave; i < NP; i += NW) {
        const double *ar = P.A + (size_t)i * ld;
        double *cr = C + (size_t)i * ld;
        for (int j = lane; j <= i; j += 64) {
            double v = (i < N) ? ar[j] : 0.0;
            if (j == i) v = (i < N) ? v + 1.0 / S.p[i] : 1.0;
            cr[j] = v;
        }
    }
    if (tid == 0) *S.flag = 0;
    __syncthreads();
    TSTAMP(0);

    // (2) right-looking blocked Cholesky, 16-wide panels
    for (int k = 0; k < nb; ++k) {
        if (wave == 0) {
            const bool ok = factor_diag_tile(C, ld, k, S.dl, lane);
            if (!ok && lane == 0) *S.flag = 1;
        }
        __syncthreads();
        TSTAMP(1);
        if (*S.flag) return false;
        const int r0 = 16 * (k + 1);
        // panel: row i solves x L_kk^T = C[i, 16k:16k+16]
        for (int i = r0 + tid; i < NP; i += KT) {
            double *cr = C + (size_t)i * ld + 16 * k;
            double x[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) x[c] = cr[c];
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                double a = x[c];
#pragma unroll
                for (int s = 0; s < c; ++s) a = fma(-x[s], S.dl[c * PS + s], a);
                x[c] = a * S.dl[c * PS + 16];
            }
            double *pr = S.pan + (size_t)(i - r0) * PS;
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                cr[c] = x[c];
                pr[c] = x[c];
            }
        }
        __syncthreads();
        TSTAMP(2);
        // trailing update C_IJ -= L_Ik L_Jk^T for k < J <= I (tiles dealt round-robin to the waves)
        int ctr = 0;
        for (int I = k + 1; I < nb; ++I) {
            for (int J = k + 1; J <= I; ++J, ++ctr) {
                if ((ctr & (NW - 1)) != wave) continue;
                double *ct = C + (size_t)(16 * I + rg) * ld + 16 * J + cl;
                v4f64 acc;
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] = ct[(size_t)(4 * r) * ld];
                const double *pa = S.pan + (size_t)((I - k - 1) * 16 + cl) * PS + rg;
                const double *pb = S.pan + (size_t)((J - k - 1) * 16 + cl) * PS + rg;
#pragma unroll
                for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-pa[4 * s], pb[4 * s], acc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) ct[(size_t)(4 * r) * ld] = acc[r];
            }
        }
        __syncthreads();
        TSTAMP(3);
    }

    // (3) W_II = L_II^-1 for every diagonal tile
    for (int I = wave; I < nb; I += NW) invert_diag_tile(C, W, ld, I, lane, S.lw + wave * 16 * PS);
    __syncthreads();
    TSTAMP(4);

    // (4) W_IJ = -W_II * sum_{K=J}^{I-1} L_IK W_KJ, block row by block row
    const int LS = NP + 1;  // LDS stride of the staged block row of L
    for (int I = 1; I < nb; ++I) {
        for (int e = tid; e < 16 * 16 * I; e += KT) {
            const int a = e / (16 * I), c = e - a * (16 * I);
            S.pan[a * LS + c] = C[(size_t)(16 * I + a) * ld + c];
        }
        for (int e = tid; e < 256; e += KT) S.dl[(e >> 4) * PS + (e & 15)] = W[(size_t)(16 * I + (e >> 4)) * ld + 16 * I + (e & 15)];
        __syncthreads();
        TSTAMP(5);
        for (int J = wave; J < I; J += NW) {
            v4f64 acc = {0.0, 0.0, 0.0, 0.0};
            for (int K = J; K < I; ++K) {
                const double *pa = S.pan + cl * LS + 16 * K + rg;
                const double *wb = W + (size_t)(16 * K + rg) * ld + 16 * J + cl;
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[4 * s], wb[(size_t)(4 * s) * ld], acc, 0, 0, 0);
            }
            // the accumulator tile is directly the B operand (register s = rows 4s..4s+3): out = -W_II * acc
            v4f64 out = {0.0, 0.0, 0.0, 0.0};
            const double *wa = S.dl + cl * PS + rg;
#pragma unroll
            for (int s = 0; s < 4; ++s) out = __builtin_amdgcn_mfma_f64_16x16x4f64(-wa[4 * s], acc[s], out, 0, 0, 0);
            double *wt = W + (size_t)(16 * I + rg) * ld + 16 * J + cl;
#pragma unroll
            for (int r = 0; r < 4; ++r) wt[(size_t)(4 * r) * ld] = out[r];
        }
        __syncthreads();
        TSTAMP(6);
    }

    // (5) y = W b
    for (int r = wave; r < N; r += NW) {
        const double *wr = W + (size_t)r * ld;
        double a = 0.0;
        for (int c = lane; c <= r; c += 64) a = fma(wr[c], S.b[c], a);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) a += __shfl_down(a, off);
        if (lane == 0) S.y[r] = a;
    }
    __syncthreads();
    // (6) m_i = sum_{r>=i} W[r,i] y_r,  tr2_i = sum_{r>=i} W[r,i]^2  (three row segments, then combined)
    {
        const int col = tid % 320, seg = tid / 320;
        if (seg < 3 && col < N) {
            const int per = (N + 2) / 3;
            const int ra = seg * per, rb = min(N, ra + per);
            double am = 0.0, at = 0.0;
            for (int r = max(ra, col); r < rb; ++r) {
                const double w = W[(size_t)r * ld + col];
                am = fma(w, S.y[r], am);
                at = fma(w, w, at);
            }
            S.red[seg * 2 * NP + col] = am;
            S.red[seg * 2 * NP + NP + col] = at;
        }
    }
    __syncthreads();
    for (int i = tid; i < N; i += KT) {
        S.m[i] = S.red[i] + S.red[2 * NP + i] + S.red[4 * NP + i];
        S.tr2[i] = S.red[NP + i] + S.red[3 * NP + i] + S.red[5 * NP + i];
    }
    __syncthreads();
    TSTAMP(7);
    return true;
}

__global__ __launch_bounds__(KT) void fit_loop_kernel(FitLoopParams P) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int N = P.N, NP = P.NP;
    const int tid = threadIdx.x;
    Smem S;
    const int panel_doubles = max(NP * PS, 16 * (NP + 1));
    S.pan = smem;
    S.lw = S.pan + panel_doubles;
    S.dl = S.lw + NW * 16 * PS;
    S.p = S.dl + 16 * PS;
    S.pold = S.p + NP;
    S.m = S.pold + NP;
    S.y = S.m + NP;
    S.tr2 = S.y + NP;
    S.rhs = S.tr2 + NP;
    S.b = S.rhs + NP;
    S.red = S.b + NP;  // 6*NP
    S.flag = reinterpret_cast<int *>(S.red + 6 * NP);
    __shared__ int s_ctl[4];  // [0] stop, [1] status

    for (int i = tid; i < NP; i += KT) {
        S.b[i] = i < N ? P.bq[i] : 0.0;
        S.p[i] = i < N ? (P.p_init ? P.p_init[i] : 1.0) : 1.0;  // radial_fitters.py:744 (p = 1)
        S.pold[i] = 0.0;                                        // radial_fitters.py:768 (pi_old = 0)
    }
    if (tid == 0) {
        s_ctl[0] = 0;
        s_ctl[1] = 0;
    }
    __syncthreads();
    int status = 0, count = 0;
    // One call site for the posterior solve.  phase 0: p = 1 (radial_fitters.py:744-747); phase 1: power-law
    // guess (:749-752); phase 2: the loop of :769-785 (FIT_MODE_STEP: exactly one pass, FIT_MODE_SOLVE: none).
    int phase = (P.mode == FIT_MODE_FULL) ? 0 : 2;
    bool in_pass = false;
    for (;;) {
        if (!solve_posterior(P, S)) {
            status = FIT_STATUS_NOT_SPD;
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
        phase = 2;
        if (in_pass) {
            if (P.diag_mu) {  // MAP of this pass: mu = Y^-1 m   (radial_fitters.py:783)
                for (int r = tid >> 6; r < N; r += NW) {
                    const double *yr = P.Yinv + (size_t)r * N;
                    double a = 0.0;
                    for (int c = tid & 63; c < N; c += 64) a = fma(yr[c], S.m[c], a);
#pragma unroll
                    for (int off = 32; off >= 1; off >>= 1) a += __shfl_down(a, off);
                    if ((tid & 63) == 0) P.diag_mu[(size_t)count * N + r] = a;
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
        for (int i = tid; i < N; i += KT) {
            const double pi = S.p[i], mi = S.m[i];
            const double beta = (P.p0 + 0.5 * (mi * mi + S.tr2[i])) / pi - (P.alpha - 1.0 + 0.5 * 1.0);
            S.rhs[i] = beta + log(pi);
            S.pold[i] = pi;
        }
        __syncthreads();
        if (tid == 0) {  // banded LU solve with the host-prepared factors; the recurrence lives in registers
            const double *f1 = P.band_lu, *f2 = f1 + N, *d0 = f2 + N, *u1 = d0 + N, *u2 = u1 + N;
            double x1 = S.rhs[0], x2 = 0.0;  // x_{i-1}, x_{i-2}
            for (int i = 1; i < N; ++i) {
                double xi = S.rhs[i];
                xi = fma(-f2[i], x2, xi);
                xi = fma(-f1[i], x1, xi);
                S.rhs[i] = xi;
                x2 = x1;
                x1 = xi;
            }
            double y1 = 0.0, y2 = 0.0;  // x_{i+1}, x_{i+2}
            for (int i = N - 1; i >= 0; --i) {
                double t = S.rhs[i];
                t = fma(-u1[i], y1, t);
                t = fma(-u2[i], y2, t);
                t = t / d0[i];
                S.rhs[i] = t;
                y2 = y1;
                y1 = t;
            }
        }
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
    }

    // outputs: mu = Y^-1 m, p, count, status
    for (int r = tid >> 6; r < N; r += NW) {
        const double *yr = P.Yinv + (size_t)r * N;
        double a = 0.0;
        for (int c = tid & 63; c < N; c += 64) a = fma(yr[c], S.m[c], a);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) a += __shfl_down(a, off);
        if ((tid & 63) == 0) P.mu_out[r] = a;
    }
    for (int i = tid; i < N; i += KT) P.p_out[i] = S.p[i];
    if (tid == 0) {
        P.result[0] = count;
        P.result[1] = status;
    }
}

// A <- (A + A^T)/2 on the leading N x N block of an ld-strided buffer; zero padding elsewhere.
__global__ void symmetrize_pad_kernel(const double *Araw, int N, int NP, double *A) {
    const size_t total = (size_t)NP * NP;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int i = (int)(e / NP), j = (int)(e - (size_t)i * NP);
        A[e] = (i < N && j < N) ? 0.5 * (Araw[(size_t)i * N + j] + Araw[(size_t)j * N + i]) : 0.0;
    }
}

}  // namespace

size_t fh_k2_loop_smem_bytes(int NP) {
    const int panel = NP * PS > 16 * (NP + 1) ? NP * PS : 16 * (NP + 1);
    return sizeof(double) * (size_t)(panel + (NW + 1) * 16 * PS + 7 * NP + 6 * NP) + 16;
}

hipError_t fh_k2_launch_loop(const FitLoopParams &P, hipStream_t s) {
    const size_t smem = fh_k2_loop_smem_bytes(P.NP);
    static size_t attr_for = 0;
    if (smem > attr_for) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&fit_loop_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
        attr_for = smem;
    }
    hipLaunchKernelGGL(fit_loop_kernel, dim3(1), dim3(KT), smem, s, P);
    return hipGetLastError();
}

hipError_t fh_k2_launch_symmetrize(const double *Araw, int N, int NP, double *A, hipStream_t s) {
    hipLaunchKernelGGL(symmetrize_pad_kernel, dim3(128), dim3(256), 0, s, Araw, N, NP, A);
    return hipGetLastError();
}
