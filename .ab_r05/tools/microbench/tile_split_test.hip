// Prototype: the factorisation of a 16 x 16 tile and the inversion of its factor on TWO waves (the factoring wave publishes
// every finished column of L and its reciprocal pivot in LDS, the inverting wave follows one column behind), against
// tilechol::chol_inv_tile_acc which does both in one wave.  Same operations per entry: the bits of L and X must agree.
//   hipcc --offload-arch=gfx950 -O3 -I frank_amd/csrc tools/microbench/tile_split_test.hip -o tools/microbench/tile_split_test
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "tile_chol.h"

using namespace tilechol;

template <int C>
__device__ __forceinline__ void factor_step(v4f64 &T, v4f64 &Lo, int rg, int cl, bool &ok, double *col, double *dv, int *flag) {
    constexpr int R = C >> 2, Q = C & 3;
    double d = bcast(T[R], Q * 16 + C);
    ok = ok && (d > 0.0);
    const double dinv = rsqrt_f64(d);
    const double b = __shfl(T[R], Q * 16 + cl) * dinv;
    const bool colC = cl == C;
#pragma unroll
    for (int r = R; r < 4; ++r) {
        const double a = dpp_row_bcast_c<C>(T[r]) * dinv;
        T[r] = fma(-a, b, T[r]);
        if (r > R) Lo[r] = colC ? a : Lo[r];
        else Lo[r] = (colC && rg >= Q) ? a : Lo[r];
        if (cl == 0) col[C * 16 + rg + 4 * r] = a;  // L[rg + 4 r][C]
    }
    if (rg == 0 && cl == 0) dv[C] = dinv;
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the column is in LDS
    if (rg == 0 && cl == 0) __hip_atomic_store(flag, C + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
template <int C>
__device__ __forceinline__ void invert_step(v4f64 &X, int rg, int cl, const double *col, const double *dv, int *flag) {
    constexpr int R = C >> 2, Q = C & 3;
    while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) <= C) {
    }
    const double dinv = dv[C];
    double a[4];
#pragma unroll
    for (int r = R; r < 4; ++r) a[r] = col[C * 16 + rg + 4 * r];
    const double xs = __shfl(X[R], Q * 16 + cl) * dinv;  // row C of X, final
#pragma unroll
    for (int r = R; r < 4; ++r) {
        if (r > R) {
            X[r] = fma(-a[r], xs, X[r]);
        } else {
            const double xu = fma(-a[r], xs, X[r]);
            double x = (rg > Q) ? xu : X[r];
            x = (rg == Q) ? xs : x;
            X[r] = x;
        }
    }
}
template <int... Cs>
__device__ __forceinline__ void factor_steps(v4f64 &T, v4f64 &Lo, int rg, int cl, bool &ok, double *col, double *dv, int *flag,
                                             std::integer_sequence<int, Cs...>) {
    (factor_step<Cs>(T, Lo, rg, cl, ok, col, dv, flag), ...);
}
template <int... Cs>
__device__ __forceinline__ void invert_steps(v4f64 &X, int rg, int cl, const double *col, const double *dv, int *flag,
                                             std::integer_sequence<int, Cs...>) {
    (invert_step<Cs>(X, rg, cl, col, dv, flag), ...);
}

__global__ void k(const double *in, double *Lout, double *Xout, long long *cyc, int reps, int split) {
    __shared__ double col[256], dv[16];
    __shared__ int flag;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int rg = lane >> 4, cl = lane & 15;
    asm volatile("" : "+v"(rg), "+v"(cl));
    const double *t = in + (size_t)blockIdx.x * 256;
    v4f64 T0;
    for (int r = 0; r < 4; ++r) T0[r] = t[(rg + 4 * r) * 16 + cl];
    v4f64 T = T0, X = T0, Lo;
    if (threadIdx.x == 0) flag = 0;
    __syncthreads();
    const long long c0 = clock64();
    for (int it = 0; it < reps; ++it) {
        if (!split) {
            if (wave == 0) {
                T = T0;
                T[0] += 1e-300 * it;
                chol_inv_tile_acc(T, X, lane, -1);
            }
        } else {
            if (wave == 0) {
                T = T0;
                T[0] += 1e-300 * it;
                Lo = v4f64{0.0, 0.0, 0.0, 0.0};
                bool ok = true;
                factor_steps(T, Lo, rg, cl, ok, col, dv, &flag, std::make_integer_sequence<int, 16>{});
                T = Lo;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) X[r] = (rg + 4 * r == cl) ? 1.0 : 0.0;
                invert_steps(X, rg, cl, col, dv, &flag, std::make_integer_sequence<int, 16>{});
            }
            __syncthreads();
            if (threadIdx.x == 0) flag = 0;
            __syncthreads();
        }
    }
    const long long c1 = clock64();
    if (wave == 0)
        for (int r = 0; r < 4; ++r) Lout[(size_t)blockIdx.x * 256 + (rg + 4 * r) * 16 + cl] = T[r];
    if (wave == (split ? 1 : 0))
        for (int r = 0; r < 4; ++r) Xout[(size_t)blockIdx.x * 256 + (rg + 4 * r) * 16 + cl] = X[r];
    if (threadIdx.x == 0) cyc[blockIdx.x] = (c1 - c0) / reps;
}

int main() {
    const int nt = 32, reps = 200;
    std::vector<double> A(nt * 256), L(nt * 256), X(nt * 256);
    unsigned long long s = 88172645463325252ULL;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) / 9007199254740992.0 - 0.5; };
    for (int b = 0; b < nt; ++b) {
        double G[16][16];
        for (auto &row : G) for (double &v : row) v = rnd();
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                double a = (i == j) ? 1e-3 * (b + 1) : 0.0;
                for (int k2 = 0; k2 < 16; ++k2) a += G[i][k2] * G[j][k2];
                A[b * 256 + i * 16 + j] = a;
            }
    }
    double *dA, *dL, *dX; long long *dc;
    hipMalloc(&dA, sizeof(double) * nt * 256); hipMalloc(&dL, sizeof(double) * nt * 256); hipMalloc(&dX, sizeof(double) * nt * 256);
    hipMalloc(&dc, sizeof(long long) * nt);
    hipMemcpy(dA, A.data(), sizeof(double) * nt * 256, hipMemcpyHostToDevice);
    unsigned long long hashes[2];
    for (int split = 0; split < 2; ++split) {
        hipLaunchKernelGGL(k, dim3(nt), dim3(128), 0, 0, dA, dL, dX, dc, reps, split);
        hipMemcpy(L.data(), dL, sizeof(double) * nt * 256, hipMemcpyDeviceToHost);
        hipMemcpy(X.data(), dX, sizeof(double) * nt * 256, hipMemcpyDeviceToHost);
        std::vector<long long> cyc(nt);
        hipMemcpy(cyc.data(), dc, sizeof(long long) * nt, hipMemcpyDeviceToHost);
        unsigned long long hsh = 1469598103934665603ULL;
        for (int i = 0; i < nt * 256; ++i) {
            unsigned long long a, b;
            memcpy(&a, &L[i], 8); memcpy(&b, &X[i], 8);
            hsh = (hsh ^ a) * 1099511628211ULL; hsh = (hsh ^ b) * 1099511628211ULL;
        }
        hashes[split] = hsh;
        long long mn = cyc[0];
        for (long long c : cyc) mn = c < mn ? c : mn;
        printf("%s: %lld cycles per tile (clock64 ticks), hash %016llx\n", split ? "two waves" : "one wave ", mn, hsh);
    }
    printf(hashes[0] == hashes[1] ? "bits agree\n" : "BITS DIFFER\n");
    return 0;
}
