// Checks tilechol::chol_inv_tile_acc (frank_amd/csrc/tile_chol.h) against a host Cholesky / inverse on random SPD tiles and
// times it:   hipcc --offload-arch=gfx950 -O3 -I frank_amd/csrc tools/microbench/tile_chol_test.hip -o /tmp/tct && /tmp/tct
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "tile_chol.h"

__global__ void k(const double *in, double *Lout, double *Xout, int *okout, long long *cyc, int reps) {
    const int lane = threadIdx.x & 63, rg = lane >> 4, cl = lane & 15;
    const double *t = in + (size_t)blockIdx.x * 256;
    v4f64 T0;
    for (int r = 0; r < 4; ++r) T0[r] = t[(rg + 4 * r) * 16 + cl];
    v4f64 T = T0, X;
    bool ok = true;
    const long long c0 = clock64();
    for (int it = 0; it < reps; ++it) {
        T = T0;
        T[0] += 1e-300 * it;  // (keeps the repetitions from being folded)
        ok = tilechol::chol_inv_tile_acc(T, X, lane, -1);
    }
    const long long c1 = clock64();
    for (int r = 0; r < 4; ++r) {
        Lout[(size_t)blockIdx.x * 256 + (rg + 4 * r) * 16 + cl] = T[r];
        Xout[(size_t)blockIdx.x * 256 + (rg + 4 * r) * 16 + cl] = X[r];
    }
    if (lane == 0) {
        okout[blockIdx.x] = ok;
        cyc[blockIdx.x] = (c1 - c0) / reps;
    }
}

int main() {
    const int nt = 64, reps = 200;
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
    double *dA, *dL, *dX; int *dok; long long *dc;
    hipMalloc(&dA, sizeof(double) * nt * 256); hipMalloc(&dL, sizeof(double) * nt * 256); hipMalloc(&dX, sizeof(double) * nt * 256);
    hipMalloc(&dok, sizeof(int) * nt); hipMalloc(&dc, sizeof(long long) * nt);
    hipMemcpy(dA, A.data(), sizeof(double) * nt * 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(nt), dim3(64), 0, 0, dA, dL, dX, dok, dc, reps);
    hipMemcpy(L.data(), dL, sizeof(double) * nt * 256, hipMemcpyDeviceToHost);
    hipMemcpy(X.data(), dX, sizeof(double) * nt * 256, hipMemcpyDeviceToHost);
    std::vector<int> ok(nt); std::vector<long long> cyc(nt);
    hipMemcpy(ok.data(), dok, sizeof(int) * nt, hipMemcpyDeviceToHost);
    hipMemcpy(cyc.data(), dc, sizeof(long long) * nt, hipMemcpyDeviceToHost);
    double worstL = 0, worstX = 0, worstU = 0;
    for (int b = 0; b < nt; ++b) {
        const double *a = &A[b * 256], *l = &L[b * 256], *x = &X[b * 256];
        double scale = 0;
        for (int i = 0; i < 256; ++i) scale = fmax(scale, fabs(a[i]));
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                double r1 = 0, r2 = 0;
                for (int k2 = 0; k2 < 16; ++k2) { r1 += l[i * 16 + k2] * l[j * 16 + k2]; r2 += l[i * 16 + k2] * x[k2 * 16 + j]; }
                worstL = fmax(worstL, fabs(r1 - a[i * 16 + j]) / scale);
                worstX = fmax(worstX, fabs(r2 - (i == j)));
                if (j > i) worstU = fmax(worstU, fmax(fabs(l[i * 16 + j]), fabs(x[i * 16 + j])));
            }
        if (!ok[b]) printf("tile %d not ok\n", b);
    }
    unsigned long long hsh = 1469598103934665603ULL;
    for (int i = 0; i < nt * 256; ++i) {
        unsigned long long a, b;
        memcpy(&a, &L[i], 8); memcpy(&b, &X[i], 8);
        hsh = (hsh ^ a) * 1099511628211ULL; hsh = (hsh ^ b) * 1099511628211ULL;
    }
    printf("bits of L and X: %016llx\n", hsh);
    printf("max |L L^T - A| / max|A| = %.2e   max |L X - I| = %.2e   max upper = %.2e   cycles per tile (s_memtime) = %lld\n", worstL, worstX,
           worstU, cyc[0]);
    return !(worstL < 1e-13 && worstX < 1e-9 && worstU == 0.0);
}
