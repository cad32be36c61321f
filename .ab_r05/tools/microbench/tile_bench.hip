// Micro-benchmark of the 16 x 16 factor-and-invert tile routines (tile_chol.h): one wave alone on a CU, cycles per tile, and the
// new forms against the round 2-4 one (max relative difference of L and L^-1; residuals against the tile itself).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I frank_amd/csrc -o /tmp/tile_bench tools/microbench/tile_bench.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "tile_chol.h"
using namespace tilechol;

template <int V>
__global__ void __launch_bounds__(64) bench(const double *tiles, int ntile, int reps, double *outL, double *outX, long long *cyc,
                                            int *okout, int force_c) {
    const int lane = threadIdx.x, rg = lane >> 4, cl = lane & 15;
    long long total = 0;
    int okall = 1;
    for (int rep = 0; rep < reps; ++rep) {
        for (int t = 0; t < ntile; ++t) {
            v4f64 T, X;
            for (int r = 0; r < 4; ++r) T[r] = tiles[t * 256 + (rg + 4 * r) * 16 + cl];
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const long long t0 = __builtin_readcyclecounter();
            bool ok;
            if constexpr (V == 0) {
                ok = chol_inv_tile_acc(T, X, lane, force_c);
            } else if constexpr (V == 3) {  // the FORCE instantiation (the augmented row of the fit loop: pivot force_c is 1)
                v4f64 Z;
                ok = chol_inv_tile_z<true, true>(T, Z, lane, force_c);
                X = Z;
            } else {
                v4f64 Z;
                ok = chol_inv_tile_z<false, (V & 1) != 0>(T, Z, lane, -1);
                X = Z;
            }
            asm volatile("" : "+v"(T), "+v"(X));
            const long long t1 = __builtin_readcyclecounter();
            total += t1 - t0;
            okall &= ok ? 1 : 0;
            if (rep == 0)
                for (int r = 0; r < 4; ++r) {
                    // V > 0 returns the transposes: element (cl, rg + 4 r)
                    const int idx = V == 0 ? (rg + 4 * r) * 16 + cl : cl * 16 + rg + 4 * r;
                    outL[t * 256 + idx] = T[r];
                    outX[t * 256 + idx] = X[r];
                }
        }
    }
    if (lane == 0) {
        *cyc = total;
        *okout = okall;
    }
}

static void check(const char *name, const std::vector<double> &tiles, const std::vector<double> &L, const std::vector<double> &X, int nt) {
    double rl = 0, rx = 0, up = 0;
    for (int t = 0; t < nt; ++t) {
        const double *A = &tiles[t * 256], *l = &L[t * 256], *x = &X[t * 256];
        double an = 0;
        for (int i = 0; i < 256; ++i) an = fmax(an, fabs(A[i]));
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                double s = 0, e = 0;
                for (int k = 0; k < 16; ++k) s += l[i * 16 + k] * l[j * 16 + k], e += x[i * 16 + k] * l[k * 16 + j];
                rl = fmax(rl, fabs(s - A[i * 16 + j]) / an);
                rx = fmax(rx, fabs(e - (i == j)));
                if (j > i) up = fmax(up, fmax(fabs(l[i * 16 + j]), fabs(x[i * 16 + j])));
            }
    }
    printf("%-10s |L L^T - A|/|A| = %.2e   |X L - I| = %.2e   above the diagonal %.1e\n", name, rl, rx, up);
}

template <int V>
static void run(const char *name, const std::vector<double> &tiles, int nt, std::vector<double> &L, std::vector<double> &X, int force_c = -1) {
    double *dT, *dL, *dX;
    long long *dc;
    int *dok;
    hipMalloc(&dT, tiles.size() * 8), hipMalloc(&dL, tiles.size() * 8), hipMalloc(&dX, tiles.size() * 8), hipMalloc(&dc, 8), hipMalloc(&dok, 4);
    hipMemcpy(dT, tiles.data(), tiles.size() * 8, hipMemcpyHostToDevice);
    const int reps = 200;
    bench<V><<<1, 64>>>(dT, nt, 2, dL, dX, dc, dok, force_c);
    bench<V><<<1, 64>>>(dT, nt, reps, dL, dX, dc, dok, force_c);
    hipDeviceSynchronize();
    long long c;
    int ok;
    hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost), hipMemcpy(&ok, dok, 4, hipMemcpyDeviceToHost);
    L.resize(tiles.size()), X.resize(tiles.size());
    hipMemcpy(L.data(), dL, tiles.size() * 8, hipMemcpyDeviceToHost), hipMemcpy(X.data(), dX, tiles.size() * 8, hipMemcpyDeviceToHost);
    printf("%-10s %8.0f shader cycles per tile (ok = %d)\n", name, (double)c / (reps * nt), ok);
    if (force_c < 0) check(name, tiles, L, X, nt);  // (with a forced pivot L L^T is not the tile)
    hipFree(dT), hipFree(dL), hipFree(dX), hipFree(dc), hipFree(dok);
}

int main() {
    const int nt = 32;
    std::vector<double> tiles(nt * 256);
    srand(7);
    for (int t = 0; t < nt; ++t) {
        double B[256];
        for (double &b : B) b = rand() / (double)RAND_MAX - 0.5;
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                double s = 0;
                for (int k = 0; k < 16; ++k) s += B[i * 16 + k] * B[j * 16 + k];
                tiles[t * 256 + i * 16 + j] = s + (i == j ? 0.5 + t : 0.0);
            }
    }
    std::vector<double> L0, X0, L1, X1, L2, X2;
    run<0>("rounds2-4", tiles, nt, L0, X0);
    run<1>("z, L", tiles, nt, L1, X1);
    run<2>("z", tiles, nt, L2, X2);
    L2 = L1;  // (not returned by that form)
    double dl = 0, dx = 0, d12 = 0;
    for (size_t i = 0; i < L0.size(); ++i) {
        dl = fmax(dl, fabs(L0[i] - L1[i])), dx = fmax(dx, fabs(X0[i] - X1[i]));
        d12 = fmax(d12, fmax(fabs(L1[i] - L2[i]), fabs(X1[i] - X2[i])));
    }
    printf("new against old: max |dL| = %.2e, max |dX| = %.2e; with against without L: %.1e (must be 0)\n", dl, dx, d12);
    // the FORCE instantiation (pivot of one column forced to 1: the row of the fit loop's padded system that carries b) against the
    // routine of rounds 2-4 with the same forced column, every column in turn being the forced one over the tiles
    double fl = 0, fx = 0;
    for (int fc : {0, 1, 5, 12, 14, 15}) {
        std::vector<double> La, Xa, Lb, Xb;
        char na[32], nb[32];
        snprintf(na, sizeof na, "old, f=%d", fc), snprintf(nb, sizeof nb, "z force f=%d", fc);
        run<0>(na, tiles, nt, La, Xa, fc);
        run<3>(nb, tiles, nt, Lb, Xb, fc);
        for (size_t i = 0; i < La.size(); ++i) fl = fmax(fl, fabs(La[i] - Lb[i])), fx = fmax(fx, fabs(Xa[i] - Xb[i]));
    }
    printf("forced pivot, new against old: max |dL| = %.2e, max |dX| = %.2e\n", fl, fx);
    return 0;
}
