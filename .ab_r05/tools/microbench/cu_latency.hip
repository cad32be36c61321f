// Single-workgroup latencies on MI355X (gfx950): what one step of a latency-bound kernel (fit_loop, lognormal) is made of.
// One workgroup of 512 threads (8 waves, 2 per SIMD) -- the shape of those kernels.  Cycles = clock64() of thread 0
// divided by the repeat count.   hipcc --offload-arch=gfx950 -O2 cu_latency.hip -o cu_latency
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/warp/warp_reduce.hpp>
#include <cstdio>
#include <vector>

constexpr int R = 2000;
struct FMax { __device__ double operator()(double a, double b) const { return __builtin_fmax(a, b); } };

__global__ __launch_bounds__(512) void lat(long long *out, double *gbuf, int *chain, double seed) {
    __shared__ double sh[1024];
    __shared__ int ich[1024];
    const int tid = threadIdx.x;
    for (int i = tid; i < 1024; i += 512) { sh[i] = seed + i; ich[i] = (i * 17 + 1) & 1023; }
    __syncthreads();
    long long t0, t1;
    double acc = seed;
    // 0: barrier only
    t0 = clock64();
    for (int r = 0; r < R; ++r) __syncthreads();
    t1 = clock64(); if (tid == 0) out[0] = t1 - t0;
    // 1: LDS write -> barrier -> LDS read (the hand-over of one value between waves)
    t0 = clock64();
    for (int r = 0; r < R; ++r) { sh[tid] = acc; __syncthreads(); acc += sh[(tid + 64) & 511]; __syncthreads(); }
    t1 = clock64(); if (tid == 0) out[1] = t1 - t0;
    // 2: wave max of a double (rocprim DPP) with the result in every lane
    t0 = clock64();
    for (int r = 0; r < R; ++r) {
        using WR = rocprim::warp_reduce<double, 64, true>;
        typename WR::storage_type st; double o; WR().reduce(acc + tid, o, st, FMax()); acc = o * 0.5;
    }
    t1 = clock64(); if (tid == 0) out[2] = t1 - t0;
    // 3: dependent LDS reads (pointer chase): LDS latency
    int idx = tid & 1023;
    t0 = clock64();
    for (int r = 0; r < R; ++r) idx = ich[idx];
    t1 = clock64(); if (tid == 0) out[3] = t1 - t0;
    acc += idx;
    // 4: dependent fp64 fma chain
    t0 = clock64();
    for (int r = 0; r < R; ++r) acc = fma(acc, 0.999999, 1e-9);
    t1 = clock64(); if (tid == 0) out[4] = t1 - t0;
    // 5: dependent fp64 divisions
    t0 = clock64();
    for (int r = 0; r < R; ++r) acc = 1.0 + 1.0 / acc;
    t1 = clock64(); if (tid == 0) out[5] = t1 - t0;
    // 6: dependent global loads (L2-resident pointer chase, one lane-uniform chain per thread)
    int g = tid;
    t0 = clock64();
    for (int r = 0; r < R; ++r) g = chain[g];
    t1 = clock64(); if (tid == 0) out[6] = t1 - t0;
    acc += g;
    // 7: global store -> barrier -> global load by another wave (the hand-over through L2 used between phases)
    t0 = clock64();
    for (int r = 0; r < R; ++r) { gbuf[tid] = acc; __syncthreads(); acc += gbuf[(tid + 64) & 511]; __syncthreads(); }
    t1 = clock64(); if (tid == 0) out[7] = t1 - t0;
    // 8: one 16x16x4 fp64 MFMA chain (dependent through the accumulator)
    typedef double v4 __attribute__((ext_vector_type(4)));
    v4 c = {acc, acc, acc, acc};
    t0 = clock64();
    for (int r = 0; r < R; ++r) c = __builtin_amdgcn_mfma_f64_16x16x4f64(acc, 1.0, c, 0, 0, 0);
    t1 = clock64(); if (tid == 0) out[8] = t1 - t0;
    // 9: readlane broadcast of a double + fma (the serial 16x16 tile chains)
    t0 = clock64();
    for (int r = 0; r < R; ++r) {
        const int lo = __builtin_amdgcn_readlane(__double2loint(acc), r & 63), hi = __builtin_amdgcn_readlane(__double2hiint(acc), r & 63);
        acc = fma(__hiloint2double(hi, lo), 0.5, acc * 0.25);
    }
    t1 = clock64(); if (tid == 0) out[9] = t1 - t0;
    gbuf[512 + tid] = acc + c[0];
}

int main() {
    long long *d; double *g; int *ch;
    hipMalloc(&d, 16 * sizeof(long long)); hipMalloc(&g, 2048 * sizeof(double)); hipMalloc(&ch, 4096 * sizeof(int));
    std::vector<int> h(4096); for (int i = 0; i < 4096; ++i) h[i] = (i * 193 + 7) & 4095;
    hipMemcpy(ch, h.data(), 4096 * sizeof(int), hipMemcpyHostToDevice);
    hipMemset(g, 0, 2048 * sizeof(double));
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(lat, dim3(1), dim3(512), 0, 0, d, g, ch, 1.5); hipDeviceSynchronize(); }
    long long o[16]; hipMemcpy(o, d, sizeof o, hipMemcpyDeviceToHost);
    const char *names[] = {"__syncthreads() alone", "LDS write -> barrier -> LDS read -> barrier", "wave max of a double (DPP, all lanes)",
                           "dependent LDS read", "dependent v_fma_f64", "dependent fp64 division (+1 add)", "dependent global load (L2 hit)",
                           "global store -> barrier -> global load -> barrier", "dependent v_mfma_f64_16x16x4", "readlane x2 + 2 fp64 ops"};
    printf("one workgroup, 512 threads (2 waves per SIMD), cycles per repetition (clock64):\n");
    for (int i = 0; i < 10; ++i) printf("  %-52s %8.1f\n", names[i], (double)o[i] / R);
    return 0;
}
