// Measures the fp64 ceilings the K1 roofline is priced against, on the device it runs on:
//   (1) v_mfma_f64_16x16x4_f64 issue rate (independent accumulators, 1 or 2 waves per SIMD),
//   (2) v_fma_f64 VALU rate,
//   (3) both together in one wave (does VALU work hide under the matrix pipe?).
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench/fp64_peaks.hip -o /tmp/fp64_peaks
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4f64 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(512) void mfma_loop(double *out, int iters, double a0, double b0) {
    v4f64 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = v4f64{0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(512) void fma_loop(double *out, int iters, double a0, double b0) {
    double x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = a0 + i + threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = fma(x[i], b0, a0);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// NV v_fma_f64 per MFMA, interleaved in one instruction stream
template <int NV>
__global__ __launch_bounds__(512) void mixed_loop(double *out, int iters, double a0, double b0) {
    v4f64 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = v4f64{0, 0, 0, 0};
    double x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = a0 + i + threadIdx.x * 1e-9;
    double a = a0, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < NV; ++k) x[(i + k) & 7] = fma(x[(i + k) & 7], b0, a0);
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
static float time_ms(F launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    double *out;
    hipMalloc(&out, sizeof(double) * 4096 * 1024);
    const int iters = 20000;
    printf("device %s, %d CUs, clock %d MHz\n", prop.name, cus, prop.clockRate / 1000);
    for (int threads : {256, 512}) {
        float ms = time_ms([&] { hipLaunchKernelGGL(mfma_loop<8>, dim3(cus), dim3(threads), 0, 0, out, iters, 1.0, 1e-3); });
        double flops = (double)cus * (threads / 64) * iters * 8 * 2048.0;
        printf("mfma_f64_16x16x4  %d waves/SIMD: %.2f ms  %.1f TFLOP/s  (%.1f cycles/MFMA/SIMD at 2.4 GHz)\n", threads / 256, ms,
               flops / ms / 1e9, ms * 1e-3 * 2.4e9 / (iters * 8.0 * (threads / 256)));
    }
    for (int threads : {256, 512}) {
        float ms = time_ms([&] { hipLaunchKernelGGL(fma_loop, dim3(cus), dim3(threads), 0, 0, out, iters, 1.0, 0.999); });
        double flops = (double)cus * threads * iters * 16 * 2.0;
        printf("v_fma_f64         %d waves/SIMD: %.2f ms  %.1f TFLOP/s\n", threads / 256, ms, flops / ms / 1e9);
    }
    {
        float m0 = time_ms([&] { hipLaunchKernelGGL(mixed_loop<0>, dim3(cus), dim3(256), 0, 0, out, iters, 1.0, 0.999); });
        float m4 = time_ms([&] { hipLaunchKernelGGL(mixed_loop<4>, dim3(cus), dim3(256), 0, 0, out, iters, 1.0, 0.999); });
        float m8 = time_ms([&] { hipLaunchKernelGGL(mixed_loop<8>, dim3(cus), dim3(256), 0, 0, out, iters, 1.0, 0.999); });
        float m12 = time_ms([&] { hipLaunchKernelGGL(mixed_loop<12>, dim3(cus), dim3(256), 0, 0, out, iters, 1.0, 0.999); });
        printf("1 wave/SIMD, MFMA + k v_fma_f64 per MFMA: k=0 %.2f ms, k=4 %.2f ms, k=8 %.2f ms, k=12 %.2f ms\n", m0, m4, m8, m12);
        float n4 = time_ms([&] { hipLaunchKernelGGL(mixed_loop<4>, dim3(cus), dim3(512), 0, 0, out, iters, 1.0, 0.999); });
        float n8 = time_ms([&] { hipLaunchKernelGGL(mixed_loop<8>, dim3(cus), dim3(512), 0, 0, out, iters, 1.0, 0.999); });
        printf("2 waves/SIMD (same total MFMAs x2): k=4 %.2f ms, k=8 %.2f ms\n", n4, n8);
    }
    return 0;
}
