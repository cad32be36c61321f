// How fast does one SIMD of gfx950 issue v_mfma_f64_16x16x4_f64?  K independent accumulator chains per wave, W waves per SIMD,
// products back to back (inline asm, accumulators tied in place): cycles per instruction per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_f64_bench mfma_f64_bench.hip && ./mfma_f64_bench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4f64 __attribute__((ext_vector_type(4)));

template <int K>
__global__ void bench(long long *out, int iters, double seed) {
    v4f64 acc[K];
    for (int c = 0; c < K; ++c) acc[c] = v4f64{seed, seed, seed, seed};
    double a = seed * 0.5, b = seed * 0.25;
    __syncthreads();
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
#pragma unroll
            for (int c = 0; c < K; ++c) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[c]) : "v"(a), "v"(b));
        }
    }
    asm volatile("s_nop 15\n\ts_nop 3");
    const long long t1 = clock64();
    double s = 0.0;
    for (int c = 0; c < K; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    if (threadIdx.x % 64 == 0) out[threadIdx.x / 64] = t1 - t0;
    if (s == 12345.678) out[63] = 1;
}

template <int K>
void run(int threads, long long *d) {
    const int iters = 2000;
    hipLaunchKernelGGL(bench<K>, dim3(1), dim3(threads), 0, 0, d, iters, 1e-3);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(bench<K>, dim3(1), dim3(threads), 0, 0, d, iters, 1e-3);
    long long h[16];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const int waves = threads / 64, per_simd = (waves + 3) / 4;
    const double cyc = (double)h[0];
    printf("%d waves (%d per SIMD), %d chains each: %.1f shader-clock cycles per instruction per wave, %.1f per instruction per SIMD\n", waves,
           per_simd, K, cyc / (iters * 4.0 * K), cyc / (iters * 4.0 * K * per_simd));
}

int main() {
    long long *d;
    hipMalloc(&d, 64 * sizeof(long long));
    for (int threads : {64, 256, 512, 768}) {
        run<1>(threads, d);
        run<2>(threads, d);
        run<3>(threads, d);
        run<4>(threads, d);
    }
    return 0;
}
