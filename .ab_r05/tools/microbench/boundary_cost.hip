// What a kernel boundary costs on one stream while persistent workgroups on ANOTHER stream keep the L2 caches full of dirty
// lines -- the situation of the binning stream beside 128 resident fit loops.
//   hipcc --offload-arch=gfx950 -O3 -o boundary_cost boundary_cost.hip && ./boundary_cost
// background modes: 0 none, 1 read-modify-write with plain stores (dirty lines in L2), 2 the same with system-scope
// (write-through) stores, 3 read only.  Foreground: 300 launches of an empty kernel, and of a small streaming kernel.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void background(double *buf, size_t doubles_per_wg, int mode, long long cycles, int *stop) {
    double *p = buf + (size_t)blockIdx.x * doubles_per_wg;
    const long long t0 = wall_clock64();
    double acc = 0.0;
    while (wall_clock64() - t0 < cycles) {
        for (size_t i = threadIdx.x; i < doubles_per_wg; i += blockDim.x) {
            const double v = p[i];
            if (mode == 1) p[i] = v + 1.0;
            else if (mode == 2) __hip_atomic_store(&p[i], v + 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            else acc += v;
        }
    }
    if (acc == 12345.678) *stop = 1;
}
__global__ void empty_kernel(int *x) { if (x && threadIdx.x == 1 << 30) *x = 0; }
__global__ void small_stream(const double *a, double *b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i] * 2.0;
}

int main() {
    const int wgs = 128;
    const size_t per = (1100 * 1024) / 8;  // 1.1 MB per workgroup, as a fit loop's working set
    double *bg, *a, *b;
    int *flag;
    CHECK(hipMalloc(&bg, wgs * per * 8));
    CHECK(hipMemset(bg, 0, wgs * per * 8));
    const size_t n = 4 << 20;  // 32 MB in, 32 MB out
    CHECK(hipMalloc(&a, n * 8));
    CHECK(hipMalloc(&b, n * 8));
    CHECK(hipMemset(a, 0, n * 8));
    CHECK(hipMalloc(&flag, 4));
    hipStream_t sa, sb;
    CHECK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipStream_t many[16];
    for (auto &m : many) CHECK(hipStreamCreateWithFlags(&m, hipStreamNonBlocking));
    for (int nl : {1, 2, 3, 4, 5, 6, 8, 12}) {  // the dirty background as nl launches on nl streams
        for (int l = 0; l < nl; ++l)
            hipLaunchKernelGGL(background, dim3(wgs / nl), dim3(768), 0, many[l], bg + (size_t)l * (wgs / nl) * per, per, 1, 20000000ll, flag);
        for (int kind = 0; kind < 2; ++kind) {
            CHECK(hipEventRecord(e0, sb));
            const int reps = 300;
            for (int r = 0; r < reps; ++r) {
                if (kind == 0) hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, sb, (int *)nullptr);
                else hipLaunchKernelGGL(small_stream, dim3(1024), dim3(256), 0, sb, a, b, n);
            }
            CHECK(hipEventRecord(e1, sb));
            CHECK(hipEventSynchronize(e1));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            printf("%2d background launches on as many streams, %s: %.1f us per launch\n", nl, kind == 0 ? "empty kernel" : "64 MB streaming kernel", 1e3 * ms / reps);
        }
        CHECK(hipDeviceSynchronize());
    }
    for (int mode = 0; mode < 8; ++mode) {
        // modes 6, 7: 8 / 16 launches on ONE stream with hipExtAnyOrderLaunch (no barrier between them)
        if (mode >= 6) {
            const int nl = mode == 6 ? 8 : 16;
            hipEvent_t b0, b1;
            CHECK(hipEventCreate(&b0));
            CHECK(hipEventCreate(&b1));
            CHECK(hipEventRecord(b0, sa));
            for (int l = 0; l < nl; ++l)
                hipExtLaunchKernelGGL(background, dim3(wgs / nl), dim3(768), 0, sa, nullptr, nullptr, hipExtAnyOrderLaunch,
                                      bg + (size_t)l * (wgs / nl) * per, per, 1, 40000000ll, flag);
            CHECK(hipEventRecord(b1, sa));
            for (int kind = 0; kind < 2; ++kind) {
                CHECK(hipEventRecord(e0, sb));
                const int reps = 300;
                for (int r = 0; r < reps; ++r) {
                    if (kind == 0) hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, sb, (int *)nullptr);
                    else hipLaunchKernelGGL(small_stream, dim3(1024), dim3(256), 0, sb, a, b, n);
                }
                CHECK(hipEventRecord(e1, sb));
                CHECK(hipEventSynchronize(e1));
                float ms = 0;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                printf("background mode %d, %s: %.1f us per launch\n", mode, kind == 0 ? "empty kernel" : "64 MB streaming kernel", 1e3 * ms / reps);
            }
            CHECK(hipDeviceSynchronize());
            float bms = 0;
            CHECK(hipEventElapsedTime(&bms, b0, b1));
            printf("   the %d any-order launches together: %.0f ms (one alone: ~400)\n", nl, bms);
            continue;
        }
        // modes 4, 5: the dirty background as 8 / 16 launches of 16 / 8 workgroups on as many streams (hardware queues)
        if (mode >= 4) {
            const int nl = mode == 4 ? 8 : 16;
            for (int l = 0; l < nl; ++l)
                hipLaunchKernelGGL(background, dim3(wgs / nl), dim3(768), 0, many[l], bg + (size_t)l * (wgs / nl) * per, per, 1, 40000000ll, flag);
        } else
        if (mode) hipLaunchKernelGGL(background, dim3(wgs), dim3(768), 0, sa, bg, per, mode, 40000000ll, flag);  // 0.4 s at 100 MHz
        for (int kind = 0; kind < 2; ++kind) {
            CHECK(hipEventRecord(e0, sb));
            const int reps = 300;
            for (int r = 0; r < reps; ++r) {
                if (kind == 0) hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, sb, (int *)nullptr);
                else hipLaunchKernelGGL(small_stream, dim3(1024), dim3(256), 0, sb, a, b, n);
            }
            CHECK(hipEventRecord(e1, sb));
            CHECK(hipEventSynchronize(e1));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            printf("background mode %d, %s: %.1f us per launch\n", mode, kind == 0 ? "empty kernel" : "64 MB streaming kernel", 1e3 * ms / reps);
        }
        CHECK(hipDeviceSynchronize());
    }
    return 0;
}
