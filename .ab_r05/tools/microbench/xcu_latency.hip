// Latency of reading what another compute unit of the same XCD has just written (MI355X): plain / device-scope stores on the
// producer, plain / device-scope (sc1) loads on the consumer; one wave each, workgroups 0 and 8 (same XCD under round-robin dispatch).
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int NT = 64;  // tiles of 2 KB
__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xf; }
template <int ST, int LD>
__global__ void __launch_bounds__(64) k(double *buf, int *flag, long long *out, int *xcc) {
    const int lane = threadIdx.x;
    if (blockIdx.x != 0 && blockIdx.x != 8) return;
    if (lane == 0) xcc[blockIdx.x ? 1 : 0] = xcc_id();
    if (blockIdx.x == 0) {  // producer
        for (int t = 0; t < NT; ++t) {
            typedef double dv2 __attribute__((ext_vector_type(2))); dv2 v = {(double)(t + lane), (double)(t - lane)};
            char *p = (char *)buf + t * 2048 + lane * 16;
            if (ST == 0) asm volatile("global_store_dwordx4 %0, %1, off\n global_store_dwordx4 %0, %1, off offset:1024" ::"v"(p), "v"(v) : "memory");
            else asm volatile("global_store_dwordx4 %0, %1, off sc1\n global_store_dwordx4 %0, %1, off offset:1024 sc1" ::"v"(p), "v"(v) : "memory");
        }
        const long long t0 = clock64();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const long long t1 = clock64();
        if (lane == 0) {
            out[200] = t1 - t0;
            __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else {
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) __builtin_amdgcn_s_sleep(1);
        double acc = 0;
        for (int rep = 0; rep < 2; ++rep)
            for (int t = 0; t < NT; ++t) {
                const char *p = (const char *)buf + t * 2048 + lane * 16;
                typedef double dv2 __attribute__((ext_vector_type(2))); dv2 a, b;
                const long long t0 = clock64();
                if (LD == 0) asm volatile("global_load_dwordx4 %0, %2, off\n global_load_dwordx4 %1, %2, off offset:1024\n s_waitcnt vmcnt(0)" : "=&v"(a), "=&v"(b) : "v"(p) : "memory");
                else asm volatile("global_load_dwordx4 %0, %2, off sc1\n global_load_dwordx4 %1, %2, off offset:1024 sc1\n s_waitcnt vmcnt(0)" : "=&v"(a), "=&v"(b) : "v"(p) : "memory");
                const long long t1 = clock64();
                acc += a[0] + b[1];
                if (lane == 0) out[rep * NT + t] = t1 - t0;
            }
        if (acc == 12345.678) out[199] = 1;
    }
}
template <int ST, int LD>
void run(const char *name) {
    double *buf;
    int *flag, *xcc, hx[2];
    long long *out, h[256];
    hipMalloc(&buf, NT * 2048), hipMalloc(&flag, 4), hipMalloc(&out, 256 * 8), hipMalloc(&xcc, 8);
    hipMemset(buf, 0, NT * 2048), hipMemset(flag, 0, 4), hipMemset(out, 0, 256 * 8);
    k<ST, LD><<<16, 64>>>(buf, flag, out, xcc);
    hipDeviceSynchronize();
    hipMemcpy(h, out, 256 * 8, hipMemcpyDeviceToHost), hipMemcpy(hx, xcc, 8, hipMemcpyDeviceToHost);
    double s1 = 0, s2 = 0;
    for (int t = 8; t < NT; ++t) s1 += h[t], s2 += h[NT + t];
    printf("%-46s first read %6.0f ns, second read %6.0f ns per tile; producer's wait for its stores %6.0f ns (XCC %d / %d)\n", name,
           s1 / (NT - 8) * 10.0, s2 / (NT - 8) * 10.0, h[200] * 10.0, hx[0], hx[1]);
    hipFree(buf), hipFree(flag), hipFree(out), hipFree(xcc);
}
int main() {
    for (int i = 0; i < 2; ++i) {
        run<0, 1>("plain stores, device-scope (sc1) loads");
        run<1, 1>("device-scope (sc1) stores, sc1 loads");
        run<0, 0>("plain stores, plain loads (cold L1)");
        run<1, 0>("sc1 stores, plain loads (cold L1)");
    }
    return 0;
}
