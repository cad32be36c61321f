// Which CUs does a CU-masked HIP stream really use on MI355X?  (development probe for the pipeline's CU partition)
// Each workgroup records (XCC_ID, HW_ID) and spins long enough that all resident slots fill; the host counts the distinct
// (xcc, se, sh, cu) tuples per mask.   hipcc --offload-arch=gfx950 -O2 cu_mask_probe.hip -o cu_mask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <set>
#include <map>
#include <vector>

__global__ void probe(uint32_t *out, long long spin) {
    const long long t0 = clock64();
    while (clock64() - t0 < spin) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // HW_REG_XCC_ID
        out[2 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg((31 << 11) | 4);  // HW_REG_HW_ID
    }
}

static void run(const char *name, const std::vector<uint32_t> &mask) {
    hipStream_t s;
    hipError_t e = mask.empty() ? hipStreamCreateWithFlags(&s, hipStreamNonBlocking)
                                : hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data());
    if (e != hipSuccess) { printf("%-28s stream creation failed: %s\n", name, hipGetErrorString(e)); return; }
    const int G = 4096;
    uint32_t *d;
    hipMalloc(&d, 2 * G * sizeof(uint32_t));
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a, s);
    hipLaunchKernelGGL(probe, dim3(G), dim3(1024), 65536, s, d, 200000LL);  // 1024 threads + 64 KB LDS: 2 per CU
    hipEventRecord(b, s);
    hipStreamSynchronize(s);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    std::vector<uint32_t> h(2 * G);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    std::map<int, std::set<uint32_t>> per;
    for (int i = 0; i < G; ++i) {
        const int xcc = h[2 * i] & 0xf;
        const uint32_t hw = h[2 * i + 1];
        per[xcc].insert((hw >> 8) & 0xff);  // cu_id[11:8], sh_id[12], se_id[15:13]
    }
    int tot = 0;
    printf("%-28s %.2f ms  CUs per XCC:", name, ms);
    for (auto &kv : per) { printf(" %d:%zu", kv.first, kv.second.size()); tot += (int)kv.second.size(); }
    printf("  total %d\n", tot);
    if (tot <= 16) {
        printf("      (xcc: se/sh/cu):");
        for (auto &kv : per) for (auto v : kv.second) printf(" %d:%u/%u/%u", kv.first, (v >> 5) & 7, (v >> 4) & 1, v & 15);
        printf("\n");
    }
    hipFree(d);
    hipStreamDestroy(s);
}

// a bin_gram-shaped grid: G workgroups of 768 threads with 158 KB of LDS (one per CU), each spinning ~1 ms: the elapsed
// time tells whether all of them were resident at once
static void run_big(const char *name, const std::vector<uint32_t> &mask, int G) {
    hipStream_t s;
    hipError_t e = mask.empty() ? hipStreamCreateWithFlags(&s, hipStreamNonBlocking)
                                : hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data());
    if (e != hipSuccess) { printf("%-28s stream creation failed\n", name); return; }
    uint32_t *d;
    hipMalloc(&d, 2 * G * sizeof(uint32_t));
    hipFuncSetAttribute(reinterpret_cast<const void *>(&probe), hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(a, s);
        hipLaunchKernelGGL(probe, dim3(G), dim3(768), 158 * 1024, s, d, 2400000LL);
        hipEventRecord(b, s);
        hipStreamSynchronize(s);
    }
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    std::vector<uint32_t> h(2 * G);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    std::map<int, int> per;
    for (int i = 0; i < G; ++i) per[h[2 * i] & 0xf]++;
    printf("%-28s G=%d  %.2f ms (1.0 = one round)  workgroups per XCC:", name, G, ms);
    for (auto &kv : per) printf(" %d:%d", kv.first, kv.second);
    printf("\n      SE of the successive workgroups of XCC 0:");
    for (int i = 0; i < G; ++i) if ((h[2 * i] & 0xf) == 0) printf(" %u", (h[2 * i + 1] >> 13) & 7);
    printf("\n      XCC of workgroups 0..15:");
    for (int i = 0; i < 16 && i < G; ++i) printf(" %u", h[2 * i] & 0xf);
    printf("\n");
    hipFree(d);
    hipStreamDestroy(s);
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int ncu = p.multiProcessorCount;
    printf("multiProcessorCount %d\n", ncu);
    auto bits = [&](int n, auto pred) { std::vector<uint32_t> m((n + 31) / 32, 0u); for (int b = 0; b < n; ++b) if (pred(b)) m[b >> 5] |= 1u << (b & 31); return m; };
    auto m240 = bits(256, [](int b) { return b < 240; });
    run_big("no mask", {}, 256);
    run_big("no mask", {}, 239);
    run_big("all but last 16", m240, 239);
    run_big("all but last 16", m240, 240);
    run_big("all but last 16", m240, 232);
    run_big("all but last 16", m240, 200);
    run_big("all but last 16", m240, 128);
    run("no mask", {});
    run("256 ones", bits(256, [](int) { return true; }));
    run("first 128", bits(256, [](int b) { return b < 128; }));
    run("first 32", bits(256, [](int b) { return b < 32; }));
    run("bits 0..7", bits(256, [](int b) { return b < 8; }));
    run("bit 0", bits(256, [](int b) { return b == 0; }));
    run("bit 1", bits(256, [](int b) { return b == 1; }));
    run("bit 8", bits(256, [](int b) { return b == 8; }));
    run("bit 32", bits(256, [](int b) { return b == 32; }));
    run("all but last 16", bits(256, [](int b) { return b < 240; }));
    run("xcd-major minus 2", bits(256, [](int b) { return (b % 32) < 30; }));
    run("last 16", bits(256, [](int b) { return b >= 240; }));
    run("xcd-major last 2", bits(256, [](int b) { return (b % 32) >= 30; }));
    run("320 ones", bits(320, [](int) { return true; }));
    run("288 ones", bits(288, [](int) { return true; }));
    return 0;
}
