// Issue cost / latency of the instructions of the tile routine, one wave alone on a CU (cycles per instruction).
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
template <int K>
__global__ void __launch_bounds__(64) k(double *o, long long *cyc) {
    double a0 = o[threadIdx.x], a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = 1.0000001, c = 0.5;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < 16; ++it) {
        if constexpr (K == 0) asm volatile(REP64("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
        if constexpr (K == 1) asm volatile(REP64("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %0, %0, %4, %5\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
        if constexpr (K == 2) asm volatile(REP64("v_fmac_f64_dpp %0, -%0, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %1, -%1, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %2, -%2, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %3, -%3, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
        if constexpr (K == 3) asm volatile(REP64("v_rcp_f64 %0, %4\n v_rcp_f64 %1, %4\n v_rcp_f64 %2, %4\n v_rcp_f64 %3, %4\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
        if constexpr (K == 4) asm volatile(REP64("v_rcp_f64 %0, %0\n v_rcp_f64 %0, %0\n v_rcp_f64 %0, %0\n v_rcp_f64 %0, %0\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
        if constexpr (K == 5) asm volatile(REP64("v_mov_b64_dpp %0, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp %1, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp %2, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp %3, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
        if constexpr (K == 6) asm volatile(REP64("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
        if constexpr (K == 7) asm volatile(REP64("v_mov_b64 %0, %1\n v_mov_b64 %1, %2\n v_mov_b64 %2, %3\n v_mov_b64 %3, %0\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
        if constexpr (K == 8) asm volatile(REP64("v_fmac_f64_dpp %0, -%0, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n s_nop 0\n s_nop 0\n v_mov_b64_dpp %1, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
        if constexpr (K == 9) asm volatile(REP64("v_mul_f64 %0, %0, %4\n v_mul_f64 %0, %0, %4\n v_add_f64 %0, %0, %4\n v_add_f64 %0, %0, %4\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
        if constexpr (K == 10) asm volatile(REP64("s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
        if constexpr (K == 11) asm volatile(REP64("v_fma_f64 %0, %0, %4, %5\n s_nop 0\n v_fma_f64 %0, %0, %4, %5\n s_nop 0\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
        if constexpr (K == 12) asm volatile(REP64("v_fma_f64 %0, %0, %4, %5\n v_mov_b64 %1, %2\n v_fma_f64 %0, %0, %4, %5\n v_mov_b64 %2, %1\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
    }
    long long t1 = __builtin_readcyclecounter();
    o[threadIdx.x] = a0 + a1 + a2 + a3;
    if (threadIdx.x == 0) cyc[K] = t1 - t0;
}
int main() {
    double *o;
    long long *c, h[16];
    hipMalloc(&o, 64 * 8), hipMalloc(&c, 16 * 8);
    hipMemset(o, 0, 64 * 8);
#define RUN(K) k<K><<<1, 64>>>(o, c); k<K><<<1, 64>>>(o, c);
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12)
    hipDeviceSynchronize();
    hipMemcpy(h, c, 16 * 8, hipMemcpyDeviceToHost);
    const char *names[] = {"v_fma_f64 independent", "v_fma_f64 dependent", "v_fmac_f64_dpp independent", "v_rcp_f64 independent", "v_rcp_f64 dependent",
                           "v_mov_b64_dpp independent", "v_mul/add_f64 independent", "v_mov_b32 (dependent ring)", "fmac_dpp -> nop nop -> mov_dpp (4 instr)", "v_mul/add_f64 dependent", "s_nop 0", "fma, nop (dependent fma)", "fma, mov_b32 (dependent fma)"};
    for (int i = 0; i < 13; ++i) printf("%-45s %6.2f cycles per instruction\n", names[i], (double)h[i] / (16.0 * 64 * 4));
    return 0;
}
