// What does one tile product of the register-resident fit loop cost a wave?  A loop of tile products T[s] += A_s B, A_s read from
// LDS one product ahead (two ds_read_b128), in the variants the kernel could use.  Cycles per tile per wave, 1 and 2 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o tile_step_bench tile_step_bench.hip && ./tile_step_bench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4f64 __attribute__((ext_vector_type(4)));
typedef double v2f64 __attribute__((ext_vector_type(2)));

#define ISSUE(lo, hi, addr) asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024" : "=&v"(lo), "=&v"(hi) : "v"(addr) : "memory")

template <int VAR>
__global__ void bench(long long *out, int iters, double seed) {
    extern __shared__ double lds[];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = seed * (i & 15);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    v4f64 T[8];
    for (int c = 0; c < 8; ++c) T[c] = v4f64{seed, seed, seed, seed};
    double b0 = seed, b1 = seed * 2, b2 = seed * 3, b3 = seed * 4;
    unsigned base = (unsigned)(size_t)lds + lane * 16u;
    v2f64 lo[2], hi[2];
    ISSUE(lo[0], hi[0], base);
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const unsigned nxt = base + (((i + s + 1) & 7u) << 11);
            if (VAR == 0 || VAR == 1) {  // the kernel's form: issue next, wait, 4 products (+ 18 wait states: VAR 0)
                ISSUE(lo[(s + 1) & 1], hi[(s + 1) & 1], nxt);
                asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
                if (VAR == 0)
                    asm volatile("s_nop 1\n\tv_mfma_f64_16x16x4_f64 %0, %1, %5, %0\n\tv_mfma_f64_16x16x4_f64 %0, %2, %6, %0\n\t"
                                 "v_mfma_f64_16x16x4_f64 %0, %3, %7, %0\n\tv_mfma_f64_16x16x4_f64 %0, %4, %8, %0\n\ts_nop 15\n\ts_nop 2"
                                 : "+v"(T[s]) : "v"(lo[s & 1][0]), "v"(lo[s & 1][1]), "v"(hi[s & 1][0]), "v"(hi[s & 1][1]), "v"(b0), "v"(b1), "v"(b2), "v"(b3));
                else
                    asm volatile("s_nop 1\n\tv_mfma_f64_16x16x4_f64 %0, %1, %5, %0\n\tv_mfma_f64_16x16x4_f64 %0, %2, %6, %0\n\t"
                                 "v_mfma_f64_16x16x4_f64 %0, %3, %7, %0\n\tv_mfma_f64_16x16x4_f64 %0, %4, %8, %0"
                                 : "+v"(T[s]) : "v"(lo[s & 1][0]), "v"(lo[s & 1][1]), "v"(hi[s & 1][0]), "v"(hi[s & 1][1]), "v"(b0), "v"(b1), "v"(b2), "v"(b3));
            } else if (VAR == 2) {  // the next operand's reads BETWEEN the products, no trailing wait states
                asm volatile("s_waitcnt lgkmcnt(0)\n\tv_mfma_f64_16x16x4_f64 %0, %3, %7, %0\n\tds_read_b128 %1, %11\n\t"
                             "v_mfma_f64_16x16x4_f64 %0, %4, %8, %0\n\tds_read_b128 %2, %11 offset:1024\n\t"
                             "v_mfma_f64_16x16x4_f64 %0, %5, %9, %0\n\tv_mfma_f64_16x16x4_f64 %0, %6, %10, %0"
                             : "+v"(T[s]), "=&v"(lo[(s + 1) & 1]), "=&v"(hi[(s + 1) & 1])
                             : "v"(lo[s & 1][0]), "v"(lo[s & 1][1]), "v"(hi[s & 1][0]), "v"(hi[s & 1][1]), "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(nxt) : "memory");
            } else if (VAR == 4) {  // one operand read per TWO products (register blocking: two rows share the tile)
                if ((s & 1) == 0) {
                    ISSUE(lo[1], hi[1], nxt);
                    asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
                }
                asm volatile("s_nop 1\n\tv_mfma_f64_16x16x4_f64 %0, %1, %5, %0\n\tv_mfma_f64_16x16x4_f64 %0, %2, %6, %0\n\t"
                             "v_mfma_f64_16x16x4_f64 %0, %3, %7, %0\n\tv_mfma_f64_16x16x4_f64 %0, %4, %8, %0"
                             : "+v"(T[s]) : "v"(lo[0][0]), "v"(lo[0][1]), "v"(hi[0][0]), "v"(hi[0][1]), "v"(b0), "v"(b1), "v"(b2), "v"(b3));
                if ((s & 1) == 1) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    lo[0] = lo[1], hi[0] = hi[1];
                }
            } else if (VAR == 5) {  // four ds_read_b64 per operand instead of two ds_read_b128
                double o0, o1, o2, o3;
                asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:512\n\tds_read_b64 %2, %4 offset:1024\n\tds_read_b64 %3, %4 offset:1536\n\t"
                             "s_waitcnt lgkmcnt(0)" : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3) : "v"(nxt - lane * 8u) : "memory");
                asm volatile("s_nop 1\n\tv_mfma_f64_16x16x4_f64 %0, %1, %5, %0\n\tv_mfma_f64_16x16x4_f64 %0, %2, %6, %0\n\t"
                             "v_mfma_f64_16x16x4_f64 %0, %3, %7, %0\n\tv_mfma_f64_16x16x4_f64 %0, %4, %8, %0"
                             : "+v"(T[s]) : "v"(o0), "v"(o1), "v"(o2), "v"(o3), "v"(b0), "v"(b1), "v"(b2), "v"(b3));
            } else if (VAR == 6) {  // wait states BETWEEN the products
                ISSUE(lo[(s + 1) & 1], hi[(s + 1) & 1], nxt);
                asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
                asm volatile("s_nop 1\n\tv_mfma_f64_16x16x4_f64 %0, %1, %5, %0\n\ts_nop 7\n\tv_mfma_f64_16x16x4_f64 %0, %2, %6, %0\n\ts_nop 7\n\t"
                             "v_mfma_f64_16x16x4_f64 %0, %3, %7, %0\n\ts_nop 7\n\tv_mfma_f64_16x16x4_f64 %0, %4, %8, %0\n\ts_nop 7"
                             : "+v"(T[s]) : "v"(lo[s & 1][0]), "v"(lo[s & 1][1]), "v"(hi[s & 1][0]), "v"(hi[s & 1][1]), "v"(b0), "v"(b1), "v"(b2), "v"(b3));
            } else if (VAR == 7) {  // two tiles per burst: four reads ahead, eight products
                if ((s & 1) == 0) {
                    v2f64 l2, h2, l3, h3;
                    ISSUE(l2, h2, nxt);
                    ISSUE(l3, h3, nxt ^ 2048u);
                    asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
                    asm volatile("s_nop 1\n\tv_mfma_f64_16x16x4_f64 %0, %2, %10, %0\n\tv_mfma_f64_16x16x4_f64 %1, %6, %10, %1\n\t"
                                 "v_mfma_f64_16x16x4_f64 %0, %3, %11, %0\n\tv_mfma_f64_16x16x4_f64 %1, %7, %11, %1\n\t"
                                 "v_mfma_f64_16x16x4_f64 %0, %4, %12, %0\n\tv_mfma_f64_16x16x4_f64 %1, %8, %12, %1\n\t"
                                 "v_mfma_f64_16x16x4_f64 %0, %5, %13, %0\n\tv_mfma_f64_16x16x4_f64 %1, %9, %13, %1"
                                 : "+v"(T[s]), "+v"(T[s + 1])
                                 : "v"(lo[0][0]), "v"(lo[0][1]), "v"(hi[0][0]), "v"(hi[0][1]), "v"(lo[1][0]), "v"(lo[1][1]), "v"(hi[1][0]), "v"(hi[1][1]), "v"(b0), "v"(b1), "v"(b2), "v"(b3));
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    lo[0] = l2, hi[0] = h2, lo[1] = l3, hi[1] = h3;
                }
            } else if (VAR == 8) {  // priority: the wave raises its priority while it has products to issue
                ISSUE(lo[(s + 1) & 1], hi[(s + 1) & 1], nxt);
                asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
                asm volatile("s_setprio 3\n\tv_mfma_f64_16x16x4_f64 %0, %1, %5, %0\n\tv_mfma_f64_16x16x4_f64 %0, %2, %6, %0\n\t"
                             "v_mfma_f64_16x16x4_f64 %0, %3, %7, %0\n\tv_mfma_f64_16x16x4_f64 %0, %4, %8, %0\n\ts_setprio 0"
                             : "+v"(T[s]) : "v"(lo[s & 1][0]), "v"(lo[s & 1][1]), "v"(hi[s & 1][0]), "v"(hi[s & 1][1]), "v"(b0), "v"(b1), "v"(b2), "v"(b3));
            } else if (VAR == 9) {  // no read ahead: read, wait for it, four products
                ISSUE(lo[0], hi[0], nxt);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                asm volatile("s_nop 1\n\tv_mfma_f64_16x16x4_f64 %0, %1, %5, %0\n\tv_mfma_f64_16x16x4_f64 %0, %2, %6, %0\n\t"
                             "v_mfma_f64_16x16x4_f64 %0, %3, %7, %0\n\tv_mfma_f64_16x16x4_f64 %0, %4, %8, %0"
                             : "+v"(T[s]) : "v"(lo[0][0]), "v"(lo[0][1]), "v"(hi[0][0]), "v"(hi[0][1]), "v"(b0), "v"(b1), "v"(b2), "v"(b3));
            } else if (VAR >= 10 && VAR <= 13) {
                if (VAR == 10) asm volatile("s_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0");
                if (VAR == 11) asm volatile("v_mov_b64 %0, %0\n\tv_mov_b64 %1, %1" : "+v"(b0), "+v"(b1));
                if (VAR == 12) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (VAR == 13) {
                    ISSUE(lo[(s + 1) & 1], hi[(s + 1) & 1], nxt);
                    if (s == 7) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %5, %0\n\tv_mfma_f64_16x16x4_f64 %0, %2, %6, %0\n\t"
                             "v_mfma_f64_16x16x4_f64 %0, %3, %7, %0\n\tv_mfma_f64_16x16x4_f64 %0, %4, %8, %0"
                             : "+v"(T[s]) : "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(b0), "v"(b1), "v"(b2), "v"(b3));
            } else if (VAR >= 20 && VAR <= 25) {  // four products, THEN the next operand's reads, then a pause
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 1\n\tv_mfma_f64_16x16x4_f64 %0, %1, %5, %0\n\tv_mfma_f64_16x16x4_f64 %0, %2, %6, %0\n\t"
                             "v_mfma_f64_16x16x4_f64 %0, %3, %7, %0\n\tv_mfma_f64_16x16x4_f64 %0, %4, %8, %0"
                             : "+v"(T[s]) : "v"(lo[s & 1][0]), "v"(lo[s & 1][1]), "v"(hi[s & 1][0]), "v"(hi[s & 1][1]), "v"(b0), "v"(b1), "v"(b2), "v"(b3));
                ISSUE(lo[(s + 1) & 1], hi[(s + 1) & 1], nxt);
                if (VAR == 20) asm volatile("s_nop 15\n\ts_nop 2");
                if (VAR == 21) asm volatile("s_nop 15\n\ts_nop 15");
                if (VAR == 22) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15");
                if (VAR == 23) asm volatile("s_sleep 1");
                if (VAR == 24) asm volatile("s_sleep 2");
            } else if (VAR >= 30 && VAR <= 36) {  // read ahead, wait, four products, a tail of wait states of varying length
                ISSUE(lo[(s + 1) & 1], hi[(s + 1) & 1], nxt);
                asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
                asm volatile("s_nop 1\n\tv_mfma_f64_16x16x4_f64 %0, %1, %5, %0\n\tv_mfma_f64_16x16x4_f64 %0, %2, %6, %0\n\t"
                             "v_mfma_f64_16x16x4_f64 %0, %3, %7, %0\n\tv_mfma_f64_16x16x4_f64 %0, %4, %8, %0"
                             : "+v"(T[s]) : "v"(lo[s & 1][0]), "v"(lo[s & 1][1]), "v"(hi[s & 1][0]), "v"(hi[s & 1][1]), "v"(b0), "v"(b1), "v"(b2), "v"(b3));
                if (VAR == 30) asm volatile("s_nop 5");
                if (VAR == 31) asm volatile("s_nop 11");
                if (VAR == 32) asm volatile("s_nop 15\n\ts_nop 7");
                if (VAR == 33) asm volatile("s_nop 15\n\ts_nop 15");
                if (VAR == 34) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15");
                if (VAR == 35) asm volatile("s_sleep 1");
                if (VAR == 36) asm volatile("s_nop 15\n\ts_nop 2\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0");
            } else if (VAR == 40 || VAR == 41) {  // ONE operand for TWO tile products (no copies): two rows of a wave share the column
                if ((s & 1) == 0) {
                    ISSUE(lo[((s >> 1) + 1) & 1], hi[((s >> 1) + 1) & 1], nxt);
                    asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
                    asm volatile("s_nop 1\n\tv_mfma_f64_16x16x4_f64 %0, %2, %6, %0\n\tv_mfma_f64_16x16x4_f64 %1, %2, %10, %1\n\t"
                                 "v_mfma_f64_16x16x4_f64 %0, %3, %7, %0\n\tv_mfma_f64_16x16x4_f64 %1, %3, %11, %1\n\t"
                                 "v_mfma_f64_16x16x4_f64 %0, %4, %8, %0\n\tv_mfma_f64_16x16x4_f64 %1, %4, %12, %1\n\t"
                                 "v_mfma_f64_16x16x4_f64 %0, %5, %9, %0\n\tv_mfma_f64_16x16x4_f64 %1, %5, %13, %1"
                                 : "+v"(T[s]), "+v"(T[s + 1])
                                 : "v"(lo[(s >> 1) & 1][0]), "v"(lo[(s >> 1) & 1][1]), "v"(hi[(s >> 1) & 1][0]), "v"(hi[(s >> 1) & 1][1]), "v"(b0), "v"(b1),
                                   "v"(b2), "v"(b3), "v"(b3), "v"(b2), "v"(b1), "v"(b0));
                    if (VAR == 41) asm volatile("s_nop 15\n\ts_nop 2");
                }
            } else if (VAR == 3) {  // products only (no LDS)
                asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %5, %0\n\tv_mfma_f64_16x16x4_f64 %0, %2, %6, %0\n\t"
                             "v_mfma_f64_16x16x4_f64 %0, %3, %7, %0\n\tv_mfma_f64_16x16x4_f64 %0, %4, %8, %0"
                             : "+v"(T[s]) : "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(b0), "v"(b1), "v"(b2), "v"(b3));
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 3");
    const long long t1 = clock64();
    double s = lo[0][0] + hi[1][1];
    for (int c = 0; c < 8; ++c) s += T[c][0] + T[c][3];
    if (lane == 0) out[threadIdx.x / 64] = t1 - t0;
    if (s == 12345.678) out[63] = 1;
}

template <int VAR>
void run(int threads, long long *d, const char *what) {
    const int iters = 500;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(bench<VAR>, dim3(1), dim3(threads), 65536, 0, d, iters, 1e-3);
    long long h[16];
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("%2d waves, %-58s %.0f cycles per tile per wave\n", threads / 64, what, (double)h[0] / (iters * 8.0));
}

int main() {
    long long *d;
    (void)hipMalloc(&d, 64 * sizeof(long long));
    for (int threads : {64, 256, 512, 768, 1024}) {
        run<3>(threads, d, "four products, no LDS:");
        run<1>(threads, d, "read ahead, wait, four products:");
        run<0>(threads, d, "read ahead, wait, four products, 18 wait states:");
        run<2>(threads, d, "reads between the products:");
        run<30>(threads, d, "read ahead, wait, products, 6 wait states:");
        run<31>(threads, d, "read ahead, wait, products, 12 wait states:");
        run<32>(threads, d, "read ahead, wait, products, 24 wait states:");
        run<33>(threads, d, "read ahead, wait, products, 32 wait states:");
        run<34>(threads, d, "read ahead, wait, products, 48 wait states:");
        run<35>(threads, d, "read ahead, wait, products, s_sleep 1:");
        run<36>(threads, d, "read ahead, wait, products, 18 wait states + 10 scalar:");
        run<40>(threads, d, "ONE operand read for TWO tile products:");
        run<41>(threads, d, "ONE operand read for TWO tile products, 18 wait states:");
        run<20>(threads, d, "products, reads, 18 wait states:");
        run<21>(threads, d, "products, reads, 32 wait states:");
        run<22>(threads, d, "products, reads, 64 wait states:");
        run<23>(threads, d, "products, reads, s_sleep 1:");
        run<24>(threads, d, "products, reads, s_sleep 2:");
        run<25>(threads, d, "products, reads:");
        run<10>(threads, d, "no LDS, 8 s_nop per tile:");
        run<11>(threads, d, "no LDS, 2 v_mov per tile:");
        run<12>(threads, d, "no LDS, s_waitcnt per tile:");
        run<13>(threads, d, "reads issued, operands unused, one wait per 8 tiles:");
        run<6>(threads, d, "8 wait states behind every product:");
        run<7>(threads, d, "two tiles per burst (four reads, eight products):");
        run<8>(threads, d, "s_setprio 3 around the products:");
        run<9>(threads, d, "no read ahead (read, wait, four products):");
        run<5>(threads, d, "four ds_read_b64 (not ahead), wait, four products:");
    }
    return 0;
}
