// Microbenchmark: does a lone wave pay for back-to-back DEPENDENT VALU instructions on gfx950, and what do the kernel's
// own instruction patterns (the DP cell chain, the key multiply-adds, the SDWA score adds) cost with 1 and 2 waves per SIMD?
// Prints cycles per instruction per wave.  Developer tool:  hipcc -O2 --offload-arch=gfx950 dep_chain.hip -o /tmp/dep_chain
#include <hip/hip_runtime.h>
#include <cstdio>

#define ITER 2048
template <int OP>
__global__ void __launch_bounds__(256) k(unsigned* out, unsigned a, unsigned b)
{
    unsigned v0 = threadIdx.x, v1 = threadIdx.x * 3 + a, v2 = threadIdx.x * 5 + b, v3 = threadIdx.x * 7, v4 = v0 ^ a, v5 = v1 ^ b, v6 = v2 + 1, v7 = v3 + 2;
    unsigned t0 = v0 + 11, t1 = v1 + 12, t2 = v2 + 13, t3 = v3 + 14, f0 = v4, f1 = v5, f2 = v6, f3 = v7, ev = a, ev2 = b, x, u, x2, u2;
    for (int it = 0; it < ITER; it++) {
        if (OP == 0)          // 16 instructions, ONE dependent chain
            asm volatile("v_pk_max_u16 %0, %0, %1\n\tv_pk_max_u16 %0, %0, %2\n\tv_pk_max_u16 %0, %0, %1\n\tv_pk_max_u16 %0, %0, %2\n\t"
                         "v_pk_max_u16 %0, %0, %1\n\tv_pk_max_u16 %0, %0, %2\n\tv_pk_max_u16 %0, %0, %1\n\tv_pk_max_u16 %0, %0, %2\n\t"
                         "v_pk_max_u16 %0, %0, %1\n\tv_pk_max_u16 %0, %0, %2\n\tv_pk_max_u16 %0, %0, %1\n\tv_pk_max_u16 %0, %0, %2\n\t"
                         "v_pk_max_u16 %0, %0, %1\n\tv_pk_max_u16 %0, %0, %2\n\tv_pk_max_u16 %0, %0, %1\n\tv_pk_max_u16 %0, %0, %2" : "+v"(v0) : "v"(a), "v"(b));
        if (OP == 1)          // 16 instructions, TWO chains interleaved
            asm volatile("v_pk_max_u16 %0, %0, %2\n\tv_pk_max_u16 %1, %1, %3\n\tv_pk_max_u16 %0, %0, %3\n\tv_pk_max_u16 %1, %1, %2\n\t"
                         "v_pk_max_u16 %0, %0, %2\n\tv_pk_max_u16 %1, %1, %3\n\tv_pk_max_u16 %0, %0, %3\n\tv_pk_max_u16 %1, %1, %2\n\t"
                         "v_pk_max_u16 %0, %0, %2\n\tv_pk_max_u16 %1, %1, %3\n\tv_pk_max_u16 %0, %0, %3\n\tv_pk_max_u16 %1, %1, %2\n\t"
                         "v_pk_max_u16 %0, %0, %2\n\tv_pk_max_u16 %1, %1, %3\n\tv_pk_max_u16 %0, %0, %3\n\tv_pk_max_u16 %1, %1, %2" : "+v"(v0), "+v"(v1) : "v"(a), "v"(b));
        if (OP == 2)          // 16 instructions, FOUR chains
            asm volatile("v_pk_max_u16 %0, %0, %4\n\tv_pk_max_u16 %1, %1, %5\n\tv_pk_max_u16 %2, %2, %4\n\tv_pk_max_u16 %3, %3, %5\n\t"
                         "v_pk_max_u16 %0, %0, %5\n\tv_pk_max_u16 %1, %1, %4\n\tv_pk_max_u16 %2, %2, %5\n\tv_pk_max_u16 %3, %3, %4\n\t"
                         "v_pk_max_u16 %0, %0, %4\n\tv_pk_max_u16 %1, %1, %5\n\tv_pk_max_u16 %2, %2, %4\n\tv_pk_max_u16 %3, %3, %5\n\t"
                         "v_pk_max_u16 %0, %0, %5\n\tv_pk_max_u16 %1, %1, %4\n\tv_pk_max_u16 %2, %2, %5\n\tv_pk_max_u16 %3, %3, %4" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(a), "v"(b));
        if (OP == 3) {        // the kernel's cell chain: 4 cells x 5 instructions, as in row_cells4 (20 instructions)
#define CELL(T, F) "v_pk_max_u16 %[x], " T ", " F "\n\tv_pk_sub_u16 %[u], " T ", %[g]\n\tv_pk_max_u16 " T ", %[x], %[ev]\n\tv_pk_max_u16 " F ", %[u], " F "\n\tv_pk_max_u16 %[ev], %[u], %[ev]\n\t"
            asm volatile(CELL("%[t0]", "%[f0]") CELL("%[t1]", "%[f1]") CELL("%[t2]", "%[f2]") CELL("%[t3]", "%[f3]")
                         : [t0] "+v"(t0), [t1] "+v"(t1), [t2] "+v"(t2), [t3] "+v"(t3), [f0] "+v"(f0), [f1] "+v"(f1), [f2] "+v"(f2), [f3] "+v"(f3), [ev] "+v"(ev), [x] "=&v"(x), [u] "=&v"(u) : [g] "s"(a));
        }
        if (OP == 4) {        // two independent cell chains interleaved instruction by instruction (40 instructions)
#define CELL2(T, F, T2, F2) "v_pk_max_u16 %[x], " T ", " F "\n\tv_pk_max_u16 %[x2], " T2 ", " F2 "\n\tv_pk_sub_u16 %[u], " T ", %[g]\n\tv_pk_sub_u16 %[u2], " T2 ", %[g]\n\t" \
                            "v_pk_max_u16 " T ", %[x], %[ev]\n\tv_pk_max_u16 " T2 ", %[x2], %[ev2]\n\tv_pk_max_u16 " F ", %[u], " F "\n\tv_pk_max_u16 " F2 ", %[u2], " F2 "\n\t" \
                            "v_pk_max_u16 %[ev], %[u], %[ev]\n\tv_pk_max_u16 %[ev2], %[u2], %[ev2]\n\t"
            asm volatile(CELL2("%[t0]", "%[f0]", "%[v0]", "%[v4]") CELL2("%[t1]", "%[f1]", "%[v1]", "%[v5]") CELL2("%[t2]", "%[f2]", "%[v2]", "%[v6]") CELL2("%[t3]", "%[f3]", "%[v3]", "%[v7]")
                         : [t0] "+v"(t0), [t1] "+v"(t1), [t2] "+v"(t2), [t3] "+v"(t3), [f0] "+v"(f0), [f1] "+v"(f1), [f2] "+v"(f2), [f3] "+v"(f3), [ev] "+v"(ev), [x] "=&v"(x), [u] "=&v"(u),
                           [v0] "+v"(v0), [v1] "+v"(v1), [v2] "+v"(v2), [v3] "+v"(v3), [v4] "+v"(v4), [v5] "+v"(v5), [v6] "+v"(v6), [v7] "+v"(v7), [ev2] "+v"(ev2), [x2] "=&v"(x2), [u2] "=&v"(u2) : [g] "s"(a));
        }
        if (OP == 5)          // the key pattern of row_keys4: 8 mads + 4 max3 (12 instructions)
            asm volatile("v_mad_u32_u16 %[x], %[h0], %[km], %[rl]\n\tv_mad_u32_u16 %[u], %[h0], %[km], %[rh] op_sel:[1,1,0,0]\n\t"
                         "v_mad_u32_u16 %[x2], %[h1], %[km], %[rl]\n\tv_mad_u32_u16 %[u2], %[h1], %[km], %[rh] op_sel:[1,1,0,0]\n\tv_max3_i32 %[a0], %[a0], %[x], %[u]\n\t"
                         "v_mad_u32_u16 %[x], %[h2], %[km], %[rl]\n\tv_mad_u32_u16 %[u], %[h2], %[km], %[rh] op_sel:[1,1,0,0]\n\tv_max3_i32 %[a1], %[a1], %[x2], %[u2]\n\t"
                         "v_mad_u32_u16 %[x2], %[h3], %[km], %[rl]\n\tv_mad_u32_u16 %[u2], %[h3], %[km], %[rh] op_sel:[1,1,0,0]\n\tv_max3_i32 %[a2], %[a2], %[x], %[u]\n\tv_max3_i32 %[a3], %[a3], %[x2], %[u2]"
                         : [a0] "+v"(v0), [a1] "+v"(v1), [a2] "+v"(v2), [a3] "+v"(v3), [x] "=&v"(x), [u] "=&v"(u), [x2] "=&v"(x2), [u2] "=&v"(u2)
                         : [h0] "v"(t0), [h1] "v"(t1), [h2] "v"(t2), [h3] "v"(t3), [km] "v"(a), [rl] "v"(b), [rh] "v"(f0));
        if (OP == 6)          // the SDWA score adds of row_add_scores (16 instructions)
            asm volatile("v_add_u16_sdwa %7, %6, sext(%10) dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:BYTE_0\n\t"
        "v_add_u16_sdwa %6, %5, sext(%9) dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:BYTE_0\n\t"
        "v_add_u16_sdwa %5, %4, sext(%10) dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:BYTE_1\n\t"
        "v_add_u16_sdwa %4, %3, sext(%9) dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:BYTE_1\n\t"
        "v_add_u16_sdwa %3, %2, sext(%10) dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:BYTE_2\n\t"
        "v_add_u16_sdwa %2, %1, sext(%9) dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:BYTE_2\n\t"
        "v_add_u16_sdwa %1, %0, sext(%10) dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:BYTE_3\n\t"
        "v_add_u16_sdwa %0, %8, sext(%9) dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:BYTE_3\n\t"
        "v_add_u16_sdwa %7, %6, sext(%12) dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:BYTE_0\n\t"
        "v_add_u16_sdwa %6, %5, sext(%11) dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:BYTE_0\n\t"
        "v_add_u16_sdwa %5, %4, sext(%12) dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:BYTE_1\n\t"
        "v_add_u16_sdwa %4, %3, sext(%11) dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:BYTE_1\n\t"
        "v_add_u16_sdwa %3, %2, sext(%12) dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:BYTE_2\n\t"
        "v_add_u16_sdwa %2, %1, sext(%11) dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:BYTE_2\n\t"
        "v_add_u16_sdwa %1, %0, sext(%12) dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:BYTE_3\n\t"
        "v_add_u16_sdwa %0, %8, sext(%11) dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:BYTE_3"
        : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(t0), "v"(f0), "v"(f1), "v"(f2), "v"(f3));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 + t0 + t1 + t2 + t3 + f0 + f1 + f2 + f3 + ev + ev2;
}

template <int OP>
void run(const char* name, int ninstr, int waves_per_simd)
{
    int blocks = 256 * waves_per_simd;
    unsigned* d; hipMalloc(&d, sizeof(unsigned) * blocks * 256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 3u, 5u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 3u, 5u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double cyc = ms * 1e-3 * 2.4e9;                       // cycles of the kernel at 2.4 GHz
    printf("%-44s waves/SIMD=%d  %.2f cycles per instruction per wave  (SIMD issues one every %.2f cycles)\n", name, waves_per_simd,
           cyc / ((double)ITER * ninstr), cyc / ((double)ITER * ninstr * waves_per_simd));
    hipFree(d);
}

int main()
{
    for (int w : {1, 2, 4}) {
        run<0>("pk_max: one dependent chain", 16, w); run<1>("pk_max: two chains interleaved", 16, w); run<2>("pk_max: four chains interleaved", 16, w);
        run<3>("DP cell chain (row_cells4)", 20, w); run<4>("two DP cell chains interleaved", 40, w);
        run<5>("key mads + max3 (row_keys4)", 12, w); run<6>("SDWA score adds (row_add_scores)", 16, w);
    }
    return 0;
}
