// Microbenchmark: sustained issue rate of the VALU ops the alignment kernel is made of, on all CUs.
// Prints lane-ops/s per op; used to fix the "VALU roofline" in DESIGN.md / bench.py.  Developer tool.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define ITER 4096
template <int OP>
__global__ void __launch_bounds__(256) k(int* out, int a, int b)
{
    int v[8];
    for (int j = 0; j < 8; j++) v[j] = threadIdx.x + j * a;
    float f[8];
    unsigned long long m64 = __builtin_amdgcn_ballot_w64(threadIdx.x & 1), m2 = 0;
    for (int j = 0; j < 8; j++) f[j] = (float)v[j];
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[j]) : "v"(b));
            if (OP == 1) asm volatile("v_max_i32 %0, %0, %1" : "+v"(v[j]) : "v"(b));
            if (OP == 2) asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(b), "v"(a));
            if (OP == 3) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(v[j]) : "v"(b));
            if (OP == 4) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(v[j]) : "v"(b));
            if (OP == 5) asm volatile("v_pk_add_i16 %0, %0, %1" : "+v"(v[j]) : "v"(b));
            if (OP == 6) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[j]) : "v"((float)b));
            if (OP == 7) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[j]) : "v"(b) : );
            if (OP == 8) asm volatile("v_add3_u32 %0, %0, %1, 3" : "+v"(v[j]) : "v"(b));
            if (OP == 9) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(double*)&f[j & 6]) : "v"(*(double*)&f[(j & 6)]));
            if (OP == 10) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(v[j]) : "v"(b), "s"(m64));
            if (OP == 11) asm volatile("v_cmp_eq_u32 vcc, %0, %1" : : "v"(v[j]), "v"(b) : "vcc");
            if (OP == 12) asm volatile("v_cmp_eq_u32_e64 %0, %1, %2" : "=s"(m2) : "v"(v[j]), "v"(b));
            if (OP == 13) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(v[j]) : "v"(b));
            if (OP == 14) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(v[j]) : "v"(b));
            if (OP == 15) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(b), "v"(a));
            if (OP == 16) asm volatile("v_bfe_u32 %0, %0, 4, 4" : "+v"(v[j]));
            if (OP == 17) asm volatile("v_mov_b32 %0, %1" : "=v"(v[j]) : "v"(b));
            if (OP == 18) asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(v[j]) : "v"(b), "v"(a));
            if (OP == 19) asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "+v"(v[j]) : "v"(b));
            if (OP == 20) { asm volatile("v_cmp_eq_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[j]) : "v"(b) : "vcc"); }
            if (OP == 21) { asm volatile("v_add_u32 %0, %0, %1\n\tv_max_i32 %0, %0, %1" : "+v"(v[j]) : "v"(b)); }
            if (OP == 22) asm volatile("v_min_i32 %0, %0, %1" : "+v"(v[j]) : "v"(b));
            if (OP == 23) asm volatile("v_bfi_b32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(b), "v"(a));
            if (OP == 24) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(b), "v"(a));
            if (OP == 25) asm volatile("v_max_i32_dpp %0, %0, %1 row_ror:1 row_mask:0xf bank_mask:0xf" : "+v"(v[j]) : "v"(b));
            if (OP == 26) asm volatile("v_max_i16 %0, %0, %1" : "+v"(v[j]) : "v"(b));
            if (OP == 27) asm volatile("v_pk_sub_i16 %0, %0, %1" : "+v"(v[j]) : "v"(b));
            if (OP == 28) asm volatile("v_pk_mad_i16 %0, %0, %1, %2" : "+v"(v[j]) : "v"(b), "v"(a));
            if (OP == 29) asm volatile("v_subrev_u32 %0, %1, %0" : "+v"(v[j]) : "s"(b));
            if (OP == 30) { if (j & 1) asm volatile("v_max_i32 %0, %0, %1" : "+v"(v[j]) : "v"(b)); else asm volatile("v_sub_u32 %0, %0, %1" : "+v"(v[j]) : "v"(b)); }
            if (OP == 31) { if (j & 4) asm volatile("v_max_i32 %0, %0, %1" : "+v"(v[j]) : "v"(b)); else asm volatile("v_sub_u32 %0, %0, %1" : "+v"(v[j]) : "v"(b)); }
            if (OP == 32) { if (j == 7) asm volatile("s_and_b64 %0, %0, %1" : "+s"(m64) : "s"(m2)); else asm volatile("v_max_i32 %0, %0, %1" : "+v"(v[j]) : "v"(b)); }
            if (OP == 34) asm volatile("v_mad_u32_u16 %0, %1, %2, %0 op_sel:[1,0,0,0]" : "+v"(v[j]) : "v"(b), "v"(a));
            if (OP == 35) asm volatile("v_mad_i32_i16 %0, %1, %2, %0 op_sel:[1,0,0,0]" : "+v"(v[j]) : "v"(b), "v"(a));
            if (OP == 36) asm volatile("v_add_u16_sdwa %0, %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:BYTE_2" : "+v"(v[j]) : "v"(b));
            if (OP == 37) asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(v[j]) : "v"(b));
            if (OP == 38) asm volatile("v_pk_sub_i16 %0, %0, %1 clamp" : "+v"(v[j]) : "v"(b));
            if (OP == 39) asm volatile("v_alignbit_b32 %0, %0, %1, 16" : "+v"(v[j]) : "v"(b));
            if (OP == 40) asm volatile("v_pk_sub_u16 %0, %0, %1" : "+v"(v[j]) : "v"(b));
            if (OP == 41) {   // one packed cell pair of the int16 kernel: 12 ops on a dependent chain, 8 chains in flight
                int t, u, k1, k2;
                asm volatile("v_add_u16_sdwa %0, %5, %6 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:BYTE_1\n\t"
                             "v_add_u16_sdwa %0, %5, %6 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:BYTE_2\n\t"
                             "v_pk_max_u16 %1, %0, %7\n\t"
                             "v_pk_max_u16 %1, %1, %4\n\t"
                             "v_pk_sub_u16 %2, %0, %6\n\t"
                             "v_pk_max_u16 %4, %2, %4\n\t"
                             "v_pk_sub_u16 %4, %4, %6\n\t"
                             "v_mad_u32_u16 %3, %1, %6, %7\n\t"
                             "v_mad_u32_u16 %2, %1, %6, %7 op_sel:[1,0,0,0]\n\t"
                             "v_max3_i32 %4, %4, %3, %2\n\t"
                             "v_pk_max_u16 %0, %1, %0\n\t"
                             "v_pk_sub_u16 %0, %0, %6"
                             : "+v"(v[j]), "=&v"(t), "=&v"(u), "=&v"(k1), "+v"(v[(j + 1) & 7]) : "v"(v[(j + 2) & 7]), "v"(b), "v"(a));
            }
            if (OP == 33) { if (j & 1) asm volatile("s_and_b64 %0, %0, %1" : "+s"(m64) : "s"(m2)); else asm volatile("v_max_i32 %0, %0, %1" : "+v"(v[j]) : "v"(b)); }
        }
    }
    int s = 0;
    for (int j = 0; j < 8; j++) s += v[j] + (int)f[j];
    s += (int)m2;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
double run(const char* name, int waves_per_simd)
{
    int blocks = 256 * waves_per_simd;   // 256 CUs x (4 waves per block = 1 per SIMD) x waves_per_simd
    int* d; hipMalloc(&d, sizeof(int) * blocks * 256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 3, 5);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 3, 5);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double ops = (double)blocks * 256 * ITER * 8;
    double rate = ops / (ms * 1e-3);
    printf("%-14s waves/SIMD=%d  %.2f T lane-ops/s  (%.2f lanes/clk/SIMD at 2.4 GHz)\n", name, waves_per_simd, rate / 1e12,
           rate / (1024 * 2.4e9));
    hipFree(d);
    return rate;
}

int main()
{
    for (int w : {1, 2}) {
        run<1>("v_max_i32", w); run<4>("v_pk_max_i16", w); run<37>("v_pk_max_u16", w); run<40>("v_pk_sub_u16", w);
        run<38>("pk_sub clamp", w); run<34>("mad_u32_u16", w); run<35>("mad_i32_i16", w); run<36>("add_u16 sdwa", w);
        run<39>("v_alignbit", w); run<2>("v_max3_i32", w); run<41>("cell pair x12", w);
    }
    return 0;
}
