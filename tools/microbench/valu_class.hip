// Microbenchmark (round 4): which VALU instructions of gfx950 issue at 32 lanes per clock and SIMD ("2-cycle class":
// round 1 found v_add_u32 / v_sub_u32 / v_xor_b32 / v_mov_b32 / v_max_i16 there) and which at 16 (every packed, SDWA and
// VOP3 form the int16 kernel is made of), whether an SGPR / constant operand changes the class, and what a row of the
// packed-int16 block costs with the carry-free packed subtract as a 32-bit subtract, without the per-cell accumulator,
// and with the two SDWA score adds replaced by v_perm_b32 + v_add_u32.  Developer tool:
//   hipcc -O2 --offload-arch=gfx950 valu_class.hip -o /tmp/valu_class && /tmp/valu_class
#include <hip/hip_runtime.h>
#include <cstdio>

#define ITER 512
#define REP8(X) X X X X X X X X
// one independent instruction per register, 8 registers, 8 times per loop trip (64 VALU per trip: the loop's scalar
// instructions are 5 % of the stream)
#define ALL8(F) F("%[v0]") F("%[v1]") F("%[v2]") F("%[v3]") F("%[v4]") F("%[v5]") F("%[v6]") F("%[v7]")
// (ONE asm statement per loop trip: between two statements the compiler puts an s_nop)
#define OPK(NAME, F, CONSTR, ...)                                                                        \
    __global__ void __launch_bounds__(256) NAME(int* out, int a, int b)                                 \
    {                                                                                                   \
        int v[8];                                                                                       \
        for (int j = 0; j < 8; j++) v[j] = threadIdx.x + j * a;                                         \
        for (int it = 0; it < ITER; it++) {                                                             \
            asm volatile(REP8(ALL8(F)) : [v0] "+v"(v[0]), [v1] "+v"(v[1]), [v2] "+v"(v[2]), [v3] "+v"(v[3]), [v4] "+v"(v[4]), \
                         [v5] "+v"(v[5]), [v6] "+v"(v[6]), [v7] "+v"(v[7]) : [b] CONSTR(b) : __VA_ARGS__); \
        }                                                                                               \
        int s = 0;                                                                                      \
        for (int j = 0; j < 8; j++) s += v[j];                                                          \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                 \
    }
#define F_add_u32(R) "v_add_u32 " R ", " R ", %[b]" "\n\t"
OPK(k_add_u32, F_add_u32, "v", "vcc")
#define F_add_u32_s(R) "v_add_u32 " R ", %[b], " R "" "\n\t"
OPK(k_add_u32_s, F_add_u32_s, "s", "vcc")
#define F_add_u32_c(R) "v_add_u32 " R ", 5, " R "" "\n\t"
OPK(k_add_u32_c, F_add_u32_c, "v", "vcc")
#define F_add_u32_lit(R) "v_add_u32 " R ", 0x12340567, " R "" "\n\t"
OPK(k_add_u32_lit, F_add_u32_lit, "v", "vcc")
#define F_sub_u32(R) "v_sub_u32 " R ", " R ", %[b]" "\n\t"
OPK(k_sub_u32, F_sub_u32, "v", "vcc")
#define F_subrev_u32_s(R) "v_subrev_u32 " R ", %[b], " R "" "\n\t"
OPK(k_subrev_u32_s, F_subrev_u32_s, "s", "vcc")
#define F_subrev_u32_v(R) "v_subrev_u32 " R ", %[b], " R "" "\n\t"
OPK(k_subrev_u32_v, F_subrev_u32_v, "v", "vcc")
#define F_and_b32(R) "v_and_b32 " R ", " R ", %[b]" "\n\t"
OPK(k_and_b32, F_and_b32, "v", "vcc")
#define F_or_b32(R) "v_or_b32 " R ", " R ", %[b]" "\n\t"
OPK(k_or_b32, F_or_b32, "v", "vcc")
#define F_xor_b32(R) "v_xor_b32 " R ", " R ", %[b]" "\n\t"
OPK(k_xor_b32, F_xor_b32, "v", "vcc")
#define F_lshlrev_b32(R) "v_lshlrev_b32 " R ", 1, " R "" "\n\t"
OPK(k_lshlrev_b32, F_lshlrev_b32, "v", "vcc")
#define F_lshrrev_b32(R) "v_lshrrev_b32 " R ", 1, " R "" "\n\t"
OPK(k_lshrrev_b32, F_lshrrev_b32, "v", "vcc")
#define F_ashrrev_i32(R) "v_ashrrev_i32 " R ", 1, " R "" "\n\t"
OPK(k_ashrrev_i32, F_ashrrev_i32, "v", "vcc")
#define F_max_u32(R) "v_max_u32 " R ", " R ", %[b]" "\n\t"
OPK(k_max_u32, F_max_u32, "v", "vcc")
#define F_max_i32(R) "v_max_i32 " R ", " R ", %[b]" "\n\t"
OPK(k_max_i32, F_max_i32, "v", "vcc")
#define F_min_u32(R) "v_min_u32 " R ", " R ", %[b]" "\n\t"
OPK(k_min_u32, F_min_u32, "v", "vcc")
#define F_max_u16(R) "v_max_u16 " R ", " R ", %[b]" "\n\t"
OPK(k_max_u16, F_max_u16, "v", "vcc")
#define F_min_u16(R) "v_min_u16 " R ", " R ", %[b]" "\n\t"
OPK(k_min_u16, F_min_u16, "v", "vcc")
#define F_max_i16(R) "v_max_i16 " R ", " R ", %[b]" "\n\t"
OPK(k_max_i16, F_max_i16, "v", "vcc")
#define F_add_u16(R) "v_add_u16 " R ", " R ", %[b]" "\n\t"
OPK(k_add_u16, F_add_u16, "v", "vcc")
#define F_sub_u16(R) "v_sub_u16 " R ", " R ", %[b]" "\n\t"
OPK(k_sub_u16, F_sub_u16, "v", "vcc")
#define F_mul_lo_u16(R) "v_mul_lo_u16 " R ", " R ", %[b]" "\n\t"
OPK(k_mul_lo_u16, F_mul_lo_u16, "v", "vcc")
#define F_lshlrev_b16(R) "v_lshlrev_b16 " R ", 1, " R "" "\n\t"
OPK(k_lshlrev_b16, F_lshlrev_b16, "v", "vcc")
#define F_max_f32(R) "v_max_f32 " R ", " R ", %[b]" "\n\t"
OPK(k_max_f32, F_max_f32, "v", "vcc")
#define F_add_f32(R) "v_add_f32 " R ", " R ", %[b]" "\n\t"
OPK(k_add_f32, F_add_f32, "v", "vcc")
#define F_mul_f32(R) "v_mul_f32 " R ", " R ", %[b]" "\n\t"
OPK(k_mul_f32, F_mul_f32, "v", "vcc")
#define F_fmac_f32(R) "v_fmac_f32 " R ", %[b], %[b]" "\n\t"
OPK(k_fmac_f32, F_fmac_f32, "v", "vcc")
#define F_max_f16(R) "v_max_f16 " R ", " R ", %[b]" "\n\t"
OPK(k_max_f16, F_max_f16, "v", "vcc")
#define F_add_f16(R) "v_add_f16 " R ", " R ", %[b]" "\n\t"
OPK(k_add_f16, F_add_f16, "v", "vcc")
#define F_mov_b32(R) "v_mov_b32 " R ", %[b]" "\n\t"
OPK(k_mov_b32, F_mov_b32, "v", "vcc")
#define F_not_b32(R) "v_not_b32 " R ", " R "" "\n\t"
OPK(k_not_b32, F_not_b32, "v", "vcc")
#define F_cndmask(R) "v_cndmask_b32 " R ", " R ", %[b], vcc" "\n\t"
OPK(k_cndmask, F_cndmask, "v", "vcc")
#define F_mul_u32_u24(R) "v_mul_u32_u24 " R ", " R ", %[b]" "\n\t"
OPK(k_mul_u32_u24, F_mul_u32_u24, "v", "vcc")
#define F_add_co_u32(R) "v_add_co_u32 " R ", vcc, " R ", %[b]" "\n\t"
OPK(k_add_co_u32, F_add_co_u32, "v", "vcc")
#define F_max_u32_e64(R) "v_max_u32_e64 " R ", " R ", %[b]" "\n\t"
OPK(k_max_u32_e64, F_max_u32_e64, "v", "vcc")
#define F_add_u32_e64(R) "v_add_u32_e64 " R ", " R ", %[b]" "\n\t"
OPK(k_add_u32_e64, F_add_u32_e64, "v", "vcc")
#define F_max_u16_e64(R) "v_max_u16_e64 " R ", " R ", %[b]" "\n\t"
OPK(k_max_u16_e64, F_max_u16_e64, "v", "vcc")
#define F_pk_max_u16(R) "v_pk_max_u16 " R ", " R ", %[b]" "\n\t"
OPK(k_pk_max_u16, F_pk_max_u16, "v", "vcc")
#define F_pk_add_u16(R) "v_pk_add_u16 " R ", " R ", %[b]" "\n\t"
OPK(k_pk_add_u16, F_pk_add_u16, "v", "vcc")
#define F_pk_sub_u16(R) "v_pk_sub_u16 " R ", " R ", %[b]" "\n\t"
OPK(k_pk_sub_u16, F_pk_sub_u16, "v", "vcc")
#define F_pk_max_f16(R) "v_pk_max_f16 " R ", " R ", %[b]" "\n\t"
OPK(k_pk_max_f16, F_pk_max_f16, "v", "vcc")
#define F_pk_add_f16(R) "v_pk_add_f16 " R ", " R ", %[b]" "\n\t"
OPK(k_pk_add_f16, F_pk_add_f16, "v", "vcc")
#define F_perm_b32(R) "v_perm_b32 " R ", " R ", %[b], %[b]" "\n\t"
OPK(k_perm_b32, F_perm_b32, "v", "vcc")
#define F_alignbit(R) "v_alignbit_b32 " R ", " R ", %[b], 16" "\n\t"
OPK(k_alignbit, F_alignbit, "v", "vcc")
#define F_bfi(R) "v_bfi_b32 " R ", %[b], " R ", %[b]" "\n\t"
OPK(k_bfi, F_bfi, "v", "vcc")
#define F_add3(R) "v_add3_u32 " R ", " R ", %[b], %[b]" "\n\t"
OPK(k_add3, F_add3, "v", "vcc")
#define F_max3_u32(R) "v_max3_u32 " R ", " R ", %[b], %[b]" "\n\t"
OPK(k_max3_u32, F_max3_u32, "v", "vcc")
#define F_add_u16_sdwa(R) "v_add_u16_sdwa " R ", " R ", sext(%[b]) dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:BYTE_2\n\t"
OPK(k_add_u16_sdwa, F_add_u16_sdwa, "v", "vcc")
#define F_add_u32_sdwa(R) "v_add_u32_sdwa " R ", " R ", %[b] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\t"
OPK(k_add_u32_sdwa, F_add_u32_sdwa, "v", "vcc")
#define F_max_u16_sdwa_hi(R) "v_max_u16_sdwa " R ", " R ", %[b] dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1\n\t"
OPK(k_max_u16_sdwa_hi, F_max_u16_sdwa_hi, "v", "vcc")
#define F_mov_dpp(R) "v_mov_b32_dpp " R ", " R " row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
OPK(k_mov_dpp, F_mov_dpp, "v", "vcc")
#define F_add_u32_dpp(R) "v_add_u32_dpp " R ", " R ", %[b] row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
OPK(k_add_u32_dpp, F_add_u32_dpp, "v", "vcc")
#define F_max_u16_dpp(R) "v_max_u16_dpp " R ", " R ", %[b] row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
OPK(k_max_u16_dpp, F_max_u16_dpp, "v", "vcc")

// ---- a row of a block pair (8 cell pairs), as the kernel issues it: 16 SDWA adds, 8 x 5 packed cell instructions, 8 packed
// maxima into the accumulators; variants of its pieces ----
#define ROW_SDWA \
    "v_add_u16_sdwa %[t7], %[h6], sext(%[wly]) dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:BYTE_0\n\t" \
    "v_add_u16_sdwa %[h6], %[h5], sext(%[wlx]) dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:BYTE_0\n\t" \
    "v_add_u16_sdwa %[h5], %[h4], sext(%[wly]) dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:BYTE_1\n\t" \
    "v_add_u16_sdwa %[h4], %[h3], sext(%[wlx]) dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:BYTE_1\n\t" \
    "v_add_u16_sdwa %[h3], %[h2], sext(%[wly]) dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:BYTE_2\n\t" \
    "v_add_u16_sdwa %[h2], %[h1], sext(%[wlx]) dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:BYTE_2\n\t" \
    "v_add_u16_sdwa %[h1], %[h0], sext(%[wly]) dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:BYTE_3\n\t" \
    "v_add_u16_sdwa %[h0], %[d0], sext(%[wlx]) dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:BYTE_3\n\t" \
    "v_add_u16_sdwa %[t7], %[h6], sext(%[why]) dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:BYTE_0\n\t" \
    "v_add_u16_sdwa %[h6], %[h5], sext(%[whx]) dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:BYTE_0\n\t" \
    "v_add_u16_sdwa %[h5], %[h4], sext(%[why]) dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:BYTE_1\n\t" \
    "v_add_u16_sdwa %[h4], %[h3], sext(%[whx]) dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:BYTE_1\n\t" \
    "v_add_u16_sdwa %[h3], %[h2], sext(%[why]) dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:BYTE_2\n\t" \
    "v_add_u16_sdwa %[h2], %[h1], sext(%[whx]) dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:BYTE_2\n\t" \
    "v_add_u16_sdwa %[h1], %[h0], sext(%[why]) dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:BYTE_3\n\t" \
    "v_add_u16_sdwa %[h0], %[d0], sext(%[whx]) dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:BYTE_3\n\t"
// scores as one packed word per cell pair: v_perm_b32 picks byte k of the low slot's and of the high slot's profile word
// (zero-extended: scores + 2 ge >= 0), one 32-bit add puts both on the diagonal values (no carry: see DESIGN)
#define PERMADD(HD, HS, WH, WL, SEL) "v_perm_b32 %[x], " WH ", " WL ", " SEL "\n\tv_add_u32 " HD ", " HS ", %[x]\n\t"
#define ROW_PERM \
    PERMADD("%[t7]", "%[h6]", "%[why]", "%[wly]", "%[s0]") PERMADD("%[h6]", "%[h5]", "%[whx]", "%[wlx]", "%[s0]") \
    PERMADD("%[h5]", "%[h4]", "%[why]", "%[wly]", "%[s1]") PERMADD("%[h4]", "%[h3]", "%[whx]", "%[wlx]", "%[s1]") \
    PERMADD("%[h3]", "%[h2]", "%[why]", "%[wly]", "%[s2]") PERMADD("%[h2]", "%[h1]", "%[whx]", "%[wlx]", "%[s2]") \
    PERMADD("%[h1]", "%[h0]", "%[why]", "%[wly]", "%[s3]") PERMADD("%[h0]", "%[d0]", "%[whx]", "%[wlx]", "%[s3]")
// (the same with the perms of a row issued first, into eight registers: what a software-pipelined row would do)
#define ROW_ADD8 \
    "v_add_u32 %[t7], %[h6], %[wly]\n\tv_add_u32 %[h6], %[h5], %[wlx]\n\tv_add_u32 %[h5], %[h4], %[why]\n\tv_add_u32 %[h4], %[h3], %[whx]\n\t" \
    "v_add_u32 %[h3], %[h2], %[wly]\n\tv_add_u32 %[h2], %[h1], %[wlx]\n\tv_add_u32 %[h1], %[h0], %[why]\n\tv_add_u32 %[h0], %[d0], %[whx]\n\t"
#define CELL(T, F, SUB) \
    "v_pk_max_u16 %[x], " T ", " F "\n\t" SUB " %[u], " T ", %[g]\n\t" \
    "v_pk_max_u16 " T ", %[x], %[ev]\n\tv_pk_max_u16 " F ", %[u], " F "\n\tv_pk_max_u16 %[ev], %[u], %[ev]\n\t"
#define CELLR(T, F) /* v_subrev: SGPR in src0 */ \
    "v_pk_max_u16 %[x], " T ", " F "\n\tv_subrev_u32 %[u], %[g], " T "\n\t" \
    "v_pk_max_u16 " T ", %[x], %[ev]\n\tv_pk_max_u16 " F ", %[u], " F "\n\tv_pk_max_u16 %[ev], %[u], %[ev]\n\t"
#define CELLS8(SUB) CELL("%[h0]", "%[f0]", SUB) CELL("%[h1]", "%[f1]", SUB) CELL("%[h2]", "%[f2]", SUB) CELL("%[h3]", "%[f3]", SUB) \
                    CELL("%[h4]", "%[f4]", SUB) CELL("%[h5]", "%[f5]", SUB) CELL("%[h6]", "%[f6]", SUB) CELL("%[t7]", "%[f7]", SUB)
#define CELLS8R CELLR("%[h0]", "%[f0]") CELLR("%[h1]", "%[f1]") CELLR("%[h2]", "%[f2]") CELLR("%[h3]", "%[f3]") \
                CELLR("%[h4]", "%[f4]") CELLR("%[h5]", "%[f5]") CELLR("%[h6]", "%[f6]") CELLR("%[t7]", "%[f7]")
#define ACC8 \
    "v_pk_max_u16 %[a0], %[a0], %[h0]\n\tv_pk_max_u16 %[a1], %[a1], %[h1]\n\tv_pk_max_u16 %[a2], %[a2], %[h2]\n\tv_pk_max_u16 %[a3], %[a3], %[h3]\n\t" \
    "v_pk_max_u16 %[a4], %[a4], %[h4]\n\tv_pk_max_u16 %[a5], %[a5], %[h5]\n\tv_pk_max_u16 %[a6], %[a6], %[h6]\n\tv_pk_max_u16 %[a7], %[a7], %[t7]\n\t"

template <int V>
__global__ void __launch_bounds__(256) krow(unsigned* out, unsigned a, unsigned b)
{
    unsigned h0 = threadIdx.x + 5000, h1 = h0 + a, h2 = h1 + a, h3 = h2 + a, h4 = h3 + a, h5 = h4 + a, h6 = h5 + a, t7 = h6 + a, d0 = 7000;
    unsigned f0 = 6000, f1 = f0 + b, f2 = f1 + b, f3 = f2 + b, f4 = f3 + b, f5 = f4 + b, f6 = f5 + b, f7 = f6 + b, ev = 6100;
    unsigned a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0, x, u;
    unsigned wlx = 0x02060206u, wly = 0x06020602u, whx = 0x02020606u, why = 0x06060202u;
    unsigned gv = 0x00040004u;
    const unsigned s0 = 0x0c040c00u, s1 = 0x0c050c01u, s2 = 0x0c060c02u, s3 = 0x0c070c03u;
    asm volatile("" : "+v"(wlx), "+v"(wly), "+v"(whx), "+v"(why), "+v"(gv));
    for (int it = 0; it < ITER; it++) {
#define OPERANDS \
        : [h0] "+v"(h0), [h1] "+v"(h1), [h2] "+v"(h2), [h3] "+v"(h3), [h4] "+v"(h4), [h5] "+v"(h5), [h6] "+v"(h6), [t7] "+v"(t7), \
          [f0] "+v"(f0), [f1] "+v"(f1), [f2] "+v"(f2), [f3] "+v"(f3), [f4] "+v"(f4), [f5] "+v"(f5), [f6] "+v"(f6), [f7] "+v"(f7), \
          [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3), [a4] "+v"(a4), [a5] "+v"(a5), [a6] "+v"(a6), [a7] "+v"(a7), \
          [ev] "+v"(ev), [x] "=&v"(x), [u] "=&v"(u) \
        : [d0] "v"(d0), [wlx] "v"(wlx), [wly] "v"(wly), [whx] "v"(whx), [why] "v"(why), [g] "s"(0x00040004u), \
          [s0] "s"(s0), [s1] "s"(s1), [s2] "s"(s2), [s3] "s"(s3)
#define OPERANDS_GV \
        : [h0] "+v"(h0), [h1] "+v"(h1), [h2] "+v"(h2), [h3] "+v"(h3), [h4] "+v"(h4), [h5] "+v"(h5), [h6] "+v"(h6), [t7] "+v"(t7), \
          [f0] "+v"(f0), [f1] "+v"(f1), [f2] "+v"(f2), [f3] "+v"(f3), [f4] "+v"(f4), [f5] "+v"(f5), [f6] "+v"(f6), [f7] "+v"(f7), \
          [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3), [a4] "+v"(a4), [a5] "+v"(a5), [a6] "+v"(a6), [a7] "+v"(a7), \
          [ev] "+v"(ev), [x] "=&v"(x), [u] "=&v"(u) \
        : [d0] "v"(d0), [wlx] "v"(wlx), [wly] "v"(wly), [whx] "v"(whx), [why] "v"(why), [g] "v"(gv), \
          [s0] "s"(s0), [s1] "s"(s1), [s2] "s"(s2), [s3] "s"(s3)
#pragma unroll
        for (int r = 0; r < 8; r++) {
            if (V == 0) asm volatile(ROW_SDWA CELLS8("v_pk_sub_u16") ACC8 OPERANDS);           // the kernel's value-step row: 64
            if (V == 1) asm volatile(ROW_SDWA CELLS8R ACC8 OPERANDS);                          // 32-bit subtract, gap_open in an SGPR (v_subrev)
            if (V == 2) asm volatile(ROW_SDWA CELLS8("v_sub_u32") ACC8 OPERANDS_GV);           // 32-bit subtract, gap_open in a VGPR
            if (V == 3) asm volatile(ROW_SDWA CELLS8("v_sub_u32") OPERANDS_GV);                // ... and no accumulators: 56
            if (V == 4) asm volatile(ROW_PERM CELLS8("v_sub_u32") OPERANDS_GV);                // ... and perm + 32-bit add for the scores: 56
            if (V == 5) asm volatile(ROW_ADD8 CELLS8("v_sub_u32") OPERANDS_GV);                // ... scores as ready packed words (lower bound): 48
            if (V == 6) asm volatile(ROW_SDWA CELLS8("v_pk_sub_u16") OPERANDS);                // packed subtract, no accumulators: 56
            if (V == 7) asm volatile(ROW_ADD8 CELLS8("v_pk_sub_u16") OPERANDS);                // packed subtract, ready words: 48
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = h0 + h1 + h2 + h3 + h4 + h5 + h6 + t7 + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + ev + a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <typename K>
static float time_kernel(K kern, int blocks)
{
    int* d; hipMalloc(&d, sizeof(int) * blocks * 256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 3, 5);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 3, 5);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipFree(d);
    return ms;
}

#define RUN(NAME) do { printf("%-18s", #NAME + 2); for (int w : {1, 2, 4, 8}) { float ms = time_kernel(NAME, 256 * w); \
        double rate = (double)256 * w * 256 * ITER * 64 / (ms * 1e-3); printf("  w%d %5.2f", w, rate / (1024 * 2.4e9)); } printf("   lanes/clk/SIMD at 2.4 GHz\n"); } while (0)

template <int V>
static void run_row(const char* name, int ninstr)
{
    printf("%-64s", name);
    for (int w : {1, 2, 4}) {
        int blocks = 256 * w;
        unsigned* d; hipMalloc(&d, sizeof(unsigned) * blocks * 256);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(krow<V>, dim3(blocks), dim3(256), 0, 0, d, 3u, 5u);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(krow<V>, dim3(blocks), dim3(256), 0, 0, d, 3u, 5u);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipFree(d);
        const double cyc = ms * 1e-3 * 2.4e9 / ((double)ITER * 8 * w);     // SIMD cycles per row (8 cell pairs)
        printf("  w%d %6.1f cyc/row (%4.2f/instr)", w, cyc, cyc / ninstr);
    }
    printf("\n");
}

int main()
{
    printf("== issue rate per opcode, lanes per clock and SIMD, with 1 / 2 / 4 / 8 waves per SIMD ==\n");
    RUN(k_add_u32); RUN(k_add_u32_s); RUN(k_add_u32_c); RUN(k_add_u32_lit); RUN(k_sub_u32); RUN(k_subrev_u32_s); RUN(k_subrev_u32_v);
    RUN(k_and_b32); RUN(k_or_b32); RUN(k_xor_b32); RUN(k_lshlrev_b32); RUN(k_lshrrev_b32); RUN(k_ashrrev_i32);
    RUN(k_max_u32); RUN(k_max_i32); RUN(k_min_u32); RUN(k_max_u16); RUN(k_min_u16); RUN(k_max_i16); RUN(k_add_u16); RUN(k_sub_u16);
    RUN(k_mul_lo_u16); RUN(k_lshlrev_b16); RUN(k_max_f32); RUN(k_add_f32); RUN(k_mul_f32); RUN(k_fmac_f32); RUN(k_max_f16); RUN(k_add_f16);
    RUN(k_mov_b32); RUN(k_not_b32); RUN(k_cndmask); RUN(k_mul_u32_u24); RUN(k_add_co_u32);
    RUN(k_max_u32_e64); RUN(k_add_u32_e64); RUN(k_max_u16_e64);
    RUN(k_pk_max_u16); RUN(k_pk_add_u16); RUN(k_pk_sub_u16); RUN(k_pk_max_f16); RUN(k_pk_add_f16);
    RUN(k_perm_b32); RUN(k_alignbit); RUN(k_bfi); RUN(k_add3); RUN(k_max3_u32);
    RUN(k_add_u16_sdwa); RUN(k_add_u32_sdwa); RUN(k_max_u16_sdwa_hi); RUN(k_mov_dpp); RUN(k_add_u32_dpp); RUN(k_max_u16_dpp);
    printf("== one row of a block pair (8 cell pairs), SIMD cycles per row ==\n");
    run_row<0>("kernel's value-step row: 16 sdwa + 8 x 5 packed + 8 acc", 64);
    run_row<1>("  v_pk_sub_u16 -> v_subrev_u32 (gap_open in an SGPR)", 64);
    run_row<2>("  v_pk_sub_u16 -> v_sub_u32 (gap_open in a VGPR)", 64);
    run_row<3>("  v_sub_u32, no accumulators", 56);
    run_row<4>("  v_sub_u32, no accumulators, perm + add_u32 for the scores", 56);
    run_row<5>("  v_sub_u32, no accumulators, scores as ready words (bound)", 48);
    run_row<6>("  v_pk_sub_u16, no accumulators", 56);
    run_row<7>("  v_pk_sub_u16, no accumulators, ready words", 48);
    return 0;
}
