// align16_kernel.hip -- packed-int16 variant of the banded affine-gap extension kernel (gfx950, wave64).
//
// Same schedule as align_kernel.hip (one pair per G-lane group, band-stationary column state, one block-anti-diagonal
// per step, eager z-drop), but every lane works on TWO adjacent column blocks at once: slots 2p (low half) and 2p+1
// (high half) of a 32-bit register, v_pk_* arithmetic, i.e. two DP cells per VALU lane-op.  What makes that possible
// and bit-exact (each point is emulated and checked against the oracle in oracle/agatha_lanes_model.c,
// agatha_model_lanes16):
//   * values are an unsigned 16-bit REPRESENTATION in a DRIFTING FRAME: rep = value + ge * (row + column) - base + 32768,
//     i.e. every quantity of cell (row, column) -- its H, the E and F that enter it -- is seen from its own
//     anti-diagonal.  A gap extension then costs nothing (E' = max(t - gap_open, E), F likewise: five packed instructions
//     per cell pair instead of seven), the two anti-diagonals between a cell and its diagonal neighbour are a constant
//     + 2 ge inside the score profile, every boundary value of the first band width is a constant, and an anti-diagonal
//     maximum can hardly sink in this frame (it loses at most ge per anti-diagonal in value).  `base` is raised whenever
//     the representation of a maximum passes R_REBASE, so sequence length does not limit the domain; maxima are moved
//     back to values where they are compared across anti-diagonals (z-drop, running maximum).  Three disjoint zones:
//     in-band cells >= R_LO, the reference's -infinity and what derives from it in [R_GLO, R_LO), cells outside the band
//     below R_GLO.
//   * no per-cell band test (an EXEC mask cannot switch off half a register): every cell of an active block is computed,
//     and the band is cut by capping E (upper-edge blocks: leaving the band to the right) or F (lower-edge blocks:
//     leaving it downwards) at R_OUT on ONE cell diagonal -- one extra v_pk_min on those eight cells -- plus R_OUT on
//     what enters an out-of-band cell from a neighbouring block.  Out-of-band cells then only hold values below R_GLO, lose
//     every max against an in-band value, and an anti-diagonal whose maximum is below R_GLO is empty, as in the
//     reference.  Lower-edge blocks hand on the reference's stale row values (agatha_kernel.h:33 skips, it does not
//     reset).  Rows past the end of the query and inactive halves are kept out of the maxima by a zero multiplier in
//     the key computation (v_mad_u32_u16).
//   * a pair whose anti-diagonal maximum comes too close to the zone borders (an in-band cell could leave its zone, or
//     the exact value of -infinity could start to matter) is abandoned and flagged for the int32 kernel.
// The kernel is compiled per cut diagonal T0 = w - 8*ceil(w/8) (0..-7), so which cells carry a cut operand is known at
// compile time (for T0 < -1 three more block kinds per pair are cut, on T0 + 8).  The launcher offers it as a candidate
// when the scores pass agatha16_scores_ok() and a (G, P) exists for the window; record_kernel (align_kernel.hip) picks
// the candidate that runs from the batch's length histogram.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <limits.h>

#include "kernels.h"
#include "device_common.h"

namespace agatha {

namespace r16 {
constexpr int BIAS = 32768;
constexpr int LO = -13000 + BIAS;        // in-band values are >= LO (bail-out rule)
constexpr int NEG = -17500 + BIAS;       // the reference's -infinity
constexpr int GLO = -22000 + BIAS;       // below: out-of-band cells / nothing
constexpr int OUT = -28000 + BIAS;       // state entering an out-of-band cell; E / F where they leave the band
constexpr int REBASE = 2048 + BIAS;      // (+ lift) rebase when an anti-diagonal maximum exceeds this
constexpr int DELTA = 2048;
// In-band cells lie up to `spread` below the maximum of their anti-diagonal.  Up to FREE_SPREAD that fits between LO and a
// representation that starts at BIAS; for steeper scores the in-band zone is lifted (a pair starts with base = -lift) into
// the range above REBASE that is otherwise unused.  Beyond MAX_SPREAD a pair would be abandoned on its first
// anti-diagonal (its in-band cells would reach down to the reference's -infinity): the launcher does not offer the kernel.
constexpr int FREE_SPREAD = 7000;
constexpr int MAX_SPREAD = 16000;
}  // namespace r16

__device__ __forceinline__ uint32_t pk2(uint32_t lo, uint32_t hi) { return (lo & 0xffffu) | (hi << 16); }
__device__ __forceinline__ uint32_t dup2(uint32_t v) { return (v & 0xffffu) * 0x10001u; }
__device__ __forceinline__ uint32_t hmask(bool lo, bool hi) { return (lo ? 0xffffu : 0u) | (hi ? 0xffff0000u : 0u); }
__device__ __forceinline__ uint32_t bfi(uint32_t m, uint32_t a, uint32_t b) { return (a & m) | (b & ~m); }   // m ? a : b

__device__ __forceinline__ uint32_t pk_max(uint32_t a, uint32_t b) { uint32_t d; asm("v_pk_max_u16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ uint32_t pk_min(uint32_t a, uint32_t b) { uint32_t d; asm("v_pk_min_u16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b) { uint32_t d; asm("v_pk_sub_u16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
// ..._c: second operand is a wave-uniform constant (SGPR)
__device__ __forceinline__ uint32_t pk_min_c(uint32_t a, uint32_t b) { uint32_t d; asm("v_pk_min_u16 %0, %1, %2" : "=v"(d) : "v"(a), "s"(b)); return d; }
__device__ __forceinline__ uint32_t pk_sub_c(uint32_t a, uint32_t b) { uint32_t d; asm("v_pk_sub_u16 %0, %1, %2" : "=v"(d) : "v"(a), "s"(b)); return d; }
__device__ __forceinline__ uint32_t pk_add_c(uint32_t a, uint32_t b) { uint32_t d; asm("v_pk_add_u16 %0, %1, %2" : "=v"(d) : "v"(a), "s"(b)); return d; }
__device__ __forceinline__ uint32_t pk_sub_sat_c(uint32_t a, uint32_t b) { uint32_t d; asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(d) : "v"(a), "s"(b)); return d; }
__device__ __forceinline__ uint32_t pk_shl_c(uint32_t a, uint32_t sh2) { uint32_t d; asm("v_pk_lshlrev_b16 %0, %1, %2" : "=v"(d) : "s"(sh2), "v"(a)); return d; }
__device__ __forceinline__ uint32_t pk_sub_sat(uint32_t a, uint32_t b) { uint32_t d; asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ uint32_t pk_shl(uint32_t a, uint32_t sh2) { uint32_t d; asm("v_pk_lshlrev_b16 %0, %1, %2" : "=v"(d) : "v"(sh2), "v"(a)); return d; }
__device__ __forceinline__ int mad_lo(uint32_t h, uint32_t m, int c) { int d; asm("v_mad_u32_u16 %0, %1, %2, %3" : "=v"(d) : "v"(h), "v"(m), "v"(c)); return d; }
__device__ __forceinline__ int mad_hi(uint32_t h, uint32_t m, int c) { int d; asm("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[1,1,0,0]" : "=v"(d) : "v"(h), "v"(m), "v"(c)); return d; }
__device__ __forceinline__ int max3i(int a, int b, int c) { int d; asm("v_max3_i32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
__device__ __forceinline__ int min3i(int a, int b, int c) { int d; asm("v_min3_i32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }

// representation of an in-band boundary value (value - base), or -infinity when it is out of the in-band zone
__device__ __forceinline__ uint32_t rep16(int v_minus_base)
{
    const int r = v_minus_base + r16::BIAS;
    return (uint32_t)(r < r16::LO ? r16::NEG : r);
}

// Initial state of column block r as 16-bit representations: H(-1, c), F(0, c) of its 8 columns and the corner
// (agatha_kernel.h:133-148, 207-215), with R_OUT wherever the value would enter a cell of the column's first row block
// that lies above the band (tu0: cells with jl - il > tu0 are outside).
__device__ __forceinline__ void init_half(int r, int R, int w, int W, int gapoe, int ge, int base, uint32_t (&hv)[8], uint32_t (&fv)[8], uint32_t& cv)
{
    const int q0 = imax(0, r - W);
    const int tu0 = w + 8 * q0 - 8 * r;
#pragma unroll
    for (int m = 0; m < 8; m++) {
        const int c = 8 * r + m;
        const bool in = (c < R) && (c <= w);
        // H(-1, c) = -(gapoe + ge c) seen from anti-diagonal c - 1, F(0, c) = that - gapoe seen from anti-diagonal c
        uint32_t h = in ? rep16(-(gapoe + ge) - base) : (uint32_t)r16::NEG;
        uint32_t f = in ? rep16(-2 * gapoe - base) : (uint32_t)r16::NEG;
        if (m > tu0) f = r16::OUT;                       // cell (0, m) of the first block is outside the band
        if (m < 7 && m + 1 > tu0) h = r16::OUT;          // so is the cell this value is the diagonal of
        hv[m] = h; fv[m] = f;
    }
    uint32_t c0 = (r == 0) ? rep16(-2 * ge - base) : ((8 * r - 1) <= w ? rep16(-(gapoe + ge) - base) : (uint32_t)r16::NEG);
    if (0 > tu0) c0 = r16::OUT;
    cv = c0;
}

// Score profile of one column block, four rows of 8 signed bytes (even columns in .x, odd in .y, column 0/1 in the
// top byte): query-base classes 0..3 = A, C, T, G.  One v_perm_b32 per word: the class (code >> 1) & 7 of each reference
// base (A 0, C 1, T 2, G 3, N 7) selects a byte of an 8-entry table {match at the query's own class, -mismatch
// elsewhere, -1 for N}.  lut_hi = entries 4..7, lut[c] = entries 0..3 of query class c.
struct ProfileLut { uint32_t hi, lo[4]; };
__device__ __forceinline__ ProfileLut make_profile_lut(int a, int b, int ge)
{
    ProfileLut t;
    // every score carries the + 2 ge of the two anti-diagonals between a cell and its diagonal neighbour (see rep above)
    const uint32_t nb = (uint32_t)(2 * ge - b) & 0xFFu, ma = (uint32_t)(2 * ge + a) & 0xFFu, nn = (uint32_t)(2 * ge - 1) & 0xFFu;
    t.hi = (nn << 24) | (nb * 0x00010101u);
#pragma unroll
    for (int c = 0; c < 4; c++) t.lo[c] = ((nb * 0x01010101u) & ~(0xFFu << (8 * c))) | (ma << (8 * c));
    return t;
}
__device__ __forceinline__ void build_profile5(uint2* __restrict__ prof, uint32_t rword, const ProfileLut& t)
{
    const uint32_t se = (rword >> 5) & 0x07070707u, so = (rword >> 1) & 0x07070707u;     // columns 0,2,4,6 / 1,3,5,7
#pragma unroll
    for (int c = 0; c < 4; c++)
        prof[c * 64] = make_uint2(__builtin_amdgcn_perm(t.hi, t.lo[c], se), __builtin_amdgcn_perm(t.hi, t.lo[c], so));
}

// packed query word -> class index per nibble: A(1)->0 C(3)->1 T(4)->2 G(7)->3.  Pairs with N in the query are not
// given to this kernel (exotic_kernel); the N padding behind the end of a query only reaches rows that do not exist.
__device__ __forceinline__ uint32_t class_word(uint32_t qword) { return (qword >> 1) & 0x33333333u; }

// ---- the three hand-scheduled pieces of a block row.  They are single asm statements on purpose: the compiler cannot
// see into inline asm and pads every short asm statement that feeds another with s_nop (it has to assume a partial-
// register writer), which cost ~18% of the issue slots when each instruction was its own statement.  Inside a statement
// the order below keeps the one real hazard of this code away: an SDWA write of half a register must be followed by at
// least one other instruction before the register is read. ----

// h[j] <- h[j-1] + score(row, column j) for both halves (h[-1] = d0: the value left of / above-left of the block), i.e.
// the column state H of the row above is turned, in its own registers, into "diagonal + score" of this row.  Scores are
// signed bytes: column j in byte 3 - j/2 of the even (.x) or odd (.y) word of the profile row (wl: low half, wh: high).
// Order: all low halves from column 7 down to 0, then all high halves: every add reads its left neighbour before that
// neighbour is overwritten, and a half-register write is never read by the next instruction.
// Column 7's sum goes to t7 instead of h[7]: H(row, 7) is only ever read by the block to the right (row hand-off), never
// as the state of column 7 (the diagonal of column 8 belongs to the next block), so it is produced where it is handed on.
__device__ __forceinline__ void row_add_scores(uint32_t (&h)[8], uint32_t& t7, uint32_t d0, uint2 wl, uint2 wh)
{
    asm("v_add_u16_sdwa %7, %6, sext(%10) dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:BYTE_0\n\t"
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
        : "+v"(h[0]), "+v"(h[1]), "+v"(h[2]), "+v"(h[3]), "+v"(h[4]), "+v"(h[5]), "+v"(h[6]), "+v"(t7)
        : "v"(d0), "v"(wl.x), "v"(wl.y), "v"(wh.x), "v"(wh.y));
}

// four cells of a row: t comes in as diagonal + score and leaves as the new H; F of the four columns and the row's E are
// advanced.  In the drifting frame neither loses the gap-extension score (see rep above): five instructions per cell
// pair.  MF / ME: two bits per cell, 1 = F / E of that cell leaves the band on the block kind's first cut diagonal (operand
// ca: R_OUT in the halves of that kind, 0xFFFF elsewhere), 2 = on its second one (operand cb); the assembler drops the rest.
template <int MF, int ME>
__device__ __forceinline__ void row_cells4(uint32_t& t0, uint32_t& t1, uint32_t& t2, uint32_t& t3, uint32_t& f0, uint32_t& f1,
                                           uint32_t& f2, uint32_t& f3, uint32_t& ev, uint32_t gapo, uint32_t fa, uint32_t fb,
                                           uint32_t ea, uint32_t eb)
{
    uint32_t x, u;
#define AGATHA16_CELL(T, F, J) \
        "v_pk_max_u16 %[x], " T ", " F "\n\t" \
        "v_pk_sub_u16 %[u], " T ", %[gapo]\n\t" \
        "v_pk_max_u16 " T ", %[x], %[ev]\n\t" \
        "v_pk_max_u16 " F ", %[u], " F "\n\t" \
        "v_pk_max_u16 %[ev], %[u], %[ev]\n\t" \
        ".if ((%[mf] >> (2 * " J ")) & 3) == 1\n\tv_pk_min_u16 " F ", " F ", %[fa]\n\t.endif\n\t" \
        ".if ((%[mf] >> (2 * " J ")) & 3) == 2\n\tv_pk_min_u16 " F ", " F ", %[fb]\n\t.endif\n\t" \
        ".if ((%[me] >> (2 * " J ")) & 3) == 1\n\tv_pk_min_u16 %[ev], %[ev], %[ea]\n\t.endif\n\t" \
        ".if ((%[me] >> (2 * " J ")) & 3) == 2\n\tv_pk_min_u16 %[ev], %[ev], %[eb]\n\t.endif\n\t"
    asm(AGATHA16_CELL("%[t0]", "%[f0]", "0")
        AGATHA16_CELL("%[t1]", "%[f1]", "1")
        AGATHA16_CELL("%[t2]", "%[f2]", "2")
        AGATHA16_CELL("%[t3]", "%[f3]", "3")
        : [t0] "+v"(t0), [t1] "+v"(t1), [t2] "+v"(t2), [t3] "+v"(t3), [f0] "+v"(f0), [f1] "+v"(f1), [f2] "+v"(f2), [f3] "+v"(f3),
          [ev] "+v"(ev), [x] "=&v"(x), [u] "=&v"(u)
        : [gapo] "s"(gapo), [fa] "v"(fa), [fb] "v"(fb), [ea] "v"(ea), [eb] "v"(eb), [mf] "i"(MF), [me] "i"(ME));
#undef AGATHA16_CELL
}

// packed maxima of four cells.  A[d] collects, for cell anti-diagonal d = il + jl of the block pair, the maximum of
// H * kmul + (relative column of the block - il): the "+ d" that turns it into H * 2^K + relative column of the cell
// is the same for every candidate of an accumulator and is added when the accumulators are read.  kmul = 2^K, or 0
// where the row does not exist.
__device__ __forceinline__ void row_keys4(int& a0, int& a1, int& a2, int& a3, uint32_t h0, uint32_t h1, uint32_t h2, uint32_t h3,
                                          uint32_t kmul, int rowc_lo, int rowc_hi)
{
    int x, y, x2, y2;
    // (the two multiply-adds of the next cell sit between a cell's multiply-adds and its v_max3: a lone wave does not issue
    // a dependent instruction back to back without a bubble)
    asm("v_mad_u32_u16 %[x], %[h0], %[km], %[rl]\n\t"
        "v_mad_u32_u16 %[y], %[h0], %[km], %[rh] op_sel:[1,1,0,0]\n\t"
        "v_mad_u32_u16 %[x2], %[h1], %[km], %[rl]\n\t"
        "v_mad_u32_u16 %[y2], %[h1], %[km], %[rh] op_sel:[1,1,0,0]\n\t"
        "v_max3_i32 %[a0], %[a0], %[x], %[y]\n\t"
        "v_mad_u32_u16 %[x], %[h2], %[km], %[rl]\n\t"
        "v_mad_u32_u16 %[y], %[h2], %[km], %[rh] op_sel:[1,1,0,0]\n\t"
        "v_max3_i32 %[a1], %[a1], %[x2], %[y2]\n\t"
        "v_mad_u32_u16 %[x2], %[h3], %[km], %[rl]\n\t"
        "v_mad_u32_u16 %[y2], %[h3], %[km], %[rh] op_sel:[1,1,0,0]\n\t"
        "v_max3_i32 %[a2], %[a2], %[x], %[y]\n\t"
        "v_max3_i32 %[a3], %[a3], %[x2], %[y2]"
        : [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3), [x] "=&v"(x), [y] "=&v"(y), [x2] "=&v"(x2), [y2] "=&v"(y2)
        : [h0] "v"(h0), [h1] "v"(h1), [h2] "v"(h2), [h3] "v"(h3), [km] "v"(kmul), [rl] "v"(rowc_lo), [rh] "v"(rowc_hi));
}

// profile row of query class (bits shift+1..shift of the class word): v_bfe_u32 + v_lshl_add_u32 (laundered so that the
// compiler does not turn the pair into shift + and + add)
__device__ __forceinline__ uint2 profile_row(const uint2* __restrict__ p, uint32_t qc, int shift)
{
    uint32_t c = __builtin_amdgcn_ubfe(qc, shift, 2);
    asm("" : "+v"(c));
    return p[c * 64u];
}

// which cut a cell's E (jl - il) / F (il - jl) carries in a block pair compiled for cut diagonal T0: 1 on T0, 2 on T0 + 8
template <int T0>
constexpr int cut_code(int d)
{
    return d == T0 ? 1 : ((T0 + 8 < 7) && d == T0 + 8) ? 2 : 0;
}
template <int T0>
constexpr int cut_mask4(int il, int j0, bool for_e)
{
    int m = 0;
    for (int j = 0; j < 4; j++) m |= cut_code<T0>(for_e ? (j0 + j) - il : il - (j0 + j)) << (2 * j);
    return m;
}

struct BlockOps {
    uint32_t gapo2, cu, cu2, cl, cl2, NRK, qc_lo, qc_hi;
    int crel_lo, crel_hi;
    const uint2* pl; const uint2* ph;
};

// one row of a block pair.  wl / wh: the row's profile words (requested a row ahead)
template <int K, int T0, int IL>
__device__ __forceinline__ void block_row16(uint32_t (&h)[8], uint32_t (&f)[8], uint32_t d0, uint32_t (&e)[8], uint32_t (&oh)[8],
                                            int (&A)[15], const BlockOps& o, uint2& wl, uint2& wh)
{
    // key multiplier of this row: 2^K where the row exists (IL < rows), 0 where it does not
    const uint32_t kmul = pk_min_c(pk_sub_sat_c(o.NRK, dup2((uint32_t)(IL << K))), dup2(1u << K));
    uint32_t& t7 = oh[IL];
    row_add_scores(h, t7, d0, wl, wh);
    if (IL < 7) {
        wl = profile_row(o.pl, o.qc_lo, 24 - 4 * IL);
        wh = profile_row(o.ph, o.qc_hi, 24 - 4 * IL);
    }
    uint32_t ev = e[IL];
    // E is cut on cell diagonal jl - il == T0 (upper edge blocks) or T0 + 8 (the block next to the corner of the band),
    // F on il - jl == T0 or T0 + 8 (lower edge blocks)
    row_cells4<cut_mask4<T0>(IL, 0, false), cut_mask4<T0>(IL, 0, true)>(h[0], h[1], h[2], h[3], f[0], f[1], f[2], f[3], ev, o.gapo2, o.cl, o.cl2, o.cu, o.cu2);
    row_cells4<cut_mask4<T0>(IL, 4, false), cut_mask4<T0>(IL, 4, true)>(h[4], h[5], h[6], t7, f[4], f[5], f[6], f[7], ev, o.gapo2, o.cl, o.cl2, o.cu, o.cu2);
    row_keys4(A[IL], A[IL + 1], A[IL + 2], A[IL + 3], h[0], h[1], h[2], h[3], kmul, o.crel_lo - IL, o.crel_hi - IL);
    row_keys4(A[IL + 4], A[IL + 5], A[IL + 6], A[IL + 7], h[4], h[5], h[6], t7, kmul, o.crel_lo - IL, o.crel_hi - IL);
    e[IL] = ev;
}

template <int K, int T0>
__device__ __forceinline__ void block_pair16(uint32_t (&h)[8], uint32_t (&f)[8], uint32_t corner, const uint32_t (&rh)[8],
                                             uint32_t (&e)[8], uint32_t (&oh)[8], int (&A)[15], const BlockOps& o)
{
    // profile rows are requested while the previous row's cells are being computed
    uint2 wl = profile_row(o.pl, o.qc_lo, 28), wh = profile_row(o.ph, o.qc_hi, 28);
    block_row16<K, T0, 0>(h, f, corner, e, oh, A, o, wl, wh);
    block_row16<K, T0, 1>(h, f, rh[0], e, oh, A, o, wl, wh);
    block_row16<K, T0, 2>(h, f, rh[1], e, oh, A, o, wl, wh);
    block_row16<K, T0, 3>(h, f, rh[2], e, oh, A, o, wl, wh);
    block_row16<K, T0, 4>(h, f, rh[3], e, oh, A, o, wl, wh);
    block_row16<K, T0, 5>(h, f, rh[4], e, oh, A, o, wl, wh);
    block_row16<K, T0, 6>(h, f, rh[5], e, oh, A, o, wl, wh);
    block_row16<K, T0, 7>(h, f, rh[6], e, oh, A, o, wl, wh);
}

// ---- packed per-half control arithmetic: column indices, row-block indices and tags are 16-bit (sequences shorter
// than 262 136 bases; longer pairs go to the int32 kernel), masks are 0xFFFF / 0 per half ----
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) { uint32_t d; asm("v_pk_add_u16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ uint32_t pk_sar15(uint32_t a, uint32_t f15) { uint32_t d; asm("v_pk_ashrrev_i16 %0, %1, %2" : "=v"(d) : "s"(f15), "v"(a)); return d; }
// a < b as signed 16-bit numbers whose difference fits 16 bits
__device__ __forceinline__ uint32_t lt_mask(uint32_t a, uint32_t b, uint32_t f15) { return pk_sar15(pk_sub(a, b), f15); }
// a == b
__device__ __forceinline__ uint32_t eq_mask(uint32_t a, uint32_t b, uint32_t one2) { return pk_sub_c(pk_min_c(a ^ b, one2), one2); }

// maxima over the G lanes of a group, result in every lane: two signed values and one unsigned at once (DPP row rotations
// inside a 16-lane row, ds_bpermute across rows)
template <int G>
__device__ __forceinline__ void group_max3(int& a, int& b, uint32_t& c, int lane)
{
#define AGATHA16_DPP_STAGE(CTRL) \
    { const int ta = __builtin_amdgcn_update_dpp(INT_MIN, a, CTRL, 0xf, 0xf, true), tb = __builtin_amdgcn_update_dpp(INT_MIN, b, CTRL, 0xf, 0xf, true); \
      const uint32_t tc = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)c, CTRL, 0xf, 0xf, true); \
      a = imax(a, ta); b = imax(b, tb); c = c > tc ? c : tc; }
    AGATHA16_DPP_STAGE(0x121) AGATHA16_DPP_STAGE(0x122) AGATHA16_DPP_STAGE(0x124) AGATHA16_DPP_STAGE(0x128)
#undef AGATHA16_DPP_STAGE
    if (G >= 32) {
        const int ta = lane_read(a, lane ^ 16), tb = lane_read(b, lane ^ 16); const uint32_t tc = (uint32_t)lane_read((int)c, lane ^ 16);
        a = imax(a, ta); b = imax(b, tb); c = c > tc ? c : tc;
    }
    if (G >= 64) {
        const int ta = lane_read(a, lane ^ 32), tb = lane_read(b, lane ^ 32); const uint32_t tc = (uint32_t)lane_read((int)c, lane ^ 32);
        a = imax(a, ta); b = imax(b, tb); c = c > tc ? c : tc;
    }
}

template <int G, int P, int T0>
__global__ void __launch_bounds__(256, 2)
align16_kernel(const AlignLaunch* __restrict__ La, AlignParams Pm, int kid)
{
    if (*La->choice != kid) return;                // another candidate takes the plain pairs of this batch (record_kernel)
    static_assert(T0 <= 0 && T0 >= -7, "T0 = w - 8 * ceil(w / 8)");
    constexpr bool CORNER_BLOCKS = (T0 + 8 < 7);   // blocks (0, W-1) and (pql-1, pql-W) are cut as well
    constexpr int S = 2 * P, GS = G * S;
    constexpr int K = KeyBits<GS>::value;
    constexpr int KMASK = (1 << K) - 1;

    __shared__ uint2 s_prof[4 * S * 4 * 64];      // [wave][slot][class][lane] score profiles
    // E(row, column right of the block) of every slot, one 16-bit value per row: what slot s wrote in step i is the E
    // input of slot s + 1 in step i + 1.  Region S double-buffers slot S-1 (read by the NEXT lane's slot 0 one step
    // later, after this lane has already written the new values): steps with odd i write region S, even i region S-1.
    // Row 8 of a region holds the column block the eight values belong to (0xFFFE = none).
    __shared__ uint16_t s_xe[4 * (S + 1) * 9 * 64];
    const int lane = threadIdx.x & 63;
    uint2* const prof0 = s_prof + (threadIdx.x >> 6) * (S * 4 * 64) + lane;
    uint16_t* const xe_wave = s_xe + (threadIdx.x >> 6) * ((S + 1) * 9 * 64);
    const int k = lane & (G - 1);
    const int gbase = lane & ~(G - 1);
    const int left_lane = gbase | ((k + G - 1) & (G - 1));

    const int gapo = Pm.gap_open, ge = Pm.gap_extend, gapoe = gapo + ge;
    const int sw = Pm.slice_width, z = Pm.z_threshold, w = Pm.band_width;
    const int W = (w + 7) >> 3;
    constexpr int t0 = T0;                         // = w - 8 * W (launcher): the cut diagonal of edge blocks
    int spread;                                    // how far below an anti-diagonal maximum an in-band cell can be
    {
        int per = 2 * ge; if (Pm.mismatch > per) per = Pm.mismatch; if (per < 1) per = 1;
        spread = gapoe + per * (w + 16) + 64;
    }
    // (+ 7 ge: the maxima of a step's eight anti-diagonals are compared after moving them to the frame of the first)
    const int bail_rep = r16::LO + spread + r16::DELTA + 7 * ge;
    const int lift = spread > r16::FREE_SPREAD ? spread - r16::FREE_SPREAD : 0;
    const int rebase_at = r16::REBASE + lift;
    const int frame_step = ge << K;                // + (7 - x) * frame_step: H field of anti-diagonal x seen from the step's last one
    const uint32_t GAPO2 = dup2((uint32_t)gapo);
    const uint32_t NEG2 = dup2(r16::NEG), OUT2 = dup2(r16::OUT), ONE2 = 0x00010001u, K2 = dup2(K), F15 = 0x000F000Fu;
    const uint32_t W2 = dup2((uint32_t)W), NOTAG = 0xFFFEFFFEu;
    const ProfileLut plut = make_profile_lut(Pm.match, Pm.mismatch, ge);
    // initial state of a column block beyond the first band width (init_half with r > W): constants per column
    const uint32_t HINIT = OUT2, H7INIT = NEG2;     // h[m], m < 7: the cell it is the diagonal of, (0, m+1), is outside for t0 <= 0
    const uint32_t F0INIT = (0 > t0) ? OUT2 : NEG2, FINIT = OUT2, CINIT = (0 > t0) ? OUT2 : NEG2;

    // ---- per-pair state (uniform inside a group) ----
    int Q = 0, R = 0, pair = 0, base = 0;
    typedef const __attribute__((address_space(1))) uint32_t* gptr_t;
    uint32_t pq = 0, pt = 0;                        // word offsets of the pair's sequences in the packed batches
    int i = 0, y = 0, ss = 0, se = 0;
    bool alive = false, exhausted = false, final_step = false;
    int best = 0, best_t = 0, best_q = 0;

    // ---- per-lane state, one register per slot PAIR (low half = slot 2p, high half = slot 2p + 1) ----
    uint32_t RC[P];                                 // column block of each slot
    uint32_t H[P][8], F[P][8], CORNER[P];
    uint32_t XH[P][8];                              // row hand-off of H (registers); E goes through s_xe
    uint32_t qcls[S];                               // packed query word of the row block each slot works on this step
    int A[15];

#pragma unroll
    for (int s = 0; s < S; s++) qcls[s] = 0;
#pragma unroll
    for (int p = 0; p < P; p++) { CORNER[p] = 0; RC[p] = 0;
#pragma unroll
        for (int m = 0; m < 8; m++) { H[p][m] = 0; F[p][m] = 0; XH[p][m] = 0; } }
#pragma unroll
    for (int x = 0; x < 15; x++) A[x] = 0;

    // First round of a full grid (two workgroups per CU, all resident): the workgroups b and b + gridDim/2 share a CU, wave
    // by wave a SIMD, so WHICH 16 queue positions a workgroup starts with decides which waves share a SIMD.  The
    // workgroups that will take a second pair (the C chunks with the shortest first-round pairs; BASELINE: 10 000 pairs on
    // 8192 lane groups) are the critical path, and they run alone -- 1.4x faster per step -- as soon as the wave they share
    // the SIMD with has finished: they get the shortest single-round chunks as partners, and the remaining chunks are
    // paired longest with shortest.  Later pairs come from the atomic queue, which then starts behind the first round.
    // (Up to two rounds; beyond that every pair comes from the queue.)
    constexpr int GPB = 4 * (64 / G);                  // lane groups per workgroup
    const int nblk = (int)gridDim.x, cap = nblk * GPB;
    int chain_chunks = (La->n - cap + GPB - 1) / GPB;
    const bool deal_full = (nblk == 2 * La->num_cus) && La->n > cap - GPB && chain_chunks <= nblk && !La->no_deal;
    // Less than one round on more workgroups than CUs: the CUs that hold one workgroup take the longest chunks (a wave
    // alone on its SIMD is the faster one), the others the shortest, again paired longest with shortest.
    const bool deal_part = nblk > La->num_cus && nblk < 2 * La->num_cus && La->n <= cap && !La->no_deal;
    const bool deal = deal_full || deal_part;
    bool first_round = deal;
    int first_idx = 0;
    if (deal_part) {
        const int Hh = La->num_cus, D = nblk - Hh, bb = (int)blockIdx.x;
        int chunk;
        if (bb >= Hh) chunk = nblk - 1 - (bb - Hh);            // second workgroup of its CU: the shortest chunks
        else if (bb < D) chunk = (Hh - D) + bb;                  // its partner: the longest of the 2 D shortest
        else chunk = bb - D;                                     // alone on its CU: the longest chunks
        first_idx = chunk * GPB + (int)(threadIdx.x >> 6) * (64 / G) + lane / G;
    }
    if (deal_full) {
        const int Hh = nblk / 2, C = chain_chunks, bb = (int)blockIdx.x;
        const int a = bb < Hh ? bb : bb - Hh;
        int chunk;
        if (C <= Hh) {
            if (bb < Hh) chunk = a < C ? nblk - C - 1 - a : a - C;
            else chunk = a < C ? nblk - 1 - a : nblk - 2 * C - 1 - (a - C);
        } else {
            // 1.5 to 2 rounds: every CU holds a chunk that takes second pairs.  The nblk - C single-round chunks are the
            // partners of the chunks with the longest two-round totals (shortest partner for the longest of them); the
            // chunks with the shortest first-round pairs share their CUs among themselves.
            const int Nn = nblk - C;
            if (a < Nn) chunk = bb < Hh ? Nn - 1 - a : Nn + a;
            else chunk = bb < Hh ? nblk - 1 - 2 * (a - Nn) : nblk - 2 - 2 * (a - Nn);
        }
        first_idx = chunk * GPB + (int)(threadIdx.x >> 6) * (64 / G) + lane / G;
    }

    for (;;) {
        // ------------------------------------------------------------------ work queue
        const bool need = !alive && !exhausted;
        if (__builtin_expect(__any(need), 0)) {
            int idx = 0;
            if (first_round) idx = first_idx;
            else {
                if (need && k == 0) idx = (int)atomicAdd(La->queue + 0, 1u) + (deal ? cap : 0);      // (deal_part: nothing is left)
                idx = lane_read(idx, gbase);
            }
            first_round = false;
            if (need) {
                if (idx >= La->n) exhausted = true;
                else {
                    pair = (int)La->order[idx];
                    if (La->exotic[pair] == 0) {           // kind 0: plain letters, not yet handed to the int32 kernel
                        Q = (int)La->qlens[pair]; R = (int)La->tlens[pair];
                        pq = La->qoffs[pair] >> 3;
                        pt = La->toffs[pair] >> 3;
                        const int pql = (Q + 7) >> 3, prl = (R + 7) >> 3;
                        best = 0; best_t = 0; best_q = 0; base = -lift;
                        // The pair starts with a dry step i = -1: no block is active in it, and the code that puts the
                        // initial column state back into slots that have not started yet (below, after each block pair)
                        // thereby initialises every slot.  Assigning that state here instead would make the register
                        // allocator copy ~100 registers on every trip round the main loop.
                        i = -1; y = -1; final_step = false;
                        ss = 0;
                        se = imin(imin(prl - 1, sw - 1), (((sw - 1) * 8 + 7 + w) / 2) / 8);
                        int kk = k;                 // laundered: nothing below is worth hoisting out of the main loop
                        asm volatile("" : "+v"(kk));
#pragma unroll
                        for (int p = 0; p < P; p++) {
                            const int ra = kk * S + 2 * p, rb = ra + 1;
                            RC[p] = pk2((uint32_t)ra, (uint32_t)rb);
#pragma unroll
                            for (int hf = 0; hf < 2; hf++) {
                                const int r = ra + hf, s = 2 * p + hf;
                                const uint32_t rw0 = (r < prl) ? ((gptr_t)La->packed_t)[pt + (uint32_t)r] : 0xEEEEEEEEu;
                                build_profile5(prof0 + s * (4 * 64), rw0, plut);
                            }
                        }
#pragma unroll
                        for (int x = 0; x < 15; x++) A[x] = 0;
                        alive = true;
                        if (Q <= 0 || R <= 0) {
                            if (k == 0) { La->score[pair] = 0; La->qend[pair] = 0; La->tend[pair] = 0; }
                            alive = false;
                        } else if (imin(W + 1, imin(pql, prl)) > GS) {
                            if (k == 0) { La->score[pair] = INT_MIN; La->qend[pair] = -1; La->tend[pair] = -1; }
                            alive = false;
                        } else if (pql + GS >= 32760 || prl + GS >= 32760) {      // indices are 16 bits wide here
                            if (k == 0) { La->exotic[pair] = 2; atomicAdd(La->kind_counts + 1, 1u); }
                            alive = false;
                        }
                    }
                }
            }
        }
        if (!__any(alive)) {
            if (__all(exhausted)) break;           // else every group drew a pair of another kind: draw again
            continue;
        }
        // The steps run in an inner loop of their own, left only when a group wants a new pair: with the queue code
        // inside the same loop the register allocator copied ~200 registers per step between two sets.
        do {

        // ------------------------------------------------------------------ one step
        const int pql = (Q + 7) >> 3, prl = (R + 7) >> 3;
        const int cb = 8 * imax(0, imax(i - pql + 1, (i - W + 1) >> 1) - 1);
        {
            const int cb_prev = 8 * imax(0, imax(i - pql, (i - W) >> 1) - 1);       // the same formula for step i - 1
            const int delta = cb - cb_prev;       // 0 or 8
#pragma unroll
            for (int x = 0; x < 7; x++) A[x] -= delta;
        }
        const int total = prl + pql - 1, lim = Q + R - 1;
        const uint32_t I2 = dup2((uint32_t)i), PQL1 = dup2((uint32_t)(pql - 1));
        const uint32_t SS2 = dup2((uint32_t)ss), SE2 = dup2((uint32_t)imin(se, prl - 1));
        const uint32_t RUNm = (alive && !final_step) ? 0xFFFFFFFFu : 0u;
        const uint32_t NRLAST = dup2((uint32_t)(Q - 8 * (pql - 1)));
        bool bail = false;

#pragma unroll
        for (int p = P - 1; p >= 0; p--) {
            __builtin_amdgcn_sched_barrier(0);      // keep the three block pairs apart: interleaving them costs registers
            const uint32_t rc = RC[p];
            const uint32_t q2 = pk_sub(I2, rc);                                    // row block of each half (signed)
            const uint32_t cs2 = pk_sub_sat(rc, W2), ce2 = pk_min(pk_add(rc, W2), PQL1);
            // active: cs <= q <= ce, ss <= r <= min(se, prl - 1), pair running
            const uint32_t ACTm = ~(lt_mask(q2, cs2, F15) | lt_mask(ce2, q2, F15) | lt_mask(rc, SS2, F15) | lt_mask(SE2, rc, F15)) & RUNm;
            {
                const int ra = (int)(rc & 0xffffu), rb = (int)(rc >> 16);
                // edge blocks: q == r - W (upper, E cut on cell diagonal T0), q == r + W (lower, F cut on -T0); for T0 < -1
                // also the two blocks next to the corners of the band: (0, W-1) (E cut on T0 + 8) and (pql-1, pql-W)
                // (F cut on -(T0 + 8)).  Every other block is uncut.
                const uint32_t UPm = eq_mask(pk_add(q2, W2), rc, ONE2) & ACTm, LOm = eq_mask(q2, pk_add(rc, W2), ONE2) & ACTm;
                uint32_t UP2m = 0u, LO2m = 0u;
                if (CORNER_BLOCKS) {
                    // E cut on T0 + 8: block (0, W-1), and the block of the LAST row block with r - q = W - 1 (a boundary block
                    // because its column ends there, although the cut is on its upper side)
                    const uint32_t lastq = eq_mask(q2, PQL1, ONE2);
                    UP2m = ((eq_mask(q2, 0u, ONE2) & eq_mask(rc, dup2((uint32_t)(W - 1)), ONE2)) |
                            (lastq & eq_mask(pk_add(q2, dup2((uint32_t)(W - 1))), rc, ONE2))) & ACTm;
                    LO2m = lastq & eq_mask(pk_add(q2, ONE2), pk_add(rc, W2), ONE2) & ACTm;
                }
                if (CORNER_BLOCKS) {
                    // safety net (only where the classification above has more than two kinds): a boundary block with a cut
                    // that is none of them must not be computed here
                    const uint32_t bnd = (eq_mask(q2, cs2, ONE2) | eq_mask(q2, ce2, ONE2)) & ACTm & ~(UPm | LOm | UP2m | LO2m);
                    if (__builtin_expect(bnd != 0u, 0)) {
                        const int qa_ = i - ra, qb_ = i - rb;
                        const bool bada = (bnd & 1u) && (w + 8 * qa_ - 8 * ra < 7 || w - 8 * qa_ + 8 * ra < 7);
                        const bool badb = (bnd >> 31) && (w + 8 * qb_ - 8 * rb < 7 || w - 8 * qb_ + 8 * rb < 7);
                        if (bada || badb) bail = true;
                    }
                }
                // slots that leave their column block after this step (q + 1 > ce) move on to column r + G*S: the reference
                // word of the new column is requested now and used after the block
                const uint32_t ADm = alive ? ~lt_mask(q2, ce2, F15) : 0u;
                uint32_t rwa = 0xEEEEEEEEu, rwb = 0xEEEEEEEEu;
                if ((ADm & 1u) && ra + GS < prl) rwa = ((gptr_t)La->packed_t)[pt + (uint32_t)(ra + GS)];
                if ((ADm >> 31) && rb + GS < prl) rwb = ((gptr_t)La->packed_t)[pt + (uint32_t)(rb + GS)];

                if (__builtin_expect(y == 0 && ACTm != 0u && (ra == prl - 1 || rb == prl - 1), 0)) {
                    // pass start: padded ref columns fall back to -infinity (agatha_kernel.h:207-215); where the value
                    // would enter an out-of-band cell of an upper edge block it is R_OUT as in init_half
                    const bool acta = (ACTm & 1u) != 0u, actb = (ACTm >> 31) != 0u;
                    const int tua = (UPm & 1u) ? t0 : (UP2m & 1u) ? t0 + 8 : 1000, tub = (UPm >> 31) ? t0 : (UP2m >> 31) ? t0 + 8 : 1000;
#pragma unroll
                    for (int m = 0; m < 8; m++) {
                        const bool pa = acta && ra == prl - 1 && 8 * ra + m >= R, pb = actb && rb == prl - 1 && 8 * rb + m >= R;
                        const uint32_t fva = (m > tua) ? r16::OUT : r16::NEG, fvb = (m > tub) ? r16::OUT : r16::NEG;
                        const uint32_t hva = (m < 7 && m + 1 > tua) ? r16::OUT : r16::NEG, hvb = (m < 7 && m + 1 > tub) ? r16::OUT : r16::NEG;
                        const uint32_t pm = hmask(pa, pb);
                        F[p][m] = bfi(pm, pk2(fva, fvb), F[p][m]);
                        H[p][m] = bfi(pm, pk2(hva, hvb), H[p][m]);
                    }
                }

                // ---- row inputs: the left neighbour's output of the previous step, or the boundary ----
                uint32_t rh[8], e[8];
                // E inputs: low half from the slot to the left (the previous lane's last slot for p == 0), high half from
                // this pair's own low slot, both as written one step ago
                uint32_t xin[8], tagin;
                {
                    const uint16_t* src_lo = (p > 0) ? xe_wave + lane + (2 * p - 1) * (9 * 64)
                                                     : xe_wave + left_lane + (((i - 1) & 1) ? S : S - 1) * (9 * 64);
                    const uint16_t* src_hi = xe_wave + lane + (2 * p) * (9 * 64);
#pragma unroll
                    for (int il = 0; il < 8; il++) xin[il] = (uint32_t)src_lo[il * 64] | ((uint32_t)src_hi[il * 64] << 16);
                    tagin = (uint32_t)src_lo[8 * 64] | ((uint32_t)src_hi[8 * 64] << 16);
                }
                const uint32_t okm = eq_mask(tagin, pk_sub(rc, ONE2), ONE2);
                // a row block inside the first w rows whose left input is missing starts from real gap scores
                const uint32_t FRMm = ACTm & ~okm & lt_mask(q2, dup2((uint32_t)(w / 8 + 1)), F15);
                if (__builtin_expect(__any(FRMm != 0u), 0)) {
                    // (agatha_kernel.h:126-131); general per-lane form of the entry rules
                    const bool oka = (okm & 1u) != 0u, okb = (okm >> 31) != 0u;
                    const int qa = i - ra, qb = i - rb;
                    const int tua = (UPm & 1u) ? t0 : (UP2m & 1u) ? t0 + 8 : 1000, tub = (UPm >> 31) ? t0 : (UP2m >> 31) ? t0 + 8 : 1000;
                    const int tla = (LOm & 1u) ? t0 : (LO2m & 1u) ? t0 + 8 : 1000, tlb = (LOm >> 31) ? t0 : (LO2m >> 31) ? t0 + 8 : 1000;
#pragma unroll
                    for (int il = 0; il < 8; il++) {
                        const int rowa = 8 * qa + il, rowb = 8 * qb + il;
                        // H(row, -1) = -(gapoe + ge row) seen from anti-diagonal row - 1, E(row, 0) = that - gapoe seen from row
                        const uint32_t hb_ = rep16(-(gapoe + ge) - base), eb_ = rep16(-2 * gapoe - base);
                        uint32_t iha = (rowa <= w) ? hb_ : (uint32_t)r16::NEG, iea = (rowa <= w) ? eb_ : (uint32_t)r16::NEG;
                        uint32_t ihb = (rowb <= w) ? hb_ : (uint32_t)r16::NEG, ieb = (rowb <= w) ? eb_ : (uint32_t)r16::NEG;
                        uint32_t vha = oka ? (XH[p][il] & 0xffffu) : iha, vea = oka ? (xin[il] & 0xffffu) : iea;
                        uint32_t vhb = okb ? (XH[p][il] >> 16) : ihb, veb = okb ? (xin[il] >> 16) : ieb;
                        // cell (il, 0) outside the band: its E is R_OUT; cell (il + 1, 0) outside: its diagonal (this H) is
                        if ((-il > tua) || (il > tla)) vea = r16::OUT;
                        if ((-il > tub) || (il > tlb)) veb = r16::OUT;
                        if (il < 7 && ((-(il + 1) > tua) || (il + 1 > tla))) vha = r16::OUT;
                        if (il < 7 && ((-(il + 1) > tub) || (il + 1 > tlb))) vhb = r16::OUT;
                        rh[il] = pk2(vha, vhb); e[il] = pk2(vea, veb);
                    }
                } else {
                    // lower edge blocks: every cell of column 0 below the cut is outside the band, so the missing left
                    // input is R_OUT there (for t0 = 0 cell (0, 0) is inside); everywhere else it is -infinity
                    const uint32_t dflt = bfi(LOm, OUT2, NEG2);
                    const uint32_t dflt_e0 = (t0 == 0) ? NEG2 : dflt;      // for T0 = 0 cell (0, 0) of a lower edge block is inside
#pragma unroll
                    for (int il = 0; il < 8; il++) {
                        rh[il] = bfi(okm, XH[p][il], il < 7 ? dflt : NEG2);
                        e[il] = bfi(okm, xin[il], il == 0 ? dflt_e0 : dflt);
                    }
                    // upper edge blocks: cells (il, 0) with -il > T0 are outside: their E, and the diagonal of the next row
#pragma unroll
                    for (int il = 0; il < 8; il++) {
                        if (il < -t0) e[il] = bfi(UPm, OUT2, e[il]);
                        if (il >= 1 && il < -t0) rh[il - 1] = bfi(UPm, OUT2, rh[il - 1]);
                    }
                    // the block next to the lower corner of the band has a left neighbour, but its cells (il, 0) with
                    // il > T0 + 8 are outside all the same
                    if (CORNER_BLOCKS) {
#pragma unroll
                        for (int il = 1; il < 8; il++) {
                            if (il > t0 + 8) { e[il] = bfi(LO2m, OUT2, e[il]); rh[il - 1] = bfi(LO2m, OUT2, rh[il - 1]); }
                        }
                    }
                }
                uint32_t corner_in = CORNER[p];
                if (t0 < 0) corner_in = bfi(UPm | LOm, OUT2, corner_in);
                // lower edge blocks: cells (0, jl) with -jl > T0 are outside: their F, and the diagonal of the next column
#pragma unroll
                for (int jl = 0; jl < 8; jl++) {
                    if (jl < -t0) F[p][jl] = bfi(LOm, OUT2, F[p][jl]);
                    if (jl >= 1 && jl < -t0) H[p][jl - 1] = bfi(LOm, OUT2, H[p][jl - 1]);
                }
                // blocks cut on T0 + 8 that are not the first row block of their column (the last row block): cells (0, jl)
                // with jl > T0 + 8 are outside (for block (0, W-1) the initial column state already says so)
                if (CORNER_BLOCKS) {
#pragma unroll
                    for (int jl = 1; jl < 8; jl++) {
                        if (jl > t0 + 8) { F[p][jl] = bfi(UP2m, OUT2, F[p][jl]); H[p][jl - 1] = bfi(UP2m, OUT2, H[p][jl - 1]); }
                    }
                }

                // ---- upper bounds of E and F on the cut diagonals: R_OUT in the halves of that block kind ----
                const uint32_t gcu = OUT2 | ~UPm, gcl = OUT2 | ~LOm;
                const uint32_t gcu2 = CORNER_BLOCKS ? (OUT2 | ~UP2m) : gcu, gcl2 = CORNER_BLOCKS ? (OUT2 | ~LO2m) : gcl;
                // rows that exist: 8, fewer in the last row block, 0 for an inactive half
                const uint32_t NR = bfi(eq_mask(q2, PQL1, ONE2), NRLAST, 0x00080008u) & ACTm;

                {
                    BlockOps o;
                    o.gapo2 = GAPO2; o.cu = gcu; o.cu2 = gcu2; o.cl = gcl; o.cl2 = gcl2; o.NRK = pk_shl_c(NR, K2);
                    o.qc_lo = class_word(qcls[2 * p]); o.qc_hi = class_word(qcls[2 * p + 1]);
                    o.crel_lo = 8 * ra - cb; o.crel_hi = 8 * rb - cb;
                    o.pl = prof0 + (2 * p) * (4 * 64); o.ph = prof0 + (2 * p + 1) * (4 * 64);
                    block_pair16<K, T0>(H[p], F[p], corner_in, rh, e, XH[p], A, o);
                }
                // lower edge blocks: the rows il > 7 + T0 end below the band; they hand on what the reference's skipped cells
                // leave in its registers: H of row 7 + T0 at column 7, and the incoming E (-infinity)
#pragma unroll
                for (int il = 1; il < 8; il++) {
                    if (il > 7 + t0) {
                        // (the same value, seen from il - (7 + T0) anti-diagonals further on)
                        XH[p][il] = bfi(LOm, pk_add_c(XH[p][7 + t0], dup2((uint32_t)(ge * (il - 7 - t0)))), XH[p][il]);
                        e[il] = bfi(LOm, NEG2, e[il]);
                    }
                }
                {
                    uint16_t* dst_lo = xe_wave + lane + (2 * p) * (9 * 64);
                    uint16_t* dst_hi = (p < P - 1) ? xe_wave + lane + (2 * p + 1) * (9 * 64) : xe_wave + lane + ((i & 1) ? S : S - 1) * (9 * 64);
#pragma unroll
                    for (int il = 0; il < 8; il++) { dst_lo[il * 64] = (uint16_t)e[il]; dst_hi[il * 64] = (uint16_t)(e[il] >> 16); }
                    const uint32_t tagout = bfi(ACTm, rc, NOTAG);
                    dst_lo[8 * 64] = (uint16_t)tagout; dst_hi[8 * 64] = (uint16_t)(tagout >> 16);
                }
                CORNER[p] = bfi(ACTm, rh[7], CORNER[p]);

                // a half that has not started its column yet was computed on garbage: put its initial state back
                // (inactive and not past the end of its column; a column that was started and then dropped by the slice
                // limits never comes back, so resetting it as well does no harm); a half that moves on starts from the
                // same constants (its new column is always beyond the first band width)
                const uint32_t RSm = ~ACTm & ~lt_mask(ce2, q2, F15);
                const uint32_t INm = RSm | ADm;
                // constants first (no branch: some lane of the wave needs it on almost every step) ...
#pragma unroll
                for (int m = 0; m < 8; m++) {
                    H[p][m] = bfi(INm, m < 7 ? HINIT : H7INIT, H[p][m]);
                    F[p][m] = bfi(INm, m == 0 ? F0INIT : FINIT, F[p][m]);
                }
                CORNER[p] = bfi(INm, CINIT, CORNER[p]);
                // ... then the real boundary values of the first band width of columns (only while a pair starts)
                if (__builtin_expect(__any(RSm != 0u && (ra <= W || rb <= W)), 0)) {
                    uint32_t ha[8], fa[8], ca, hb[8], fb[8], cbv;
                    int ral = ra, rbl = rb;
                    asm volatile("" : "+v"(ral), "+v"(rbl));
                    init_half(ral, R, w, W, gapoe, ge, base, ha, fa, ca);
                    init_half(rbl, R, w, W, gapoe, ge, base, hb, fb, cbv);
#pragma unroll
                    for (int m = 0; m < 8; m++) { H[p][m] = bfi(RSm, pk2(ha[m], hb[m]), H[p][m]); F[p][m] = bfi(RSm, pk2(fa[m], fb[m]), F[p][m]); }
                    CORNER[p] = bfi(RSm, pk2(ca, cbv), CORNER[p]);
                }
                // the reference words requested before the block have long arrived: consume them here, so that no later
                // register reuse has to wait for them together with the younger loads below
                asm volatile("" : : "v"(rwa), "v"(rwb));
                if (__any(ADm != 0u)) {
                    if (ADm & 1u) build_profile5(prof0 + (2 * p) * (4 * 64), rwa, plut);
                    if (ADm >> 31) build_profile5(prof0 + (2 * p + 1) * (4 * 64), rwb, plut);
                    RC[p] = pk_add(rc, ADm & dup2((uint32_t)GS));
                }
                // query words of the row blocks this pair of slots works on in step i + 1: requested now, used a step
                // later (raw: converting them here would wait for the load)
                {
                    const uint32_t rcn = RC[p];
                    // A half whose row block does not exist (yet) is inactive in that step and computes on garbage anyway, so
                    // the address is only clamped into the pair's query (no branch, no select on the loaded word).
                    const int qhi = imax(pql - 1, 0);
                    const int qna = imin(imax(i + 1 - (int)(rcn & 0xffffu), 0), qhi), qnb = imin(imax(i + 1 - (int)(rcn >> 16), 0), qhi);
                    qcls[2 * p] = ((gptr_t)La->packed_q)[pq + (uint32_t)qna];
                    qcls[2 * p + 1] = ((gptr_t)La->packed_q)[pq + (uint32_t)qnb];
                }
            }
        }

        __builtin_amdgcn_sched_barrier(0);
        if (__builtin_expect(__any(bail), 0)) {     // the decision belongs to the whole group
            const unsigned long long bm = __builtin_amdgcn_ballot_w64(bail);
            const unsigned long long gm = (G == 64) ? ~0ull : (((1ull << (G & 63)) - 1ull) << gbase);
            bail = (bm & gm) != 0ull;
        }

        // ---- hand-off: slot s feeds slot s + 1 (half swap inside a register pair, or the next register pair);
        //      the last slot feeds slot 0 of the next lane ----
        {
            uint32_t th[8];
#pragma unroll
            for (int il = 0; il < 8; il++) {
                // a 16-lane group is one DPP row: row_ror:1 hands every lane the value of its left neighbour (wrapping)
                // (a 64-lane group is the wave: wave_ror:1 does the same there)
                if (G == 16) th[il] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)XH[P - 1][il], 0x121, 0xf, 0xf, true);
                else if (G == 64) th[il] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)XH[P - 1][il], 0x13C, 0xf, 0xf, true);
                else th[il] = (uint32_t)lane_read((int)XH[P - 1][il], left_lane);
            }
#pragma unroll
            for (int p = P - 1; p >= 0; p--) {
#pragma unroll
                for (int il = 0; il < 8; il++) {
                    const uint32_t ph_ = (p > 0) ? XH[p > 0 ? p - 1 : 0][il] : th[il];
                    XH[p][il] = __builtin_amdgcn_alignbit(XH[p][il], ph_, 16);     // low = previous pair's high half, high = own low half
                }
            }
        }

        // ------------------------------------------------------------------ anti-diagonals 8i..8i+7 are complete
        bool stopped = false;
        int vred[8];
        // H fields are moved to the frame of the step's LAST anti-diagonal (+ (7 - x) ge): they only grow, so that the key
        // of an accumulator without a cell stays a small non-negative number (see row_keys4 for the + x)
#pragma unroll
        for (int x = 0; x < 8; x++) vred[x] = A[x] + (x + (7 - x) * frame_step);
        const int base_i = base - ge * (8 * i + 7);                         // value = rep - BIAS + base_i there
        // Per lane first, then ONE reduction of three numbers over the group instead of eight: the largest key of the step
        // (rebase, calm test), a LOWER BOUND of the smallest anti-diagonal maximum (the largest, over the lanes, of a
        // lane's smallest accumulator: the lane that holds the best cells has cells on all eight anti-diagonals), and the
        // best (H : earliest anti-diagonal : column) -- each key's order is kept by its own transformation, so the maximum
        // over lanes and anti-diagonals of the transformed keys is the transformed maximum.
        static_assert(K + 3 + 16 <= 32, "key of the fast path");
        int hi8 = max3i(vred[0], vred[1], vred[2]), lo8 = min3i(vred[0], vred[1], vred[2]);
        hi8 = max3i(hi8, vred[3], vred[4]); lo8 = min3i(lo8, vred[3], vred[4]);
        hi8 = max3i(hi8, vred[5], vred[6]); lo8 = min3i(lo8, vred[5], vred[6]);
        hi8 = imax(hi8, vred[7]); lo8 = imin(lo8, vred[7]);
        uint32_t mk = 0u;
#pragma unroll
        for (int x = 0; x < 8; x++) {
            const uint32_t v = (uint32_t)vred[x];
            const uint32_t k3 = ((v & ~(uint32_t)KMASK) << 3) | ((uint32_t)(7 - x) << K) | (v & (uint32_t)KMASK);
            mk = k3 > mk ? k3 : mk;
        }
        group_max3<G>(hi8, lo8, mk, lane);
        // Fast path (wave-uniform): every anti-diagonal of this step has an in-band maximum well inside its zone, inside
        // the pair, and within z of the running maximum, so neither z-drop nor the bail-out can fire.
        bool calm = !final_step && (8 * i + 7 < lim) && !bail;
        {
            const int lo_rep = lo8 >> K, lo_abs = lo_rep - r16::BIAS + base_i, hi_abs = (hi8 >> K) - r16::BIAS + base_i;
            calm = calm && lo_rep >= bail_rep && lo_abs >= NEG_INF2 + spread && (z < 0 || imax(best, hi_abs) - lo_abs <= z);
        }
        if (__builtin_expect(__all(calm || !alive), 1)) {
            // only the running maximum moves: the anti-diagonal with the largest H wins, the earliest one among equals
            // (the reference walks them in order and updates on H > max only)
            const int Hm = (int)(mk >> (K + 3)) - r16::BIAS + base_i;
            if (alive && Hm > best) {
                best = Hm; best_t = (int)(mk & (uint32_t)KMASK) + cb; best_q = 8 * i + 7 - (int)((mk >> K) & 7u) - best_t;
            }
        } else {
            group_max8<G>(vred, lane);
#pragma unroll
            for (int x = 0; x < 8; x++) {
                const int v = vred[x];
                const int d = 8 * i + x;
                const bool chk = alive && !stopped && !bail && (final_step || d < lim);      // agatha_kernel.h:293-294 / 337
                const int rep = v >> K;
                int Hv = rep - r16::BIAS + base_i, c = (v & KMASK) + cb;
                if (rep < r16::GLO) { Hv = -32768; c = 0; }                                // empty, or only out-of-band cells
                else if (chk && (rep < bail_rep || Hv < NEG_INF2 + spread)) bail = true;
                if (chk && !bail) {                                                        // agatha_kernel.h:297-309
                    if (Hv > best) { best = Hv; best_t = c; best_q = d - c; }
                    else if (c >= best_t && (d - c) >= best_q) {
                        const int tlen = c - best_t, qlen = (d - c) - best_q;
                        const int l = tlen > qlen ? tlen - qlen : qlen - tlen;
                        if (z >= 0 && best - Hv > z + l * ge) stopped = true;
                    }
                }
            }
        }
        bool finished = alive && (stopped || final_step);

        // carry dl 8..14 into the next step
#pragma unroll
        for (int x = 0; x < 7; x++) A[x] = A[8 + x] + 8;        // anti-diagonal 8 + x of this step is x of the next
#pragma unroll
        for (int x = 7; x < 15; x++) A[x] = 0;

        // ---- rebase: keep the representation of the running maximum small ----
        {
            const bool reb = alive && !finished && !bail && (hi8 >> K) > rebase_at;
            if (__builtin_expect(__any(reb), 0)) {
                const uint32_t D2 = reb ? dup2(r16::DELTA) : 0u;
                const uint32_t CAP2 = dup2(r16::LO - 1);
                // only in-band values follow the base: max(v - D, min(v, LO - 1)) is v - D for them, v for the rest
#pragma unroll
                for (int p = 0; p < P; p++) {
#pragma unroll
                    for (int m = 0; m < 8; m++) {
                        H[p][m] = pk_max(pk_sub(H[p][m], D2), pk_min(H[p][m], CAP2));
                        F[p][m] = pk_max(pk_sub(F[p][m], D2), pk_min(F[p][m], CAP2));
                        XH[p][m] = pk_max(pk_sub(XH[p][m], D2), pk_min(XH[p][m], CAP2));
                    }
                    CORNER[p] = pk_max(pk_sub(CORNER[p], D2), pk_min(CORNER[p], CAP2));
                }
                // the E hand-off values of this lane in LDS (all regions; the ones not in flight are dead)
                {
                    const int dl = reb ? r16::DELTA : 0;
                    uint16_t* xe_my = xe_wave + lane;
                    for (int g = 0; g < (S + 1) * 9; g++) {
                        if (g % 9 == 8) continue;       // the tag row
                        const int v = (int)xe_my[g * 64];
                        xe_my[g * 64] = (uint16_t)imax(v - dl, imin(v, r16::LO - 1));
                    }
                }
                const int dk = reb ? (r16::DELTA << K) : 0, capk = (r16::LO << K) - 1;
#pragma unroll
                for (int x = 0; x < 7; x++) A[x] = imax(A[x] + x - dk, imin(A[x] + x, capk)) - x;
                if (reb) base += r16::DELTA;
            }
        }

        // next step / next slice (agatha_kernel.h:183-191, 330-334)
        i++; y++;
        if (y == sw) {
            y = 0;
            if (i >= total) final_step = true;
            else {
                ss = imax(imax(0, i - pql + 1), ((i * 8 + 8 - w) / 2) / 8);
                se = imin(imin(prl - 1, i + sw - 1), (((i + sw - 1) * 8 + 7 + w) / 2) / 8);
                if (ss > se) finished = alive;       // empty slice: stop without checking it (:189-191)
            }
        }
        if (__builtin_expect(bail && alive, 0)) {
            // hand the pair to the int32 kernel (launched after this one on the same stream)
            if (k == 0) { La->exotic[pair] = 2; atomicAdd(La->kind_counts + 1, 1u); }
            alive = false;
        } else if (__builtin_expect(finished, 0)) {
            if (k == 0) { La->score[pair] = best; La->qend[pair] = best_q; La->tend[pair] = best_t; }   // :359-363
            alive = false;
        }
        } while (!__any(!alive && !exhausted));
    }
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
template <int G, int P, int T0>
static hipError_t launch_align16_t(const AlignLaunch& L, int kid, hipStream_t st)
{
    const int groups_per_block = (256 / 64) * (64 / G);
    int blocks = (L.n + groups_per_block - 1) / groups_per_block;
    int max_blocks = L.num_cus * 2;
    if (L.max_blocks_override > 0) max_blocks = L.max_blocks_override;
    if (blocks > max_blocks) blocks = max_blocks;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL((align16_kernel<G, P, T0>), dim3(blocks), dim3(256), 0, st, L.self_dev, L.p, kid);
    return hipGetLastError();
}

typedef hipError_t (*launch16_fn)(const AlignLaunch&, int, hipStream_t);
struct Cfg16 { int G, P; launch16_fn fn[8]; };        // fn[-T0]
#define AGATHA16_CFG(G, P) {G, P, {launch_align16_t<G, P, 0>, launch_align16_t<G, P, -1>, launch_align16_t<G, P, -2>, launch_align16_t<G, P, -3>, \
                                   launch_align16_t<G, P, -4>, launch_align16_t<G, P, -5>, launch_align16_t<G, P, -6>, launch_align16_t<G, P, -7>}}
static const Cfg16 kCfgs16[] = {       // ascending G * 2P: windows of 32, 64, 96, 128, 192 blocks; then the latency shapes
    AGATHA16_CFG(16, 1), AGATHA16_CFG(16, 2), AGATHA16_CFG(16, 3), AGATHA16_CFG(32, 2), AGATHA16_CFG(32, 3),
    AGATHA16_CFG(64, 1), AGATHA16_CFG(64, 2),
};

bool agatha16_scores_ok(const AlignParams& p)
{
    if (p.band_width < 16) return false;
    if (p.match < 0 || p.match > 16 || p.mismatch < 0 || p.mismatch > 32) return false;
    if (p.gap_open < 0 || p.gap_open > 64 || p.gap_extend < 0 || p.gap_extend > 16) return false;
    int per = 2 * p.gap_extend; if (p.mismatch > per) per = p.mismatch; if (per < 1) per = 1;
    const int spread = p.gap_open + p.gap_extend + per * (p.band_width + 16) + 64;
    return spread <= r16::MAX_SPREAD;
}

// the smallest packed-int16 configuration that holds the window, unless it would leave most of its slots idle
static const Cfg16* pick16(const AlignParams& p, int window_blocks)
{
    if (!agatha16_scores_ok(p)) return nullptr;
    for (const Cfg16& c : kCfgs16)
        if (c.G < 64 && c.G * 2 * c.P >= window_blocks) return (c.G * 2 * c.P <= 2 * window_blocks + 32) ? &c : nullptr;
    return nullptr;
}

bool align16_config(const AlignParams& p, int window_blocks, int* G, int* P, int* GL, int* PL)
{
    const Cfg16* c = pick16(p, window_blocks);
    if (!c) return false;
    *G = c->G; *P = c->P;
    *GL = 0; *PL = 0;
    for (const Cfg16& l : kCfgs16)            // latency shape: 64 lanes per pair, fewer register pairs per lane
        if (l.G == 64 && l.G * 2 * l.P >= window_blocks && l.P < c->P) { *GL = l.G; *PL = l.P; break; }
    return true;
}

hipError_t launch_align16(const AlignLaunch& L, int G, int P, int kid, hipStream_t st)
{
    const int W = (L.p.band_width + 7) / 8, t0 = L.p.band_width - 8 * W;
    for (const Cfg16& c : kCfgs16)
        if (c.G == G && c.P == P) return c.fn[-t0](L, kid, st);
    return hipErrorInvalidValue;
}

}  // namespace agatha
