/*
 * seq_ops_ref.c -- CPU restatement of the reference's per-sequence reverse / complement kernel, AS WRITTEN, next to
 * the semantics the product (agatha::seq_ops_kernel behind agatha_amd_seq_ops) implements.
 *
 * TEST INFRASTRUCTURE (like everything under oracle/): only tests/ may call it.
 *
 * Reference: AGAThA/src/kernels/pack_rc_seqs.h:56-212 (gasal_reversecomplement_kernel), one thread per pair, in place on
 * the packed words (8 bases per uint32, first base in bits 31-28).  Dead code in the reference (no flag ever sets
 * isReverseComplement, args_parser.cpp:28), so its behaviour is defined by its text alone:
 *   - :111-116 count the padding bases of the last word by comparing each NIBBLE with N_CODE, which the build defines
 *     as 0x4E (AGAThA/Makefile:4): a nibble never equals 0x4E, so nbr_N is always 0;
 *   - with nbr_N = 0 the funnel shifts at :137,:155,:156 shift a 32-bit word by 32.  That is undefined in C; on the
 *     hardware the reference targets a 32-bit shift by >= 32 gives 0 (PTX shl.b32 / shr.b32 clamp the amount), which is
 *     what shl32 / shr32 below restate.  (g++ on x86 would mask the amount to 0 instead, which is why this kernel is
 *     NOT run under oracle/ref_shim: the emulation would not be the reference's behaviour.)
 *   Net effect as written: the whole PADDED sequence (8 * ceil(len/8) nibbles, padding included) is reversed, so for
 *   len % 8 != 0 the padding Ns end up in FRONT of the sequence.  For len % 8 == 0 it is the plain reversal.
 * The product reverses exactly `len` bases and keeps the padding behind them; the two agree iff
 * len % 8 == 0, and tests/test_oracle.py + tests/test_gpu_parity.py pin both facts.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define REF_N_CODE 0x4E              /* AGAThA/Makefile:4 */
#define A_PAK ('A' & 0x0F)           /* pack_rc_seqs.h:5-8 */
#define C_PAK ('C' & 0x0F)
#define G_PAK ('G' & 0x0F)
#define T_PAK ('T' & 0x0F)

static uint32_t shl32(uint32_t x, uint32_t n) { return n >= 32u ? 0u : x << n; }
static uint32_t shr32(uint32_t x, uint32_t n) { return n >= 32u ? 0u : x >> n; }

/* one side (query or target) of one pair: pack_rc_seqs.h:107-205.  `base` points at the first packed word of the
 * sequence. */
static void ref_side_as_written(uint32_t* base, uint32_t len, uint8_t op)
{
    const uint32_t regs = (len >> 3) + ((len & 7u) ? 1u : 0u);          /* :70-71 */
    const uint32_t regs_to_swap = (regs >> 1) + (regs & 1u);            /* :73-74 */
    if (regs == 0u) return;
    if (op & 0x01) {                                                    /* reverse, :107-166 */
        uint32_t nbr_N = 0;
        for (int j = 0; j < 32; j += 4)                                 /* :113-116: a nibble against 0x4E */
            nbr_N += (((base[regs - 1] & (0x0Fu << j)) >> j) == REF_N_CODE);
        nbr_N <<= 2;                                                    /* :121 */
        for (uint32_t i = 0; i < regs_to_swap; i++) {                   /* :123-165 */
            const uint32_t w_tail1 = base[(int64_t)regs - 1 - i];
            /* (for a one-word sequence the reference reads the word BEFORE the sequence here; with nbr_N = 0 that value is
             *  shifted out below and never written back, so 0 stands in for it) */
            const int64_t i2 = (int64_t)regs - 2 - (int64_t)i;
            const uint32_t w_tail2 = i2 >= 0 ? base[i2] : 0u;
            const uint32_t rpac_1 = base[i];                                                   /* :136 */
            const uint32_t rpac_2 = shl32(w_tail2, 32u - nbr_N) | shr32(w_tail1, nbr_N);       /* :137 */
            uint32_t rev1 = 0, rev2 = 0;
            for (int k = 28; k >= 0; k -= 4) {                                                 /* :145-149 */
                rev1 |= ((rpac_1 & (0x0Fu << k)) >> k) << (28 - k);
                rev2 |= ((rpac_2 & (0x0Fu << k)) >> k) << (28 - k);
            }
            const uint32_t to_queue_1 = shl32(rev1, nbr_N) | (w_tail1 & (shl32(1u, nbr_N) - 1u));                   /* :153 */
            const uint32_t to_queue_2 = (w_tail2 & (0xFFFFFFFFu - (shl32(1u, nbr_N) - 1u))) | shr32(rev1, 32u - nbr_N); /* :154 */
            base[i] = rev2;                                                                    /* :161 */
            base[(int64_t)regs - 1 - i] = to_queue_1;                                          /* :162 */
            if (i != regs_to_swap - 1u) base[i2] = to_queue_2;                                 /* :163-164 */
        }
    }
    if (op & 0x02) {                                                    /* complement, :168-205 */
        for (uint32_t i = 0; i < regs; i++) {
            uint32_t rpac = base[i];
            for (int k = 28; k >= 0; k -= 4) {
                uint32_t nt = (rpac & (0x0Fu << k)) >> k;
                switch (nt) {                                           /* :180-197: every other code is left alone */
                    case A_PAK: nt = T_PAK; break;
                    case C_PAK: nt = G_PAK; break;
                    case T_PAK: nt = A_PAK; break;
                    case G_PAK: nt = C_PAK; break;
                    default: break;
                }
                rpac = (rpac & (0xFFFFFFFFu - (0x0Fu << k))) | (nt << k);
            }
            base[i] = rpac;
        }
    }
}

/* the reference kernel over one side of a batch (its thread loop handles both sides of pair `tid` with the same code;
 * they are independent, so one side at a time is the same thing).  packed: words of the packed batch, in place. */
void agatha_ref_seq_ops_as_written(uint32_t* packed, const uint32_t* lens, const uint32_t* offsets, const uint8_t* ops, int n)
{
    for (int t = 0; t < n; t++)
        if (ops[t] & 3u) ref_side_as_written(packed + (offsets[t] >> 3), lens[t], ops[t]);
}

/* what the product implements: reverse exactly `len` bases, complement A<->T, C<->G, padding (N) stays at the end */
void agatha_seq_ops_product_semantics(uint32_t* packed, const uint32_t* lens, const uint32_t* offsets, const uint8_t* ops, int n)
{
    for (int t = 0; t < n; t++) {
        const uint8_t op = ops[t] & 3u;
        if (!op) continue;
        uint32_t* base = packed + (offsets[t] >> 3);
        const uint32_t len = lens[t], regs = (len + 7u) >> 3;
        if (regs == 0u) continue;
        uint32_t* old = (uint32_t*)malloc((size_t)regs * sizeof(uint32_t));
        memcpy(old, base, (size_t)regs * sizeof(uint32_t));
        for (uint32_t wi = 0; wi < regs; wi++) {
            uint32_t v = 0;
            for (uint32_t k = 0; k < 8u; k++) {
                const uint32_t p = 8u * wi + k;
                uint32_t c = 14u;                                       /* 'N' & 15 */
                if (p < len) {
                    const uint32_t src = (op & 1u) ? (len - 1u - p) : p;
                    c = (old[src >> 3] >> (28u - 4u * (src & 7u))) & 15u;
                    if (op & 2u) c = c == A_PAK ? T_PAK : c == T_PAK ? A_PAK : c == C_PAK ? G_PAK : c == G_PAK ? C_PAK : c;
                }
                v |= c << (28u - 4u * k);
            }
            base[wi] = v;
        }
        free(old);
    }
}
