/*
 * agatha_oracle.c -- CPU restatement of AGAThA's guided-alignment kernel.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped product
 * path: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and there only as the checker / reported CPU baseline.
 *
 * PARITY STATUS: the reference (readwrite112/AGAThA) ships no tests, no golden
 * vectors and no dataset, and it is CUDA-only, so it cannot be built natively in
 * this image ("parity unpinned" by a native reference build).  What this oracle IS
 * pinned against: (1) the known-answer vectors of SURVEY.md Appendix E
 * (tests/golden/kat_appendix_e.json), several of them hand-checked; (2) golden
 * vectors obtained by executing the UNMODIFIED reference kernel source
 * (AGAThA/src/kernels/agatha_kernel.h) under the CPU warp emulator in
 * oracle/ref_shim/ (an emulation of the CUDA execution model, not a native CUDA
 * build -- see DESIGN.md "Oracle").
 *
 * Two implementations of the same semantics live here and are cross-checked:
 *
 *   agatha_model_slices()   literal sequential restatement of one pair, following
 *                           the reference's slice/pass order
 *                             cell update      agatha_kernel.h:20-46
 *                             state init       agatha_kernel.h:126-153
 *                             slice range      agatha_kernel.h:183-191
 *                             block load       agatha_kernel.h:204-227
 *                             block sweep      agatha_kernel.h:230-281
 *                             max / z-drop     agatha_kernel.h:288-314, 337-356
 *                             outputs          agatha_kernel.h:359-363
 *                             N scoring, -inf  gasal_kernels.h:38-50
 *                             packing/padding  pack_rc_seqs.h:21-33, host_batch.cpp:100-146
 *   agatha_model_steps()    the same arithmetic re-scheduled one block-anti-diagonal
 *                           ("step") at a time with eager z-drop checks: the schedule
 *                           the HIP kernel uses (agatha_amd/csrc/align_kernel.hip).
 *
 * mode: 0 = "faithful" (int16 storage of strip state, 16:16 packed anti-diagonal
 *           maxima, exactly the reference's arithmetic incl. its wrap-around),
 *       1 = "wide" (int32 state, (H, col) compared lexicographically): identical to
 *           faithful whenever every H fits int16 and every length < 32768.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>

#define NEG_INF2 (-16384)           /* SHRT_MIN/2, gasal_kernels.h:39 */
#define N_VALUE 14                  /* 'N' & 0xF, AGAThA/Makefile:4, gasal_kernels.h:41 */

typedef struct {
    int32_t match, mismatch, gap_open, gap_extend, slice_width, z_threshold, band_width;
} oracle_params_t;                  /* field order of gasal_subst_scores, gasal.h:165-173 */

typedef struct { int32_t score, query_end, target_end; } oracle_result_t;

static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int imin(int a, int b) { return a < b ? a : b; }

/* gasal_kernels.h:48-50 with N_PENALTY=1 */
static inline int sub_score(int x, int y, int a, int b)
{
    int s = (x == y) ? a : -b;
    return (x == N_VALUE || y == N_VALUE) ? -1 : s;
}

/* ASCII -> 4-bit code, padded with N to a multiple of 8 (pack_rc_seqs.h:21-33, host_batch.cpp:143-146) */
static uint8_t *encode_padded(const char *s, int len, int *padded_len)
{
    int p8 = (len + 7) & ~7;
    uint8_t *c = (uint8_t *)malloc((size_t)p8 + 8);
    for (int i = 0; i < len; i++) c[i] = (uint8_t)(s[i] & 15);
    for (int i = len; i < p8 + 8; i++) c[i] = N_VALUE;
    *padded_len = p8;
    return c;
}

/* per-anti-diagonal maximum: faithful = one packed int32; wide = (H, col) */
typedef struct { int32_t h; int32_t c; int32_t packed; } dmax_t;

static inline void dmax_reset(dmax_t *d) { d->packed = INT_MIN; d->h = -32768; d->c = 0; }

static inline void dmax_update(dmax_t *d, int h, int c, int wide)
{
    if (!wide) {
        int32_t v = (int32_t)(((uint32_t)h << 16) + (uint32_t)c);   /* agatha_kernel.h:30 */
        if (v > d->packed) d->packed = v;
    } else {
        if (d->packed == INT_MIN || h > d->h || (h == d->h && c > d->c)) { d->h = h; d->c = c; d->packed = 0; }
    }
}
static inline void dmax_get(const dmax_t *d, int wide, int *h, int *c)
{
    if (!wide) { *h = d->packed >> 16; *c = d->packed & 65535; }    /* agatha_kernel.h:297-299 */
    else if (d->packed == INT_MIN) { *h = -32768; *c = 0; }
    else { *h = d->h; *c = d->c; }
}

typedef struct {
    int best, best_t, best_q, stopped;
} zstate_t;

/* agatha_kernel.h:297-309 (and 341-353) */
static inline void zdrop_check(zstate_t *zs, int H, int c, int d, int z, int ge)
{
    if (H > zs->best) { zs->best = H; zs->best_t = c; zs->best_q = d - c; }
    else if (c >= zs->best_t && (d - c) >= zs->best_q) {
        int tl = c - zs->best_t, ql = (d - c) - zs->best_q;
        int l = tl > ql ? tl - ql : ql - tl;
        if (z >= 0 && zs->best - H > z + l * ge) zs->stopped = 1;
    }
}

#define ST16(x) (wide ? (int32_t)(x) : (int32_t)(int16_t)(x))

/* One 8x8 block (q, r): agatha_kernel.h:204-281 restricted to one y-step.
 * h[1..8], f[1..8], p[1..8] persist across consecutive blocks of a pass exactly as the
 * reference's registers do.  Returns nothing; updates strips and dmax. */
typedef struct {
    int32_t *rowH, *rowE, *colH, *colF, *corner;
    dmax_t *dmax; int dmax_mod;            /* ring size (power of two not required here) */
    const uint8_t *qc, *rc;
    int Q, R, a, b, gapoe, ge, w, wide;
    uint8_t *tb; int tb_half;              /* traceback codes of the computed cells, band-indexed (NULL: not recorded) */
} ctx_t;

/* Traceback code of a cell (not in the reference, which never fills gasal_res.cigar, gasal.h:91-92):
 * bits 0-1 = which term H took (1 the diagonal term t, 2 E, 3 F; on ties in that order; 0 = the cell was never computed),
 * bit 2 = the E that leaves the cell to the right extends the E that entered it (e - ge > t - gapoe, opening wins ties),
 * bit 3 = the same for F. */
static inline size_t tb_index(const ctx_t *cx, int i, int c) { return (size_t)i * (size_t)(2 * cx->tb_half + 1) + (size_t)(c - i + cx->tb_half); }

static void block_rows(ctx_t *cx, int q, int r, int cs, int ce, int32_t *h, int32_t *f, int32_t *p)
{
    const int wide = cx->wide;
    int boundary = (q == cs || q == ce);                            /* :243 */
    for (int i = 8 * q; i < 8 * q + 8 && i < cx->Q; i++) {          /* :236 */
        int qb = cx->qc[i];
        int e;
        h[0] = cx->rowH[i]; e = cx->rowE[i];                        /* :239-241 */
        for (int m = 1; m <= 8; m++) {
            int c = 8 * r + m - 1;
            if (boundary && (i + cx->w < c || i - cx->w > c)) {     /* :33-35 */
                p[m] = h[m - 1];
            } else {
                int t = sub_score(qb, cx->rc[c], cx->a, cx->b) + p[m];
                h[m] = imax(imax(t, f[m]), e);
                if (cx->tb) {
                    int code = (h[m] == t) ? 1 : (h[m] == e ? 2 : 3);
                    if (e - cx->ge > t - cx->gapoe) code |= 4;
                    if (f[m] - cx->ge > t - cx->gapoe) code |= 8;
                    cx->tb[tb_index(cx, i, c)] = (uint8_t)code;
                }
                f[m] = imax(t - cx->gapoe, f[m] - cx->ge);
                e = imax(t - cx->gapoe, e - cx->ge);
                p[m] = h[m - 1];
                dmax_update(&cx->dmax[(i + c) % cx->dmax_mod], h[m], c, wide);
            }
        }
        cx->rowH[i] = ST16(h[8]); cx->rowE[i] = ST16(e);            /* :256-258 */
    }
}

static void load_colstate(ctx_t *cx, int r, int32_t *h, int32_t *f, int32_t *p)
{
    p[1] = cx->corner[r];                                           /* :204 */
    for (int m = 1; m <= 8; m++) {
        int c = 8 * r + m - 1;
        if (c < cx->R) { h[m] = cx->colH[c]; f[m] = cx->colF[c]; }  /* :207-215 */
        else { h[m] = NEG_INF2; f[m] = NEG_INF2; }
    }
    for (int m = 2; m <= 8; m++) p[m] = h[m - 1];                   /* :219-221 */
}
static void store_colstate(ctx_t *cx, int r, const int32_t *h, const int32_t *f, const int32_t *p)
{
    const int wide = cx->wide;
    for (int m = 1; m <= 8; m++) {
        int c = 8 * r + m - 1;
        if (c < cx->R) { cx->colH[c] = ST16(h[m]); cx->colF[c] = ST16(f[m]); }   /* :272-279 */
    }
    cx->corner[r] = p[1];                                           /* :280 */
}

static void init_strips(ctx_t *cx, int L)
{
    const int wide = cx->wide;
    for (int l = 0; l < L; l++) {                                   /* :126-148 */
        int k = -(cx->gapoe + cx->ge * l);
        if (l <= cx->w) { cx->rowH[l] = ST16(k); cx->rowE[l] = ST16(k - cx->gapoe); }
        else { cx->rowH[l] = NEG_INF2; cx->rowE[l] = NEG_INF2; }
        cx->colH[l] = cx->rowH[l]; cx->colF[l] = cx->rowE[l];
        int kk = -(cx->gapoe + cx->ge * (l * 8 - 1));
        cx->corner[l] = (l == 0) ? 0 : ((l * 8 - 1) <= cx->w ? kk : NEG_INF2);
    }
}

static int setup_ctx(ctx_t *cx, const char *qs, int Q, const char *rs, int R, const oracle_params_t *pr,
                     int wide, uint8_t **qc, uint8_t **rc, int ring)
{
    int qp, rp;
    *qc = encode_padded(qs, Q, &qp);
    *rc = encode_padded(rs, R, &rp);
    int L = imax(qp, rp) + 8;
    cx->rowH = (int32_t *)malloc(sizeof(int32_t) * (size_t)L * 5);
    cx->rowE = cx->rowH + L; cx->colH = cx->rowE + L; cx->colF = cx->colH + L; cx->corner = cx->colF + L;
    cx->dmax = (dmax_t *)malloc(sizeof(dmax_t) * (size_t)ring);
    cx->dmax_mod = ring;
    for (int i = 0; i < ring; i++) dmax_reset(&cx->dmax[i]);
    cx->qc = *qc; cx->rc = *rc; cx->Q = Q; cx->R = R;
    cx->a = pr->match; cx->b = pr->mismatch; cx->gapoe = pr->gap_open + pr->gap_extend; cx->ge = pr->gap_extend;
    cx->w = pr->band_width; cx->wide = wide; cx->tb = NULL; cx->tb_half = 0;
    init_strips(cx, L);
    return L;
}

/* ------------------------------------------------------------------------------------------ */
/* Literal slice/pass order.                                            */
/* ------------------------------------------------------------------------------------------ */
static void model_slices(const char *qs, int Q, const char *rs, int R, const oracle_params_t *pr,
                         int wide, oracle_result_t *out, uint8_t *tb, int tb_half)
{
    ctx_t cx; uint8_t *qc, *rc;
    const int sw = pr->slice_width, w = pr->band_width, z = pr->z_threshold, ge = pr->gap_extend;
    const int ring = 8 * (sw + 1);                                  /* total_shm, :83 */
    setup_ctx(&cx, qs, Q, rs, R, pr, wide, &qc, &rc, ring);
    cx.tb = tb; cx.tb_half = tb_half;
    const int pql = (Q + 7) / 8, prl = (R + 7) / 8;
    int total = prl + pql - 1;                                      /* :165 */
    zstate_t zs = {0, 0, 0, 0};
    int32_t h[9], f[9], p[9];
    int i0 = 0;
    while (i0 < total) {                                            /* :180 */
        int ss = imax(0, i0 - pql + 1);
        ss = imax(ss, (i0 * 8 + 8 - 1 + 1 - w) / 2 / 8);            /* :184 */
        int se = imin(prl - 1, i0 + sw - 1);
        se = imin(se, ((i0 + sw - 1) * 8 + 8 - 1 + w) / 2 / 8);     /* :186 */
        if (ss > se) zs.stopped = 1;                                /* :189-191 */
        if (!zs.stopped) {
            for (int r = ss; r <= se; r++) {                        /* :193-284, any r-ascending order */
                load_colstate(&cx, r, h, f, p);
                int cs = imax(0, r * 8 - w) / 8;                    /* :224 */
                int ce = imin(pql - 1, (r * 8 + 8 - 1 + w) / 8);    /* :225 */
                for (int y = 0; y < sw; y++) {                      /* :230 */
                    int q = i0 - r + y;
                    if (q >= cs && q <= ce) block_rows(&cx, q, r, cs, ce, h, f, p);
                }
                store_colstate(&cx, r, h, f, p);
            }
            int last = (i0 + sw) * 8, lim = Q + R - 1;              /* :288-289 */
            for (int d = i0 * 8; d < last; d++) {                   /* :293 */
                if (d < lim) {
                    int H, c; dmax_get(&cx.dmax[d % ring], wide, &H, &c);
                    zdrop_check(&zs, H, c, d, z, ge);
                    if (zs.stopped) break;                          /* :306-307 (no reset after break) */
                    dmax_reset(&cx.dmax[d % ring]);                 /* :311 */
                }
            }
        }
        if (zs.stopped) total = i0;                                 /* :319-322 */
        i0 += sw;                                                   /* :330 */
        if (i0 >= total) {                                          /* :334 */
            if (!zs.stopped) {                                      /* :337-356 */
                for (int k = i0 * 8; k < i0 * 8 + 8; k++) {
                    int H, c; dmax_get(&cx.dmax[k % ring], wide, &H, &c);
                    zdrop_check(&zs, H, c, k, z, ge);
                    if (zs.stopped) break;
                    dmax_reset(&cx.dmax[k % ring]);
                }
            }
        }
    }
    out->score = zs.best; out->query_end = zs.best_q; out->target_end = zs.best_t;   /* :359-363 */
    free(cx.rowH); free(cx.dmax); free(qc); free(rc);
}

void agatha_model_slices(const char *qs, int Q, const char *rs, int R, const oracle_params_t *pr,
                         int wide, oracle_result_t *out)
{
    model_slices(qs, Q, rs, R, pr, wide, out, NULL, 0);
}

/* ------------------------------------------------------------------------------------------ */
/* Traceback.  The reference declares gasal_res.cigar / n_cigar_ops (gasal.h:91-92) and never fills    */
/* them (res.cpp:27-28); GASAL2, which it derives from, publishes the byte format used here: one byte per run,           */
/* (count << 2) | op with op 0 = match, 1 = mismatch, 2 = D (target base against a gap), 3 = I (query base against a     */
/* gap), count <= 63, longer runs split greedily from the start; the bytes run from the alignment's first column (always */
/* the origin: this is an extension alignment) to its end cell (query_end, target_end).                                   */
/* The path is the one the recurrence of block_rows() took, with its tie-breaks (diagonal, then E, then F; opening a     */
/* gap before extending one) and its rule that gaps open from the diagonal TERM of a cell, not from its H.               */
/* A pair whose score is 0 never had a positive cell: its alignment is empty (0 ops).                                    */
/* The reference's block-granular band lets a boundary block read the stale register of a cell it skipped                */
/* (agatha_kernel.h:33-35): the few scores that come through such a cell belong to no alignment, the    */
/* walk then meets a cell that was never computed and the pair is reported as having no path.                             */
/* Returns the number of bytes written, or -1 (no path, or `cap` too small).                                              */
/* ------------------------------------------------------------------------------------------ */
static int tb_is_match(const uint8_t *qc, const uint8_t *rc, int i, int j)
{
    return qc[i] == rc[j] && qc[i] != N_VALUE;
}

int agatha_model_traceback(const char *qs, int Q, const char *rs, int R, const oracle_params_t *pr,
                           oracle_result_t *out, uint8_t *cigar, int cap)
{
    const int half = pr->band_width + 16;
    if (Q <= 0 || R <= 0) { out->score = 0; out->query_end = 0; out->target_end = 0; return 0; }
    uint8_t *tb = (uint8_t *)malloc((size_t)Q * (size_t)(2 * half + 1));
    memset(tb, 0, (size_t)Q * (size_t)(2 * half + 1));
    model_slices(qs, Q, rs, R, pr, 1, out, tb, half);
    if (out->score <= 0) { free(tb); return 0; }
    ctx_t ix; ix.tb_half = half;
    int qp, rp;
    uint8_t *qc = encode_padded(qs, Q, &qp), *rc = encode_padded(rs, R, &rp);
    uint8_t *ops = (uint8_t *)malloc((size_t)Q + (size_t)R + 8);     /* single ops, end to start */
    int n = 0, i = out->query_end, j = out->target_end, state = 0, bad = 0;
    while (i >= 0 && j >= 0 && !bad) {
        int code = tb[tb_index(&ix, i, j)];
        if (code == 0) { bad = 1; break; }
        if (state == 0) {
            int d = code & 3;
            if (d == 1) { ops[n++] = tb_is_match(qc, rc, i, j) ? 0 : 1; i--; j--; }
            else state = d - 1;                   /* 1: H = E(i, j), 2: H = F(i, j) */
        } else if (state == 1) {                  /* E(i, j): target base j against a gap; formed in cell (i, j - 1) */
            ops[n++] = 2; j--;
            if (j < 0) break;                     /* opened from the boundary column H(i, -1) */
            code = tb[tb_index(&ix, i, j)];
            if (code == 0) { bad = 1; break; }
            if (!(code & 4)) { ops[n++] = tb_is_match(qc, rc, i, j) ? 0 : 1; i--; j--; state = 0; }
        } else {                                  /* F(i, j): query base i against a gap; formed in cell (i - 1, j) */
            ops[n++] = 3; i--;
            if (i < 0) break;                     /* opened from the boundary row H(-1, j) */
            code = tb[tb_index(&ix, i, j)];
            if (code == 0) { bad = 1; break; }
            if (!(code & 8)) { ops[n++] = tb_is_match(qc, rc, i, j) ? 0 : 1; i--; j--; state = 0; }
        }
    }
    /* what is left of one sequence lies against the boundary gap */
    while (!bad && i >= 0) { ops[n++] = 3; i--; }
    while (!bad && j >= 0) { ops[n++] = 2; j--; }
    int nb = 0;
    for (int k = n - 1; k >= 0 && !bad; ) {
        int op = ops[k], run = 0;
        while (k >= 0 && ops[k] == op && run < 63) { run++; k--; }
        if (nb >= cap) { bad = 1; break; }
        cigar[nb++] = (uint8_t)((run << 2) | op);
    }
    free(ops); free(tb); free(qc); free(rc);
    return bad ? -1 : nb;
}

/* ------------------------------------------------------------------------------------------ */
/* Step order: one block-anti-diagonal at a time, column state "stationary" per column block,   */
/* row state handed from block (q,r) to block (q,r+1), eager z-drop checks.  This is the         */
/* schedule of the HIP kernel; it must agree with agatha_model_slices() bit for bit.             */
/* ------------------------------------------------------------------------------------------ */
/* Step statistics for the design of the kernel's fast path (tools/step_stats.py; not a result): when non-NULL,
 * agatha_model_steps() fills st[0] = steps run, st[1] = step of the last rise of the running maximum (-1: none),
 * st[2] = steps (with all 8 anti-diagonals inside the pair) on which z-drop could fire judging by the values alone
 * (max(best, largest H) - smallest anti-diagonal maximum > z), st[3] = the first such step (-1: none),
 * st[4] = 1 if the pair stopped on z-drop, st[5] = steps that raised the maximum, st[6] = total steps of the pair. */
__thread int32_t *agatha_steps_stats = 0;
__thread int32_t *agatha_steps_trace = 0;      /* tools/cliff_sweep.py: the running maximum behind every step */

void agatha_model_steps(const char *qs, int Q, const char *rs, int R, const oracle_params_t *pr,
                        int wide, oracle_result_t *out)
{
    ctx_t cx; uint8_t *qc, *rc;
    const int sw = pr->slice_width, w = pr->band_width, z = pr->z_threshold, ge = pr->gap_extend;
    const int pql = (Q + 7) / 8, prl = (R + 7) / 8;
    const int total = prl + pql - 1;
    const int ring = 8 * (total + sw + 4);        /* no aliasing: one entry per anti-diagonal */
    setup_ctx(&cx, qs, Q, rs, R, pr, wide, &qc, &rc, ring);
    zstate_t zs = {0, 0, 0, 0};
    /* per-column-block registers (the HIP kernel keeps these in VGPRs) */
    int32_t (*H)[9] = (int32_t (*)[9])malloc(sizeof(int32_t[9]) * (size_t)(prl + 1) * 3);
    int32_t (*F)[9] = H + (prl + 1), (*P)[9] = F + (prl + 1);
    uint8_t *started = (uint8_t *)calloc((size_t)prl + 1, 1);
    const int lim = Q + R - 1;
    int i0 = 0, done = 0;
    while (i0 < total && !done) {
        int ss = imax(0, i0 - pql + 1);
        ss = imax(ss, (i0 * 8 + 8 - w) / 2 / 8);
        int se = imin(prl - 1, i0 + sw - 1);
        se = imin(se, ((i0 + sw - 1) * 8 + 7 + w) / 2 / 8);
        if (ss > se) { zs.stopped = 1; break; }
        /* slice start: the reference reloads column state from the strips here; the only
         * observable effect is that padded ref columns (c >= R) fall back to -inf and the
         * diagonal chain p[m] is rebuilt from h[m-1]  (agatha_kernel.h:204-221). */
        for (int y = 0; y < sw && !zs.stopped; y++) {
            int i = i0 + y;
            for (int r = ss; r <= se; r++) {
                int q = i - r;
                int cs = imax(0, r * 8 - w) / 8;
                int ce = imin(pql - 1, (r * 8 + 7 + w) / 8);
                if (q < cs || q > ce) continue;
                if (!started[r]) {            /* first touch of this column block: formula init */
                    started[r] = 1;
                    load_colstate(&cx, r, H[r], F[r], P[r]);
                } else if (y == 0 || i - 1 - r < cs /* not contiguous: cannot happen */) {
                    /* new pass on an already-started column block */
                    for (int m = 1; m <= 8; m++) if (8 * r + m - 1 >= R) { H[r][m] = NEG_INF2; F[r][m] = NEG_INF2; }
                    for (int m = 2; m <= 8; m++) P[r][m] = H[r][m - 1];
                    if (!wide) { for (int m = 1; m <= 8; m++) { H[r][m] = (int16_t)H[r][m]; F[r][m] = (int16_t)F[r][m]; }
                                 for (int m = 2; m <= 8; m++) P[r][m] = H[r][m - 1]; }
                }
                block_rows(&cx, q, r, cs, ce, H[r], F[r], P[r]);
            }
            if (agatha_steps_stats) {
                int32_t *st = agatha_steps_stats;
                int hmin = INT_MAX, hmax = INT_MIN;
                for (int d = 8 * i; d < 8 * i + 8 && d < lim; d++) {
                    int Hh, c; dmax_get(&cx.dmax[d % ring], wide, &Hh, &c);
                    if (Hh < hmin) hmin = Hh;
                    if (Hh > hmax) hmax = Hh;
                }
                st[0] = i + 1;
                if (8 * i + 7 < lim && z >= 0 && imax(zs.best, hmax) - hmin > z) { st[2]++; if (st[3] < 0) st[3] = i; }
                if (hmax > zs.best) { st[1] = i; st[5]++; }
                if (agatha_steps_trace) agatha_steps_trace[i] = imax(zs.best, hmax);
            }
            /* anti-diagonals 8i..8i+7 are complete after step i */
            for (int d = 8 * i; d < 8 * i + 8 && d < lim; d++) {
                int Hh, c; dmax_get(&cx.dmax[d % ring], wide, &Hh, &c);
                zdrop_check(&zs, Hh, c, d, z, ge);
                if (zs.stopped) break;
            }
        }
        if (zs.stopped) break;
        /* steps of this slice beyond its last block row still own anti-diagonals (:293-294) */
        i0 += sw;
        if (i0 >= total) {
            for (int k = i0 * 8; k < i0 * 8 + 8; k++) {
                int Hh, c; dmax_get(&cx.dmax[k % ring], wide, &Hh, &c);
                zdrop_check(&zs, Hh, c, k, z, ge);
                if (zs.stopped) break;
            }
            done = 1;
        }
    }
    out->score = zs.best; out->query_end = zs.best_q; out->target_end = zs.best_t;
    if (agatha_steps_stats) { agatha_steps_stats[4] = zs.stopped && !done && i0 < total; agatha_steps_stats[6] = total; }
    free(H); free(started); free(cx.rowH); free(cx.dmax); free(qc); free(rc);
}

/* one pair: stats7 as below, trace[i] = the running maximum behind step i (trace holds ceil(Q/8) + ceil(R/8) entries) */
void agatha_steps_trace_pair(const char *qs, int Q, const char *rs, int R, const oracle_params_t *pr, int32_t *stats7, int32_t *trace)
{
    oracle_result_t res;
    stats7[0] = 0; stats7[1] = -1; stats7[2] = 0; stats7[3] = -1; stats7[4] = 0; stats7[5] = 0; stats7[6] = 0;
    agatha_steps_stats = stats7; agatha_steps_trace = trace;
    agatha_model_steps(qs, Q, rs, R, pr, 1, &res);
    agatha_steps_stats = 0; agatha_steps_trace = 0;
}

void agatha_steps_stats_batch(const uint8_t *qbatch, const uint8_t *tbatch, const uint32_t *qoff, const uint32_t *toff,
                              const uint32_t *qlen, const uint32_t *tlen, int n, const oracle_params_t *pr, int threads,
                              int32_t *stats7)
{
    (void)threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 0 ? threads : 1)
#endif
    for (int k = 0; k < n; k++) {
        oracle_result_t res;
        int32_t *st = stats7 + 7 * (size_t)k;
        st[0] = 0; st[1] = -1; st[2] = 0; st[3] = -1; st[4] = 0; st[5] = 0; st[6] = 0;
        agatha_steps_stats = st;
        agatha_model_steps((const char *)qbatch + qoff[k], (int)qlen[k], (const char *)tbatch + toff[k], (int)tlen[k], pr, 1, &res);
        agatha_steps_stats = 0;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* Textbook model: exact |i-j|<=w band, same recurrence/tie-break/z-drop, no block quirks.      */
/* Used only to quantify how often the block-granular band matters (SURVEY.md App. B #1,#2).    */
/* ------------------------------------------------------------------------------------------ */
/* (stop_diag: the last cell anti-diagonal the z-drop walk looked at -- Q + R - 2 when it never stopped; what bench.py's
   "effective cells" count up to) */
void agatha_model_exactband_stop(const char *qs, int Q, const char *rs, int R, const oracle_params_t *pr,
                                 oracle_result_t *out, int *stop_diag)
{
    const int a = pr->match, b = pr->mismatch, gapoe = pr->gap_open + pr->gap_extend, ge = pr->gap_extend;
    const int w = pr->band_width, z = pr->z_threshold;
    int qp, rp;
    uint8_t *qc = encode_padded(qs, Q, &qp), *rc = encode_padded(rs, R, &rp);
    int32_t *Hrow = (int32_t *)malloc(sizeof(int32_t) * (size_t)(R + 2) * 2);
    int32_t *Frow = Hrow + (R + 2);
    int nd = Q + R;
    int32_t *dh = (int32_t *)malloc(sizeof(int32_t) * (size_t)nd * 2), *dc = dh + nd;
    for (int d = 0; d < nd; d++) { dh[d] = INT_MIN; dc[d] = 0; }
    /* Hrow[j+1] = H(i-1, j), Hrow[0] = H(i-1,-1); Frow[j+1] = F(i, j) */
    Hrow[0] = 0;
    for (int j = 0; j < R; j++) {
        int k = -(gapoe + ge * j);
        Hrow[j + 1] = j <= w ? k : NEG_INF2; Frow[j + 1] = j <= w ? k - gapoe : NEG_INF2;
    }
    for (int i = 0; i < Q; i++) {
        int k = -(gapoe + ge * i);
        int hleft = i <= w ? k : NEG_INF2, e = i <= w ? k - gapoe : NEG_INF2;
        int diag = Hrow[0];
        Hrow[0] = hleft;
        int jlo = imax(0, i - w), jhi = imin(R - 1, i + w);
        if (jlo > jhi) continue;               /* the band has left the matrix on this row */
        if (jlo > 0) { diag = Hrow[jlo]; }
        for (int j = jlo; j <= jhi; j++) {
            int up = Hrow[j + 1];
            int t = sub_score(qc[i], rc[j], a, b) + diag;
            int h = imax(imax(t, Frow[j + 1]), e);
            Frow[j + 1] = imax(t - gapoe, Frow[j + 1] - ge);
            e = imax(t - gapoe, e - ge);
            diag = up; Hrow[j + 1] = h;
            int d = i + j;
            if (dh[d] == INT_MIN || h > dh[d] || (h == dh[d] && j > dc[d])) { dh[d] = h; dc[d] = j; }
        }
        if (jlo > 0) Hrow[jlo] = NEG_INF2;   /* cell left of the band on the next row */
    }
    zstate_t zs = {0, 0, 0, 0};
    int last = -1;
    for (int d = 0; d < Q + R - 1 && !zs.stopped; d++) {
        int H = dh[d] == INT_MIN ? -32768 : dh[d];
        zdrop_check(&zs, H, dh[d] == INT_MIN ? 0 : dc[d], d, z, ge);
        last = d;
    }
    if (stop_diag) *stop_diag = last;
    out->score = zs.best; out->query_end = zs.best_q; out->target_end = zs.best_t;
    free(Hrow); free(dh); free(qc); free(rc);
}

void agatha_model_exactband(const char *qs, int Q, const char *rs, int R, const oracle_params_t *pr,
                            oracle_result_t *out)
{
    agatha_model_exactband_stop(qs, Q, rs, R, pr, out, NULL);
}

/* ------------------------------------------------------------------------------------------ */
/* Batch drivers over the GASAL host-batch wire format (host_batch.cpp:79-154): raw ASCII,      */
/* sequences at byte offsets (multiples of 8), true lengths.  OpenMP over pairs.                 */
/* ------------------------------------------------------------------------------------------ */
void agatha_oracle_batch(const uint8_t *qbatch, const uint8_t *tbatch,
                         const uint32_t *qoff, const uint32_t *toff,
                         const uint32_t *qlen, const uint32_t *tlen, int n,
                         const oracle_params_t *pr, int wide, int model, int threads,
                         int32_t *score, int32_t *qend, int32_t *tend)
{
    (void)threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 0 ? threads : 1)
#endif
    for (int k = 0; k < n; k++) {
        oracle_result_t res;
        const char *q = (const char *)qbatch + qoff[k], *t = (const char *)tbatch + toff[k];
        if (model == 0) agatha_model_slices(q, (int)qlen[k], t, (int)tlen[k], pr, wide, &res);
        else if (model == 1) agatha_model_steps(q, (int)qlen[k], t, (int)tlen[k], pr, wide, &res);
        else agatha_model_exactband(q, (int)qlen[k], t, (int)tlen[k], pr, &res);
        score[k] = res.score; qend[k] = res.query_end; tend[k] = res.target_end;
    }
}

/* traceback over a host batch: pair k's bytes start at cigar + cigar_off[k], n_ops[k] of them (-1: failed) */
void agatha_traceback_batch(const uint8_t *qbatch, const uint8_t *tbatch,
                            const uint32_t *qoff, const uint32_t *toff,
                            const uint32_t *qlen, const uint32_t *tlen, int n,
                            const oracle_params_t *pr, int threads,
                            int32_t *score, int32_t *qend, int32_t *tend,
                            uint8_t *cigar, const uint64_t *cigar_off, int32_t *n_ops)
{
    (void)threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 0 ? threads : 1)
#endif
    for (int k = 0; k < n; k++) {
        oracle_result_t res;
        const char *q = (const char *)qbatch + qoff[k], *t = (const char *)tbatch + toff[k];
        n_ops[k] = agatha_model_traceback(q, (int)qlen[k], t, (int)tlen[k], pr, &res, cigar + cigar_off[k],
                                          (int)qlen[k] + (int)tlen[k] + 8);
        score[k] = res.score; qend[k] = res.query_end; tend[k] = res.target_end;
    }
}

/* pack_rc_seqs.h:21-33: 8 ASCII bytes -> one uint32, first base in bits 31-28 */
void agatha_oracle_pack(const uint8_t *unpacked, uint32_t nbytes, uint32_t *packed)
{
    for (uint32_t wi = 0; wi < nbytes / 8; wi++) {
        uint32_t v = 0;
        for (int k = 0; k < 8; k++) v |= (uint32_t)(unpacked[8 * wi + k] & 15) << (28 - 4 * k);
        packed[wi] = v;
    }
}

/* in-band cells of a pair on the cell anti-diagonals 0 .. stop_diag (exact band |i - j| <= w): the "effective cells" of
   SURVEY.md 8(d) for a pair whose z-drop walk ended on stop_diag */
int64_t agatha_cells_upto_diag(int Q, int R, int w, int stop_diag)
{
    int64_t n = 0;
    if (Q <= 0 || R <= 0) return 0;
    for (int d = 0; d <= stop_diag && d < Q + R - 1; d++) {
        const int lo = imax(imax(0, d - (R - 1)), (d - w + 1) >> 1), hi = imin(imin(Q - 1, d), (d + w) >> 1);
        if (hi >= lo) n += hi - lo + 1;
    }
    return n;
}

/* nominal in-band cells of a pair, SURVEY.md 8(d) */
int64_t agatha_nominal_cells(int Q, int R, int w)
{
    int64_t n = 0;
    for (int i = 0; i < Q; i++) {
        int lo = imax(0, i - w), hi = imin(R - 1, i + w);
        if (hi >= lo) n += hi - lo + 1;
    }
    return n;
}
