/*
 * agatha_lanes_model.c -- CPU emulation of the SCHEDULE of the HIP alignment kernel
 * (agatha_amd/csrc/align_kernel.hip), lane for lane.
 *
 * TEST INFRASTRUCTURE ONLY (see agatha_oracle.c).  It exists so that the kernel's design --
 * band-stationary column state per (lane, slot), row state handed to the right neighbour one
 * block-anti-diagonal later, relative-column packed maxima with a moving column base, eager
 * z-drop checks -- can be checked against the oracle on the CPU, where there is no GPU.
 *
 * Mapping: column block r lives in slot r % S of lane (r / S) % G of a G-lane group; a slot
 * moves on to column r + G*S once q = i - r has run past the band.  Requires
 * G*S >= min(W + 1, ceil(Q/8), ceil(R/8)), W = (band_width + 7) / 8.
 * Semantics are the oracle's "wide" mode (int32 state), identical to the reference wherever
 * the reference is defined.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>
#include <stdio.h>
#include <math.h>

#define NEG_INF2 (-16384)
#define N_VALUE 14
#define MAXG 64
#define MAXS 8

int32_t *agatha_dbg_dump = 0; int agatha_dbg_stride = 0;   /* debugging aid: H of every in-band cell */
typedef struct { int32_t match, mismatch, gap_open, gap_extend, slice_width, z_threshold, band_width; } lm_params_t;

static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int32_t ssub_sat(int32_t a, int32_t b)
{
    int64_t r = (int64_t)a - b;
    return r < INT_MIN ? INT_MIN : (r > INT_MAX ? INT_MAX : (int32_t)r);
}

typedef struct {
    int rcur[MAXS];
    int32_t h[MAXS][8], f[MAXS][8], corner[MAXS];
    uint32_t rword[MAXS];
    /* hand-off slots X[0..S]: X[s] = row input of slot s, X[S] = output of slot S-1 */
    int32_t xh[MAXS + 1][8], xe[MAXS + 1][8];
    int xr[MAXS + 1];
    int32_t A[15];
    int32_t AV[2][15];      /* the int16 kernel's maxima of H alone, per half (even / odd slots) of the lane: this step's own cells */
    int32_t CAR[2][7];      /* ... and what the previous step's blocks left on this step's anti-diagonals 0..6 (carried) */
    int32_t bhi[MAXS], blo[MAXS];   /* value steps: largest / smallest-over-anti-diagonals boundary value of the slot's block (see below) */
} lane_t;

static void init_col(lane_t *ln, int s, int r, int R, int prl, int w, int gapoe, int ge, const uint32_t *pt)
{
    for (int m = 0; m < 8; m++) {
        int c = 8 * r + m;
        if (c < R && c <= w) { ln->h[s][m] = -(gapoe + ge * c); ln->f[s][m] = ln->h[s][m] - gapoe; }
        else { ln->h[s][m] = NEG_INF2; ln->f[s][m] = NEG_INF2; }
    }
    ln->corner[s] = (r == 0) ? 0 : ((8 * r - 1) <= w ? -(gapoe + ge * (8 * r - 1)) : NEG_INF2);
    ln->rword[s] = (r < prl) ? pt[r] : 0xEEEEEEEEu;
    ln->rcur[s] = r;
}

/* packed words: base k of a word in bits 31-4k..28-4k (pack_rc_seqs.h:21-33) */
static void pack_words(const char *s, int len, uint32_t *out, int nwords)
{
    for (int wv = 0; wv < nwords; wv++) {
        uint32_t v = 0;
        for (int k = 0; k < 8; k++) {
            int idx = 8 * wv + k;
            uint32_t code = idx < len ? (uint32_t)(s[idx] & 15) : N_VALUE;
            v |= code << (28 - 4 * k);
        }
        out[wv] = v;
    }
}

int agatha_model_lanes(const char *qs, int Q, const char *rs, int R, const lm_params_t *pr,
                       int G, int S, int32_t *out3)
{
    const int a = pr->match, b = pr->mismatch, gapoe = pr->gap_open + pr->gap_extend, ge = pr->gap_extend;
    const int sw = pr->slice_width, z = pr->z_threshold, w = pr->band_width;
    const int W = (w + 7) / 8, GS = G * S;
    if (G > MAXG || S > MAXS) return -1;
    int K = 0; while ((1 << K) < 8 * (GS + 2)) K++;
    const int32_t KMASK = (1 << K) - 1;
    const int pql = (Q + 7) / 8, prl = (R + 7) / 8, total = prl + pql - 1, lim = Q + R - 1;
    if (GS < imin(W + 1, imin(pql, prl))) return -1;     /* at most min(W+1, pql, prl) blocks per anti-diagonal */
    uint32_t *pq = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(pql + prl + 2)), *pt = pq + pql + 1;
    pack_words(qs, Q, pq, pql); pack_words(rs, R, pt, prl);

    lane_t *L = (lane_t *)calloc((size_t)G, sizeof(lane_t));
    for (int k = 0; k < G; k++) {
        for (int s = 0; s < S; s++) init_col(&L[k], s, k * S + s, R, prl, w, gapoe, ge, pt);
        for (int s = 0; s <= S; s++) L[k].xr[s] = -2;
        for (int x = 0; x < 15; x++) L[k].A[x] = INT_MIN;
    }
    int best = 0, best_t = 0, best_q = 0, stopped = 0;
    int i = 0, y = 0, final = 0, cb_prev = 0;
    int ss = 0, se = imin(imin(prl - 1, sw - 1), ((sw - 1) * 8 + 7 + w) / 2 / 8);   /* slice 0 */
    if (Q <= 0 || R <= 0) { out3[0] = out3[1] = out3[2] = 0; free(pq); free(L); return 0; }

    for (;;) {
        /* column base of the packed maxima: one block left of the lowest active column block */
        const int cb = 8 * imax(0, imax(i - pql + 1, (i - W + 1) >> 1) - 1);
        /* rebase the carried maxima (anti-diagonals 8i..8i+6 already hold step i-1's dl 8..14) */
        for (int k = 0; k < G; k++) {
            for (int x = 0; x < 7; x++) L[k].A[x] = ssub_sat(L[k].A[x], cb - cb_prev);
        }
        /* ---- blocks of block-anti-diagonal i ---- */
        for (int k = 0; k < G; k++) {
            lane_t *ln = &L[k];
            for (int s = S - 1; s >= 0; s--) {
                const int r = ln->rcur[s], q = i - r;
                const int cs = imax(0, r - W), ce = imin(pql - 1, r + W);
                const int active = !final && r < prl && q >= cs && q <= ce && r >= ss && r <= se;
                if (!active) { ln->xr[s + 1] = -2; continue; }
                if (y == 0)   /* pass start: padded ref columns fall back to -inf (agatha_kernel.h:207-215) */
                    for (int m = 0; m < 8; m++) if (8 * r + m >= R) { ln->h[s][m] = NEG_INF2; ln->f[s][m] = NEG_INF2; }
                int32_t xh[8], xe[8];
                const int left_ok = (ln->xr[s] == r - 1);
                for (int il = 0; il < 8; il++) {
                    int row = 8 * q + il;
                    if (left_ok) { xh[il] = ln->xh[s][il]; xe[il] = ln->xe[s][il]; }
                    else if (row <= w) { xh[il] = -(gapoe + ge * row); xe[il] = xh[il] - gapoe; }
                    else { xh[il] = NEG_INF2; xe[il] = NEG_INF2; }
                }
                const uint32_t qword = pq[q], rword = ln->rword[s];
                const int nrows = imin(8, Q - 8 * q);
                const int boundary = (q == cs || q == ce);
                const int tu = boundary ? w + 8 * q - 8 * r : 1000;
                const int tl = boundary ? w - 8 * q + 8 * r : 1000;
                const int crel0 = 8 * r - cb;
                int32_t *h = ln->h[s], *f = ln->f[s];
                int32_t oh[8], oe[8];
                for (int il = 0; il < 8; il++) {
                    oh[il] = 0; oe[il] = 0;
                    if (il >= nrows) continue;
                    const int qb = (qword >> (28 - 4 * il)) & 15;
                    int32_t t[8];
                    for (int jl = 0; jl < 8; jl++) {
                        const int rb = (rword >> (28 - 4 * jl)) & 15;
                        int sc = (qb == rb) ? a : -b;
                        if (qb == N_VALUE || rb == N_VALUE) sc = -1;
                        const int32_t d = jl == 0 ? (il == 0 ? ln->corner[s] : xh[il - 1]) : h[jl - 1];
                        t[jl] = sc + d;
                    }
                    int32_t e = xe[il];
                    for (int jl = 0; jl < 8; jl++) {
                        if ((jl - il) <= tu && (il - jl) <= tl) {
                            const int32_t hn = imax(imax(t[jl], f[jl]), e);
                            const int32_t tg = t[jl] - gapoe;
                            f[jl] = imax(tg, f[jl] - ge);
                            e = imax(tg, e - ge);
                            h[jl] = hn;
                            if (agatha_dbg_dump) agatha_dbg_dump[(size_t)(8 * q + il) * agatha_dbg_stride + 8 * r + jl] = hn;
                            const int32_t key = (int32_t)((uint32_t)hn << K) + crel0 + jl;
                            ln->A[il + jl] = imax(ln->A[il + jl], key);
                        }
                    }
                    oh[il] = h[7]; oe[il] = e;
                }
                ln->corner[s] = xh[nrows - 1];
                memcpy(ln->xh[s + 1], oh, sizeof(oh)); memcpy(ln->xe[s + 1], oe, sizeof(oe));
                ln->xr[s + 1] = r;
            }
        }
        /* ---- X[S] of lane k-1 becomes X[0] of lane k (rotate within the group) ---- */
        {
            int32_t th[MAXG][8], te[MAXG][8]; int tr[MAXG];
            for (int k = 0; k < G; k++) { memcpy(th[k], L[k].xh[S], sizeof(th[k])); memcpy(te[k], L[k].xe[S], sizeof(te[k])); tr[k] = L[k].xr[S]; }
            for (int k = 0; k < G; k++) {
                int src = (k + G - 1) % G;
                memcpy(L[k].xh[0], th[src], sizeof(th[src])); memcpy(L[k].xe[0], te[src], sizeof(te[src])); L[k].xr[0] = tr[src];
            }
        }
        /* ---- anti-diagonals 8i..8i+7 are complete: reduce over the group, z-drop checks ---- */
        for (int x = 0; x < 8 && !stopped; x++) {
            int32_t v = INT_MIN;
            for (int k = 0; k < G; k++) v = imax(v, L[k].A[x]);
            const int d = 8 * i + x;
            if (!final && d >= lim) continue;
            int H, c;
            if (v == INT_MIN) { H = -32768; c = 0; } else { H = v >> K; c = (v & KMASK) + cb; }
            if (H > best) { best = H; best_t = c; best_q = d - c; }
            else if (c >= best_t && (d - c) >= best_q) {
                int tlen = c - best_t, qlen = (d - c) - best_q;
                int l = tlen > qlen ? tlen - qlen : qlen - tlen;
                if (z >= 0 && best - H > z + l * ge) stopped = 1;
            }
        }
        if (stopped || final) break;
        /* carry dl 8..14 into the next step, clear the rest */
        for (int k = 0; k < G; k++) {
            for (int x = 0; x < 7; x++) L[k].A[x] = L[k].A[8 + x];
            for (int x = 7; x < 15; x++) L[k].A[x] = INT_MIN;
        }
        cb_prev = cb;
        /* slots whose column has left the band move on to column r + G*S */
        for (int k = 0; k < G; k++)
            for (int s = 0; s < S; s++) {
                int r = L[k].rcur[s];
                if (i + 1 - r > imin(pql - 1, r + W)) init_col(&L[k], s, r + GS, R, prl, w, gapoe, ge, pt);
            }
        /* next step / next slice (agatha_kernel.h:183-191, 330-334) */
        i++; y++;
        if (y == sw) {
            y = 0;
            if (i >= total) final = 1;
            else {
                ss = imax(imax(0, i - pql + 1), (i * 8 + 8 - w) / 2 / 8);
                se = imin(imin(prl - 1, i + sw - 1), ((i + sw - 1) * 8 + 7 + w) / 2 / 8);
                if (ss > se) break;          /* empty slice: stop without checking it (:189-191) */
            }
        }
    }
    out3[0] = best; out3[1] = best_q; out3[2] = best_t;
    free(pq); free(L);
    return 0;
}


/*
 * ---------------------------------------------------------------------------------------------------
 * agatha_model_lanes16 -- emulation of the PACKED-INT16 kernel (agatha_amd/csrc/align16_kernel.hip).
 *
 * Same schedule as above; the arithmetic differs in three ways, all of which this model checks against the
 * int32 model on the CPU before the kernel relies on them:
 *   (1) all DP state is an int16 REPRESENTATION in a drifting frame: rep = value + ge * (row + column) - base, every
 *       quantity of a cell seen from the cell's own anti-diagonal.  E and F then carry over unchanged where the value
 *       loses the gap-extension score (E' = max(t - gap_open, E)), the diagonal step adds a constant 2 ge to the score,
 *       and the boundary values of the first band width are constants.  `base` is raised by L16_DELTA whenever the
 *       representation of an anti-diagonal maximum exceeds L16_REBASE, so sequence length does not limit the domain;
 *       values are recovered (+ base - ge * d) where maxima of different anti-diagonals meet (z-drop, running
 *       maximum).  Three disjoint zones: in-band values live in [L16_LO, ~L16_REBASE + 50 per anti-diagonal of a step];
 *       the reference's -infinity and what is derived from it in [L16_GLO, L16_LO); cells outside the band below L16_GLO.
 *   (2) no per-cell band test.  Every cell of an active block is computed.  In a boundary block the band is cut by
 *       replacing E / F with L16_OUT on one cell diagonal: E leaving the band to the right (cells with jl - il == tu)
 *       and F leaving it downwards (cells with il - jl == tl); state entering an
 *       out-of-band cell from a neighbouring block is replaced by L16_OUT.  Out-of-band cells therefore only ever
 *       hold values below L16_GLO, which lose every max against an in-band value, so in-band cells are unchanged;
 *       an anti-diagonal whose maximum is below L16_GLO has no in-band cell and is reported empty, as the
 *       reference does.
 *       The rows of a lower boundary block whose last cell is outside the band hand on the same STALE values as
 *       the reference (agatha_kernel.h:33 leaves its registers untouched).  Rows past the end of the query are
 *       computed but kept out of the anti-diagonal maxima.
 *   (3) the pair is abandoned (return 1 = "bail": the int32 kernel takes it) when an anti-diagonal maximum comes
 *       within `spread` + L16_DELTA of L16_LO (an in-band cell could then drop out of its zone), or within `spread`
 *       of the reference's -infinity in absolute terms, or is itself derived from -infinity (their exact values
 *       would start to matter).
 * stats[0] = min rep seen, stats[1] = max rep seen, stats[2] = largest garbage rep, stats[3] = smallest in-band rep.
 * ---------------------------------------------------------------------------------------------------
 */
#define L16_LO     (-13000)    /* in-band values are >= L16_LO (enforced by the bail-out rule)                     */
#define L16_NEG    (-17500)    /* the reference's -infinity; values derived from it stay in [L16_GLO, L16_LO)     */
#define L16_GLO    (-22000)
#define L16_OUT    (-28000)    /* state entering an out-of-band cell, and E / F where they leave the band         */
#define L16_REBASE 2048        /* + lift: rebase when the representation of an anti-diagonal maximum exceeds this    */
#define L16_DELTA  2048
/* In-band cells can lie up to `spread` below the maximum of their anti-diagonal.  Up to 7000 that fits between L16_LO and
 * a representation that starts at 0; for steeper scores / wider bands the whole in-band zone is lifted (the pair starts
 * with base = -lift), which the unused range above L16_REBASE has room for.  Beyond L16_MAX_SPREAD a pair would be
 * abandoned at once: its in-band cells reach down to the reference's -infinity from the first anti-diagonal on. */
#define L16_FREE_SPREAD 7000
#define L16_MAX_SPREAD  16000

int agatha_lanes16_spread(const lm_params_t *pr)
{
    const int gapoe = pr->gap_open + pr->gap_extend, ge = pr->gap_extend;
    int per = 2 * ge; if (pr->mismatch > per) per = pr->mismatch; if (per < 1) per = 1;
    return gapoe + per * (pr->band_width + 16) + 64;
}

/* 1 = these scores may run on the int16 kernel (a property of the launch, not of the pair) */
int agatha_lanes16_eligible(const lm_params_t *pr)
{
    if (pr->band_width < 16) return 0;
    if (pr->match < 0 || pr->match > 16 || pr->mismatch < 0 || pr->mismatch > 32) return 0;
    if (pr->gap_open < 0 || pr->gap_open > 64 || pr->gap_extend < 0 || pr->gap_extend > 16) return 0;
    if (agatha_lanes16_spread(pr) > L16_MAX_SPREAD) return 0;
    return 1;
}

static inline int32_t rep16(int64_t v) { return v < L16_LO ? L16_NEG : (int32_t)v; }

static void init_col16(lane_t *ln, int s, int r, int R, int prl, int w, int gapoe, int ge, int base, const uint32_t *pt)
{
    for (int m = 0; m < 8; m++) {
        int c = 8 * r + m;
        /* H(-1, c) = -(gapoe + ge c) in the frame of anti-diagonal c - 1, F(0, c) = that - gapoe in frame c: constants */
        if (c < R && c <= w) { ln->h[s][m] = rep16(-(gapoe + ge) - base); ln->f[s][m] = rep16(-2 * gapoe - base); }
        else { ln->h[s][m] = L16_NEG; ln->f[s][m] = L16_NEG; }
    }
    ln->corner[s] = (r == 0) ? rep16(-2 * ge - base) : ((8 * r - 1) <= w ? rep16(-(gapoe + ge) - base) : L16_NEG);
    ln->rword[s] = (r < prl) ? pt[r] : 0xEEEEEEEEu;
    ln->rcur[s] = r;
}

/* Value steps (align16_body.inc, FAST), round 4: with agatha_lanes16_margin > 0 every step from the pair's second one to the
 * start of its window of key steps computes NO anti-diagonal maxima inside the block.  It only looks, behind each block, at the
 * 15 cells of the block's last row and last column -- state the kernel holds in registers anyway (the column state H of row 7, the
 * row hand-off of column 7) -- which lie on the block's cell anti-diagonals 7..14:
 *   HI = the largest of them over the group (rows behind a query's end included) bounds EVERY cell of the step's blocks from
 *        above once `slack` = 7 max(mismatch, 1) is added: H(i+1, j+1) >= H(i, j) + s, so a cell is at most 7 diagonal moves
 *        below a cell of the last row or column; taken in the frame of anti-diagonal 8i + 7 (the earliest) it can only be too large.
 *   LO = the largest, over lanes and slots, of a block's SMALLEST anti-diagonal value (the larger of its one or two valid cells
 *        on each of 7..14): a lower bound of the maximum of every anti-diagonal 8i + 7 .. 8i + 14 (the block that holds the best
 *        cells has cells on all eight); taken in the frame of 8i + 14 (the latest) it can only be too small.
 * `best` becomes an UPPER BOUND of the running maximum on a value step that may have raised it (pos_known = 0: neither its cell nor its
 * exact value is known); it is exact again as soon as a key step sees a cell above it (everything before that cell is <= the bound).
 * A value step is calm -- may decide "nothing but the running maximum moves" -- when LO of this step AND of the previous one
 * (the anti-diagonals 8i .. 8i + 6 have their boundary cells in the previous step's blocks) are within z of the bound, well inside
 * their zone, inside the pair.  A step that is not calm, a key step that needs cells it does not know, or an end without the cell
 * of the maximum returns 2: the caller runs the pair again on key steps only -- what the kernel does by starting the pair over.
 * Where the two forms meet: the first key step behind value steps ("stale") knows of its anti-diagonals 0..6 only the lower bound LO
 * of the step before; the first value step behind key steps takes lower and upper bound of those anti-diagonals from the carried
 * keys.  Steps i <= 0 are key steps (the anti-diagonals 0..6 of a pair have no boundary cell before them).
 * The model keeps exact keys all along; what it checks is the DECISION logic: whenever it returns 0 with a margin, the result
 * must be the oracle's. */
__thread int agatha_lanes16_margin = 0;
int agatha_lanes16_win_cap_min = 128, agatha_lanes16_win_cap_div = 16;     /* the kernel's debug options of the same names */
/* Checkpoints (align16_checkpoint.inc / align16_acquire.inc, the shapes without bookkeeping): with agatha_lanes16_ck_span > 0 the model keeps
 * the state of every span-th step in two slots, and a pair that gives up goes back to the NEWER one when the bound of its maximum has since
 * risen by more than a bound can lie above a cell (slack + 14 ge), to the older one otherwise, and runs key steps only from there; if it gives
 * up again (or has no checkpoint) it starts from its first step.  Return codes of agatha_model_lanes16: 0 = as it came, 1 = bailed out, 2 =
 * started over from its first step, 3 = went back to a checkpoint, 4 = went back to a checkpoint, gave up again, started over.
 * agatha_lanes16_ck_counts: [0] pairs that went back to the newer checkpoint, [1] to the older one (tests). */
/* (flat batches, align16_body.inc widen_window: a pair says once, at its look between its 64th and 128th step, whether it would need more than
 * four times the window it may have; the kernel runs a batch on key steps when 30 % of its pairs say so.  The model counts; tools/cliff_sweep.py) */
int agatha_lanes16_asked = 0, agatha_lanes16_flat = 0;
int agatha_lanes16_ck_span = 0;
int agatha_lanes16_ck_counts[2] = {0, 0};
/* Probation (round 5, align16_acquire.inc / align16_step_maxima.inc, PROB): with agatha_lanes16_probation = 1 a pair that goes back to a checkpoint
 * runs key steps from there until 32 steps behind the step it gave up on -- on while z-drop is not out of reach by more than the bounds of
 * a value step can blur --, then value steps with the window it started with; a pair that gives up ON probation starts from its first step,
 * on key steps for good.  agatha_lanes16_left_probation counts the pairs' returns to value steps, agatha_lanes16_steps the value / key steps
 * of everything the model ran (a pair that is run again counts twice: that is what it costs). */
int agatha_lanes16_probation = 0, agatha_lanes16_left_probation = 0;
/* A what-if, not the kernel (tools/cliff_sweep.py --bursts --slots): a RING of agatha_lanes16_ck_slots > 2 checkpoints instead of
 * the kernel's two.  A pair that gives up goes back to the NEWEST one since which the bound of its maximum has risen by more than slack + 14 ge
 * (it lies before the last rise: the kernel's rule for the newer of its two), to the oldest one it has otherwise. */
int agatha_lanes16_ck_slots = 2;
long long agatha_lanes16_steps[2] = {0, 0};
/* (tools/cliff_sweep.py --bursts: the same two counts per pair, written by agatha_lanes16_batch when the pointer is set -- 2 n entries) */
long long *agatha_lanes16_pair_steps = 0;
static __thread long long l16_tl_steps[2];
int agatha_lanes16_trace = 0;             /* tools: print every step's mode, bound and whether the cell of the maximum is known (stderr) */
int agatha_lanes16_old_window = 0;        /* tools: round 4's rule for the window of key steps (3/2 (slack + 7 ge) i / best) */
int agatha_lanes16_lazy_max = 8;          /* lazy value steps (round 6, one pair per wave: G >= 64): a passed test answers for at most this many steps behind it (0 = every step is tested) */
long long agatha_lanes16_lazy_steps = 0;  /* tests: value steps on which nothing was tested (all threads; not atomic -- a count that is zero or not) */
int agatha_lanes16_lazy_any_shape = 0;    /* tests / tools: the rule on every shape (its arithmetic does not depend on the shape; the kernel uses it where a wave holds one pair) */

int agatha_model_lanes16(const char *qs, int Q, const char *rs, int R, const lm_params_t *pr,
                         int G, int S, int32_t *out3, int32_t *stats)
{
    const int margin = agatha_lanes16_margin;
    int pos_known = 1, prev_fast = 0, again = 0;
    int keys_only = 0, rolled = 0;      /* (checkpoints) the pair went back to one and runs key steps for good */
    int prob = 0, prob_until = 0;        /* (probation) ... or until step prob_until, see agatha_lanes16_probation */
    long long n_value = 0, n_key = 0;
    const int a = pr->match, b = pr->mismatch, gapoe = pr->gap_open + pr->gap_extend, ge = pr->gap_extend;
    const int gapo = pr->gap_open;
    const int sw = pr->slice_width, z = pr->z_threshold, w = pr->band_width;
    const int W = (w + 7) / 8, GS = G * S;
    if (G > MAXG || S > MAXS) return -1;
    int K = 0; while ((1 << K) < 8 * (GS + 2)) K++;
    const int32_t KMASK = (1 << K) - 1;
    const int pql = (Q + 7) / 8, prl = (R + 7) / 8, total = prl + pql - 1, lim = Q + R - 1;
    if (GS < imin(W + 1, imin(pql, prl))) return -1;
    if (Q <= 0 || R <= 0) { out3[0] = out3[1] = out3[2] = 0; return 0; }
    if (!agatha_lanes16_eligible(pr)) return 1;
    const int spread = agatha_lanes16_spread(pr);
    uint32_t *pq = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(pql + prl + 2)), *pt = pq + pql + 1;
    pack_words(qs, Q, pq, pql); pack_words(rs, R, pt, prl);

    lane_t *L = (lane_t *)calloc((size_t)G, sizeof(lane_t));
    const int lift = spread > L16_FREE_SPREAD ? spread - L16_FREE_SPREAD : 0;
    int base = -lift;
    for (int k = 0; k < G; k++) {
        for (int s = 0; s < S; s++) init_col16(&L[k], s, k * S + s, R, prl, w, gapoe, ge, base, pt);
        for (int s = 0; s <= S; s++) L[k].xr[s] = -2;
        for (int x = 0; x < 15; x++) { L[k].A[x] = INT_MIN; L[k].AV[0][x] = INT_MIN; L[k].AV[1][x] = INT_MIN; }
        for (int x = 0; x < 7; x++) { L[k].CAR[0][x] = INT_MIN; L[k].CAR[1][x] = INT_MIN; }
    }
    /* (round 6: + 8 match where the one-cell / last-column bounds of boundary_bounds<true> may be used -- the upper bound then comes from the blocks'
     *  LAST COLUMNS alone, and a cell that leaves its block downwards is known only as "at most eight diagonal moves above a cell of an earlier step") */
    const int one_cell_ok = margin > 0 && (z < 0 || 40 * gapo <= z);
    const int slack = 7 * imax(b, 1) + (one_cell_ok ? 8 * imax(a, 0) : 0);
    /* first step of the window of key steps: what a read with 15 % errors needs at this scoring (capi.cpp: win_prior), capped */
    const int win_anchor = 2 * (pql < prl ? pql : prl) - 1 - margin - ((pql + prl) >> 7);
    int ewin = win_anchor;
    if (!agatha_lanes16_old_window) {
        const double pen_ = a + b > 0.5 * a + gapoe ? a + b : 0.5 * a + gapoe, mu_ = 4.0 * (a - 0.15 * pen_), V_ = 4.0 * 0.15 * pen_ * pen_;
        double n_ = 4096.0;
        if (mu_ > 0.0) { const double r_ = (sqrt(12.0 * V_) + sqrt(12.0 * V_ + 4.0 * mu_ * (slack + 7 * ge))) / (2.0 * mu_); n_ = r_ * r_ < 4096.0 ? r_ * r_ : 4096.0; }
        ewin = win_anchor - imin((int)ceil(n_), imax(agatha_lanes16_win_cap_min, (pql + prl) / imax(agatha_lanes16_win_cap_div, 1)));
    }
    const int ewin0 = ewin;             /* (the window a pair starts with: first_window of align16_body.inc) */
    int64_t lo_prev_abs = INT_MIN;      /* value steps: lower bound (absolute score) of the maxima of this step's anti-diagonals 0..6 */
    /* Lazy value steps (round 6, align16_body.inc: the shapes with one pair per wave, G >= 64): a test that passed with room to spare answers for
     * the next steps as well.  From the cell the lower bound was read off, every later anti-diagonal holds a cell at most one gap lower (gap_open once,
     * 8 ge per step), and the running maximum rises by at most 8 m per step; so after a test at step j with need = (bound of the running maximum) -
     * (lower bound), the steps j + 1 .. j + k cannot z-drop while need + k (8 m + 8 ge) <= z.  On those steps the kernel computes no lower bound,
     * reduces nothing and tests nothing: it keeps each lane's largest last-column cell (acc_ub here: its absolute bound) for the next test and lets
     * the lower bound sink by 8 ge.  At most agatha_lanes16_lazy_max steps, never next to the window of key steps, a checkpoint or the pair's last
     * rows and columns. */
    int skip_until = 0;
    int64_t acc_ub = INT_MIN;
    int best = 0, best_t = 0, best_q = 0, stopped = 0, bail = 0;
    int i = 0, y = 0, final = 0, cb_prev = 0;
    int ss = 0, se = imin(imin(prl - 1, sw - 1), ((sw - 1) * 8 + 7 + w) / 2 / 8);
    int32_t vmin = 0, vmax = 0, gmax = INT_MIN, rmin = INT_MAX;
#define REB(v) imax((v) - L16_DELTA, imin((v), L16_LO - 1))   /* only in-band values follow the base (v_pk_max / v_pk_min form of the kernel) */
#define TRACK(v) do { if ((v) < vmin) vmin = (v); if ((v) > vmax) vmax = (v); } while (0)

    /* (checkpoints) two slots: the state before step c, c a multiple of the span */
    typedef struct { int valid, i, y, final, cb_prev, ss, se, base, best, best_t, best_q, pos_known, prev_fast, ewin; int64_t lo_prev_abs; lane_t *L; } snap_t;
#define L16_MAX_SLOTS 16
    snap_t snap[L16_MAX_SLOTS];
    for (int sl = 0; sl < L16_MAX_SLOTS; sl++) { snap[sl].valid = 0; snap[sl].L = 0; }
    const int nslots = agatha_lanes16_ck_slots > 2 ? imin(agatha_lanes16_ck_slots, L16_MAX_SLOTS) : 2;
    const int ck_span = margin > 0 ? agatha_lanes16_ck_span : 0;
#define SNAP_SAVE(sn) do { if (!(sn).L) (sn).L = (lane_t *)malloc(sizeof(lane_t) * (size_t)G); memcpy((sn).L, L, sizeof(lane_t) * (size_t)G); (sn).valid = 1; \
        (sn).i = i; (sn).y = y; (sn).final = final; (sn).cb_prev = cb_prev; (sn).ss = ss; (sn).se = se; (sn).base = base; (sn).best = best; (sn).best_t = best_t; \
        (sn).best_q = best_q; (sn).pos_known = pos_known; (sn).prev_fast = prev_fast; (sn).ewin = ewin; (sn).lo_prev_abs = lo_prev_abs; } while (0)
    /* a pair gives up at step i: the checkpoint it goes back to (the kernel's rule), or -1 */
#define SNAP_PICK(out) do { (out) = -1; if (ck_span > 0 && !keys_only && !prob && nslots == 2) { const int c0 = (i / ck_span) * ck_span - ck_span, cN = c0 + ck_span; \
        const snap_t *sN = &snap[(cN / ck_span) & 1], *sO = &snap[(c0 / ck_span) & 1]; \
        if (cN >= ck_span && cN < i && sN->valid && sN->i == cN && best - sN->best > slack + 14 * ge) (out) = (cN / ck_span) & 1; \
        else if (c0 >= ck_span && sO->valid && sO->i == c0) (out) = (c0 / ck_span) & 1; } \
      else if (ck_span > 0 && !keys_only && !prob) { /* (the ring: newest first) */ \
        for (int kk_ = 0; kk_ < nslots; kk_++) { const int c_ = (i / ck_span) * ck_span - kk_ * ck_span; if (c_ < ck_span) break; if (c_ >= i) continue; \
            const snap_t *s_ = &snap[(c_ / ck_span) % nslots]; if (!s_->valid || s_->i != c_) continue; \
            (out) = (c_ / ck_span) % nslots; if (best - s_->best > slack + 14 * ge) break; } } } while (0)
#define SNAP_LOAD(sn) do { const int gave_up_at = i; skip_until = 0; acc_ub = INT_MIN; memcpy(L, (sn).L, sizeof(lane_t) * (size_t)G); i = (sn).i; y = (sn).y; final = (sn).final; cb_prev = (sn).cb_prev; ss = (sn).ss; se = (sn).se; \
        base = (sn).base; best = (sn).best; best_t = (sn).best_t; best_q = (sn).best_q; pos_known = (sn).pos_known; prev_fast = (sn).prev_fast; ewin = (sn).ewin; \
        lo_prev_abs = (sn).lo_prev_abs; stopped = 0; bail = 0; rolled = 1; \
        if (agatha_lanes16_probation) { prob = 1; prob_until = gave_up_at + 33; } else keys_only = 1; } while (0)
run_again:
    for (;;) {
        if (ck_span > 0 && !keys_only && i >= ck_span && i % ck_span == 0 && (prob || i < ewin)) SNAP_SAVE(snap[(i / ck_span) % nslots]);
        const int cb = 8 * imax(0, imax(i - pql + 1, (i - W + 1) >> 1) - 1);
        int n_in_flight = 0;        /* a block of this step works on a query word that holds an N (or the padding behind the query's end) */
        const int one_cell = one_cell_ok && (i + W + 6 < 2 * imin(pql, prl));      /* (the kernel's wave-uniform guard, here for the one pair; only where a gap open is small against z) */
        for (int k = 0; k < G; k++)
            for (int x = 0; x < 7; x++) L[k].A[x] = ssub_sat(L[k].A[x], cb - cb_prev);
        for (int k = 0; k < G; k++) {
            lane_t *ln = &L[k];
            for (int s = S - 1; s >= 0; s--) {
                const int r = ln->rcur[s], q = i - r;
                const int cs = imax(0, r - W), ce = imin(pql - 1, r + W);
                const int active = !final && r < prl && q >= cs && q <= ce && r >= ss && r <= se;
                ln->bhi[s] = INT_MIN; ln->blo[s] = INT_MIN;
                if (!active) { ln->xr[s + 1] = -2; continue; }
                /* (an N of the query itself: the pair is flagged by exotic_kernel; the N padding behind the query's end does not count) */
                { const int real = imin(8, Q - 8 * q); if (real > 0 && (pq[q] & 0x88888888u & (real >= 8 ? 0xFFFFFFFFu : ~(0xFFFFFFFFu >> (4 * real))))) n_in_flight = 1; }
                if (y == 0)
                    for (int m = 0; m < 8; m++) if (8 * r + m >= R) { ln->h[s][m] = L16_NEG; ln->f[s][m] = L16_NEG; }
                int32_t xh[8], xe[8], xe_in[8];
                const int left_ok = (ln->xr[s] == r - 1);
                for (int il = 0; il < 8; il++) {
                    int row = 8 * q + il;
                    if (left_ok) { xh[il] = ln->xh[s][il]; xe[il] = ln->xe[s][il]; }
                    else if (row <= w) { xh[il] = rep16(-(gapoe + ge) - base); xe[il] = rep16(-2 * gapoe - base); }
                    else { xh[il] = L16_NEG; xe[il] = L16_NEG; }
                }
                memcpy(xe_in, xe, sizeof(xe_in));
                const uint32_t qword = pq[q], rword = ln->rword[s];
                const int nrows = imin(8, Q - 8 * q);
                const int boundary = (q == cs || q == ce);
                const int tu = boundary ? w + 8 * q - 8 * r : 1000;
                const int tl = boundary ? w - 8 * q + 8 * r : 1000;
                const int crel0 = 8 * r - cb;
                int32_t *h = ln->h[s], *f = ln->f[s];
                int32_t oh[8], oe[8];
                int32_t cornerv = ln->corner[s];
                /* entry cuts: state that enters an out-of-band cell of this block from a neighbouring block */
                if (0 > tu || 0 > tl) cornerv = L16_OUT;
                for (int m = 0; m < 8; m++) {
                    const int out_c0 = (-m > tu) || (m > tl);           /* cell (m, 0) is outside the band */
                    const int out_r0 = (m > tu) || (-m > tl);           /* cell (0, m) is outside the band */
                    if (out_c0) { xe[m] = L16_OUT; if (m > 0) xh[m - 1] = L16_OUT; }    /* its E and its diagonal */
                    if (out_r0) { f[m] = L16_OUT; if (m > 0) h[m - 1] = L16_OUT; }      /* its F and its diagonal */
                }
                for (int il = 0; il < 8; il++) {
                    int qb = (qword >> (28 - 4 * il)) & 15;             /* N beyond the query: packing pads with N */
                    /* (rows behind the query's end: the kernel's value steps read the N padding as class 3 = G, class_word();
                     *  nothing valid depends on them, but the upper bound HI of a value step sees them) */
                    if (margin > 0 && il >= nrows && qb == N_VALUE) qb = 7;
                    int32_t t[8];
                    for (int jl = 0; jl < 8; jl++) {
                        const int rb = (rword >> (28 - 4 * jl)) & 15;
                        int sc = (qb == rb) ? a : -b;
                        if (qb == N_VALUE || rb == N_VALUE) sc = -1;
                        const int32_t d = jl == 0 ? (il == 0 ? cornerv : xh[il - 1]) : h[jl - 1];
                        t[jl] = sc + 2 * ge + d; TRACK(t[jl]);           /* two anti-diagonals further: + 2 ge */
                    }
                    int32_t e = xe[il];
                    for (int jl = 0; jl < 8; jl++) {
                        const int out = !((jl - il) <= tu && (il - jl) <= tl);
                        const int32_t hn = imax(imax(t[jl], f[jl]), e);
                        const int32_t u = t[jl] - gapo; TRACK(u);
                        f[jl] = (il - jl) == tl ? L16_OUT : imax(u, f[jl]); TRACK(f[jl]);
                        e = (jl - il) == tu ? L16_OUT : imax(u, e); TRACK(e);
                        h[jl] = hn;
                        if (out) { if (hn > gmax) gmax = hn; }
                        else if (il < nrows) { if (hn < rmin) rmin = hn; }
                        if (agatha_dbg_dump && !out && il < nrows) agatha_dbg_dump[(size_t)(8 * q + il) * agatha_dbg_stride + 8 * r + jl] = hn + base;
                        if (il < nrows) {
                            const int32_t key = (int32_t)((uint32_t)hn << K) + crel0 + jl;
                            ln->A[il + jl] = imax(ln->A[il + jl], key);
                            ln->AV[s & 1][il + jl] = imax(ln->AV[s & 1][il + jl], hn);
                        }
                    }
                    oh[il] = h[7]; oe[il] = e;
                }
                {
                    /* the block's last row (7, x) = h[x] and last column (x, 7) = oh[x], cell anti-diagonal 7 + x, as computed
                     * (before the stale values of a lower edge block go into the hand-off) */
                    int32_t hi = INT_MIN, lo = INT_MAX;
                    for (int x = 0; x < 8; x++) {
                        hi = imax(hi, imax(h[x], oh[x]));
                        int32_t v = INT_MIN;                    /* valid cells only: rows that exist */
                        if (nrows == 8) v = imax(v, h[x]);
                        if (x < nrows) v = imax(v, oh[x]);
                        lo = imin(lo, v);
                    }
                    /* (round 6, boundary_bounds<true> of align16_block.inc: away from the pair's last rows and columns the lower bound comes from the
                     *  block's two cells on its cell anti-diagonal 7 alone -- one gap below them lies a cell of each of the next seven anti-diagonals) */
                    if (one_cell) {
                        lo = INT_MIN; if (nrows == 8) lo = imax(lo, h[0]); if (0 < nrows) lo = imax(lo, oh[0]);
                        /* ... and the upper bound from the block's last column alone: every cell reaches the last column of its column block within
                         * seven diagonal moves, in this row block or in the next one (which the next step measures) */
                        hi = INT_MIN; for (int x = 0; x < 8; x++) hi = imax(hi, oh[x]);
                    }
                    ln->bhi[s] = hi; ln->blo[s] = lo;
                }
                for (int il = 0; il < 8; il++)
                    if (il - 7 > tl) {       /* the same VALUE is handed on for a row il - (tl + 7) anti-diagonals further down */
                        const int32_t sv = oh[imax(0, tl + 7)];
                        oh[il] = sv + ge * (il - imax(0, tl + 7)); oe[il] = xe_in[il];
                    }
                ln->corner[s] = xh[7];
                memcpy(ln->xh[s + 1], oh, sizeof(oh)); memcpy(ln->xe[s + 1], oe, sizeof(oe));
                ln->xr[s + 1] = r;
            }
        }
        {
            int32_t th[MAXG][8], te[MAXG][8]; int tr[MAXG];
            for (int k = 0; k < G; k++) { memcpy(th[k], L[k].xh[S], sizeof(th[k])); memcpy(te[k], L[k].xe[S], sizeof(te[k])); tr[k] = L[k].xr[S]; }
            for (int k = 0; k < G; k++) {
                int src = (k + G - 1) % G;
                memcpy(L[k].xh[0], th[src], sizeof(th[src])); memcpy(L[k].xe[0], te[src], sizeof(te[src])); L[k].xr[0] = tr[src];
            }
        }
        /* ---- value steps: the mode of this step, the calm test ---- */
        /* (the kernel's window of key steps, align16_body.inc `ewin`: from `margin` + 1/128 of the steps before the block anti-diagonal of
         *  the corner the shorter sequence ends in, to the pair's end; and the pair's first step) */
        /* (the bound of a value step lies up to slack + 7 ge above the running maximum, and the key steps must see the score rise by
         *  more than that: every 64 steps the window is widened by the steps that takes at the pair's rate so far, 3/2 of them) */
        /* (round 5: widen_window of align16_body.inc -- the steps a random walk with the pair's rate of rise and the variance its error
         *  rate implies needs to rise by more than slack + 7 ge except with the probability of a 3.5-sigma event) */
        if (margin > 0 && !prob && i > 0 && (i & 63) == 0 && (best > 0 || !agatha_lanes16_old_window) && (agatha_lanes16_old_window || i < ewin)) {
            const int64_t X_ = slack + 7 * ge, pen2 = imax(2 * (a + b), a + 2 * gapoe);
            /* (`best` without a cell is a bound, up to X above the running maximum: the rate is taken from what is certain) */
            const int64_t best_ = agatha_lanes16_old_window || pos_known ? best : best - X_;
            const int64_t err4 = imax(4 * i * a - best_, 0);
            int64_t more = best_ > 0 ? ((6 * err4 * pen2 + 2 * best_ * X_) * i) / (best_ * best_) : (1 << 30);
            if (agatha_lanes16_old_window) {
                more = ((int64_t)3 * X_ * i) / (2 * (int64_t)best);
                ewin = imin(ewin, win_anchor - (int)(more < 4096 ? more : 4096));
            } else {
                /* (what the pair shows of itself may move the window either way while it is still on value steps) */
                const int64_t cap = imax(agatha_lanes16_win_cap_min, (pql + prl) / imax(agatha_lanes16_win_cap_div, 1));
                if (i >= 64 && i < 128) { __sync_fetch_and_add(&agatha_lanes16_asked, 1); if (more > 4 * cap) __sync_fetch_and_add(&agatha_lanes16_flat, 1); }
                ewin = imax(win_anchor - (int)(more < cap ? more : cap), i);
            }
        }
        /* (a wave runs key steps while a pair whose query holds an N has an N row in flight: align16_body.inc, want_keys) */
        const int fast = margin > 0 && i >= 1 && i < ewin && !n_in_flight && !keys_only && !prob;
        if (fast) n_value++; else n_key++;
        /* (a key step behind lazy value steps: what they kept for the next test goes into the bound of the running maximum now) */
        if (!fast && acc_ub != INT_MIN) { if (acc_ub > best) { best = (int)acc_ub; pos_known = 0; } acc_ub = INT_MIN; skip_until = 0; }
        const int lazy = fast && prev_fast && one_cell && (G >= 64 || agatha_lanes16_lazy_any_shape) && agatha_lanes16_lazy_max > 0 && i + 1 < skip_until && i + 2 < ewin &&
                         !(ck_span > 0 && (i + 1) % ck_span == 0);
        int calm = 0, stale = 0;
        int32_t HI = INT_MIN;
        if (margin > 0) {
            stale = !fast && prev_fast;
            if (fast) {
                if (!prev_fast) {
                    /* first value step behind key steps: what the key step's blocks left on this step's anti-diagonals 0..6 */
                    int32_t clo = INT_MIN, chi = INT_MIN;
                    for (int k = 0; k < G; k++)
                        for (int hf = 0; hf < 2; hf++) {
                            int32_t lo = INT_MAX, hi = INT_MIN;
                            for (int x = 0; x < 7; x++) { lo = imin(lo, L[k].CAR[hf][x]); hi = imax(hi, L[k].CAR[hf][x]); }
                            clo = imax(clo, lo); chi = imax(chi, hi);
                        }
                    lo_prev_abs = clo == INT_MIN ? INT_MIN : (int64_t)clo + base - (int64_t)ge * (8 * i + 6);
                    if (chi != INT_MIN && (int64_t)chi + base - (int64_t)ge * (8 * i) > best) { best = (int)((int64_t)chi + base - (int64_t)ge * (8 * i)); pos_known = 0; }
                }
                int32_t LO = INT_MIN;
                for (int k = 0; k < G; k++)
                    for (int sx = 0; sx < S; sx++) { LO = imax(LO, L[k].blo[sx]); HI = imax(HI, L[k].bhi[sx]); }
                if (one_cell && LO != INT_MIN) LO -= gapo;          /* (one gap below the best cell of anti-diagonal 8i + 7, in the drifting frame) */
                const int64_t lo_abs = LO == INT_MIN ? INT_MIN : (int64_t)LO + base - (int64_t)ge * (8 * i + 14);
                int64_t ub = HI == INT_MIN ? INT_MIN : (int64_t)HI + base - (int64_t)ge * (8 * i + 7) + slack;
                if (lazy) {
                    /* nothing is tested: the last test answers for this step */
                    agatha_lanes16_lazy_steps++;
                    if (ub > acc_ub) acc_ub = ub;
                    if (lo_prev_abs != INT_MIN) lo_prev_abs -= 8 * (int64_t)ge;
                    HI = INT_MIN;                                   /* (no rebase on such a step either) */
                    prev_fast = fast;
                    goto step_done;
                }
                if (acc_ub > ub) ub = acc_ub;
                acc_ub = INT_MIN;
                const int64_t nb = ub > best ? ub : best, lo_both = lo_abs < lo_prev_abs ? lo_abs : lo_prev_abs;
                calm = !final && (8 * i + 7 < lim) && LO != INT_MIN && lo_prev_abs != INT_MIN && LO >= L16_LO + spread + L16_DELTA + 7 * ge &&
                       lo_abs >= NEG_INF2 + spread && (z < 0 || nb - lo_both <= z);
                if (!calm) { int pk; SNAP_PICK(pk); if (pk >= 0) { __sync_fetch_and_add(&agatha_lanes16_ck_counts[pk == ((i / ck_span) % nslots) ? 0 : 1], 1); SNAP_LOAD(snap[pk]); continue; } again = 1; break; }
                if (ub > best) { best = (int)ub; pos_known = 0; }
                lo_prev_abs = lo_abs;
                {   /* how many steps this test answers for */
                    const int64_t d_ = 8 * (int64_t)imax(a, 0) + 8 * (int64_t)ge, need_ = (int64_t)best - lo_both;
                    int64_t k_ = z < 0 ? agatha_lanes16_lazy_max : (d_ > 0 ? ((int64_t)z - need_) / d_ : 0);
                    if (k_ > agatha_lanes16_lazy_max) k_ = agatha_lanes16_lazy_max;
                    skip_until = i + 1 + (int)(k_ > 0 ? k_ : 0);
                }
            } else {
                int32_t lo8 = INT_MIN, mk = INT_MIN;
                for (int k = 0; k < G; k++)
                    for (int hf = 0; hf < 2; hf++) {
                        int32_t lo = INT_MAX, hi = INT_MIN;
                        for (int x = 0; x < 8; x++) {
                            int32_t a = L[k].AV[hf][x];
                            if (x < 7) {
                                /* (stale: the value steps carried no maxima, only the lower bound of the step before) */
                                const int32_t c = !stale ? L[k].CAR[hf][x] : (lo_prev_abs == INT_MIN ? INT_MIN : (int32_t)(lo_prev_abs - base + (int64_t)ge * (8 * i + x)));
                                a = imax(a, c);
                            }
                            const int32_t v = a == INT_MIN ? INT_MIN : a + (7 - x) * ge;      /* the frame of the step's last anti-diagonal */
                            if (v < lo) lo = v;
                            if (v > hi) hi = v;
                        }
                        if (lo > lo8) lo8 = lo;
                        if (hi > mk) mk = hi;
                    }
                const int64_t base_i = (int64_t)base - (int64_t)ge * (8 * i + 7);
                calm = !final && (8 * i + 7 < lim) && lo8 != INT_MIN && lo8 >= L16_LO + spread + L16_DELTA + 7 * ge &&
                       lo8 + base_i >= NEG_INF2 + spread && (z < 0 || imax(best, (int)(mk + base_i)) - (int)(lo8 + base_i) <= z);
                if (prob) {
                    /* z-drop out of reach by more than a value step's bounds can blur, and the cell of the maximum known: back to value steps */
                    const int comfy = calm && pos_known && (z < 0 || imax(best, (int)(mk + base_i)) - (int)(lo8 + base_i) + slack + 14 * ge + 32 <= z);
                    if (!comfy) prob_until = imax(prob_until, i + 33);
                    else if (i + 1 >= prob_until) { prob = 0; ewin = ewin0; __sync_fetch_and_add(&agatha_lanes16_left_probation, 1); }
                }
                if (!calm && (stale || !pos_known)) { int pk; SNAP_PICK(pk); if (pk >= 0) { __sync_fetch_and_add(&agatha_lanes16_ck_counts[pk == ((i / ck_span) % nslots) ? 0 : 1], 1); SNAP_LOAD(snap[pk]); continue; } again = 1; break; }
            }
            prev_fast = fast;
        }
step_done:;
        int hi_rep = INT_MIN;
        for (int x = 0; x < 8 && !stopped; x++) {
            int32_t v = INT_MIN;
            for (int k = 0; k < G; k++) v = imax(v, L[k].A[x]);
            const int d = 8 * i + x;
            if (v != INT_MIN && (v >> K) >= L16_LO) hi_rep = imax(hi_rep, v >> K);
            if (fast) { hi_rep = HI; continue; }    /* a value step looks at no anti-diagonal by itself (its rebase follows the boundary cells) */
            if (!final && d >= lim) continue;
            int H, c;
            if (v == INT_MIN || (v >> K) < L16_GLO) { H = -32768; c = 0; }      /* empty, or only out-of-band cells */
            else {
                H = (v >> K) + base - ge * d; c = (v & KMASK) + cb;
                /* (round 4) only cells that derive from -infinity are left on this anti-diagonal (padded columns behind a short target's
                 * end, rows behind the band's last block): no real anti-diagonal maximum falls from the in-band zone to below L16_LO in
                 * one anti-diagonal, and nothing real follows one that has: the result is final, the pair ends here (align16_body.inc) */
                if ((v >> K) < L16_LO && pos_known && !stale &&
                    imax(imax(0, d - (R - 1)), (d - w + 1) >> 1) > imin(imin(Q - 1, d), (d + w) >> 1)) { stopped = 1; H = -32768; c = 0; }      /* (... and the geometry agrees: no in-band cell of the pair is left on d) */
                else
                if ((v >> K) < L16_LO + spread + L16_DELTA || H < NEG_INF2 + spread) { bail = 1; break; }
            }
            if (H > best) { best = H; best_t = c; best_q = d - c; pos_known = !stale; }
            else if (c >= best_t && (d - c) >= best_q) {
                int tlen = c - best_t, qlen = (d - c) - best_q;
                int l = tlen > qlen ? tlen - qlen : qlen - tlen;
                if (z >= 0 && best - H > z + l * ge) stopped = 1;
            }
        }
        if (agatha_lanes16_trace && i >= agatha_lanes16_trace) fprintf(stderr, "step %d of %d fast %d stale %d calm %d best %d pos_known %d ewin %d HI %d base %d\n", i, total, fast, stale, calm, best, pos_known, ewin, HI, base);
        if (bail || stopped || final) break;
        for (int k = 0; k < G; k++) {
            for (int x = 0; x < 7; x++) { L[k].A[x] = L[k].A[8 + x]; L[k].CAR[0][x] = L[k].AV[0][8 + x]; L[k].CAR[1][x] = L[k].AV[1][8 + x]; }
            for (int x = 7; x < 15; x++) L[k].A[x] = INT_MIN;
            for (int x = 0; x < 15; x++) { L[k].AV[0][x] = INT_MIN; L[k].AV[1][x] = INT_MIN; }
        }
        cb_prev = cb;
        /* rebase: keep the representation of the running maximum small */
        if (hi_rep > L16_REBASE + lift) {
            base += L16_DELTA;
            for (int k = 0; k < G; k++) {
                lane_t *ln = &L[k];
                for (int s = 0; s < S; s++) {
                    for (int m = 0; m < 8; m++) { ln->h[s][m] = REB(ln->h[s][m]); ln->f[s][m] = REB(ln->f[s][m]); }
                    ln->corner[s] = REB(ln->corner[s]);
                }
                for (int s = 0; s <= S; s++)
                    for (int m = 0; m < 8; m++) { ln->xh[s][m] = REB(ln->xh[s][m]); ln->xe[s][m] = REB(ln->xe[s][m]); }
                for (int x = 0; x < 7; x++) if (ln->A[x] != INT_MIN) ln->A[x] = imax(ln->A[x] - (L16_DELTA << K), imin(ln->A[x], ((L16_LO + 32768) << K) - 1 - (32768 << K)));
                for (int hf = 0; hf < 2; hf++)
                    for (int x = 0; x < 7; x++) if (ln->CAR[hf][x] != INT_MIN) ln->CAR[hf][x] = REB(ln->CAR[hf][x]);
            }
        }
        for (int k = 0; k < G; k++)
            for (int s = 0; s < S; s++) {
                int r = L[k].rcur[s];
                if (i + 1 - r > imin(pql - 1, r + W)) init_col16(&L[k], s, r + GS, R, prl, w, gapoe, ge, base, pt);
            }
        i++; y++;
        if (y == sw) {
            y = 0;
            if (i >= total) final = 1;
            else {
                ss = imax(imax(0, i - pql + 1), (i * 8 + 8 - w) / 2 / 8);
                se = imin(imin(prl - 1, i + sw - 1), ((i + sw - 1) * 8 + 7 + w) / 2 / 8);
                if (ss > se) break;
            }
        }
    }
#undef TRACK
#undef REB
    if (margin > 0 && !bail && !pos_known && !again) {          /* the pair ends without the cell of its maximum */
        int pk; SNAP_PICK(pk);
        if (pk >= 0) { __sync_fetch_and_add(&agatha_lanes16_ck_counts[pk == ((i / ck_span) % nslots) ? 0 : 1], 1); SNAP_LOAD(snap[pk]); goto run_again; }
        again = 1;
    }
    __sync_fetch_and_add(&agatha_lanes16_steps[0], n_value); __sync_fetch_and_add(&agatha_lanes16_steps[1], n_key);
    l16_tl_steps[0] += n_value; l16_tl_steps[1] += n_key;
    out3[0] = best; out3[1] = best_q; out3[2] = best_t;
    if (stats) { stats[0] = vmin; stats[1] = vmax; stats[2] = bail ? INT_MIN : gmax; stats[3] = bail ? INT_MAX : rmin; }
    free(pq); free(L); for (int sl = 0; sl < L16_MAX_SLOTS; sl++) free(snap[sl].L);
#undef SNAP_SAVE
#undef SNAP_PICK
#undef SNAP_LOAD
    if (again && !bail) {           /* started over, on key steps only */
        agatha_lanes16_margin = 0;
        const int rc = agatha_model_lanes16(qs, Q, rs, R, pr, G, S, out3, stats);
        agatha_lanes16_margin = margin;
        return rc == 0 ? (rolled ? 4 : 2) : rc;
    }
    return bail ? 1 : (rolled ? 3 : 0);
}

/* kind[k]: 0 = aligned by the int16 model, 1 = ineligible / bailed out (out arrays hold the int32 model's answer),
 * 2 = aligned by the int16 model after being started over (value steps, agatha_lanes16_set_margin) */
static int l16_batch_margin = 0;
void agatha_lanes16_set_margin(int margin) { l16_batch_margin = margin; }
void agatha_lanes16_batch(const uint8_t *qbatch, const uint8_t *tbatch, const uint32_t *qoff, const uint32_t *toff,
                          const uint32_t *qlen, const uint32_t *tlen, int n, const lm_params_t *pr, int G, int S,
                          int threads, int32_t *score, int32_t *qend, int32_t *tend, int32_t *kind, int32_t *stats4)
{
    (void)threads;
    int32_t gmin = 0, gmax = 0, garb = INT_MIN, rmin = INT_MAX;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 0 ? threads : 1) reduction(min:gmin) reduction(max:gmax) reduction(max:garb) reduction(min:rmin)
#endif
    for (int k = 0; k < n; k++) {
        int32_t o[3] = {0, 0, 0}, st[4] = {0, 0, INT_MIN, INT_MAX};
        agatha_lanes16_margin = l16_batch_margin;
        l16_tl_steps[0] = l16_tl_steps[1] = 0;
        int rc = agatha_model_lanes16((const char *)qbatch + qoff[k], (int)qlen[k], (const char *)tbatch + toff[k],
                                      (int)tlen[k], pr, G, S, o, st);
        if (rc == 1) rc = agatha_model_lanes((const char *)qbatch + qoff[k], (int)qlen[k], (const char *)tbatch + toff[k],
                                             (int)tlen[k], pr, G, S, o) != 0 ? -1 : 1;
        kind[k] = rc;
        if (agatha_lanes16_pair_steps) { agatha_lanes16_pair_steps[2 * k] = l16_tl_steps[0]; agatha_lanes16_pair_steps[2 * k + 1] = l16_tl_steps[1]; }
        score[k] = o[0]; qend[k] = o[1]; tend[k] = o[2];
        if (st[0] < gmin) gmin = st[0];
        if (st[1] > gmax) gmax = st[1];
        if (st[2] > garb) garb = st[2];
        if (st[3] < rmin) rmin = st[3];
    }
    stats4[0] = gmin; stats4[1] = gmax; stats4[2] = garb; stats4[3] = rmin;
}

void agatha_lanes_batch(const uint8_t *qbatch, const uint8_t *tbatch, const uint32_t *qoff, const uint32_t *toff,
                        const uint32_t *qlen, const uint32_t *tlen, int n, const lm_params_t *pr, int G, int S,
                        int threads, int32_t *score, int32_t *qend, int32_t *tend, int *rc)
{
    int bad = 0;
    (void)threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 0 ? threads : 1) reduction(|:bad)
#endif
    for (int k = 0; k < n; k++) {
        int32_t o[3] = {0, 0, 0};
        bad |= agatha_model_lanes((const char *)qbatch + qoff[k], (int)qlen[k], (const char *)tbatch + toff[k],
                                  (int)tlen[k], pr, G, S, o) != 0;
        score[k] = o[0]; qend[k] = o[1]; tend[k] = o[2];
    }
    *rc = bad;
}
