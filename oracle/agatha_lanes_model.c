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

#define NEG_INF2 (-16384)
#define N_VALUE 14
#define MAXG 64
#define MAXS 8

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
