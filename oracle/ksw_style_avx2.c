/*
 * ksw_style_avx2.c -- anti-diagonal SIMD (AVX2, 16 x int16) CPU implementation of the extension DP, in the manner
 * of minimap2's ksw_extz2_sse (which is neither in the reference tree nor in this image).
 *
 * TEST / BENCH INFRASTRUCTURE ONLY (cpu_baseline leg of bench.py).  Semantics = oracle/agatha_oracle.c's
 * agatha_model_exactband(): the reference's recurrence (gaps open from the diagonal term), exact |i-j| <= w band,
 * largest-column tie-break of the per-anti-diagonal maximum, ksw2's z-drop rule.  That equals the reference kernel
 * whenever the alignment does not hug the band edge (always at the BASELINE bands) and all scores
 * fit int16.  tests/test_oracle.py asserts bit-equality with agatha_model_exactband.
 *
 * Layout: everything is indexed by row i (+2 so that the virtual boundary row i = -1 is addressable); one vector
 * covers 16 consecutive rows of one anti-diagonal d = i + j.  Per diagonal the left neighbour (i, j-1) is the same
 * row of diagonal d-1, the upper neighbour (i-1, j) is row i-1 of d-1 and the diagonal neighbour row i-1 of d-2.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>

#if defined(__AVX2__)
#include <immintrin.h>

#define NEG16 ((int16_t)-16384)
#define PAD 2            /* array index = i + PAD: i = -1 (boundary row) and i = -2 stay addressable */
#define TAIL 48          /* slack after the last row for whole-vector stores */

typedef struct { int32_t match, mismatch, gap_open, gap_extend, slice_width, z_threshold, band_width; } ksw_params_t;

static inline int imin_(int a, int b) { return a < b ? a : b; }
static inline int imax_(int a, int b) { return a > b ? a : b; }

/* returns 0 on success, -1 if the pair is outside the int16 domain (caller falls back to the scalar model) */
static int ksw_style_avx2_pair_stop(const char *qs, int Q, const char *rs, int R, const ksw_params_t *pr, int32_t *out3, int *stop_diag)
{
    const int a = pr->match, b = pr->mismatch, oe = pr->gap_open + pr->gap_extend, ge = pr->gap_extend;
    const int w = pr->band_width, z = pr->z_threshold;
    out3[0] = out3[1] = out3[2] = 0;
    if (stop_diag) *stop_diag = -1;
    if (Q <= 0 || R <= 0) return 0;
    if ((long)a * imin_(Q, R) >= 32000 || 16384 + (long)(2 * w + 4) * ge + oe + b >= 32000) return -1;

    const int n = Q + PAD + TAIL;
    int16_t *buf = (int16_t *)aligned_alloc(64, ((sizeof(int16_t) * (size_t)n * 8 + 127) / 64) * 64);
    int16_t *H0 = buf, *H1 = buf + n, *H2 = buf + 2 * n, *E0 = buf + 3 * n, *E1 = buf + 4 * n, *F0 = buf + 5 * n,
            *F1 = buf + 6 * n, *qc = buf + 7 * n;
    for (int k = 0; k < 7 * n; k++) buf[k] = NEG16;
    /* query codes by row; reference codes reversed so that r[d - i] is contiguous in i */
    for (int i = 0; i < n; i++) qc[i] = 0x100;
    for (int i = 0; i < Q; i++) qc[i + PAD] = (int16_t)(qs[i] & 15);
    int16_t *rrev = (int16_t *)malloc(sizeof(int16_t) * (size_t)(R + Q + 2 * TAIL + 64));
    /* rr[x] = code of r[R-1-x]; r[d-i] = rr[R-1-d+i]; out-of-range -> 0x200 (never equal to a query code) */
    int16_t *rr = rrev + Q + TAIL;
    for (int x = -(Q + TAIL); x < R + TAIL; x++) rr[x] = (x >= 0 && x < R) ? (int16_t)(rs[R - 1 - x] & 15) : 0x200;

    const __m256i vA = _mm256_set1_epi16((short)a), vNB = _mm256_set1_epi16((short)-b), vM1 = _mm256_set1_epi16(-1),
                  vOE = _mm256_set1_epi16((short)oe), vGE = _mm256_set1_epi16((short)ge), vN = _mm256_set1_epi16(14);

    /* diagonal d = -2 holds only the corner H(-1,-1) = 0 (row -1 -> index PAD-1): it is "H2" of diagonal 0.
       diagonal d = -1 holds H(-1,0) at row -1 and H(0,-1) at row 0, with their F / E.                           */
    H2[PAD - 1] = 0;
    {
        const int16_t k0 = (int16_t)-(oe);                    /* k(0) */
        H1[PAD - 1] = k0; F1[PAD - 1] = (int16_t)(k0 - oe);   /* (-1, 0): H and the F flowing down into (0,0) */
        H1[PAD + 0] = k0; E1[PAD + 0] = (int16_t)(k0 - oe);   /* (0, -1): H and the E flowing right into (0,0) */
    }
    int best = 0, best_t = 0, best_q = 0, stopped = 0;
    const int lim = Q + R - 1;
    for (int d = 0; d < lim && !stopped; d++) {
        if (stop_diag) *stop_diag = d;
        const int lo = imax_(imax_(0, d - (R - 1)), (d - w + 1) >> 1);      /* ceil((d-w)/2) */
        const int hi = imin_(imin_(Q - 1, d), (d + w) >> 1);                /* floor((d+w)/2) */
        /* boundary cells of THIS diagonal: (-1, d+1) and (d+1, -1) */
        {
            const int c = d + 1;
            const int16_t kc = (c <= w) ? (int16_t)-(oe + ge * c) : NEG16;
            H0[PAD - 1] = kc; F0[PAD - 1] = (c <= w) ? (int16_t)(kc - oe) : NEG16; E0[PAD - 1] = NEG16;
        }
        int vmax = INT_MIN, vrow = -1;
        if (lo <= hi) {
            const int16_t *rd = rr + (R - 1 - d);
            __m256i mx = _mm256_set1_epi16(-32768);
            for (int i = lo; i <= hi; i += 16) {
                const int x = i + PAD;
                const __m256i q = _mm256_loadu_si256((const __m256i *)(qc + x));
                const __m256i r = _mm256_loadu_si256((const __m256i *)(rd + i));
                __m256i s = _mm256_blendv_epi8(vNB, vA, _mm256_cmpeq_epi16(q, r));
                const __m256i isn = _mm256_or_si256(_mm256_cmpeq_epi16(q, vN), _mm256_cmpeq_epi16(r, vN));
                s = _mm256_blendv_epi8(s, vM1, isn);
                const __m256i t = _mm256_add_epi16(_mm256_loadu_si256((const __m256i *)(H2 + x - 1)), s);
                const __m256i fin = _mm256_loadu_si256((const __m256i *)(F1 + x - 1));
                const __m256i ein = _mm256_loadu_si256((const __m256i *)(E1 + x));
                const __m256i h = _mm256_max_epi16(_mm256_max_epi16(t, fin), ein);
                const __m256i tg = _mm256_sub_epi16(t, vOE);
                _mm256_storeu_si256((__m256i *)(F0 + x), _mm256_max_epi16(tg, _mm256_sub_epi16(fin, vGE)));
                _mm256_storeu_si256((__m256i *)(E0 + x), _mm256_max_epi16(tg, _mm256_sub_epi16(ein, vGE)));
                _mm256_storeu_si256((__m256i *)(H0 + x), h);
                if (i + 16 > hi + 1) {          /* partial last vector: lanes past hi must not win the maximum */
                    int16_t tmp[16];
                    _mm256_storeu_si256((__m256i *)tmp, h);
                    for (int k2 = hi + 1 - i; k2 < 16; k2++) tmp[k2] = -32768;
                    mx = _mm256_max_epi16(mx, _mm256_loadu_si256((const __m256i *)tmp));
                } else mx = _mm256_max_epi16(mx, h);
            }
            /* lanes computed past hi are garbage: restore the sentinels the next diagonals will read */
            for (int k2 = hi + 1; k2 < hi + 1 + 17 && k2 + PAD < n; k2++) { E0[k2 + PAD] = NEG16; F0[k2 + PAD] = NEG16; H0[k2 + PAD] = NEG16; }
            /* horizontal maximum, then the smallest row (= largest column) holding it */
            int16_t m16[16];
            _mm256_storeu_si256((__m256i *)m16, mx);
            int m = -32768;
            for (int k2 = 0; k2 < 16; k2++) m = imax_(m, m16[k2]);
            vmax = m;
            const __m256i vm = _mm256_set1_epi16((short)m);
            for (int i = lo; i <= hi && vrow < 0; i += 16) {
                unsigned msk = (unsigned)_mm256_movemask_epi8(_mm256_cmpeq_epi16(_mm256_loadu_si256((const __m256i *)(H0 + i + PAD)), vm));
                while (msk) {
                    const int k2 = __builtin_ctz(msk) >> 1;
                    if (i + k2 <= hi) { vrow = i + k2; break; }
                    msk &= msk - 1; msk &= msk - 1;
                }
            }
        }
        /* the virtual boundary column cell (d+1, -1) sits right after the last real row while hi == d */
        {
            const int c = d + 1;
            if (c < Q + 1 && hi == d) {
                const int16_t kc = (c <= w) ? (int16_t)-(oe + ge * c) : NEG16;
                H0[c + PAD] = kc; E0[c + PAD] = (c <= w) ? (int16_t)(kc - oe) : NEG16; F0[c + PAD] = NEG16;
            }
            if (lo > 0 && lo <= hi) { E0[lo - 1 + PAD] = NEG16; F0[lo - 1 + PAD] = NEG16; }
        }
        /* z-drop bookkeeping (agatha_kernel.h:297-309); an empty diagonal reads (-32768, 0) */
        {
            const int H = (vrow >= 0) ? vmax : -32768;
            const int c = (vrow >= 0) ? d - vrow : 0;
            if (H > best) { best = H; best_t = c; best_q = d - c; }
            else if (c >= best_t && (d - c) >= best_q) {
                const int tl = c - best_t, ql = (d - c) - best_q;
                const int l = tl > ql ? tl - ql : ql - tl;
                if (z >= 0 && best - H > z + l * ge) stopped = 1;
            }
        }
        /* rotate: d -> d-1 -> d-2 */
        int16_t *tH = H2; H2 = H1; H1 = H0; H0 = tH;
        int16_t *tE = E1; E1 = E0; E0 = tE;
        int16_t *tF = F1; F1 = F0; F0 = tF;
    }
    out3[0] = best; out3[1] = best_q; out3[2] = best_t;
    free(buf); free(rrev);
    return 0;
}
int ksw_style_avx2_pair(const char *qs, int Q, const char *rs, int R, const ksw_params_t *pr, int32_t *out3)
{
    return ksw_style_avx2_pair_stop(qs, Q, rs, R, pr, out3, NULL);
}
#else
typedef struct { int32_t match, mismatch, gap_open, gap_extend, slice_width, z_threshold, band_width; } ksw_params_t;
int ksw_style_avx2_pair(const char *qs, int Q, const char *rs, int R, const ksw_params_t *pr, int32_t *out3)
{ (void)qs; (void)Q; (void)rs; (void)R; (void)pr; (void)out3; return -1; }
static int ksw_style_avx2_pair_stop(const char *qs, int Q, const char *rs, int R, const ksw_params_t *pr, int32_t *out3, int *stop_diag)
{ (void)qs; (void)Q; (void)rs; (void)R; (void)pr; (void)out3; (void)stop_diag; return -1; }
#endif

/* from agatha_oracle.c */
typedef struct { int32_t score, query_end, target_end; } oracle_result_t_;
void agatha_model_exactband(const char *qs, int Q, const char *rs, int R, const void *pr, oracle_result_t_ *out);
void agatha_model_exactband_stop(const char *qs, int Q, const char *rs, int R, const void *pr, oracle_result_t_ *out, int *stop_diag);

void ksw_style_batch(const uint8_t *qbatch, const uint8_t *tbatch, const uint32_t *qoff, const uint32_t *toff,
                     const uint32_t *qlen, const uint32_t *tlen, int n, const ksw_params_t *pr, int threads,
                     int32_t *score, int32_t *qend, int32_t *tend, int *n_fallback)
{
    int fb = 0;
    (void)threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 0 ? threads : 1) reduction(+:fb)
#endif
    for (int k = 0; k < n; k++) {
        int32_t o[3];
        const char *q = (const char *)qbatch + qoff[k], *t = (const char *)tbatch + toff[k];
        if (ksw_style_avx2_pair(q, (int)qlen[k], t, (int)tlen[k], pr, o) != 0) {
            oracle_result_t_ r;
            agatha_model_exactband(q, (int)qlen[k], t, (int)tlen[k], pr, &r);
            o[0] = r.score; o[1] = r.query_end; o[2] = r.target_end; fb++;
        }
        score[k] = o[0]; qend[k] = o[1]; tend[k] = o[2];
    }
    if (n_fallback) *n_fallback = fb;
}

/* the same, which also reports the cell anti-diagonal each pair's z-drop walk ended on (Q + R - 2: it never stopped):
   bench.py's "effective cells"  */
void ksw_style_batch_stops(const uint8_t *qbatch, const uint8_t *tbatch, const uint32_t *qoff, const uint32_t *toff,
                           const uint32_t *qlen, const uint32_t *tlen, int n, const ksw_params_t *pr, int threads,
                           int32_t *score, int32_t *qend, int32_t *tend, int32_t *stop_diag, int *n_fallback)
{
    int fb = 0;
    (void)threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 0 ? threads : 1) reduction(+:fb)
#endif
    for (int k = 0; k < n; k++) {
        int32_t o[3];
        int sd = -1;
        const char *q = (const char *)qbatch + qoff[k], *t = (const char *)tbatch + toff[k];
        if (ksw_style_avx2_pair_stop(q, (int)qlen[k], t, (int)tlen[k], pr, o, &sd) != 0) {
            oracle_result_t_ r;
            agatha_model_exactband_stop(q, (int)qlen[k], t, (int)tlen[k], pr, &r, &sd);
            o[0] = r.score; o[1] = r.query_end; o[2] = r.target_end; fb++;
        }
        score[k] = o[0]; qend[k] = o[1]; tend[k] = o[2]; stop_diag[k] = sd;
    }
    if (n_fallback) *n_fallback = fb;
}
