// align_kernel.hip -- banded affine-gap extension DP with z-drop for MI355X (gfx950, wave64).
//
// Replaces the reference's agatha_kernel (AGAThA/src/kernels/agatha_kernel.h:49-431) and its
// length sort (agatha_sort :434-458 + the host std::sort in gasal_align.cu:14-18).  Results are
// bit-identical to the reference wherever the reference is defined (lengths < 32768, |H| < 32768);
// outside that domain the arithmetic simply stays int32 ("wide" semantics).
//
// Design (not a translation of the CUDA kernel):
//   * one sequence pair per G-lane sub-wavefront (G = 16/32/64), 64/G pairs per wave, no LDS state,
//     no global scratch (the reference keeps 3 strips of L entries per subwarp in global memory).
//   * the DP advances one BLOCK-anti-diagonal ("step") at a time; column block r is owned for its
//     whole life by slot r % S of lane (r / S) % G, so the column state (H, F of 8 columns) never
//     leaves VGPRs.  The row state (H, E of 8 rows) of block (q, r) is the input of block (q, r+1)
//     one step later: next slot of the same lane, or lane+1 through one cross-lane rotate.
//   * the per-anti-diagonal maximum (value + largest column) is a packed key (H << K) + column
//     relative to a moving base, accumulated in 15 registers per lane and reduced over the group
//     once per step; z-drop is tested eagerly every step, which is equivalent to the reference's
//     per-slice test (oracle/agatha_oracle.c: agatha_model_steps == agatha_model_slices).
//   * pairs are pulled longest-first from an atomic queue by whichever group is free
//     (the reference's uneven bucketing + subwarp rejoining solve the same imbalance differently).
//
// The lane-for-lane CPU emulation of exactly this schedule is oracle/agatha_lanes_model.c.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <limits.h>

#include "kernels.h"
#include "device_common.h"

namespace agatha {

// initial column state of column block r: H(-1, c), F(0, c)  (agatha_kernel.h:133-148, 207-215)
__device__ __forceinline__ void init_col(int r, int R, int w, int gapoe, int ge, int neg, int (&h)[8], int (&f)[8], int& corner)
{
#pragma unroll
    for (int m = 0; m < 8; m++) {
        const int c = 8 * r + m;
        const int k = -(gapoe + ge * c);
        const bool in = (c < R) && (c <= w);
        h[m] = in ? k : neg;
        f[m] = in ? k - gapoe : neg;
    }
    const int kc = -(gapoe + ge * (8 * r - 1));
    corner = (r == 0) ? 0 : ((8 * r - 1) <= w ? kc : neg);
}

// One 8x8 block (q, r): the reference's CORE_COMPUTE / CORE_COMPUTE_BOUNDARY sweep
// (agatha_kernel.h:20-46, 230-269) on register state.
//   h, f    column state H(row above, c), F(next row, c) of the 8 columns       (in/out)
//   corner  H(row above, column left of the block)                             (in/out)
//   rh      H(row, column left of the block) for the 8 rows                    (in)
//   e       E(row, first column) in, E(row, column right of the block) out     (in/out)
//   oh      H(row, last column) out
// Substitution scores come from a per-(lane, slot) PROFILE in LDS: for the 8 reference bases of this column block,
// one row of 8 signed bytes per query-base class (A, C, G, T, N), built once when the column block starts.  A row of
// the block then costs one ds_read_b64 and each cell's "score + diagonal" is a single v_add_u32_sdwa (sign-extended
// byte operand) instead of compare + select + add.  Pairs that contain letters outside ACGTN (flagged by
// exotic_kernel) take the compare path (use_cmp, wave-uniform), which also carries the N rule of
// gasal_kernels.h:48-50.
// MASKED: per-cell band test of boundary blocks and the row limit of the last row block, as EXEC masks
//         built once per block (km: one lane mask per cell diagonal jl-il).
template <bool MASKED, bool CMP, int K>
__device__ __forceinline__ void block8x8(int (&h)[8], int (&f)[8], int& corner, const int (&rh)[8], int (&e)[8],
                                         int (&oh)[8], int (&A)[15], uint32_t qword, uint32_t rword,
                                         const uint2* __restrict__ prof, int va, int vnb, int gapoe, int ge,
                                         int crel0, int nrows, int tu, int tl, int t0)
{
    // all eight profile rows are requested up front: their LDS latency then hides behind the mask set-up and the
    // first rows instead of stalling every row (class of a query base = bits 3..1 of its code:
    // A(1)->0 C(3)->1 T(4)->2 G(7)->3 N(14)->7)
    uint2 pw[8];
    if (!CMP) {
#pragma unroll
        for (int il = 0; il < 4; il++) pw[il] = prof[((qword >> (29 - 4 * il)) & 7u) * 64u];
    }
    int cj[8];
#pragma unroll
    for (int jl = 0; jl < 8; jl++) cj[jl] = crel0 + jl;
    unsigned long long km[15];
    if (MASKED) {
        // Away from the matrix corners an edge block has tu == t0 (upper edge) or tl == t0 (lower edge), t0 = w - 8W,
        // and every other block has no skipped cell: then the 15 lane masks are two ballots combined with
        // wave-uniform predicates in SALU.  Anything else (clamped corners, bands narrower than a block) takes the
        // general per-lane compares.
        const bool up = (tu == t0) && (tl >= 7), lo = (tl == t0) && (tu >= 7), none = (tu >= 7) && (tl >= 7);
        if (__builtin_expect(__all(up || lo || none), 1)) {
            const unsigned long long mu = __builtin_amdgcn_ballot_w64(up), ml = __builtin_amdgcn_ballot_w64(lo);
#pragma unroll
            for (int kk = 0; kk < 15; kk++)
                km[kk] = ~(((kk - 7) > t0 ? mu : 0ull) | ((7 - kk) > t0 ? ml : 0ull));
        } else {
#pragma unroll
            for (int kk = 0; kk < 15; kk++)     // tu, tl >= -7 always, so kk = 0 never fails the first test and kk = 14 never the second
                km[kk] = (kk == 0 ? ~0ull : __builtin_amdgcn_ballot_w64((kk - 7) <= tu)) &
                         (kk == 14 ? ~0ull : __builtin_amdgcn_ballot_w64((7 - kk) <= tl));
        }
    }
#pragma unroll
    for (int il = 0; il < 8; il++) {
        if (!CMP && il == 1) {                  // second half of the profile rows: requested three rows ahead of use
#pragma unroll
            for (int i2 = 4; i2 < 8; i2++) pw[i2] = prof[((qword >> (29 - 4 * i2)) & 7u) * 64u];
        }
        if (!MASKED || il < nrows) {            // rows past the end of the query exist only in the last row block
            int t[8];
            if (CMP) {
                const uint32_t qb = (qword >> (28 - 4 * il)) & 15u;
#pragma unroll
                for (int jl = 0; jl < 8; jl++) {
                    const uint32_t rb = (rword >> (28 - 4 * jl)) & 15u;
                    int sc = (qb == rb) ? va : vnb;
                    sc = (qb == N_VALUE || rb == N_VALUE) ? -1 : sc;
                    const int d = (jl == 0) ? ((il == 0) ? corner : rh[il - 1]) : h[jl - 1];
                    t[jl] = sc + d;
                }
            } else {
                const uint2 w = pw[il];
#pragma unroll
                for (int jl = 0; jl < 8; jl++) {
                    const uint32_t word = (jl & 1) ? w.y : w.x;            // even columns in .x, odd in .y
                    const int sc = (int)(int8_t)((word >> (8 * (3 - (jl >> 1)))) & 0xffu);
                    const int d = (jl == 0) ? ((il == 0) ? corner : rh[il - 1]) : h[jl - 1];
                    t[jl] = sc + d;
                }
            }
            // every diagonal term is taken from the PREVIOUS row's H: pin them before H is overwritten
#pragma unroll
            for (int jl = 0; jl < 8; jl++) asm volatile("" : "+v"(t[jl]));
            int ev = e[il];
#pragma unroll
            for (int jl = 0; jl < 8; jl++) {
                if (!MASKED || __builtin_amdgcn_inverse_ballot_w64(km[jl - il + 7])) {
                    const int hn = imax3(t[jl], f[jl], ev);
                    const int tg = t[jl] - gapoe;
                    f[jl] = imax(tg, f[jl] - ge);
                    ev = imax(tg, ev - ge);
                    h[jl] = hn;
                    A[il + jl] = imax(A[il + jl], (int)(((uint32_t)hn << K) + (uint32_t)cj[jl]));
                }
            }
            oh[il] = h[7]; e[il] = ev;
        }
    }
    // p[1] = h[0] of the last processed row (agatha_kernel.h:28).  A block with fewer than 8 rows is the last
    // block of its column (q == pql-1), after which the corner is never read again, so row 7 is always right.
    corner = rh[7];
}

// Score profile of one column block: rows of 8 signed bytes (even columns in .x, odd in .y, column 0/1 in the top
// byte) for the query-base classes 0..3 = A, C, T, G and 7 = N.  SWAR on the 8 packed reference codes.
__device__ __forceinline__ void build_profile(uint2* __restrict__ prof, uint32_t rword, int a, int b)
{
    const uint32_t Re = (rword >> 4) & 0x0F0F0F0Fu, Ro = rword & 0x0F0F0F0Fu;  // columns 0,2,4,6 / 1,3,5,7
    const uint32_t A4 = ((uint32_t)a & 0xFFu) * 0x01010101u, NB4 = ((uint32_t)(-b) & 0xFFu) * 0x01010101u;
    const uint32_t BMe = NB4 | eq_bytes(Re, 0x0E0E0E0Eu), BMo = NB4 | eq_bytes(Ro, 0x0E0E0E0Eu);   // -b, or -1 where ref is N
    const uint32_t codes[4] = {0x01010101u, 0x03030303u, 0x04040404u, 0x07070707u};              // A C T G
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const uint32_t me = eq_bytes(Re, codes[c]), mo = eq_bytes(Ro, codes[c]);
        prof[c * 64] = make_uint2((A4 & me) | (BMe & ~me), (A4 & mo) | (BMo & ~mo));
    }
    prof[7 * 64] = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);                                          // query N: always -1
}

template <int G, int S, bool CMP>
__global__ void __launch_bounds__(256, (S <= 3 ? 2 : 1))
align_kernel(const AlignLaunch* __restrict__ La, AlignParams P, int kid, int takes2)
{
    // kid: index of this kernel among the candidates for the plain (kind-0) pairs, chosen on the device (record_kernel);
    // takes2: this launch also takes the pairs the int16 kernel handed over / could not take (kind 2)
    if (!CMP && !takes2 && *La->choice != kid) return;
    if (!CMP && takes2 && *La->choice != kid && La->kind_counts[1] == 0u) return;      // nothing was handed over
    if (CMP && !La->force_cmp && La->kind_counts[0] == 0u) return;                      // no pair with other letters
    // Batch pointers are read from the launch record only where a pair starts or ends: keeping a dozen 64-bit
    // pointers live through the DP loop would push the band masks (30 SGPRs) into spills.
    constexpr int GS = G * S;
    constexpr int K = KeyBits<GS>::value;
    constexpr int KMASK = (1 << K) - 1;
    constexpr int NEGK = NEG_INF2;

    __shared__ uint2 s_prof[4 * S * 8 * 64];      // [wave][slot][class][lane] score profiles (block8x8)
    const int lane = threadIdx.x & 63;
    uint2* const prof0 = s_prof + (threadIdx.x >> 6) * (S * 8 * 64) + lane;
    const int k = lane & (G - 1);                 // lane inside the group
    const int gbase = lane & ~(G - 1);            // first lane of the group
    const int left_lane = gbase | ((k + G - 1) & (G - 1));

    const int gapoe = P.gap_open + P.gap_extend, ge = P.gap_extend;
    const int sw = P.slice_width, z = P.z_threshold, w = P.band_width;
    const int W = (w + 7) >> 3;
    int va = P.match, vnb = -P.mismatch;               // kept in VGPRs: both arms of the score select (compare path)

    // ---- per-pair state (uniform inside a group) ----
    int Q = 0, R = 0, pql = 0, prl = 0, total = 0, lim = 0, pair = 0;
    // sequence words are read through explicit global (address space 1) pointers: a pointer loaded from the launch
    // record is otherwise treated as generic and costs flat_load + a wait on both memory counters
    typedef const __attribute__((address_space(1))) uint32_t* gptr_t;
    gptr_t pq = nullptr;
    gptr_t pt = nullptr;
    int i = 0, y = 0, ss = 0, se = 0, cb_prev = 0;
    bool alive = false, exhausted = false, final_step = false;
    int best = 0, best_t = 0, best_q = 0;

    // ---- per-lane state ----
    int rcur[S], corner[S];
    int h[S][8], f[S][8];
    uint32_t rword[S];
    uint32_t qcur[S];                              // packed query word of the row block each slot works on this step
    int xh[S + 1][8], xe[S + 1][8], xr[S + 1];   // hand-off: X[s] feeds slot s, X[S] leaves slot S-1
    int A[15];

#pragma unroll
    for (int s = 0; s < S; s++) { rcur[s] = 0; corner[s] = 0; rword[s] = 0; qcur[s] = 0;
#pragma unroll
        for (int m = 0; m < 8; m++) { h[s][m] = 0; f[s][m] = 0; } }
#pragma unroll
    for (int s = 0; s <= S; s++) { xr[s] = -2;
#pragma unroll
        for (int m = 0; m < 8; m++) { xh[s][m] = 0; xe[s][m] = 0; } }
#pragma unroll
    for (int x = 0; x < 15; x++) A[x] = INT_MIN;

    for (;;) {
        // ------------------------------------------------------------------ work queue
        const bool need = !alive && !exhausted;
        if (__builtin_expect(__any(need), 0)) {
            int idx = 0;
            if (need && k == 0) idx = (int)atomicAdd(La->queue + (CMP ? 2 : takes2 ? 1 : 3), 1u);      // one queue head per launch
            idx = lane_read(idx, gbase);
            if (need) {
                if (idx >= La->n) exhausted = true;
                else {
                    pair = (int)La->order[idx];
                    // two launches share the work: this instantiation only takes the pairs of its kind (the profile
                    // kernel skips pairs with letters outside ACGTN, the compare kernel takes exactly those)
                    // pair kinds: 0 = plain letters (the chosen candidate kernel), 1 = letters outside ACGTN (compare
                    // kernel), 2 = handed over by / withheld from the packed-int16 kernel (int32 profile kernel)
                    const int kind = La->exotic[pair];
                    const bool mine = CMP ? (La->force_cmp || kind == 1) : ((takes2 && kind == 2) || (kind == 0 && *La->choice == kid));
                    if (mine) {
                    Q = (int)La->qlens[pair]; R = (int)La->tlens[pair];
                    pq = (gptr_t)(La->packed_q + (La->qoffs[pair] >> 3));
                    pt = (gptr_t)(La->packed_t + (La->toffs[pair] >> 3));
                    pql = (Q + 7) >> 3; prl = (R + 7) >> 3;
                    total = prl + pql - 1; lim = Q + R - 1;
                    best = 0; best_t = 0; best_q = 0;
                    i = 0; y = 0; cb_prev = 0; final_step = false;
                    ss = 0;
                    se = imin(imin(prl - 1, sw - 1), (((sw - 1) * 8 + 7 + w) / 2) / 8);
#pragma unroll
                    for (int s = 0; s < S; s++) {
                        rcur[s] = k * S + s;
                        init_col(rcur[s], R, w, gapoe, ge, NEGK, h[s], f[s], corner[s]);
                        rword[s] = (rcur[s] < prl) ? pt[rcur[s]] : 0xEEEEEEEEu;
                        if (!CMP) build_profile(prof0 + s * (8 * 64), rword[s], P.match, P.mismatch);
                        const int q0 = 0 - rcur[s];                      // row block of step 0 (only column block 0 has one)
                        qcur[s] = (q0 >= 0 && q0 < pql) ? pq[q0] : 0u;
                    }
#pragma unroll
                    for (int s = 0; s <= S; s++) xr[s] = -2;
#pragma unroll
                    for (int x = 0; x < 15; x++) A[x] = INT_MIN;
                    alive = true;
                    if (Q <= 0 || R <= 0) {           // nothing to align
                        if (k == 0) { La->score[pair] = 0; La->qend[pair] = 0; La->tend[pair] = 0; }
                        alive = false;
                    } else if (imin(W + 1, imin(pql, prl)) > GS) {
                        // the caller's length hint was too small for this pair: refuse loudly instead of aligning
                        // with a window that does not hold the band (include/agatha_amd.h: AGATHA_AMD_BAD_RESULT)
                        if (k == 0) { La->score[pair] = INT_MIN; La->qend[pair] = -1; La->tend[pair] = -1; }
                        alive = false;
                    }
                    }
                }
            }
        }
        if (!__any(alive)) {
            if (__all(exhausted)) break;
            continue;                              // every group drew a pair of the other kind: draw again
        }
        // the steps run in an inner loop of their own, left only when a group wants a new pair (with the queue code in
        // the same loop the register allocator spills)
        do {

        // ------------------------------------------------------------------ one step
        // column base of the packed maxima: one block left of the lowest active column block
        const int cb = 8 * imax(0, imax(i - pql + 1, (i - W + 1) >> 1) - 1);
        {
            const int delta = cb - cb_prev;       // 0 or 8; saturating so that the empty marker INT_MIN survives
#pragma unroll
            for (int x = 0; x < 7; x++) A[x] = __builtin_elementwise_sub_sat(A[x], delta);
        }

        // Which slots leave their column after this step is known now (used by the prefetch after the slot loop).
        bool adv[S];
        bool any_adv = false;
#pragma unroll
        for (int s = 0; s < S; s++) {
            adv[s] = alive && (i + 1 - rcur[s] > imin(pql - 1, rcur[s] + W));
            any_adv |= adv[s];
        }

#pragma unroll
        for (int s = S - 1; s >= 0; s--) {
            const int r = rcur[s], q = i - r;
            const int cs = imax(0, r - W), ce = imin(pql - 1, r + W);
            const bool active = alive && !final_step && r < prl && q >= cs && q <= ce && r >= ss && r <= se;
            xr[s + 1] = active ? r : -2;
            const uint32_t qword = qcur[s];
            const uint32_t rw = rword[s];
            const int nrows = imin(8, Q - 8 * q);
            const bool boundary = (q == cs) || (q == ce);               // agatha_kernel.h:243
            if (active) {
                if (__builtin_expect(y == 0 && r == prl - 1, 0)) {   // pass start: padded ref columns fall back to -inf (agatha_kernel.h:207-215)
#pragma unroll
                    for (int m = 0; m < 8; m++) if (8 * r + m >= R) { h[s][m] = NEGK; f[s][m] = NEGK; }
                }
                const bool left_ok = (xr[s] == r - 1);
                int rh[8];
                if (__builtin_expect(__any(!left_ok && 8 * q <= w), 0)) {
                    // a row block starts inside the first w rows: its left boundary holds real gap scores
#pragma unroll
                    for (int il = 0; il < 8; il++) {
                        const int row = 8 * q + il;
                        const int kk = -(gapoe + ge * row);
                        const int ih = (row <= w) ? kk : NEGK;                // H(row, -1)   (agatha_kernel.h:126-131)
                        const int ie = (row <= w) ? kk - gapoe : NEGK;        // E(row, 0)
                        rh[il] = left_ok ? xh[s][il] : ih;
                        xe[s + 1][il] = left_ok ? xe[s][il] : ie;             // E travels in place through the block
                    }
                } else {
#pragma unroll
                    for (int il = 0; il < 8; il++) {
                        rh[il] = left_ok ? xh[s][il] : NEGK;
                        xe[s + 1][il] = left_ok ? xe[s][il] : NEGK;
                    }
                }
                const int tu = boundary ? w + 8 * q - 8 * r : 1000;     // cell skipped when jl - il > tu (:33)
                const int tl = boundary ? w - 8 * q + 8 * r : 1000;     //                or il - jl > tl
                const int crel0 = 8 * r - cb;
                // Edge blocks (band test) exist on every anti-diagonal, so a mask-free variant would rarely run for
                // a whole wave; Ns are rare (padding of the last column block, occasional N in a read).
                block8x8<true, CMP, K>(h[s], f[s], corner[s], rh, xe[s + 1], xh[s + 1], A, qword, rw, prof0 + s * (8 * 64),
                                       va, vnb, gapoe, ge, crel0, nrows, tu, tl, w - 8 * W);
            }
        }
        // Prefetch for step i + 1, issued once per step: every block of this step is done, so the registers are free,
        // and the reduce / z-drop tail below hides the latency (nothing else waits on vmcnt in between).
#pragma unroll
        for (int s = 0; s < S; s++) {
            const int rn = adv[s] ? rcur[s] + GS : rcur[s];
            const int qn = i + 1 - rn;
            uint32_t qv = 0u;
            if (alive && qn >= 0 && qn < pql) qv = pq[qn];
            qcur[s] = qv;
            if (adv[s]) rword[s] = (rn < prl) ? pt[rn] : 0xEEEEEEEEu;
        }
        // X[S] of the left neighbour lane becomes X[0]
#pragma unroll
        for (int il = 0; il < 8; il++) { xh[0][il] = lane_read(xh[S][il], left_lane); xe[0][il] = lane_read(xe[S][il], left_lane); }
        xr[0] = lane_read(xr[S], left_lane);

        // ------------------------------------------------------------------ anti-diagonals 8i..8i+7 are complete
        bool stopped = false;
        int vred[8];
#pragma unroll
        for (int x = 0; x < 8; x++) vred[x] = A[x];
        group_max8<G>(vred, lane);
        // Fast path (wave-uniform): every anti-diagonal of this step is non-empty, inside the pair, and within z of the
        // running maximum, so z-drop cannot fire (agatha_kernel.h:304 needs best - H > z + l*ge >= z) and only the
        // running maximum and its position have to be advanced.
        bool calm = !final_step && (8 * i + 7 < lim);
        {
            int lo8 = vred[0];
#pragma unroll
            for (int x = 1; x < 8; x++) lo8 = imin(lo8, vred[x]);
            int hi8 = vred[0];
#pragma unroll
            for (int x = 1; x < 8; x++) hi8 = imax(hi8, vred[x]);
            calm = calm && lo8 != INT_MIN && (z < 0 || imax(best, hi8 >> K) - (lo8 >> K) <= z);
        }
        if (__builtin_expect(__all(calm || !alive), 1)) {
#pragma unroll
            for (int x = 0; x < 8; x++) {
                const int H = vred[x] >> K;
                if (alive && H > best) { best = H; best_t = (vred[x] & KMASK) + cb; best_q = 8 * i + x - best_t; }
            }
        } else {
#pragma unroll
            for (int x = 0; x < 8; x++) {
                const int v = vred[x];
                const int d = 8 * i + x;
                const bool chk = alive && !stopped && (final_step || d < lim);       // agatha_kernel.h:293-294 / 337
                int H = v >> K, c = (v & KMASK) + cb;
                if (v == INT_MIN) { H = -32768; c = 0; }                              // empty anti-diagonal
                if (chk) {                                                           // agatha_kernel.h:297-309
                    if (H > best) { best = H; best_t = c; best_q = d - c; }
                    else if (c >= best_t && (d - c) >= best_q) {
                        const int tlen = c - best_t, qlen = (d - c) - best_q;
                        const int l = tlen > qlen ? tlen - qlen : qlen - tlen;
                        if (z >= 0 && best - H > z + l * ge) stopped = true;
                    }
                }
            }
        }
        bool finished = alive && (stopped || final_step);

        // carry dl 8..14 into the next step
#pragma unroll
        for (int x = 0; x < 7; x++) A[x] = A[8 + x];
#pragma unroll
        for (int x = 7; x < 15; x++) A[x] = INT_MIN;
        cb_prev = cb;

        // slots whose column block has left the band move on to column r + G*S
        if (__any(any_adv)) {
#pragma unroll
            for (int s = 0; s < S; s++) {
                if (adv[s]) {
                    const int rn = rcur[s] + GS;
                    rcur[s] = rn;
                    init_col(rn, R, w, gapoe, ge, NEGK, h[s], f[s], corner[s]);
                    if (!CMP) build_profile(prof0 + s * (8 * 64), rword[s], P.match, P.mismatch);
                }
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
        if (__builtin_expect(finished, 0)) {
            if (k == 0) { La->score[pair] = best; La->qend[pair] = best_q; La->tend[pair] = best_t; }   // :359-363
            alive = false;
        }
        } while (!__any(!alive && !exhausted));
    }
}

// ---------------------------------------------------------------------------------------------------
// Length sort on the device (the reference does it on the host inside the timed region,
// gasal_align.cu:14-18): counting sort of pair ids by step count, longest first.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t sort_key(uint32_t ql, uint32_t tl, uint32_t nbuckets)
{
    const uint32_t steps = ((ql + 7) >> 3) + ((tl + 7) >> 3);       // total block anti-diagonals + 1
    const uint32_t b = steps >> 2;                                   // 32-base granularity is plenty
    return b < nbuckets ? b : nbuckets - 1;
}

__global__ void sort_hist_kernel(const uint32_t* __restrict__ qlens, const uint32_t* __restrict__ tlens, int n,
                                 uint32_t* __restrict__ hist, uint32_t nbuckets)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) atomicAdd(&hist[sort_key(qlens[t], tlens[t], nbuckets)], 1u);
}

// one workgroup: exclusive scan of the histogram from the LONGEST bucket down
__global__ void sort_scan_kernel(uint32_t* __restrict__ hist, uint32_t nbuckets, float* __restrict__ totals)
{
    __shared__ uint32_t part[256];
    __shared__ float fsum[256], fmax[256];
    const uint32_t per = (nbuckets + 255) / 256;
    const uint32_t t = threadIdx.x;
    uint32_t sum = 0;
    float steps_sum = 0.f, steps_max = 0.f;          // what the kernel choice needs: total and longest step count
    for (uint32_t j = 0; j < per; j++) {
        const uint32_t b = t * per + j;
        if (b < nbuckets) {
            const uint32_t bucket = nbuckets - 1 - b, c = hist[bucket];
            sum += c;
            const float st = 4.f * (float)bucket + 2.f;   // sort_key: bucket = steps >> 2
            steps_sum += (float)c * st;
            if (c) steps_max = fmaxf(steps_max, st);
        }
    }
    part[t] = sum; fsum[t] = steps_sum; fmax[t] = steps_max;
    __syncthreads();
    if (t == 0) {
        float a = 0.f, m = 0.f;
        for (int j = 0; j < 256; j++) { a += fsum[j]; m = fmaxf(m, fmax[j]); }
        totals[0] = a; totals[1] = m;
    }
    if (t == 0) { uint32_t acc = 0; for (int j = 0; j < 256; j++) { const uint32_t v = part[j]; part[j] = acc; acc += v; } }
    __syncthreads();
    uint32_t acc = part[t];
    for (uint32_t j = 0; j < per; j++) {
        const uint32_t b = t * per + j;
        if (b < nbuckets) { const uint32_t v = hist[nbuckets - 1 - b]; hist[nbuckets - 1 - b] = acc; acc += v; }
    }
}

__global__ void sort_scatter_kernel(const uint32_t* __restrict__ qlens, const uint32_t* __restrict__ tlens, int n,
                                    uint32_t* __restrict__ cursor, uint32_t nbuckets, uint32_t* __restrict__ order)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) order[atomicAdd(&cursor[sort_key(qlens[t], tlens[t], nbuckets)], 1u)] = (uint32_t)t;
}

// ---------------------------------------------------------------------------------------------------
// ASCII -> 4-bit packing, 8 bases per uint32, first base in bits 31-28 (replaces gasal_pack_kernel,
// pack_rc_seqs.h:13-53).  16 input bytes -> 8 output bytes per lane per iteration, fully coalesced.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t pack8(uint32_t lo, uint32_t hi)
{
    // byte k of (lo, hi) -> nibble 7-k
    uint32_t v = 0;
    v |= (lo & 15u) << 28; v |= ((lo >> 8) & 15u) << 24; v |= ((lo >> 16) & 15u) << 20; v |= ((lo >> 24) & 15u) << 16;
    v |= (hi & 15u) << 12; v |= ((hi >> 8) & 15u) << 8; v |= ((hi >> 16) & 15u) << 4; v |= ((hi >> 24) & 15u);
    return v;
}

__global__ void __launch_bounds__(256)
pack_kernel(const uint4* __restrict__ in16, uint2* __restrict__ out8, uint32_t n16,
            const uint2* __restrict__ in8_tail, uint32_t* __restrict__ out4_tail, uint32_t has_tail)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n16; t += stride) {
        const uint4 v = in16[t];
        uint2 o; o.x = pack8(v.x, v.y); o.y = pack8(v.z, v.w);
        out8[t] = o;
    }
    if (has_tail && blockIdx.x == 0 && threadIdx.x == 0) { const uint2 v = *in8_tail; *out4_tail = pack8(v.x, v.y); }
}

// ---------------------------------------------------------------------------------------------------
// Prepass: flag the pairs whose sequences hold a letter outside {A, C, G, T, N} (any case).  The alignment kernel's
// score profile only has rows for those five classes; flagged pairs use its compare path.  One wave per pair.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool word_is_plain(uint32_t v)
{
    bool ok = true;
#pragma unroll
    for (int k2 = 0; k2 < 8; k2++) {
        const uint32_t c = (v >> (4 * k2)) & 15u;
        ok = ok && ((0x409Au >> c) & 1u);          // bits 1, 3, 4, 7, 14
    }
    return ok;
}

// true if one of the first `nbases` (1..8) bases of the packed word is N
__device__ __forceinline__ bool word_has_n(uint32_t v, uint32_t nbases)
{
    bool has = false;
#pragma unroll
    for (uint32_t k2 = 0; k2 < 8; k2++) has = has || (k2 < nbases && ((v >> (28 - 4 * k2)) & 15u) == N_VALUE);
    return has;
}

// kind of each pair: 1 = a sequence holds a letter outside {A, C, G, T, N} (compare kernel); 2 = the QUERY holds an N
// (the packed-int16 kernel's score profile has no row for it: int32 profile kernel); 0 = everything else
__global__ void __launch_bounds__(256)
exotic_kernel(const uint32_t* __restrict__ packed_q, const uint32_t* __restrict__ packed_t,
              const uint32_t* __restrict__ qlens, const uint32_t* __restrict__ tlens,
              const uint32_t* __restrict__ qoffs, const uint32_t* __restrict__ toffs, uint8_t* __restrict__ exotic, int n,
              unsigned int* __restrict__ kind_counts, AlignParams P, long long score_limit,
              int32_t* __restrict__ score, int32_t* __restrict__ qend, int32_t* __restrict__ tend)
{
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int p = wave; p < n; p += nwaves) {
        bool plain = true, qn = false;
        const uint32_t* a = packed_q + (qoffs[p] >> 3);
        const uint32_t ql = qlens[p], na = (ql + 7u) >> 3;
        for (uint32_t i = lane; i < na; i += 64u) {
            const uint32_t v = a[i];
            plain = plain && word_is_plain(v);
            qn = qn || word_has_n(v, (i + 1u < na) ? 8u : ql - 8u * i);
        }
        const uint32_t* b = packed_t + (toffs[p] >> 3);
        const uint32_t nb = (tlens[p] + 7u) >> 3;
        for (uint32_t i = lane; i < nb; i += 64u) plain = plain && word_is_plain(b[i]);
        const bool all_plain = __all(plain), any_qn = __any(qn);
        if (lane == 0) {
            int kind = all_plain ? (any_qn ? 2 : 0) : 1;
            // kind 3: the scores this pair can reach do not fit the kernels' H << K keys (only possible when the caller
            // gave no length hints: with hints the host refuses the whole call, AGATHA_AMD_ERANGE): no kernel takes it
            const long long Q = ql, R = tlens[p], lmin = Q < R ? Q : R, lmax = Q > R ? Q : R;
            const long long top = lmin * (P.match > 1 ? P.match : 1) + 16384 + 2ll * (P.band_width + 8) * P.gap_extend;
            const long long per = P.mismatch > 2 * P.gap_extend ? P.mismatch : 2 * P.gap_extend;
            const long long low = P.z_threshold < 0 ? lmax * (per > 1 ? per : 1) + P.gap_open + 16384 : 0;
            if (top >= score_limit || low >= score_limit) {
                kind = 3;
                score[p] = INT_MIN; qend[p] = -1; tend[p] = -1;          // AGATHA_AMD_BAD_RESULT
            }
            exotic[p] = (uint8_t)kind;
            if (kind == 1 || kind == 2) atomicAdd(kind_counts + (kind - 1), 1u);
        }
    }
}

hipError_t launch_exotic(const AlignLaunch& L, hipStream_t st)
{
    int blocks = (L.n + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(exotic_kernel, dim3(blocks), dim3(256), 0, st, L.packed_q, L.packed_t, L.qlens, L.tlens, L.qoffs,
                       L.toffs, L.exotic, L.n, L.kind_counts, L.p, L.score_limit, L.score, L.qend, L.tend);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Per-sequence reverse / complement (replaces gasal_reversecomplement_kernel, pack_rc_seqs.h:56-212, which the
// reference runs in place on the packed words with one thread per pair).  Here the packed words of a sequence
// whose op code is non-zero are simply re-derived from the unpacked ASCII that is still in HBM: one wave per
// sequence, coalesced word stores, no in-place hazards.  op bit 0 = reverse, bit 1 = complement (A<->T, C<->G;
// every other code is left alone, pack_rc_seqs.h:183-199).  Reversal uses the TRUE length: base p becomes old base
// len-1-p, positions >= len of the last word stay N.  (As written, the reference counts padding Ns by comparing a
// nibble with N_CODE = 0x4E and therefore always finds zero, so for len % 8 != 0 it rotates the padding to the
// front; for len % 8 == 0 both agree.  See DESIGN.md.)
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t comp_code(uint32_t c)
{
    return c == 1u ? 4u : c == 4u ? 1u : c == 3u ? 7u : c == 7u ? 3u : c;
}

__global__ void __launch_bounds__(256)
seq_ops_kernel(const uint8_t* __restrict__ unpacked, uint32_t* __restrict__ packed, const uint32_t* __restrict__ lens,
               const uint32_t* __restrict__ offsets, const uint8_t* __restrict__ ops, uint32_t n)
{
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t s = wave; s < n; s += nwaves) {
        const uint32_t op = ops[s] & 3u;
        if (op == 0u) continue;
        const uint32_t len = lens[s], off = offsets[s];
        const uint32_t nw = (len + 7u) >> 3;
        const uint8_t* src = unpacked + off;
        uint32_t* dst = packed + (off >> 3);
        for (uint32_t wi = lane; wi < nw; wi += 64u) {
            uint32_t v = 0;
#pragma unroll
            for (uint32_t k = 0; k < 8u; k++) {
                const uint32_t p = 8u * wi + k;
                uint32_t c = N_VALUE;
                if (p < len) {
                    c = (uint32_t)src[(op & 1u) ? (len - 1u - p) : p] & 15u;
                    if (op & 2u) c = comp_code(c);
                }
                v |= c << (28u - 4u * k);
            }
            dst[wi] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Start positions (the reference declares query_batch_start / target_batch_start, gasal.h:89-90, and leaves them NULL,
// res.cpp:27-28): the same banded extension is run BACKWARDS from the end cell, on the reversed prefixes q[0..qend],
// t[0..tend].  reverse_prefix_kernel writes those prefixes (packed, one wave per sequence, at the sequence's own offset
// in a second packed buffer) and their lengths end + 1; starts_kernel turns the backward run's end cell into the start.
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
reverse_prefix_kernel(const uint32_t* __restrict__ packed, uint32_t* __restrict__ rev, const uint32_t* __restrict__ offsets,
                      const int32_t* __restrict__ ends, uint32_t* __restrict__ rev_lens, uint32_t n)
{
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t s = wave; s < n; s += nwaves) {
        const int32_t e = ends[s];
        const uint32_t len = e >= 0 ? (uint32_t)e + 1u : 0u;
        if (lane == 0) rev_lens[s] = len;
        const uint32_t* src = packed + (offsets[s] >> 3);
        uint32_t* dst = rev + (offsets[s] >> 3);
        const uint32_t nw = (len + 7u) >> 3;
        for (uint32_t wi = lane; wi < nw; wi += 64u) {
            uint32_t v = 0;
#pragma unroll
            for (uint32_t k = 0; k < 8u; k++) {
                const uint32_t p = 8u * wi + k;
                uint32_t c = N_VALUE;
                if (p < len) { const uint32_t q = len - 1u - p; c = (src[q >> 3] >> (28u - 4u * (q & 7u))) & 15u; }
                v |= c << (28u - 4u * k);
            }
            dst[wi] = v;
        }
    }
}

__global__ void starts_kernel(const int32_t* __restrict__ qend, const int32_t* __restrict__ tend, const int32_t* __restrict__ bq,
                              const int32_t* __restrict__ bt, int32_t* __restrict__ qstart, int32_t* __restrict__ tstart, uint32_t n)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) { qstart[t] = qend[t] - bq[t]; tstart[t] = tend[t] - bt[t]; }
}

hipError_t launch_reverse_prefix(const uint32_t* packed, uint32_t* rev, const uint32_t* offsets, const int32_t* ends,
                                 uint32_t* rev_lens, uint32_t n, hipStream_t st)
{
    uint32_t blocks = (n + 3u) / 4u;
    if (blocks > 2048u) blocks = 2048u;
    if (blocks < 1u) blocks = 1u;
    hipLaunchKernelGGL(reverse_prefix_kernel, dim3(blocks), dim3(256), 0, st, packed, rev, offsets, ends, rev_lens, n);
    return hipGetLastError();
}

hipError_t launch_starts(const int32_t* qend, const int32_t* tend, const int32_t* bq, const int32_t* bt, int32_t* qstart,
                         int32_t* tstart, uint32_t n, hipStream_t st)
{
    hipLaunchKernelGGL(starts_kernel, dim3((n + 255u) / 256u), dim3(256), 0, st, qend, tend, bq, bt, qstart, tstart, n);
    return hipGetLastError();
}

hipError_t launch_seq_ops(const uint8_t* unpacked, uint32_t* packed, const uint32_t* lens, const uint32_t* offsets,
                          const uint8_t* ops, uint32_t n, hipStream_t st)
{
    uint32_t blocks = (n + 3u) / 4u;
    if (blocks > 2048u) blocks = 2048u;
    if (blocks < 1u) blocks = 1u;
    hipLaunchKernelGGL(seq_ops_kernel, dim3(blocks), dim3(256), 0, st, unpacked, packed, lens, offsets, ops, n);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// host-side launchers (called from capi.cpp through kernels.h)
// ---------------------------------------------------------------------------------------------------
// which: 0 = profile kernel for candidate `kid`, 1 = profile kernel that also takes kind 2, 2 = compare kernel
template <int G, int S>
static hipError_t launch_align_t(const AlignLaunch& L, int which, int kid, hipStream_t st)
{
    // enough groups for every pair, capped by what the chip can keep resident (2 waves/SIMD = 8 waves/CU)
    const int groups_per_block = (256 / 64) * (64 / G);
    int blocks = (L.n + groups_per_block - 1) / groups_per_block;
    int max_blocks = L.num_cus * (S <= 3 ? 2 : 1);
    if (L.max_blocks_override > 0) max_blocks = L.max_blocks_override;
    if (blocks > max_blocks) blocks = max_blocks;
    if (blocks < 1) blocks = 1;
    if (which == 2) hipLaunchKernelGGL((align_kernel<G, S, true>), dim3(blocks), dim3(256), 0, st, L.self_dev, L.p, -1, 0);
    else hipLaunchKernelGGL((align_kernel<G, S, false>), dim3(blocks), dim3(256), 0, st, L.self_dev, L.p, kid, which);
    return hipGetLastError();
}

struct Cfg { int G, S; hipError_t (*fn)(const AlignLaunch&, int, int, hipStream_t); };
static const Cfg kCfgs[] = {       // ascending G*S
    {16, 1, launch_align_t<16, 1>}, {16, 2, launch_align_t<16, 2>}, {16, 3, launch_align_t<16, 3>},
    {32, 2, launch_align_t<32, 2>}, {64, 1, launch_align_t<64, 1>}, {32, 3, launch_align_t<32, 3>},
    {64, 2, launch_align_t<64, 2>}, {64, 3, launch_align_t<64, 3>}, {64, 4, launch_align_t<64, 4>},
    {64, 6, launch_align_t<64, 6>},
};

int max_window_blocks() { return 64 * 6; }

// bits of the packed-maximum key that hold the relative column for the (G, S) chosen for `window_blocks`
int key_bits_for_window(int window_blocks)
{
    for (const Cfg& c : kCfgs)
        if (c.G * c.S >= window_blocks) { int k = 7; while ((1 << k) < 8 * (c.G * c.S + 2)) k++; return k; }
    return -1;
}

// Writes the launch record, resets the queue heads and picks the kernel for the plain pairs: the candidate with the
// smallest max(longest pair alone, whole batch spread over the candidate's lane groups).  A batch with a few very
// long pairs is bound by their latency (fewer blocks per lane and step win), a uniform one by throughput.
__global__ void record_kernel(AlignLaunch L, AlignLaunch* rec)
{
    *rec = L; L.queue[0] = 0u; L.queue[1] = 0u; L.queue[2] = 0u; L.queue[3] = 0u;
    const float total = L.totals[0], longest = L.totals[1];
    int best = 0; float bestc = 3.4e38f;
    for (int c = 0; c < L.ncand; c++) {
        const float cost = fmaxf(longest * L.cand[c].t_lat, total * L.cand[c].t_load / (float)L.cand[c].capacity);
        if (cost < bestc) { bestc = cost; best = c; }
    }
    *L.choice = (L.force_choice >= 0 && L.force_choice < L.ncand) ? L.force_choice : best;
}

hipError_t launch_record(const AlignLaunch& L, AlignLaunch* rec, hipStream_t st)
{
    hipLaunchKernelGGL(record_kernel, dim3(1), dim3(1), 0, st, L, rec);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Preemptive static schedule for the packed-int16 throughput shape (the reference's subwarp rejoining,
// agatha_kernel.h:365-408, re-derived for independent lane groups: instead of idle lanes joining a pair in flight, pairs
// in flight move to lane groups that would otherwise idle).  With n pairs on m < n lane groups and every pair's step count
// p_j known from its lengths, McNaughton's wrap-around rule gives the optimal preemptive makespan T = max(p_max,
// ceil(sum p_j / m)): the pairs are laid end to end (sorted order) on a line, lane group s owns [s T, (s+1) T).  A pair
// that crosses a boundary b T is split: group b runs its FIRST steps at the start of its life and suspends it (state ->
// HBM), group b - 1 resumes it at the end of its own; p_j <= T keeps the two parts apart in time.  This kernel computes the
// step counts (0 for pairs the int16 kernel skips), their prefix sums and T, and decides whether the schedule is used.
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024)
schedule_kernel(AlignLaunch L, int GS, int G)
{
    __shared__ uint32_t part[1024];
    __shared__ uint32_t pmaxs[1024];
    __shared__ uint32_t nzs[1024];
    const int n = L.n, m = L.mig_slots, t = threadIdx.x;
    if (!L.mig_enabled || m <= 0 || n <= m || ((long long)n > 16ll * m && L.mig_enabled != 2)) { if (t == 0) L.sched[0] = 0; return; }
    const int W = (L.p.band_width + 7) >> 3, sw = L.p.slice_width;
    const int chunk = (n + 1023) / 1024;
    const int j0 = t * chunk, j1 = min(n, j0 + chunk);
    auto steps_of = [&](int j) -> uint32_t {
        const uint32_t pair = L.order[j];
        if (L.exotic[pair] != 0) return 0u;                 // another kernel's pair: skipped when drawn
        const int Q = (int)L.qlens[pair], R = (int)L.tlens[pair];
        const int pql = (Q + 7) >> 3, prl = (R + 7) >> 3;
        if (Q <= 0 || R <= 0 || min(W + 1, min(pql, prl)) > GS || pql + GS >= 32760 || prl + GS >= 32760) return 1u;
        const int total = pql + prl - 1;
        // dry step + whole slices + the final check step + what starting it costs (every start stalls all 64 / G groups of the wave)
        return (uint32_t)(((total + sw - 1) / sw) * sw + 2 + kMigPairOverheadSteps * (64 / G));
    };
    uint32_t sum = 0, mx = 0, nz = 0;
    for (int j = j0; j < j1; j++) { const uint32_t p = steps_of(j); sum += p; mx = max(mx, p); nz += p > 1u; }
    part[t] = sum; pmaxs[t] = mx; nzs[t] = nz;
    __syncthreads();
    if (t == 0) {
        unsigned long long acc = 0;          // (the 32-bit prefix sums are only used when the total stays below 2^30)
        uint32_t pm = 0, npairs = 0;
        for (int k = 0; k < 1024; k++) { const uint32_t v = part[k]; part[k] = (uint32_t)acc; acc += v; pm = max(pm, pmaxs[k]); npairs += nzs[k]; }
        const long long P = (long long)acc;
        long long T = (P + m - 1) / m; if (T < (long long)pm) T = pm; if (T < 1) T = 1;
        L.cum[n] = (uint32_t)acc;
        // When is the static schedule the better one (measured): always up to ~2 rounds of pairs (the work queue
        // then ends in a long tail on a few waves); up to 16 rounds when the pairs are long against the band -- the four lane
        // groups of a wave then start their pairs at different times, and every pair start is ~W steps on slower code paths,
        // which short pairs (a bundled-dataset-like batch: 750 steps each) pay for more than they gain (24.7 against 21.1 ms);
        // beyond that the queue balances by itself (70 000 pairs: 223.7 against 227.2 ms) and also follows pairs that z-drop early.
        const long long pavg = npairs ? P / (long long)npairs : 0;
        const bool use = pm > 0 && P < (1ll << 30) && (L.mig_enabled == 2 || 10ll * n <= 22ll * m || pavg >= 12ll * (W + 1));
        L.sched[0] = use ? 1 : 0; L.sched[1] = (int)T; L.sched[2] = (int)((P + T - 1) / T);
    }
    __syncthreads();
    uint32_t acc = part[t];
    for (int j = j0; j < j1; j++) { L.cum[j] = acc; acc += steps_of(j); }
}

hipError_t launch_schedule(const AlignLaunch& L, hipStream_t st)
{
    if (!L.mig_enabled) return hipSuccess;
    const KernelChoice& k = L.cand[0];
    hipLaunchKernelGGL(schedule_kernel, dim3(1), dim3(1024), 0, st, L, k.G * k.S, k.G);
    return hipGetLastError();
}

hipError_t launch_sort(const uint32_t* qlens, const uint32_t* tlens, int n, uint32_t* hist, uint32_t nbuckets,
                       uint32_t* order, float* totals, hipStream_t st)
{
    hipError_t e = hipMemsetAsync(hist, 0, sizeof(uint32_t) * nbuckets, st);
    if (e != hipSuccess) return e;
    const int blocks = (n + 255) / 256;
    hipLaunchKernelGGL(sort_hist_kernel, dim3(blocks), dim3(256), 0, st, qlens, tlens, n, hist, nbuckets);
    hipLaunchKernelGGL(sort_scan_kernel, dim3(1), dim3(256), 0, st, hist, nbuckets, totals);
    hipLaunchKernelGGL(sort_scatter_kernel, dim3(blocks), dim3(256), 0, st, qlens, tlens, n, hist, nbuckets, order);
    return hipGetLastError();
}

// Throughput shape: the smallest G*S that holds the window (most pairs per wave, fullest lanes).  Latency shape: 64 lanes
// per pair with fewer slots, so that a lane sweeps fewer blocks per step (results do not depend on the shape).
static void pick_shapes(int window_blocks, const Cfg** thr, const Cfg** lat)
{
    *thr = nullptr; *lat = nullptr;
    for (const Cfg& c : kCfgs)
        if (c.G * c.S >= window_blocks) { *thr = &c; break; }
    if (!*thr) return;
    for (const Cfg& c : kCfgs)
        if (c.G == 64 && c.G * c.S >= window_blocks && c.S < (*thr)->S) { *lat = &c; break; }
}

// Step-time model of the candidates (microseconds per step of one pair, measured on MI355X with
// tools/bench_candidates.py: profiles/r01_v5/candidates.txt): a lane sweeps S blocks (P register pairs) per step; two
// waves share a SIMD when the chip is full (S <= 3: 2 waves/SIMD resident).
static KernelChoice int32_choice(const Cfg& c, int num_cus)
{
    KernelChoice k;
    k.kind = 0; k.G = c.G; k.S = c.S;
    k.t_lat = 2.3f * c.S + 0.2f; k.t_load = (c.S <= 3 ? 3.3f : 2.3f) * c.S + 0.2f;
    k.capacity = num_cus * (c.S <= 3 ? 8 : 4) * (64 / c.G);
    return k;
}

hipError_t plan_align(AlignLaunch& L, int window_blocks, bool disable16, bool force16)
{
    const Cfg *thr, *lat;
    pick_shapes(window_blocks, &thr, &lat);
    if (!thr) return hipErrorInvalidValue;
    L.ncand = 0;
    int G16 = 0, P16 = 0, GL16 = 0, PL16 = 0;
    const bool have16 = !disable16 && !L.force_cmp && align16_config(L.p, window_blocks, &G16, &P16, &GL16, &PL16);
    if (have16) {
        KernelChoice k;
        k.kind = 1; k.G = G16; k.S = 2 * P16;
        // (microseconds per step, round 2: a lone wave 7.2 / 6.7 / 3.6 for P = 3 / 2 / 1; two waves per SIMD taking turns at
        //  the issue priority 10.3 / 7.8 / ~5)
        k.t_lat = 1.8f * P16 + 1.8f; k.t_load = 2.55f * P16 + 2.6f;
        k.capacity = L.num_cus * 8 * (64 / G16);
        L.cand[L.ncand++] = k;
        if (GL16 && !force16) {
            k.G = GL16; k.S = 2 * PL16;
            k.t_lat = 1.8f * PL16 + 1.8f + (GL16 == 128 ? 0.4f : 0.f); k.t_load = 2.55f * PL16 + 2.6f;
            k.capacity = GL16 == 128 ? L.num_cus * 4 : L.num_cus * 8 * (64 / GL16);
            L.cand[L.ncand++] = k;
        }
    }
    if (!(have16 && force16)) {
        L.cand[L.ncand++] = int32_choice(*thr, L.num_cus);
        if (lat) L.cand[L.ncand++] = int32_choice(*lat, L.num_cus);
    }
    return hipSuccess;
}

hipError_t launch_align(const AlignLaunch& L, int window_blocks, int* G_out, int* S_out, hipStream_t st)
{
    const Cfg *thr, *lat;
    pick_shapes(window_blocks, &thr, &lat);
    if (!thr) return hipErrorInvalidValue;
    if (G_out) *G_out = thr->G;
    if (S_out) *S_out = thr->S;
    if (L.force_cmp) return thr->fn(L, 2, -1, st);
    // candidates for the plain pairs in record order; the throughput shape also takes the kind-2 pairs, so it always runs
    // (after the int16 kernel, which produces them) even when it is not a candidate
    int kid_thr = -2;
    hipError_t e = hipSuccess;
    for (int c = 0; c < L.ncand && e == hipSuccess; c++) {
        const KernelChoice& k = L.cand[c];
        if (k.kind == 1) e = launch_align16(L, k.G, k.S / 2, c, st);
        else if (k.G == thr->G && k.S == thr->S) kid_thr = c;
        else if (lat) e = lat->fn(L, 0, c, st);
    }
    if (e != hipSuccess) return e;
    e = thr->fn(L, 1, kid_thr, st);
    if (e != hipSuccess) return e;
    return thr->fn(L, 2, -1, st);          // compare kernel: walks the queue only if pairs with other letters exist
}

hipError_t launch_pack(const uint8_t* unpacked, uint32_t nbytes, uint32_t* packed, hipStream_t st)
{
    const uint32_t n16 = nbytes / 16, tail = (nbytes % 16) ? 1u : 0u;    // nbytes is a multiple of 8
    uint32_t blocks = (n16 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(pack_kernel, dim3(blocks), dim3(256), 0, st,
                       reinterpret_cast<const uint4*>(unpacked), reinterpret_cast<uint2*>(packed), n16,
                       reinterpret_cast<const uint2*>(unpacked + (size_t)n16 * 16), packed + (size_t)n16 * 2, tail);
    return hipGetLastError();
}

}  // namespace agatha
