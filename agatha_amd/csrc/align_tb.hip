// align_tb.hip -- traceback (alignment paths) for MI355X (gfx950, wave64).
//
// The reference declares gasal_res.cigar / n_cigar_ops (AGAThA/src/gasal.h:91-92) and never fills them (res.cpp:27-28);
// SURVEY.md 8 f4 lists them as the step after the start positions.  Two kernels:
//   * the int32 compare kernel of align_body.inc with TB = true: the same schedule, band, tie-breaks and z-drop as the
//     scoring pass (it IS the scoring kernel), which additionally stores a 4-bit code per computed cell -- 32 bytes per
//     8x8 block, written once, never re-read by the pass: a pure HBM write stream of 0.5 byte per cell;
//   * backtrace_kernel: one wave per pair walks the codes from the end cell to the origin as a scalar program (a chain of
//     dependent loads, one per 8x8 block on the path; the pairs of a batch run side by side) and writes GASAL2-style bytes.
// The CPU statement of both is oracle/agatha_oracle.c: agatha_model_traceback.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <limits.h>

#include "kernels.h"
#include "device_common.h"

namespace agatha {

#include "align_body.inc"

template <int G, int S>
static hipError_t launch_tb_t(const AlignLaunch& L, int pass, hipStream_t st)
{
    const int groups_per_block = (256 / 64) * (64 / G);
    int blocks = (L.n + groups_per_block - 1) / groups_per_block;
    int max_blocks = L.num_cus * (S <= 3 ? 2 : 1);
    if (L.max_blocks_override > 0) max_blocks = L.max_blocks_override;
    if (blocks > max_blocks) blocks = max_blocks;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL((align_kernel<G, S, true, true>), dim3(blocks), dim3(256), 0, st, L.self_dev, L.p, pass, 0);
    return hipGetLastError();
}

struct TbCfg { int G, S; hipError_t (*fn)(const AlignLaunch&, int, hipStream_t); };
static const TbCfg kTbCfgs[] = {       // ascending G*S: the smallest one that holds the window is used
    {16, 3, launch_tb_t<16, 3>}, {32, 3, launch_tb_t<32, 3>}, {64, 3, launch_tb_t<64, 3>}, {64, 6, launch_tb_t<64, 6>},
};

static const TbCfg* tb_cfg(int window_blocks)
{
    for (const TbCfg& c : kTbCfgs)
        if (c.G * c.S >= window_blocks) return &c;
    return nullptr;
}

int tb_group_slots(int window_blocks) { const TbCfg* c = tb_cfg(window_blocks); return c ? c->G * c->S : 0; }

int tb_key_bits(int window_blocks)
{
    const TbCfg* c = tb_cfg(window_blocks);
    if (!c) return -1;
    int k = 7;
    while ((1 << k) < 8 * (c->G * c->S + 2)) k++;
    return k;
}

hipError_t launch_align_tb(const AlignLaunch& L, int window_blocks, int pass, hipStream_t st)
{
    const TbCfg* c = tb_cfg(window_blocks);
    if (!c || !L.force_cmp || !L.tb_codes) return hipErrorInvalidValue;
    return c->fn(L, pass, st);
}

// ---------------------------------------------------------------------------------------------------
// The plan: every pair's code area is (row blocks + column blocks) * G*S * 8 words, from its TRUE lengths; the pairs are
// packed into the area in input order, and when the next one does not fit the area is started over -- a new pass.  One
// wave: 64 sizes per coalesced load, then a scalar walk over the 64 lanes (the running offset depends on every pair before).
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
tb_plan_kernel(AlignLaunch L, int GS, unsigned long long cap_words, int max_passes, unsigned long long* __restrict__ off,
               int* __restrict__ pass, int* __restrict__ plan)
{
    const int lane = threadIdx.x;
    unsigned long long cur = 0ull;
    int p = 0;
    for (int base = 0; base < L.n; base += 64) {
        const int k = base + lane;
        unsigned long long size = 0ull;
        if (k < L.n) {
            const unsigned long long Q = L.qlens[k], R = L.tlens[k];
            if (Q > 0 && R > 0) size = (((Q + 7) >> 3) + ((R + 7) >> 3)) * (unsigned long long)(GS * 8);
        }
        unsigned long long my_off = 0ull;
        int my_pass = 0;
        const int cnt = (L.n - base < 64) ? L.n - base : 64;
        for (int l = 0; l < cnt; l++) {
            const unsigned long long s = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(size >> 32), l) << 32) |
                                         (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)size, l);
            int ps;
            unsigned long long os = 0ull;
            if (s > cap_words) ps = -1;                       // does not fit even alone
            else {
                if (cur + s > cap_words) { p++; cur = 0ull; }
                ps = p < max_passes ? p : -1;                 // (more passes than the host launches: the length hints were wrong)
                os = cur; cur += s;
            }
            if (l == lane) { my_off = os; my_pass = ps; }
        }
        if (k < L.n) {
            off[k] = my_off; pass[k] = my_pass;
            if (my_pass < 0) { L.score[k] = INT_MIN; L.qend[k] = -1; L.tend[k] = -1; }       // AGATHA_AMD_BAD_RESULT
        }
    }
    if (lane == 0) plan[0] = p + 1 < max_passes ? p + 1 : max_passes;
}

hipError_t launch_tb_plan(const AlignLaunch& L, int group_slots, unsigned long long cap_words, int max_passes,
                          unsigned long long* off, int* pass, int* plan, hipStream_t st)
{
    hipLaunchKernelGGL(tb_plan_kernel, dim3(1), dim3(64), 0, st, L, group_slots, cap_words, max_passes, off, pass, plan);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// The walk.  States: 0 = in H of cell (i, j); 1 = in E(i, j) (target base j against a gap; E(i, j) was formed in cell
// (i, j-1)); 2 = in F(i, j) (query base i against a gap; formed in cell (i-1, j)).  Gaps open from the DIAGONAL TERM of a
// cell, not from its H, so leaving a gap always consumes the cell's diagonal.  Whatever is left of one sequence when the
// other is used up lies against the boundary gap (H(i, -1) / H(-1, j), agatha_kernel.h:126-148).
// A score that came through a cell the block-granular band skipped (the stale-register reads of agatha_kernel.h:33-35,
// SURVEY.md App. B) belongs to no alignment: the walk meets a cell without a code and the pair gets
// n_ops = 0xFFFFFFFF (AGATHA_AMD_NO_PATH), as does a pair whose result is AGATHA_AMD_BAD_RESULT.
// Output bytes: (count << 2) | op, op 0 = match, 1 = mismatch, 2 = D, 3 = I, count <= 63, longer runs split greedily from
// the start; first byte = first column of the alignment.  A pair with score 0 has an empty alignment: 0 bytes.
// ---------------------------------------------------------------------------------------------------
// One WAVE per pair, and the walk is SCALAR: every quantity of it (cell, state, run length) is the same in all 64 lanes, so
// it lives in SGPRs and costs scalar instructions; the lanes are only used as storage and memory ports -- lanes 0..7 hold the
// eight code words of the block the path is in (one coalesced 32-byte load when the path enters a block, v_readlane to pick
// the row), lane 0 writes the bytes, all lanes turn the byte string round at the end.  (History: one pair per LANE, 24 ms for
// 2 000 pairs of 10 kb -- the 64 walks of a wave are never in the same state or block, so every lane paid for every branch
// and every load; one pair per wave with lane 0 walking in vector registers, 9 ms.)
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint32_t uniu(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

__global__ void __launch_bounds__(64)
backtrace_kernel(AlignLaunch L, int GS, int pass, uint8_t* __restrict__ cigar, uint32_t* __restrict__ n_ops)
{
    const int pair = blockIdx.x;
    const int lane = threadIdx.x;
    if (pair >= L.n) return;
    {
        // this launch walks the pairs whose codes the pass just recorded (the first one also marks the pairs without codes)
        const int pp = uni(L.tb_pass[pair]);
        if (pp != pass && !(pass == 0 && pp < 0)) return;
    }
    const int score = uni(L.score[pair]);
    const int Q = uni((int)L.qlens[pair]), R = uni((int)L.tlens[pair]);
    if (score == INT_MIN) { if (lane == 0) n_ops[pair] = 0xFFFFFFFFu; return; }
    if (score <= 0 || Q <= 0 || R <= 0) { if (lane == 0) n_ops[pair] = 0u; return; }
    const uint32_t qo = uniu(L.qoffs[pair]), to = uniu(L.toffs[pair]);
    const uint32_t* pq = L.packed_q + (qo >> 3);
    const uint32_t* pt = L.packed_t + (to >> 3);
    const uint32_t* tb = L.tb_codes + L.tb_off[pair];
    uint8_t* out = cigar + (size_t)qo + (size_t)to;
    const int w = L.p.band_width, sw = L.p.slice_width, W = (w + 7) >> 3;
    const int pql = (Q + 7) >> 3, prl = (R + 7) >> 3;

    // Lanes 0..7 hold the eight code words of the block the path is in; lanes 8..31 those of the three blocks it can go to from there
    // -- above, left, above-left --, requested when the path ENTERS a block: the load of the next block is then under way while
    // this one is walked (the walk is a chain of dependent loads otherwise: measured 16 ms for 6 000 pairs of 10 kb, 4 us per
    // block).  A block the pass never computed is all zero.
    int cq = -1, cr = -1;
    uint32_t cur = 0u, nxt = 0u;               // lane l < 8: word of row l of the current block; lane 8 g + l, g = 1..3: of neighbour g
    bool cedge = false;                        // the current block is a boundary block of the band (agatha_kernel.h:243): the band is tested per cell
    const int lg = lane >> 3, ll = lane & 7;
    const int ldq = (lg == 1 || lg == 3) ? 1 : 0, ldr = (lg == 2 || lg == 3) ? 1 : 0;
    // (per lane) word ll of block (q, r), 0 if the scoring pass never computed the block
    auto block_word = [&](int q, int r) -> uint32_t {
        if (q < 0 || r < 0 || q >= pql || r >= prl) return 0u;
        const int step = q + r;
        const int cs = imax(0, r - W), ce = imin(pql - 1, r + W);
        const int i0 = (step / sw) * sw;                              // first step of the slice (agatha_kernel.h:183-187)
        const int ss = imax(imax(0, i0 - pql + 1), ((i0 * 8 + 8 - w) / 2) / 8);
        const int se = imin(imin(prl - 1, i0 + sw - 1), (((i0 + sw - 1) * 8 + 7 + w) / 2) / 8);
        if (q < cs || q > ce || r < ss || r > se) return 0u;
        if (L.tb_lanes == 0) return tb[((size_t)step * GS + (size_t)(r % GS)) * 8 + (size_t)ll];
        const int slot = r % GS, s16 = GS / L.tb_lanes, k16 = slot / s16, rem = slot % s16;
        return tb[(size_t)step * GS * 8 + (size_t)(((((rem >> 1) * 4 + (rem & 1) * 2 + (ll >> 2)) * L.tb_lanes + k16) * 4) + (ll & 3))];
    };
    int wqi = -1, wti = -1;
    uint32_t wq = 0u, wt = 0u;
    // code of cell (i, j), 0 if the scoring pass never computed it
    auto code_of = [&](int i, int j) -> uint32_t {
        const int q = i >> 3, r = j >> 3;
        if (q != cq || r != cr) {
            int hit = 0;
            if (cq >= 0) hit = (q == cq - 1 && r == cr) ? 1 : (q == cq && r == cr - 1) ? 2 : (q == cq - 1 && r == cr - 1) ? 3 : 0;
            cq = q; cr = r;
            if (hit) cur = (uint32_t)__shfl((int)nxt, ll + 8 * hit);
            else cur = (lane < 8) ? block_word(q, r) : 0u;
            nxt = (lane >= 8 && lane < 32) ? block_word(q - ldq, r - ldr) : 0u;
            cedge = (q == imax(0, r - W)) || (q == imin(pql - 1, r + W));
        }
        // (the int32 kernel leaves 0 in the cells of a boundary block that lie outside the band; the int16 kernel computes -- and
        //  codes -- every cell of a block it computes, so the band is looked at here)
        if (cedge && (j - i > w || i - j > w)) return 0u;
        const uint32_t word = (uint32_t)__builtin_amdgcn_readlane((int)cur, i & 7);
        return (word >> (4 * (j & 7))) & 15u;
    };
    auto diag_op = [&](int i, int j) -> uint32_t {
        if ((i >> 3) != wqi) { wqi = i >> 3; wq = uniu(pq[wqi]); }
        if ((j >> 3) != wti) { wti = j >> 3; wt = uniu(pt[wti]); }
        const uint32_t a = (wq >> (28 - 4 * (i & 7))) & 15u, b = (wt >> (28 - 4 * (j & 7))) & 15u;
        return (a == b && a != N_VALUE) ? 0u : 1u;
    };
    // Runs are produced end to start.  A run longer than 63 leaves as its remainder FIRST and its full bytes after it, so that
    // the string turned round reads greedily from the start (63, 63, ..., remainder) like the oracle's.
    uint32_t nb = 0, run_op = 4u, run = 0;
    auto flush = [&]() {
        uint32_t rem = run % 63u, full = run / 63u;
        if (rem) { if (lane == 0) out[nb] = (uint8_t)((rem << 2) | run_op); nb++; }
        for (; full; full--) { if (lane == 0) out[nb] = (uint8_t)((63u << 2) | run_op); nb++; }
    };
    auto emit = [&](uint32_t op) {
        if (op == run_op) { run++; return; }
        if (run) flush();
        run_op = op; run = 1;
    };

    int i = uni(L.qend[pair]), j = uni(L.tend[pair]), state = 0;
    bool bad = false;
    for (int guard = Q + R + 16; i >= 0 && j >= 0 && guard > 0; guard--) {
        uint32_t code = code_of(i, j);
        if (code == 0u) { bad = true; break; }
        if (state == 0) {
            const uint32_t d = code & 3u;
            if (d == 1u) {
                // Most of a path is diagonal moves, and the scalar unit -- one per CU, one instruction per cycle for all of its
                // waves -- is what the walk runs on: the diagonal from (i, j) to the edge of the block is looked at by the lanes
                // at once (lane l: the cell of row l on it), two ballots say how far the path follows it and where bases match,
                // and the scalar program only sees the ends of the runs.
                const int il = i & 7, jl = j & 7, m = il < jl ? il : jl;
                if (cq != wqi) { wqi = cq; wq = uniu(pq[wqi]); }
                if (cr != wti) { wti = cr; wt = uniu(pt[wti]); }
                const int t = il - lane, c = jl - t;                                  // (per lane) steps back along the diagonal, column in the block
                const bool valid = t >= 0 && t <= m;
                uint32_t nibv = valid ? ((cur >> (4 * (c & 7))) & 15u) : 0u;
                if (cedge) { const int row = 8 * cq + lane, col = 8 * cr + c; if (col - row > w || row - col > w) nibv = 0u; }
                const uint32_t av = (wq >> (28 - 4 * (lane & 7))) & 15u, bv = (wt >> (28 - 4 * (c & 7))) & 15u;
                const uint32_t dm = (uint32_t)__builtin_amdgcn_ballot_w64(valid && (nibv & 3u) == 1u);
                const uint32_t mm = (uint32_t)__builtin_amdgcn_ballot_w64(valid && av == bv && av != N_VALUE);
                const uint32_t stop = ~dm & ((2u << il) - 1u);                       // lanes at or below il whose cell leaves the diagonal (or is none)
                const int n = stop ? il - (31 - __builtin_clz(stop)) : il + 1;       // >= 1: the cell (i, j) itself is on it
                const int pend = il - n;
                const uint32_t lowm = pend >= 0 ? ((2u << pend) - 1u) : 0u;
                for (int p = il; p > pend;) {
                    const uint32_t bit = (mm >> p) & 1u;
                    const uint32_t z = (bit ? ~mm : mm) & ((2u << p) - 1u) & ~lowm;   // lanes in (pend, p] with the other op
                    const int k = z ? p - (31 - __builtin_clz(z)) : p - pend;
                    const uint32_t op = bit ? 0u : 1u;
                    if (op == run_op) run += (uint32_t)k;
                    else { if (run) flush(); run_op = op; run = (uint32_t)k; }
                    p -= k;
                }
                i -= n; j -= n; guard -= n - 1;
            }
            else state = (int)d - 1;
        } else if (state == 1) {
            emit(2u); j--;
            if (j < 0) break;
            code = code_of(i, j);
            if (code == 0u) { bad = true; break; }
            if (!(code & 4u)) { emit(diag_op(i, j)); i--; j--; state = 0; }
        } else {
            emit(3u); i--;
            if (i < 0) break;
            code = code_of(i, j);
            if (code == 0u) { bad = true; break; }
            if (!(code & 8u)) { emit(diag_op(i, j)); i--; j--; state = 0; }
        }
    }
    if (bad || (i >= 0 && j >= 0)) { if (lane == 0) n_ops[pair] = 0xFFFFFFFFu; return; }
    for (; i >= 0; i--) emit(3u);
    for (; j >= 0; j--) emit(2u);
    if (run) flush();
    // turn the string round (lane 0's byte stores first, then all lanes swap)
    __threadfence();
    for (uint32_t a = (uint32_t)lane; 2u * a + 1u < nb; a += 64u) {
        const uint32_t b = nb - 1u - a;
        const uint8_t x = out[a], y = out[b];
        out[a] = y; out[b] = x;
    }
    if (lane == 0) n_ops[pair] = nb;
}

hipError_t launch_backtrace(const AlignLaunch& L, int group_slots, int pass, uint8_t* cigar, uint32_t* n_ops, hipStream_t st)
{
    hipLaunchKernelGGL(backtrace_kernel, dim3(L.n), dim3(64), 0, st, L, group_slots, pass, cigar, n_ops);
    return hipGetLastError();
}

}  // namespace agatha
