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

#include "align_body.inc"

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
// 2-bit codes + N mask -> the 4-bit words the kernels read (round 4: the "2-bit-packed" input of north_star; SURVEY.md 8 f3's
// second half).  Per 8 bases one uint16 of codes (A 0, C 1, G 2, T 3; base k in bits 15-2k..14-2k, first base on top, like the
// 4-bit layout) and one byte of mask (bit 7-k: base k is N -- or padding, or any other letter: the format carries ACGT + N only):
// 3 bits per base over PCIe instead of 4 (isPacked) or 8 (ASCII).  One lane per 16 bases: 32-bit codes word + 16-bit mask in,
// two 4-bit words out, coalesced.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t unpack2_word(uint32_t codes16, uint32_t mask8)
{
    uint32_t v = 0;
#pragma unroll
    for (int k2 = 0; k2 < 8; k2++) {
        const uint32_t c = (codes16 >> (14 - 2 * k2)) & 3u;
        const uint32_t nib = ((mask8 >> (7 - k2)) & 1u) ? 14u : ((0x4731u >> (4 * c)) & 15u);       // A 1, C 3, G 7, T 4 (c & 0xF of the ASCII letters)
        v |= nib << (28 - 4 * k2);
    }
    return v;
}

__global__ void __launch_bounds__(256)
unpack2_kernel(const uint16_t* __restrict__ codes, const uint8_t* __restrict__ nmask, uint32_t nwords, uint32_t* __restrict__ out)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < nwords; t += stride) out[t] = unpack2_word(codes[t], nmask[t]);
}

hipError_t launch_unpack2(const uint16_t* codes, const uint8_t* nmask, uint32_t nwords, uint32_t* packed, hipStream_t st)
{
    uint32_t blocks = (nwords + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(unpack2_kernel, dim3(blocks), dim3(256), 0, st, codes, nmask, nwords, packed);
    return hipGetLastError();
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

// kind of each pair (low 7 bits): 1 = a sequence holds a letter outside {A, C, G, T, N} (compare kernel); 0 = everything else,
// with bit 7 set when the QUERY holds an N (2 and 3 are given later / below)
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
            // (bit 7 on a plain pair: its query -- the DP rows -- holds an N.  Every kernel scores N in line, gasal_kernels.h:48-50;
            //  the packed-int16 kernel needs to know because its score profile has no row for it: align16_body.inc, NROW)
            int kind = all_plain ? (any_qn ? 0x80 : 0) : 1;
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
            if (kind == 1) atomicAdd(kind_counts + 0, 1u);
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
// Round 4, a batch of mixed lengths (BASELINE configs[4]; the purpose of the reference's uneven bucketing, agatha_kernel.h:113,
// and subwarp rejoining, :365-408, re-derived): ONE shape per launch made 19 000 short pairs run on the latency shape because a
// few hundred were long, or the long ones crawl on the throughput shape.  When both int16 shapes are candidates the batch may
// be SPLIT by length: the sorted order's first `n_long` pairs go to the latency shape (candidate 1, launched on a second
// stream, one pair per wave), the rest to the throughput shape (candidate 0) on its work queue, side by side on the chip.  The
// threshold is the bucket of the length histogram that minimises
//     max(longest * t_lat(latency shape), longest short pair * t_load(throughput shape),
//         (steps of the short pairs * t_load(thr) + steps of the long pairs * t_load(lat) * lane groups a wave displaces) / lane groups)
// over all 16 384 buckets (256 threads, 64 buckets each, a suffix scan over the threads); it is taken when that beats the best single
// shape by 7 %.  queue[11] = n_long (0: no split); the static schedule is switched off with it.
__global__ void __launch_bounds__(256)
record_kernel(AlignLaunch L, AlignLaunch* rec, const uint32_t* __restrict__ hist, uint32_t nbuckets, int allow_split)
{
    __shared__ float s_steps[256];
    __shared__ float s_cost[256];
    __shared__ uint32_t s_bucket[256];
    const int t = threadIdx.x;
    const float total = L.totals[0], longest = L.totals[1];
    int best = 0; float bestc = 3.4e38f;
    for (int c = 0; c < L.ncand; c++) {
        // (the longest pair alone: on a shape with several pairs per wave it never is -- its wave's other lane groups keep drawing
        //  pairs --, so it advances at the loaded step time there unless it is the whole batch)
        const float tl = (L.cand[c].G < 64 && L.n > L.cand[c].capacity / 2) ? L.cand[c].t_load : L.cand[c].t_lat;
        const float cost = fmaxf(longest * tl, total * L.cand[c].t_load / (float)L.cand[c].capacity);
        if (cost < bestc) { bestc = cost; best = c; }
    }
    // (... and only a batch whose longest pair is more than twice its average one can gain by a split: a uniform batch -- the headline
    //  workload's longest pair is 1.2 x the average -- skips the search over the histogram, 40 us of this one-workgroup kernel)
    const bool can_split = allow_split && hist != nullptr && L.ncand >= 2 && L.cand[0].kind == 1 && L.cand[1].kind == 1 && L.cand[0].G < 64 &&
                           L.force_choice < 0 && L.n > 64 && (L.force_split > 0 || longest * (float)L.n > 2.0f * total);
    uint32_t n_long = 0;
    if (can_split) {          // (uniform: every thread takes the branch or none)
        const uint32_t per = (nbuckets + 255u) / 256u;
        // thread t owns the buckets [t * per, (t + 1) * per); after the scatter hist[b] = pairs in buckets >= b
        const uint32_t b0 = (uint32_t)t * per, b1 = min(b0 + per, nbuckets);
        float mine = 0.f;
        for (uint32_t b = b0; b < b1; b++) {
            const uint32_t c = hist[b] - (b + 1 < nbuckets ? hist[b + 1] : 0u);
            mine += (float)c * (4.f * (float)b + 2.f);
        }
        s_steps[t] = mine;
        __syncthreads();
        if (t == 0) { float acc = 0.f; for (int j = 255; j >= 0; j--) { const float v = s_steps[j]; s_steps[j] = acc; acc += v; } }   // steps in the buckets ABOVE thread j's
        __syncthreads();
        const KernelChoice &A = L.cand[1], &B = L.cand[0];
        const float rf = (float)A.G / (float)B.G, capB = (float)B.capacity;
        float above = s_steps[t];                       // steps of the pairs in buckets > b (walking down from b1 - 1)
        float bc = 3.4e38f; uint32_t bb = 0;
        for (uint32_t b = b1; b-- > b0;) {
            // threshold: buckets >= b + 1 are long (steps `above`), bucket b holds the longest short pair
            const uint32_t cnt = hist[b] - (b + 1 < nbuckets ? hist[b + 1] : 0u);
            const uint32_t nl = b + 1 < nbuckets ? hist[b + 1] : 0u;
            // (every split has one bucket that holds its longest short pair: only those are looked at)
            // ... and the long pairs must leave the chip to the others: a pair on the latency shape holds a whole wave slot for its
            // life, and the throughput shape's workgroups cannot start on a CU whose slots are taken (C4 at 20 000 pairs with a quarter
            // of the steps on 1 900 waves: 204 ms against 161 on the latency shape alone)
            if (cnt > 0u && nl > 0u && nl < (uint32_t)L.n && 4u * nl <= (uint32_t)A.capacity) {
                const float lshort = 4.f * (float)b + 2.f;
                const float c = fmaxf(fmaxf(longest * A.t_lat, lshort * B.t_load), ((total - above) * B.t_load + above * A.t_load * rf) / capB);
                if (c < bc) { bc = c; bb = b + 1; }
            }
            above += (float)cnt * (4.f * (float)b + 2.f);
        }
        s_cost[t] = bc; s_bucket[t] = bb;
        __syncthreads();
        if (t == 0) {
            float c = 3.4e38f; uint32_t b = 0;
            for (int j = 0; j < 256; j++) if (s_cost[j] < c) { c = s_cost[j]; b = s_bucket[j]; }
            if (c < 0.93f * bestc && b > 0u && b < nbuckets) n_long = hist[b];
            if (L.force_split > 0) n_long = (uint32_t)min(L.force_split, L.n - 1);
        }
    }
    if (t == 0) {
        *rec = L; L.queue[0] = 0u; L.queue[1] = 0u; L.queue[2] = 0u; L.queue[3] = 0u; L.queue[4] = 0u; L.queue[5] = 0u; L.queue[6] = 0u; L.queue[7] = 0u;
        L.queue[11] = n_long;
        if (n_long) { best = 0; L.sched[0] = 0; }
        *L.choice = (L.force_choice >= 0 && L.force_choice < L.ncand) ? L.force_choice : best;
    }
}

// A batch split by length: the throughput shape's 512 workgroups fill every CU (two of them take 159 of its 160 KB of LDS), and a
// workgroup of the latency shape that is dispatched a microsecond later finds no room until one of them ends -- the 64 long pairs
// of a 40 000-pair batch then ran in four rounds on the handful of workgroups that had got in, 157 ms instead of 65.  So the
// throughput shape's launch waits, on its own stream, behind this one-wave kernel: until every long pair has been drawn by a
// resident wave of the latency shape (queue[5] counts them), or 300 us have passed.
__global__ void split_gate_kernel(AlignLaunch L)
{
    const unsigned int n_long = L.queue[11];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (n_long == 0u) {
        // No split.  If the throughput shape takes the batch, its grid -- every workgroup resident, two per CU, the static schedule
        // counts on it -- must not be dispatched among the latency shape's workgroups that are still on their way out (measured: one
        // launch in two took 30.4 instead of 26.2 ms): wait for the last of them (queue[7]), at most 100 us.
        if (*L.choice != 0) return;
        while (__hip_atomic_load(L.queue + 7, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u && __builtin_amdgcn_s_memrealtime() - t0 < 10000ull)
            __builtin_amdgcn_s_sleep(8);
        // (what the gate cost this call, in 10 ns ticks: agatha_amd_flat_stats out[3] -- on a runtime that does not run the two streams'
        //  kernels side by side it is the whole time-out every time, and a regression there should be visible)
        if (threadIdx.x == 0) L.queue[63] = (unsigned int)(__builtin_amdgcn_s_memrealtime() - t0);
        return;
    }
    while (__hip_atomic_load(L.queue + 5, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < n_long &&
           __builtin_amdgcn_s_memrealtime() - t0 < 30000ull)
        __builtin_amdgcn_s_sleep(32);
    if (threadIdx.x == 0) L.queue[63] = (unsigned int)(__builtin_amdgcn_s_memrealtime() - t0);
}

hipError_t launch_record(const AlignLaunch& L, AlignLaunch* rec, hipStream_t st, const uint32_t* hist, uint32_t nbuckets, int allow_split)
{
    hipLaunchKernelGGL(record_kernel, dim3(1), dim3(256), 0, st, L, rec, hist, nbuckets, allow_split);
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
    if (!L.mig_enabled || m <= 0 || n <= m || ((long long)n > 16ll * m && L.mig_enabled != 2)) { if (t == 0) { L.sched[0] = 0; L.sched[1] = 0; L.sched[2] = 0; } return; }
    const int W = (L.p.band_width + 7) >> 3, sw = L.p.slice_width;
    const int chunk = (n + 1023) / 1024;
    const int j0 = t * chunk, j1 = min(n, j0 + chunk);
    auto steps_of = [&](int j) -> uint32_t {
        const uint32_t pair = L.order[j];
        if ((L.exotic[pair] & 0x7f) != 0) return 0u;        // another kernel's pair: skipped when drawn
        const int Q = (int)L.qlens[pair], R = (int)L.tlens[pair];
        const int pql = (Q + 7) >> 3, prl = (R + 7) >> 3;
        if (Q <= 0 || R <= 0 || min(W + 1, min(pql, prl)) > GS || pql + GS >= 32760 || prl + GS >= 32760) return 1u;
        const int total = pql + prl - 1;
        // dry step + whole slices + the final check step + what starting it costs (every start stalls all 64 / G groups of the wave)
        uint32_t st = (uint32_t)(((total + sw - 1) / sw) * sw + 2 + kMigPairOverheadSteps * (64 / G));
        return st;
    };
    // (round 5: a pair's step count is looked up once -- three dependent loads -- and parked in cum[j]; the sums over the 1 024 threads are
    //  a parallel scan, block_scan_1024 below: this one-workgroup kernel stands in front of every align call)
    uint32_t sum = 0, mx = 0, nz = 0;
    for (int j = j0; j < j1; j++) { const uint32_t p = steps_of(j); L.cum[j] = p; sum += p; mx = max(mx, p); nz += p > 1u; }
    // exclusive prefix sums of one value per thread (Hillis-Steele over LDS; totals beyond 2^32 are refused below by P < 2^30 --
    // the 64-bit total comes from a second scan of the high parts' carries: sums of 1 024 values below 2^22 fit 32 bits)
    auto block_scan_1024 = [&](uint32_t* buf, uint32_t v) -> uint32_t {
        buf[t] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const uint32_t add = t >= off ? buf[t - off] : 0u;
            __syncthreads();
            buf[t] += add;
            __syncthreads();
        }
        return buf[t] - v;          // exclusive
    };
    // (a thread's sum is at most chunk * 2^16 steps; with n <= 16 m <= 2^18 pairs the total stays below 2^34: carried in two halves)
    const uint32_t ex_lo = block_scan_1024(part, sum & 0xFFFFu), tot_lo = part[1023];
    __syncthreads();
    const uint32_t ex_hi = block_scan_1024(part, sum >> 16), tot_hi = part[1023];
    __syncthreads();
    const unsigned long long my_ex = (unsigned long long)ex_lo + ((unsigned long long)ex_hi << 16);
    const unsigned long long total_steps = (unsigned long long)tot_lo + ((unsigned long long)tot_hi << 16);
    pmaxs[t] = mx; nzs[t] = nz;
    __syncthreads();
    for (int off = 512; off > 0; off >>= 1) { if (t < off) { pmaxs[t] = max(pmaxs[t], pmaxs[t + off]); nzs[t] += nzs[t + off]; } __syncthreads(); }
    part[t] = (uint32_t)my_ex;
    __syncthreads();
    if (t == 0) {
        const unsigned long long acc = total_steps;          // (the 32-bit prefix sums are only used when the total stays below 2^30)
        const uint32_t pm = pmaxs[0], npairs = nzs[0];
        const long long P = (long long)acc;
        long long T = (P + m - 1) / m; if (T < (long long)pm) T = pm; if (T < 1) T = 1;
        L.cum[n] = (uint32_t)acc;
        // When is the static schedule the better one (measured): always up to ~2 rounds of pairs (the work queue
        // then ends in a long tail on a few waves); up to 16 rounds when the pairs are long against the band -- the four lane
        // groups of a wave then start their pairs at different times, and every pair start is ~W steps on slower code paths,
        // which short pairs (a bundled-dataset-like batch: 750 steps each) pay for more than they gain (24.7 against 21.1 ms);
        // beyond that the queue balances by itself (70 000 pairs: 223.7 against 227.2 ms) and also follows pairs that z-drop early.
        const long long pavg = npairs ? P / (long long)npairs : 0;
        // ... and not when the longest pair is several times the average one (a batch of mixed lengths: T is then that pair, most lane
        // groups stand idle for most of it, and a pair that z-drops early frees nobody: C4 at 20 000 pairs 378 ms against 160)
        const bool use = pm > 0 && P < (1ll << 30) && (L.mig_enabled == 2 || ((10ll * n <= 22ll * m || pavg >= 12ll * (W + 1)) && (long long)pm <= 3ll * pavg));
        L.sched[0] = use ? 1 : 0; L.sched[1] = use ? (int)T : 0; L.sched[2] = use ? (int)((P + T - 1) / T) : 0;
    }
    __syncthreads();
    uint32_t acc = part[t];
    for (int j = j0; j < j1; j++) { const uint32_t p = L.cum[j]; L.cum[j] = acc; acc += p; }
    // ---- the pool of suspended pairs' rests, and which physical lane group owns which interval (round 5) ----
    // Boundary b T (b = 1 .. m - 1) cuts the pair that lies across it: lane group (interval) b runs its first steps and suspends it; its
    // REST, the L_b = b T - cum[pair] steps before the boundary, is resumed by whoever draws it from the pool: the boundaries sorted by
    // L_b, longest first (mig_late: [0] the pool's head, [1] how many, [2..] the boundaries).  A lane group is done with the fixed part of
    // its interval after T - L_(g+1) steps, so with every group on time each one finds its own interval's rest on top of the pool --
    // McNaughton's schedule --, and a group that is ahead or behind finds a longer or a shorter one.
    // A wave runs key steps while ANY of its lane groups' pairs is in the window at its end (align16_body.inc, want_keys), so the four
    // lane groups of a wave should end their pairs together: the intervals are dealt to the physical lane groups in the order of their
    // rest L_(g+1) -- the four groups of a wave are then done with their fixed parts within a step or two of each other, draw
    // neighbouring rests of the pool, and end those together --, consecutive waves of that order on one CU (workgroups b and b + half
    // share a CU on a full persistent grid).  Counting sorts over 2048 bins of the step, one workgroup.
    if (L.mig_perm == nullptr) return;
    const bool have_pool = L.mig_late != nullptr;        // (debug option no_pool: the permutation alone)
    __shared__ uint32_t bins[2048];
    __threadfence_block();
    __syncthreads();
    const int T_ = L.sched[1];
    const bool use_ = L.sched[0] != 0 && T_ > 0;
    if (t == 0 && have_pool) { L.mig_late[0] = 0; L.mig_late[1] = 0; }
    if (!use_) { for (int g = t; g < m; g += 1024) L.mig_perm[g] = g; return; }
    // rest of the pair across boundary b (0: the boundary falls between two pairs, or b is no boundary), for all b once: thread t takes
    // the boundaries [t per, (t + 1) per) -- one binary search for the first, a short walk along the line for the others (a boundary lies
    // about one pair behind the one before) -- and leaves them in L.mig_rest[0 .. m]
    {
        const int per = (m + 1 + 1023) / 1024;
        int j = -1;
        for (int b = t * per; b < min(m + 1, (t + 1) * per); b++) {
            uint32_t r = 0u;
            if (b > 0 && b < m) {
                const uint32_t at = (uint32_t)b * (uint32_t)T_;
                if (j < 0) { int lo = 0, hi = n; while (lo < hi) { const int mid = (lo + hi) >> 1; if (L.cum[mid + 1] > at) hi = mid; else lo = mid + 1; } j = lo; }
                else while (j < n && L.cum[j + 1] <= at) j++;
                if (j < n) { const uint32_t c = L.cum[j]; r = c < at ? at - c : 0u; }
            }
            L.mig_rest[b] = r;
        }
    }
    __threadfence_block();
    __syncthreads();
    // descending: bin 0 holds the longest rests
    auto bin_of = [&](uint32_t rest) -> uint32_t { return 2047u - (uint32_t)(((unsigned long long)rest * 2047ull) / (unsigned long long)T_); };
    for (int pass = have_pool ? 0 : 1; pass < 2; pass++) {
        // pass 0: the pool (boundaries with a rest); pass 1: the permutation (every interval g by the rest of boundary g + 1)
        for (int b = t; b < 2048; b += 1024) bins[b] = 0u;
        __syncthreads();
        for (int g = t; g < m; g += 1024) {
            const uint32_t r = L.mig_rest[pass == 0 ? g : g + 1];
            if (pass == 1 || r > 0u) atomicAdd(&bins[bin_of(r)], 1u);
        }
        __syncthreads();
        {   // exclusive prefix sums of the 2 048 bins: two per thread
            const uint32_t b0 = bins[2 * t], b1 = bins[2 * t + 1];
            __syncthreads();
            const uint32_t ex = block_scan_1024(part, b0 + b1);
            bins[2 * t] = ex; bins[2 * t + 1] = ex + b0;
            if (t == 1023 && pass == 0) L.mig_late[1] = (int)(ex + b0 + b1);
        }
        __syncthreads();
        const int gpb = m / (2 * L.num_cus) > 0 ? m / (2 * L.num_cus) : 1, half = L.num_cus;     // lane groups per workgroup; workgroups per half of the grid
        for (int g = t; g < m; g += 1024) {
            const uint32_t r = L.mig_rest[pass == 0 ? g : g + 1];
            if (pass == 0) { if (r > 0u) L.mig_late[2 + atomicAdd(&bins[bin_of(r)], 1u)] = g; continue; }
            if (L.mig_identity || m != 2 * L.num_cus * gpb) { L.mig_perm[g] = g; continue; }
            const uint32_t u = atomicAdd(&bins[bin_of(r)], 1u);       // rank of interval g
            // rank u -> physical lane group: CU u / (2 gpb), its first workgroup for the first gpb ranks, its second for the rest
            const int cu = (int)(u / (uint32_t)(2 * gpb)), within = (int)(u % (uint32_t)(2 * gpb));
            const int blk = within < gpb ? cu : cu + half;
            L.mig_perm[blk * gpb + within % gpb] = g;
        }
        __syncthreads();
    }
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
        // (round 4, value steps without maxima inside the block: a lone wave 2.6 / 5.6 for P = 1 / 3, two waves per SIMD 3.3 / 6.4 / 8.3
        //  for P = 1 / 2 / 3: profiles/r04_v1/other_configs.txt)
        k.t_lat = 1.5f * P16 + 1.1f; k.t_load = 2.45f * P16 + 1.0f;
        k.capacity = L.num_cus * 8 * (64 / G16);
        L.cand[L.ncand++] = k;
        if (GL16 && !force16) {
            k.G = GL16; k.S = 2 * PL16;
            k.t_lat = 1.5f * PL16 + 1.1f + (GL16 == 128 ? 0.6f : 0.f); k.t_load = 2.45f * PL16 + 1.0f;
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

hipError_t launch_align(const AlignLaunch& L, int window_blocks, int* G_out, int* S_out, hipStream_t st, hipStream_t aux, hipEvent_t fork,
                        hipEvent_t join)
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
    // The int16 latency shape (candidate 1 behind an int16 candidate 0) goes to the second stream, FIRST, so that a batch split by
    // length has its long pairs on the chip before the throughput shape's workgroups fill it; the first stream waits for it behind
    // its own int16 kernel.  (When the device chose one shape, the other returns at once wherever it was launched.)
    const bool two_streams = aux != nullptr && fork != nullptr && join != nullptr && L.ncand >= 2 && L.cand[0].kind == 1 && L.cand[1].kind == 1;
    if (two_streams) {
        e = hipEventRecord(fork, st);
        if (e == hipSuccess) e = hipStreamWaitEvent(aux, fork, 0);
        if (e == hipSuccess) e = launch_align16(L, L.cand[1].G, L.cand[1].S / 2, 1, aux);
        if (e == hipSuccess) e = hipEventRecord(join, aux);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(split_gate_kernel, dim3(1), dim3(64), 0, st, L);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    for (int c = 0; c < L.ncand && e == hipSuccess; c++) {
        const KernelChoice& k = L.cand[c];
        if (k.kind == 1) { if (!(two_streams && c == 1)) e = launch_align16(L, k.G, k.S / 2, c, st); }
        else if (k.G == thr->G && k.S == thr->S) kid_thr = c;
        else if (lat) e = lat->fn(L, 0, c, st);
    }
    if (e != hipSuccess) return e;
    if (two_streams) { e = hipStreamWaitEvent(st, join, 0); if (e != hipSuccess) return e; }
    // the clean-up launch of the int16 latency shape (pairs of kind 5; returns at once when there is none)
    if (L.cleanup_ok && L.ncand >= 2 && L.cand[0].kind == 1 && L.cand[1].kind == 1) { e = launch_align16(L, L.cand[1].G, L.cand[1].S / 2, 100, st); if (e != hipSuccess) return e; }
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
