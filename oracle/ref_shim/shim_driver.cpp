// shim_driver.cpp -- runs the reference agatha_kernel (included from /root/reference at build time,
// with one cooperative yield inserted by the Makefile) warp by warp on the CPU.
// TEST INFRASTRUCTURE, build-container only; see cuda_shim.h.
#include "cuda_shim.h"
#include <ucontext.h>
#include <vector>
#include <cstdio>
#include <cstdlib>
#include <cstring>

shim_dim blockIdx, blockDim, gridDim;
int32_t shared_maxHH[1 << 20];

#include "agatha_kernel_yield.h"   // generated: the reference kernel + SHIM_YIELD() in its y-loop

namespace {
enum Kind { K_NONE, K_SYNC, K_MATCH, K_REDUCE };
struct Lane {
    ucontext_t ctx; std::vector<char> stack;
    bool done = false, waiting = false;
    Kind kind = K_NONE; unsigned mask = 0; int value = 0; unsigned uret = 0; int iret = 0;
};
Lane lanes[32];
ucontext_t sched_ctx;
int cur = 0, cur_warp_in_block = 0;
struct KArgs { uint32_t *pq, *pr, *ql, *tl, *qo, *to; gasal_res_t *res; int n; uint32_t L; short2 *gb; } kargs;

unsigned live_mask() { unsigned m = 0; for (int l = 0; l < 32; l++) if (!lanes[l].done) m |= 1u << l; return m; }
void block_on(Kind k, unsigned mask, int v) {
    Lane &L = lanes[cur]; L.waiting = true; L.kind = k; L.mask = mask; L.value = v;
    swapcontext(&L.ctx, &sched_ctx);
}
void lane_entry() {
    agatha_kernel(kargs.pq, kargs.pr, kargs.ql, kargs.tl, kargs.qo, kargs.to, kargs.res, nullptr, nullptr,
                  kargs.n, kargs.L, kargs.gb);
    lanes[cur].done = true;
    swapcontext(&lanes[cur].ctx, &sched_ctx);
}
// release every group whose live members all wait on the same primitive with the same mask
bool release_groups() {
    bool any = false;
    for (int l = 0; l < 32; l++) {
        Lane &A = lanes[l];
        if (A.done || !A.waiting) continue;
        unsigned M = A.mask; bool ready = true;
        for (int j = 0; j < 32 && ready; j++) if (M >> j & 1) {
            Lane &B = lanes[j];
            if (B.done) continue;
            if (!(B.waiting && B.kind == A.kind && B.mask == M)) ready = false;
        }
        if (!ready) continue;
        int mx = INT_MIN;
        for (int j = 0; j < 32; j++) if ((M >> j & 1) && !lanes[j].done) mx = std::max(mx, lanes[j].value);
        for (int j = 0; j < 32; j++) if ((M >> j & 1) && !lanes[j].done) {
            Lane &B = lanes[j];
            if (A.kind == K_MATCH) {
                unsigned r = 0;
                for (int t = 0; t < 32; t++) if ((M >> t & 1) && !lanes[t].done && lanes[t].value == B.value) r |= 1u << t;
                B.uret = r;
            } else if (A.kind == K_REDUCE) B.iret = mx;
        }
        for (int j = 0; j < 32; j++) if ((M >> j & 1) && !lanes[j].done) lanes[j].waiting = false;
        any = true;
    }
    return any;
}
int run_warp() {
    for (int l = 0; l < 32; l++) {
        Lane &L = lanes[l]; L.done = false; L.waiting = false;
        if (L.stack.empty()) L.stack.resize(256 * 1024);
        getcontext(&L.ctx); L.ctx.uc_stack.ss_sp = L.stack.data(); L.ctx.uc_stack.ss_size = L.stack.size();
        L.ctx.uc_link = &sched_ctx; makecontext(&L.ctx, lane_entry, 0);
    }
    for (;;) {
        bool progressed = false, alive = false;
        for (int l = 0; l < 32; l++) {
            if (lanes[l].done) continue;
            alive = true;
            if (lanes[l].waiting) continue;
            cur = l; swapcontext(&sched_ctx, &lanes[l].ctx); progressed = true;
        }
        if (!alive) return 0;
        if (release_groups()) progressed = true;
        if (!progressed) return -1;   // deadlock
    }
}
}  // namespace

shim_dim shim_thread_idx() { shim_dim d; d.x = (unsigned)(cur_warp_in_block * 32 + cur); return d; }
void shim_syncwarp() { block_on(K_SYNC, 0xffffffffu, 0); }
unsigned shim_activemask() { return live_mask(); }
unsigned shim_match_any(unsigned mask, int v) { block_on(K_MATCH, mask, v); return lanes[cur].uret; }
int shim_reduce_max(unsigned mask, int v) { block_on(K_REDUCE, mask, v); return lanes[cur].iret; }
void shim_yield() { swapcontext(&lanes[cur].ctx, &sched_ctx); }

struct shim_params { int32_t match, mismatch, gap_open, gap_extend, slice_width, z_threshold, band_width; };

// Batch in the GASAL host wire format (ASCII padded with 'N' to x8, byte offsets, true lengths).
extern "C" int refshim_align(const uint8_t *qb, const uint8_t *tb, const uint32_t *qoff, const uint32_t *toff,
                             const uint32_t *qlen, const uint32_t *tlen, int n, const shim_params *p,
                             int blocks, int threads, int32_t *score, int32_t *qend, int32_t *tend)
{
    if (threads <= 0) threads = 256;
    int subwarps_per_block = threads / 8;
    if (blocks <= 0) blocks = (n + subwarps_per_block - 1) / subwarps_per_block;   // job_per_warp == 1
    // device constants, as gasal_copy_subst_scores (reference gasal_align.cu:295-309)
    _cudaGapO = p->gap_open; _cudaGapExtend = p->gap_extend; _cudaGapOE = p->gap_open + p->gap_extend;
    _cudaMatchScore = p->match; _cudaMismatchScore = p->mismatch; _cudaSliceWidth = p->slice_width;
    _cudaZThreshold = p->z_threshold; _cudaBandWidth = p->band_width;
    // pack (own loop; layout of reference pack_rc_seqs.h:21-33)
    uint32_t qbytes = 0, tbytes = 0, L = 0;
    for (int k = 0; k < n; k++) {
        qbytes = std::max(qbytes, qoff[k] + ((qlen[k] + 7) & ~7u)); tbytes = std::max(tbytes, toff[k] + ((tlen[k] + 7) & ~7u));
        L = std::max(L, std::max(qlen[k], tlen[k]));
    }
    std::vector<uint32_t> pq(qbytes / 8 + 1), pt(tbytes / 8 + 1);
    for (uint32_t w = 0; w < qbytes / 8; w++) { uint32_t v = 0; for (int k = 0; k < 8; k++) v |= (uint32_t)(qb[8 * w + k] & 15) << (28 - 4 * k); pq[w] = v; }
    for (uint32_t w = 0; w < tbytes / 8; w++) { uint32_t v = 0; for (int k = 0; k < 8; k++) v |= (uint32_t)(tb[8 * w + k] & 15) << (28 - 4 * k); pt[w] = v; }
    // scratch layout of reference ctors.cpp:89
    size_t strip = (size_t)L * (threads / 8) * blocks;
    std::vector<short2> gbuf(strip * 3 + n + 8);
    // sort keys: agatha_sort (agatha_kernel.h:434-458) + host std::sort (gasal_align.cu:17)
    short2 *keys = gbuf.data() + strip * 3;
    for (int k = 0; k < n; k++) keys[k] = make_short2((int)((qlen[k] + 7) / 8 + (tlen[k] + 7) / 8 - 1), k);
    std::sort(keys, keys + n, [](short2 a, short2 b) { return a.x < b.x; });
    std::vector<uint32_t> ql(qlen, qlen + n), tl(tlen, tlen + n), qo(qoff, qoff + n), to(toff, toff + n);
    gasal_res_t res; memset(&res, 0, sizeof(res));
    res.aln_score = score; res.query_batch_end = qend; res.target_batch_end = tend;
    kargs = {pq.data(), pt.data(), ql.data(), tl.data(), qo.data(), to.data(), &res, n, L, gbuf.data()};
    blockDim.x = threads; gridDim.x = blocks;
    for (int b = 0; b < blocks; b++) {
        blockIdx.x = b;
        for (int w = 0; w < threads / 32; w++) {
            int first_subwarp = (b * threads + w * 32) / 8;
            if (first_subwarp >= n) continue;           // no job in this warp (job_per_warp == 1)
            cur_warp_in_block = w;
            if (run_warp() != 0) return -1;
        }
    }
    return 0;
}
