// align16_kernel.hip -- host side of the packed-int16 kernel: which (G, P) shapes exist, when the kernel may run, and the
// dispatch to the instantiations (align16_inst.hip, one translation unit per cut diagonal; the kernel itself is in
// align16_body.inc).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace agatha {


hipError_t align16_entry_0(const AlignLaunch&, int, int, int, hipStream_t);
hipError_t align16_entry_1(const AlignLaunch&, int, int, int, hipStream_t);
hipError_t align16_entry_2(const AlignLaunch&, int, int, int, hipStream_t);
hipError_t align16_entry_3(const AlignLaunch&, int, int, int, hipStream_t);
hipError_t align16_entry_4(const AlignLaunch&, int, int, int, hipStream_t);
hipError_t align16_entry_5(const AlignLaunch&, int, int, int, hipStream_t);
hipError_t align16_entry_6(const AlignLaunch&, int, int, int, hipStream_t);
hipError_t align16_entry_7(const AlignLaunch&, int, int, int, hipStream_t);

hipError_t align16_tb_entry_0(const AlignLaunch&, int, int, int, hipStream_t);
hipError_t align16_tb_entry_1(const AlignLaunch&, int, int, int, hipStream_t);
hipError_t align16_tb_entry_2(const AlignLaunch&, int, int, int, hipStream_t);
hipError_t align16_tb_entry_3(const AlignLaunch&, int, int, int, hipStream_t);
hipError_t align16_tb_entry_4(const AlignLaunch&, int, int, int, hipStream_t);
hipError_t align16_tb_entry_5(const AlignLaunch&, int, int, int, hipStream_t);
hipError_t align16_tb_entry_6(const AlignLaunch&, int, int, int, hipStream_t);
hipError_t align16_tb_entry_7(const AlignLaunch&, int, int, int, hipStream_t);

struct Cfg16 { int G, P; };
static const Cfg16 kCfgs16[] = {       // ascending G * 2P: windows of 32, 64, 96, 128, 192 blocks; then the latency shapes
    {16, 1}, {16, 2}, {16, 3}, {32, 2}, {32, 3}, {64, 1}, {64, 2}, {128, 1},
    {64, 3},                           // (round 4) windows of 257..384 blocks: bands up to 3064 on long pairs
};

int align16_mig_fields(int P) { return mig_fields(P); }

bool agatha16_scores_ok(const AlignParams& p)
{
    if (p.band_width < 16) return false;
    if (p.match < 0 || p.match > kAlign16MaxMatch || p.mismatch < 0 || p.mismatch > kAlign16MaxMismatch) return false;
    if (p.gap_open < 0 || p.gap_open > kAlign16MaxGapOpen || p.gap_extend < 0 || p.gap_extend > kAlign16MaxGapExtend) return false;
    int per = 2 * p.gap_extend; if (p.mismatch > per) per = p.mismatch; if (per < 1) per = 1;
    const int spread = p.gap_open + p.gap_extend + per * (p.band_width + 16) + 64;
    return spread <= kAlign16MaxSpread;
}

// the smallest packed-int16 configuration that holds the window, unless it would leave most of its slots idle
static const Cfg16* pick16(const AlignParams& p, int window_blocks)
{
    if (!agatha16_scores_ok(p)) return nullptr;
    for (const Cfg16& c : kCfgs16)
        if (c.G < 64 && c.G * 2 * c.P >= window_blocks) return (c.G * 2 * c.P <= 2 * window_blocks + 32) ? &c : nullptr;
    // (round 4) wider windows -- bands 1529..3064 on long pairs, which used to run on the int32 kernel only: one pair per wave,
    // two or three register pairs per lane, as the THROUGHPUT shape
    for (const Cfg16& c : kCfgs16)
        if (c.G == 64 && c.P >= 2 && c.G * 2 * c.P >= window_blocks) return &c;
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
    // ... or one pair on two cooperating waves when that halves the register pairs per lane again (windows of 129..256 blocks)
    if (*GL == 64 && *PL == 2 && window_blocks <= 256) { *GL = 128; *PL = 1; }
    if (c->G == 64 && c->P == 2 && window_blocks <= 256) { *GL = 128; *PL = 1; }       // (throughput shape <64, 2>: windows of 193..256 blocks)
    return true;
}

hipError_t launch_align16(const AlignLaunch& L, int G, int P, int kid, hipStream_t st)
{
    typedef hipError_t (*entry_fn)(const AlignLaunch&, int, int, int, hipStream_t);
    static const entry_fn entries[8] = {align16_entry_0, align16_entry_1, align16_entry_2, align16_entry_3,
                                        align16_entry_4, align16_entry_5, align16_entry_6, align16_entry_7};
    const int W = (L.p.band_width + 7) / 8, t0 = L.p.band_width - 8 * W;       // the cut diagonal of edge blocks, 0..-7
    return entries[-t0](L, G, P, kid, st);
}

// Traceback pass: the int16 shape whose G * 2P equals the slot count of the int32 traceback kernel for this window (both write
// the code area of a pass, one layout), if the scores allow the kernel at all.
bool align16_tb_config(const AlignParams& p, int window_blocks, int group_slots, int* G, int* P)
{
    if (!agatha16_scores_ok(p)) return false;
    for (const Cfg16& c : kCfgs16)
        if (c.P == 3 && c.G < 64 && c.G * 2 * c.P == group_slots && group_slots >= window_blocks) { *G = c.G; *P = c.P; return true; }
    return false;
}

hipError_t launch_align16_tb(const AlignLaunch& L, int G, int P, int pass, hipStream_t st)
{
    typedef hipError_t (*entry_fn)(const AlignLaunch&, int, int, int, hipStream_t);
    static const entry_fn entries[8] = {align16_tb_entry_0, align16_tb_entry_1, align16_tb_entry_2, align16_tb_entry_3,
                                        align16_tb_entry_4, align16_tb_entry_5, align16_tb_entry_6, align16_tb_entry_7};
    if (!L.tb_codes || !L.tb_off || !L.tb_pass || !L.tb_plan) return hipErrorInvalidValue;
    const int W = (L.p.band_width + 7) / 8, t0 = L.p.band_width - 8 * W;
    return entries[-t0](L, G, P, pass, st);
}

}  // namespace agatha
