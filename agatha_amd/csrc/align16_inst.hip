// align16_inst.hip -- instantiations of the packed-int16 kernel (align16_body.inc) for ONE cut diagonal T0 = -AGATHA16_NT0.
// The eight cut diagonals are separate translation units so that they compile in parallel (make -j); the tools define
// AGATHA16_ONLY_G / AGATHA16_ONLY_P to build a single shape.
#include "align16_body.inc"

#ifndef AGATHA16_NT0
#error "compile with -DAGATHA16_NT0=0..7"
#endif
#define AGATHA16_CAT2(a, b) a##b
#define AGATHA16_CAT(a, b) AGATHA16_CAT2(a, b)

namespace agatha {

hipError_t AGATHA16_CAT(align16_entry_, AGATHA16_NT0)(const AlignLaunch& L, int G, int P, int kid, hipStream_t st)
{
    constexpr int T0 = -(AGATHA16_NT0);
#define AGATHA16_SHAPE(GG, PP) if (G == GG && P == PP) return launch_align16_t<GG, PP, T0>(L, kid, st);
#ifdef AGATHA16_ONLY_G
    AGATHA16_SHAPE(AGATHA16_ONLY_G, AGATHA16_ONLY_P)
#else
    // windows of 32, 64, 96, 128, 192 blocks; then the latency shapes
    AGATHA16_SHAPE(16, 1) AGATHA16_SHAPE(16, 2) AGATHA16_SHAPE(16, 3) AGATHA16_SHAPE(32, 2) AGATHA16_SHAPE(32, 3)
    AGATHA16_SHAPE(64, 1) AGATHA16_SHAPE(64, 2) AGATHA16_SHAPE(128, 1) AGATHA16_SHAPE(64, 3)
#endif
#undef AGATHA16_SHAPE
    return hipErrorInvalidValue;
}

}  // namespace agatha
