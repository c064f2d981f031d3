// align16_tb_inst.hip -- the traceback pass on the packed-int16 kernel (align16_body.inc with TBK = true) for ONE cut diagonal
// T0 = -AGATHA16_NT0: the two shapes whose slot count (96, 192) is also one of the int32 traceback kernel's, so that both
// kernels of a pass write the same code layout (align_tb.hip).
#include "align16_body.inc"

#ifndef AGATHA16_NT0
#error "compile with -DAGATHA16_NT0=0..7"
#endif
#define AGATHA16_CAT2(a, b) a##b
#define AGATHA16_CAT(a, b) AGATHA16_CAT2(a, b)

namespace agatha {

hipError_t AGATHA16_CAT(align16_tb_entry_, AGATHA16_NT0)(const AlignLaunch& L, int G, int P, int pass, hipStream_t st)
{
    constexpr int T0 = -(AGATHA16_NT0);
    if (G == 16 && P == 3) return launch_align16_t<16, 3, T0, true>(L, pass, st);
    if (G == 32 && P == 3) return launch_align16_t<32, 3, T0, true>(L, pass, st);
    return hipErrorInvalidValue;
}

}  // namespace agatha
