// GEMMs with an fp32 A operand (gate MLPs, up-convs) and implicit-GEMM convolutions over bf16 NHWC.  See hd_dispatch.hpp.
#include "hd_dispatch.hpp"
namespace hd {
hipError_t dispatch_gemm_misc(const GemmP& p, LdKind lk, EpKind ek, int mode, hipStream_t s) {
    if (lk == LK_F32 && ek == EK_BIASF32) return launch_tile<LdF32Plain, EpBiasF32, false>(p, mode, s);
    if (lk == LK_F32 && ek == EK_PIXSHUF) return launch_tile<LdF32Plain, EpPixShufF32, false>(p, mode, s);
    if (lk == LK_CONV_BF16 && ek == EK_BIASF32) return launch_tile<LdConv<true, false>, EpBiasF32, false>(p, mode, s);
    if (lk == LK_CONV_BF16 && ek == EK_BIASBF16) return launch_tile<LdConv<true, false>, EpBiasBF16, false>(p, mode, s);
    if (lk == LK_CONV_BF16 && ek == EK_RESID) return launch_tile<LdConv<true, false>, EpResidF32, false>(p, mode, s);      // VAE ResnetBlock2D conv2 + shortcut
    return hipErrorInvalidValue;
}
}  // namespace hd
