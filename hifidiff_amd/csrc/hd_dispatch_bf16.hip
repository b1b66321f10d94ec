// GEMMs whose A operand is a bf16 matrix (conv3 / conv5 / SCA of the NAF blocks, ResNet 1x1 convs, VAE projections).  See hd_dispatch.hpp.
#include "hd_dispatch.hpp"
namespace hd {
hipError_t dispatch_gemm_bf16(const GemmP& p, LdKind lk, EpKind ek, int mode, hipStream_t s) {
    if (lk == LK_BF16S && ek == EK_RESID) return launch_tile<LdBF16Scale, EpResidF32, false>(p, mode, s);
    if (lk != LK_BF16) return hipErrorInvalidValue;
    if (ek == EK_RESID) return launch_tile<LdBF16Plain, EpResidF32, false>(p, mode, s);
    if (ek == EK_BIASBF16) return launch_tile<LdBF16Plain, EpBiasBF16, false>(p, mode, s);
    if (ek == EK_PIXSHUF) return launch_tile<LdBF16Plain, EpPixShufF32, false>(p, mode, s);
    if (ek == EK_SCA) {
        static const bool no_dw1 = hd_env("HD_NO_DW1") != nullptr;
        if (p.scale_hw == 1 && !no_dw1) return launch_skinny_auto<1, 1, false, LdBF16Plain, EpSca1BF16>(p, s);
        return launch_skinny_auto<1, 1, false, LdBF16Plain, EpScaBF16>(p, s);   // its in-place G scaling is a skinny tile epilogue
    }
    if (ek == EK_BIASF32) return launch_tile<LdBF16Plain, EpBiasF32, false>(p, mode, s);                  // VAE attention projections
    return hipErrorInvalidValue;
}
}  // namespace hd
