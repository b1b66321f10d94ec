// LayerNorm + FiLM GEMMs with per-face timesteps (training-style batches of hd_eps).  See hd_dispatch.hpp.
#include "hd_dispatch.hpp"
namespace hd {
hipError_t dispatch_gemm_ln_face(const GemmP& p, EpKind ek, int mode, hipStream_t s) { return dispatch_ln<LdF32LNFace>(p, ek, mode, s); }
}  // namespace hd
