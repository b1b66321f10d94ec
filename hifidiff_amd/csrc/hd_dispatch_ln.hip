// LayerNorm + FiLM GEMMs with one FiLM row for all faces (the sampling loop).  See hd_dispatch.hpp.
#include "hd_dispatch.hpp"
namespace hd {
hipError_t dispatch_gemm_ln_shared(const GemmP& p, EpKind ek, int mode, hipStream_t s) { return dispatch_ln<LdF32LN>(p, ek, mode, s); }
}  // namespace hd
