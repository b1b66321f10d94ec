// hd_strip.hip -- translation unit of the strip kernel (hd_strip.hpp): the front of a NAFBlock on 32 x 32 faces (latent 32, level 0).
// Entry point: hd_stage_api.hpp.
#include "hd_strip.hpp"

namespace hd {

hipError_t run_strip_dwgate(const StripP& p, hipStream_t s) { return launch_strip_dwgate(p, s); }

}  // namespace hd
