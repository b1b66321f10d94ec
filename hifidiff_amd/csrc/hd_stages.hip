// hd_stages.hip -- translation unit of the persistent stage kernels (gfx950): a run of ConditionalNAFBlocks
// (models/denoiser/conditional_naf.py:108-136) of one level as one launch.  Entry points: hd_stage_api.hpp.
#include "hd_face.hpp"
#include "hd_xcd.hpp"
#include "hd_xcd2.hpp"

namespace hd {

hipError_t run_xcd_stage(int C, const XStageP& p, hipStream_t s) {
    if (C == 1024) return launch_xcd_stage<1024, 4>(p, s);
    if (C == 512) return launch_xcd_stage<512, 16>(p, s);
    return hipErrorInvalidValue;
}

hipError_t run_xcd2_stage(int C, const X2StageP& p, hipStream_t s) {
    if (C == 1024) return launch_xcd2_stage<1024, 4>(p, s);
    if (C == 512) return launch_xcd2_stage<512, 16>(p, s);
    return hipErrorInvalidValue;
}

hipError_t run_face_stage(int C, int own_rows, const FStageP& p, hipStream_t s) {
    if (C == 128 && own_rows == 32) return launch_face_stage<128, 32>(p, s);
    if (C == 256 && own_rows == 32) return launch_face_stage<256, 32>(p, s);
    if (C == 256 && own_rows == 16) return launch_face_stage<256, 16>(p, s);
    return hipErrorInvalidValue;
}

}  // namespace hd
