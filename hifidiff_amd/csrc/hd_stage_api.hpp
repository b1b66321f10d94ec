// hd_stage_api.hpp -- parameter blocks and host entry points of the persistent stage kernels (hd_xcd.hpp, hd_xcd2.hpp, hd_face.hpp).
// The kernels are compiled in their own translation unit (hd_stages.hip); the library's host side (hd_lib.hip) sees only this.
#pragma once
#include <hip/hip_runtime.h>

namespace hd {

struct XBlockW {
    const uint4 *w1, *wsca, *w3, *w4, *w5;           // packed bf16 B fragments (hd_kernels.hpp: pack_weight_kernel)
    const float *b1, *bsca, *b3, *b4, *b5, *beta, *gamma;
    const float *dw_w, *dw_b;                         // depthwise 3x3 weights tap-major [9][2C], bias [2C]
    int film_off, pad_;                               // this block's 4C FiLM values: [bias_att | gain_att | bias_ffn | gain_ffn]
};

constexpr int XS_MAXBLK = 8;
constexpr int XS_THREADS = 512;
constexpr int XS_GROUPS = 8, XS_GROUP_WG = 32, XS_FACES = 8;      // 8 XCDs x 32 CUs; faces per group
constexpr unsigned XS_SPINS = 1u << 21;                            // polls (>= 0.3 us each) before a wait gives up

struct XStageP {
    int B, nblocks;                       // faces in the batch (<= 64), blocks in this stage (<= XS_MAXBLK)
    const XBlockW* blocks;                // device array [nblocks]
    // level buffers, standard layouts ([rows][C] channels-last over the whole batch)
    float* X; unsigned short* Xb; float2* sx;            // residual stream: entry (from the previous launch) and exit
    unsigned short* G; unsigned short* Yb; float2* sy;   // hand-off buffers between phases
    unsigned short* pooled16; float* pooled; float* S;   // pooled16: hand-off; pooled / S: introspection copies (may be NULL)
    const float* film; float ln_eps;                     // FiLM row shared by all faces
    unsigned short* outg16; const float* gate_c; const float* gate_s; const float* add_src;   // HCA input after the last block (or NULL)
    unsigned *flags, *hello, *gstate;     // [8][32] words each: one 128-byte line per group
    unsigned* tmo;                        // host-visible timeout word (pinned, device-mapped)
    unsigned* abort_dev;                  // the same code in device memory: every stage launch reads it at entry and steps aside when it is
                                          // set (the call's results are then poisoned by poison_if_abort_kernel, hd_kernels.hpp)
    int test_abort;                       // fault injection (hd_set_option "stage_test_abort"): group 0 gives up its wait for phase test_abort - 1
    int phase_limit;                      // introspection: stop after this many phases (<= 0: all)
    int force_global;                     // test: use the placement-independent hand-off even when the group shares an XCD
#ifdef HD_STAMPS
    unsigned long long* stamps;           // [phase][workgroup][8]: 0 start, 1 barrier passed, 2 K loop done, 3 epilogue stores issued, 4 drained, 5 published
    int dbg_no_a, dbg_no_w;               // timing-only what-ifs (results are garbage): activation loads through zero-record descriptors / no weight loads
#endif
};

struct FStageP {
    int B, nblocks;                        // faces (<= 64), blocks of this run
    const XBlockW* blocks;                 // device array [nblocks]
    float* X;                              // [M][C] fp32: entry, per-block hand-off of x', exit
    unsigned short* Xb;                    // exit: bf16 copy of x' (what the down conv gathers), or NULL
    unsigned short* outg16; const float *gate_c, *gate_s;     // exit: (x') * (1 + w_c + w_s) for the HCA conv, or NULL
    float* pool_part;                      // [faces][CL][C] channel sums of the gate of each workgroup's rows
    const float* film; float ln_eps;
    unsigned *flags, *gstate;              // [64 faces][16] words each
    unsigned* tmo;                         // host-visible timeout word (pinned, device-mapped)
    unsigned* abort_dev;                   // the same code in device memory, read at entry by every stage launch (hd_xcd.hpp)
    int test_abort;                        // fault injection: 1000 + b = face 0 gives up its pool wait of block b
    int block_limit;                       // introspection: stop after this many blocks (0: all; < 0: none -- entry and exit only)
    // level 0, first stage of the encoder: the intro conv (Conv2d(4,128,3,pad 1), models/denoiser/model.py:159-167,235) as the stage's ENTRY --
    // every workgroup computes x for its own and its halo image rows from the NCHW latents instead of reading X (one launch less per step)
    const float* intro_lat;                // [B][4][16][16] latents, or NULL: X is read as written by the previous launch
    const float *intro_wT, *intro_b;       // weights re-laid [36][128] (intro_weight_layout_kernel), bias [128]
    int* intro_step; int intro_advance;    // the loop's step counter (StepState::step), advanced by the first workgroup when intro_advance
    // level 1, first stage of the encoder: the down conv of level 0 (Conv2d(128,256,2,2), models/denoiser/model.py:236-241) as the stage's ENTRY --
    // every workgroup gathers the 2 x 2 patches of its own and its halo image rows from level 0's bf16 copy and runs the K = 512 GEMM itself
    const unsigned short* down_A;          // level 0's Xb [B * 256][128] bf16, or NULL: X is read as written by the previous launch
    const uint4* down_W; const float* down_b;            // packed B fragments [8 tiles][32 k-steps][64] (k = tap * 128 + c), bias [256]
    // level 0, decoder stage: the last up conv (1x1 conv 256 -> 512 + PixelShuffle(2) + encoder skip, models/denoiser/model.py:249-251) as the stage's
    // ENTRY -- X holds the skip; every workgroup adds the up conv of the 24 level-1 pixels under its own and halo image rows (one 32-row MFMA tile)
    const unsigned short* up_A;            // level 1's bf16 rows [B * 64][256] (HCA output, or the blocks' own copy), or NULL: X already holds the sum
    const uint4* up_W;                     // packed B fragments [16 tiles][16 k-steps][64], output channel n' = sub-pixel * 128 + c (no bias)
#ifdef HD_STAMPS
    unsigned long long* stamps;            // [block][workgroup][8]
    int dbg_no_w;                          // timing-only what-if (results are garbage): no weight loads
#endif
};
struct X2StageP {
    int B, nblocks;
    const XBlockW* blocks;                 // device array [nblocks]; weights in the 16x16x32 packing (pack_weight16_kernel)
    float* X; unsigned short* Xb; const float2* sx;      // entry (standard layouts, sx: [M][C/32] partials of 32) and exit (X, Xb)
    uint4 *hX, *hG, *hY;                   // hand-off, fragment order: [row block][C/32][64 lanes] uint4
    float2 *hsx, *hsy;                     // hand-off statistics [row][C/16]: (mean, M2) of 16 channels
    unsigned short* pooled16;              // hand-off [B][C] bf16 (standard)
    // introspection copies in the standard layouts (written by the phase a phase_limit stops at; may be NULL)
    unsigned short *dG, *dYb; float *dpooled, *dS;
    const float* film; float ln_eps;
    unsigned short* outg16; const float *gate_c, *gate_s, *add_src;
    unsigned *flags, *hello, *gstate;      // [8 groups][128] | [8][32] | [8][32]
    unsigned *tmo, *abort_dev; int test_abort;           // test_abort: n in 1..: compute waves of group 0 give up the wait for phase n - 1; 2000 + p: loader 0 of group 0 before LayerNorm phase p
    int phase_limit, force_global;
#ifdef HD_STAMPS
    unsigned long long* stamps;            // [phase][workgroup][8] of compute wave 0
    int dbg_no_w;                          // timing-only what-if (results are garbage): the loaders count their steps but move no weights
#endif
};

// LayerNorm -> conv1 -> depthwise 3x3 -> SimpleGate of 32 x 32 faces, C = 128, by strips of 4 image rows (hd_strip.hpp)
struct StripP {
    int faces, side, C;                    // faces in the batch; face side (32) and channels (128): what the kernel is written for
    const unsigned short* Xb;              // [faces * side^2][C] bf16 copy of the block input
    const float2* stats_in; int stats_np, stats_cnt;     // its LayerNorm partials [rows][np] (mean, M2) of cnt channels each
    const uint4* W1; const float* b1;      // conv1: packed bf16 B fragments (N = 2C), bias [2C]
    const float *dw_wT, *dw_b;             // depthwise 3x3 weights tap-major [9][2C], bias [2C]
    const float* film; int film_gain_off, film_bias_off, film_face_stride, face0; float ln_eps;
    unsigned short* G;                     // [rows][C] bf16 gate
    float* pool_part;                      // [faces][side / 4][C] channel sums of the gate per strip
};

// host entry points (hd_stages.hip).  C selects the instantiation: XCD-local stages 1024 (level 3, 2 x 2 faces) / 512 (level 2,
// 4 x 4 faces); face-cluster stages 128 (level 0, 16 x 16 faces) / 256 (level 1, 8 x 8 faces: 16 rows per workgroup put it on all 256 CUs).  hipErrorInvalidValue otherwise.
hipError_t run_xcd_stage(int C, const XStageP& p, hipStream_t s);
hipError_t run_xcd2_stage(int C, const X2StageP& p, hipStream_t s);
hipError_t run_strip_dwgate(const StripP& p, hipStream_t s);                        // hipErrorInvalidValue unless side = 32, C = 128
hipError_t run_face_stage(int C, int own_rows, const FStageP& p, hipStream_t s);   // own_rows: pixel rows per workgroup, 32 (C = 128, 256) or 16 (C = 256)

}  // namespace hd
