// hd_face.hpp — a run of ConditionalNAFBlocks of a SHALLOW level (level 0: C = 128, 16 x 16 faces; level 1: C = 256, 8 x 8) as
// ONE launch, rows split over the workgroups of a face (gfx950 only).
//
// At these levels a block is row-local except for two things (models/denoiser/conditional_naf.py:108-136): the depthwise
// 3x3 conv2 reads the image rows above and below (:116), and the SCA pool averages over the whole face (:119).  A workgroup
// owns 32 pixel rows of one face (2 image rows at level 0, 4 at level 1) and ALL channels, one 32-column tile per wave as in
// hd_chain.hpp; the residual stream tile (x -> y -> x') lives in LDS for the whole run.  Per block the workgroups of a face
// (8 at level 0, 2 at level 1: a "cluster") meet twice:
//   A  x' rows of the previous block are published (fp32, the level's X buffer) -> the neighbours' boundary image rows are
//      read back, LayerNorm2d + FiLM is applied to own + halo rows, conv1 runs on 32 + 2S rows, so that the depthwise conv of
//      the own rows finds its halo in LDS (recomputing conv1 for the halo instead of exchanging T1: half the exchanged bytes,
//      and the exchange then sits at the block boundary where x' has to leave anyway);
//   B  per-workgroup channel sums of the gate -> the face's pooled mean.
// Everything else is the chain kernel's arithmetic (hd_chain.hpp): SCA GEMV on the MFMA, G * s, conv3, y, LayerNorm (two-pass,
// fp32 statistics, bf16 value), conv4, SimpleGate, conv5, x'.  Replaces 2 launches per block (fused conv1, chain kernel).
// The encoder's first stage also takes the intro conv as its entry (FStageP::intro_lat): own and halo image rows of x straight from the latents;
// its second stage (level 1) takes the down conv of level 0 the same way (FStageP::down_A: a 32-row x K = 512 GEMM per workgroup), and the
// level-0 decoder stage the last up conv (FStageP::up_A: 24 level-1 pixels x K = 256 x 512 sub-pixel-major columns, added to the skip X holds).
//
// Hand-off (MI355X_MICROARCH.md "Valid forms" row 1, placement-independent): hand-off data is stored write-through (sc1) by
// every wave, every storing wave drains (s_waitcnt vmcnt(0)), the workgroup's barrier, ONE lane stores the workgroup's flag
// (sc1); a consumer polls the cluster's flags (one line per face) with sc1 loads and, after its workgroup barrier, reads the
// handed-off rows with sc1 loads only.  Flags carry an epoch the kernel advances itself (launch counter per face).  The
// members of a cluster get block ids that are equal mod 8 (one XCD under round-robin dispatch: speed only).  Every spin is
// bounded; all workgroups of a face must be resident (512 workgroups at two per CU at level 0, 128 at level 1).
#pragma once
#include "hd_chain.hpp"
#include "hd_xcd.hpp"

namespace hd {

#ifdef HD_STAMPS
#define HD_FSTAMP(i) do { if (p.stamps && tid == 0) p.stamps[((size_t)blk * gridDim.x + blockIdx.x) * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define HD_FSTAMP(i) do { } while (0)
#endif

// OWN = pixel rows a workgroup owns: 32 (level 0: 2 image rows of 16, 512 workgroups at two per CU; level 1 in its first form:
// 4 image rows of 8, 128 workgroups = half the chip) or 16 (level 1: 2 image rows, 256 workgroups = every CU; own + halo rows
// are then exactly one 32-row MFMA tile, the GEMMs after the depthwise stage use half of theirs)
template <int C, int OWN_ = 32>
struct FaceCfg {
    static constexpr int S = (C == 128) ? 16 : 8;              // face side
    static constexpr int OWN = OWN_;
    static constexpr int HW = S * S, CL = HW / OWN, RI = OWN / S;   // pixels per face, workgroups per face, image rows per workgroup
    static constexpr int NT = C / 32, THREADS = 64 * NT, KS = C / 16;
    static constexpr int ROWS = OWN + 2 * S, MT1 = (ROWS + 31) / 32;   // own + halo rows; 32-row MFMA tiles of conv1
    static constexpr int AROW = C * 2 + 16;                    // bytes per bf16 tile row (padded)
    static constexpr int XROW = C + 4;                         // floats per fp32 tile row (padded)
    static constexpr int XT_OFF = 0;                           // x / y / x' tile fp32 [OWN][XROW]
    static constexpr int ALN_OFF = XT_OFF + OWN * XROW * 4;    // LayerNorm output bf16 [MT1 * 32][AROW]: rows 0..OWN-1 own, OWN.. above halo, OWN+S.. below
    // gate tile bf16 [32][AROW].  OWN = 32: rows 32..63 of the LN tile (the halo rows are dead once conv1 has run; a barrier separates
    // the two uses) -> two workgroups per CU at C = 128.  OWN = 16: the LN tile is one MFMA tile, the gate tile gets its own
    static constexpr int A1_OFF = ALN_OFF + 32 * AROW;
    static constexpr int T1_OFF = ALN_OFF + 64 * AROW;         // per wave: conv1 half tile fp32 [MT1 * 32][32]
    static constexpr int SV_OFF = T1_OFF + NT * MT1 * 32 * 32 * 4;   // sca vector [C]
    static constexpr int GB_OFF = SV_OFF + C * 4;              // FiLM gain | bias of norm1 and norm2 [4][C]
    static constexpr int PL_OFF = GB_OFF + 4 * C * 4;          // pooled mean [C]
    static constexpr int PP_OFF = PL_OFF + C * 4;              // this workgroup's channel sums [C]
    static constexpr int SMEM = PP_OFF + C * 4;
    static_assert(ROWS <= 64 && (OWN == 32 || OWN == 16) && RI >= 2 && RI % 2 == 0, "own + halo rows fit two MFMA row tiles; two lane halves share the image rows");
};

typedef __attribute__((address_space(1))) unsigned fs_gu32;

// all B fragments of one column tile (K = C); the pointer comes from the LDS copy of the block table: say it is global
template <int C>
__device__ __forceinline__ void face_load_b(const uint4* W, int tile, int lane, uint4* b) {
#ifdef HD_STAMPS
    if (!W) return;                                                  // what-if build: FStageP::dbg_no_w hands in null weight pointers
#endif
    const uint4* Wl = W + (size_t)tile * (C / 16) * 64 + lane;
#pragma unroll
    for (int ks = 0; ks < C / 16; ++ks) b[ks] = xs_ldg_u4(Wl + ks * 64);
}

template <int C, int OWN = 32>
__global__ __launch_bounds__((FaceCfg<C, OWN>::THREADS), 2) void naf_face_stage_kernel(const FStageP p) {      // two waves per SIMD: two 4-wave workgroups per CU at C = 128
    typedef FaceCfg<C, OWN> K;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ XBlockW s_blk[XS_MAXBLK];
    __shared__ unsigned s_abort, s_base;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bid = blockIdx.x, xcd = bid & 7, jj = bid >> 3;
    const int fl = jj / K::CL, kk = jj - fl * K::CL;
    const int face = xcd * 8 + fl;
    if (face >= p.B) return;                                        // nobody of this face takes part
    const int M = p.B * K::HW;
    const int row0 = face * K::HW + kk * OWN;                       // own rows
    const bool has_up = kk > 0, has_dn = kk < K::CL - 1;
    const int tile = wave, col = tile * 32 + (lane & 31);
    float* xt = reinterpret_cast<float*>(smem + K::XT_OFF);
    float* s_vec = reinterpret_cast<float*>(smem + K::SV_OFF);
    float* gb = reinterpret_cast<float*>(smem + K::GB_OFF);
    float* pooled = reinterpret_cast<float*>(smem + K::PL_OFF);
    float* poolp = reinterpret_cast<float*>(smem + K::PP_OFF);
    float* t1w = reinterpret_cast<float*>(smem + K::T1_OFF) + wave * K::MT1 * 32 * 32;

    {
        const unsigned* src = reinterpret_cast<const unsigned*>(p.blocks);
        unsigned* dst = reinterpret_cast<unsigned*>(s_blk);
        for (int i = tid; i < p.nblocks * (int)(sizeof(XBlockW) / 4); i += K::THREADS) dst[i] = src[i];
        if (tid == 0) {
            s_abort = __hip_atomic_load((fs_gu32*)p.abort_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // an earlier stage of this call gave up
            s_base = __hip_atomic_load((fs_gu32*)(p.gstate + face * 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * 64u;
        }
    }
    const __amdgpu_buffer_rsrc_t rs_X = __builtin_amdgcn_make_buffer_rsrc(p.X, 0, (int)((size_t)M * C * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_PP = __builtin_amdgcn_make_buffer_rsrc(p.pool_part, 0, p.B * K::CL * C * 4, 0x00020000);
    fs_gu32* flags = (fs_gu32*)(p.flags + face * 16);

    // ---- entry: own rows of x (written by the previous launch), or -- first stage of the encoder -- the intro conv of own + halo rows ----
    bool intro = false;                                              // x (own + halo rows) is computed by this stage's entry: no launch wrote it
    float* hb = reinterpret_cast<float*>(smem + K::T1_OFF);          // x of the halo image rows (above: rows 0..S-1, below: S..2S-1), until block 0's LayerNorm
    if constexpr (C == 128 && OWN == 32) intro = p.intro_lat != nullptr;
    bool from_x = !intro;
    if constexpr (C == 256 && OWN == 16) from_x = p.down_A == nullptr;
    if (from_x) {
        for (int u = tid; u < OWN * (C / 4); u += K::THREADS) {
            const int r = u / (C / 4), q = u - r * (C / 4);
            *reinterpret_cast<float4*>(xt + r * K::XROW + q * 4) = *reinterpret_cast<const float4*>(p.X + (size_t)(row0 + r) * C + q * 4);
        }
    }
    if constexpr (C == 128 && OWN == 32) {
        if (intro) {
            // intro_conv_kernel's arithmetic (hd_kernels.hpp) for the 4 image rows this workgroup needs: wave w = image row y0 + {0, 1, -1, 2};
            // lanes over output channels (lane, lane + 64), the lane's 72 weights in registers, the 6 x 18 x 4 latent patch in LDS (broadcast reads)
            constexpr int S = K::S;
            if (p.intro_advance && bid == 0 && tid == 0) p.intro_step[0] += 1;
            float* patch = gb;                                           // [4][6][S + 2] floats: the FiLM rows land here only in block 0
            const int y0 = kk * K::RI;
            for (int i = tid; i < 4 * 6 * (S + 2); i += K::THREADS) {
                const int ci = i / (6 * (S + 2)), rr = (i - ci * 6 * (S + 2)) / (S + 2), xx = i - ci * 6 * (S + 2) - rr * (S + 2);
                const int yy = y0 - 2 + rr, x = xx - 1;
                patch[i] = (yy >= 0 && yy < S && x >= 0 && x < S) ? p.intro_lat[((size_t)(face * 4 + ci) * S + yy) * S + x] : 0.f;
            }
            f32x2_t wl[36];                                              // (channel lane, channel lane + 64): packed fp32 FMAs, both channels per instruction
#pragma unroll
            for (int r = 0; r < 36; ++r) wl[r] = (f32x2_t){p.intro_wT[r * 128 + lane], p.intro_wT[r * 128 + lane + 64]};
            const f32x2_t b01 = {p.intro_b[lane], p.intro_b[lane + 64]};
            __syncthreads();
            const int dy = wave == 0 ? 0 : wave == 1 ? 1 : wave == 2 ? -1 : 2;      // image row y0 + dy; patch row 2 + dy is its centre
            const bool needed = wave < 2 || (wave == 2 ? has_up : has_dn);
            float* dst = wave < 2 ? xt + wave * S * K::XROW : hb + (wave - 2) * S * K::XROW;
            if (needed) {
                // sliding 3 x 3 x 4 window along the image row: 12 new patch values per pixel (LDS broadcast reads), 36 packed FMAs
                float win[4][3][3];
#pragma unroll
                for (int ci = 0; ci < 4; ++ci)
#pragma unroll
                    for (int r = 0; r < 3; ++r) { win[ci][r][1] = patch[(ci * 6 + 1 + dy + r) * (S + 2)]; win[ci][r][2] = patch[(ci * 6 + 1 + dy + r) * (S + 2) + 1]; }
#pragma unroll
                for (int px = 0; px < S; ++px) {
                    f32x2_t a = b01;
#pragma unroll
                    for (int ci = 0; ci < 4; ++ci)
#pragma unroll
                        for (int r = 0; r < 3; ++r) {
                            win[ci][r][0] = win[ci][r][1]; win[ci][r][1] = win[ci][r][2];
                            win[ci][r][2] = patch[(ci * 6 + 1 + dy + r) * (S + 2) + px + 2];
#pragma unroll
                            for (int k = 0; k < 3; ++k) a = __builtin_elementwise_fma((f32x2_t){win[ci][r][k], win[ci][r][k]}, wl[ci * 9 + r * 3 + k], a);
                        }
                    const float a0 = a[0], a1 = a[1];
                    dst[px * K::XROW + lane] = a0; dst[px * K::XROW + lane + 64] = a1;
                }
            }
        }
    }
    if constexpr (C == 128 && OWN == 32) {
        if (p.up_A) {
            // rows of the A tile: 0..7 the level-1 image row under the own rows (y1 = kk), 8..15 the one above, 16..23 the one below, 24..31 zero.
            // Wave w = sub-pixel (dy, dx) = (w >> 1, w & 1): its four 32-column tiles are the 128 channels of that sub-pixel; the output of level-1
            // pixel (y1, x1) lands at level-0 pixel (2 y1 + dy, 2 x1 + dx).  Of the halo rows only the adjacent image row is kept (above: dy = 1, below: dy = 0).
            constexpr int UROW = 256 * 2 + 16;
            static_assert(32 * UROW <= 64 * K::AROW, "the up conv's A tile fits the LayerNorm + gate tiles");
            char* sA = smem + K::ALN_OFF;
            for (int u = tid; u < 32 * 32; u += K::THREADS) {
                const int r = u >> 5, q = u & 31, x1 = r & 7, grp = r >> 3;
                const int y1 = grp == 0 ? kk : grp == 1 ? kk - 1 : kk + 1;
                const bool ok = grp == 0 || (grp == 1 && has_up) || (grp == 2 && has_dn);
                uint4 v = make_uint4(0, 0, 0, 0);
                if (ok) v = *reinterpret_cast<const uint4*>(p.up_A + (size_t)(face * 64 + y1 * 8 + x1) * 256 + q * 8);
                *reinterpret_cast<uint4*>(sA + r * UROW + q * 16) = v;
            }
            // weights in half tiles of 8 k-steps, three register sets: two halves (16 KB per wave) are in flight ahead of the MFMAs -- with one
            // ahead the entry was a chain of eight L2 round trips (5.7 us)
            uint4 wq[3][8];
            const uint4* Wl = p.up_W + (size_t)(wave * 4) * 16 * 64 + lane;
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) wq[h][ks] = xs_ldg_u4(Wl + (h * 8 + ks) * 64);
            __syncthreads();                                         // the A tile is staged, and the skip rows are in xt
            const char* ap = sA + (lane & 31) * UROW + (lane >> 5) * 16;
            const int dyw = wave >> 1, dxw = wave & 1;
            f32x16_t acc;
#pragma unroll
            for (int ch = 0; ch < 8; ++ch) {                           // chunk ch = half (ch & 1) of tile ch >> 1
                if (ch + 2 < 8) {
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks) wq[(ch + 2) % 3][ks] = xs_ldg_u4(Wl + ((ch + 2) * 8 + ks) * 64);
                }
                if ((ch & 1) == 0) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
                }
#pragma unroll
                for (int ks = 0; ks < 8; ++ks)
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_t*>(ap + ((ch & 1) * 8 + ks) * 32),
                                                                  __builtin_bit_cast(bf16x8_t, wq[ch % 3][ks]), acc, 0, 0, 0);
                if (ch & 1) {
                    const int c = (ch >> 1) * 32 + (lane & 31);
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int r = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5), grp = r >> 3, px = 2 * (r & 7) + dxw;
                        if (grp == 0) xt[(dyw * K::S + px) * K::XROW + c] += acc[i];
                        else if (grp == 1 && dyw == 1) hb[px * K::XROW + c] = acc[i];
                        else if (grp == 2 && dyw == 0) hb[(K::S + px) * K::XROW + c] = acc[i];
                    }
                }
            }
            // the halo rows' skip (X holds the encoder's output) joins their up term in hb: block 0 then takes its halo from LDS, as with the intro entry
            __syncthreads();
            for (int u = tid; u < 2 * K::S * (C / 4); u += K::THREADS) {
                const int r = u / (C / 4), q = u - r * (C / 4);
                const bool up = r < K::S;
                if (up ? has_up : has_dn) {
                    const int grow = up ? row0 - K::S + r : row0 + OWN + (r - K::S);
                    const float4 v = *reinterpret_cast<const float4*>(p.X + (size_t)grow * C + q * 4);
                    float4* d = reinterpret_cast<float4*>(hb + r * K::XROW + q * 4);
                    float4 t = *d;
                    t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
                    *d = t;
                }
            }
            intro = true;
        }
    }
    if constexpr (C == 256 && OWN == 16) {
        if (p.down_A) {
            // down conv of level 0 (2 x 2, stride 2, 128 -> 256) for the 32 level-1 pixels this workgroup needs: rows 0..15 own (2 image rows of 8),
            // 16..23 the image row above, 24..31 the one below -- the LayerNorm tile's row order.  A: the 2 x 2 patches gathered from level 0's bf16
            // copy into LDS ([32][4 taps x 128 channels], over the LayerNorm + gate tiles, which block 0 fills later); one 32-column tile per wave.
            intro = true;
            constexpr int DROW = 4 * 128 * 2 + 16;
            static_assert(32 * DROW <= 64 * K::AROW, "the patch tile fits the LayerNorm + gate tiles");
            char* sA = smem + K::ALN_OFF;
            for (int u = tid; u < 32 * 64; u += K::THREADS) {
                const int r = u >> 6, tap = (u >> 4) & 3, q = u & 15;
                const int x1 = r & 7;
                int y1 = kk * K::RI + (r >> 3);
                bool ok = true;
                if (r >= 24) { y1 = kk * K::RI + K::RI; ok = has_dn; } else if (r >= 16) { y1 = kk * K::RI - 1; ok = has_up; }
                uint4 v = make_uint4(0, 0, 0, 0);
                if (ok) {
                    const int l0row = face * 256 + (2 * y1 + (tap >> 1)) * 16 + 2 * x1 + (tap & 1);
                    v = *reinterpret_cast<const uint4*>(p.down_A + (size_t)l0row * 128 + q * 8);
                }
                *reinterpret_cast<uint4*>(sA + r * DROW + (tap * 128 + q * 8) * 2) = v;
            }
            uint4 wa[16], wb[16];
            {
                const uint4* Wl = p.down_W + (size_t)tile * 32 * 64 + lane;
#pragma unroll
                for (int ks = 0; ks < 16; ++ks) wa[ks] = xs_ldg_u4(Wl + ks * 64);
#pragma unroll
                for (int ks = 0; ks < 16; ++ks) wb[ks] = xs_ldg_u4(Wl + (16 + ks) * 64);
            }
            const float bias = p.down_b[col];
            __syncthreads();
            f32x16_t acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
            const char* ap = sA + (lane & 31) * DROW + (lane >> 5) * 16;
#pragma unroll
            for (int ks = 0; ks < 16; ++ks)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_t*>(ap + ks * 32), __builtin_bit_cast(bf16x8_t, wa[ks]), acc, 0, 0, 0);
#pragma unroll
            for (int ks = 0; ks < 16; ++ks)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_t*>(ap + (16 + ks) * 32), __builtin_bit_cast(bf16x8_t, wb[ks]), acc, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int r = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
                float* dst = r < OWN ? xt + r * K::XROW : hb + (r - OWN) * K::XROW;
                dst[col] = acc[i] + bias;
            }
        }
    }
    __syncthreads();
    const unsigned base = s_base;
    bool dead = false;

    // publish: every storing wave has drained, then one lane stores the flag; wait: wave 0 polls the cluster's flags
    auto publish = [&](unsigned value) __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(flags + kk, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto poll = [&](unsigned value, unsigned code) __attribute__((always_inline)) {
        if (wave == 0) {
            for (unsigned spins = 0;; ++spins) {
                const unsigned v = lane < K::CL ? __hip_atomic_load(flags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : value;
                if (__all((int)(v - value) >= 0)) break;
                if (spins > XS_SPINS) {
                    if (lane == 0) {
                        s_abort = 1u;
                        __hip_atomic_store((fs_gu32*)p.abort_dev, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store((fs_gu32*)p.tmo, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
    };
    auto wait_for = [&](unsigned value, unsigned code) __attribute__((always_inline)) {
        if (wave == 0) {
            for (unsigned spins = 0;; ++spins) {
                const unsigned v = lane < K::CL ? __hip_atomic_load(flags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : value;
                const bool inject = p.test_abort >= 1000 && (int)(code & 0xffu) == p.test_abort - 1000 && face == 0;
                if (!inject && __all((int)(v - value) >= 0)) break;
                if (spins > XS_SPINS || inject) {
                    if (lane == 0) {
                        s_abort = 1u;
                        __hip_atomic_store((fs_gu32*)p.abort_dev, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store((fs_gu32*)p.tmo, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        dead = s_abort != 0u;
    };
    // LayerNorm2d + FiLM of one row held by a group of 16 lanes: two-pass fp32 statistics, the bf16 copy is what is normalised
    // (utils.py:16-24, conditional_naf.py:114-115,126-127; same rounding points as hd_chain.hpp)
    constexpr int V4 = C / 64;                                       // float4 per lane per row
    auto ln_row = [&](const float4 (&v)[V4], const float* g, const float* b, char* dst_row, bool valid) __attribute__((always_inline)) {
        const int l16 = lane & 15;
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < V4; ++i) sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        const float mean = row16_sum(sum) * (1.0f / C);
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < V4; ++i) {
            const float d0 = v[i].x - mean, d1 = v[i].y - mean, d2 = v[i].z - mean, d3 = v[i].w - mean;
            q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        }
        const float rstd = 1.0f / sqrtf(row16_sum(q) * (1.0f / C) + p.ln_eps);
        const float nmr = -mean * rstd;
#pragma unroll
        for (int i = 0; i < V4; ++i) {
            const int k = 4 * l16 + 64 * i;
            const float x4[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float xq = bf16_bits_to_f32(f32_to_bf16_bits(x4[e]));
                o[e] = valid ? fmaf(fmaf(xq, rstd, nmr), g[k + e], b[k + e]) : 0.f;
            }
            *reinterpret_cast<uint2*>(dst_row + k * 2) = make_uint2(pack2(o[0], o[1]), pack2(o[2], o[3]));
        }
    };

    const int nb_run = p.block_limit < 0 ? 0 : (p.block_limit > 0 && p.block_limit < p.nblocks) ? p.block_limit : p.nblocks;
    uint4 bw[K::KS], bw2[K::KS];
    for (int blk = 0; blk < nb_run; ++blk) {
        const XBlockW& B = s_blk[blk];
        HD_FSTAMP(0);
        // ---- FiLM rows of both LayerNorms: requested by waves 1.. (wave 0 polls), copied to LDS behind the hand-off wait ----
        float4 gbv = make_float4(0.f, 0.f, 0.f, 0.f);                // [bias_att | gain_att | bias_ffn | gain_ffn] at film_off: 4C floats
        const int gi = tid - 64;
        static_assert(K::THREADS - 64 >= C, "one float4 of the FiLM rows per thread of waves 1..");
        if (gi >= 0 && gi < C) gbv = *reinterpret_cast<const float4*>(p.film + B.film_off + 4 * gi);
        face_load_b<C>(B.w1, tile, lane, bw);
        if constexpr (C == 128) face_load_b<C>(B.w1, tile + K::NT, lane, bw2);    // C = 256: one gate half at a time (registers)
        // per-channel constants of the depthwise stage: requested with the weights at C = 128; at C = 256 the two weight sets
        // already hold half the register file, so they (and the next GEMMs' weights) wait until conv1 has consumed them
        constexpr bool kEarly = C == 128;                            // (C = 256 with 16-row tiles and both halves up front: 107 spilled registers)
        float c_b1a = 0.f, c_b1b = 0.f, dwa[9], dwb[9], dba = 0.f, dbb = 0.f;
        auto load_dw_consts = [&]() __attribute__((always_inline)) {
            c_b1a = xs_ldg_f(B.b1 + col); c_b1b = xs_ldg_f(B.b1 + col + C);
#pragma unroll
            for (int t = 0; t < 9; ++t) { dwa[t] = xs_ldg_f(B.dw_w + (size_t)t * 2 * C + col); dwb[t] = xs_ldg_f(B.dw_w + (size_t)t * 2 * C + col + C); }
            dba = xs_ldg_f(B.dw_b + col); dbb = xs_ldg_f(B.dw_b + col + C);
        };
        if constexpr (kEarly) load_dw_consts();
        // ---- A: the neighbours' boundary rows of x' (previous block of this launch, or the previous launch) ----
        if (blk > 0) poll(base + 2u * (unsigned)(blk - 1) + 2u, 0x200u + (unsigned)blk);
        if (gi >= 0 && gi < C) *reinterpret_cast<float4*>(gb + 4 * gi) = gbv;   // gb[0..C) bias1, [C..2C) gain1, [2C..3C) bias2, [3C..4C) gain2
        __syncthreads();
        if (s_abort) return;
        {
            const int l16 = lane & 15;
            constexpr int ROWS = K::ROWS;
            constexpr int HRI = (ROWS - OWN + K::NT * 4 - 1) / (K::NT * 4);      // halo rows per 16-lane group
            constexpr int HSH = (K::NT * 4 - OWN % (K::NT * 4)) % (K::NT * 4);   // OWN = 16 at 8 waves: waves 0..3 take the own rows, 4..7 the halo rows
            float4 hv[HRI][V4];
            bool hok[HRI];
#pragma unroll
            for (int hi = 0; hi < HRI; ++hi) {                           // requested first: they arrive while the own rows are normalised
                const int r = OWN + hi * K::NT * 4 + (wave * 4 + (lane >> 4) + HSH) % (K::NT * 4);
                const bool up = r < OWN + K::S;
                hok[hi] = r < ROWS && (up ? has_up : has_dn);
                const int grow = up ? row0 - K::S + (r - OWN) : row0 + OWN + (r - OWN - K::S);
                const int gr = hok[hi] ? grow : row0;
                if (intro && blk == 0) {                               // computed at the entry (no launch wrote X)
                    const int hr = r < ROWS ? r - OWN : 0;
#pragma unroll
                    for (int i = 0; i < V4; ++i) hv[hi][i] = *reinterpret_cast<const float4*>(hb + hr * K::XROW + 4 * l16 + 64 * i);
                } else {
#pragma unroll
                    for (int i = 0; i < V4; ++i) {
                        const xs_u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(rs_X, (gr * C + 4 * l16 + 64 * i) * 4, 0, 16);
                        hv[hi][i] = make_float4(__uint_as_float(raw.x), __uint_as_float(raw.y), __uint_as_float(raw.z), __uint_as_float(raw.w));
                    }
                }
            }
            for (int r = wave * 4 + (lane >> 4); r < OWN; r += K::NT * 4) {
                float4 v[V4];
#pragma unroll
                for (int i = 0; i < V4; ++i) v[i] = *reinterpret_cast<const float4*>(xt + r * K::XROW + 4 * l16 + 64 * i);
                ln_row(v, gb + C, gb, smem + K::ALN_OFF + r * K::AROW, true);
            }
#pragma unroll
            for (int hi = 0; hi < HRI; ++hi) {
                const int r = OWN + hi * K::NT * 4 + (wave * 4 + (lane >> 4) + HSH) % (K::NT * 4);
                if (r < ROWS) ln_row(hv[hi], gb + C, gb, smem + K::ALN_OFF + r * K::AROW, hok[hi]);
            }
        }
        __syncthreads();
        HD_FSTAMP(1);
        // ---- conv1 on own + halo rows: column tile j and j + C/32 (the two SimpleGate halves) ----
        f32x16_t acc_a[K::MT1], acc_b[K::MT1];
        chain_mma<C, K::MT1>(smem + K::ALN_OFF, bw, lane, acc_a);
        if constexpr (kEarly) {
            chain_mma<C, K::MT1>(smem + K::ALN_OFF, bw2, lane, acc_b);
            face_load_b<C>(B.wsca, tile, lane, bw);      // the next GEMMs' weights fly during the depthwise stage
            face_load_b<C>(B.w3, tile, lane, bw2);
        } else {
            face_load_b<C>(B.w1, tile + K::NT, lane, bw);                         // second half's weights fly during the first half's depthwise pass
            load_dw_consts();
        }
        // ---- depthwise 3x3 (pad 1) of the own rows, half by half through the wave's LDS tile; SimpleGate ----
        HD_FSTAMP(2);
        constexpr int NU = (K::RI / 2) * K::S;                       // gate values per lane: image rows per lane half x face side
        float u1[NU], u2[NU];
        auto dw_half = [&](const f32x16_t (&acc)[K::MT1], float bias1, const float (&w)[9], float bias2, float (&u)[NU]) __attribute__((always_inline)) {
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int mt = 0; mt < K::MT1; ++mt)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int r = mt * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
                    t1w[r * 32 + (lane & 31)] = acc[mt][i] + bias1;
                }
            __builtin_amdgcn_wave_barrier();
            const int j = lane & 31, h = lane >> 5;
            constexpr int S = K::S, IRL = K::RI / 2;                  // image rows per lane: 1 (level 0) or 2 (level 1)
#pragma unroll
            for (int q = 0; q < IRL; ++q) {
                const int ir = h * IRL + q;                           // image row inside the workgroup's 32 rows
                // three image rows x (S + 2) columns, zero outside the face
                float v[3][S + 2];
#pragma unroll
                for (int rr = 0; rr < 3; ++rr) { v[rr][0] = 0.f; v[rr][S + 1] = 0.f; }
                const bool up_in = ir > 0, dn_in = ir < K::RI - 1;
                const bool up_ok = up_in || has_up, dn_ok = dn_in || has_dn;
                const int up_base = up_in ? (ir - 1) * S : OWN, dn_base = dn_in ? (ir + 1) * S : OWN + S;
#pragma unroll
                for (int x = 0; x < S; ++x) {
                    v[0][x + 1] = up_ok ? t1w[(up_base + x) * 32 + j] : 0.f;
                    v[1][x + 1] = t1w[(ir * S + x) * 32 + j];
                    v[2][x + 1] = dn_ok ? t1w[(dn_base + x) * 32 + j] : 0.f;
                }
#pragma unroll
                for (int x = 0; x < S; ++x) {
                    float a = bias2;
#pragma unroll
                    for (int rr = 0; rr < 3; ++rr) {
                        a = fmaf(w[rr * 3], v[rr][x], a); a = fmaf(w[rr * 3 + 1], v[rr][x + 1], a); a = fmaf(w[rr * 3 + 2], v[rr][x + 2], a);
                    }
                    u[q * S + x] = a;
                }
            }
        };
        dw_half(acc_a, c_b1a, dwa, dba, u1);
        if constexpr (!kEarly) {
            chain_mma<C, K::MT1>(smem + K::ALN_OFF, bw, lane, acc_b);
            face_load_b<C>(B.wsca, tile, lane, bw);                  // conv3's weights follow after the depthwise pass (registers)
        }
        dw_half(acc_b, c_b1b, dwb, dbb, u2);
        if constexpr (!kEarly) face_load_b<C>(B.w3, tile, lane, bw2);
        __syncthreads();                                             // every wave has read the halo rows of the LN tile: the gate tile overwrites them
        {
            const int j = lane & 31, h = lane >> 5;
            constexpr int IRL = K::RI / 2;
            float rsum = 0.f;
#pragma unroll
            for (int q = 0; q < IRL; ++q)
#pragma unroll
                for (int x = 0; x < K::S; ++x) {
                    const float g = u1[q * K::S + x] * u2[q * K::S + x];
                    rsum += g;
                    const int r = (h * IRL + q) * K::S + x;
                    *reinterpret_cast<unsigned short*>(smem + K::A1_OFF + r * K::AROW + (tile * 32 + j) * 2) = f32_to_bf16_bits(g);
                }
            const float other = __shfl_xor(rsum, 32);
            if (h == 0) poolp[tile * 32 + j] = rsum + other;          // rows of image-row group 0 first, then group 1
        }
        __syncthreads();
        HD_FSTAMP(3);
        // ---- B: channel sums of the face ----
        if (tid < C / 4) {
            const float4 v = *reinterpret_cast<const float4*>(poolp + 4 * tid);
            const xs_u32x4 x = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
            __builtin_amdgcn_raw_buffer_store_b128(x, rs_PP, ((face * K::CL + kk) * C + 4 * tid) * 4, 0, 16);
        }
        publish(base + 2u * (unsigned)blk + 1u);
        wait_for(base + 2u * (unsigned)blk + 1u, 0x300u + (unsigned)blk);
        if (dead) return;
        if (tid < C / 4) {
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int m = 0; m < K::CL; ++m) {                          // fixed order: every workgroup of the face gets the same bits
                const xs_u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(rs_PP, ((face * K::CL + m) * C + 4 * tid) * 4, 0, 16);
                s.x += __uint_as_float(raw.x); s.y += __uint_as_float(raw.y); s.z += __uint_as_float(raw.z); s.w += __uint_as_float(raw.w);
            }
            const float inv = 1.0f / (float)K::HW;
            *reinterpret_cast<float4*>(pooled + 4 * tid) = make_float4(s.x * inv, s.y * inv, s.z * inv, s.w * inv);
        }
        const float c_bsca = xs_ldg_f(B.bsca + col), c_b3 = xs_ldg_f(B.b3 + col), c_beta = xs_ldg_f(B.beta + col), c_b4a = xs_ldg_f(B.b4 + col),
                    c_b4b = xs_ldg_f(B.b4 + col + C), c_b5 = xs_ldg_f(B.b5 + col), c_gamma = xs_ldg_f(B.gamma + col);
        __syncthreads();
        HD_FSTAMP(4);
        // ---- SCA: s = Wsca * pooled + b on the MFMA (row 0 of the A tile holds the pooled vector) ----
        {
            f32x16_t acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
            for (int ks = 0; ks < K::KS; ++ks) {
                const float* q = pooled + ks * 16 + 8 * (lane >> 5);
                const float4 v0 = *reinterpret_cast<const float4*>(q), v1 = *reinterpret_cast<const float4*>(q + 4);
                const uint4 r0 = make_uint4(pack2(v0.x, v0.y), pack2(v0.z, v0.w), pack2(v1.x, v1.y), pack2(v1.z, v1.w));
                const uint4 a = (lane & 31) == 0 ? r0 : make_uint4(0, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, bw[ks]), acc, 0, 0, 0);
            }
            if (lane < 32) s_vec[col] = acc[0] + c_bsca;
        }
        face_load_b<C>(B.w4, tile, lane, bw);
        __syncthreads();
        // ---- A1 <- bf16(bf16(g) * s) in place ----
        for (int u = tid; u < OWN * (C / 8); u += K::THREADS) {
            const int r = u / (C / 8), q = u - r * (C / 8);
            uint4* gp = reinterpret_cast<uint4*>(smem + K::A1_OFF + r * K::AROW + q * 16);
            float v[8];
            unpack8(*gp, v);
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] *= s_vec[q * 8 + i];
            *gp = pack8(v);
        }
        __syncthreads();
        // ---- conv3 -> y = x + beta * (acc + b3), in place over x ----
        {
            f32x16_t acc[1];
            chain_mma<C, 1>(smem + K::A1_OFF, bw2, lane, acc);
            face_load_b<C>(B.w4, tile + K::NT, lane, bw2);
#pragma unroll
            for (int i = 0; i < OWN / 2; ++i) {                           // tile rows (i & 3) + 8 (i >> 2) + 4 (lane >> 5) < OWN
                const int r = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
                xt[r * K::XROW + col] = xt[r * K::XROW + col] + (acc[0][i] + c_b3) * c_beta;
            }
        }
        __syncthreads();
        // ---- LayerNorm + FiLM on y -> rows 0..31 of the LN tile ----
        {
            const int l16 = lane & 15;
            for (int r = wave * 4 + (lane >> 4); r < OWN; r += K::NT * 4) {
                float4 v[V4];
#pragma unroll
                for (int i = 0; i < V4; ++i) v[i] = *reinterpret_cast<const float4*>(xt + r * K::XROW + 4 * l16 + 64 * i);
                ln_row(v, gb + 3 * C, gb + 2 * C, smem + K::ALN_OFF + r * K::AROW, true);
            }
        }
        __syncthreads();
        // ---- conv4 (both halves) -> SimpleGate -> A1 ----
        {
            f32x16_t a1[1], a2[1];
            chain_mma<C, 1>(smem + K::ALN_OFF, bw, lane, a1);
            chain_mma<C, 1>(smem + K::ALN_OFF, bw2, lane, a2);
            face_load_b<C>(B.w5, tile, lane, bw);
#pragma unroll
            for (int i = 0; i < OWN / 2; ++i) {
                const int r = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
                *reinterpret_cast<unsigned short*>(smem + K::A1_OFF + r * K::AROW + col * 2) = f32_to_bf16_bits((a1[0][i] + c_b4a) * (a2[0][i] + c_b4b));
            }
        }
        __syncthreads();
        // ---- conv5 -> x' = y + gamma * (acc + b5), in place ----
        {
            f32x16_t acc[1];
            chain_mma<C, 1>(smem + K::A1_OFF, bw, lane, acc);
#pragma unroll
            for (int i = 0; i < OWN / 2; ++i) {
                const int r = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
                xt[r * K::XROW + col] = xt[r * K::XROW + col] + (acc[0][i] + c_b5) * c_gamma;
            }
        }
        __syncthreads();
        HD_FSTAMP(5);
        // ---- x' leaves: fp32 rows for the neighbours' halo (and the next launch); at the end of the run the bf16 / gated copies ----
        const bool last = blk == nb_run - 1;
        int row0e = row0, facee = face;                              // opaque here: the exit addresses are formed at the exit, not hoisted to the kernel's start and spilled
        int tide = tid;
        asm volatile("" : "+s"(row0e), "+s"(facee), "+v"(tide));
        for (int u = tide; u < OWN * (C / 8); u += K::THREADS) {
            const int r = u / (C / 8), q = u - r * (C / 8);
            const float* xv = xt + r * K::XROW + q * 8;
            const float4 a = *reinterpret_cast<const float4*>(xv), b = *reinterpret_cast<const float4*>(xv + 4);
            const int off = ((row0 + r) * C + q * 8) * 4;
            __builtin_amdgcn_raw_buffer_store_b128((xs_u32x4){__float_as_uint(a.x), __float_as_uint(a.y), __float_as_uint(a.z), __float_as_uint(a.w)}, rs_X, off, 0, 16);
            __builtin_amdgcn_raw_buffer_store_b128((xs_u32x4){__float_as_uint(b.x), __float_as_uint(b.y), __float_as_uint(b.z), __float_as_uint(b.w)}, rs_X, off + 16, 0, 16);
            if (last) {
                const float x8[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
                const size_t o = (size_t)(row0e + r) * C + q * 8;
                if (p.Xb) *reinterpret_cast<uint4*>(p.Xb + o) = pack8(x8);
                if (p.outg16 && blk == p.nblocks - 1) {                // f_d * (1 + w_c + w_s): the HCA conv input (hca.py:28)
                    float gv[8];
                    const float gsr = p.gate_s[row0e + r];
                    const float* gc = p.gate_c + (size_t)facee * C + q * 8;
#pragma unroll
                    for (int i = 0; i < 8; ++i) gv[i] = x8[i] * (1.0f + gc[i] + gsr);
                    *reinterpret_cast<uint4*>(p.outg16 + o) = pack8(gv);
                }
            }
        }
        if (!last) publish(base + 2u * (unsigned)blk + 2u);
        HD_FSTAMP(6);
    }
    if constexpr ((C == 128 && OWN == 32) || (C == 256 && OWN == 16)) {
        if (nb_run == 0) {                                           // introspection (block_limit < 0): the entry's x as the stage holds it
            int row0e = row0, tide = tid;                            // opaque: the addresses are formed here, not at the kernel's start (hd_xcd.hpp's exit)
            asm volatile("" : "+s"(row0e), "+v"(tide));
            for (int u = tide; u < OWN * (C / 4); u += K::THREADS) {
                const int r = u / (C / 4), q = u - r * (C / 4);
                const float4 v = *reinterpret_cast<const float4*>(xt + r * K::XROW + q * 4);
                const size_t o = (size_t)(row0e + r) * C + q * 4;
                *reinterpret_cast<float4*>(p.X + o) = v;
                if (p.Xb) *reinterpret_cast<uint2*>(p.Xb + o) = make_uint2(pack2(v.x, v.y), pack2(v.z, v.w));
            }
        }
    }
    if (kk == 0 && tid == 0) __hip_atomic_store((fs_gu32*)(p.gstate + face * 16), base / 64u + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int C, int OWN = 32>
inline hipError_t launch_face_stage(const FStageP& p, hipStream_t s) {
    typedef FaceCfg<C, OWN> K;
    if (p.B < 1 || p.B > 64 || p.nblocks < 1 || p.nblocks > XS_MAXBLK) return hipErrorInvalidValue;
    static std::atomic<unsigned long long> granted{0};
        { const hipError_t e = grant_dynamic_lds(reinterpret_cast<const void*>(&naf_face_stage_kernel<C, OWN>), K::SMEM, granted); if (e != hipSuccess) return e; }
    hipLaunchKernelGGL((naf_face_stage_kernel<C, OWN>), dim3(64 * K::CL), dim3(K::THREADS), K::SMEM, s, p);
    return hipGetLastError();
}

}  // namespace hd
