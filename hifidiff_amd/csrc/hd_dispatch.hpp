// hd_dispatch.hpp -- the GEMM launch table, split by loader family so that the kernel instantiations compile in parallel
// translation units (hd_dispatch_*.hip; one hipcc process each, linked into libhifidiff_hip.so).
#pragma once
#include <type_traits>
#include "hd_gemm.hpp"
#include "hd_wide.hpp"

namespace hd {

enum LdKind { LK_F32, LK_LN, LK_BF16, LK_BF16S, LK_CONV_F32, LK_CONV_F32G, LK_CONV_BF16 };
enum EpKind { EK_BIASF32, EK_RESID, EK_GATE, EK_PIXSHUF, EK_BIASBF16, EK_DWGATE, EK_SCA };

// mode: 0 = tall T128, 1 = tall T64, 2 = skinny 64 rows, 3 = skinny 32 rows, 4 = tall T32W (32 rows x 256 cols),
//       5 / 6 = skinny with 4 / 8 M-split waves (128 / 256 rows per workgroup; long-K gathers at large M)
//       + 16: the deep-prefetch tall kernel where the loader / epilogue pair has it and the shape fits (latent 32: levels 2 / 3)
//       + 32: the role-split wide kernel (hd_wide.hpp) for LayerNorm -> gate and bf16 -> residual at 1024 rows x K = 1024 / 4096 rows x K = 512 (latent 32: levels 3 / 2)
template <class LD, class EP, bool PAIR>
hipError_t launch_tile(const GemmP& p, int mode, hipStream_t s) {
    constexpr bool kWideLN = std::is_same<LD, LdF32LN_T<false>>::value && std::is_same<EP, EpGateBF16>::value && PAIR;
    constexpr bool kWidePlain = std::is_same<LD, LdBF16Plain>::value && std::is_same<EP, EpResidF32>::value && !PAIR;
    if constexpr (kWideLN || kWidePlain) {
        // (the 256-row form measured no gain for these two: 15.9 against 15.8 us and 12.9 against 12.4 us at level 2; tools/deep_bench)
        // (the 64-row form: the LayerNorm pair GEMM of the middle level only)
        const int form = (mode & 32) ? wide_shape_ok<PAIR>(p) : 0;
        if ((form == 1 || (form == 3 && kWideLN)) && (!kWideLN || wide_stats_ok<PAIR>(p))) return launch_gemm_wide<kWideLN, EP, PAIR>(p, s);
    }
    if constexpr (ld_is_deep<LD>::value && ep_is_deep<EP>::value) {
        if ((mode & 16) && deep_shape_ok<PAIR>(p)) return launch_gemm_deep<LD, EP, PAIR>(p, s);
    }
    mode &= 15;
    if constexpr (PAIR) {
        switch (mode) {
            case 0: return launch_gemm<T128P, LD, EP>(p, s);
            case 1: return launch_gemm<T64P, LD, EP>(p, s);
            case 2: return launch_skinny_auto<1, 2, true, LD, EP>(p, s);
            case 4: return launch_gemm<T32WP, LD, EP>(p, s);
            case 5: return launch_skinny_auto<4, 1, true, LD, EP>(p, s);
            case 6: return launch_skinny_auto<8, 1, true, LD, EP>(p, s);
            default: return launch_skinny_auto<1, 1, true, LD, EP>(p, s);
        }
    } else {
        switch (mode) {
            case 0: return launch_gemm<T128, LD, EP>(p, s);
            case 1: return launch_gemm<T64, LD, EP>(p, s);
            case 2: return launch_skinny_auto<1, 2, false, LD, EP>(p, s);
            case 4: return launch_gemm<T32W, LD, EP>(p, s);
            case 5: return launch_skinny_auto<4, 1, false, LD, EP>(p, s);
            case 6: return launch_skinny_auto<8, 1, false, LD, EP>(p, s);
            default: return launch_skinny_auto<1, 1, false, LD, EP>(p, s);
        }
    }
}

// conv1 with the depthwise 3x3 + SimpleGate + pool fused: workgroup = whole faces (BM = max(32, hw) rows)
template <class LN>
hipError_t dispatch_dwgate(const GemmP& p, hipStream_t s) {
    // many rows (latent 32: levels 2 / 3, M = 4096 / 1024): the 128-row deep-prefetch tile with the same epilogue -- every 128-row tile is whole faces
    if constexpr (std::is_same<LN, LdF32LN_T<false>>::value) {      // level 3 of latent 32: the role-split wide kernel with the same tile epilogue (hd_wide.hpp)
        static const bool no_wide = hd_env("HD_NO_WIDE") != nullptr;
        const int form = wide_shape_ok<true>(p);
        if (!no_wide && form && wide_stats_ok<true>(p) && p.hw >= 4 && wide_form_rows(form) % p.hw == 0 && p.side * p.side == p.hw)
            return launch_gemm_wide<true, EpDwGate, true>(p, s);
    }
    if constexpr (ld_is_deep<LN>::value) {
        static const bool no_deep_dw = hd_env("HD_NO_DEEP_DW") != nullptr;
        if (!no_deep_dw && deep_shape_ok<true>(p) && p.M % 128 == 0 && p.hw >= 4 && p.hw <= 128 && 128 % p.hw == 0 && p.side * p.side == p.hw)
            return launch_gemm_deep<LN, EpDwGate, true>(p, s);
    }
    // more than two 32-row workgroups per CU: 64-row tiles halve the weight re-reads (latent 32, levels 3 / middle).  At K = 2048
    // (latent 32, middle level: 256 rows) already from two per CU on: 512 workgroups of 256 + 128 KB against 256 of 256 + 256 KB
    if (p.hw <= 32 && 64 % p.hw == 0 && p.M % 64 == 0 && (p.M / 32) * (p.N / 64) >= (p.Kp >= 2048 ? 512 : 1024)) return launch_skinny_auto<1, 2, true, LN, EpDwGate>(p, s);
    if (p.hw <= 32) return launch_skinny_auto<1, 1, true, LN, EpDwGate>(p, s);
    static const bool big64 = hd_env("HD_NO_DW64_WM8") == nullptr;      // 256-row tiles (4 faces of 8x8) when 64-row tiles would put >= 4 workgroups on a CU
    if (p.hw == 64 && big64 && p.M % 256 == 0 && (p.M / 64) * (p.N / 64) >= 1024) return launch_skinny_auto<8, 1, true, LN, EpDwGate>(p, s);
    if (p.hw == 64) return launch_skinny_auto<2, 1, true, LN, EpDwGate>(p, s);
    if (p.hw == 256) return launch_skinny_auto<8, 1, true, LN, EpDwGate>(p, s);
    return hipErrorInvalidValue;
}

// LayerNorm GEMMs (LK_LN): which FiLM source the loader reads is a property of the kernel (LdF32LN: one row for all faces, in
// LDS; LdF32LNFace: per-face timesteps, rows of the global table) -- chosen by the caller, never inside the K loop
template <class LN>
hipError_t dispatch_ln(const GemmP& p, EpKind ek, int mode, hipStream_t s) {
    if (ek == EK_DWGATE) {
        static const bool no_dw1 = hd_env("HD_NO_DW1") != nullptr;
        if (p.hw == 1 && !no_dw1) return launch_skinny_auto<1, 1, true, LN, EpDwGate1>(p, s);     // one pixel per face: element-wise
        return dispatch_dwgate<LN>(p, s);
    }
    if (ek == EK_BIASF32) return launch_tile<LN, EpBiasF32, false>(p, mode, s);
    if (ek == EK_GATE) return launch_tile<LN, EpGateBF16, true>(p, mode, s);
    return hipErrorInvalidValue;
}

hipError_t dispatch_gemm_ln_shared(const GemmP& p, EpKind ek, int mode, hipStream_t s);     // hd_dispatch_ln.hip
hipError_t dispatch_gemm_ln_face(const GemmP& p, EpKind ek, int mode, hipStream_t s);       // hd_dispatch_lnface.hip
hipError_t dispatch_gemm_bf16(const GemmP& p, LdKind lk, EpKind ek, int mode, hipStream_t s);   // hd_dispatch_bf16.hip: LK_BF16, LK_BF16S
hipError_t dispatch_gemm_misc(const GemmP& p, LdKind lk, EpKind ek, int mode, hipStream_t s);   // hd_dispatch_misc.hip: LK_F32, LK_CONV_BF16

}  // namespace hd
