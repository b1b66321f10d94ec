// hd_fm.hpp — EXPERIMENT, not part of libhifidiff_hip.so (see README.md in this directory for the measurement).
//
// The SimpleGate pair GEMMs of the middle level (conv1, conv4: LayerNorm + FiLM -> 2048 -> 4096 -> gate;
// models/denoiser/conditional_naf.py:114-118,126-131) at one pixel per face, "face-major": every workgroup owns
// 8 + 8 output channels (both gate halves) for ALL faces, so each weight byte is fetched by exactly one CU (64 KiB per
// workgroup, non-temporal) and the activations (<= 64 rows x 2048, 256 KiB, the same bytes for every workgroup) come
// out of L2.  The MFMA runs transposed: A operand = 16 weight rows x 32 k, B operand = 32 k x 16 faces
// (v_mfma_f32_16x16x32_bf16), four face groups per k-step, K split over the 8 waves; both operands are loaded straight
// into registers in fragment order.  It was wired into dispatch_gemm for (LK_LN, EK_DWGATE, hw == 1) and (LK_LN, EK_GATE)
// at M <= 64, K = 2048, N = 4096 with a second packing of the weights (pack_weight_fm_kernel).
#pragma once
#include "../../hifidiff_amd/csrc/hd_gemm.hpp"

namespace hd {

typedef __attribute__((ext_vector_type(4))) float fm_f32x4;
typedef unsigned fm_u32x4 __attribute__((ext_vector_type(4)));

constexpr int FM_K = 2048, FM_C = 2048;            // K of the GEMM, gate channels (N = 2 * FM_C)
constexpr int FM_WG = FM_C / 8;                    // 256 workgroups: 8 gate channels each

struct FmP {
    int M;                                         // faces (rows), <= 64
    const uint4* W;                                // [256 tiles][K/32][64 lanes]: lane l holds W[row r][k = 32 ks + 8 (l>>4) + j], row r = l & 15:
                                                   //   r < 8: channel 8 t + r, r >= 8: channel C + 8 t + (r - 8)   (pack_weight_fm_kernel)
    const unsigned short* A; int lda;              // bf16 copy of the residual stream [M][K]
    const float2* stats_in; int stats_np, stats_cnt;
    const float* film; int film_gain_off, film_bias_off;
    float ln_eps;
    const float* bias;                             // [2C] conv bias
    const float *dw_c, *dw_b;                      // conv1: depthwise centre tap [2C] and bias [2C]; conv4: null
    unsigned short* out; int ldo;                  // G (bf16) [M][C]
    float* pooled; unsigned short* pooled16;       // conv1 only
};

struct FmLds {
    float red[8][4][256];                          // [wave][face group][16 rows x 16 faces]   32 KiB
    float gb[2][FM_K];                             // FiLM gain / bias                         16 KiB
    float2 st[64];                                 // per face: (-mean * rstd, rstd)
};

template <bool DW>
__global__ __launch_bounds__(512) void fm_pair_kernel(const FmP p) {
    __shared__ __attribute__((aligned(16))) FmLds L;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t = blockIdx.x;
    const int fl = lane & 15, kq = lane >> 4;
    const uint4* Wl = p.W + ((size_t)t * (FM_K / 32) + 8 * wave) * 64 + lane;
    fm_u32x4 wf[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) wf[s] = __builtin_nontemporal_load(reinterpret_cast<const fm_u32x4*>(Wl + s * 64));
    fm_u32x4 af[8][4];
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int f = 16 * g + fl;
            af[s][g] = (fm_u32x4){0u, 0u, 0u, 0u};
            if (f < p.M) af[s][g] = *reinterpret_cast<const fm_u32x4*>(p.A + (size_t)f * p.lda + 256 * wave + 32 * s + 8 * kq);
        }
    {   // LayerNorm statistics of all faces: 8 threads per face merge the partials (equal counts); FiLM row to LDS
        const int f = tid >> 3, part = tid & 7, per = p.stats_np >> 3;
        float sm = 0.f;
        float2 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            v[i] = make_float2(0.f, 0.f);
            if (i < per && f < p.M) v[i] = p.stats_in[(size_t)f * p.stats_np + part * per + i];
            sm += v[i].x;
        }
        sm += __shfl_xor(sm, 1); sm += __shfl_xor(sm, 2); sm += __shfl_xor(sm, 4);
        const float mean = sm / (float)p.stats_np;
        const float cnt = (float)p.stats_cnt;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) { const float d = v[i].x - mean; if (i < per) q += fmaf(cnt * d, d, v[i].y); }
        q += __shfl_xor(q, 1); q += __shfl_xor(q, 2); q += __shfl_xor(q, 4);
        const float var = q / ((float)p.stats_np * cnt);
        const float rstd = (f < p.M) ? __frsqrt_rn(var + p.ln_eps) : 0.f;
        if (part == 0) L.st[f] = make_float2(-mean * rstd, rstd);
        const float4 g4 = *reinterpret_cast<const float4*>(p.film + p.film_gain_off + 4 * tid);
        const float4 b4 = *reinterpret_cast<const float4*>(p.film + p.film_bias_off + 4 * tid);
        *reinterpret_cast<float4*>(&L.gb[0][4 * tid]) = g4;
        *reinterpret_cast<float4*>(&L.gb[1][4 * tid]) = b4;
    }
    __syncthreads();
    float2 fs[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) fs[g] = L.st[16 * g + fl];
    fm_f32x4 acc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = (fm_f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int k0 = 256 * wave + 32 * s + 8 * kq;
        const float4 g0 = *reinterpret_cast<const float4*>(&L.gb[0][k0]), g1 = *reinterpret_cast<const float4*>(&L.gb[0][k0 + 4]);
        const float4 b0 = *reinterpret_cast<const float4*>(&L.gb[1][k0]), b1 = *reinterpret_cast<const float4*>(&L.gb[1][k0 + 4]);
        const bf16x8_t wfrag = __builtin_bit_cast(bf16x8_t, wf[s]);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float v[8];
            unpack8(make_uint4(af[s][g].x, af[s][g].y, af[s][g].z, af[s][g].w), v);
            const float mu = fs[g].x, rs = fs[g].y;
            v[0] = fmaf(fmaf(v[0], rs, mu), g0.x, b0.x); v[1] = fmaf(fmaf(v[1], rs, mu), g0.y, b0.y);
            v[2] = fmaf(fmaf(v[2], rs, mu), g0.z, b0.z); v[3] = fmaf(fmaf(v[3], rs, mu), g0.w, b0.w);
            v[4] = fmaf(fmaf(v[4], rs, mu), g1.x, b1.x); v[5] = fmaf(fmaf(v[5], rs, mu), g1.y, b1.y);
            v[6] = fmaf(fmaf(v[6], rs, mu), g1.z, b1.z); v[7] = fmaf(fmaf(v[7], rs, mu), g1.w, b1.w);
            acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfrag, __builtin_bit_cast(bf16x8_t, pack8(v)), acc[g], 0, 0, 0);
        }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) L.red[wave][g][(4 * kq + i) * 16 + fl] = acc[g][i];     // D[row = 4 kq + i][face = fl]
    __syncthreads();
    const int f = tid >> 3, c = tid & 7;                                                   // one (face, gate channel) per thread
    float v1 = 0.f, v2 = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) { v1 += L.red[w][f >> 4][c * 16 + (f & 15)]; v2 += L.red[w][f >> 4][(8 + c) * 16 + (f & 15)]; }
    if (f >= p.M) return;
    const int ch = 8 * t + c;
    float g;
    if (DW) {
        const float wa = p.dw_c[ch], wb = p.dw_c[ch + FM_C];
        const float A1 = p.dw_b[ch] + wa * p.bias[ch], A2 = p.dw_b[ch + FM_C] + wb * p.bias[ch + FM_C];
        g = fmaf(wa, v1, A1) * fmaf(wb, v2, A2);
    } else {
        g = (v1 + p.bias[ch]) * (v2 + p.bias[ch + FM_C]);
    }
    const size_t o = (size_t)f * p.ldo + ch;
    p.out[o] = f32_to_bf16_bits(g);
    if (DW) { p.pooled[o] = g; if (p.pooled16) p.pooled16[o] = f32_to_bf16_bits(g); }
}

// [2C][K] fp32 1x1 weight -> the fragment order of FmP::W
struct PackFmP { const float* src; uint4* dst; int C, K; };
__global__ void pack_weight_fm_kernel(const PackFmP p) {
    const int ksteps = p.K >> 5;
    const size_t total = (size_t)(p.C >> 3) * ksteps * 64;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int lane = (int)(e & 63);
        const size_t tk = e >> 6;
        const int ks = (int)(tk % ksteps), t = (int)(tk / ksteps);
        const int r = lane & 15;
        const int n = r < 8 ? 8 * t + r : p.C + 8 * t + (r - 8);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = p.src[(size_t)n * p.K + ks * 32 + 8 * (lane >> 4) + j];
        p.dst[e] = pack8(v);
    }
}

template <bool DW>
inline hipError_t launch_fm_pair(const FmP& p, hipStream_t s) {
    if (p.M < 1 || p.M > 64 || (p.stats_np != 64 && p.stats_np != 16 && p.stats_np != 8)) return hipErrorInvalidValue;
    hipLaunchKernelGGL((fm_pair_kernel<DW>), dim3(FM_WG), dim3(512), 0, s, p);
    return hipGetLastError();
}

}  // namespace hd
