// hd_strip.hpp — the attention half's front of a (Conditional)NAFBlock on 32 x 32 faces with C = 128 (latent 32, level 0, and the
// FPG's first level there) as ONE launch: LayerNorm2d + FiLM -> conv1 (1x1, C -> 2C) -> depthwise 3x3 (pad 1) -> SimpleGate -> G,
// plus the per-strip channel sums of the SCA pool (models/denoiser/conditional_naf.py:114-119, models/naf.py:108-113).  gfx950 only.
//
// The fused GEMM epilogue that does this at the other levels needs a whole face inside one workgroup's rows (<= 256 pixels);
// a 32 x 32 face has 1024, so this level ran as three launches with T1 (fp32 [rows][2C], 67 MB at batch 64) written and read
// back in between: 55 us per block.  Here a workgroup owns a STRIP of 4 image rows (128 pixel rows) of one face and all 2C
// conv1 columns (4 waves x (tile j, tile j + 4)), and recomputes conv1 for the image row above and below its strip (6 MFMA row
// tiles for 4: MFMA time is not what bounds this kernel).  With a face side of 32 an MFMA 32 x 32 accumulator IS one image row
// of one channel per lane pair: lane (j, h) holds x = 8b + 4h + (0..3), b = 0..3, of column j.  The vertical taps are therefore
// the same register of the neighbouring accumulator tile, the horizontal ones the neighbouring register or, at the 4-pixel
// group edges, a register of lane ^ 32 — the depthwise conv runs out of the accumulators, T1 never exists, not even in LDS.
//
// LayerNorm input as in the GEMM loader it replaces (hd_gemm.hpp: LdF32LN_T): statistics exact fp32 from the producer's
// (mean, M2) partials (equal counts: two short sums), the value normalised is the bf16 copy Xb; same rounding points.
// Traffic per block: Xb 1.5 x 16.8 MB in (halo rows), G 16.8 MB out, against 117 MB for the three launches.
#pragma once
#include <type_traits>
#include "hd_chain.hpp"
#include "hd_stage_api.hpp"

namespace hd {

struct StripCfg {
    static constexpr int C = 128, S = 32, HW = S * S;
    static constexpr int RI = 4, OWN = RI * S, ROWS = OWN + 2 * S, MT = ROWS / 32;   // image rows per strip; own rows; with halo; MFMA row tiles
    static constexpr int NSTRIP = S / RI;
    static constexpr int NT = C / 32, THREADS = 64 * NT, KS = C / 16;
    static constexpr int AROW = ChainCfg<C>::AROW;               // bytes per bf16 tile row (padded): chain_mma's layout
    static constexpr int ALN_OFF = 0;                            // LayerNorm output bf16 [ROWS][AROW]: image row above | own rows | image row below;
                                                                 // later the gate tile bf16 [OWN][AROW] on its way out
    static constexpr int SMEM = ALN_OFF + ROWS * AROW;
    static_assert(MT == 6 && NT == 4 && NSTRIP == 8, "one accumulator tile per image row");
};

// value of lane ^ 32 (the other 4-pixel group of the same column)
__device__ __forceinline__ float strip_other_half(float v) { return __shfl_xor(v, 32); }

// conv1 of the six row tiles against one column tile: the A fragments of one k-step are read while the previous step's MFMAs
// run.  (chain_mma leaves the order to the compiler, which reads all 48 fragments -- 192 registers -- ahead of the first MFMA.)
__device__ __forceinline__ void strip_mma(const char* sA, const uint4* b, int lane, f32x16_t (&acc)[StripCfg::MT]) {
    typedef StripCfg K;
    const char* ap = sA + (lane & 31) * K::AROW + (lane >> 5) * 16;
#pragma unroll
    for (int mt = 0; mt < K::MT; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[mt][i] = 0.f;
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    u32x4_t a[2][K::MT];
#pragma unroll
    for (int mt = 0; mt < K::MT; ++mt) a[0][mt] = *reinterpret_cast<const u32x4_t*>(ap + mt * 32 * K::AROW);
#pragma unroll
    for (int ks = 0; ks < K::KS; ++ks) {
        if (ks + 1 < K::KS) {
#pragma unroll
            for (int mt = 0; mt < K::MT; ++mt) a[(ks + 1) & 1][mt] = *reinterpret_cast<const u32x4_t*>(ap + mt * 32 * K::AROW + (ks + 1) * 32);
        }
#pragma unroll
        for (int mt = 0; mt < K::MT; ++mt) {
            asm volatile("" : "+v"(a[ks & 1][mt]) :: "memory");        // this step's fragments are consumed here, not hoisted reads of later steps
            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a[ks & 1][mt]), __builtin_bit_cast(bf16x8_t, b[ks]), acc[mt], 0, 0, 0);
        }
    }
}

__global__ __launch_bounds__(StripCfg::THREADS, 2) void naf_strip_dwgate_kernel(const StripP p) {
    typedef StripCfg K;
    constexpr int C = K::C;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // workgroups are dealt round-robin to the 8 XCDs: give each XCD a contiguous run of (face, strip) pairs, so that the halo rows
    // a strip re-reads are the own rows of a neighbour on the same L2 (gridDim.x = 8 * faces)
    const int lid = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int face = lid / K::NSTRIP, kk = lid - face * K::NSTRIP;
    const bool has_up = kk > 0, has_dn = kk < K::NSTRIP - 1;
    const int row_first = face * K::HW + kk * K::OWN - K::S;      // global row of local row 0 (the image row above the strip)
    const int tile = wave, col = tile * 32 + (lane & 31);

    uint4 bw[K::KS];
    chain_load_b<C>(p.W1, tile, lane, bw);
    // ---- LayerNorm2d + FiLM of own + halo rows: 16 lanes per row, 8 channels per lane ----
    {
        const int l16 = lane & 15, g16 = tid >> 4;                    // 16 row groups per pass
        constexpr int PASSES = K::ROWS / 16;
        const float* f = p.film + (size_t)(p.face0 + face) * p.film_face_stride;
        float gn[8], bs[8];
        {
            const float4 g0 = *reinterpret_cast<const float4*>(f + p.film_gain_off + 8 * l16), g1 = *reinterpret_cast<const float4*>(f + p.film_gain_off + 8 * l16 + 4);
            const float4 b0 = *reinterpret_cast<const float4*>(f + p.film_bias_off + 8 * l16), b1 = *reinterpret_cast<const float4*>(f + p.film_bias_off + 8 * l16 + 4);
            gn[0] = g0.x; gn[1] = g0.y; gn[2] = g0.z; gn[3] = g0.w; gn[4] = g1.x; gn[5] = g1.y; gn[6] = g1.z; gn[7] = g1.w;
            bs[0] = b0.x; bs[1] = b0.y; bs[2] = b0.z; bs[3] = b0.w; bs[4] = b1.x; bs[5] = b1.y; bs[6] = b1.z; bs[7] = b1.w;
        }
        uint4 raw[PASSES];
        float2 part[PASSES];
        const int np = p.stats_np;
#pragma unroll
        for (int it = 0; it < PASSES; ++it) {                          // every load first: one round trip for the whole tile
            const int r = it * 16 + g16;
            const bool ok = (r >= K::S || has_up) && (r < K::S + K::OWN || has_dn);
            const int row = ok ? row_first + r : row_first + K::S;
            raw[it] = *reinterpret_cast<const uint4*>(p.Xb + (size_t)row * C + 8 * l16);
            part[it] = l16 < np ? p.stats_in[(size_t)row * np + l16] : make_float2(0.f, 0.f);
        }
        const float inv_np = 1.0f / (float)np, cnt = (float)p.stats_cnt;
#pragma unroll
        for (int it = 0; it < PASSES; ++it) {
            const int r = it * 16 + g16;
            const bool ok = (r >= K::S || has_up) && (r < K::S + K::OWN || has_dn);
            // equal-count partials: mean = sum(mean_i) / np, M2 = sum(M2_i + cnt (mean_i - mean)^2)   (hd_gemm.hpp: block_finish)
            const float mean = row16_sum(part[it].x) * inv_np;
            const float d = part[it].x - mean;
            const float q = l16 < np ? fmaf(cnt * d, d, part[it].y) : 0.f;
            const float var = row16_sum(q) * (inv_np / cnt);
            float rstd = __frsqrt_rn(var + p.ln_eps), mu = -mean * rstd;
            asm volatile("" : "+v"(rstd), "+v"(mu));                  // two registers of their own (hd_gemm.hpp: unit_stats)
            float v[8];
            unpack8(raw[it], v);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaf(fmaf(v[e], rstd, mu), gn[e], bs[e]);
            *reinterpret_cast<uint4*>(smem + K::ALN_OFF + r * K::AROW + 16 * l16) = ok ? pack8(v) : make_uint4(0, 0, 0, 0);
        }
    }
    // per-channel constants of both gate halves
    float c_b1 = p.b1[col], c_db = p.dw_b[col], w[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) w[t] = p.dw_wT[(size_t)t * 2 * C + col];
    __syncthreads();

    // depthwise 3x3 of the four own image rows out of the six accumulator tiles of one gate half
    const bool hi = lane >= 32;
    // second == false: u <- depthwise output; second == true: u <- u * depthwise output (SimpleGate)
    auto dw_half = [&](f32x16_t (&acc)[K::MT], float bias1, float bias2, float (&u)[K::RI][16], auto second) __attribute__((always_inline)) {
        // T1 = conv1 + bias inside the face, 0 outside (the conv's zero padding)
#pragma unroll
        for (int m = 0; m < K::MT; ++m) {
            const bool ok = (m > 0 || has_up) && (m < K::MT - 1 || has_dn);
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][i] = ok ? acc[m][i] + bias1 : 0.f;
        }
        // the pixels left of register 4b and right of register 4b+3 live in lane ^ 32:
        //   h = 0 (x = 8b..8b+3):   left = other[4(b-1)+3] (0 at b = 0), right = other[4b]
        //   h = 1 (x = 8b+4..8b+7): left = other[4b+3],                   right = other[4(b+1)] (0 at b = 3)
        float el[K::MT][4], er[K::MT][4];
        auto edges = [&](int m) __attribute__((always_inline)) {
            float o0[4], o3[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) { o0[b] = strip_other_half(acc[m][4 * b]); o3[b] = strip_other_half(acc[m][4 * b + 3]); }
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                el[m][b] = hi ? o3[b] : (b > 0 ? o3[b > 0 ? b - 1 : 0] : 0.f);
                er[m][b] = hi ? (b < 3 ? o0[b < 3 ? b + 1 : 3] : 0.f) : o0[b];
            }
        };
        edges(0); edges(1);
#pragma unroll
        for (int r = 0; r < K::RI; ++r) {
            edges(r + 2);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int b = i >> 2, e = i & 3;
                float a = bias2;
#pragma unroll
                for (int rr = 0; rr < 3; ++rr) {
                    const int m = r + rr;
                    const float left = e == 0 ? el[m][b] : acc[m][i > 0 ? i - 1 : 0];
                    const float right = e == 3 ? er[m][b] : acc[m][i < 15 ? i + 1 : 15];
                    a = fmaf(w[rr * 3], left, a); a = fmaf(w[rr * 3 + 1], acc[m][i], a); a = fmaf(w[rr * 3 + 2], right, a);
                }
                u[r][i] = decltype(second)::value ? u[r][i] * a : a;
            }
        }
    };

    f32x16_t acc[K::MT];
    float g[K::RI][16];
    strip_mma(smem + K::ALN_OFF, bw, lane, acc);
    chain_load_b<C>(p.W1, tile + K::NT, lane, bw);                     // the second half's weights fly during the first half's depthwise pass
    dw_half(acc, c_b1, c_db, g, std::false_type());
    c_b1 = p.b1[col + C]; c_db = p.dw_b[col + C];
#pragma unroll
    for (int t = 0; t < 9; ++t) w[t] = p.dw_wT[(size_t)t * 2 * C + col + C];
    strip_mma(smem + K::ALN_OFF, bw, lane, acc);
    dw_half(acc, c_b1, c_db, g, std::true_type());
    __syncthreads();                                                   // every wave has read the LayerNorm tile: the gate tile overwrites it

    // ---- SimpleGate -> gate tile (bf16, LDS) and the strip's channel sums ----
    float rsum = 0.f;
#pragma unroll
    for (int r = 0; r < K::RI; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            rsum += g[r][i];
            const int x = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
            *reinterpret_cast<unsigned short*>(smem + K::ALN_OFF + (r * K::S + x) * K::AROW + col * 2) = f32_to_bf16_bits(g[r][i]);
        }
    {
        const float other = strip_other_half(rsum);
        if (!hi) p.pool_part[((size_t)face * K::NSTRIP + kk) * C + col] = rsum + other;
    }
    __syncthreads();
    {
        unsigned short* Gp = p.G + (size_t)(row_first + K::S) * C;
        int tide = threadIdx.x;
        asm volatile("" : "+v"(tide));                                 // addresses computed here, not carried (spilled) through the kernel
#pragma unroll
        for (int it = 0; it < K::OWN * (C / 8) / K::THREADS; ++it) {
            const int u = tide + it * K::THREADS;
            const int r = u / (C / 8), q = u - r * (C / 8);
            *reinterpret_cast<uint4*>(Gp + (size_t)r * C + q * 8) = *reinterpret_cast<const uint4*>(smem + K::ALN_OFF + r * K::AROW + q * 16);
        }
    }
}

// host side: shapes this kernel is written for
inline bool strip_shape_ok(const StripP& p) {
    return p.side == StripCfg::S && p.C == StripCfg::C && p.faces > 0 && p.stats_np >= 1 && p.stats_np <= 16 && p.stats_np * p.stats_cnt == StripCfg::C;
}

inline hipError_t launch_strip_dwgate(const StripP& p, hipStream_t s) {
    typedef StripCfg K;
    if (!strip_shape_ok(p)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(naf_strip_dwgate_kernel, dim3(p.faces * K::NSTRIP), dim3(K::THREADS), K::SMEM, s, p);
    return hipGetLastError();
}

}  // namespace hd
