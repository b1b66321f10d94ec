// hd_strip.hpp — the attention half's front of a (Conditional)NAFBlock on faces too large for one workgroup, as ONE launch:
// LayerNorm2d + FiLM -> conv1 (1x1, C -> 2C) -> depthwise 3x3 (pad 1) -> SimpleGate -> G, plus the per-strip channel sums of the
// SCA pool (models/denoiser/conditional_naf.py:114-119, models/naf.py:108-113).  gfx950 only.  Two instantiations, both latent 32:
// level 0 (C = 128, 32 x 32 faces) and level 1 (C = 256, 16 x 16 faces); the FPG's levels of the same shapes use them too.
//
// The fused GEMM epilogue that does this at the other levels needs whole faces inside one workgroup's rows.  A 32 x 32 face has
// 1024 pixels: level 0 ran as three launches with T1 (fp32 [rows][2C], 67 MB at batch 64) written and read back in between, 55 us
// per block.  A 16 x 16 face fits a 256-row workgroup, but then 64 faces are 64 workgroups per column tile group: 23.5 us per block
// on a quarter of the chip.  Here a workgroup owns a STRIP of 4 image rows of one face and 128 gate columns (four waves, one per
// column tile j, which also takes tile j + C/32: the two SimpleGate halves; at C = 256 two workgroups share a strip -- one
// 8-wave workgroup per CU had nothing to overlap its phases with: 16.9 us), and recomputes conv1 for the image row above and
// below its strip (6 image rows for 4: MFMA time is not what bounds this kernel).  Rows are pixels in image order, so an MFMA
// 32 x 32 accumulator tile is 32 / S whole image rows of one channel per lane pair: lane (j, h) holds, per image row, x = 8b + 4h +
// (0..3), b = 0 .. S/8 - 1, of column j.  The vertical taps are therefore the same register position of the neighbouring image
// row (same or neighbouring accumulator tile), the horizontal ones the neighbouring register or, at the 4-pixel group edges, a
// register of lane ^ 32 — the depthwise conv runs out of the accumulators, T1 never exists, not even in LDS.
//
// LayerNorm input as in the GEMM loader it replaces (hd_gemm.hpp: LdF32LN_T): statistics exact fp32 from the producer's
// (mean, M2) partials (equal counts: two short sums), the value normalised is the bf16 copy Xb; same rounding points.
// Traffic per block at level 0: Xb 1.5 x 16.8 MB in (halo rows), G 16.8 MB out, against 117 MB for the three launches.
#pragma once
#include <type_traits>
#include "hd_chain.hpp"
#include "hd_stage_api.hpp"

namespace hd {

template <int C_, int S_>
struct StripCfg {
    static constexpr int C = C_, S = S_, HW = S * S;
    static constexpr int RI = 4, OWN = RI * S, ROWS = OWN + 2 * S, MT = ROWS / 32;   // image rows per strip; own rows; with halo; MFMA row tiles
    static constexpr int NSTRIP = S / RI;
    static constexpr int NT = C / 32, KS = C / 16;                // 32-column tiles of a C-wide output
    static constexpr int WAVES = 4, THREADS = 64 * WAVES, CG = NT / WAVES;   // a workgroup takes 4 column tiles (+ their gate partners); CG workgroups share a strip
    static constexpr int RT = 32 / S, NB = S / 8, RPI = S / 2;     // image rows per MFMA tile; 4-pixel groups and registers per image row per lane
    static constexpr int AROW = ChainCfg<C>::AROW;               // bytes per bf16 tile row (padded): chain_mma's layout
    static constexpr int ALN_OFF = 0;                            // LayerNorm output bf16 [ROWS][AROW]: image row above | own rows | image row below;
                                                                 // later the gate tile bf16 [OWN][AROW] on its way out
    static constexpr int SMEM = ALN_OFF + ROWS * AROW;
    static constexpr int V = C / 128;                            // 8-channel groups per lane of a 16-lane row group
    static_assert((S == 32 || S == 16) && ROWS % 32 == 0 && (C == 128 || C == 256) && NSTRIP <= 8, "shapes this kernel is written for");
    static_assert((OWN * (32 * WAVES / 8)) % THREADS == 0 && ROWS % (THREADS / 16) == 0 && NT % WAVES == 0, "copy-out and LayerNorm passes");
};

// value of lane ^ 32 (the other 4-pixel group of the same column)
__device__ __forceinline__ float strip_other_half(float v) { return __shfl_xor(v, 32); }

// conv1 of all row tiles against one column tile: the A fragments of one k-step are read while the previous step's MFMAs
// run.  (chain_mma leaves the order to the compiler, which reads all fragments -- 192 registers at level 0 -- ahead of the first MFMA.)
template <class K>
__device__ __forceinline__ void strip_mma(const char* sA, const uint4* b, int lane, f32x16_t (&acc)[K::MT]) {
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

template <int C, int S>
__global__ __launch_bounds__((StripCfg<C, S>::THREADS), 2) void naf_strip_dwgate_kernel(const StripP p) {
    typedef StripCfg<C, S> K;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // workgroups are dealt round-robin to the 8 XCDs: give each XCD a contiguous run of (face, strip) pairs, so that the halo rows
    // a strip re-reads are the own rows of a neighbour on the same L2
    const int lid = (gridDim.x & 7) == 0 ? (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    const int cg = lid % K::CG, sid = lid / K::CG;                 // column group (C = 256: two workgroups per strip, neighbours on one XCD: they read the same rows)
    const int face = sid / K::NSTRIP, kk = sid - face * K::NSTRIP;
    const bool has_up = kk > 0, has_dn = kk < K::NSTRIP - 1;
    const int row_first = face * K::HW + kk * K::OWN - K::S;      // global row of local row 0 (the image row above the strip)
    const int tile = cg * K::WAVES + wave, col = tile * 32 + (lane & 31);

    uint4 bw[K::KS];
    chain_load_b<C>(p.W1, tile, lane, bw);
    // ---- LayerNorm2d + FiLM of own + halo rows: 16 lanes per row, 8 V channels per lane (k = 8 l16 + 128 v + 0..7) ----
    {
        const int l16 = lane & 15, g16 = tid >> 4;
        constexpr int RG = K::THREADS / 16, PASSES = K::ROWS / RG;     // row groups per pass
        const float* f = p.film + (size_t)(p.face0 + face) * p.film_face_stride;
        float gn[K::V][8], bs[K::V][8];
#pragma unroll
        for (int v = 0; v < K::V; ++v) {
            const int k = 8 * l16 + 128 * v;
            const float4 g0 = *reinterpret_cast<const float4*>(f + p.film_gain_off + k), g1 = *reinterpret_cast<const float4*>(f + p.film_gain_off + k + 4);
            const float4 b0 = *reinterpret_cast<const float4*>(f + p.film_bias_off + k), b1 = *reinterpret_cast<const float4*>(f + p.film_bias_off + k + 4);
            gn[v][0] = g0.x; gn[v][1] = g0.y; gn[v][2] = g0.z; gn[v][3] = g0.w; gn[v][4] = g1.x; gn[v][5] = g1.y; gn[v][6] = g1.z; gn[v][7] = g1.w;
            bs[v][0] = b0.x; bs[v][1] = b0.y; bs[v][2] = b0.z; bs[v][3] = b0.w; bs[v][4] = b1.x; bs[v][5] = b1.y; bs[v][6] = b1.z; bs[v][7] = b1.w;
        }
        uint4 raw[PASSES][K::V];
        float2 part[PASSES];
        const int np = p.stats_np;
#pragma unroll
        for (int it = 0; it < PASSES; ++it) {                          // every load first: one round trip for the whole tile
            const int r = it * RG + g16;
            const bool ok = (r >= K::S || has_up) && (r < K::S + K::OWN || has_dn);
            const int row = ok ? row_first + r : row_first + K::S;
#pragma unroll
            for (int v = 0; v < K::V; ++v) raw[it][v] = *reinterpret_cast<const uint4*>(p.Xb + (size_t)row * C + 8 * l16 + 128 * v);
            part[it] = l16 < np ? p.stats_in[(size_t)row * np + l16] : make_float2(0.f, 0.f);
        }
        const float inv_np = 1.0f / (float)np, cnt = (float)p.stats_cnt;
#pragma unroll
        for (int it = 0; it < PASSES; ++it) {
            const int r = it * RG + g16;
            const bool ok = (r >= K::S || has_up) && (r < K::S + K::OWN || has_dn);
            // equal-count partials: mean = sum(mean_i) / np, M2 = sum(M2_i + cnt (mean_i - mean)^2)   (hd_gemm.hpp: block_finish)
            const float mean = row16_sum(part[it].x) * inv_np;
            const float d = part[it].x - mean;
            const float q = l16 < np ? fmaf(cnt * d, d, part[it].y) : 0.f;
            const float var = row16_sum(q) * (inv_np / cnt);
            float rstd = __frsqrt_rn(var + p.ln_eps), mu = -mean * rstd;
            asm volatile("" : "+v"(rstd), "+v"(mu));                  // two registers of their own (hd_gemm.hpp: unit_stats)
#pragma unroll
            for (int v = 0; v < K::V; ++v) {
                float x[8];
                unpack8(raw[it][v], x);
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = fmaf(fmaf(x[e], rstd, mu), gn[v][e], bs[v][e]);
                *reinterpret_cast<uint4*>(smem + K::ALN_OFF + r * K::AROW + 16 * l16 + 256 * v) = ok ? pack8(x) : make_uint4(0, 0, 0, 0);
            }
        }
    }
    // per-channel constants of the first gate half
    float c_b1 = p.b1[col], c_db = p.dw_b[col], w[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) w[t] = p.dw_wT[(size_t)t * 2 * C + col];
    __syncthreads();

    // depthwise 3x3 of the own image rows out of the accumulator tiles of one gate half.  Local image row y = 0 is the one above the
    // strip, 1 .. RI the own ones, RI + 1 the one below; its registers are acc[y / RT][(y % RT) * RPI + 4 b + e].
    const bool hi = lane >= 32;
    // second == false: u <- depthwise output; second == true: u <- u * depthwise output (SimpleGate)
    auto dw_half = [&](f32x16_t (&acc)[K::MT], float bias1, float bias2, float (&u)[K::RI][K::RPI], auto second) __attribute__((always_inline)) {
        // T1 = conv1 + bias inside the face, 0 outside (the conv's zero padding)
#pragma unroll
        for (int y = 0; y < K::RI + 2; ++y) {
            const bool ok = (y > 0 || has_up) && (y < K::RI + 1 || has_dn);
#pragma unroll
            for (int k = 0; k < K::RPI; ++k) {
                const int m = y / K::RT, i = (y % K::RT) * K::RPI + k;
                acc[m][i] = ok ? acc[m][i] + bias1 : 0.f;
            }
        }
        // the pixels left of register 4b and right of register 4b+3 live in lane ^ 32:
        //   h = 0 (x = 8b..8b+3):   left = other[4(b-1)+3] (0 at b = 0), right = other[4b]
        //   h = 1 (x = 8b+4..8b+7): left = other[4b+3],                   right = other[4(b+1)] (0 at the last b)
        float el[K::RI + 2][K::NB], er[K::RI + 2][K::NB];
        auto edges = [&](int y) __attribute__((always_inline)) {
            float o0[K::NB], o3[K::NB];
#pragma unroll
            for (int b = 0; b < K::NB; ++b) {
                o0[b] = strip_other_half(acc[y / K::RT][(y % K::RT) * K::RPI + 4 * b]);
                o3[b] = strip_other_half(acc[y / K::RT][(y % K::RT) * K::RPI + 4 * b + 3]);
            }
#pragma unroll
            for (int b = 0; b < K::NB; ++b) {
                el[y][b] = hi ? o3[b] : (b > 0 ? o3[b > 0 ? b - 1 : 0] : 0.f);
                er[y][b] = hi ? (b < K::NB - 1 ? o0[b < K::NB - 1 ? b + 1 : 0] : 0.f) : o0[b];
            }
        };
        edges(0); edges(1);
#pragma unroll
        for (int r = 0; r < K::RI; ++r) {
            edges(r + 2);
#pragma unroll
            for (int k = 0; k < K::RPI; ++k) {
                const int b = k >> 2, e = k & 3;
                float a = bias2;
#pragma unroll
                for (int rr = 0; rr < 3; ++rr) {
                    const int y = r + rr;
                    const f32x16_t& t = acc[y / K::RT];
                    const int base = (y % K::RT) * K::RPI;
                    const float left = e == 0 ? el[y][b] : t[base + (k > 0 ? k - 1 : 0)];
                    const float right = e == 3 ? er[y][b] : t[base + (k < K::RPI - 1 ? k + 1 : 0)];
                    a = fmaf(w[rr * 3], left, a); a = fmaf(w[rr * 3 + 1], t[base + k], a); a = fmaf(w[rr * 3 + 2], right, a);
                }
                u[r][k] = decltype(second)::value ? u[r][k] * a : a;
            }
        }
    };

    f32x16_t acc[K::MT];
    float g[K::RI][K::RPI];
    strip_mma<K>(smem + K::ALN_OFF, bw, lane, acc);
    chain_load_b<C>(p.W1, tile + K::NT, lane, bw);                     // the second half's weights fly during the first half's depthwise pass
    // (C = 256: the compiler starts the second half's MFMAs under the first half's depthwise pass, on a second set of accumulators:
    // 12 registers spilled around that overlap, with or without this prefetch; the overlap is worth more)
    dw_half(acc, c_b1, c_db, g, std::false_type());
    c_b1 = p.b1[col + C]; c_db = p.dw_b[col + C];
#pragma unroll
    for (int t = 0; t < 9; ++t) w[t] = p.dw_wT[(size_t)t * 2 * C + col + C];
    strip_mma<K>(smem + K::ALN_OFF, bw, lane, acc);
    dw_half(acc, c_b1, c_db, g, std::true_type());
    __syncthreads();                                                   // every wave has read the LayerNorm tile: the gate tile overwrites it

    // ---- SimpleGate -> gate tile (bf16, LDS) and the strip's channel sums ----
    float rsum = 0.f;
#pragma unroll
    for (int r = 0; r < K::RI; ++r)
#pragma unroll
        for (int k = 0; k < K::RPI; ++k) {
            rsum += g[r][k];
            const int x = (k & 3) + 8 * (k >> 2) + 4 * (lane >> 5);
            *reinterpret_cast<unsigned short*>(smem + K::ALN_OFF + (r * K::S + x) * K::AROW + col * 2) = f32_to_bf16_bits(g[r][k]);
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
        constexpr int CW = 32 * K::WAVES;                              // this workgroup's columns of the gate tile
#pragma unroll
        for (int it = 0; it < K::OWN * (CW / 8) / K::THREADS; ++it) {
            const int u = tide + it * K::THREADS;
            const int r = u / (CW / 8), q = u - r * (CW / 8);
            *reinterpret_cast<uint4*>(Gp + (size_t)r * C + cg * CW + q * 8) = *reinterpret_cast<const uint4*>(smem + K::ALN_OFF + r * K::AROW + (cg * CW + q * 8) * 2);
        }
    }
}

// host side: shapes this kernel is written for
inline bool strip_shape_ok(int C, int side) { return (C == 128 && side == 32) || (C == 256 && side == 16); }

template <int C, int S>
inline hipError_t launch_strip_inst(const StripP& p, hipStream_t s) {
    typedef StripCfg<C, S> K;
    hipLaunchKernelGGL((naf_strip_dwgate_kernel<C, S>), dim3(p.faces * K::NSTRIP * K::CG), dim3(K::THREADS), K::SMEM, s, p);
    return hipGetLastError();
}

inline hipError_t launch_strip_dwgate(const StripP& p, hipStream_t s) {
    if (!strip_shape_ok(p.C, p.side) || p.faces <= 0 || p.stats_np < 1 || p.stats_np > 16 || p.stats_np * p.stats_cnt != p.C) return hipErrorInvalidValue;
    return p.C == 128 ? launch_strip_inst<128, 32>(p, s) : launch_strip_inst<256, 16>(p, s);
}

}  // namespace hd
