// hd_chain.hpp — the row-local tail of a (Conditional)NAFBlock as ONE kernel, for the levels whose channel
// count fits a workgroup (C = 128 or 256: levels 0/1):
//
//   s = sca(pooled)                               per-face GEMV on the MFMA (conditional_naf.py:54-65,119)
//   y = x + beta * conv3(g * s)                   (conditional_naf.py:119-123)
//   a = LN(y) * (1 + scale_ffn) + shift_ffn       (utils.py:16-24, conditional_naf.py:126-127)
//   x' = y + gamma * conv5(gate(conv4(a)))        (conditional_naf.py:128-134)
//
// A workgroup owns 32 pixel rows (inside one face) and ALL channels, so nothing here needs another
// workgroup: y, the LayerNorm statistics, the normalised tile and the gated tile live in LDS; only x' (fp32),
// its bf16 copy, the per-tile LayerNorm partials for the next block and (optionally) the HCA-gated copy go to
// memory.  Replaces four launches (sca, conv3, conv4, conv5).  Rounding points are those of the unfused path.
#pragma once
#include "hd_gemm.hpp"

namespace hd {

struct ChainP {
    int M, hw, face0;                 // rows, rows per face, first face of this launch in the batch
    const unsigned short* G;          // [M][C] bf16 gate output of the fused conv1 kernel
    const float* pooled;              // [faces][C]
    const float* pool_part; int pool_nparts; float pool_scale; float* pooled_out;   // or (pool_part != NULL): [faces][nparts][C] sums to add up, x scale; the mean is also stored
    const float* X;                   // [M][C] block input (residual)
    const uint4 *Wsca, *W3, *W4, *W5; // packed bf16 weights (K = C)
    const float *bsca, *b3, *b4, *b5, *beta, *gamma;
    const float* film;                // FiLM table; this block's [bias_ffn | gain_ffn] at film_bias_off / film_gain_off
    int film_face_stride, film_step_stride, film_gain_off, film_bias_off;
    const int* step_ptr;
    float ln_eps;
    float* Xout;                      // [M][C] fp32
    unsigned short* Xout16;           // bf16 copy (next LayerNorm GEMM / down conv), or NULL
    float2* stats_out;                // [M][C/32] (mean, M2) partials of x', or NULL
    unsigned short* outg16;           // HCA-gated copy (x' + add) * (1 + w_c + w_s), or NULL
    const float *gate_c, *gate_s, *add_src;
#ifdef HD_STAMPS
    unsigned long long* stamps;       // diagnostic build (tools/gemm_bench): [workgroup][8] s_memrealtime ticks
#endif
};
#ifdef HD_STAMPS
#define HD_CSTAMP(i) do { if (p.stamps && threadIdx.x == 0) p.stamps[(size_t)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define HD_CSTAMP(i) do { } while (0)
#endif

// MT: 32-row MFMA tiles per wave (BM = 32*MT rows per workgroup).  A workgroup re-reads all 5*C*C weights of the
// block; MT = 2 halves that traffic at level 0 but leaves a single 4-wave workgroup per CU, whose barrier-separated
// phases then have nothing to overlap with (measured slower: 21.4 vs 16.8 us), so MT = 1 is what runs.
template <int C, int MT_ = 1>
struct ChainCfg {
    static constexpr int NT = C / 32;                    // 32-column tiles of a C-wide output
    static constexpr int MT = MT_;
    static constexpr int WAVES = NT, THREADS = 64 * WAVES, BM = 32 * MT;   // one column tile per wave (4 waves at C=128, 8 at C=256)
    static constexpr int TPW = 1;
    static constexpr int KS = C / 16;                    // k-steps of a K = C GEMM
    static constexpr int AROW = C * 2 + 16;              // bytes per bf16 A-tile row (padded)
    static constexpr int YROW = C + 4;                   // floats per y row (padded)
    static constexpr int A1_OFF = 0;                     // A tile of conv3 / later of conv5 (gated conv4 output)
    static constexpr int A2_OFF = A1_OFF + BM * AROW;    // A tile of conv4 (normalised y)
    static constexpr int Y_OFF = A2_OFF + BM * AROW;     // y tile fp32
    static constexpr int S_OFF = Y_OFF + BM * YROW * 4;  // sca vector [C]
    static constexpr int GB_OFF = S_OFF + C * 4;         // FiLM gain | bias [2][C]
    static constexpr int SMEM = GB_OFF + 2 * C * 4;
    static_assert(TPW >= 1, "C must be a multiple of 128");
};

// All B fragments of one column tile (K = C) are requested at once: a GEMM of the chain costs one memory
// round trip, and the next GEMM's weights are requested before the current epilogue / barrier.
template <int C>
__device__ __forceinline__ void chain_load_b(const uint4* W, int tile, int lane, uint4* b) {
    const uint4* Wl = W + (size_t)tile * ChainCfg<C>::KS * 64 + lane;
#pragma unroll
    for (int ks = 0; ks < ChainCfg<C>::KS; ++ks) b[ks] = Wl[ks * 64];
}
template <int C, int MT>
__device__ __forceinline__ void chain_mma(const char* sA, const uint4* b, int lane, f32x16_t (&acc)[MT]) {
    const char* ap = sA + (lane & 31) * ChainCfg<C>::AROW + (lane >> 5) * 16;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[mt][i] = 0.f;
#pragma unroll
    for (int ks = 0; ks < ChainCfg<C>::KS; ++ks)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_t*>(ap + mt * 32 * ChainCfg<C>::AROW + ks * 32),
                                                              __builtin_bit_cast(bf16x8_t, b[ks]), acc[mt], 0, 0, 0);
}

// C = 128, 32-row tiles: asked for three waves per SIMD the compiler needs 155 registers and no scratch (left alone it takes
// 208 and two workgroups per CU); the kernel is a chain of dependent round trips, a third workgroup per CU fills them
#ifndef HD_CHAIN_OCC
#define HD_CHAIN_OCC 3
#endif
template <int C, int MT>
__global__ __launch_bounds__((ChainCfg<C, MT>::THREADS), (C == 128 && MT == 1 ? HD_CHAIN_OCC : 1)) void naf_chain_kernel(const ChainP p) {
    typedef ChainCfg<C, MT> K;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row0 = blockIdx.x * K::BM;
    const int face = row0 / p.hw;                                  // BM rows never straddle faces (hw % BM == 0: host)
    float* s_vec = reinterpret_cast<float*>(smem + K::S_OFF);
    float* gb = reinterpret_cast<float*>(smem + K::GB_OFF);
    float* yt = reinterpret_cast<float*>(smem + K::Y_OFF);
    const int tile = wave;                                         // this wave's 32 output columns
    const int col = tile * 32 + (lane & 31);
    const bool full = row0 + K::BM <= p.M;

    HD_CSTAMP(0);
    uint4 bw[K::KS], bw2[K::KS];
    chain_load_b<C>(p.Wsca, tile, lane, bw);                       // SCA weights first,
    chain_load_b<C>(p.W3, tile, lane, bw2);                        // conv3's right behind them (second register set)
    // ---- FiLM gain/bias of norm2 for this face/step into LDS ----
    {
        const int step = p.step_ptr ? *p.step_ptr : 0;
        const float* f = p.film + (size_t)step * p.film_step_stride + (size_t)(p.face0 + face) * p.film_face_stride;
        for (int k = tid; k < C; k += K::THREADS) { gb[k] = f[p.film_gain_off + k]; gb[C + k] = f[p.film_bias_off + k]; }
    }
    // per-column constants of all four GEMM epilogues: requested now.  Read at the top of each epilogue they were a dependent
    // round trip per phase (the barriers keep the compiler from hoisting them)
    const float c_bsca = p.bsca[col], c_b3 = p.b3[col], c_beta = p.beta[col], c_b4a = p.b4[col], c_b4b = p.b4[col + C], c_b5 = p.b5[col],
                c_gamma = p.gamma[col];
    // the gate tile G of this workgroup's rows (scaled by s and staged after the SCA phase): requested now, with the weights,
    // not after the phase's barrier (one more dependent round trip there); rows beyond M re-read row0 and are zeroed later
    constexpr int kGIt = K::BM * (C / 8) / K::THREADS;
    static_assert(kGIt * K::THREADS == K::BM * (C / 8), "G tile / threads");
    uint4 gx[kGIt];
#pragma unroll
    for (int it = 0; it < kGIt; ++it) {
        const int u = tid + it * K::THREADS;
        const int rl = u / (C / 8), kq = u - rl * (C / 8);
        const int row = (full || row0 + rl < p.M) ? row0 + rl : row0;
        gx[it] = *reinterpret_cast<const uint4*>(p.G + (size_t)row * C + kq * 8);
    }
    // ---- SCA: s = Wsca * pooled[face] + b on the MFMA (row 0 of the A tile holds the pooled vector) ----
    {
        // the pooled vector goes through LDS (the y tile is free until conv3's epilogue).  Read from global memory by the two
        // lanes that hold row 0, under a branch, it was 16 load -> wait -> pack round trips in a row: most of this phase.
        float* pl = yt;
        const float* pv = p.pooled + (size_t)face * C;
        if (p.pool_part) {                                             // per-strip channel sums (hd_strip.hpp): fixed order, every workgroup of the face gets the same bits
            const float* pp = p.pool_part + (size_t)face * p.pool_nparts * C;
            for (int k = tid; k < C; k += K::THREADS) {
                float v[8], s = 0.f;                                    // pool_nparts <= 8 (host): all loads in flight, then the sum in index order
#pragma unroll
                for (int b = 0; b < 8; ++b) v[b] = b < p.pool_nparts ? pp[b * C + k] : 0.f;
#pragma unroll
                for (int b = 0; b < 8; ++b) s += v[b];
                pl[k] = s * p.pool_scale;
                if (row0 == face * p.hw) p.pooled_out[(size_t)face * C + k] = pl[k];      // introspection copy (the face's first workgroup)
            }
        } else {
            for (int k = tid; k < C; k += K::THREADS) pl[k] = pv[k];
        }
        __syncthreads();
        uint4 a[K::KS];
#pragma unroll
        for (int ks = 0; ks < K::KS; ++ks) {
            const float* q = pl + ks * 16 + 8 * (lane >> 5);              // LDS, same address for the 32 lanes of a half-wave
            const float4 v0 = *reinterpret_cast<const float4*>(q), v1 = *reinterpret_cast<const float4*>(q + 4);
            const uint4 row0 = make_uint4(pack2(v0.x, v0.y), pack2(v0.z, v0.w), pack2(v1.x, v1.y), pack2(v1.z, v1.w));
            a[ks] = (lane & 31) == 0 ? row0 : make_uint4(0, 0, 0, 0);
        }
        f32x16_t acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int ks = 0; ks < K::KS; ++ks)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a[ks]), __builtin_bit_cast(bf16x8_t, bw[ks]), acc, 0, 0, 0);
        if (lane < 32) s_vec[col] = acc[0] + c_bsca;               // C/D row 0 = reg 0 of lanes 0..31
    }
    chain_load_b<C>(p.W4, tile, lane, bw);                         // conv4's first gate half flies during the staging and conv3
    // residual x for this wave's tile (needed by the conv3 epilogue): request it now as well
    float xr[MT][16];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int rl = mt * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
            xr[mt][i] = (full || row0 + rl < p.M) ? p.X[(size_t)(row0 + rl) * C + col] : 0.f;
        }
    __syncthreads();

    HD_CSTAMP(1);                                                  // SCA done
    // ---- A1 = bf16(G * s): 32 rows x C, whole 128-byte lines ----
    {
        char* sA = smem + K::A1_OFF;
#pragma unroll
        for (int it = 0; it < kGIt; ++it) {                            // G was requested before the SCA phase (gx)
            const int u = tid + it * K::THREADS;
            const int rl = u / (C / 8), kq = u - rl * (C / 8);
            float v[8]; unpack8(gx[it], v);
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] *= s_vec[kq * 8 + i];
            *reinterpret_cast<uint4*>(sA + rl * K::AROW + kq * 16) = (full || row0 + rl < p.M) ? pack8(v) : make_uint4(0, 0, 0, 0);
        }
    }
    __syncthreads();

    HD_CSTAMP(2);                                                  // A1 staged
    // ---- conv3 -> y = x + beta * (acc + b3) into LDS ----
    {
        f32x16_t acc[MT];
        chain_mma<C, MT>(smem + K::A1_OFF, bw2, lane, acc);
        chain_load_b<C>(p.W4, tile + K::NT, lane, bw2);            // second gate half, now that conv3 has consumed its registers
        const float bb = c_b3, be = c_beta;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int rl = mt * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
                yt[rl * K::YROW + col] = xr[mt][i] + (acc[mt][i] + bb) * be;
            }
    }
    __syncthreads();

    HD_CSTAMP(3);                                                  // conv3 done
    // ---- LayerNorm + FiLM on y (statistics fp32 two-pass; the normalised value is the bf16 copy of y, as in
    //      the unfused path) -> A2 ----
    {
        char* sA = smem + K::A2_OFF;
        // 16 lanes per row (one DPP row: row16_sum leaves the sum in all 16 lanes), four rows per wave pass
        constexpr int PER = C / 16;                                  // values per lane per row
        constexpr int RPW = K::BM / K::WAVES;                        // rows per wave
        static_assert(RPW % 4 == 0, "rows per wave");
        const int l16 = lane & 15;
        for (int rl = wave * RPW + (lane >> 4); rl < wave * RPW + RPW; rl += 4) {
            float v[PER];
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < PER; ++i) { v[i] = yt[rl * K::YROW + l16 + 16 * i]; sum += v[i]; }
            const float mean = row16_sum(sum) * (1.0f / C);
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < PER; ++i) { const float d = v[i] - mean; q += d * d; }
            const float rstd = 1.0f / sqrtf(row16_sum(q) * (1.0f / C) + p.ln_eps);
            const float nmr = -mean * rstd;
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int k = l16 + 16 * i;
                const float yq = bf16_bits_to_f32(f32_to_bf16_bits(v[i]));
                const float a = fmaf(fmaf(yq, rstd, nmr), gb[k], gb[C + k]);
                *reinterpret_cast<unsigned short*>(sA + rl * K::AROW + k * 2) = f32_to_bf16_bits(a);
            }
        }
    }
    __syncthreads();

    HD_CSTAMP(4);                                                  // LN done
    // ---- conv4 (tile j and tile j + C/32) -> SimpleGate -> A3 (reuses the A1 region) ----
    {
        f32x16_t acc1[MT], acc2[MT];
        chain_mma<C, MT>(smem + K::A2_OFF, bw, lane, acc1);
        chain_mma<C, MT>(smem + K::A2_OFF, bw2, lane, acc2);
        chain_load_b<C>(p.W5, tile, lane, bw);                     // conv5 weights
        char* sA = smem + K::A1_OFF;
        const float b1 = c_b4a, b2 = c_b4b;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int rl = mt * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
                *reinterpret_cast<unsigned short*>(sA + rl * K::AROW + col * 2) = f32_to_bf16_bits((acc1[mt][i] + b1) * (acc2[mt][i] + b2));
            }
    }
    __syncthreads();

    HD_CSTAMP(5);                                                  // conv4 + gate done
    // ---- conv5 -> x' = y + gamma * (acc + b5): fp32, bf16 copy, LayerNorm partials, optional HCA-gated copy ----
    // The tile goes back through LDS (in place over y) and leaves in whole 16-byte units: as 4-byte / 2-byte stores per
    // accumulator element the epilogue was 32-48 store instructions per lane (store-issue bound: 3.2-3.6 us of the kernel).
    {
        f32x16_t accs[MT];
        chain_mma<C, MT>(smem + K::A1_OFF, bw, lane, accs);
        const float bb = c_b5, ga = c_gamma;
        float2* st_lds = reinterpret_cast<float2*>(smem + K::A2_OFF);           // [BM][NT] partials (the normalised tile is dead)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const f32x16_t& acc = accs[mt];
            float v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int rl = mt * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
                v[i] = yt[rl * K::YROW + col] + (acc[i] + bb) * ga;
                if (!(full || row0 + rl < p.M)) v[i] = 0.f;
                yt[rl * K::YROW + col] = v[i];                             // each lane rewrites only what it read
            }
            if (p.stats_out) {
                float2 ms[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) ms[i] = halfwave_mean_m2(v[i]);
                if ((lane & 31) == kStatLane) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) st_lds[(mt * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5)) * K::NT + tile] = ms[i];
                }
            }
        }
        __syncthreads();
        constexpr int UPR = C / 8;                                          // 16-byte bf16 units per row
        constexpr int kIt = K::BM * UPR / K::THREADS;
        static_assert(kIt * K::THREADS == K::BM * UPR, "tile / threads");
#pragma unroll
        for (int it = 0; it < kIt; ++it) {
            const int u = tid + it * K::THREADS;
            const int rl = u / UPR, kq = u - rl * UPR;
            const int row = row0 + rl;
            if (full || row < p.M) {
                const float* xv = yt + rl * K::YROW + kq * 8;
                const float4 a = *reinterpret_cast<const float4*>(xv), b = *reinterpret_cast<const float4*>(xv + 4);
                float* xo = p.Xout + (size_t)row * C + kq * 8;
                *reinterpret_cast<float4*>(xo) = a;
                *reinterpret_cast<float4*>(xo + 4) = b;
                const float x8[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
                if (p.Xout16) *reinterpret_cast<uint4*>(p.Xout16 + (size_t)row * C + kq * 8) = pack8(x8);
                if (p.outg16) {                                             // f_d * (1 + w_c + w_s) (+ add): the HCA conv input (hca.py:28)
                    float gv[8];
                    const float gsr = p.gate_s[row];
                    const float* gc = p.gate_c + (size_t)face * C + kq * 8;     // gates are per launch (chain-local faces)
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float ad = p.add_src ? p.add_src[(size_t)row * C + kq * 8 + i] : 0.f;
                        gv[i] = (x8[i] + ad) * (1.0f + gc[i] + gsr);
                    }
                    *reinterpret_cast<uint4*>(p.outg16 + (size_t)row * C + kq * 8) = pack8(gv);
                }
            }
        }
        if (p.stats_out) {                                                  // [row][NT] partials: 16 bytes = two partials per store
            constexpr int SPR = K::NT / 2;
            for (int u = tid; u < K::BM * SPR; u += K::THREADS) {
                const int rl = u / SPR, q2 = u - rl * SPR;
                if (full || row0 + rl < p.M)
                    *reinterpret_cast<float4*>(p.stats_out + (size_t)(row0 + rl) * K::NT + 2 * q2) = *reinterpret_cast<const float4*>(st_lds + rl * K::NT + 2 * q2);
            }
        }
    }
    HD_CSTAMP(6);
}

template <int C, int MT>
inline hipError_t launch_chain(const ChainP& p, hipStream_t s) {
    typedef ChainCfg<C, MT> K;
    if (K::SMEM > 65536) {
        static std::atomic<unsigned long long> granted{0};
        { const hipError_t e = grant_dynamic_lds(reinterpret_cast<const void*>(&naf_chain_kernel<C, MT>), K::SMEM, granted); if (e != hipSuccess) return e; }
    }
    hipLaunchKernelGGL((naf_chain_kernel<C, MT>), dim3((p.M + K::BM - 1) / K::BM), dim3(K::THREADS), K::SMEM, s, p);
    return hipGetLastError();
}

}  // namespace hd
