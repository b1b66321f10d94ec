// hd_conv.hpp — 3x3 (pad 1) convolution C -> C on small faces as an implicit GEMM whose A operand never leaves
// LDS: the HCA conv `fused_mlp` of models/fpg/hca.py:21-23,29 (BatchNorm folded, ReLU).
//
// The gather form of this conv (hd_gemm.hpp, LdConv) re-reads every input pixel nine times through L1 and, with
// waves stacked along M, every weight fragment once per wave: at level 0 that is 1.15 MB of loads per workgroup
// for 138 KB of distinct bytes, and the kernel runs at the per-CU ingest limit (25 us).  Here a workgroup stages
// the whole faces that contain its BM rows once (bf16, NP pixels + one zero row), each wave builds its MFMA A
// fragments from LDS with the tap shift applied to the row address, and the only global stream in the K loop is
// the packed weights (k = tap*C + c order, as packed for LdConv).
//
//   grid (M / BM, C / 32); 8 waves = RG row groups (MT 32-row tiles each) x KS slices of the input channels (every
//   wave walks all nine taps over its C/KS channels, so the tap -- and with it the LDS row shift -- is a
//   compile-time constant of the unrolled loop); KS > 1: partial tiles are summed through LDS in slice order.
#pragma once
#include "hd_gemm.hpp"
#ifndef HD_CONV_DEPTH
#define HD_CONV_DEPTH 8
#endif
#ifndef HD_CONV_NT_MINC
#define HD_CONV_NT_MINC 1024
#endif

namespace hd {

struct ConvP {
    int M;                            // rows (pixels) of the level
    const unsigned short* X;          // [M][C] bf16 input (pre-gated f_d)
    const uint4* W;                   // packed [C/32][9*C/16][64] uint4
    const float* bias;                // [C] (BN folded)
    float* out;                       // [M][C] fp32, ReLU applied
    unsigned short* out16;            // bf16 copy or NULL
};

// STRIP: faces too large for LDS (32x32 pixels at level 0 of latent 32).  A workgroup's BM rows are whole image rows of one
// face; it stages them plus the image row above and the one below (rows outside the face are never read: their taps
// point at the zero row).
// NT: 32-column output tiles per workgroup (each wave holds MT x NT accumulators and streams NT weight tiles).  One tile per workgroup
// means C/32 workgroups stage the same faces; at latent 32 the staged faces fill most of the LDS (one workgroup per CU), so those
// workgroups ran one after the other: 2-4 rounds of stage -> K loop -> epilogue.  NT = 2 / 4 makes every level one round there.
template <int C_, int S_, int BM_, int MT_, int KS_, int DEPTH_ = 8, bool STRIP_ = false, int NT_ = 1>
struct ConvCfg {
    static constexpr int C = C_, S = S_, HW = S_ * S_, BM = BM_, MT = MT_, KS = KS_, NT = NT_;
    static constexpr bool STRIP = STRIP_;
    static constexpr int RG = BM / (32 * MT);                    // row groups
    static constexpr int WAVES = RG * KS, THREADS = 64 * WAVES;
    static constexpr int NP = STRIP ? BM + 2 * S : (BM > HW ? BM : HW);   // staged pixels: whole faces covering the BM rows / the strip and its halo
    static constexpr int ROWB = C * 2 + 16;                      // bytes per staged pixel (padded against bank conflicts)
    static constexpr int KSTEPS = 9 * C / 16;                    // k-steps of the whole K = 9*C
    static constexpr int SPT = C / 16, CPW = SPT / KS;           // k-steps per tap: all channels / this wave's slice
    static constexpr int NSTEP = 9 * CPW;                        // k-steps per wave
    static constexpr int XIN = (NP + 1) * ROWB;                  // + one zero row for taps outside the face
    static constexpr int RED = KS > 1 ? KS * BM * 32 * 4 : 0;    // partial tiles (aliases the staging area)
    static constexpr int SMEM = XIN > RED ? XIN : RED;
    static constexpr int DEPTH = DEPTH_;                         // weight fragments (1 KiB each) in flight per wave and column tile
    static constexpr int OCC = (NT_ > 1 && C_ == 512) ? 4 : 1;   // waves per SIMD asked of the compiler: level 2 of latent 32 keeps two workgroups per CU (135 -> 128 registers)
    static_assert(WAVES == 8 || WAVES == 4, "4 or 8 waves");
    static_assert(NSTEP >= DEPTH, "prefetch ring longer than the loop");
    static_assert(SPT % KS == 0 && BM % (32 * MT) == 0, "shape");
    static_assert((HW >= BM && HW % BM == 0) || (BM % HW == 0), "row tiles are whole faces or a face is whole row tiles");
    static_assert(!STRIP || (BM % S == 0 && HW % BM == 0 && HW > BM), "a strip is whole image rows of one face");
};

template <class K>
__global__ __launch_bounds__((K::THREADS), (K::OCC)) void hca_conv_kernel(const ConvP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int C = K::C, S = K::S, HW = K::HW, MT = K::MT, KS = K::KS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rg = wave / KS, ksl = wave - rg * KS;
    // XCD-aware block -> (row group, weight tile) map (as in gemm_skinny_kernel): the row groups that stream the
    // same weight tile get linear ids that are equal mod 8, i.e. run on one XCD and share the tile through its L2
    int bx = blockIdx.x, tile = blockIdx.y;
    if (K::C >= 1024 && (gridDim.y & 7) == 0 && gridDim.x > 1) {      // only where the weights dwarf the activations (measured)
        const int lin = blockIdx.y * gridDim.x + blockIdx.x, j = lin >> 3;
        bx = j % (int)gridDim.x;
        tile = (j / (int)gridDim.x) * 8 + (lin & 7);
    }
    const int row0 = bx * K::BM;
    const int pix0 = K::STRIP ? row0 - S : (row0 / K::NP) * K::NP;   // first staged pixel (face aligned / one image row above the strip: may lie outside the face)
    // weights first: independent of everything else.  Step n = tap * CPW + j reads k-step tap * SPT + ksl * CPW + j.
    uint4 bq[K::DEPTH][K::NT];
    const uint4* Wl = p.W + ((size_t)tile * K::NT * K::KSTEPS + ksl * K::CPW) * 64 + lane;     // + t * KSTEPS * 64: column tile tile * NT + t
    // level 3 and deeper: 19+ MB of weights that only eight row groups share -> non-temporal stream (see gemm_skinny_kernel)
#define HD_CONV_B(n, t) (K::C >= HD_CONV_NT_MINC ? nt_load_u4(&Wl[(size_t)((t) * K::KSTEPS + ((n) / K::CPW) * K::SPT + ((n) % K::CPW)) * 64]) \
                                      : Wl[(size_t)((t) * K::KSTEPS + ((n) / K::CPW) * K::SPT + ((n) % K::CPW)) * 64])
#pragma unroll
    for (int d = 0; d < K::DEPTH; ++d)
#pragma unroll
        for (int t = 0; t < K::NT; ++t) bq[d][t] = HD_CONV_B(d, t);

    // ---- stage the faces (16-byte pieces, whole lines) and the zero row ----
    {
        constexpr int PPR = C / 8;                                 // pieces per pixel
        const uint4* src = reinterpret_cast<const uint4*>(p.X) + (long long)pix0 * PPR;
        // all loads first, then all LDS stores: written as one loop the compiler kept it rolled, one load -> wait -> store
        // per trip, i.e. eight serial round trips ahead of the first MFMA (6 of the kernel's 12 us).  The address is clamped
        // and the value masked instead of the load being predicated, so that nothing sits under a branch.
        constexpr int NIT = (K::NP * PPR + K::THREADS - 1) / K::THREADS;
        static_assert(NIT <= 16, "staging registers");
        uint4 stage[NIT];
        // strip: pieces of the halo rows that fall outside the strip's face are clamped into it and zeroed (never read anyway)
        const int first = K::STRIP ? ((row0 / HW) * HW - pix0) * PPR : 0;
        const int last = K::STRIP ? ((row0 / HW + 1) * HW - pix0) * PPR - 1
                                  : (p.M - pix0) * PPR - 1;        // last valid piece of this workgroup's faces (>= 0: row0 < M)
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid + it * K::THREADS;
            stage[it] = src[i > last ? last : (i < first ? first : i)];
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid + it * K::THREADS;
            if ((K::NP * PPR) % K::THREADS == 0 || i < K::NP * PPR) {
                const int px = i / PPR, q = i - px * PPR;
                *reinterpret_cast<uint4*>(smem + px * K::ROWB + q * 16) = (i <= last && i >= first) ? stage[it] : make_uint4(0, 0, 0, 0);
            }
        }
        for (int i = tid; i < K::ROWB / 16; i += K::THREADS) *reinterpret_cast<uint4*>(smem + K::NP * K::ROWB + i * 16) = make_uint4(0, 0, 0, 0);
    }
    // ---- per lane: LDS byte offset of the source pixel of (row tile mt, tap), zero row when outside the face ----
    int src_off[MT][9];
    {
        const int r = lane & 31, h = lane >> 5;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            // staged-pixel index of this row; strip: pixel index inside the face, staged index = that - (first pixel of the strip) + S
            const int pl = (K::STRIP ? row0 % HW : row0 - pix0) + (rg * MT + mt) * 32 + r;
            const int f = pl / HW, rem = pl - f * HW, y = rem / S, x = rem - y * S;
            const int sbase = K::STRIP ? S - row0 % HW : f * HW;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
                const bool in = yy >= 0 && yy < S && xx >= 0 && xx < S;
                src_off[mt][t] = (in ? (sbase + yy * S + xx) : K::NP) * K::ROWB + h * 16;
            }
        }
    }
    f32x16_t acc[MT][K::NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < K::NT; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][t][i] = 0.f;
    __syncthreads();

    // ---- K loop, fully unrolled: 9 taps x CPW k-steps of this wave's channel slice ----
    const int cbase = ksl * K::CPW * 32;                           // byte offset of the slice inside a staged pixel
#pragma unroll
    for (int n = 0; n < K::NSTEP; ++n) {
        const int tap = n / K::CPW, j = n % K::CPW;                // compile-time after unrolling
        bf16x8_t a[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a[mt] = *reinterpret_cast<const bf16x8_t*>(smem + src_off[mt][tap] + cbase + j * 32);
#pragma unroll
        for (int t = 0; t < K::NT; ++t) {
            const bf16x8_t b = __builtin_bit_cast(bf16x8_t, bq[n % K::DEPTH][t]);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt], b, acc[mt][t], 0, 0, 0);
        }
        if (n + K::DEPTH < K::NSTEP) {
#pragma unroll
            for (int t = 0; t < K::NT; ++t) bq[n % K::DEPTH][t] = HD_CONV_B(n + K::DEPTH, t);
            // keep the refill HERE: left to itself the scheduler sinks it to just before its use DEPTH steps later (shorter
            // live range), i.e. issues it and waits for it at once -- a ring of depth 1-2 instead of DEPTH
            asm volatile("" ::: "memory");
        }
    }
#undef HD_CONV_B

    // ---- K-slice partials through LDS (slice order), then bias + ReLU + stores; one column tile at a time ----
    if constexpr (KS > 1) __syncthreads();                         // staged faces are dead
#pragma unroll
    for (int t = 0; t < K::NT; ++t) {
        const int col = (tile * K::NT + t) * 32 + (lane & 31);
        const float bias = p.bias[col];
        if constexpr (KS > 1) {
            float* red = reinterpret_cast<float*>(smem);
            if (t > 0) __syncthreads();                            // the previous tile's partials have been read
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int r = (rg * MT + mt) * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
                    red[(ksl * K::BM + r) * 32 + (lane & 31)] = acc[mt][t][i];
                }
            __syncthreads();
            for (int e = tid; e < K::BM * 32; e += K::THREADS) {
                float v = bias;                                    // e & 31 == lane & 31 (THREADS % 32 == 0): same column
#pragma unroll
                for (int s = 0; s < KS; ++s) v += red[s * K::BM * 32 + e];
                v = fmaxf(v, 0.f);
                const int row = row0 + (e >> 5);
                if (row < p.M) {
                    p.out[(size_t)row * C + col] = v;
                    if (p.out16) p.out16[(size_t)row * C + col] = f32_to_bf16_bits(v);
                }
            }
        } else {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int row = row0 + (rg * MT + mt) * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
                    const float v = fmaxf(acc[mt][t][i] + bias, 0.f);
                    if (row < p.M) {
                        p.out[(size_t)row * C + col] = v;
                        if (p.out16) p.out16[(size_t)row * C + col] = f32_to_bf16_bits(v);
                    }
                }
        }
    }
}

template <class K>
inline hipError_t launch_hca_conv(const ConvP& p, hipStream_t s) {
    if (K::SMEM > 65536) {
        static std::atomic<unsigned long long> granted{0};
        { const hipError_t e = grant_dynamic_lds(reinterpret_cast<const void*>(&hca_conv_kernel<K>), K::SMEM, granted); if (e != hipSuccess) return e; }
    }
    hipLaunchKernelGGL((hca_conv_kernel<K>), dim3((p.M + K::BM - 1) / K::BM, K::C / (32 * K::NT)), dim3(K::THREADS), K::SMEM, s, p);
    return hipGetLastError();
}

// Shapes of the refiner at latent 16: level l has C = 128 << l channels and faces of side 16 >> l.
typedef ConvCfg<128, 16, 256, 2, 2, HD_CONV_DEPTH> ConvL0;    // one face per workgroup: 4 row groups x 2 K-halves
typedef ConvCfg<256, 8, 128, 2, 4, HD_CONV_DEPTH> ConvL1;     // two faces: 2 row groups x 4 K-quarters
typedef ConvCfg<512, 4, 64, 2, 8, HD_CONV_DEPTH> ConvL2;      // four faces: 1 row group x 8 K-slices
typedef ConvCfg<1024, 2, 32, 1, 8, 16> ConvL3; // eight faces; 590 KB of weights per workgroup: deeper ring
// latent 32: faces of side 32 >> l; level 0's 32x32 faces do not fit LDS: strips of 8 image rows + halo (87 KB)
// (r04: one column tile per workgroup was 46.0 / 30.3 / 26.5 / 30.2 / 40.2 us for hcas.0 .. hcas.4 at batch 64; NT = 2 everywhere and 4 at
// level 0: 43.8 / 26.4 / 29.4 / 27.3 / 29.1 -- level 2 loses its second workgroup per CU (167 registers; with half the ring and 128 registers asked of the compiler: 26.4, 5 spilled) and keeps NT = 1)
#ifndef HD_CONV_L2X32_NT
#define HD_CONV_L2X32_NT 1
#endif
typedef ConvCfg<128, 32, 256, 2, 2, 4, true, 4> ConvL0x32;
typedef ConvCfg<256, 16, 256, 2, 2, 8, false, 2> ConvL1x32;
typedef ConvCfg<512, 8, 64, 2, 8, (HD_CONV_L2X32_NT > 1 ? 4 : 8), false, HD_CONV_L2X32_NT> ConvL2x32;   // one face per workgroup, two per CU (two faces, one workgroup per CU: 28.9 against 26.6 us)
typedef ConvCfg<1024, 4, 64, 2, 8, 8, false, 2> ConvL3x32;   // four faces (134 KB): 16 row groups instead of 32 re-read every weight tile (42.9 -> 30.2 us)
typedef ConvCfg<2048, 2, 32, 1, 8, 8, false, 2> ConvL4x32;

}  // namespace hd
