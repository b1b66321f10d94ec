// hd_end.hpp — the step's last two launches as one (latent 16, conditional refiner): the HCA conv of level 0 (hcas.4: 3x3 conv 128 -> 128
// on the gated decoder output, BatchNorm folded, ReLU; models/fpg/hca.py:21-23,29) and the ending conv (Conv2d(128,4,3,pad 1),
// models/denoiser/model.py:168-176,261) with the scheduler update in its epilogue (hd_kernels.hpp: ending_conv_kernel).
//
// The ending conv is the only reader of the HCA conv's output, and both are 3x3 convs inside one 16 x 16 face: a workgroup that owns 4 image
// rows of a face (64 pixels, 256 workgroups at batch 64) computes the HCA output for those rows and the image row above and below
// (96 pixels = three 32-row MFMA tiles: 1.5x the MFMA work of the conv alone, which is 5 % of the pipe either way), keeps it in LDS and runs
// the ending conv + scheduler update of its 64 pixels from there.  What disappears: a launch boundary with its cold start, and the fp32 round
// trip of the HCA output through HBM (8.4 MB written and read).
//
// Arithmetic is that of the two launches, operation for operation: the HCA conv's K = 9 x 128 is split over two wave groups by input channel
// (ConvL0's KS = 2: the same slices, the same order of taps and k-steps, partials added in slice order on top of the bias), the ending conv
// is ending_conv_kernel's loop with its column loader reading LDS instead of global memory.  Results are bit-identical to the two launches.
#pragma once
#include "hd_conv.hpp"
#include "hd_kernels.hpp"

namespace hd {

struct EndP {
    int B;                            // faces
    const unsigned short* Xg;         // [B * 256][128] bf16: the gated decoder output (f_d * (1 + w_c + w_s))
    const uint4* W;                   // HCA conv weights, packed [4 tiles][72 k-steps][64] (k = tap * 128 + c)
    const float* bias;                // [128] (BatchNorm folded)
    const float* ewT; const float* eb;   // ending conv weights re-laid [tap][co][ci] (ending_weight_layout_kernel), bias [4]
    float* eps;                       // [B][4][16][16] NCHW
    SchedArgs sa;                     // scheduler update + FiLM staging of the sampling loop (lat == NULL: plain evaluation)
};

struct EndCfg {
    static constexpr int C = 128, S = 16, OWNR = 4, YR = OWNR + 2, XR = OWNR + 4;      // image rows: own, with the HCA output's halo, with the input's halo
    static constexpr int ROWB = C * 2 + 16;                          // bytes per staged input pixel
    static constexpr int NP = XR * S;                                // staged input pixels (+ one zero row)
    static constexpr int MT = YR * S / 32;                           // 32-row MFMA tiles of the HCA output: 3
    static constexpr int SPT = C / 16, KSL = 2, CPW = SPT / KSL, NSTEP = 9 * CPW, KSTEPS = 9 * SPT, DEPTH = 8;
    static constexpr int THREADS = 512;
    static constexpr int XIN = (NP + 1) * ROWB;                      // 35,088 B
    static constexpr int RED = KSL * YR * S * C * 4;                 // 98,304 B: Y (fp32 [96][128]) and the partial tiles of K slice 1
    static constexpr int SMEM = RED > XIN ? RED : XIN;
};

static __global__ __launch_bounds__(EndCfg::THREADS) void hca_ending_conv_kernel(const EndP p) {
    typedef EndCfg K;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int C = K::C, S = K::S;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int face = blockIdx.x >> 2, qd = blockIdx.x & 3;
    const int ct = wave & 3, ksl = wave >> 2;                        // this wave's 32 output channels and K slice (input channels 64 ksl ..)
    const int y0 = qd * K::OWNR;                                     // first own image row

    // ---- the next step's FiLM row to its fixed address (ending_conv_kernel's trailing workgroups): every workgroup moves its share ----
    if (p.sa.lat) {
        const int step = p.sa.st->step;
        if (step + 1 < p.sa.st->n_steps) {
            const float4* src = reinterpret_cast<const float4*>(p.sa.film_table + (size_t)(step + 1) * p.sa.film_total);
            for (int i = blockIdx.x * K::THREADS + tid; i < p.sa.film_total / 4; i += gridDim.x * K::THREADS) reinterpret_cast<float4*>(p.sa.film_cur)[i] = src[i];
        }
    }
    // ---- the ending conv's 72 weights of this lane (input channels lane, lane + 64): requested now, used after the HCA conv ----
    float wl[9][4][2];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int co = 0; co < 4; ++co) {
            wl[tap][co][0] = p.ewT[(tap * 4 + co) * 128 + lane];
            wl[tap][co][1] = p.ewT[(tap * 4 + co) * 128 + lane + 64];
        }
    // ---- weights first (hca_conv_kernel's ring): step n = tap * CPW + j reads k-step tap * SPT + ksl * CPW + j of column tile ct ----
    uint4 bq[K::DEPTH];
    const uint4* Wl = p.W + ((size_t)ct * K::KSTEPS + ksl * K::CPW) * 64 + lane;
#define HD_END_B(n) Wl[(size_t)(((n) / K::CPW) * K::SPT + ((n) % K::CPW)) * 64]
#pragma unroll
    for (int d = 0; d < K::DEPTH; ++d) bq[d] = HD_END_B(d);

    // ---- stage the input rows y0 - 2 .. y0 + 5 of the face (rows outside the image: zeros) and the zero row ----
    {
        constexpr int PPR = C / 8, NIT = K::NP * PPR / K::THREADS;   // 16 pieces per pixel, 4 per thread
        static_assert(K::NP * PPR % K::THREADS == 0, "staging");
        uint4 stage[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid + it * K::THREADS, px = i / PPR, q = i - px * PPR;
            const int yy = y0 - 2 + px / S;
            const bool in = yy >= 0 && yy < S;
            const uint4 v = reinterpret_cast<const uint4*>(p.Xg)[((size_t)face * 256 + (in ? yy : y0) * S + px % S) * PPR + q];
            stage[it] = in ? v : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid + it * K::THREADS, px = i / PPR, q = i - px * PPR;
            *reinterpret_cast<uint4*>(smem + px * K::ROWB + q * 16) = stage[it];
        }
        for (int i = tid; i < K::ROWB / 16; i += K::THREADS) *reinterpret_cast<uint4*>(smem + K::NP * K::ROWB + i * 16) = make_uint4(0, 0, 0, 0);
    }
    // ---- per lane: LDS byte offset of the source pixel of (row tile mt, tap); outside the image: the zero row ----
    int src_off[K::MT][9];
    {
        const int r = lane & 31, h = lane >> 5;
#pragma unroll
        for (int mt = 0; mt < K::MT; ++mt) {
            const int pl = mt * 32 + r, yi = pl / S, x = pl - yi * S;              // HCA output pixel: image row y0 - 1 + yi
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int yy = y0 - 1 + yi + t / 3 - 1, xx = x + t % 3 - 1;
                const bool in = yy >= 0 && yy < S && xx >= 0 && xx < S;
                src_off[mt][t] = (in ? ((yy - (y0 - 2)) * S + xx) : K::NP) * K::ROWB + h * 16;
            }
        }
    }
    f32x16_t acc[K::MT];
#pragma unroll
    for (int mt = 0; mt < K::MT; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[mt][i] = 0.f;
    __syncthreads();

    // ---- K loop, fully unrolled: 9 taps x CPW k-steps of this wave's channel slice ----
    const int cbase = ksl * K::CPW * 32;
#pragma unroll
    for (int n = 0; n < K::NSTEP; ++n) {
        const int tap = n / K::CPW, j = n % K::CPW;
        const bf16x8_t b = __builtin_bit_cast(bf16x8_t, bq[n % K::DEPTH]);
#pragma unroll
        for (int mt = 0; mt < K::MT; ++mt)
            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_t*>(smem + src_off[mt][tap] + cbase + j * 32), b, acc[mt], 0, 0, 0);
        if (n + K::DEPTH < K::NSTEP) {
            bq[n % K::DEPTH] = HD_END_B(n + K::DEPTH);
            asm volatile("" ::: "memory");                           // keep the refill here (hca_conv_kernel)
        }
    }
#undef HD_END_B

    // ---- Y = relu((bias + slice 0) + slice 1): slice 1's partial tiles through LDS, slice 0's waves add theirs from registers (hca_conv_kernel's
    //      order: bias, slice 0, slice 1) and write Y, fp32 [96][128]; rows outside the image: 0 (the ending conv's padding) ----
    __syncthreads();                                                 // the staged input is dead
    float* red = reinterpret_cast<float*>(smem);                     // Y
    float* red1 = red + K::YR * S * C;                               // slice 1's partials
    if (ksl == 1) {
#pragma unroll
        for (int mt = 0; mt < K::MT; ++mt)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int r = mt * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
                red1[r * C + ct * 32 + (lane & 31)] = acc[mt][i];
            }
    }
    __syncthreads();
    if (ksl == 0) {
        const int col = ct * 32 + (lane & 31);
        const float bias = p.bias[col];
#pragma unroll
        for (int mt = 0; mt < K::MT; ++mt)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int r = mt * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
                const int yy = y0 - 1 + r / S;
                float v = bias;
                v += acc[mt][i];
                v += red1[r * C + col];
                v = fmaxf(v, 0.f);
                red[r * C + col] = (yy >= 0 && yy < S) ? v : 0.f;
            }
    }
    __syncthreads();

    // ---- ending conv + scheduler update of the 64 own pixels: ending_conv_kernel's loop, one run of 8 pixels per wave, columns from LDS ----
    constexpr int PXS = 8;
    const int ro = wave >> 1, x0 = (wave & 1) * PXS;                 // own image row y0 + ro = Y row ro + 1
    auto load_col = [&](int x, float (&c)[3][2]) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const bool in = x >= 0 && x < S;                          // wave-uniform; rows outside the image are zero rows of Y
            const float* q = red + ((ro + r) * S + (in ? x : x0)) * C;
            const float v0 = q[lane], v1 = q[lane + 64];
            c[r][0] = in ? v0 : 0.f; c[r][1] = in ? v1 : 0.f;
        }
    };
    float cl[3][2], cc[3][2], cr[3][2];
    load_col(x0 - 1, cl);
    load_col(x0, cc);
    float part[PXS][4];
#pragma unroll
    for (int px = 0; px < PXS; ++px) {
        load_col(x0 + px + 1, cr);
#pragma unroll
        for (int co = 0; co < 4; ++co) {
            float a = 0.f;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                a += cl[r][0] * wl[r * 3 + 0][co][0] + cl[r][1] * wl[r * 3 + 0][co][1];
                a += cc[r][0] * wl[r * 3 + 1][co][0] + cc[r][1] * wl[r * 3 + 1][co][1];
                a += cr[r][0] * wl[r * 3 + 2][co][0] + cr[r][1] * wl[r * 3 + 2][co][1];
            }
            part[px][co] = a;
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) { cl[r][0] = cc[r][0]; cl[r][1] = cc[r][1]; cc[r][0] = cr[r][0]; cc[r][1] = cr[r][1]; }
    }
    float mine = 0.f;
#pragma unroll
    for (int px = 0; px < PXS; ++px)
#pragma unroll
        for (int co = 0; co < 4; ++co) {
            const float t = wave_sum(part[px][co]);
            if (lane == px * 4 + co) mine = t;
        }
    if (lane >= 4 * PXS) return;
    const int co = lane & 3, x = x0 + (lane >> 2), y = y0 + ro;
    const size_t o = (((size_t)face * 4 + co) * S + y) * S + x;       // NCHW
    const float e = mine + p.eb[co];
    p.eps[o] = e;
    if (p.sa.lat) {
        const int step = p.sa.st->step;
        p.sa.lat[o] = sched_update(p.sa.lat[o], e, p.sa.coef + (size_t)step * 7, p.sa.st, step, (size_t)p.sa.elem0 + o, p.sa.n_total);
    }
}

inline hipError_t launch_hca_ending(const EndP& p, hipStream_t s) {
    static std::atomic<unsigned long long> granted{0};
    { const hipError_t e = grant_dynamic_lds(reinterpret_cast<const void*>(&hca_ending_conv_kernel), EndCfg::SMEM, granted); if (e != hipSuccess) return e; }
    hipLaunchKernelGGL(hca_ending_conv_kernel, dim3(p.B * 4), dim3(EndCfg::THREADS), EndCfg::SMEM, s, p);
    return hipGetLastError();
}

}  // namespace hd
