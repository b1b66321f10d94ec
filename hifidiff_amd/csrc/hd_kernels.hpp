// hd_kernels.hpp — the non-GEMM kernels of the refiner path (gfx950).  All activations are
// channels-last fp32 [rows = (face, y, x)][C] unless stated; see DESIGN.md for the buffer map.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hd_gemm.hpp"

namespace hd {

// Device-resident loop state read by the captured per-step graph (so one graph serves every step).
struct StepState {
    int step;                 // index into the schedule; incremented by the first kernel of a step
    int n_steps;              // length of the schedule (0: single evaluation)
    const float* noise;       // [n_steps][n_elems] or NULL -> Philox
    unsigned long long seed;
};

// ----------------------------------------------------------------------------------- weight packing
// fp32 conv/linear weight [N][Cin][KH][KW] -> bf16 MFMA B-fragment order [nt][kstep][lane][8] with
// k = tap*Cin_pad + c, optional per-output-channel scale (folded BatchNorm), optional centre-tap
// slice (3x3 conv on a 1x1 map: only the centre tap ever sees data), optional output permutation
// n' = pos*Cg + c  <-  n = c*S2 + pos  (idc_conv: (B,2048*s*s,1,1) viewed as (B,2048,s,s), model.py:246).
struct PackP {
    const float* src; uint4* dst;
    int N, Cin, Cin_pad, KH, KW, ntaps, centre_only, S2, Kp, nt_total;
    const float* nscale;
};
static __global__ void pack_weight_kernel(const PackP p) {
    const int ksteps = p.Kp >> 4;
    const size_t total = (size_t)p.nt_total * ksteps * 64;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int lane = (int)(e & 63);
        const size_t tk = e >> 6;
        const int ks = (int)(tk % ksteps), nt = (int)(tk / ksteps);
        const int np = nt * 32 + (lane & 31);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = ks * 16 + 8 * (lane >> 5) + j;
            const int tap = k / p.Cin_pad, c = k - tap * p.Cin_pad;
            float x = 0.f;
            if (np < p.N && tap < p.ntaps && c < p.Cin) {
                int n = np;
                if (p.S2 > 1) { const int cg = p.N / p.S2; n = (np % cg) * p.S2 + np / cg; }
                int ky, kx;
                if (p.centre_only) { ky = p.KH >> 1; kx = p.KW >> 1; }
                else { ky = tap / p.KW; kx = tap - ky * p.KW; }
                x = p.src[(((size_t)n * p.Cin + c) * p.KH + ky) * p.KW + kx];
                if (p.nscale) x *= p.nscale[n];
            }
            v[j] = x;
        }
        p.dst[e] = pack8(v);
    }
}

// 1x1 conv weight [N][K] fp32 -> bf16 in the A-operand order of v_mfma_f32_16x16x32_bf16 (hd_xcd2.hpp: the WEIGHTS are the A operand,
// 16 output channels x 32 k per fragment): dst[(mb * K/32 + ks) * 64 + lane] = W[mb*16 + (lane & 15)][ks*32 + 8*(lane >> 4) .. + 7]
static __global__ void pack_weight16_kernel(const float* __restrict__ src, uint4* __restrict__ dst, int N, int K) {
    const int ksteps = K >> 5;
    const size_t total = (size_t)(N >> 4) * ksteps * 64;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int lane = (int)(e & 63);
        const size_t tk = e >> 6;
        const int ks = (int)(tk % ksteps), mb = (int)(tk / ksteps);
        const float* q = src + (size_t)(mb * 16 + (lane & 15)) * K + ks * 32 + 8 * (lane >> 4);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = q[j];
        dst[e] = pack8(v);
    }
}

// depthwise weights [2C][9] -> tap-major [9][2C]: the fused conv1 epilogue reads them with one coalesced load per tap
static __global__ void dw_weight_layout_kernel(const float* __restrict__ w, float* __restrict__ wT, int n2c) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n2c * 9) return;
    const int ch = i / 9, t = i - ch * 9;
    wT[(size_t)t * n2c + ch] = w[i];
}

// ----------------------------------------------------------------------------------- intro / ending
// intro: Conv2d(4,128,3,pad 1) on the NCHW latent -> channels-last fp32 + bf16 copy + LayerNorm partial
// (models/denoiser/model.py:159-167,235).  fp32 FMA (K = 36 is too small for MFMA).  One wave per run of 16
// pixels of an image row, lanes over output channels (co = lane, lane + 64): the lane's 72 weights stay in
// registers (wT is the weight re-laid as [ci*9 + tap][co]: coalesced loads), the 3 x 18 x 4 latent patch of
// the run sits in LDS and is read as broadcasts.  The first thread also advances the loop's step counter.
static __global__ void intro_weight_layout_kernel(const float* __restrict__ w, float* __restrict__ wT) {    // w[co][36] -> wT[36][co]
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 128 * 36) return;
    const int co = i / 36;
    wT[(i - co * 36) * 128 + co] = w[i];
}
// PXS: pixels per wave.  One wave per run of 16 pixels gave one wave per SIMD at the benchmark batch (1024 waves): a serial
// chain of LDS reads, FMAs, DPP reductions and stores with nothing to overlap it; runs of 8 put two waves on a SIMD
// (measured per launch, runs of 16 / 8 / 4 / 2: intro 13.4 / 9.5 / 9.5 / 10.8 us, ending 11.8 / 8.8 / 9.8 / 12.8 us).  From
// kLongRunRows rows on (latent 32 at batch 64) runs of 16 already give four waves per SIMD and re-read fewer halo columns.
constexpr int kIntroPx = 8, kEndingPx = 8, kLongRunRows = 32768;
template <int PXS>
__global__ __launch_bounds__(256) void intro_conv_kernel(const float* __restrict__ lat, const float* __restrict__ wT,
                                                          const float* __restrict__ b, float* __restrict__ out,
                                                          unsigned short* __restrict__ out16, float2* __restrict__ stats, int B, int L,
                                                          StepState* st, int advance) {
    __shared__ float patch[4][4][3][PXS + 4];                            // [wave][ci][row][x0-1 .. x0+PXS] (+pad)
    if (advance && blockIdx.x == 0 && threadIdx.x == 0) st->step += 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nseg = B * L * (L / PXS);
    const int seg = blockIdx.x * 4 + wave;
    if (seg >= nseg) return;                                             // whole wave; no block-wide barrier below
    const int spr = L / PXS;
    const int bb = seg / (L * spr), rem = seg - bb * L * spr, y = rem / spr, x0 = (rem - y * spr) * PXS;
    for (int i = lane; i < 4 * 3 * (PXS + 2); i += 64) {
        const int ci = i / (3 * (PXS + 2)), r = (i - ci * 3 * (PXS + 2)) / (PXS + 2), xx = i - ci * 3 * (PXS + 2) - r * (PXS + 2);
        const int yy = y + r - 1, x = x0 + xx - 1;
        patch[wave][ci][r][xx] = (yy >= 0 && yy < L && x >= 0 && x < L) ? lat[((size_t)(bb * 4 + ci) * L + yy) * L + x] : 0.f;
    }
    float wl[36][2];
#pragma unroll
    for (int r = 0; r < 36; ++r) { wl[r][0] = wT[r * 128 + lane]; wl[r][1] = wT[r * 128 + lane + 64]; }
    const float b0 = b[lane], b1 = b[lane + 64];
    __builtin_amdgcn_wave_barrier();
    const size_t row0 = ((size_t)bb * L + y) * L + x0;
#pragma unroll
    for (int px = 0; px < PXS; ++px) {
        float a0 = b0, a1 = b1;
#pragma unroll
        for (int ci = 0; ci < 4; ++ci)
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float v = patch[wave][ci][r][px + k];
                    a0 += v * wl[ci * 9 + r * 3 + k][0];
                    a1 += v * wl[ci * 9 + r * 3 + k][1];
                }
        const size_t o = (row0 + px) * 128;
        out[o + lane] = a0; out[o + lane + 64] = a1;
        out16[o + lane] = f32_to_bf16_bits(a0); out16[o + lane + 64] = f32_to_bf16_bits(a1);
        // LayerNorm statistics of the 128-channel row for the first block's norm1: (mean, M2), one partial
        const float s1 = wave_sum(a0 + a1), s2 = wave_sum(a0 * a0 + a1 * a1);
        const float mean = s1 * (1.0f / 128.0f);
        if (lane == 0) stats[row0 + px] = make_float2(mean, fmaxf(s2 - s1 * mean, 0.f));
    }
}

// ending: Conv2d(128,4,3,pad 1) channels-last fp32 -> NCHW eps (models/denoiser/model.py:168-176,261), and --
// in the sampling loop -- the scheduler update of the same elements (sched_update below) plus the staging of
// the next step's FiLM row, so the step ends with this launch.
// One wave per run of 16 pixels of an image row; lanes over input channels (ci = lane, lane + 64).  The lane's
// 72 weights stay in registers for the whole run (wT is the weight re-laid as [tap][co][ci], so these are
// coalesced loads) and the 3x3 window slides along x (one new column = 6 loads per pixel).  The 64 per-lane
// partial sums (16 pixels x 4 outputs) are reduced over the wave with DPP.
struct SchedArgs {
    float* lat;                   // chain-local latents x_t -> x_{t-1} (NULL: plain eps evaluation)
    const float* coef;            // [n_steps][7]
    const StepState* st;
    int elem0, n_total;           // position of this chain's elements inside the whole batch (noise indexing)
    const float* film_table; float* film_cur; int film_total;
};
__device__ __forceinline__ float philox_normal(unsigned long long seed, unsigned step, unsigned idx);
// x0 = clamp((x - c0*eps)/c1, +-c2);  x <- c3*x0 + c4*x + c5*eps + c6*z      (hd_schedule in the C-ABI)
__device__ __forceinline__ float sched_update(float xv, float e, const float* c, const StepState* st, int step, size_t gi, int n_total) {
    float x0 = (xv - c[0] * e) / c[1];
    x0 = fminf(fmaxf(x0, -c[2]), c[2]);
    float r = c[3] * x0 + c[4] * xv + c[5] * e;
    if (c[6] != 0.f) {
        // (the pointer is read from device memory: say that it is a global pointer, or the load is a flat load with a full wait)
        const float z = st->noise ? ((const __attribute__((address_space(1))) float*)st->noise)[(size_t)step * n_total + gi] : philox_normal(st->seed, (unsigned)step, (unsigned)gi);
        r += c[6] * z;
    }
    return r;
}
static __global__ void ending_weight_layout_kernel(const float* __restrict__ w, float* __restrict__ wT) {   // w[co][ci][tap] -> wT[tap][co][ci]
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 4 * 128 * 9) return;
    const int co = i / (128 * 9), r = i - co * 128 * 9, ci = r / 9, tap = r - ci * 9;
    wT[(tap * 4 + co) * 128 + ci] = w[i];
}
template <int PXS>
__global__ __launch_bounds__(256) void ending_conv_kernel(const float* __restrict__ X, const float* __restrict__ wT,
                                                           const float* __restrict__ b, float* __restrict__ eps,
                                                           int B, int L, const SchedArgs sa) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nseg = B * L * (L / PXS);                                  // runs of PXS pixels (L is a multiple of 16)
    const int nb_conv = (nseg + 3) >> 2;
    if ((int)blockIdx.x >= nb_conv) {
        // trailing workgroups: stage the NEXT step's FiLM row at a fixed address, so that no LayerNorm loader of
        // the next replay has to chase the step index through memory before it can fetch its gain/bias
        const int step = sa.st->step;
        if (step + 1 >= sa.st->n_steps) return;
        const float4* src = reinterpret_cast<const float4*>(sa.film_table + (size_t)(step + 1) * sa.film_total);
        const int i = ((int)blockIdx.x - nb_conv) * 256 + (int)threadIdx.x;
        if (i < sa.film_total / 4) reinterpret_cast<float4*>(sa.film_cur)[i] = src[i];
        return;
    }
    const int seg = blockIdx.x * 4 + wave;
    if (seg >= nseg) return;                                             // whole wave; no block-wide barrier below
    const int spr = L / PXS;                                             // runs per image row
    const int bb = seg / (L * spr), rem = seg - bb * L * spr, y = rem / spr, x0 = (rem - y * spr) * PXS;
    float wl[9][4][2];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int co = 0; co < 4; ++co) {
            wl[tap][co][0] = wT[(tap * 4 + co) * 128 + lane];
            wl[tap][co][1] = wT[(tap * 4 + co) * 128 + lane + 64];
        }
    // column loader: the three rows y-1, y, y+1 at image column x (zeros outside the image)
    const float* face = X + (size_t)bb * L * L * 128;
    auto load_col = [&](int x, float (&c)[3][2]) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int yy = y + r - 1;
            const bool in = yy >= 0 && yy < L && x >= 0 && x < L;      // wave-uniform
            const float* q = face + ((size_t)(in ? yy : y) * L + (in ? x : x0)) * 128;
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
    // wave reduction of the 4 PXS partial sums; lane j keeps value j = (pixel j >> 2, output j & 3)
    float mine = 0.f;
#pragma unroll
    for (int px = 0; px < PXS; ++px)
#pragma unroll
        for (int co = 0; co < 4; ++co) {
            const float t = wave_sum(part[px][co]);
            if (lane == px * 4 + co) mine = t;
        }
    if (lane >= 4 * PXS) return;
    const int co = lane & 3, x = x0 + (lane >> 2);
    const size_t o = (((size_t)bb * 4 + co) * L + y) * L + x;           // NCHW
    const float e = mine + b[co];
    eps[o] = e;
    if (sa.lat) {
        const int step = sa.st->step;
        sa.lat[o] = sched_update(sa.lat[o], e, sa.coef + (size_t)step * 7, sa.st, step, (size_t)sa.elem0 + o, sa.n_total);
    }
}

// ----------------------------------------------------------------------- depthwise 3x3 + gate + pool
// conv2 (depthwise 3x3, pad 1) -> SimpleGate -> G (bf16) and the SCA global average pool
// (conditional_naf.py:116-119 / naf.py:109-112).  T1 is the fp32 conv1 output [rows][2C].
// grid (C/32, faces); a workgroup owns one face x 32 gate channels, so the pooled mean is complete
// without atomics.
// Unfused form (faces too large for the fused conv1 epilogue: side > 16).  grid (C/32, faces, bands): a workgroup
// owns 32 gate channels x a band of 8 image rows of one face; each of its 8 row workers slides a 3x3 window along
// its row (6 new values per pixel per gate half instead of 18).  Band sums go to pool_part[face][band][C]; the
// mean over the face is finished by dwconv_pool_finish_kernel in band order (no atomics: reproducible).
static __global__ __launch_bounds__(256) void dwconv_gate_pool_kernel(const float* __restrict__ T1, const float* __restrict__ w2,
                                                                const float* __restrict__ b2,
                                                                unsigned short* __restrict__ G, float* __restrict__ pool_part,
                                                                int H, int W, int C) {
    __shared__ float red[8][32];
    const int j = blockIdx.x * 32 + (threadIdx.x & 31);
    const int rw = threadIdx.x >> 5;
    const int face = blockIdx.y, band = blockIdx.z, nbands = gridDim.z;
    const int HW = H * W, C2 = 2 * C;
    float wa[9], wb[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) { wa[t] = w2[(size_t)j * 9 + t]; wb[t] = w2[(size_t)(j + C) * 9 + t]; }
    const float ba = b2[j], bb = b2[j + C];
    const float* base = T1 + (size_t)face * HW * C2;
    const int y = band * 8 + rw;
    float sum = 0.f;
    if (y < H) {
        // column x of the three rows y-1, y, y+1 for both gate halves (zeros outside the image)
        auto load_col = [&](int x, float (&a)[3], float (&b)[3]) {
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int yy = y + r - 1;
                const bool in = yy >= 0 && yy < H && x >= 0 && x < W;
                const float* q = base + (size_t)(in ? yy * W + x : 0) * C2;
                const float va = q[j], vb = q[j + C];
                a[r] = in ? va : 0.f; b[r] = in ? vb : 0.f;
            }
        };
        float la[3], lb[3], ca[3], cb[3], ra[3], rb[3];
        load_col(-1, la, lb);
        load_col(0, ca, cb);
        for (int x = 0; x < W; ++x) {
            load_col(x + 1, ra, rb);
            float u1 = ba, u2 = bb;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                u1 += wa[r * 3] * la[r] + wa[r * 3 + 1] * ca[r] + wa[r * 3 + 2] * ra[r];
                u2 += wb[r * 3] * lb[r] + wb[r * 3 + 1] * cb[r] + wb[r * 3 + 2] * rb[r];
            }
            const float g = u1 * u2;
            G[((size_t)face * HW + y * W + x) * C + j] = f32_to_bf16_bits(g);
            sum += g;
#pragma unroll
            for (int r = 0; r < 3; ++r) { la[r] = ca[r]; lb[r] = cb[r]; ca[r] = ra[r]; cb[r] = rb[r]; }
        }
    }
    red[rw][threadIdx.x & 31] = sum;
    __syncthreads();
    if (rw == 0) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) s += red[i][threadIdx.x];
        pool_part[((size_t)face * nbands + band) * C + j] = s;
    }
}
static __global__ void dwconv_pool_finish_kernel(const float* __restrict__ pool_part, float* __restrict__ pooled, int faces, int nbands, int C, float inv_hw) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= faces * C) return;
    const int face = i / C, j = i - face * C;
    float s = 0.f;
    for (int b = 0; b < nbands; ++b) s += pool_part[((size_t)face * nbands + b) * C + j];
    pooled[i] = s * inv_hw;
}

// -------------------------------------------------------------------------------------- FiLM path
// sinusoidal embedding: [sin(t f_k), cos(t f_k)], k < 64 (models/denoiser/model.py:22-29).
// f_k comes from the host (expf in fp32, like torch.exp on a float32 tensor).
static __global__ void time_embed_kernel(const float* __restrict__ t, const float* __restrict__ freq, float* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * 64) return;
    const int r = i >> 6, k = i & 63;
    const float a = t[r] * freq[k];
    out[(size_t)r * 128 + k] = sinf(a);
    out[(size_t)r * 128 + 64 + k] = cosf(a);
}

// out[m][n] = bias[n] + sum_k in'[m][k] W[n][k], in' = in[m][k]*in[m][k+K] when GATE_IN (SimpleGate on
// the input).  fp32 FMA: time_mlp and the per-block FiLM Linear(256,4C) stay in fp32
// (models/denoiser/model.py:152-157, conditional_naf.py:18-22).  64x64 tile, 4x4 per thread.
template <bool GATE_IN>
__global__ __launch_bounds__(256) void linear_f32_kernel(const float* __restrict__ in, int ldin, const float* __restrict__ Wt,
                                                          const float* __restrict__ bias, float* __restrict__ out, int ldo,
                                                          int M, int N, int K) {
    __shared__ float sa[16][65];
    __shared__ float sb[16][65];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    float acc[4][4] = {};
    for (int k0 = 0; k0 < K; k0 += 16) {
        for (int i = threadIdx.x; i < 64 * 16; i += 256) {
            const int r = i >> 4, kk = i & 15;
            float a = 0.f, b = 0.f;
            if (m0 + r < M && k0 + kk < K) {
                a = in[(size_t)(m0 + r) * ldin + k0 + kk];
                if (GATE_IN) a *= in[(size_t)(m0 + r) * ldin + K + k0 + kk];
            }
            if (n0 + r < N && k0 + kk < K) b = Wt[(size_t)(n0 + r) * K + k0 + kk];
            sa[kk][r] = a; sb[kk][r] = b;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { a[i] = sa[kk][ty * 4 + i]; b[i] = sb[kk][tx * 4 + i]; }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] += a[i] * b[j];
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = m0 + ty * 4 + i, n = n0 + tx * 4 + j;
            if (m < M && n < N) out[(size_t)m * ldo + n] = acc[i][j] + bias[n];
        }
}

// Fold LayerNorm2d's affine into FiLM, in place on the raw Linear output of one block:
//   raw  = [shift_att, scale_att, shift_ffn, scale_ffn]            (conditional_naf.py:110)
//   out  = [b1*(1+scale_att)+shift_att, w1*(1+scale_att), b2*(1+scale_ffn)+shift_ffn, w2*(1+scale_ffn)]
// ln holds [b1, w1, b2, w2] at the same offsets.  grid (ceil(C/256), n_blocks, rows).
struct FilmBlock { int off, C; };
static __global__ void film_fold_kernel(float* __restrict__ film, const float* __restrict__ ln, const FilmBlock* __restrict__ blocks,
                                 int film_total) {
    const FilmBlock fb = blocks[blockIdx.y];
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= fb.C) return;
    float* f = film + (size_t)blockIdx.z * film_total + fb.off;
    const float* l = ln + fb.off;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const float shift = f[2 * h * fb.C + c], scale = f[(2 * h + 1) * fb.C + c];
        const float lb = l[2 * h * fb.C + c], lw = l[(2 * h + 1) * fb.C + c];
        f[2 * h * fb.C + c] = lb * (scale + 1.0f) + shift;
        f[(2 * h + 1) * fb.C + c] = lw * (scale + 1.0f);
    }
}

// ------------------------------------------------------------------------------------- scheduler
// Philox4x32-10 (Salmon et al. 2011), counter (elem, step, 0, 0), key (seed lo, seed hi); two
// Box-Muller normals per call, the first is used.  Restated identically in tests (numpy).
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                                              unsigned* o) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1;
        const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}
__device__ __forceinline__ float philox_normal(unsigned long long seed, unsigned step, unsigned elem) {
    unsigned o[4];
    philox4x32_10(elem, step, 0u, 0u, (unsigned)seed, (unsigned)(seed >> 32), o);
    const float u1 = ((float)(o[0] >> 8) + 1.0f) * (1.0f / 16777216.0f);      // (0, 1]
    const float u2 = (float)(o[1] >> 8) * (1.0f / 16777216.0f);               // [0, 1)
    return sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
}

// The sampling loop applies the scheduler update inside ending_conv_kernel (sched_update); this kernel serves
// hd_scheduler_step.
struct Coef7 { float c[7]; };
static __global__ void sched_step_direct_kernel(float* __restrict__ x, const float* __restrict__ eps, const Coef7 k,
                                         const float* __restrict__ noise, unsigned long long seed, int step, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float xv = x[i], e = eps[i];
    float x0 = (xv - k.c[0] * e) / k.c[1];
    x0 = fminf(fmaxf(x0, -k.c[2]), k.c[2]);
    float r = k.c[3] * x0 + k.c[4] * xv + k.c[5] * e;
    if (k.c[6] != 0.f) r += k.c[6] * (noise ? noise[i] : philox_normal(seed, (unsigned)step, (unsigned)i));
    x[i] = r;
}

// A persistent stage of this call gave up a hand-off wait (abort word set): everything computed since is invalid, so the
// call's result buffer is filled with NaN -- the caller sees the failure in the very tensors it gets back, whatever it does
// with the return code (hd_check() reports it after a synchronisation).  One launch per hd_eps / hd_sample call.
static __global__ void poison_if_abort_kernel(const unsigned* __restrict__ abort_dev, float* __restrict__ out, size_t n) {
    if (*abort_dev == 0u) return;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = __builtin_nanf("");
}

// ------------------------------------------------------------------------------- pooling / layout
// channels-last fp32 -> NCHW fp32 (priors handed out by hd_fpg)
static __global__ void nhwc_to_nchw_f32_kernel(const float* __restrict__ in, float* __restrict__ out, int C, int HW, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int p = (int)(i % HW);
    const size_t r = i / HW;
    const int c = (int)(r % C);
    const size_t b = r / C;
    out[i] = in[(b * HW + p) * C + c];
}

// avg + max pool over the face: HCA channel gate input (models/fpg/hca.py:34-36).  X fp32 [B][HW][C].
static __global__ void pool_avgmax_kernel(const float* __restrict__ X, float* __restrict__ out, int HW, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float* p = X + (size_t)blockIdx.y * HW * C + c;
    float s = 0.f, m = -3.402823466e38f;
    for (int i = 0; i < HW; ++i) { const float v = p[(size_t)i * C]; s += v; m = fmaxf(m, v); }
    out[(size_t)blockIdx.y * C + c] = s / (float)HW + m;
}

// out[row] = sigmoid(dot(Hd[row], w) + bias): spatial gate's Conv2d(C/2,1,1)+BN (folded)+Sigmoid
// (models/fpg/hca.py:16-18).  One wave per row, fp32.
static __global__ __launch_bounds__(256) void rowdot_sigmoid_kernel(const float* __restrict__ Hd, const float* __restrict__ w, float bias,
                                                              float* __restrict__ out, int M, int K) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    float s = 0.f;
    for (int k = lane; k < K; k += 64) s += Hd[(size_t)row * K + k] * w[k];
    s = wave_sum(s);
    if (lane == 0) out[row] = 1.0f / (1.0f + expf(-(s + bias)));
}

// NCHW fp32 -> channels-last fp32 (priors handed in by hd_prepare_from_priors)
static __global__ void nchw_to_nhwc_f32_kernel(const float* __restrict__ in, float* __restrict__ out, int C, int HW, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C);
    const size_t r = i / C;
    const int p = (int)(r % HW);
    const size_t b = r / HW;
    out[i] = in[(b * C + c) * HW + p];
}

// cr_face NCHW fp32 [B,3,H,W] -> channels-last bf16 padded to 8 channels (ResNet conv1 input)
static __global__ void nchw3_to_nhwc8_bf16_kernel(const float* __restrict__ in, uint4* __restrict__ out, int HW, size_t npix) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const size_t b = i / HW, p = i - b * HW;
    float v[8] = {in[(b * 3 + 0) * HW + p], in[(b * 3 + 1) * HW + p], in[(b * 3 + 2) * HW + p], 0, 0, 0, 0, 0};
    out[i] = pack8(v);
}

// MaxPool2d(3, stride 2, pad 1) on channels-last bf16 (models/idc/model.py:112,124)
static __global__ void maxpool3x3s2_bf16_kernel(const unsigned short* __restrict__ in, unsigned short* __restrict__ out,
                                         int B, int H, int W, int C) {
    const int Ho = H / 2, Wo = W / 2;
    const size_t total = (size_t)B * Ho * Wo * C;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C);
    size_t r = i / C;
    const int ox = (int)(r % Wo); r /= Wo;
    const int oy = (int)(r % Ho);
    const int b = (int)(r / Ho);
    float m = -3.402823466e38f;
    for (int ky = 0; ky < 3; ++ky)
        for (int kx = 0; kx < 3; ++kx) {
            const int y = oy * 2 - 1 + ky, x = ox * 2 - 1 + kx;
            if (y < 0 || y >= H || x < 0 || x >= W) continue;
            m = fmaxf(m, bf16_bits_to_f32(in[(((size_t)b * H + y) * W + x) * C + c]));
        }
    out[i] = f32_to_bf16_bits(m);
}

// AdaptiveAvgPool2d(1) on channels-last bf16 -> fp32 [B][C] (models/idc/model.py:131)
static __global__ void avgpool_bf16_kernel(const unsigned short* __restrict__ in, float* __restrict__ out, int HW, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const unsigned short* p = in + (size_t)blockIdx.y * HW * C + c;
    float s = 0.f;
    for (int i = 0; i < HW; ++i) s += bf16_bits_to_f32(p[(size_t)i * C]);
    out[(size_t)blockIdx.y * C + c] = s / (float)HW;
}

}  // namespace hd
