// hd_vae.hpp — kernels of the VAE boundary either side of the sampling loop (reference call sites
// test_refiner.py:78-83,93 and train_refiner.py:72-83,122-123): bicubic resize, AutoencoderKL encode -> posterior sample
// -> x 0.18215, and / 0.18215 -> decode.  The network is diffusers' AutoencoderKL (0.32.2; third party, absent here:
// "parity unpinned"): ResnetBlock2D = GroupNorm(32, eps 1e-6) -> SiLU -> conv3x3 twice + shortcut, one single-head
// attention block in each mid block, stride-2 / nearest-2x resampling convs.  The 3x3 / 1x1 convolutions run on the
// implicit-GEMM MFMA kernels of hd_gemm.hpp (bf16 operands, fp32 accumulate, fp32 residual stream); this file holds
// what is not a GEMM.  Activations are channels-last [face, y, x][C].
#pragma once
#include <hip/hip_runtime.h>
#include "hd_gemm.hpp"
#include "hd_kernels.hpp"

namespace hd {

// ---- F.interpolate(x, R, mode="bicubic", align_corners=False) on NCHW fp32 (ATen upsample_bicubic2d: A = -0.75,
// source index (dst + 0.5) * scale - 0.5, taps clamped to the border) ----
__device__ __forceinline__ void cubic_coeffs(float t, float* w) {
    const float A = -0.75f;
    const float x0 = t + 1.f, x1 = t, x2 = 1.f - t, x3 = 2.f - t;
    w[0] = ((A * x0 - 5.f * A) * x0 + 8.f * A) * x0 - 4.f * A;
    w[1] = ((A + 2.f) * x1 - (A + 3.f)) * x1 * x1 + 1.f;
    w[2] = ((A + 2.f) * x2 - (A + 3.f)) * x2 * x2 + 1.f;
    w[3] = ((A * x3 - 5.f * A) * x3 + 8.f * A) * x3 - 4.f * A;
}
static __global__ void bicubic_resize_kernel(const float* __restrict__ in, float* __restrict__ out, int planes, int Hin, int Win, int Hout, int Wout) {
    const size_t total = (size_t)planes * Hout * Wout;
    const float sy = (float)Hin / (float)Hout, sx = (float)Win / (float)Wout;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ox = (int)(i % Wout), oy = (int)((i / Wout) % Hout);
        const size_t pl = i / ((size_t)Wout * Hout);
        const float fy = (oy + 0.5f) * sy - 0.5f, fx = (ox + 0.5f) * sx - 0.5f;
        const int iy = (int)floorf(fy), ix = (int)floorf(fx);
        float wy[4], wx[4];
        cubic_coeffs(fy - (float)iy, wy); cubic_coeffs(fx - (float)ix, wx);
        const float* src = in + pl * Hin * Win;
        float acc = 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int yy = min(max(iy - 1 + a, 0), Hin - 1);
            float r = 0.f;
#pragma unroll
            for (int b = 0; b < 4; ++b) r += wx[b] * src[(size_t)yy * Win + min(max(ix - 1 + b, 0), Win - 1)];
            acc += wy[a] * r;
        }
        out[i] = acc;
    }
}

// NCHW fp32 (C <= 8 channels) -> channels-last bf16 with 8 channels per pixel (zero padded): the conv_in gather source
// vae_range: first x.clamp(0, 1) * 2 - 1 (to_vae_range, train_refiner.py:60-65)
static __global__ void nchw_to_nhwc8_bf16_kernel(const float* __restrict__ in, uint4* __restrict__ out, int C, int HW, size_t npix, int vae_range) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const size_t b = i / HW, px = i - b * HW;
    float v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        float x = c < C ? in[(b * C + c) * HW + px] : 0.f;
        if (vae_range && c < C) x = fminf(fmaxf(x, 0.f), 1.f) * 2.f - 1.f;
        v[c] = x;
    }
    out[i] = pack8(v);
}

// decoder entry: z / 0.18215 -> post_quant_conv (1x1, 4 -> 4) -> channels-last bf16 x 8
static __global__ void vae_decode_entry_kernel(const float* __restrict__ z, const float* __restrict__ w, const float* __restrict__ b, uint4* __restrict__ out,
                                        int HW, size_t npix, float inv_scale) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const size_t f = i / HW, px = i - f * HW;
    float x[4], v[8];
#pragma unroll
    for (int c = 0; c < 4; ++c) x[c] = z[(f * 4 + c) * HW + px] * inv_scale;
#pragma unroll
    for (int o = 0; o < 4; ++o) v[o] = b[o] + w[o * 4] * x[0] + w[o * 4 + 1] * x[1] + w[o * 4 + 2] * x[2] + w[o * 4 + 3] * x[3];
    v[4] = v[5] = v[6] = v[7] = 0.f;
    out[i] = pack8(v);
}

// encoder exit: moments [pixel][8] (conv_out) -> quant_conv (1x1, 8 -> 8) -> DiagonalGaussianDistribution.sample():
// mean + exp(0.5 * clamp(logvar, -30, 20)) * noise, times the scaling factor; NCHW out.  noise: NCHW tensor or Philox.
static __global__ void vae_sample_kernel(const float* __restrict__ mom, const float* __restrict__ w, const float* __restrict__ b,
                                  const float* __restrict__ noise, unsigned long long seed, float* __restrict__ out, int HW, size_t npix,
                                  float scale, int ld) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const size_t f = i / HW, px = i - f * HW;
    float m[8], qv[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) m[c] = mom[i * ld + c];
#pragma unroll
    for (int o = 0; o < 8; ++o) {
        float a = b[o];
#pragma unroll
        for (int c = 0; c < 8; ++c) a = fmaf(w[o * 8 + c], m[c], a);
        qv[o] = a;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float logvar = fminf(fmaxf(qv[4 + c], -30.f), 20.f);
        const size_t e = (f * 4 + c) * HW + px;                     // NCHW element index (also the Philox counter)
        const float z = noise ? noise[e] : philox_normal(seed, 0u, (unsigned)e);
        out[e] = (qv[c] + expf(0.5f * logvar) * z) * scale;
    }
}

// the posterior's parameters themselves: quant_conv(moments) as NCHW [B,8,L,L] (mean | logvar), what
// AutoencoderKL.encode(x).latent_dist is built from
static __global__ void vae_moments_kernel(const float* __restrict__ mom, const float* __restrict__ w, const float* __restrict__ b, float* __restrict__ out,
                                   int HW, size_t npix) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const size_t f = i / HW, px = i - f * HW;
    float m[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) m[c] = mom[i * 8 + c];
#pragma unroll
    for (int o = 0; o < 8; ++o) {
        float a = b[o];
#pragma unroll
        for (int c = 0; c < 8; ++c) a = fmaf(w[o * 8 + c], m[c], a);
        out[(f * 8 + o) * HW + px] = a;
    }
}

// fp32 channels-last [pixel][ld] (first C columns) -> NCHW
static __global__ void nhwc_to_nchw_kernel(const float* __restrict__ in, float* __restrict__ out, int C, int ld, int HW, size_t npix) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix * C) return;
    const size_t px = i % HW, c = (i / HW) % C, f = i / ((size_t)HW * C);
    out[i] = in[(f * HW + px) * ld + c];
}

// nearest 2x upsampling of a channels-last fp32 map into bf16 (the gather source of Upsample2D's conv)
static __global__ void upsample2x_bf16_kernel(const float* __restrict__ in, unsigned short* __restrict__ out, int B, int H, int W, int C) {
    const size_t total = (size_t)B * 2 * H * 2 * W * (C / 8);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % (C / 8));
        const size_t p = i / (C / 8);
        const int ox = (int)(p % (2 * W)), oy = (int)((p / (2 * W)) % (2 * H));
        const size_t f = p / ((size_t)4 * H * W);
        const float* src = in + ((f * H + (oy >> 1)) * W + (ox >> 1)) * C + c8 * 8;
        const float4 a = *reinterpret_cast<const float4*>(src), bq = *reinterpret_cast<const float4*>(src + 4);
        float v[8] = {a.x, a.y, a.z, a.w, bq.x, bq.y, bq.z, bq.w};
        *reinterpret_cast<uint4*>(out + p * C + c8 * 8) = pack8(v);
    }
}

// ---- GroupNorm(32 groups, eps) [+ SiLU] on a channels-last fp32 map -> bf16 (the next conv's MFMA operand) ----
// pass 1: per (face, chunk of pixels): sum and sum of squares of every group.  Coalesced float4 reads; thread t always
// sees the same channel quad (t mod C/4), so its partial sums belong to one group.
constexpr int GN_GROUPS = 32;
static __global__ __launch_bounds__(256) void groupnorm_partial_kernel(const float* __restrict__ x, double* __restrict__ part, int HW, int C, int chunk_px) {
    __shared__ double red[2][256];
    const int f = blockIdx.y, chunk = blockIdx.x, t = threadIdx.x;
    const int quads = C >> 2;                                     // float4 per pixel (C in {128, 256, 512})
    const int px_per_pass = 256 / quads > 0 ? 256 / quads : 1;    // quads <= 256 here
    const int q = t % quads, pofs = t / quads;
    const int p0 = chunk * chunk_px, p1 = min(p0 + chunk_px, HW);
    float s1 = 0.f, s2 = 0.f;
    if (pofs < px_per_pass)
        for (int p = p0 + pofs; p < p1; p += px_per_pass) {
            const float4 v = *reinterpret_cast<const float4*>(x + ((size_t)f * HW + p) * C + 4 * q);
            s1 += (v.x + v.y) + (v.z + v.w);
            s2 += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        }
    red[0][t] = (double)s1; red[1][t] = (double)s2;
    __syncthreads();
    if (t < GN_GROUPS) {
        const int qpg = quads / GN_GROUPS;                        // channel quads per group (1, 2 or 4)
        double a = 0.0, b = 0.0;
        for (int pp = 0; pp < px_per_pass; ++pp)
            for (int j = 0; j < qpg; ++j) { const int idx = pp * quads + t * qpg + j; a += red[0][idx]; b += red[1][idx]; }
        double* o = part + (((size_t)f * gridDim.x + chunk) * GN_GROUPS + t) * 2;
        o[0] = a; o[1] = b;
    }
}
// pass 2: y = (x - mean_g) * rstd_g * gamma_c + beta_c [, SiLU] -> bf16
static __global__ __launch_bounds__(256) void groupnorm_apply_kernel(const float* __restrict__ x, const double* __restrict__ part, int nchunks,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              unsigned short* __restrict__ y, int HW, int C, int chunk_px, float eps, int silu) {
    __shared__ float mr[2][GN_GROUPS];
    const int f = blockIdx.y, chunk = blockIdx.x, t = threadIdx.x;
    if (t < GN_GROUPS) {
        double a = 0.0, b = 0.0;
        for (int k = 0; k < nchunks; ++k) { const double* o = part + (((size_t)f * nchunks + k) * GN_GROUPS + t) * 2; a += o[0]; b += o[1]; }
        const double n = (double)HW * (C / GN_GROUPS);
        const double mean = a / n, var = fmax(b / n - mean * mean, 0.0);
        mr[0][t] = (float)mean; mr[1][t] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    const int oct = C >> 3;                                       // 8-channel units per pixel
    const int p0 = chunk * chunk_px, p1 = min(p0 + chunk_px, HW);
    const size_t units = (size_t)(p1 - p0) * oct;
    for (size_t u = t; u < units; u += 256) {
        const int p = p0 + (int)(u / oct), c0 = (int)(u % oct) * 8;
        const float* src = x + ((size_t)f * HW + p) * C + c0;
        const float4 a = *reinterpret_cast<const float4*>(src), bq = *reinterpret_cast<const float4*>(src + 4);
        float v[8] = {a.x, a.y, a.z, a.w, bq.x, bq.y, bq.z, bq.w};
        const int g = c0 / (C / GN_GROUPS);                       // 8 channels never straddle a group (C/32 in {4, 8, 16}: see below)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ge = (C / GN_GROUPS >= 8) ? g : (c0 + e) / (C / GN_GROUPS);
            float h = (v[e] - mr[0][ge]) * mr[1][ge] * gamma[c0 + e] + beta[c0 + e];
            if (silu) h = h / (1.0f + __expf(-h));
            v[e] = h;
        }
        *reinterpret_cast<uint4*>(y + ((size_t)f * HW + p) * C + c0) = pack8(v);
    }
}

// ---- single-head attention of the mid block (attention_processor.Attention with heads = 1, scale 1/sqrt(C)) ----
// q, k, v: fp32 [face][T][C] (C = 512).  One workgroup = 16 queries of one face; keys are visited in tiles of 64 with an
// online softmax; K is staged in LDS in 128-channel slices, V is streamed.  out: bf16 [face][T][C] (to_out's operand).
constexpr int AT_C = 512, AT_Q = 16, AT_K = 64;
static __global__ __launch_bounds__(256) void vae_attention_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                            unsigned short* __restrict__ out, int T, float scale) {
    extern __shared__ __attribute__((aligned(16))) float at_smem[];
    float* Qs = at_smem;                                          // [16][512]
    float* Ks = Qs + AT_Q * AT_C;                                 // [64][129] slice (padded: conflict-free column reads)
    float* Ps = Ks + AT_K * 129;                                  // [16][64] scores / probabilities
    float* Ms = Ps + AT_Q * AT_K;                                 // [16] running max, [16] running sum, [16] rescale
    const int f = blockIdx.y, q0 = blockIdx.x * AT_Q, t = threadIdx.x;
    const float* qf = q + ((size_t)f * T + q0) * AT_C;
    for (int i = t; i < AT_Q * AT_C / 4; i += 256) reinterpret_cast<float4*>(Qs)[i] = reinterpret_cast<const float4*>(qf)[i];
    if (t < AT_Q) { Ms[t] = -INFINITY; Ms[16 + t] = 0.f; Ms[32 + t] = 0.f; }
    float acc[AT_Q][2];
#pragma unroll
    for (int i = 0; i < AT_Q; ++i) { acc[i][0] = 0.f; acc[i][1] = 0.f; }
    const int sq = t >> 4, sj = t & 15;                           // score ownership: query sq, keys sj + 16 r
    __syncthreads();
    for (int k0 = 0; k0 < T; k0 += AT_K) {
        float s[4] = {0.f, 0.f, 0.f, 0.f};
        for (int cs = 0; cs < AT_C; cs += 128) {
            __syncthreads();
            for (int i = t; i < AT_K * 32; i += 256) {            // 64 keys x 32 float4
                const int key = i >> 5, c4 = i & 31;
                const float4 kv = (k0 + key < T) ? *reinterpret_cast<const float4*>(k + ((size_t)f * T + k0 + key) * AT_C + cs + 4 * c4) : make_float4(0, 0, 0, 0);
                float* d = Ks + key * 129 + 4 * c4;
                d[0] = kv.x; d[1] = kv.y; d[2] = kv.z; d[3] = kv.w;
            }
            __syncthreads();
            const float* qr = Qs + sq * AT_C + cs;
#pragma unroll 4
            for (int c = 0; c < 128; ++c) {
                const float qv = qr[c];
#pragma unroll
                for (int r = 0; r < 4; ++r) s[r] = fmaf(qv, Ks[(sj + 16 * r) * 129 + c], s[r]);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) Ps[sq * AT_K + sj + 16 * r] = (k0 + sj + 16 * r < T) ? s[r] * scale : -INFINITY;
        __syncthreads();
        if (t < AT_Q) {                                           // online softmax bookkeeping of query t
            float m = Ms[t];
            float mx = m;
            for (int j = 0; j < AT_K; ++j) mx = fmaxf(mx, Ps[t * AT_K + j]);
            const float resc = (m == -INFINITY) ? 0.f : __expf(m - mx);
            float sum = 0.f;
            for (int j = 0; j < AT_K; ++j) { const float pv = __expf(Ps[t * AT_K + j] - mx); Ps[t * AT_K + j] = pv; sum += pv; }
            Ms[t] = mx; Ms[16 + t] = Ms[16 + t] * resc + sum; Ms[32 + t] = resc;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < AT_Q; ++i) { const float r = Ms[32 + i]; acc[i][0] *= r; acc[i][1] *= r; }
        const int nk = min(AT_K, T - k0);
        for (int j = 0; j < nk; ++j) {
            const float* vr = v + ((size_t)f * T + k0 + j) * AT_C;
            const float v0 = vr[t], v1 = vr[t + 256];
#pragma unroll
            for (int i = 0; i < AT_Q; ++i) { const float pv = Ps[i * AT_K + j]; acc[i][0] = fmaf(pv, v0, acc[i][0]); acc[i][1] = fmaf(pv, v1, acc[i][1]); }
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < AT_Q; ++i) {
        if (q0 + i < T) {
            const float inv = 1.0f / Ms[16 + i];
            unsigned short* o = out + ((size_t)f * T + q0 + i) * AT_C;
            o[t] = f32_to_bf16_bits(acc[i][0] * inv); o[t + 256] = f32_to_bf16_bits(acc[i][1] * inv);
        }
    }
}
constexpr int AT_SMEM = (AT_Q * AT_C + AT_K * 129 + AT_Q * AT_K + 48) * 4;

}  // namespace hd
