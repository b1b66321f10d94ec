// hd_cr.hpp — kernels that only the CoarseRestoration network needs (models/cr/model.py:33-88,
// models/cr/stn.py:9-52).  Its 32 NAF blocks, down- and up-convs run on the kernels of the refiner path
// (hd_gemm.hpp / hd_chain.hpp); here are the 3<->32 channel image convs, the STN localisation network (kept in
// fp32: its six outputs steer a resampling grid) and the affine bilinear resampler.  This network runs once per
// face before the diffusion loop, so these kernels are written for clarity, not for the roofline.
#pragma once
#include "hd_gemm.hpp"

namespace hd {

// intro: Conv2d(3, 32, 3, pad 1) on the NCHW image -> channels-last fp32 + bf16 copy + LayerNorm partial (1 x 32).
// 32 lanes = the 32 output channels of one pixel.
static __global__ __launch_bounds__(256) void cr_intro_kernel(const float* __restrict__ img, const float* __restrict__ w,
                                                        const float* __restrict__ b, float* __restrict__ out,
                                                        unsigned short* __restrict__ out16, float2* __restrict__ stats, int B, int H) {
    const int co = threadIdx.x & 31;
    const size_t pix = (size_t)blockIdx.x * 8 + (threadIdx.x >> 5);
    const size_t M = (size_t)B * H * H;
    const bool ok = pix < M;
    float acc = 0.f;
    if (ok) {
        const int bb = (int)(pix / ((size_t)H * H)), rem = (int)(pix - (size_t)bb * H * H), y = rem / H, x = rem - y * H;
        acc = b[co];
#pragma unroll
        for (int ci = 0; ci < 3; ++ci)
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
                if (yy >= 0 && yy < H && xx >= 0 && xx < H)
                    acc += img[((size_t)(bb * 3 + ci) * H + yy) * H + xx] * w[(co * 3 + ci) * 9 + t];   // w[co][ci][ky][kx]
            }
        out[pix * 32 + co] = acc;
        out16[pix * 32 + co] = f32_to_bf16_bits(acc);
    }
    const float2 ms = halfwave_mean_m2(ok ? acc : 0.f);            // all 64 lanes take part
    if (ok && co == kStatLane) stats[pix] = ms;
}

// outro: Conv2d(32, 3, 3, pad 1) channels-last fp32 -> NCHW image.  One thread per output pixel.
static __global__ __launch_bounds__(256) void cr_outro_kernel(const float* __restrict__ X, const float* __restrict__ w,
                                                        const float* __restrict__ b, float* __restrict__ out, int B, int H) {
    __shared__ float wt[9][3][32];                                // [tap][co][ci]
    for (int i = threadIdx.x; i < 3 * 32 * 9; i += 256) {
        const int co = i / (32 * 9), r = i - co * 32 * 9, ci = r / 9, t = r - ci * 9;
        wt[t][co][ci] = w[i];
    }
    __syncthreads();
    const size_t pix = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (pix >= (size_t)B * H * H) return;
    const int bb = (int)(pix / ((size_t)H * H)), rem = (int)(pix - (size_t)bb * H * H), y = rem / H, x = rem - y * H;
    float a0 = b[0], a1 = b[1], a2 = b[2];
    for (int t = 0; t < 9; ++t) {
        const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
        if (yy < 0 || yy >= H || xx < 0 || xx >= H) continue;
        const float4* q = reinterpret_cast<const float4*>(X + (((size_t)bb * H + yy) * H + xx) * 32);
#pragma unroll
        for (int c4 = 0; c4 < 8; ++c4) {
            const float4 v = q[c4];
            const float* w0 = &wt[t][0][c4 * 4]; const float* w1 = &wt[t][1][c4 * 4]; const float* w2 = &wt[t][2][c4 * 4];
            a0 += v.x * w0[0] + v.y * w0[1] + v.z * w0[2] + v.w * w0[3];
            a1 += v.x * w1[0] + v.y * w1[1] + v.z * w1[2] + v.w * w1[3];
            a2 += v.x * w2[0] + v.y * w2[1] + v.z * w2[2] + v.w * w2[3];
        }
    }
    const size_t hw = (size_t)H * H, o = (size_t)bb * 3 * hw + (size_t)y * H + x;
    out[o] = a0; out[o + hw] = a1; out[o + 2 * hw] = a2;
}

// STN localisation stage: valid k x k conv -> MaxPool2d(2, 2) -> ReLU (stn.py:23-30), fp32, output NCHW
// [B][Cout][Hp][Hp] (the order .view(-1, fc_size) flattens).  The input is addressed through strides so that the
// same kernel reads the channels-last feature map (first stage) and the NCHW output of the first stage (second).
// One wave per pooled pixel, all COUT output channels: lanes split the Cin*k*k products (each input value is
// loaded once for all channels), DPP wave reductions.
struct StnConvP {
    const float* in; long long sb, sc, sy, sx;                    // element strides of the input
    const float *w, *bias;                                        // w[Cout][Cin][k][k]
    float* out;
    int B, Cin, Hin, k, Cout, Hp;                                 // Hp = (Hin - k + 1) / 2
};
template <int COUT>
__global__ __launch_bounds__(256) void stn_conv_pool_relu_kernel(const StnConvP p) {
    const int lane = threadIdx.x & 63;
    const long long o = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);      // pooled pixel (b, py, px): all COUT channels
    const long long total = (long long)p.B * p.Hp * p.Hp;
    if (o >= total) return;                                        // whole wave
    const int px = (int)(o % p.Hp), py = (int)((o / p.Hp) % p.Hp), bb = (int)(o / ((long long)p.Hp * p.Hp));
    const int kk = p.k * p.k, nprod = p.Cin * kk;
    float acc[COUT][4];
#pragma unroll
    for (int co = 0; co < COUT; ++co) { acc[co][0] = 0.f; acc[co][1] = 0.f; acc[co][2] = 0.f; acc[co][3] = 0.f; }
    const float* base = p.in + bb * p.sb;
    for (int i = lane; i < nprod; i += 64) {                       // the four conv outputs under one pooling window share w
        const int ci = i / kk, r = i - ci * kk, ky = r / p.k, kx = r - ky * p.k;
        const float* q = base + ci * p.sc + (2 * py + ky) * p.sy + (2 * px + kx) * p.sx;
        const float v00 = q[0], v01 = q[p.sx], v10 = q[p.sy], v11 = q[p.sy + p.sx];
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
            const float wv = p.w[(size_t)co * nprod + i];
            acc[co][0] += wv * v00; acc[co][1] += wv * v01; acc[co][2] += wv * v10; acc[co][3] += wv * v11;
        }
    }
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
        const float a = wave_sum(acc[co][0]), b = wave_sum(acc[co][1]), c = wave_sum(acc[co][2]), d = wave_sum(acc[co][3]);
        if (lane == 0)                                             // max commutes with the shared bias
            p.out[(((size_t)bb * COUT + co) * p.Hp + py) * p.Hp + px] = fmaxf(fmaxf(fmaxf(a, b), fmaxf(c, d)) + p.bias[co], 0.f);
    }
}

// fc_loc: theta = W2 relu(W1 xs + b1) + b2 (stn.py:32-36,46-48).  One workgroup per face.
static __global__ __launch_bounds__(256) void stn_fc_kernel(const float* __restrict__ xs, int fc, const float* __restrict__ w1,
                                                      const float* __restrict__ b1, int n1, const float* __restrict__ w2,
                                                      const float* __restrict__ b2, float* __restrict__ theta) {
    __shared__ float h[128];                                      // n1 = floor(sqrt(fc)) <= 85
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* x = xs + (size_t)blockIdx.x * fc;
    for (int j = wave; j < n1; j += 4) {
        float a = 0.f;
        for (int i = lane; i < fc; i += 64) a += w1[(size_t)j * fc + i] * x[i];
        a = wave_sum(a);
        if (lane == 0) h[j] = fmaxf(a + b1[j], 0.f);
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        float a = b2[threadIdx.x];
        for (int j = 0; j < n1; ++j) a += w2[threadIdx.x * n1 + j] * h[j];
        theta[blockIdx.x * 6 + threadIdx.x] = a;
    }
}

// F.affine_grid(theta, size, align_corners=False) + F.grid_sample(bilinear, zeros, align_corners=False) on a
// channels-last map (stn.py:50-51): base grid x_j = (2j + 1)/W - 1; source ix = ((gx + 1) W - 1) / 2.
// Writes fp32 and the bf16 copy the following down/up GEMM loads.  One thread per (pixel, 4 channels).
static __global__ __launch_bounds__(256) void stn_grid_sample_kernel(const float* __restrict__ X, const float* __restrict__ theta,
                                                               float* __restrict__ Y, unsigned short* __restrict__ Y16,
                                                               int B, int H, int C) {
    const int c4n = C >> 2;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)B * H * H * c4n) return;
    const int c4 = (int)(i % c4n);
    const size_t pix = i / c4n;
    const int bb = (int)(pix / ((size_t)H * H)), rem = (int)(pix - (size_t)bb * H * H), y = rem / H, x = rem - y * H;
    const float* t = theta + bb * 6;
    const float xs = (2.0f * x + 1.0f) / (float)H - 1.0f, ys = (2.0f * y + 1.0f) / (float)H - 1.0f;
    const float gx = t[0] * xs + t[1] * ys + t[2], gy = t[3] * xs + t[4] * ys + t[5];
    const float ix = ((gx + 1.0f) * (float)H - 1.0f) * 0.5f, iy = ((gy + 1.0f) * (float)H - 1.0f) * 0.5f;
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy;
    const float wx1 = ix - fx, wy1 = iy - fy, wx0 = 1.0f - wx1, wy0 = 1.0f - wy1;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* face = X + (size_t)bb * H * H * C + c4 * 4;
    auto tap = [&](int yy, int xx, float wgt) {
        if (yy < 0 || yy >= H || xx < 0 || xx >= H) return;
        const float4 v = *reinterpret_cast<const float4*>(face + ((size_t)yy * H + xx) * C);
        acc.x += v.x * wgt; acc.y += v.y * wgt; acc.z += v.z * wgt; acc.w += v.w * wgt;
    };
    tap(y0, x0, wy0 * wx0); tap(y0, x0 + 1, wy0 * wx1); tap(y0 + 1, x0, wy1 * wx0); tap(y0 + 1, x0 + 1, wy1 * wx1);
    const size_t o = pix * C + c4 * 4;
    *reinterpret_cast<float4*>(Y + o) = acc;
    *reinterpret_cast<uint2*>(Y16 + o) = make_uint2(pack2(acc.x, acc.y), pack2(acc.z, acc.w));
}

// x = a + b (decoder input = previous stage + encoder skip, model.py:82-83): fp32, bf16 copy and the LayerNorm
// partial (one per row) of the sum.  One wave per row.
static __global__ __launch_bounds__(256) void add_rows_stats_kernel(const float* __restrict__ A, const float* __restrict__ Bv, float* __restrict__ X,
                                                              unsigned short* __restrict__ X16, float2* __restrict__ stats, int M, int C) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const size_t r0 = (size_t)row * C;
    float s1 = 0.f, s2 = 0.f;
    for (int k = lane * 4; k < C; k += 256) {
        const float4 a = *reinterpret_cast<const float4*>(A + r0 + k), b = *reinterpret_cast<const float4*>(Bv + r0 + k);
        const float4 v = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
        *reinterpret_cast<float4*>(X + r0 + k) = v;
        *reinterpret_cast<uint2*>(X16 + r0 + k) = make_uint2(pack2(v.x, v.y), pack2(v.z, v.w));
        s1 += (v.x + v.y) + (v.z + v.w);
        s2 += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    const float mean = s1 / (float)C;
    if (lane == 0) stats[row] = make_float2(mean, fmaxf(s2 - s1 * mean, 0.f));
}

}  // namespace hd
