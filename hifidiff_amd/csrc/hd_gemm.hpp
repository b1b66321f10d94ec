// hd_gemm.hpp — the one MFMA GEMM / implicit-conv kernel family of the refiner path (gfx950 only).
//
//   Out[M,N] = epilogue( loader(A)[M,K] * W[K,N] )        bf16 operands, fp32 accumulate
//
// * rows  = pixels (channels-last activations, row = (face, y, x)), cols = output channels.
// * W is pre-packed once (hd_pack.hpp) in MFMA B-fragment order: [N/32][K/16][64 lanes][8 bf16], so a
//   wave streams its weight tiles with fully coalesced 1 KiB loads straight into registers — no LDS
//   round trip for the operand that is read exactly once (the HBM-bound levels 3/mid).
// * A is staged global -> registers -> (loader transform, bf16) -> LDS, double buffered, one barrier
//   per 64-deep K chunk; rows are padded to 144 B so the ds_read_b128 fragment reads are conflict-free.
// * The loader fuses what precedes the conv in the reference: LayerNorm2d + FiLM (utils.py:16-24,
//   conditional_naf.py:114-115), the SCA channel scale (conditional_naf.py:119), the HCA gate
//   (hca.py:28) and the im2col gather of the 2x2/3x3/7x7 convs.
// * The epilogue fuses what follows: bias, SimpleGate (utils.py:57-60; the wave computes column tile j
//   and tile j+N/2 so the product is register-local), beta/gamma residual (conditional_naf.py:123,134),
//   PixelShuffle + skip add (models/denoiser/model.py:204-208,256-257), BN(eval)+ReLU.
// * Two kernels share the loaders/epilogues:
//     gemm_kernel        tall M (levels 0/1, ResNet): A staged through LDS, waves tile M x N.
//     gemm_skinny_kernel small M (levels 2..mid, gates): one 32-column weight tile per workgroup, the
//                        waves split K inside the workgroup (A fragments straight to registers, partial
//                        tiles summed through LDS in wave order: bitwise deterministic, no inter-workgroup
//                        protocol), so a 64 x 2048 x 4096 GEMM still spreads over the chip.
// * LayerNorm statistics are never recomputed by the consumer: every producer of a residual-stream
//   tensor emits per-row (mean, M2) partials per 32-column tile (stats_out), and the LN loader merges
//   them with Chan's parallel-variance update.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hd {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

constexpr int BK = 64;                 // K chunk staged per barrier
constexpr int LDS_ROW = BK * 2 + 16;   // bytes per staged A row (128 B data + 16 B pad)

struct GemmP {
    // problem
    int M, N, K, Kp;              // rows, real output columns, real K, K padded to a multiple of 64
    int nt_total;                 // 32-column tiles present in the packed weight
    const uint4* W;               // packed bf16 weight
    // LayerNorm statistics hand-off (mean, M2) per row per partial
    const float2* stats_in;       // consumer: [M][stats_np] partials of the rows of A
    int stats_np, stats_cnt;      // partials per row, elements per partial
    float2* stats_out;            // producer: [M][N/32] partials of the rows of out (or NULL)
    // loader
    const void* A;
    int lda;
    float a_scale;
    int hw;                       // rows per face
    const float* film;            // FiLM table (gain/bias rows)
    int film_face_stride, film_step_stride, film_gain_off, film_bias_off;
    const int* step_ptr;          // device step index (NULL -> 0)
    float ln_eps;
    const float* rowscale;        // [faces][K] SCA scale
    int Hin, Win, Cin, KH, KW, stride, pad, Hout, Wout, ntaps;
    const float* gate_c;          // [faces][Cin]
    const float* gate_s;          // [faces*Hin*Win]
    const float* add_src;         // same layout as A (fp32), or NULL
    // epilogue
    void* out;
    int ldo;
    const float* bias;
    const float* rscale;
    const void* resid;
    int ldr;
    int act;                      // 0 none, 1 relu, 2 sigmoid
    int shuffle_r;                // pixel-shuffle factor (1 or 2)
};

__device__ __forceinline__ unsigned pack2(float lo, float hi) {
    unsigned short a = __builtin_bit_cast(unsigned short, (__bf16)lo);
    unsigned short b = __builtin_bit_cast(unsigned short, (__bf16)hi);
    return (unsigned)a | ((unsigned)b << 16);
}
__device__ __forceinline__ uint4 pack8(const float* v) {
    return make_uint4(pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7]));
}
__device__ __forceinline__ void unpack8(uint4 u, float* v) {
    v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
    v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
    v[4] = __uint_as_float(u.z << 16); v[5] = __uint_as_float(u.z & 0xffff0000u);
    v[6] = __uint_as_float(u.w << 16); v[7] = __uint_as_float(u.w & 0xffff0000u);
}
__device__ __forceinline__ float bf16_bits_to_f32(unsigned short v) { return __uint_as_float((unsigned)v << 16); }
__device__ __forceinline__ unsigned short f32_to_bf16_bits(float f) {
    return __builtin_bit_cast(unsigned short, (__bf16)f);
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <int WM_, int WN_, int MT_, int TN_, bool PAIR_>
struct Cfg {
    static constexpr int WM = WM_, WN = WN_, MT = MT_, TN = TN_;
    static constexpr bool PAIR = PAIR_;
    static constexpr int WAVES = WM * WN;
    static constexpr int THREADS = 64 * WAVES;
    static constexpr int BM = WM * MT * 32;
    static constexpr int TNT = PAIR ? 2 * TN : TN;
    static constexpr int NCOLS = WN * TN * 32;          // (gate) columns per workgroup
    static constexpr int UNITS = BM * 8 / THREADS;      // (row, 8 k) staging units per thread
    static constexpr int A_BUF = BM * LDS_ROW;
    static constexpr int STATS_OFF = 2 * A_BUF;
    static constexpr int GB_OFF = STATS_OFF + BM * 8;      // + 2*Kp floats of FiLM gain/bias for the LN loader
    static constexpr int SMEM = GB_OFF;
    static_assert(UNITS >= 1 && UNITS * THREADS == BM * 8, "tile/threads mismatch");
};

// ------------------------------------------------------------------------------------------ loaders
struct F8 { float4 a, b; };
__device__ __forceinline__ F8 ldg8(const float* p) {
    F8 r; r.a = *reinterpret_cast<const float4*>(p); r.b = *reinterpret_cast<const float4*>(p + 4); return r;
}
__device__ __forceinline__ F8 zero8() { F8 r; r.a = make_float4(0, 0, 0, 0); r.b = r.a; return r; }
__device__ __forceinline__ void f8_to_arr(const F8& x, float* v) {
    v[0] = x.a.x; v[1] = x.a.y; v[2] = x.a.z; v[3] = x.a.w; v[4] = x.b.x; v[5] = x.b.y; v[6] = x.b.z; v[7] = x.b.w;
}

// fp32 rows, optional scalar scale (SCA pooled input, up-conv input, gate MLPs)
struct LdF32Plain {
    struct St { const float* rowp; bool valid; };
    struct Raw { F8 x; };
    static constexpr int kRawRegs = 8;
    static constexpr bool kGainBiasLds = false;
    template <int BM, int THREADS> static __device__ void block_init(const GemmP&, int, char*, float*, int) {}
    static __device__ __forceinline__ void unit_init(const GemmP& p, St& st, int row, int, const char*, const float*) {
        st.valid = row < p.M;
        st.rowp = reinterpret_cast<const float*>(p.A) + (size_t)(st.valid ? row : 0) * p.lda;
    }
    static __device__ __forceinline__ void fetch(const GemmP& p, const St& st, int k0, Raw& r) {
        r.x = (st.valid && k0 < p.K) ? ldg8(st.rowp + k0) : zero8();
    }
    static __device__ __forceinline__ uint4 finish(const GemmP& p, const St&, int, const Raw& r) {
        float v[8]; f8_to_arr(r.x, v);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] *= p.a_scale;
        return pack8(v);
    }
};

// fp32 rows -> LayerNorm2d over the row (two-pass, biased variance, eps inside the sqrt: utils.py:18-22)
// -> folded LN-affine/FiLM gain and bias (conditional_naf.py:114-115,126-127)
struct LdF32LN {
    static constexpr int kRawRegs = 8;
    static constexpr bool kGainBiasLds = true;
    struct St { const float* rowp; const float* gain; const float* bias; const float* gbl; float mu, rstd; bool valid; };
    struct Raw { F8 x; };
    // (1) merge the producer's per-tile (mean, M2) partials of each row into (mean, rstd) in LDS: 4 threads
    //     per row, Chan's update (Chan, Golub, LeVeque 1979), fixed order -> deterministic;
    // (2) when every row shares one FiLM row (sampling: same t for all faces) copy gain/bias[K] to LDS so
    //     finish() never waits on global memory.
    template <int BM, int THREADS> static __device__ void block_init(const GemmP& p, int row0, char* stats, float* gb, int tid) {
        float2* st = reinterpret_cast<float2*>(stats);
        if (gb && p.film_face_stride == 0) {
            const int step = p.step_ptr ? *p.step_ptr : 0;
            const float* f = p.film + (size_t)step * p.film_step_stride;
            for (int k = tid * 4; k < p.K; k += THREADS * 4) {
                *reinterpret_cast<float4*>(gb + k) = *reinterpret_cast<const float4*>(f + p.film_gain_off + k);
                *reinterpret_cast<float4*>(gb + p.Kp + k) = *reinterpret_cast<const float4*>(f + p.film_bias_off + k);
            }
        }
        for (int base = 0; base < BM; base += THREADS / 4) {
            const int rl = base + (tid >> 2), part = tid & 3;
            const int row = row0 + rl;
            float n = 0.f, mean = 0.f, m2 = 0.f;
            if (rl < BM && row < p.M) {
                const float2* sp = p.stats_in + (size_t)row * p.stats_np;
                const float cnt = (float)p.stats_cnt;
                for (int i = part; i < p.stats_np; i += 4) {
                    const float2 v = sp[i];
                    const float d = v.x - mean, nn = n + cnt;
                    mean += d * (cnt / nn);
                    m2 += v.y + d * d * (n * cnt / nn);
                    n = nn;
                }
            }
#pragma unroll
            for (int o = 1; o < 4; o <<= 1) {
                const float n2 = __shfl_xor(n, o, 64), mean2 = __shfl_xor(mean, o, 64), m22 = __shfl_xor(m2, o, 64);
                const float nn = n + n2;
                if (nn > 0.f) {
                    const float d = mean2 - mean;
                    m2 = m2 + m22 + d * d * (n * n2 / nn);
                    mean = mean + d * (n2 / nn);
                    n = nn;
                }
            }
            if (part == 0 && rl < BM) st[rl] = make_float2(mean, n > 0.f ? 1.0f / sqrtf(m2 / n + p.ln_eps) : 0.f);
        }
        __syncthreads();
    }
    static __device__ __forceinline__ void unit_init(const GemmP& p, St& st, int row, int row_local, const char* stats, const float* gb) {
        st.valid = row < p.M;
        const int r = st.valid ? row : 0;
        st.rowp = reinterpret_cast<const float*>(p.A) + (size_t)r * p.lda;
        const float2 s = reinterpret_cast<const float2*>(stats)[row_local];
        st.mu = s.x; st.rstd = s.y;
        st.gbl = (gb && p.film_face_stride == 0) ? gb : nullptr;      // LDS copy of the shared FiLM row
        const int step = p.step_ptr ? *p.step_ptr : 0;
        const float* f = p.film + (size_t)step * p.film_step_stride + (size_t)(r / p.hw) * p.film_face_stride;
        st.gain = f + p.film_gain_off;
        st.bias = f + p.film_bias_off;
    }
    static __device__ __forceinline__ void fetch(const GemmP& p, const St& st, int k0, Raw& r) {
        r.x = (st.valid && k0 < p.K) ? ldg8(st.rowp + k0) : zero8();
    }
    static __device__ __forceinline__ uint4 finish(const GemmP& p, const St& st, int k0, const Raw& r) {
        float v[8], g[8], b[8];
        if (!(st.valid && k0 < p.K)) return make_uint4(0, 0, 0, 0);
        f8_to_arr(r.x, v);
        if (st.gbl) {
            const float4* gp = reinterpret_cast<const float4*>(st.gbl + k0);
            const float4* bp = reinterpret_cast<const float4*>(st.gbl + p.Kp + k0);
            F8 gg, bb; gg.a = gp[0]; gg.b = gp[1]; bb.a = bp[0]; bb.b = bp[1];
            f8_to_arr(gg, g); f8_to_arr(bb, b);
        } else {
            f8_to_arr(ldg8(st.gain + k0), g); f8_to_arr(ldg8(st.bias + k0), b);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (v[i] - st.mu) * st.rstd * g[i] + b[i];
        return pack8(v);
    }
};

// bf16 rows, copied as they are (conv5 input G2, ResNet 1x1 convs)
struct LdBF16Plain {
    struct St { const unsigned short* rowp; bool valid; };
    struct Raw { uint4 x; };
    static constexpr int kRawRegs = 4;
    static constexpr bool kGainBiasLds = false;
    template <int BM, int THREADS> static __device__ void block_init(const GemmP&, int, char*, float*, int) {}
    static __device__ __forceinline__ void unit_init(const GemmP& p, St& st, int row, int, const char*, const float*) {
        st.valid = row < p.M;
        st.rowp = reinterpret_cast<const unsigned short*>(p.A) + (size_t)(st.valid ? row : 0) * p.lda;
    }
    static __device__ __forceinline__ void fetch(const GemmP& p, const St& st, int k0, Raw& r) {
        r.x = (st.valid && k0 < p.K) ? *reinterpret_cast<const uint4*>(st.rowp + k0) : make_uint4(0, 0, 0, 0);
    }
    static __device__ __forceinline__ uint4 finish(const GemmP&, const St&, int, const Raw& r) { return r.x; }
};

// bf16 rows times a per-(face, k) fp32 scale: x * sca(x) feeding conv3 (conditional_naf.py:119-120)
struct LdBF16Scale {
    struct St { const unsigned short* rowp; const float* srow; bool valid; };
    struct Raw { uint4 x; F8 s; };
    static constexpr int kRawRegs = 12;
    static constexpr bool kGainBiasLds = false;
    template <int BM, int THREADS> static __device__ void block_init(const GemmP&, int, char*, float*, int) {}
    static __device__ __forceinline__ void unit_init(const GemmP& p, St& st, int row, int, const char*, const float*) {
        st.valid = row < p.M;
        const int r = st.valid ? row : 0;
        st.rowp = reinterpret_cast<const unsigned short*>(p.A) + (size_t)r * p.lda;
        st.srow = p.rowscale + (size_t)(r / p.hw) * p.K;
    }
    static __device__ __forceinline__ void fetch(const GemmP& p, const St& st, int k0, Raw& r) {
        if (st.valid && k0 < p.K) { r.x = *reinterpret_cast<const uint4*>(st.rowp + k0); r.s = ldg8(st.srow + k0); }
        else { r.x = make_uint4(0, 0, 0, 0); r.s = zero8(); }
    }
    static __device__ __forceinline__ uint4 finish(const GemmP&, const St&, int, const Raw& r) {
        float v[8], s[8]; unpack8(r.x, v); f8_to_arr(r.s, s);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] *= s[i];
        return pack8(v);
    }
};

// im2col gather over a channels-last image: k = tap*Cin + c.  SRC_BF16: ResNet activations;
// fp32 + GATED: the HCA 3x3 conv input f_d*(1 + w_c + w_s) (+ idc term) (hca.py:28, model.py:245-246)
template <bool SRC_BF16, bool GATED>
struct LdConv {
    struct St { int b, iy0, ix0; bool valid; };
    struct Raw { F8 x; F8 add; F8 gc; float gs; uint4 xb; bool inb; };
    static constexpr int kRawRegs = (SRC_BF16 ? 4 : 8) + (GATED ? 17 : 0) + 1;
    static constexpr bool kGainBiasLds = false;
    template <int BM, int THREADS> static __device__ void block_init(const GemmP&, int, char*, float*, int) {}
    static __device__ __forceinline__ void unit_init(const GemmP& p, St& st, int row, int, const char*, const float*) {
        st.valid = row < p.M;
        const int r = st.valid ? row : 0;
        const int hwo = p.Hout * p.Wout;
        st.b = r / hwo;
        const int rem = r - st.b * hwo;
        const int oy = rem / p.Wout;
        st.iy0 = oy * p.stride - p.pad;
        st.ix0 = (rem - oy * p.Wout) * p.stride - p.pad;
    }
    static __device__ __forceinline__ void fetch(const GemmP& p, const St& st, int k0, Raw& r) {
        const int tap = k0 / p.Cin;
        const int c0 = k0 - tap * p.Cin;
        const int ky = tap / p.KW;
        const int iy = st.iy0 + ky, ix = st.ix0 + (tap - ky * p.KW);
        r.inb = st.valid && tap < p.ntaps && iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win;
        if (!r.inb) return;
        const size_t srow = ((size_t)st.b * p.Hin + iy) * p.Win + ix;
        if (SRC_BF16) {
            r.xb = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(p.A) + srow * p.Cin + c0);
        } else {
            r.x = ldg8(reinterpret_cast<const float*>(p.A) + srow * p.Cin + c0);
        }
        if (GATED) {
            r.gs = p.gate_s[srow];
            r.gc = ldg8(p.gate_c + (size_t)st.b * p.Cin + c0);
            r.add = p.add_src ? ldg8(p.add_src + srow * p.Cin + c0) : zero8();
        }
    }
    static __device__ __forceinline__ uint4 finish(const GemmP&, const St&, int, const Raw& r) {
        if (!r.inb) return make_uint4(0, 0, 0, 0);
        if (SRC_BF16 && !GATED) return r.xb;
        float v[8];
        if (SRC_BF16) unpack8(r.xb, v); else f8_to_arr(r.x, v);
        if (GATED) {
            float a[8], g[8]; f8_to_arr(r.add, a); f8_to_arr(r.gc, g);
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = (v[i] + a[i]) * (1.0f + g[i] + r.gs);
        }
        return pack8(v);
    }
};

// ---------------------------------------------------------------------------------------- epilogues
__device__ __forceinline__ float activate(float v, int act) {
    if (act == 1) return fmaxf(v, 0.f);
    if (act == 2) return 1.0f / (1.0f + expf(-v));
    return v;
}

// out(fp32)[row][col] = act(acc + bias)
struct EpBiasF32 {
    static __device__ __forceinline__ float store(const GemmP& p, int row, int col, float v) {
        if (p.bias) v += p.bias[col];
        v = activate(v, p.act);
        reinterpret_cast<float*>(p.out)[(size_t)row * p.ldo + col] = v;
        return v;
    }
};
// out(fp32) = resid + rscale[col] * (acc + bias)      (y = inp + x*beta, out = y + x*gamma)
struct EpResidF32 {
    static __device__ __forceinline__ float store(const GemmP& p, int row, int col, float v) {
        v += p.bias[col];
        const float r = reinterpret_cast<const float*>(p.resid)[(size_t)row * p.ldr + col];
        v = r + v * p.rscale[col];
        reinterpret_cast<float*>(p.out)[(size_t)row * p.ldo + col] = v;
        return v;
    }
};
// PAIR: out(bf16)[row][col] = (acc1 + bias[col]) * (acc2 + bias[col + N/2])   (conv4 -> SimpleGate)
struct EpGateBF16 {
    static __device__ __forceinline__ void store2(const GemmP& p, int row, int col, float v1, float v2) {
        v1 += p.bias[col]; v2 += p.bias[col + (p.N >> 1)];
        reinterpret_cast<unsigned short*>(p.out)[(size_t)row * p.ldo + col] = f32_to_bf16_bits(v1 * v2);
    }
};
// 1x1 conv (no bias) -> PixelShuffle(r) -> + skip : out[b, r*h+i, r*w+j, c] = acc[n = c*r*r + i*r + j]
struct EpPixShufF32 {
    static __device__ __forceinline__ float store(const GemmP& p, int row, int col, float v) {
        float* out = reinterpret_cast<float*>(p.out);
        size_t o;
        if (p.shuffle_r == 2) {
            const int hw = p.Hin * p.Win;
            const int b = row / hw, rem = row - b * hw;
            const int h = rem / p.Win, w = rem - h * p.Win;
            const int c = col >> 2, i = (col >> 1) & 1, j = col & 1;
            o = (((size_t)b * (2 * p.Hin) + (2 * h + i)) * (2 * p.Win) + (2 * w + j)) * p.ldo + c;
        } else {
            o = (size_t)row * p.ldo + col;
        }
        if (p.resid) v += reinterpret_cast<const float*>(p.resid)[o];
        out[o] = v;
        return v;
    }
};
// out(bf16) = act(acc + bias (+ resid bf16))           (ResNet conv+BN(+identity)+ReLU, BN folded)
struct EpBiasBF16 {
    static __device__ __forceinline__ float store(const GemmP& p, int row, int col, float v) {
        v += p.bias[col];
        if (p.resid) v += bf16_bits_to_f32(reinterpret_cast<const unsigned short*>(p.resid)[(size_t)row * p.ldr + col]);
        v = activate(v, p.act);
        reinterpret_cast<unsigned short*>(p.out)[(size_t)row * p.ldo + col] = f32_to_bf16_bits(v);
        return v;
    }
};

// ------------------------------------------------------------------------------ statistics emission
// (mean, M2) of 32 values spread over the 32 lanes of a half-wave (Chan partial for one 32-column tile)
__device__ __forceinline__ float2 halfwave_mean_m2(float v) {
    float s = v;
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mean = s * (1.0f / 32.0f);
    float d = v - mean;
    d *= d;
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
    return make_float2(mean, d);
}

// Does this epilogue's store() return the value it wrote (needed to emit LN statistics of `out`)?
template <class EP> struct EpTraits { static constexpr bool kStats = false; };
template <> struct EpTraits<EpBiasF32> { static constexpr bool kStats = true; };
template <> struct EpTraits<EpResidF32> { static constexpr bool kStats = true; };

// ------------------------------------------------------------------------------------- tall kernel
template <class C, class LD, class EP>
__global__ __launch_bounds__(C::THREADS) void gemm_kernel(const GemmP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int w_m = wave / C::WN, w_n = wave - w_m * C::WN;
    const int row0 = blockIdx.x * C::BM;
    const int ksteps_total = p.Kp >> 4;
    const int c_end = p.Kp >> 6;

    // weight tiles of this wave
    int tile[C::TNT];
#pragma unroll
    for (int tn = 0; tn < C::TN; ++tn) {
        const int t = (blockIdx.y * C::WN + w_n) * C::TN + tn;
        tile[tn] = t;
        if (C::PAIR) tile[C::TN + tn] = t + (p.N >> 6);    // second half starts at N/2 = 32*(N/64)
    }
    const int tiles_half = C::PAIR ? (p.N >> 6) : p.nt_total;

    float* gb = LD::kGainBiasLds ? reinterpret_cast<float*>(smem + C::GB_OFF) : nullptr;
    LD::template block_init<C::BM, C::THREADS>(p, row0, smem + C::STATS_OFF, gb, tid);

    typename LD::St st[C::UNITS];
    int u_ldsoff[C::UNITS], u_k[C::UNITS];
#pragma unroll
    for (int u = 0; u < C::UNITS; ++u) {
        const int unit = tid + u * C::THREADS;
        const int rl = unit >> 3, kq = unit & 7;
        u_ldsoff[u] = rl * LDS_ROW + kq * 16;
        u_k[u] = kq * 8;
        LD::unit_init(p, st[u], row0 + rl, rl, smem + C::STATS_OFF, gb);
    }

    f32x16_t acc[C::MT][C::TNT];
#pragma unroll
    for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
        for (int tn = 0; tn < C::TNT; ++tn)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][tn][i] = 0.f;

    typename LD::Raw raw[C::UNITS];
    uint4 bcur[C::TNT][4], bnxt[C::TNT][4];
    const uint4* Wl = p.W + lane;

#define HD_LOAD_B(dst, chunk)                                                                         \
    _Pragma("unroll") for (int tn = 0; tn < C::TNT; ++tn) {                                           \
        const bool tv = (tn < C::TN ? tile[tn] : tile[tn] - (p.N >> 6)) < tiles_half &&               \
                        tile[tn] < p.nt_total;                                                        \
        _Pragma("unroll") for (int s = 0; s < 4; ++s)                                                 \
            dst[tn][s] = tv ? Wl[((size_t)tile[tn] * ksteps_total + (chunk) * 4 + s) * 64]            \
                            : make_uint4(0, 0, 0, 0);                                                 \
    }
#define HD_FETCH_A(chunk)                                                                             \
    _Pragma("unroll") for (int u = 0; u < C::UNITS; ++u) LD::fetch(p, st[u], (chunk) * BK + u_k[u], raw[u]);
#define HD_WRITE_A(chunk, buf)                                                                        \
    _Pragma("unroll") for (int u = 0; u < C::UNITS; ++u)                                              \
        *reinterpret_cast<uint4*>(smem + (buf) * C::A_BUF + u_ldsoff[u]) =                            \
            LD::finish(p, st[u], (chunk) * BK + u_k[u], raw[u]);

    HD_FETCH_A(0);
    HD_LOAD_B(bcur, 0);
    HD_WRITE_A(0, 0);
    __syncthreads();

    const int a_lane_off = (lane & 31) * LDS_ROW + (lane >> 5) * 16;
    for (int c = 0; c < c_end; ++c) {
        const int buf = c & 1;
        const bool has_next = (c + 1) < c_end;
        if (has_next) {
            HD_FETCH_A(c + 1);
            HD_LOAD_B(bnxt, c + 1);
        }
        const char* sA = smem + buf * C::A_BUF + a_lane_off;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            bf16x8_t a[C::MT];
#pragma unroll
            for (int mt = 0; mt < C::MT; ++mt)
                a[mt] = *reinterpret_cast<const bf16x8_t*>(sA + ((w_m * C::MT + mt) * 32) * LDS_ROW + s * 32);
#pragma unroll
            for (int tn = 0; tn < C::TNT; ++tn)
#pragma unroll
                for (int mt = 0; mt < C::MT; ++mt)
                    acc[mt][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                        a[mt], __builtin_bit_cast(bf16x8_t, bcur[tn][s]), acc[mt][tn], 0, 0, 0);
        }
        if (has_next) {
            HD_WRITE_A(c + 1, buf ^ 1);
        }
        __syncthreads();
        if (has_next) {
#pragma unroll
            for (int tn = 0; tn < C::TNT; ++tn)
#pragma unroll
                for (int s = 0; s < 4; ++s) bcur[tn][s] = bnxt[tn][s];
        }
    }
#undef HD_LOAD_B
#undef HD_FETCH_A
#undef HD_WRITE_A

    // ---- epilogue: C/D map of mfma_f32_32x32x16: col = lane&31, row = (i&3) + 8*(i>>2) + 4*(lane>>5) ----
    const int ncols = C::PAIR ? (p.N >> 1) : p.N;
#pragma unroll
    for (int mt = 0; mt < C::MT; ++mt) {
        const int rbase = row0 + (w_m * C::MT + mt) * 32 + 4 * (lane >> 5);
#pragma unroll
        for (int tn = 0; tn < C::TN; ++tn) {
            const int col = tile[tn] * 32 + (lane & 31);
            if (tile[tn] * 32 >= ncols) continue;                      // wave-uniform
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = rbase + (i & 3) + 8 * (i >> 2);
                float v = 0.f;
                if (row < p.M && col < ncols) {
                    if constexpr (C::PAIR) EP::store2(p, row, col, acc[mt][tn][i], acc[mt][C::TN + tn][i]);
                    else v = EP::store(p, row, col, acc[mt][tn][i]);
                }
                if constexpr (EpTraits<EP>::kStats) {
                    if (p.stats_out) {                                 // rows of `out` feed a LayerNorm next
                        const float2 ms = halfwave_mean_m2(v);
                        if ((lane & 31) == 0 && row < p.M) p.stats_out[(size_t)row * (p.N >> 5) + tile[tn]] = ms;
                    }
                }
            }
        }
    }
}

// ----------------------------------------------------------------------------------- skinny kernel
// One 32-column weight tile (PAIR: tile j and tile j + N/64) x MT*32 rows per workgroup; the K dimension
// is split over the workgroup's WAVES waves (wave w owns k-steps [w*ksw, (w+1)*ksw)).  A fragments come
// straight from global memory in MFMA layout (lane (r,h) holds A[row r][k0 + 8h .. +8] = one loader
// unit), B fragments from the packed weight.  Each wave issues G k-steps of loads in one burst (two
// bursts in flight), so with ksw <= 2G the whole K-slice of a wave is requested up front: the kernel is
// one memory round trip deep instead of K/32.  Partial tiles are summed through LDS in wave order.
template <int MT, int WAVES, bool PAIR, int G_>
struct SkinnyCfg {
    static constexpr int THREADS = 64 * WAVES, BM = MT * 32, TNT = PAIR ? 2 : 1, G = G_;
    static constexpr int RED = WAVES * BM * 32 * TNT * 4;        // cross-wave reduction buffer (bytes)
    static constexpr int STATS_OFF = RED;
    static constexpr int GB_OFF = STATS_OFF + BM * 8;            // + 2*Kp floats of gain/bias for the LN loader
};

template <class C, class LD, class EP>
__global__ __launch_bounds__(C::THREADS) void gemm_skinny_kernel(const GemmP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int MT = C::BM / 32, TNT = C::TNT, WAVES = C::THREADS / 64, G = C::G;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row0 = blockIdx.x * C::BM;
    const int ksteps_total = p.Kp >> 4;
    const int ksw = ksteps_total / WAVES;                    // host: Kp % (16*WAVES) == 0
    const int ks0 = wave * ksw, ks_end = ks0 + ksw;
    int tile[TNT];
    tile[0] = blockIdx.y;
    if (TNT == 2) tile[1] = blockIdx.y + (p.N >> 6);
    const uint4* Wl = p.W + lane;

    f32x16_t acc[MT][TNT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int tn = 0; tn < TNT; ++tn)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][tn][i] = 0.f;

    // weights of the first burst go out before anything that waits (statistics merge, row setup)
    uint4 bq[2][G][TNT];
#define HD_SK_LOAD_B(slot, ks)                                                                         \
    _Pragma("unroll") for (int g = 0; g < G; ++g) _Pragma("unroll") for (int tn = 0; tn < TNT; ++tn)   \
        bq[slot][g][tn] = ((ks) + g < ks_end) ? Wl[((size_t)tile[tn] * ksteps_total + (ks) + g) * 64] : make_uint4(0, 0, 0, 0);
    HD_SK_LOAD_B(0, ks0);

    float* gb = LD::kGainBiasLds ? reinterpret_cast<float*>(smem + C::GB_OFF) : nullptr;
    LD::template block_init<C::BM, C::THREADS>(p, row0, smem + C::STATS_OFF, gb, tid);
    typename LD::St st[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
        LD::unit_init(p, st[mt], row0 + mt * 32 + (lane & 31), mt * 32 + (lane & 31), smem + C::STATS_OFF, gb);
    const int k_lane = 8 * (lane >> 5);

    typename LD::Raw aq[2][G][MT];
#define HD_SK_K(ks, g) (((ks) + (g) < ks_end) ? ((ks) + (g)) * 16 + k_lane : p.Kp)
#define HD_SK_FETCH_A(slot, ks)                                                                        \
    _Pragma("unroll") for (int g = 0; g < G; ++g) _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)    \
        LD::fetch(p, st[mt], HD_SK_K(ks, g), aq[slot][g][mt]);
#define HD_SK_COMPUTE(slot, ks)                                                                        \
    _Pragma("unroll") for (int g = 0; g < G; ++g) {                                                    \
        bf16x8_t a[MT];                                                                                \
        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                              \
            a[mt] = __builtin_bit_cast(bf16x8_t, LD::finish(p, st[mt], HD_SK_K(ks, g), aq[slot][g][mt])); \
        _Pragma("unroll") for (int tn = 0; tn < TNT; ++tn) _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) \
            acc[mt][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt], __builtin_bit_cast(bf16x8_t, bq[slot][g][tn]), acc[mt][tn], 0, 0, 0); \
    }
    HD_SK_FETCH_A(0, ks0);
    for (int ks = ks0; ks < ks_end; ks += 2 * G) {            // two bursts per trip: static register slots
        if (ks + G < ks_end) { HD_SK_LOAD_B(1, ks + G); HD_SK_FETCH_A(1, ks + G); }
        HD_SK_COMPUTE(0, ks);
        if (ks + G < ks_end) {
            if (ks + 2 * G < ks_end) { HD_SK_LOAD_B(0, ks + 2 * G); HD_SK_FETCH_A(0, ks + 2 * G); }
            HD_SK_COMPUTE(1, ks + G);
        }
    }
#undef HD_SK_LOAD_B
#undef HD_SK_FETCH_A
#undef HD_SK_COMPUTE
#undef HD_SK_K

    // ---- cross-wave reduction through LDS, in wave order; then a row-major epilogue ----
    float* red = reinterpret_cast<float*>(smem);
    constexpr int TILE_F = C::BM * 32 * TNT;                  // floats per wave partial, layout [tn][row][col]
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int tn = 0; tn < TNT; ++tn)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int r = mt * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
                red[wave * TILE_F + (tn * C::BM + r) * 32 + (lane & 31)] = acc[mt][tn][i];
            }
    __syncthreads();
    const int ncols = (TNT == 2) ? (p.N >> 1) : p.N;
    const int col = tile[0] * 32 + (tid & 31);
    for (int e0 = 0; e0 < C::BM * 32; e0 += C::THREADS) {      // e -> (row_local, col_local): 32 lanes = one row
        const int e = e0 + tid;
        const bool ev = e < C::BM * 32;
        const int row = row0 + (e >> 5);
        float v1 = 0.f, v2 = 0.f;
        if (ev) {
#pragma unroll
            for (int w = 0; w < WAVES; ++w) {
                v1 += red[w * TILE_F + e];
                if (TNT == 2) v2 += red[w * TILE_F + C::BM * 32 + e];
            }
        }
        float v = 0.f;
        if (ev && row < p.M && col < ncols) {
            if constexpr (TNT == 2) EP::store2(p, row, col, v1, v2);
            else v = EP::store(p, row, col, v1);
        }
        if constexpr (EpTraits<EP>::kStats) {
            if (p.stats_out) {
                const float2 ms = halfwave_mean_m2(v);
                if (ev && (tid & 31) == 0 && row < p.M) p.stats_out[(size_t)row * (p.N >> 5) + tile[0]] = ms;
            }
        }
    }
}

// Tile shapes.  T128: tall GEMMs (levels 0/1, ResNet) — 4 waves stacked along M, each 32 rows x 64 cols.
//               T64 : 2x2 waves, 64 rows x 64 cols.
typedef Cfg<4, 1, 1, 2, false> T128;
typedef Cfg<4, 1, 1, 1, true> T128P;
typedef Cfg<2, 2, 1, 1, false> T64;
typedef Cfg<2, 2, 1, 1, true> T64P;
//               T32W: 32 rows x (4 waves x 64 cols): the whole N of a level-0/1 GEMM in one workgroup, so the
//               activation tile is read once and >= 512 small workgroups keep the CUs occupied.
typedef Cfg<1, 4, 1, 2, false> T32W;
typedef Cfg<1, 4, 1, 1, true> T32WP;

template <class C, class LD, class EP>
inline hipError_t launch_gemm(const GemmP& p, hipStream_t s) {
    const int ncols = C::PAIR ? p.N / 2 : p.N;
    const int smem = C::SMEM + (LD::kGainBiasLds ? 2 * p.Kp * 4 : 0);
    if (smem > 65536) {                                      // above the default dynamic-LDS limit
        static int granted = 0;
        if (smem > granted) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<C, LD, EP>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
            granted = 160 * 1024;
        }
    }
    dim3 grid((p.M + C::BM - 1) / C::BM, (ncols + C::NCOLS - 1) / C::NCOLS, 1);
    hipLaunchKernelGGL((gemm_kernel<C, LD, EP>), grid, dim3(C::THREADS), smem, s, p);
    return hipGetLastError();
}

template <class C, class LD, class EP>
inline hipError_t launch_skinny(const GemmP& p, hipStream_t s) {
    const int smem = C::GB_OFF + (LD::kGainBiasLds ? 2 * p.Kp * 4 : 0);
    if (smem > 160 * 1024) return hipErrorInvalidValue;
    if (smem > 65536) {
        static int granted = 0;
        if (smem > granted) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_skinny_kernel<C, LD, EP>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
            granted = 160 * 1024;
        }
    }
    const int ncols = (C::TNT == 2) ? p.N / 2 : p.N;
    dim3 grid((p.M + C::BM - 1) / C::BM, (ncols + 31) / 32, 1);
    hipLaunchKernelGGL((gemm_skinny_kernel<C, LD, EP>), grid, dim3(C::THREADS), smem, s, p);
    return hipGetLastError();
}

// Burst depth from a register budget: accumulators + two slots of G k-steps of (B fragments + raw A units)
// must stay well inside the 256 VGPRs a wave gets at <= 8 waves per workgroup.
template <int MT, bool PAIR, class LD>
struct BurstDepth {
    static constexpr int TNT = PAIR ? 2 : 1;
    static constexpr int fixed = MT * TNT * 16 + 40;
    static constexpr int per_g = 2 * (TNT * 4 + MT * LD::kRawRegs);
    static constexpr int G = (fixed + 4 * per_g <= 216) ? 4 : ((fixed + 2 * per_g <= 216) ? 2 : 1);
};

// Skinny launch with the K-split chosen at run time: as many waves (<= 8) as give each >= 4 k-steps.
template <int MT, bool PAIR, class LD, class EP>
inline hipError_t launch_skinny_auto(const GemmP& p, hipStream_t s) {
    constexpr int G = BurstDepth<MT, PAIR, LD>::G;
    const int ksteps = p.Kp / 16;
    if (ksteps >= 64) return launch_skinny<SkinnyCfg<MT, 8, PAIR, G>, LD, EP>(p, s);
    if (ksteps >= 32) return launch_skinny<SkinnyCfg<MT, 4, PAIR, G>, LD, EP>(p, s);
    if (ksteps >= 16) return launch_skinny<SkinnyCfg<MT, 2, PAIR, G>, LD, EP>(p, s);
    return launch_skinny<SkinnyCfg<MT, 1, PAIR, G>, LD, EP>(p, s);
}

}  // namespace hd
