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
// * Split-K (gridDim.z > 1) for the skinny weight-streaming GEMMs: every slice writes its fp32 partial
//   tile in accumulator order; the last arriver (agent-scope release/acquire ticket) sums all slices in
//   slice order — bitwise deterministic — and runs the epilogue.
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
    int ksplit;                   // gridDim.z
    float* slab;                  // split-K partials
    unsigned* counters;           // split-K tickets (self-resetting)
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
    static constexpr int FLAG_OFF = STATS_OFF + BM * 8;
    static constexpr int SMEM = FLAG_OFF + 16;
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
    template <class C> static __device__ void block_init(const GemmP&, int, char*, int) {}
    static __device__ __forceinline__ void unit_init(const GemmP& p, St& st, int row, int, const char*) {
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
    struct St { const float* rowp; const float* gain; const float* bias; float mu, rstd; bool valid; };
    struct Raw { F8 x; };
    template <class C> static __device__ void block_init(const GemmP& p, int row0, char* stats, int tid) {
        const int lane = tid & 63, wave = tid >> 6;
        float2* st = reinterpret_cast<float2*>(stats);
        for (int rl = wave; rl < C::BM; rl += C::WAVES) {
            const int row = row0 + rl;
            float mu = 0.f, rstd = 0.f;
            if (row < p.M) {                                  // wave-uniform
                const float* rp = reinterpret_cast<const float*>(p.A) + (size_t)row * p.lda;
                float s = 0.f;
                for (int k = lane * 4; k < p.K; k += 256) {
                    float4 v = *reinterpret_cast<const float4*>(rp + k);
                    s += (v.x + v.y) + (v.z + v.w);
                }
                mu = wave_sum(s) / (float)p.K;
                float q = 0.f;
                for (int k = lane * 4; k < p.K; k += 256) {
                    float4 v = *reinterpret_cast<const float4*>(rp + k);
                    float a = v.x - mu, b = v.y - mu, c = v.z - mu, d = v.w - mu;
                    q += (a * a + b * b) + (c * c + d * d);
                }
                rstd = 1.0f / sqrtf(wave_sum(q) / (float)p.K + p.ln_eps);
            }
            if (lane == 0) st[rl] = make_float2(mu, rstd);
        }
        __syncthreads();
    }
    static __device__ __forceinline__ void unit_init(const GemmP& p, St& st, int row, int row_local, const char* stats) {
        st.valid = row < p.M;
        const int r = st.valid ? row : 0;
        st.rowp = reinterpret_cast<const float*>(p.A) + (size_t)r * p.lda;
        const float2 s = reinterpret_cast<const float2*>(stats)[row_local];
        st.mu = s.x; st.rstd = s.y;
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
        f8_to_arr(r.x, v); f8_to_arr(ldg8(st.gain + k0), g); f8_to_arr(ldg8(st.bias + k0), b);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (v[i] - st.mu) * st.rstd * g[i] + b[i];
        return pack8(v);
    }
};

// bf16 rows, copied as they are (conv5 input G2, ResNet 1x1 convs)
struct LdBF16Plain {
    struct St { const unsigned short* rowp; bool valid; };
    struct Raw { uint4 x; };
    template <class C> static __device__ void block_init(const GemmP&, int, char*, int) {}
    static __device__ __forceinline__ void unit_init(const GemmP& p, St& st, int row, int, const char*) {
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
    template <class C> static __device__ void block_init(const GemmP&, int, char*, int) {}
    static __device__ __forceinline__ void unit_init(const GemmP& p, St& st, int row, int, const char*) {
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
    template <class C> static __device__ void block_init(const GemmP&, int, char*, int) {}
    static __device__ __forceinline__ void unit_init(const GemmP& p, St& st, int row, int, const char*) {
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
    static __device__ __forceinline__ void store(const GemmP& p, int row, int col, float v) {
        if (p.bias) v += p.bias[col];
        reinterpret_cast<float*>(p.out)[(size_t)row * p.ldo + col] = activate(v, p.act);
    }
};
// out(fp32) = resid + rscale[col] * (acc + bias)      (y = inp + x*beta, out = y + x*gamma)
struct EpResidF32 {
    static __device__ __forceinline__ void store(const GemmP& p, int row, int col, float v) {
        v += p.bias[col];
        const float r = reinterpret_cast<const float*>(p.resid)[(size_t)row * p.ldr + col];
        reinterpret_cast<float*>(p.out)[(size_t)row * p.ldo + col] = r + v * p.rscale[col];
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
    static __device__ __forceinline__ void store(const GemmP& p, int row, int col, float v) {
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
    }
};
// out(bf16) = act(acc + bias (+ resid bf16))           (ResNet conv+BN(+identity)+ReLU, BN folded)
struct EpBiasBF16 {
    static __device__ __forceinline__ void store(const GemmP& p, int row, int col, float v) {
        v += p.bias[col];
        if (p.resid) v += bf16_bits_to_f32(reinterpret_cast<const unsigned short*>(p.resid)[(size_t)row * p.ldr + col]);
        reinterpret_cast<unsigned short*>(p.out)[(size_t)row * p.ldo + col] = f32_to_bf16_bits(activate(v, p.act));
    }
};

// ------------------------------------------------------------------------------------------- kernel
template <class C, class LD, class EP>
__global__ __launch_bounds__(C::THREADS) void gemm_kernel(const GemmP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int w_m = wave / C::WN, w_n = wave - w_m * C::WN;
    const int row0 = blockIdx.x * C::BM;
    const int ksteps_total = p.Kp >> 4;
    const int chunks_total = p.Kp >> 6;
    const int cps = chunks_total / p.ksplit;               // host guarantees divisibility
    const int c_begin = blockIdx.z * cps, c_end = c_begin + cps;

    // weight tiles of this wave
    int tile[C::TNT];
#pragma unroll
    for (int tn = 0; tn < C::TN; ++tn) {
        const int t = (blockIdx.y * C::WN + w_n) * C::TN + tn;
        tile[tn] = t;
        if (C::PAIR) tile[C::TN + tn] = t + (p.N >> 6);    // second half starts at N/2 = 32*(N/64)
    }
    const int tiles_half = C::PAIR ? (p.N >> 6) : p.nt_total;

    LD::template block_init<C>(p, row0, smem + C::STATS_OFF, tid);

    typename LD::St st[C::UNITS];
    int u_ldsoff[C::UNITS], u_k[C::UNITS];
#pragma unroll
    for (int u = 0; u < C::UNITS; ++u) {
        const int unit = tid + u * C::THREADS;
        const int rl = unit >> 3, kq = unit & 7;
        u_ldsoff[u] = rl * LDS_ROW + kq * 16;
        u_k[u] = kq * 8;
        LD::unit_init(p, st[u], row0 + rl, rl, smem + C::STATS_OFF);
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

    if (c_begin < c_end) {
        HD_FETCH_A(c_begin);
        HD_LOAD_B(bcur, c_begin);
        HD_WRITE_A(c_begin, 0);
    }
    __syncthreads();

    const int a_lane_off = (lane & 31) * LDS_ROW + (lane >> 5) * 16;
    for (int c = c_begin; c < c_end; ++c) {
        const int buf = (c - c_begin) & 1;
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

    // ---- split-K: publish partial, last arriver reduces in slice order (deterministic) ----
    if (p.ksplit > 1) {
        constexpr int NACC4 = C::MT * C::TNT * 4;                      // float4 per lane
        const int tile_id = blockIdx.y * gridDim.x + blockIdx.x;
        float4* slab = reinterpret_cast<float4*>(p.slab);
        const size_t wg_stride = (size_t)C::WAVES * NACC4 * 64;        // float4 per (tile, slice)
        float4* mine = slab + ((size_t)tile_id * p.ksplit + blockIdx.z) * wg_stride + (size_t)wave * NACC4 * 64 + lane;
#pragma unroll
        for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
            for (int tn = 0; tn < C::TNT; ++tn)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    mine[((mt * C::TNT + tn) * 4 + q) * 64] =
                        make_float4(acc[mt][tn][4 * q], acc[mt][tn][4 * q + 1], acc[mt][tn][4 * q + 2], acc[mt][tn][4 * q + 3]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        unsigned* flag = reinterpret_cast<unsigned*>(smem + C::FLAG_OFF);
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned old = __hip_atomic_fetch_add(p.counters + tile_id, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *flag = (old == (unsigned)(p.ksplit - 1)) ? 1u : 0u;
        }
        __syncthreads();
        if (*flag == 0u) return;
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(p.counters + tile_id, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
            for (int tn = 0; tn < C::TNT; ++tn)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[mt][tn][i] = 0.f;
        for (int ks = 0; ks < p.ksplit; ++ks) {
            const float4* src = slab + ((size_t)tile_id * p.ksplit + ks) * wg_stride + (size_t)wave * NACC4 * 64 + lane;
#pragma unroll
            for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
                for (int tn = 0; tn < C::TNT; ++tn)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 v = src[((mt * C::TNT + tn) * 4 + q) * 64];
                        acc[mt][tn][4 * q] += v.x; acc[mt][tn][4 * q + 1] += v.y;
                        acc[mt][tn][4 * q + 2] += v.z; acc[mt][tn][4 * q + 3] += v.w;
                    }
        }
    }

    // ---- epilogue: C/D map of mfma_f32_32x32x16: col = lane&31, row = (i&3) + 8*(i>>2) + 4*(lane>>5) ----
    const int ncols = C::PAIR ? (p.N >> 1) : p.N;
#pragma unroll
    for (int mt = 0; mt < C::MT; ++mt) {
        const int rbase = row0 + (w_m * C::MT + mt) * 32 + 4 * (lane >> 5);
#pragma unroll
        for (int tn = 0; tn < C::TN; ++tn) {
            const int col = tile[tn] * 32 + (lane & 31);
            if (col >= ncols) continue;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = rbase + (i & 3) + 8 * (i >> 2);
                if (row >= p.M) continue;
                if constexpr (C::PAIR) EP::store2(p, row, col, acc[mt][tn][i], acc[mt][C::TN + tn][i]);
                else EP::store(p, row, col, acc[mt][tn][i]);
            }
        }
    }
}

// Tile shapes.  T128: tall GEMMs (levels 0/1) — 4 waves stacked along M, each 32 rows x 64 cols.
//               T64 : skinny GEMMs (levels 2..mid, prologue) — 2x2 waves, 64 rows x 64 cols, used with split-K.
typedef Cfg<4, 1, 1, 2, false> T128;
typedef Cfg<4, 1, 1, 1, true> T128P;
typedef Cfg<2, 2, 1, 1, false> T64;
typedef Cfg<2, 2, 1, 1, true> T64P;

template <class C, class LD, class EP>
inline hipError_t launch_gemm(const GemmP& p, hipStream_t s) {
    const int ncols = C::PAIR ? p.N / 2 : p.N;
    dim3 grid((p.M + C::BM - 1) / C::BM, (ncols + C::NCOLS - 1) / C::NCOLS, p.ksplit);
    hipLaunchKernelGGL((gemm_kernel<C, LD, EP>), grid, dim3(C::THREADS), C::SMEM, s, p);
    return hipGetLastError();
}

}  // namespace hd
