// hd_gemm.hpp — the MFMA GEMM / implicit-conv kernel family of the refiner path (gfx950 only).
//
//   Out[M,N] = epilogue( loader(A)[M,K] * W[K,N] )        bf16 operands, fp32 accumulate
//
// * rows  = pixels (channels-last activations, row = (face, y, x)), cols = output channels.
// * W is pre-packed once (hd_kernels.hpp: pack_weight_kernel) in MFMA B-fragment order
//   [N/32][K/16][64 lanes][8 bf16], so a wave streams its weight tiles with fully coalesced 1 KiB loads
//   straight into registers — no LDS round trip for the operand that is read exactly once.
// * A is loaded in whole cache lines (8 lanes per row; fp32 sources as two 128-byte halves per row so
//   that every wave-load covers complete lines), transformed by the loader, converted to bf16 and staged
//   through LDS in rows padded to 144 B (conflict-free ds_read_b128 fragment reads).
// * The loader fuses what precedes the conv in the reference: LayerNorm2d + FiLM (utils.py:16-24,
//   conditional_naf.py:114-115), the SCA channel scale (conditional_naf.py:119), the HCA gate
//   (hca.py:28) and the im2col gather of the 2x2/3x3/7x7 convs.
// * The epilogue fuses what follows: bias, SimpleGate (utils.py:57-60; a wave computes column tile j and
//   tile j+N/2 so the product is register-local), beta/gamma residual (conditional_naf.py:123,134),
//   PixelShuffle + skip add (models/denoiser/model.py:204-208,256-257), BN(eval)+ReLU, and for conv1 the
//   whole depthwise 3x3 + SimpleGate + SCA average pool (conditional_naf.py:116-119) on the LDS tile.
// * Two kernels share the loaders/epilogues:
//     gemm_kernel        tall M (levels 0/1 at full batch, ResNet): workgroup-shared LDS tile, waves tile M x N.
//     gemm_skinny_kernel small M: one 32-column weight tile per workgroup; its waves split M (WM) and K (WK);
//                        every wave stages its own A sub-tile in a private LDS region (no barrier in the
//                        K loop); partial tiles are summed through LDS in wave order (bitwise
//                        deterministic, no inter-workgroup protocol).
// * LayerNorm statistics are never recomputed by the consumer: every producer of a residual-stream
//   tensor emits per-row (mean, M2) partials per 32-column tile (stats_out), and the LN loader merges
//   them with Chan's parallel-variance update.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <stdlib.h>

#include <atomic>

// No implicit fused multiply-add in this file: whether `a * b + c` becomes one v_fma_f32 is otherwise decided per
// instantiation (the same epilogue gave different low bits in two tile shapes), and the XCD-local persistent stages
// (hd_xcd.hpp) must reproduce these kernels bit for bit.  Every fused operation below is written out (fmaf).
#pragma clang fp contract(off)

namespace hd {

// The dynamic-LDS limit of a kernel (hipFuncAttributeMaxDynamicSharedMemorySize) is a per-DEVICE attribute of the function:
// it is granted once per (kernel instantiation, device) -- `mask` is the instantiation's own function-local word, bit d =
// granted on device d -- so that a second context on another GPU of the same process gets it as well (thread-safe).
inline hipError_t grant_dynamic_lds(const void* fn, int bytes, std::atomic<unsigned long long>& mask) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (mask.load(std::memory_order_acquire) & bit) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return e;
    mask.fetch_or(bit, std::memory_order_release);
    return hipSuccess;
}


// Tile-shape and fusion rules are FIXED in the product.  The HD_* experiment switches that were used to measure them
//  are read only when HD_EXPERIMENTS=1 is set; otherwise they are inert (HD_CHAINS, independent sub-batches,
// among them).  The one product switch is HD_NO_XCD=1: levels 2 / 3 as one launch per GEMM instead of XCD-local stages.
inline const char* hd_env(const char* name) {
    return getenv("HD_EXPERIMENTS") != nullptr ? getenv(name) : nullptr;
}

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

constexpr int BK = 64;                 // K chunk
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
    int face0;                    // first face of this launch inside the batch (per-face FiLM rows)
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
    unsigned short* out16;        // optional bf16 copy of `out` (same ld): what the next LayerNorm GEMM reads
    unsigned short* outg16;       // optional bf16 (out + add_src) * (1 + gate_c[face] + gate_s[row]): the HCA conv input (hca.py:28)
    const float* bias;
    const float* rscale;
    const void* resid;
    int ldr;
    int act;                      // 0 none, 1 relu, 2 sigmoid
    int shuffle_r;                // pixel-shuffle factor (1 or 2)
    int xcd_tile_affine;          // skinny kernel: co-locate the row groups of a weight tile on one XCD
    // fused depthwise 3x3 + SimpleGate + average pool (EpDwGate)
    const float* dw_w;            // [N][9]
    const float* dw_b;            // [N]
    float* pooled;                // [faces][N/2]
    unsigned short* pooled16;     // bf16 copy of pooled (A operand of the SCA GEMM), or NULL
    int side;                     // face side (hw = side*side)
    // SCA epilogue (EpScaBF16): rows of this GEMM are faces; scale_G[(face*scale_hw + r)][col] *= s in place
    unsigned short* scale_G;
    int scale_hw;
    // next-kernel weight prefetch (skinny kernel): the packed tiles of the GEMM that runs after this one; tile t
    // is consumed on XCD t % 8 (xcd_tile_affine map), so workgroups with linear id % 8 == x touch the tiles
    // t % 8 == x and leave them in that XCD's L2.  NULL: nothing to prefetch.
    int w_nt;                     // skinny kernel: stream the weight fragments with the non-temporal cache policy
    const uint4* pf_base;
    unsigned pf_tile_u4;          // uint4 per tile
    int pf_ntiles;
#ifdef HD_STAMPS
    unsigned long long* stamps;   // diagnostic build (tools/gemm_bench): [workgroup][8] s_memrealtime ticks (100 MHz)
#endif
};

#ifdef HD_STAMPS
#define HD_STAMP(i)                                                                                     \
    do {                                                                                                \
        if (p.stamps && threadIdx.x == 0)                                                               \
            p.stamps[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define HD_STAMP(i) do { } while (0)
#endif

typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
// two fp32 -> packed bf16 (round to nearest even) in one v_cvt_pk_bf16_f32; two scalar conversions cost two of them plus a merge
__device__ __forceinline__ unsigned pack2(f32x2_t v) { return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t)); }
__device__ __forceinline__ unsigned pack2(float lo, float hi) { return pack2((f32x2_t){lo, hi}); }
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
// In-wave reductions on the DPP cross-lane path (a few cycles each).  __shfl_xor lowers to ds_bpermute, an
// LDS-latency operation: ten dependent ones per row made the statistics epilogue cost 11 us per kernel.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// every lane gets the sum over its 16-lane row: quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    v += dpp_mov<0x141>(v);
    v += dpp_mov<0x140>(v);
    return v;
}
__device__ __forceinline__ float lane_value(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
// row_bcast15: lane 15 of rows 0/2 is added into every lane of rows 1/3 -> after row16_sum the lanes
// 16..31 and 48..63 hold the sum over their 32-lane half (all 64 lanes must be active)
__device__ __forceinline__ float halfwave_sum_hi(float v) {
    v = row16_sum(v);
    const int t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xA, 0xF, false);
    return v + __builtin_bit_cast(float, t);
}
__device__ __forceinline__ float wave_sum(float v) {
    v = halfwave_sum_hi(v);                                   // lanes 16-31: half 0, lanes 48-63: half 1
    return lane_value(v, 31) + lane_value(v, 63);
}

// ------------------------------------------------------------------------------------------ loaders
// A "unit" is what one lane stages per 64-deep chunk for one row: 8 k-values.
//   kSplit == false (bf16 sources): k = kc + 8*kq + {0..7}            -> one 16-byte LDS write at 16*kq
//   kSplit == true  (fp32 sources): k = kc + 4*kq + {0..3} and kc + 32 + 4*kq + {0..3}: the 8 lanes of a
//       row read two complete 128-byte lines                           -> two 8-byte LDS writes (8*kq, 64 + 8*kq)
// fetch() issues the global loads, finish() transforms and packs (lo 4 | hi 4).  kc >= K yields zeros.
struct F8 { float4 a, b; };
__device__ __forceinline__ F8 ldg44(const float* p) {             // 4 floats at p, 4 floats at p + 32
    F8 r; r.a = *reinterpret_cast<const float4*>(p); r.b = *reinterpret_cast<const float4*>(p + 32); return r;
}
__device__ __forceinline__ F8 zero8() { F8 r; r.a = make_float4(0, 0, 0, 0); r.b = r.a; return r; }
__device__ __forceinline__ void f8_to_arr(const F8& x, float* v) {
    v[0] = x.a.x; v[1] = x.a.y; v[2] = x.a.z; v[3] = x.a.w; v[4] = x.b.x; v[5] = x.b.y; v[6] = x.b.z; v[7] = x.b.w;
}
template <bool SPLIT>
__device__ __forceinline__ void lds_write_unit(char* row_base, int kq, uint4 v) {
    if (SPLIT) {
        *reinterpret_cast<uint2*>(row_base + 8 * kq) = make_uint2(v.x, v.y);
        *reinterpret_cast<uint2*>(row_base + 64 + 8 * kq) = make_uint2(v.z, v.w);
    } else {
        *reinterpret_cast<uint4*>(row_base + 16 * kq) = v;
    }
}

// fp32 rows, optional scalar scale (SCA pooled input, up-conv input, gate MLPs)
struct LdF32Plain {
    static constexpr int kRawRegs = 8;
    static constexpr bool kGainBiasLds = false, kSplit = true, kStaticK = false;
    struct St { const float* rowp; bool valid; };
    struct Raw { F8 x; };
    struct Pre {};
    template <int BM, int THREADS> static __device__ __forceinline__ void block_issue(const GemmP&, int, float*, int, Pre&) {}
    template <int BM, int THREADS> static __device__ __forceinline__ void block_finish(const GemmP&, int, char*, float*, int, const Pre&) {}
    template <int BM, int THREADS> static __device__ void block_init(const GemmP&, int, char*, float*, int) {}
    template <class S> static __device__ __forceinline__ void unit_stats(S&, int, const char*) {}
    static __device__ __forceinline__ void unit_init(const GemmP& p, St& st, int row, int, const char*, const float*) {
        st.valid = row < p.M;
        st.rowp = reinterpret_cast<const float*>(p.A) + (size_t)(st.valid ? row : 0) * p.lda;
    }
    static __device__ __forceinline__ void fetch(const GemmP& p, const St& st, int kc, int kq, Raw& r) {
        // K is a multiple of 8, not necessarily of 64 (CoarseRestoration level 0: K = 32): each half is checked
        if (st.valid && kc + 32 + 4 * kq < p.K) r.x = ldg44(st.rowp + kc + 4 * kq);
        else { r.x = zero8(); if (st.valid && kc + 4 * kq < p.K) r.x.a = *reinterpret_cast<const float4*>(st.rowp + kc + 4 * kq); }
    }
    static __device__ __forceinline__ uint4 finish(const GemmP& p, const St&, int, int, const Raw& r) {
        float v[8]; f8_to_arr(r.x, v);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] *= p.a_scale;
        return pack8(v);
    }
};

// bf16 copy of the fp32 residual stream -> LayerNorm2d over the row (biased variance, eps inside the sqrt:
// utils.py:18-22) -> folded LN-affine/FiLM gain and bias (conditional_naf.py:114-115,126-127).  The row
// statistics are exact fp32 (producer partials of the unrounded values); only the value being normalised
// is the bf16 copy, which halves the bytes every workgroup ingests.
// PER_FACE is a compile-time property of the kernel: false = every face shares one FiLM row, copied to LDS once per
// workgroup (the sampling loop, scalar timesteps); true = per-face timesteps, gain / bias read from the rows of the global
// table.  A run-time choice between the two sources inside finish() costs the whole K loop its prefetch depth: the pointer
// becomes generic (flat loads + s_waitcnt vmcnt(0) lgkmcnt(0)), and with two typed branches the wait for the global branch
// still lands after the join -- every weight / A load in flight was drained in every chunk (tools/gemm_bench: the LayerNorm
// GEMMs cost 2.6-4.1 us more than the same GEMM with the plain loader).
template <bool PER_FACE>
struct LdF32LN_T {
    static constexpr int kRawRegs = 4;
    static constexpr bool kGainBiasLds = !PER_FACE, kSplit = false, kStaticK = true;
    static constexpr bool kDeep = !PER_FACE;                           // gemm_deep_kernel: the shared-FiLM form only
    struct St { const unsigned short* rowp; const float* gain; const float* bias; const float* gbl; float mu, rstd; bool valid; };   // mu holds -mean*rstd
    struct Raw { uint4 x; };
    // block prologue in two halves so that its (small) loads can be issued before the weight / A streams and
    // consumed while those are still arriving (vector loads return in issue order):
    // (1) merge the producer's per-tile (mean, M2) partials of each row into (mean, rstd) in LDS: TPR threads
    //     per row, Chan's update (Chan, Golub, LeVeque 1979), fixed order -> deterministic;
    // (2) when every row shares one FiLM row (sampling: same t for all faces) copy gain/bias[K] to LDS so
    //     finish() never waits on global memory.
    static constexpr int kPP = 4;                                       // partials held per thread
    struct Pre { float2 s[kPP]; float4 g[2], b[2]; };
    template <int BM, int THREADS> static constexpr int tpr() { return THREADS / BM >= 16 ? 16 : (THREADS / BM >= 1 ? THREADS / BM : 1); }
    template <int BM, int THREADS> static __device__ __forceinline__ bool fast(const GemmP& p) {
        return p.stats_np <= kPP * tpr<BM, THREADS>() && BM * tpr<BM, THREADS>() <= THREADS;
    }
    static __device__ __forceinline__ const float* film_row(const GemmP& p) {
        const int step = p.step_ptr ? *p.step_ptr : 0;
        return p.film + (size_t)step * p.film_step_stride;
    }
    template <int BM, int THREADS> static __device__ __forceinline__ void block_issue(const GemmP& p, int row0, float* gb, int tid, Pre& pre) {
        constexpr int TPR = tpr<BM, THREADS>();
        if (fast<BM, THREADS>(p)) {
            const int rl = tid / TPR, part = tid % TPR, row = row0 + rl;
            const bool rv = rl < BM && row < p.M;
            const float2* sp = p.stats_in + (size_t)(rv ? row : 0) * p.stats_np;
#pragma unroll
            for (int i = 0; i < kPP; ++i) {
                const int j = part + i * TPR;
                pre.s[i] = (rv && j < p.stats_np) ? sp[j] : make_float2(0.f, -1.f);     // M2 < 0 marks "no partial"
            }
        }
        if (gb && p.film_face_stride == 0) {
            const float* f = film_row(p);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int k = (tid + i * THREADS) * 4;
                if (k < p.K) {
                    pre.g[i] = *reinterpret_cast<const float4*>(f + p.film_gain_off + k);
                    pre.b[i] = *reinterpret_cast<const float4*>(f + p.film_bias_off + k);
                }
            }
        }
    }
    template <int BM, int THREADS> static __device__ __forceinline__ void block_finish(const GemmP& p, int row0, char* stats, float* gb, int tid, const Pre& pre) {
        constexpr int TPR = tpr<BM, THREADS>();
        float2* st = reinterpret_cast<float2*>(stats);
        if (gb && p.film_face_stride == 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int k = (tid + i * THREADS) * 4;
                if (k < p.K) {
                    *reinterpret_cast<float4*>(gb + k) = pre.g[i];
                    *reinterpret_cast<float4*>(gb + p.Kp + k) = pre.b[i];
                }
            }
            if (p.K > THREADS * 8) {                                    // longer rows than two passes cover
                const float* f = film_row(p);
                for (int k = (tid + 2 * THREADS) * 4; k < p.K; k += THREADS * 4) {
                    *reinterpret_cast<float4*>(gb + k) = *reinterpret_cast<const float4*>(f + p.film_gain_off + k);
                    *reinterpret_cast<float4*>(gb + p.Kp + k) = *reinterpret_cast<const float4*>(f + p.film_bias_off + k);
                }
            }
        }
        if (fast<BM, THREADS>(p)) {
            // all partials of a row cover the same number of channels, so the exact decomposition
            //   mean = sum(mean_i) / np,   M2 = sum(M2_i + cnt * (mean_i - mean)^2)
            // needs two short sums (own partials, then a DPP butterfly over the row's TPR lanes: lane^1, lane^2,
            // half-row mirror, row mirror) instead of Chan's pairwise updates with their divisions -- this runs
            // on every thread of every LayerNorm GEMM, two waves per SIMD, ahead of the first MFMA.
            const int rl = tid / TPR, part = tid % TPR;
            auto row_sum = [](float v) {
                if (TPR > 1) v += dpp_mov<0xB1>(v);
                if (TPR > 2) v += dpp_mov<0x4E>(v);
                if (TPR > 4) v += dpp_mov<0x141>(v);
                if (TPR > 8) v += dpp_mov<0x140>(v);
                return v;
            };
            float sm = 0.f;
#pragma unroll
            for (int i = 0; i < kPP; ++i) sm += pre.s[i].y >= 0.f ? pre.s[i].x : 0.f;
            const float inv_np = 1.0f / (float)p.stats_np;               // wave-uniform (scalar unit)
            const float mean = row_sum(sm) * inv_np;
            const float cnt = (float)p.stats_cnt;
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < kPP; ++i) {
                const float d = pre.s[i].x - mean;
                q += pre.s[i].y >= 0.f ? fmaf(cnt * d, d, pre.s[i].y) : 0.f;
            }
            const float var = row_sum(q) * (inv_np / cnt);
            if (part == 0 && rl < BM) st[rl] = make_float2(mean, (row0 + rl < p.M) ? __frsqrt_rn(var + p.ln_eps) : 0.f);
        } else {
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
                for (int o = 1; o < 4; o <<= 1) {                       // butterfly over the quad: lane ^ 1, lane ^ 2
                    const float n2 = (o == 1) ? dpp_mov<0xB1>(n) : dpp_mov<0x4E>(n);
                    const float mean2 = (o == 1) ? dpp_mov<0xB1>(mean) : dpp_mov<0x4E>(mean);
                    const float m22 = (o == 1) ? dpp_mov<0xB1>(m2) : dpp_mov<0x4E>(m2);
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
        }
        HD_STAMP(7);
        __syncthreads();
    }
    template <int BM, int THREADS> static __device__ void block_init(const GemmP& p, int row0, char* stats, float* gb, int tid) {
        Pre pre;
        block_issue<BM, THREADS>(p, row0, gb, tid, pre);
        block_finish<BM, THREADS>(p, row0, stats, gb, tid, pre);
    }
    static __device__ __forceinline__ void unit_init(const GemmP& p, St& st, int row, int row_local, const char* stats, const float* gb) {
        st.valid = row < p.M;
        const int r = st.valid ? row : 0;
        st.rowp = reinterpret_cast<const unsigned short*>(p.A) + (size_t)r * p.lda;
        st.mu = 0.f; st.rstd = 0.f;                                  // filled by unit_stats() after block_finish()
        st.gbl = (gb && p.film_face_stride == 0) ? gb : nullptr;      // LDS copy of the shared FiLM row
        const float* f = film_row(p);
        if (p.film_face_stride != 0) f += (size_t)(p.face0 + r / p.hw) * p.film_face_stride;   // per-face timesteps (wave-uniform test)
        st.gain = f + p.film_gain_off;
        st.bias = f + p.film_bias_off;
    }
    static __device__ __forceinline__ void unit_stats(St& st, int row_local, const char* stats) {
        float2 s = reinterpret_cast<const float2*>(stats)[row_local];
        // The row statistics are read once and live in registers for the whole K loop.  In the straight-line K loop, without
        // the statement below, rows 8j+6 / 8j+7 of a tile (lanes 48-63 of a unit) came out different in >= 1 launch of 60 on
        // MI355X, and only when the transform used these values (constants instead: reproducible).  With it: 0 of 5400
        // launches (tools/det_bench).  The cause is not established.  It is not the wait as such -- the kernel built with
        // every compiler wait forced to zero still differs -- but how the code around the first use is laid out: the asm makes
        // the two values opaque, so they are consumed here and kept in registers of their own instead of being read, with
        // op_sel, out of the pair the LDS read delivered.  Checks of the hardware rules this could have broken all came back
        // clean (tools/vmorder_bench, ldswar_bench, mfma_overlap_bench, pkfma_bench).
        // r03 (profiles/r03_unit_stats_isa, tools/det_bench with -DHD_UNIT_STATS_PLAIN=n): opaque values WITHOUT the wait are clean in
        // 299 launches, the wait + idle cycles WITHOUT opacity still differ in 59-93 of 299; the listings differ in one operand:
        // v_pk_fma_f32 ... op_sel:[0,1,0] op_sel_hi:[1,1,0] taking rstd from the high register of the ds_read_b64 pair.
        // tests/test_gpu_parity.py::test_every_launch_is_reproducible guards it.
#if !defined(HD_UNIT_STATS_PLAIN)                                   // tools/det_bench builds the other forms for profiles/r03_unit_stats_isa
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 3" : "+v"(s.x), "+v"(s.y));
#elif HD_UNIT_STATS_PLAIN == 2                                      // opaque values only: no wait, no idle cycles
        asm volatile("" : "+v"(s.x), "+v"(s.y));
#elif HD_UNIT_STATS_PLAIN == 3                                      // wait + idle cycles, values NOT opaque
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 3" ::: "memory");
#endif
        st.mu = -s.x * s.y; st.rstd = s.y;                            // x_hat = fma(x, rstd, -mean*rstd)
    }
    static __device__ __forceinline__ void fetch(const GemmP& p, const St& st, int kc, int kq, Raw& r) {
        r.x = (st.valid && kc + 8 * kq < p.K) ? *reinterpret_cast<const uint4*>(st.rowp + kc + 8 * kq) : make_uint4(0, 0, 0, 0);
    }
    // K % 64 == 0 (host): no bound to test, the load is not under a branch (rows beyond M re-read row 0 and are zeroed by
    // finish()), so the compiler can count the loads in flight (see gemm_skinny_kernel, CPW)
    static constexpr bool kStatic8 = false;
    static bool static_ok(const GemmP&) { return true; }
    static __device__ __forceinline__ void fetch_nc(const GemmP&, const St& st, int kc, int kq, Raw& r) {
        r.x = *reinterpret_cast<const uint4*>(st.rowp + kc + 8 * kq);
    }
    static __device__ __forceinline__ uint4 finish(const GemmP& p, const St& st, int kc, int kq, const Raw& r) {
        float v[8], g[8], b[8];
        if (!(st.valid && kc + 8 * kq < p.K)) return make_uint4(0, 0, 0, 0);
        unpack8(r.x, v);
        const int k = kc + 8 * kq;
        typedef float f4v __attribute__((ext_vector_type(4)));
        f4v g0, g1, b0, b1;
        if constexpr (!PER_FACE) {                                    // shared FiLM row in LDS (block_finish)
            typedef __attribute__((address_space(3))) const f4v lds_f4;
            typedef __attribute__((address_space(3))) const float lds_f1;
            lds_f1* gl = (lds_f1*)st.gbl;
            g0 = *(lds_f4*)(gl + k); g1 = *(lds_f4*)(gl + k + 4);
            b0 = *(lds_f4*)(gl + p.Kp + k); b1 = *(lds_f4*)(gl + p.Kp + k + 4);
        } else {                                                      // rows of the global table
            typedef __attribute__((address_space(1))) const f4v gl_f4;
            typedef __attribute__((address_space(1))) const float gl_f1;
            gl_f1 *gg = (gl_f1*)st.gain, *bg = (gl_f1*)st.bias;
            g0 = *(gl_f4*)(gg + k); g1 = *(gl_f4*)(gg + k + 4);
            b0 = *(gl_f4*)(bg + k); b1 = *(gl_f4*)(bg + k + 4);
        }
        g[0] = g0.x; g[1] = g0.y; g[2] = g0.z; g[3] = g0.w; g[4] = g1.x; g[5] = g1.y; g[6] = g1.z; g[7] = g1.w;
        b[0] = b0.x; b[1] = b0.y; b[2] = b0.z; b[3] = b0.w; b[4] = b1.x; b[5] = b1.y; b[6] = b1.z; b[7] = b1.w;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = fmaf(fmaf(v[i], st.rstd, st.mu), g[i], b[i]);
        return pack8(v);
    }
    // the same arithmetic (two fused multiply-adds per element, round to nearest even) for the straight-line K loop: no
    // bounds (K % 64 == 0; a row beyond M re-reads row 0 with rstd = 0 and is never stored), two elements per
    // instruction (v_pk_fma_f32, v_cvt_pk_bf16_f32).  This transform runs on every workgroup of every LayerNorm GEMM
    // for all of its rows x K: it is VALU time on the critical path, not hidden behind the weight stream.
    // gain / bias of the lane's eight k: the same for every row when all faces share one FiLM row, so they are read from
    // LDS once per chunk, not once per (row, 8 k) unit (the compiler cannot merge the reads itself: an LDS store of the
    // staging tile sits between two units)
    struct ChunkC { f32x2_t g[4], b[4]; };
    static __device__ __forceinline__ void film8(const GemmP& p, const St& st, int k, ChunkC& c) {
        if constexpr (!PER_FACE) {
            typedef __attribute__((address_space(3))) const f32x2_t lds_f2;
            typedef __attribute__((address_space(3))) const float lds_f1;
            lds_f1* gl = (lds_f1*)st.gbl;
#pragma unroll
            for (int i = 0; i < 4; ++i) { c.g[i] = *(lds_f2*)(gl + k + 2 * i); c.b[i] = *(lds_f2*)(gl + p.Kp + k + 2 * i); }
        } else {
            typedef __attribute__((address_space(1))) const f32x2_t gl_f2;
            typedef __attribute__((address_space(1))) const float gl_f1;
            gl_f1 *gg = (gl_f1*)st.gain, *bg = (gl_f1*)st.bias;
#pragma unroll
            for (int i = 0; i < 4; ++i) { c.g[i] = *(gl_f2*)(gg + k + 2 * i); c.b[i] = *(gl_f2*)(bg + k + 2 * i); }
        }
    }
    static __device__ __forceinline__ void chunk_consts(const GemmP& p, const St& st0, int kc, int kq, ChunkC& c) {
        if constexpr (!PER_FACE) film8(p, st0, kc + 8 * kq, c);
    }
    static __device__ __forceinline__ uint4 finish_nc(const GemmP& p, const St& st, int kc, int kq, const Raw& r, const ChunkC& cc) {
        ChunkC own;
        if constexpr (PER_FACE) film8(p, st, kc + 8 * kq, own);      // rows of different faces: per unit
        const ChunkC& c = PER_FACE ? own : cc;
        const unsigned w[4] = {r.x.x, r.x.y, r.x.z, r.x.w};
        const f32x2_t rs = {st.rstd, st.rstd}, mu = {st.mu, st.mu};
        unsigned o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x2_t x = {__uint_as_float(w[i] << 16), __uint_as_float(w[i] & 0xffff0000u)};
            o[i] = pack2(__builtin_elementwise_fma(__builtin_elementwise_fma(x, rs, mu), c.g[i], c.b[i]));
        }
        return make_uint4(o[0], o[1], o[2], o[3]);
    }
};

typedef LdF32LN_T<false> LdF32LN;          // one FiLM row for all faces
typedef LdF32LN_T<true> LdF32LNFace;       // per-face timesteps

// bf16 rows, copied as they are (conv5 input G2, ResNet 1x1 convs)
struct LdBF16Plain {
    static constexpr int kRawRegs = 4;
    static constexpr bool kGainBiasLds = false, kSplit = false, kStaticK = true;
    static constexpr bool kDeep = true;
    struct St { const unsigned short* rowp; bool valid; };
    struct Raw { uint4 x; };
    struct Pre {};
    template <int BM, int THREADS> static __device__ __forceinline__ void block_issue(const GemmP&, int, float*, int, Pre&) {}
    template <int BM, int THREADS> static __device__ __forceinline__ void block_finish(const GemmP&, int, char*, float*, int, const Pre&) {}
    template <int BM, int THREADS> static __device__ void block_init(const GemmP&, int, char*, float*, int) {}
    template <class S> static __device__ __forceinline__ void unit_stats(S&, int, const char*) {}
    static __device__ __forceinline__ void unit_init(const GemmP& p, St& st, int row, int, const char*, const float*) {
        st.valid = row < p.M;
        st.rowp = reinterpret_cast<const unsigned short*>(p.A) + (size_t)(st.valid ? row : 0) * p.lda;
    }
    static __device__ __forceinline__ void fetch(const GemmP& p, const St& st, int kc, int kq, Raw& r) {
        r.x = (st.valid && kc + 8 * kq < p.K) ? *reinterpret_cast<const uint4*>(st.rowp + kc + 8 * kq) : make_uint4(0, 0, 0, 0);
    }
    static constexpr bool kStatic8 = false;
    static bool static_ok(const GemmP&) { return true; }
    static __device__ __forceinline__ void fetch_nc(const GemmP&, const St& st, int kc, int kq, Raw& r) {
        r.x = *reinterpret_cast<const uint4*>(st.rowp + kc + 8 * kq);
    }
    static __device__ __forceinline__ uint4 finish(const GemmP&, const St& st, int, int, const Raw& r) {
        return st.valid ? r.x : make_uint4(0, 0, 0, 0);
    }
    struct ChunkC {};
    static __device__ __forceinline__ void chunk_consts(const GemmP&, const St&, int, int, ChunkC&) {}
    static __device__ __forceinline__ uint4 finish_nc(const GemmP& p, const St& st, int kc, int kq, const Raw& r, const ChunkC&) { return finish(p, st, kc, kq, r); }
};

// bf16 rows times a per-(face, k) fp32 scale: x * sca(x) feeding conv3 (conditional_naf.py:119-120)
struct LdBF16Scale {
    static constexpr int kRawRegs = 12;
    static constexpr bool kGainBiasLds = false, kSplit = false, kStaticK = false;
    struct St { const unsigned short* rowp; const float* srow; bool valid; };
    struct Raw { uint4 x; F8 s; };
    struct Pre {};
    template <int BM, int THREADS> static __device__ __forceinline__ void block_issue(const GemmP&, int, float*, int, Pre&) {}
    template <int BM, int THREADS> static __device__ __forceinline__ void block_finish(const GemmP&, int, char*, float*, int, const Pre&) {}
    template <int BM, int THREADS> static __device__ void block_init(const GemmP&, int, char*, float*, int) {}
    template <class S> static __device__ __forceinline__ void unit_stats(S&, int, const char*) {}
    static __device__ __forceinline__ void unit_init(const GemmP& p, St& st, int row, int, const char*, const float*) {
        st.valid = row < p.M;
        const int r = st.valid ? row : 0;
        st.rowp = reinterpret_cast<const unsigned short*>(p.A) + (size_t)r * p.lda;
        st.srow = p.rowscale + (size_t)(r / p.hw) * p.K;
    }
    static __device__ __forceinline__ void fetch(const GemmP& p, const St& st, int kc, int kq, Raw& r) {
        const bool ok = st.valid && kc + 8 * kq < p.K;
        r.x = ok ? *reinterpret_cast<const uint4*>(st.rowp + kc + 8 * kq) : make_uint4(0, 0, 0, 0);
        F8 s;
        s.a = ok ? *reinterpret_cast<const float4*>(st.srow + kc + 8 * kq) : make_float4(0, 0, 0, 0);
        s.b = ok ? *reinterpret_cast<const float4*>(st.srow + kc + 8 * kq + 4) : make_float4(0, 0, 0, 0);
        r.s = s;
    }
    static __device__ __forceinline__ uint4 finish(const GemmP&, const St&, int, int, const Raw& r) {
        float v[8], sc[8]; unpack8(r.x, v); f8_to_arr(r.s, sc);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] *= sc[i];
        return pack8(v);
    }
};

// im2col gather over a channels-last image: k = tap*Cin + c.  SRC_BF16: ResNet activations (8 consecutive
// k per lane); fp32 (4 + 4 split, each half resolved to its own tap);
// GATED: the HCA 3x3 conv input f_d*(1 + w_c + w_s) (+ idc term) (hca.py:28, model.py:245-246)
template <bool SRC_BF16, bool GATED>
struct LdConv {
    static constexpr int kRawRegs = (SRC_BF16 ? 4 : 8) + (GATED ? 18 : 0) + 2;
    static constexpr bool kGainBiasLds = false, kSplit = !SRC_BF16, kStaticK = SRC_BF16 && !GATED;
    struct St { int b, iy0, ix0; bool valid; };
    struct Raw { float4 x[2]; float4 add[2]; float4 gc[2]; float gs[2]; uint4 xb; bool inb[2]; };
    // straight-line K loop: every 64-deep chunk lies inside one tap and no tap falls outside the image (the stride-2 2x2
    // down-convs and the 1x1 convs), so the gather needs no bounds
    static constexpr bool kStatic8 = true;                      // K = 4096 (8 chunks per wave) occurs: downs.3
    static bool static_ok(const GemmP& p) {
        return p.pad == 0 && p.Cin % 64 == 0 && p.ntaps == p.KH * p.KW && p.ntaps * p.Cin == p.K &&
               (p.Hout - 1) * p.stride + p.KH <= p.Hin && (p.Wout - 1) * p.stride + p.KW <= p.Win;
    }
    struct ChunkC {};
    static __device__ __forceinline__ void chunk_consts(const GemmP&, const St&, int, int, ChunkC&) {}
    static __device__ __forceinline__ void fetch_nc(const GemmP& p, const St& st, int kc, int kq, Raw& r) {
        const int k = kc + 8 * kq, tap = k / p.Cin, c0 = k - tap * p.Cin, ky = tap / p.KW;
        const size_t srow = ((size_t)st.b * p.Hin + st.iy0 + ky) * p.Win + st.ix0 + (tap - ky * p.KW);
        r.xb = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(p.A) + srow * p.Cin + c0);
    }
    static __device__ __forceinline__ uint4 finish_nc(const GemmP&, const St& st, int, int, const Raw& r, const ChunkC&) {
        return st.valid ? r.xb : make_uint4(0, 0, 0, 0);
    }
    struct Pre {};
    template <int BM, int THREADS> static __device__ __forceinline__ void block_issue(const GemmP&, int, float*, int, Pre&) {}
    template <int BM, int THREADS> static __device__ __forceinline__ void block_finish(const GemmP&, int, char*, float*, int, const Pre&) {}
    template <int BM, int THREADS> static __device__ void block_init(const GemmP&, int, char*, float*, int) {}
    template <class S> static __device__ __forceinline__ void unit_stats(S&, int, const char*) {}
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
    // source row index of (tap-resolved) k; returns false if padding / out of range
    static __device__ __forceinline__ bool locate(const GemmP& p, const St& st, int k, size_t& srow, int& c0) {
        const int tap = k / p.Cin;
        c0 = k - tap * p.Cin;
        const int ky = tap / p.KW;
        const int iy = st.iy0 + ky, ix = st.ix0 + (tap - ky * p.KW);
        const bool ok = st.valid && tap < p.ntaps && iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win;
        srow = ok ? ((size_t)st.b * p.Hin + iy) * p.Win + ix : 0;
        return ok;
    }
    static __device__ __forceinline__ void fetch(const GemmP& p, const St& st, int kc, int kq, Raw& r) {
        if (SRC_BF16) {
            size_t srow; int c0;
            r.inb[0] = locate(p, st, kc + 8 * kq, srow, c0);
            if (r.inb[0]) r.xb = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(p.A) + srow * p.Cin + c0);
        } else {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                size_t srow; int c0;
                r.inb[h] = locate(p, st, kc + 32 * h + 4 * kq, srow, c0);
                if (!r.inb[h]) continue;
                r.x[h] = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.A) + srow * p.Cin + c0);
                if (GATED) {
                    r.gs[h] = p.gate_s[srow];
                    r.gc[h] = *reinterpret_cast<const float4*>(p.gate_c + (size_t)st.b * p.Cin + c0);
                    r.add[h] = p.add_src ? *reinterpret_cast<const float4*>(p.add_src + srow * p.Cin + c0) : make_float4(0, 0, 0, 0);
                }
            }
        }
    }
    static __device__ __forceinline__ uint4 finish(const GemmP&, const St&, int, int, const Raw& r) {
        if (SRC_BF16) return r.inb[0] ? r.xb : make_uint4(0, 0, 0, 0);
        float v[8];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float4 x = r.inb[h] ? r.x[h] : make_float4(0, 0, 0, 0);
            if (GATED && r.inb[h]) {
                const float4 a = r.add[h], g = r.gc[h];
                const float s = 1.0f + r.gs[h];
                x.x = (x.x + a.x) * (s + g.x); x.y = (x.y + a.y) * (s + g.y);
                x.z = (x.z + a.z) * (s + g.z); x.w = (x.w + a.w) * (s + g.w);
            }
            v[4 * h] = x.x; v[4 * h + 1] = x.y; v[4 * h + 2] = x.z; v[4 * h + 3] = x.w;
        }
        return pack8(v);
    }
};

// ---------------------------------------------------------------------------------------- epilogues
// Protocol (element-wise epilogues): per output column the lane loads its constants once (col_init); the
// loads that may alias `out` in the compiler's eyes (residual / skip tensors) are issued for the whole
// tile first (pre), then all stores follow — otherwise every store->load pair is serialised by a
// vmcnt(0) wait (measured: 4.6 us per kernel).
__device__ __forceinline__ float activate(float v, int act) {
    if (act == 1) return fmaxf(v, 0.f);
    if (act == 2) return 1.0f / (1.0f + expf(-v));
    return v;
}
struct ColC { float bias, bias2, rscale, rscale2; };

// out(fp32)[row][col] = act(acc + bias)
struct EpBiasF32 {
    static constexpr bool kStats = true, kTile = false;
    static __device__ __forceinline__ size_t stat_index(const GemmP& p, int row, int tile_idx) { return (size_t)row * (p.N >> 5) + tile_idx; }
    static __device__ __forceinline__ ColC col_init(const GemmP& p, int col) { ColC c; c.bias = p.bias ? p.bias[col] : 0.f; c.bias2 = 0.f; c.rscale = 1.f; return c; }
    static __device__ __forceinline__ float pre(const GemmP&, int, int) { return 0.f; }
    static __device__ __forceinline__ float store(const GemmP& p, int row, int col, float v, float, const ColC& c) {
        v = activate(v + c.bias, p.act);
        reinterpret_cast<float*>(p.out)[(size_t)row * p.ldo + col] = v;
        if (p.out16) p.out16[(size_t)row * p.ldo + col] = f32_to_bf16_bits(v);
        return v;
    }
};
// SCA: s = acc + bias (fp32 out, rows = faces) and, in place, G[pixels of the face][col] <- bf16(G * s): the
// product x * sca(x) of conditional_naf.py:119 with the same rounding as the scale loader, so conv3 reads G plainly.
// Runs as a tile epilogue in the skinny kernel (all threads sweep the faces' pixel rows, 16 B per access).
struct EpScaBF16 {
    static constexpr bool kStats = false, kTile = false, kScaTile = true;
    static __device__ __forceinline__ ColC col_init(const GemmP& p, int col) { ColC c; c.bias = p.bias[col]; c.bias2 = 0.f; c.rscale = 1.f; return c; }
    static __device__ __forceinline__ float pre(const GemmP&, int, int) { return 0.f; }
    static __device__ __forceinline__ float store(const GemmP& p, int row, int col, float v, float, const ColC& c) {
        v += c.bias;
        reinterpret_cast<float*>(p.out)[(size_t)row * p.ldo + col] = v;
        return v;
    }
};
// out(fp32) = resid + rscale[col] * (acc + bias)      (y = inp + x*beta, out = y + x*gamma)
struct EpResidF32 {
    static constexpr bool kStats = true, kTile = false;
    static constexpr bool kDeep = true;
    static __device__ __forceinline__ size_t stat_index(const GemmP& p, int row, int tile_idx) { return (size_t)row * (p.N >> 5) + tile_idx; }
    static __device__ __forceinline__ ColC col_init(const GemmP& p, int col) { ColC c; c.bias = p.bias[col]; c.bias2 = 0.f; c.rscale = p.rscale[col]; return c; }
    static __device__ __forceinline__ float pre(const GemmP& p, int row, int col) {
        return reinterpret_cast<const float*>(p.resid)[(size_t)row * p.ldr + col];
    }
    static __device__ __forceinline__ float store(const GemmP& p, int row, int col, float v, float r, const ColC& c) {
        v = fmaf(v + c.bias, c.rscale, r);
        reinterpret_cast<float*>(p.out)[(size_t)row * p.ldo + col] = v;
        if (p.out16) p.out16[(size_t)row * p.ldo + col] = f32_to_bf16_bits(v);
        if (p.outg16) {                                       // f_d * (1 + w_c + w_s) (+ idc term) for the following HCA
            const float a = p.add_src ? p.add_src[(size_t)row * p.ldo + col] : 0.f;
            const float g = 1.0f + p.gate_c[(size_t)(row / p.hw) * p.N + col] + p.gate_s[row];
            p.outg16[(size_t)row * p.ldo + col] = f32_to_bf16_bits((v + a) * g);
        }
        return v;
    }
};
// PAIR: out(bf16)[row][col] = (acc1 + bias[col]) * (acc2 + bias[col + N/2])   (conv4 -> SimpleGate)
struct EpGateBF16 {
    static constexpr bool kStats = false, kTile = false;
    static constexpr bool kDeep = true;
    static __device__ __forceinline__ ColC col_init(const GemmP& p, int col) { ColC c; c.bias = p.bias[col]; c.bias2 = p.bias[col + (p.N >> 1)]; c.rscale = 1.f; return c; }
    static __device__ __forceinline__ float pre(const GemmP&, int, int) { return 0.f; }
    static __device__ __forceinline__ void store2(const GemmP& p, int row, int col, float v1, float v2, const ColC& c) {
        reinterpret_cast<unsigned short*>(p.out)[(size_t)row * p.ldo + col] = f32_to_bf16_bits((v1 + c.bias) * (v2 + c.bias2));
    }
};
// PAIR, faces of ONE pixel (middle level at latent 16): the depthwise 3x3 sees only its centre tap, the pool is the
// value itself, so conv1 -> conv2 -> SimpleGate -> pool is element-wise on the two accumulators:
//   g = (b2[j] + w2[j][centre] * (acc1 + b1[j])) * (b2[j+C] + w2[j+C][centre] * (acc2 + b1[j+C]))
struct EpDwGate1 {
    static constexpr bool kStats = false, kTile = false;
    static __device__ __forceinline__ ColC col_init(const GemmP& p, int col) {
        // fold: g = (A1 + wa * acc1) * (A2 + wb * acc2) with A = b2 + w * b1
        const int C2 = p.N >> 1;
        const float wa = p.dw_w[(size_t)4 * p.N + col], wb = p.dw_w[(size_t)4 * p.N + col + C2];      // tap-major copy
        ColC c; c.bias = p.dw_b[col] + wa * p.bias[col]; c.bias2 = p.dw_b[col + C2] + wb * p.bias[col + C2];
        c.rscale = wa; c.rscale2 = wb;
        return c;
    }
    static __device__ __forceinline__ float pre(const GemmP&, int, int) { return 0.f; }
    static __device__ __forceinline__ void store2(const GemmP& p, int row, int col, float v1, float v2, const ColC& c) {
        // same operation order as the tile path: t1 = acc + b1; u = b2 + w * t1 is evaluated as (b2 + w*b1) + w*acc
        const float g = fmaf(c.rscale, v1, c.bias) * fmaf(c.rscale2, v2, c.bias2);
        const size_t o = (size_t)row * p.ldo + col;
        reinterpret_cast<unsigned short*>(p.out)[o] = f32_to_bf16_bits(g);
        p.pooled[o] = g;                                          // one pixel per face: row == face
        if (p.pooled16) p.pooled16[o] = f32_to_bf16_bits(g);
    }
};
// SCA at one pixel per face: s = acc + bias and G[face][col] <- bf16(G * s), element-wise (pre() fetches G early)
struct EpSca1BF16 {
    static constexpr bool kStats = false, kTile = false;
    static __device__ __forceinline__ ColC col_init(const GemmP& p, int col) { ColC c; c.bias = p.bias[col]; c.bias2 = 0.f; c.rscale = 1.f; c.rscale2 = 0.f; return c; }
    static __device__ __forceinline__ float pre(const GemmP& p, int row, int col) { return bf16_bits_to_f32(p.scale_G[(size_t)row * p.ldo + col]); }
    static __device__ __forceinline__ float store(const GemmP& p, int row, int col, float v, float g, const ColC& c) {
        v += c.bias;
        reinterpret_cast<float*>(p.out)[(size_t)row * p.ldo + col] = v;
        p.scale_G[(size_t)row * p.ldo + col] = f32_to_bf16_bits(g * v);
        return v;
    }
};
// 1x1 conv (no bias) -> PixelShuffle(r) -> + skip.  For r = 2 the weight columns are packed sub-pixel major
// (n' = (2i + j) * Cout + c  <-  n = c*4 + 2i + j, PackOpts::S2 = 4), so a 32-column tile is a run of 32 output
// channels of ONE output pixel: out[b, 2h+i, 2w+j, c..c+31] is a contiguous 128-byte store and the tile carries
// a LayerNorm partial of that output row like every other producer of the residual stream.
struct EpPixShufF32 {
    static constexpr bool kStats = true, kTile = false;
    static __device__ __forceinline__ ColC col_init(const GemmP&, int) { ColC c; c.bias = 0.f; c.bias2 = 0.f; c.rscale = 1.f; return c; }
    static __device__ __forceinline__ size_t out_row(const GemmP& p, int row, int sub) {       // Hin == Win == a power of two
        const int lw = 31 - __builtin_clz(p.Win);
        const int b = row >> (2 * lw), rem = row & ((1 << (2 * lw)) - 1);
        const int h = rem >> lw, w = rem & (p.Win - 1);
        return (((size_t)b << (lw + 1)) + (2 * h + (sub >> 1))) * (2 * p.Win) + (2 * w + (sub & 1));
    }
    static __device__ __forceinline__ size_t index(const GemmP& p, int row, int col) {
        if (p.shuffle_r == 2) {
            const int lc = 31 - __builtin_clz(p.ldo);                                         // Cout is a power of two
            return out_row(p, row, col >> lc) * p.ldo + (col & (p.ldo - 1));
        }
        return (size_t)row * p.ldo + col;
    }
    static __device__ __forceinline__ size_t stat_index(const GemmP& p, int row, int tile_idx) {
        if (p.shuffle_r == 2) {
            const int lt = 31 - __builtin_clz(p.ldo >> 5);                                    // tiles per output row
            return out_row(p, row, tile_idx >> lt) * (p.ldo >> 5) + (tile_idx & ((p.ldo >> 5) - 1));
        }
        return (size_t)row * (p.N >> 5) + tile_idx;
    }
    static __device__ __forceinline__ float pre(const GemmP& p, int row, int col) {
        return p.resid ? reinterpret_cast<const float*>(p.resid)[index(p, row, col)] : 0.f;
    }
    static __device__ __forceinline__ float store(const GemmP& p, int row, int col, float v, float r, const ColC&) {
        v += r;
        const size_t o = index(p, row, col);
        reinterpret_cast<float*>(p.out)[o] = v;
        if (p.out16) p.out16[o] = f32_to_bf16_bits(v);
        return v;
    }
};
// out(bf16) = act(acc + bias (+ resid bf16))           (ResNet conv+BN(+identity)+ReLU, BN folded)
struct EpBiasBF16 {
    static constexpr bool kStats = false, kTile = false;
    static __device__ __forceinline__ ColC col_init(const GemmP& p, int col) { ColC c; c.bias = p.bias[col]; c.bias2 = 0.f; c.rscale = 1.f; return c; }
    static __device__ __forceinline__ float pre(const GemmP& p, int row, int col) {
        return p.resid ? bf16_bits_to_f32(reinterpret_cast<const unsigned short*>(p.resid)[(size_t)row * p.ldr + col]) : 0.f;
    }
    static __device__ __forceinline__ float store(const GemmP& p, int row, int col, float v, float r, const ColC& c) {
        v = activate(v + c.bias + r, p.act);
        reinterpret_cast<unsigned short*>(p.out)[(size_t)row * p.ldo + col] = f32_to_bf16_bits(v);
        return v;
    }
};
// PAIR, skinny kernel only: the workgroup's T1 tile (conv1 output of whole faces, channels j and j+N/2)
// stays in LDS; depthwise 3x3 (pad 1) on both halves, SimpleGate, G (bf16) and the per-face average pool
// (conditional_naf.py:116-119 / naf.py:109-112).  Needs BM % hw == 0.
struct EpDwGate {
    static constexpr bool kStats = false, kTile = true;
    static __device__ __forceinline__ ColC col_init(const GemmP&, int) { ColC c; c.bias = 0.f; c.bias2 = 0.f; c.rscale = 1.f; return c; }
    static __device__ __forceinline__ float pre(const GemmP&, int, int) { return 0.f; }
    static __device__ __forceinline__ void store2(const GemmP&, int, int, float, float, const ColC&) {}
};

// LayerNorm partial of one 32-column tile: the 32 values of a row sit in the 32 lanes of a half-wave.
// Single pass (sum, sum of squares); the result is valid in lanes 16..31 / 48..63 (kStatLane).
constexpr int kStatLane = 16;
__device__ __forceinline__ float2 halfwave_mean_m2(float v) {
    const float s1 = halfwave_sum_hi(v), s2 = halfwave_sum_hi(v * v);
    const float mean = s1 * (1.0f / 32.0f);
    return make_float2(mean, fmaxf(fmaf(-s1, mean, s2), 0.f));
}

// Epilogue of one 32x32 accumulator tile in MFMA layout (col = lane&31, row = (i&3) + 8*(i>>2) + 4*(lane>>5)).
// FULL: the tile lies completely inside [M) x [ncols) (wave-uniform) -> no per-element predicates, so the
// 16 loads, 16 stores and the statistics math are emitted as straight-line batches.
template <bool FULL, bool PAIR, class EP>
__device__ __forceinline__ void tile_epilogue_mfma(const GemmP& p, const f32x16_t& a1, const f32x16_t& a2, int rbase, int col,
                                                   int ncols, int tile_idx, int lane) {
    const bool cv = FULL || col < ncols;
    const ColC cc = EP::col_init(p, cv ? col : 0);
    float pre[16], v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int row = rbase + (i & 3) + 8 * (i >> 2);
        pre[i] = (FULL || (cv && row < p.M)) ? EP::pre(p, row, col) : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int row = rbase + (i & 3) + 8 * (i >> 2);
        v[i] = 0.f;
        if (FULL || (cv && row < p.M)) {
            if constexpr (PAIR) EP::store2(p, row, col, a1[i], a2[i], cc);
            else v[i] = EP::store(p, row, col, a1[i], pre[i], cc);
        }
    }
    if constexpr (EP::kStats) {
        if (p.stats_out) {                                     // rows of `out` feed a LayerNorm next
            float2 ms[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) ms[i] = halfwave_mean_m2(v[i]);
            if ((lane & 31) == kStatLane) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int row = rbase + (i & 3) + 8 * (i >> 2);
                    if (FULL || row < p.M) p.stats_out[EP::stat_index(p, row, tile_idx)] = ms[i];
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------- tall kernel
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
    static constexpr int GB_OFF = STATS_OFF + BM * 8;   // + 2*Kp floats of FiLM gain/bias for the LN loader
    static constexpr int SMEM = GB_OFF;
    static_assert(UNITS >= 1 && UNITS * THREADS == BM * 8, "tile/threads mismatch");
};

template <class C, class LD, class EP>
__global__ __launch_bounds__(C::THREADS) void gemm_kernel(const GemmP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int w_m = wave / C::WN, w_n = wave - w_m * C::WN;
    const int row0 = blockIdx.x * C::BM;
    const int ksteps_total = p.Kp >> 4;
    const int c_end = p.Kp >> 6;

    int tile[C::TNT];
#pragma unroll
    for (int tn = 0; tn < C::TN; ++tn) {
        const int t = (blockIdx.y * C::WN + w_n) * C::TN + tn;
        tile[tn] = t;
        if (C::PAIR) tile[C::TN + tn] = t + (p.N >> 6);    // second half starts at N/2 = 32*(N/64)
    }
    const int tiles_half = C::PAIR ? (p.N >> 6) : p.nt_total;

    HD_STAMP(0);
    float* gb = LD::kGainBiasLds ? reinterpret_cast<float*>(smem + C::GB_OFF) : nullptr;
    typename LD::St st[C::UNITS];
    int u_ldsoff[C::UNITS];
    const int kq = tid & 7;
#pragma unroll
    for (int u = 0; u < C::UNITS; ++u) {                   // row pointers (statistics filled in below)
        const int rl = (tid >> 3) + u * (C::THREADS / 8);
        u_ldsoff[u] = rl * LDS_ROW;
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
    _Pragma("unroll") for (int u = 0; u < C::UNITS; ++u) LD::fetch(p, st[u], (chunk) * BK, kq, raw[u]);
#define HD_WRITE_A(chunk, buf)                                                                        \
    _Pragma("unroll") for (int u = 0; u < C::UNITS; ++u)                                              \
        lds_write_unit<LD::kSplit>(smem + (buf) * C::A_BUF + u_ldsoff[u], kq, LD::finish(p, st[u], (chunk) * BK, kq, raw[u]));

    HD_FETCH_A(0);
    HD_LOAD_B(bcur, 0);
    // the loads above do not depend on the LayerNorm statistics; merge them while the loads fly
    LD::template block_init<C::BM, C::THREADS>(p, row0, smem + C::STATS_OFF, gb, tid);
#pragma unroll
    for (int u = 0; u < C::UNITS; ++u) LD::unit_stats(st[u], (tid >> 3) + u * (C::THREADS / 8), smem + C::STATS_OFF);
    HD_STAMP(1);
    HD_WRITE_A(0, 0);
    __syncthreads();
    HD_STAMP(2);

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
    HD_STAMP(3);
    HD_STAMP(4);

    // ---- epilogue ----
    const int ncols = C::PAIR ? (p.N >> 1) : p.N;
#pragma unroll
    for (int mt = 0; mt < C::MT; ++mt) {
        const int rtile = row0 + (w_m * C::MT + mt) * 32;
        const int rbase = rtile + 4 * (lane >> 5);
#pragma unroll
        for (int tn = 0; tn < C::TN; ++tn) {
            if (tile[tn] * 32 >= ncols) continue;                      // wave-uniform
            const int col = tile[tn] * 32 + (lane & 31);
            const bool full = (rtile + 32 <= p.M) && (tile[tn] * 32 + 32 <= ncols);   // wave-uniform
            if (full) tile_epilogue_mfma<true, C::PAIR, EP>(p, acc[mt][tn], acc[mt][C::PAIR ? C::TN + tn : tn], rbase, col, ncols, tile[tn], lane);
            else tile_epilogue_mfma<false, C::PAIR, EP>(p, acc[mt][tn], acc[mt][C::PAIR ? C::TN + tn : tn], rbase, col, ncols, tile[tn], lane);
        }
    }
    HD_STAMP(5);
}


// ------------------------------------------------------------------------------------- tall kernel, deep prefetch
// The tall kernel above keeps ONE chunk in flight and drains it at every chunk's __syncthreads(): with K = 512 .. 1024 that is
// 8 .. 16 exposed memory round trips, which is why the skinny kernel (whole K slice of a wave in flight) was used for every long-K
// GEMM.  At latent 32 the LayerNorm / plain GEMMs of levels 2 and 3 have M = 1024 .. 4096 rows: 64-row skinny tiles put
// (64 + 64) rows of operands through every workgroup and two workgroups on a CU.  This form is the tall kernel's tiling
// (128 rows x 32 (gate) columns, A through LDS, one row tile per wave: (128 + 32 or 64) operand rows per CU) with a straight-line
// K loop: NCH = K / 64 is a template parameter, loads are unconditional (host checks: K % 64 == 0, whole column tiles; rows
// beyond M re-read row 0 and are never stored) and P chunks ahead (register rings of raw A units and of this wave's quarter of
// the chunk's B fragments: each 1 KiB fragment is requested by ONE wave and shared through LDS -- with every wave loading its own
// copy the four-fold B requests made the pair GEMM slower than the skinny kernel, 19.4 against 17.6 us), the chunk barrier is
// LDS-only (no vector-memory drain), so the compiler's counted waits leave P chunks in flight.
template <class LD, class = void> struct ld_is_deep { static constexpr bool value = false; };
template <class LD> struct ld_is_deep<LD, decltype((void)LD::kDeep)> { static constexpr bool value = LD::kDeep; };
template <class EP, class = void> struct ep_is_deep { static constexpr bool value = false; };
template <class EP> struct ep_is_deep<EP, decltype((void)EP::kDeep)> { static constexpr bool value = EP::kDeep; };

template <class C, class LD, class EP, int NCH, int P>
__global__ __launch_bounds__(C::THREADS) void gemm_deep_kernel(const GemmP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    static_assert(C::WN == 1 && C::TN == 1 && C::WM == 4 && NCH >= P + 1, "one 32-column (pair) tile per workgroup, four row tiles");
    constexpr int NF = C::TNT * 4;                         // 1 KiB B fragments per chunk: (gate half, k-step)
    constexpr int FPW = NF / 4;                            // ... requested per wave: every fragment crosses the CU's load path once,
    constexpr int B_BUF = NF * 1024;                       // the four waves share it through LDS
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int w_m = wave;
    const int ksteps_total = p.Kp >> 4;
    // block -> (row group, column tile): the row groups that stream one weight tile get linear ids that are equal mod 8 (one
    // XCD, one L2 fill per weight tile) when the weights outweigh the activations (gemm_skinny_kernel's map)
    int bx = blockIdx.x, by = blockIdx.y;
    if (p.xcd_tile_affine && (gridDim.y & 7) == 0) {
        const int lin = blockIdx.y * gridDim.x + blockIdx.x, j = lin >> 3;
        bx = j % (int)gridDim.x;
        by = (j / (int)gridDim.x) * 8 + (lin & 7);
    }
    const int row0 = bx * C::BM;
    const int tile0 = by, tile1 = by + (p.N >> 6);         // pair: the second gate half starts at N/2 = 32 * (N/64)

    float* gb = LD::kGainBiasLds ? reinterpret_cast<float*>(smem + C::GB_OFF) : nullptr;
    char* sB = smem + C::GB_OFF + (LD::kGainBiasLds ? 2 * p.Kp * 4 : 0);
    typename LD::St st[C::UNITS];
    int u_ldsoff[C::UNITS];
    const int kq = tid & 7;
#pragma unroll
    for (int u = 0; u < C::UNITS; ++u) {
        const int rl = (tid >> 3) + u * (C::THREADS / 8);
        u_ldsoff[u] = rl * LDS_ROW;
        LD::unit_init(p, st[u], row0 + rl, rl, smem + C::STATS_OFF, gb);
    }
    f32x16_t acc[C::TNT];
#pragma unroll
    for (int tn = 0; tn < C::TNT; ++tn)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[tn][i] = 0.f;

    typename LD::Raw raw[P][C::UNITS];
    typedef unsigned dq_u32x4 __attribute__((ext_vector_type(4)));
    dq_u32x4 braw0[P], braw1[P];
    // this wave's fragments of a chunk: f = wave * FPW + i -> (gate half f / 4, k-step f % 4)
    const int f0 = wave * FPW, f1 = wave * FPW + (FPW - 1);
    const uint4* Wf0 = p.W + ((size_t)((f0 >> 2) ? tile1 : tile0) * ksteps_total + (f0 & 3)) * 64 + lane;
    const uint4* Wf1 = p.W + ((size_t)((f1 >> 2) ? tile1 : tile0) * ksteps_total + (f1 & 3)) * 64 + lane;
    char* sBw = sB + (wave * FPW) * 1024 + lane * 16;      // where this wave's fragments go
#define HD_DEEP_LOAD(chunk)                                                                               \
    {                                                                                                      \
        _Pragma("unroll") for (int u = 0; u < C::UNITS; ++u) LD::fetch_nc(p, st[u], (chunk) * BK, kq, raw[(chunk) % P][u]); \
        braw0[(chunk) % P] = *reinterpret_cast<const dq_u32x4*>(Wf0 + (size_t)(chunk) * 4 * 64);           \
        if (FPW == 2) braw1[(chunk) % P] = *reinterpret_cast<const dq_u32x4*>(Wf1 + (size_t)(chunk) * 4 * 64); \
    }
#define HD_DEEP_WRITE(chunk)                                                                              \
    {                                                                                                      \
        typename LD::ChunkC cc;                                                                            \
        LD::chunk_consts(p, st[0], (chunk) * BK, kq, cc);                                                  \
        _Pragma("unroll") for (int u = 0; u < C::UNITS; ++u)                                               \
            lds_write_unit<LD::kSplit>(smem + ((chunk) & 1) * C::A_BUF + u_ldsoff[u], kq,                  \
                                       LD::finish_nc(p, st[u], (chunk) * BK, kq, raw[(chunk) % P][u], cc)); \
        *reinterpret_cast<dq_u32x4*>(sBw + ((chunk) & 1) * B_BUF) = braw0[(chunk) % P];                    \
        if (FPW == 2) *reinterpret_cast<dq_u32x4*>(sBw + ((chunk) & 1) * B_BUF + 1024) = braw1[(chunk) % P]; \
    }
    HD_STAMP(0);
#pragma unroll
    for (int d = 0; d < P; ++d) HD_DEEP_LOAD(d);
    HD_STAMP(1);
    // LayerNorm statistics / FiLM row: do not depend on the loads above (its workgroup barrier drains them once)
    LD::template block_init<C::BM, C::THREADS>(p, row0, smem + C::STATS_OFF, gb, tid);
#pragma unroll
    for (int u = 0; u < C::UNITS; ++u) LD::unit_stats(st[u], (tid >> 3) + u * (C::THREADS / 8), smem + C::STATS_OFF);
    HD_DEEP_WRITE(0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    HD_STAMP(2);

    const int a_lane_off = (lane & 31) * LDS_ROW + (lane >> 5) * 16;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        if (c + P < NCH) {
            HD_DEEP_LOAD(c + P);
            // keep the requests HERE: left to itself the scheduler sinks them to just before their use P chunks later (shorter
            // live ranges), i.e. issues them and waits for them at once
            asm volatile("" ::: "memory");
        }
        const char* sA = smem + (c & 1) * C::A_BUF + a_lane_off + (w_m * 32) * LDS_ROW;
        const char* sBc = sB + (c & 1) * B_BUF + lane * 16;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const bf16x8_t a = *reinterpret_cast<const bf16x8_t*>(sA + s4 * 32);
#pragma unroll
            for (int tn = 0; tn < C::TNT; ++tn) {
                const bf16x8_t b = *reinterpret_cast<const bf16x8_t*>(sBc + (tn * 4 + s4) * 1024);
                acc[tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[tn], 0, 0, 0);
            }
        }
        if (c + 1 < NCH) HD_DEEP_WRITE(c + 1);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
#undef HD_DEEP_LOAD
#undef HD_DEEP_WRITE
    HD_STAMP(3); HD_STAMP(4);

    const int ncols = C::PAIR ? (p.N >> 1) : p.N;
    const int rtile = row0 + w_m * 32;
    const int rbase = rtile + 4 * (lane >> 5);
    const int col = tile0 * 32 + (lane & 31);
    if (rtile + 32 <= p.M) tile_epilogue_mfma<true, C::PAIR, EP>(p, acc[0], acc[C::PAIR ? 1 : 0], rbase, col, ncols, tile0, lane);
    else tile_epilogue_mfma<false, C::PAIR, EP>(p, acc[0], acc[C::PAIR ? 1 : 0], rbase, col, ncols, tile0, lane);
    HD_STAMP(5);
}

// The same for a PAIR GEMM (LayerNorm -> conv4 -> SimpleGate) with EIGHT waves: four row tiles x the two gate halves.  A wave
// owns one half (one accumulator, four MFMAs per chunk instead of eight behind each other), requests ONE of the chunk's eight B
// fragments, and the LayerNorm transform of the A chunk is spread over 512 threads; the second half's accumulators meet the first
// half's through LDS once, in the epilogue.  (With four waves a 128-row workgroup is one wave per SIMD and the chunk chain --
// LDS reads, eight dependent MFMAs, transform, LDS writes, barrier -- has nothing to overlap with: 18.3 us against 17.6 us for
// the skinny kernel at M = 1024.)
template <int S>
__device__ __forceinline__ float dw_gate_row(const float* t1a, const float* t1b, int p0, bool up, bool dn, const float* wa,
                                             const float* wb, float ba, float bb, unsigned short* gout, int ldo, bool store_ok,
                                             int rows_left);          // below (skinny kernel's tile epilogue)

// (r04, tools/deep_bench.hip: at M = 1024, K = 1024 the K loop takes 10.3 of the launch's 17 us = 0.63 us per 64-deep chunk, and neither the
// prefetch depth (P = 4 / 6 / 7: 10.3 / 10.1 / 9.8) nor requesting the fragments ahead of the next chunk's transform (10.3) nor two workgroups
// per CU for the depthwise epilogue (128 registers, 4 spilled: 19.1 -> 20.9 us) moves it: per chunk ~120 KB cross the CU's LDS -- fragments read
// by two (A) and four (B) waves each, FiLM gain / bias by every thread -- for 32 MFMAs)
#ifndef HD_DEEP_SKEW
#define HD_DEEP_SKEW 0
#endif
template <class LD, class EP, int NCH, int P>
__global__ __launch_bounds__(512) void gemm_deep_pair8_kernel(const GemmP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BM = 128, THREADS = 512, UNITS = BM * 8 / THREADS;           // 2 (row, 8 k) staging units per thread
    constexpr int A_BUF = BM * LDS_ROW, STATS_OFF = 2 * A_BUF, GB_OFF = STATS_OFF + BM * 8, B_BUF = 8 * 1024;
    static_assert(NCH >= P + 1 && 4 * 16 * 64 * 4 <= 2 * A_BUF, "prefetch depth; the epilogue exchange fits the dead A buffers");
    typedef unsigned dq_u32x4 __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int w_m = wave & 3, half = wave >> 2;
    const int ksteps_total = p.Kp >> 4;
    int bx = blockIdx.x, by = blockIdx.y;
    if (p.xcd_tile_affine && (gridDim.y & 7) == 0) {                   // gemm_deep_kernel's block map
        const int lin = blockIdx.y * gridDim.x + blockIdx.x, j = lin >> 3;
        bx = j % (int)gridDim.x;
        by = (j / (int)gridDim.x) * 8 + (lin & 7);
    }
    const int row0 = bx * BM;
    const int tile0 = by, tile1 = by + (p.N >> 6);

    float* gb = LD::kGainBiasLds ? reinterpret_cast<float*>(smem + GB_OFF) : nullptr;
    char* sB = smem + GB_OFF + (LD::kGainBiasLds ? 2 * p.Kp * 4 : 0);
    typename LD::St st[UNITS];
    int u_ldsoff[UNITS];
    const int kq = tid & 7;
#pragma unroll
    for (int u = 0; u < UNITS; ++u) {
        const int rl = (tid >> 3) + u * (THREADS / 8);
        u_ldsoff[u] = rl * LDS_ROW;
        LD::unit_init(p, st[u], row0 + rl, rl, smem + STATS_OFF, gb);
    }
    f32x16_t acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;

    typename LD::Raw raw[P][UNITS];
    dq_u32x4 braw[P];
    // this wave's B fragment of a chunk: f = wave -> (gate half wave / 4, k-step wave % 4)
    const uint4* Wf = p.W + ((size_t)(half ? tile1 : tile0) * ksteps_total + (wave & 3)) * 64 + lane;
    char* sBw = sB + wave * 1024 + lane * 16;
#define HD_DEEP_LOAD(chunk)                                                                               \
    {                                                                                                      \
        _Pragma("unroll") for (int u = 0; u < UNITS; ++u) LD::fetch_nc(p, st[u], (chunk) * BK, kq, raw[(chunk) % P][u]); \
        braw[(chunk) % P] = *reinterpret_cast<const dq_u32x4*>(Wf + (size_t)(chunk) * 4 * 64);            \
    }
#define HD_DEEP_WRITE(chunk)                                                                              \
    {                                                                                                      \
        typename LD::ChunkC cc;                                                                            \
        LD::chunk_consts(p, st[0], (chunk) * BK, kq, cc);                                                  \
        _Pragma("unroll") for (int u = 0; u < UNITS; ++u)                                                  \
            lds_write_unit<LD::kSplit>(smem + ((chunk) & 1) * A_BUF + u_ldsoff[u], kq,                     \
                                       LD::finish_nc(p, st[u], (chunk) * BK, kq, raw[(chunk) % P][u], cc)); \
        *reinterpret_cast<dq_u32x4*>(sBw + ((chunk) & 1) * B_BUF) = braw[(chunk) % P];                     \
    }
    HD_STAMP(0);
#pragma unroll
    for (int d = 0; d < P; ++d) HD_DEEP_LOAD(d);
    HD_STAMP(1);
    // fused depthwise epilogue (EpDwGate): its per-channel constants are requested now, not behind the K loop
    float dw_w9[9], dw_bias = 0.f, dw_b1 = 0.f;
    if constexpr (EP::kTile) {
        const int ncols_e = p.N >> 1, ce = tile0 * 32 + (tid & 31) + ((tid >> 5) & 1) * ncols_e;      // odd half-waves: the second gate half's channel
#pragma unroll
        for (int t = 0; t < 9; ++t) dw_w9[t] = p.dw_w[(size_t)t * p.N + ce];
        dw_bias = p.dw_b[ce];
        dw_b1 = p.bias[(half ? tile1 : tile0) * 32 + (lane & 31)];     // conv1 bias of THIS wave's accumulator columns
    }
    LD::template block_init<BM, THREADS>(p, row0, smem + STATS_OFF, gb, tid);
#pragma unroll
    for (int u = 0; u < UNITS; ++u) LD::unit_stats(st[u], (tid >> 3) + u * (THREADS / 8), smem + STATS_OFF);
    HD_DEEP_WRITE(0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    HD_STAMP(2);

    const int a_lane_off = (lane & 31) * LDS_ROW + (lane >> 5) * 16;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        if (c + P < NCH) {
            HD_DEEP_LOAD(c + P);
            asm volatile("" ::: "memory");                              // keep the requests here (see gemm_deep_kernel)
        }
        const char* sA = smem + (c & 1) * A_BUF + a_lane_off + (w_m * 32) * LDS_ROW;
        const char* sBc = sB + (c & 1) * B_BUF + (half * 4) * 1024 + lane * 16;
#if HD_DEEP_SKEW
        // the two waves of a SIMD (w and w + 4: the two gate halves) take the chunk's two jobs in opposite order -- one runs its four
        // MFMAs while the other transforms and stores the next chunk -- instead of both queueing for the matrix pipe and then both for the VALU
        dq_u32x4 fa[4], fb[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) { fa[s4] = *reinterpret_cast<const dq_u32x4*>(sA + s4 * 32); fb[s4] = *reinterpret_cast<const dq_u32x4*>(sBc + s4 * 1024); }
        if (half) {
            if (c + 1 < NCH) HD_DEEP_WRITE(c + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fa[s4]), __builtin_bit_cast(bf16x8_t, fb[s4]), acc, 0, 0, 0);
        } else {
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fa[s4]), __builtin_bit_cast(bf16x8_t, fb[s4]), acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (c + 1 < NCH) HD_DEEP_WRITE(c + 1);
        }
#else
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const bf16x8_t a = *reinterpret_cast<const bf16x8_t*>(sA + s4 * 32);
            const bf16x8_t b = *reinterpret_cast<const bf16x8_t*>(sBc + s4 * 1024);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        }
        if (c + 1 < NCH) HD_DEEP_WRITE(c + 1);
#endif
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
#undef HD_DEEP_LOAD
#undef HD_DEEP_WRITE
    HD_STAMP(3); HD_STAMP(4);
    if constexpr (EP::kTile) {
        // ====== conv1 bias -> depthwise 3x3 -> SimpleGate -> G, pooled on the 128-row tile (the tile epilogue of gemm_skinny_kernel;
        // conditional_naf.py:116-119).  Both gate halves' T1 tiles go to LDS (the A buffers are dead: the loop ended with a barrier):
        // t1[half][128 rows][32] fp32 = 32 KB, then one work item = (channel j, image row) as there.  With 32-row skinny tiles this
        // GEMM put 1024 workgroups of 128 + 64 KB of operands on the chip (768 KB per CU: 21.8 - 23.6 us at latent 32, levels 3 / 2);
        // on 128-row tiles it is the conv4 GEMM (16.4 - 16.8 us) plus this epilogue.
        float* t1 = reinterpret_cast<float*>(smem);
        float* rs = reinterpret_cast<float*>(smem + 2 * 128 * 32 * 4);     // [128 / S][32] row sums (<= 8 KB: behind the T1 tiles, over dead statistics / FiLM rows)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int r = w_m * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
            t1[(half * 128 + r) * 32 + (lane & 31)] = acc[i] + dw_b1;
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const int C2 = p.N >> 1, ncols = C2;
        const int j = tid & 31, col = tile0 * 32 + j;
        const int S = p.side, ls = 31 - __builtin_clz(S), HW = p.hw;
        // every thread holds the taps of ONE half (even half-waves: a, odd: b); the work item needs both: the partner's through LDS
        float* wx = rs + (128 >> ls) * 32;                                  // [2][10][32] taps + bias of both halves
        {
            const int hb = (tid >> 5) & 1;
            if ((tid >> 6) == 0) {                                           // wave 0: half-wave 0 holds half a, half-wave 1 half b
#pragma unroll
                for (int t = 0; t < 9; ++t) wx[(hb * 10 + t) * 32 + j] = dw_w9[t];
                wx[(hb * 10 + 9) * 32 + j] = dw_bias;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        float wa[9], wb[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) { wa[t] = wx[t * 32 + j]; wb[t] = wx[(10 + t) * 32 + j]; }
        const float ba = wx[9 * 32 + j], bb = wx[19 * 32 + j];
        const float* t1a = t1 + j;
        const float* t1b = t1 + 128 * 32 + j;
        const int nrows_img = 128 >> ls;
        for (int rr = tid >> 5; rr < nrows_img; rr += THREADS / 32) {
            const int p0 = rr << ls;
            const int y = (p0 & (HW - 1)) >> ls;
            const bool up = y > 0, dn = y < S - 1;
            const int row = row0 + p0;
            unsigned short* gout = reinterpret_cast<unsigned short*>(p.out) + (size_t)row * p.ldo + col;
            const int left = p.M - row;
            float rsum;
            switch (S) {
                case 16: rsum = dw_gate_row<16>(t1a, t1b, p0, up, dn, wa, wb, ba, bb, gout, p.ldo, true, left); break;
                case 8: rsum = dw_gate_row<8>(t1a, t1b, p0, up, dn, wa, wb, ba, bb, gout, p.ldo, true, left); break;
                case 4: rsum = dw_gate_row<4>(t1a, t1b, p0, up, dn, wa, wb, ba, bb, gout, p.ldo, true, left); break;
                default: rsum = dw_gate_row<2>(t1a, t1b, p0, up, dn, wa, wb, ba, bb, gout, p.ldo, true, left); break;
            }
            rs[rr * 32 + j] = rsum;
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const int faces = 128 / HW;
        for (int idx = tid; idx < faces * 32; idx += THREADS) {
            const int f = idx >> 5;
            float sacc = 0.f;
            for (int r = 0; r < S; ++r) sacc += rs[(f * S + r) * 32 + j];
            const int face = row0 / HW + f;
            if (face * HW < p.M) {
                const float pm = sacc / (float)HW;
                p.pooled[(size_t)face * C2 + col] = pm;
                if (p.pooled16) p.pooled16[(size_t)face * C2 + col] = f32_to_bf16_bits(pm);
            }
        }
        (void)ncols;
        return;
    }
    // second gate half -> LDS (the A buffers are dead: the loop ended with a barrier), first half runs the pair epilogue
    float* xch = reinterpret_cast<float*>(smem);
    if (half) {
#pragma unroll
        for (int i = 0; i < 16; ++i) xch[(w_m * 16 + i) * 64 + lane] = acc[i];
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (!half) {
        f32x16_t acc2;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc2[i] = xch[(w_m * 16 + i) * 64 + lane];
        const int ncols = p.N >> 1;
        const int rtile = row0 + w_m * 32;
        const int rbase = rtile + 4 * (lane >> 5);
        const int col = tile0 * 32 + (lane & 31);
        if (rtile + 32 <= p.M) tile_epilogue_mfma<true, true, EP>(p, acc, acc2, rbase, col, ncols, tile0, lane);
        else tile_epilogue_mfma<false, true, EP>(p, acc, acc2, rbase, col, ncols, tile0, lane);
    }
    HD_STAMP(5);
}

template <class EP, class = void> struct ep_is_sca_tile { static constexpr bool value = false; };
template <class EP> struct ep_is_sca_tile<EP, decltype((void)EP::kScaTile)> { static constexpr bool value = EP::kScaTile; };

// Touch this workgroup's share of the next GEMM's weight tiles so that they are in this XCD's L2 when that
// kernel starts (its per-CU ingest is bound by the latency of the source: HBM ~2 us vs L2 ~0.6 us).  The loads
// all target one 4-VGPR sink that stays reserved until prefetch_drain() at the end of the kernel; loads return
// in order, so the extra (compiler-invisible) loads only ever make the compiler's own vmcnt waits stricter.
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void prefetch_issue(const GemmP& p, int lin, int nwg, int tid, int nthreads, u32x4_t& sink) {
    if (!p.pf_base) return;
    const int x = lin & 7, r = lin >> 3, per_x = nwg >> 3;                 // nwg % 8 == 0 (caller)
    const int tiles_x = (p.pf_ntiles - x + 7) >> 3;                          // tiles t = x, x+8, ...
    if (tiles_x <= 0 || r >= per_x) return;
    const unsigned total = (unsigned)tiles_x * p.pf_tile_u4;                 // uint4 units on this XCD (< 2^32: host)
    const unsigned share = (total + per_x - 1) / per_x;
    const unsigned lo = share * r, hi = (lo + share < total) ? lo + share : total;
    for (unsigned i = lo + tid; i < hi; i += nthreads) {
        const unsigned t = i / p.pf_tile_u4, o = i - t * p.pf_tile_u4;
        const uint4* q = p.pf_base + ((size_t)(x + 8 * t) * p.pf_tile_u4 + o);
        asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(sink) : "v"(q) : "memory");
    }
}
__device__ __forceinline__ void prefetch_drain(u32x4_t& sink) {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(sink) : : "memory");
}

typedef unsigned nt_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 nt_load_u4(const uint4* q) {
    const nt_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const nt_u32x4*>(q));
    return make_uint4(v.x, v.y, v.z, v.w);
}
// ----------------------------------------------------------------------------------- skinny kernel
// HALF: 16-row workgroup tiles (WM = MT = 1 only): at M = 64 this doubles the workgroups (4 row groups instead
// of 2) so all 256 CUs pull weights; rows 16..31 of the MFMA tile stay zero in LDS.
template <int WM_, int WK_, int MT_, bool PAIR_, int D_, bool HALF_ = false>
struct SkinnyCfg {
    static constexpr int WM = WM_, WK = WK_, MT = MT_, D = D_;
    static constexpr bool HALF = HALF_;
    static_assert(!HALF_ || (WM_ == 1 && MT_ == 1), "HALF tiles are single-row-tile workgroups");
    static constexpr int WROWS = HALF_ ? 16 : MT_ * 32;          // rows of one wave's sub-tile that hold data
    static constexpr int WAVES = WM * WK, THREADS = 64 * WAVES, BM = WM * WROWS, TNT = PAIR_ ? 2 : 1;
    static constexpr int UN = WROWS / 8;                         // (row, 8 k) units per lane per chunk
    static constexpr int A_WAVE = MT * 32 * LDS_ROW;             // private staging tile of one wave
    static constexpr int RED = WK * BM * 32 * TNT * 4;           // [wk][tn][row][32] partial tiles (aliases staging)
    static constexpr int STAGE = (WAVES * A_WAVE > RED) ? WAVES * A_WAVE : RED;
    static constexpr int GT_OFF = STAGE;                         // EpDwGate: gate tile [BM][32] fp32
    static constexpr int STATS_OFF = GT_OFF + BM * 32 * 4;
    static constexpr int GB_OFF = STATS_OFF + BM * 8;            // + 2*Kp floats of gain/bias for the LN loader
};

// Depthwise 3x3 + SimpleGate over one image row of S pixels for channel j (both gate halves), window
// sliding along x.  t1a/t1b point at column j of the two T1 half-tiles ([row][32] floats).  Returns the row
// sum of the unrounded gate; stores G as bf16.
template <int S>
__device__ __forceinline__ float dw_gate_row(const float* t1a, const float* t1b, int p0, bool up, bool dn, const float* wa,
                                             const float* wb, float ba, float bb, unsigned short* gout, int ldo, bool store_ok,
                                             int rows_left) {
    float va[3][S + 2], vb[3][S + 2];                                    // 3 image rows x (S + 2) columns, zero padded
#pragma unroll
    for (int r = 0; r < 3; ++r) { va[r][0] = 0.f; va[r][S + 1] = 0.f; vb[r][0] = 0.f; vb[r][S + 1] = 0.f; }
#pragma unroll
    for (int x = 0; x < S; ++x) {
        va[0][x + 1] = up ? t1a[(p0 + x - S) * 32] : 0.f; vb[0][x + 1] = up ? t1b[(p0 + x - S) * 32] : 0.f;
        va[1][x + 1] = t1a[(p0 + x) * 32];                vb[1][x + 1] = t1b[(p0 + x) * 32];
        va[2][x + 1] = dn ? t1a[(p0 + x + S) * 32] : 0.f; vb[2][x + 1] = dn ? t1b[(p0 + x + S) * 32] : 0.f;
    }
    float rsum = 0.f;
#pragma unroll
    for (int x = 0; x < S; ++x) {
        float u1 = ba, u2 = bb;
#pragma unroll
        for (int r = 0; r < 3; ++r) {                                        // nine taps, one fused multiply-add each, in tap order
            u1 = fmaf(wa[r * 3], va[r][x], u1); u1 = fmaf(wa[r * 3 + 1], va[r][x + 1], u1); u1 = fmaf(wa[r * 3 + 2], va[r][x + 2], u1);
            u2 = fmaf(wb[r * 3], vb[r][x], u2); u2 = fmaf(wb[r * 3 + 1], vb[r][x + 1], u2); u2 = fmaf(wb[r * 3 + 2], vb[r][x + 2], u2);
        }
        const float g = u1 * u2;
        rsum += g;
        if (store_ok && x < rows_left) gout[(size_t)x * ldo] = f32_to_bf16_bits(g);
    }
    return rsum;
}

// What the element-wise epilogue reads from global memory (per-column constants, residual / gate operands): requested
// before the K loop so that the epilogue does not start with a dependent round trip.
template <class C> struct SkinnyPre { static constexpr int NIT = (C::BM * 32 + C::THREADS - 1) / C::THREADS; ColC cc; float pre[NIT]; };
template <bool FULL, class C, class EP>
__device__ __forceinline__ void skinny_rows_pre(const GemmP& p, int row0, int col, int ncols, int tid, SkinnyPre<C>& sp) {
    constexpr int NIT = SkinnyPre<C>::NIT;
    constexpr bool EVEN = (C::BM * 32) % C::THREADS == 0;
    const bool cv = FULL || col < ncols;
    sp.cc = EP::col_init(p, cv ? col : 0);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int e = it * C::THREADS + tid;
        const int row = row0 + (e >> 5);
        const bool ok = (EVEN || e < C::BM * 32) && (FULL || (row < p.M && cv));
        sp.pre[it] = ok ? EP::pre(p, row, col) : 0.f;
    }
}
template <bool FULL, class C, class EP>
__device__ __forceinline__ void skinny_rows_epilogue(const GemmP& p, const float* red, int row0, int col, int ncols, int tile_idx, int tid,
                                                     const SkinnyPre<C>& sp) {
    constexpr int NIT = (C::BM * 32 + C::THREADS - 1) / C::THREADS, TNT = C::TNT, WK = C::WK;
    constexpr int TILE_F = C::BM * 32 * TNT;
    constexpr bool EVEN = (C::BM * 32) % C::THREADS == 0;
    const bool cv = FULL || col < ncols;
    const ColC cc = sp.cc;
    const float (&pre)[NIT] = sp.pre;
    float v[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int e = it * C::THREADS + tid;
        const int row = row0 + (e >> 5);
        const bool ev = EVEN || e < C::BM * 32;
        float v1 = 0.f, v2 = 0.f;
        if (ev) {
#pragma unroll
            for (int w = 0; w < WK; ++w) {
                v1 += red[w * TILE_F + e];
                if (TNT == 2) v2 += red[w * TILE_F + C::BM * 32 + e];
            }
        }
        v[it] = 0.f;
        if (ev && (FULL || (row < p.M && cv))) {
            if constexpr (TNT == 2) EP::store2(p, row, col, v1, v2, cc);
            else v[it] = EP::store(p, row, col, v1, pre[it], cc);
        }
    }
    if constexpr (EP::kStats) {
        if (p.stats_out) {
            float2 ms[NIT];
#pragma unroll
            for (int it = 0; it < NIT; ++it) ms[it] = halfwave_mean_m2(v[it]);
            if ((tid & 31) == kStatLane) {
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const int e = it * C::THREADS + tid;
                    const int row = row0 + (e >> 5);
                    if ((EVEN || e < C::BM * 32) && (FULL || row < p.M)) p.stats_out[EP::stat_index(p, row, tile_idx)] = ms[it];
                }
            }
        }
    }
}

// CPW > 0: the number of 64-deep chunks per wave is a compile-time constant (host: Kp / 64 / WK == CPW, K == Kp) and the
// K loop is straight-line code whose loads are not under any branch.  With a run-time trip count every prefetch sits
// behind a bounds test, the compiler can then no longer tell how many loads are in flight at the top of a chunk and
// drains all of them (s_waitcnt vmcnt(0)): each chunk waits for the prefetch issued one chunk earlier, i.e. a full
// memory round trip per chunk instead of D chunks in flight.
template <class C, class LD, class EP, bool W_NT = false, int CPW = 0>
__global__ __launch_bounds__(C::THREADS) void gemm_skinny_kernel(const GemmP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int MT = C::MT, TNT = C::TNT, WK = C::WK, D = C::D, UN = C::UN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WK, wk = wave - wm * WK;
    // XCD-aware block -> (row group, weight tile) map: workgroups are dealt round-robin over the 8 XCDs
    // (private L2s), so all row groups that stream the same weight tile are given ids that are equal mod 8
    // and adjacent in dispatch order: the tile is fetched from HBM once and re-read from that XCD's L2.
    // Placement only affects speed, never results.
    int bx = blockIdx.x, by = blockIdx.y;
    if (p.xcd_tile_affine && (gridDim.y & 7) == 0 && gridDim.x > 1) {
        const int lin = blockIdx.y * gridDim.x + blockIdx.x;
        const int j = lin >> 3;
        bx = j % (int)gridDim.x;
        by = (j / (int)gridDim.x) * 8 + (lin & 7);
    }
    const int row0 = bx * C::BM;
    const int ksteps_total = p.Kp >> 4;
    const int cpw = CPW > 0 ? CPW : (p.Kp >> 6) / WK;        // host: Kp % (64*WK) == 0
    const int c0 = wk * cpw, c_end = c0 + cpw;
    int tile[TNT];
    tile[0] = by;
    if (TNT == 2) tile[1] = by + (p.N >> 6);
    const uint4* Wl = p.W + lane;
    // weight tiles that only a handful of row groups share are streamed with the non-temporal policy (they are not
    // needed again this step; measured: middle level -2.4 us per block, level 3 -1 us); tiles shared by many row
    // groups (level 2: 32) must stay in L2 and are loaded normally
    constexpr bool w_nt = W_NT;                                // compile-time: p.w_nt picks the instantiation (launch_skinny)
    HD_STAMP(0);

    f32x16_t acc[MT][TNT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int tn = 0; tn < TNT; ++tn)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][tn][i] = 0.f;
    uint4 bq[D][4][TNT];
    typename LD::Raw aq[D][UN];
#define HD_SK_LOAD_B(slot, chunk)                                                                      \
    _Pragma("unroll") for (int ss = 0; ss < 4; ++ss) _Pragma("unroll") for (int tn = 0; tn < TNT; ++tn) \
        bq[slot][ss][tn] = ((chunk) < c_end) ? (w_nt ? nt_load_u4(&Wl[((size_t)tile[tn] * ksteps_total + (chunk) * 4 + ss) * 64]) : Wl[((size_t)tile[tn] * ksteps_total + (chunk) * 4 + ss) * 64]) : make_uint4(0, 0, 0, 0);
    // the block prologue's small loads (LayerNorm partials, FiLM row) go out first, then the weights: neither
    // depends on anything this kernel computes, and loads return in issue order
    float* gb = LD::kGainBiasLds ? reinterpret_cast<float*>(smem + C::GB_OFF) : nullptr;
    typename LD::Pre pre;
    LD::template block_issue<C::BM, C::THREADS>(p, row0, gb, tid, pre);
#define HD_SK_LOAD_B_NC(slot, chunk)                                                                   \
    _Pragma("unroll") for (int ss = 0; ss < 4; ++ss) _Pragma("unroll") for (int tn = 0; tn < TNT; ++tn) \
        bq[slot][ss][tn] = w_nt ? nt_load_u4(&Wl[((size_t)tile[tn] * ksteps_total + (chunk) * 4 + ss) * 64]) : Wl[((size_t)tile[tn] * ksteps_total + (chunk) * 4 + ss) * 64];
    if constexpr (CPW == 0) {
#pragma unroll
        for (int d = 0; d < D; ++d) { HD_SK_LOAD_B(d, c0 + d); }
    }

    typename LD::St st[UN];
    int u_off[UN];
    const int kq = lane & 7;
    // row pointers first (the statistics part of St is filled in after block_init): the A loads do not
    // depend on the LayerNorm statistics, only finish() does
#pragma unroll
    for (int u = 0; u < UN; ++u) {
        const int rw = (lane >> 3) + 8 * u;                      // row inside the wave's sub-tile
        const int rl = wm * C::WROWS + rw;
        u_off[u] = rw * LDS_ROW;
        LD::unit_init(p, st[u], row0 + rl, rl, smem + C::STATS_OFF, gb);
    }
#define HD_SK_FETCH_A(slot, chunk)                                                                     \
    _Pragma("unroll") for (int u = 0; u < UN; ++u)                                                     \
        LD::fetch(p, st[u], ((chunk) < c_end) ? (chunk) * BK : p.Kp, kq, aq[slot][u]);
#define HD_SK_FETCH_A_NC(slot, chunk) \
    _Pragma("unroll") for (int u = 0; u < UN; ++u) LD::fetch_nc(p, st[u], (chunk) * BK, kq, aq[slot][u]);
    if constexpr (CPW > 0) {                                   // chunk by chunk: loads return in issue order, chunk 0 is needed first
#pragma unroll
        for (int d = 0; d < (D < CPW ? D : CPW); ++d) { HD_SK_LOAD_B_NC(d, c0 + d); HD_SK_FETCH_A_NC(d, c0 + d); }
    } else {
#pragma unroll
        for (int d = 0; d < D; ++d) { HD_SK_FETCH_A(d, c0 + d); }
    }
    // fused depthwise epilogue: its per-channel weights (tap-major copy: coalesced) are requested now, not after the
    // K loop (a dependent global round trip inside the epilogue otherwise)
    // (only in the 8-wave shapes: in the 4-wave shapes, two workgroups per CU, the extra live registers cost more)
    constexpr bool kEarlyDw = EP::kTile && C::THREADS >= 512;
    float dw_wa[9], dw_wb[9], dw_ba = 0.f, dw_bb = 0.f, dw_b1a = 0.f, dw_b1b = 0.f;
    auto load_dw = [&]() {
        const int ncols_e = p.N >> 1, col_e = tile[0] * 32 + (tid & 31);
        const bool cok = col_e < ncols_e;
#pragma unroll
        for (int t = 0; t < 9; ++t) { dw_wa[t] = cok ? p.dw_w[(size_t)t * p.N + col_e] : 0.f; dw_wb[t] = cok ? p.dw_w[(size_t)t * p.N + col_e + ncols_e] : 0.f; }
        if (cok) { dw_ba = p.dw_b[col_e]; dw_bb = p.dw_b[col_e + ncols_e]; dw_b1a = p.bias[col_e]; dw_b1b = p.bias[col_e + ncols_e]; }
    };
    if constexpr (kEarlyDw) load_dw();
    // SCA tile epilogue: the G rows it rescales in place are requested now as well (first two 16-byte units per thread)
    uint4 sca_g[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
    if constexpr (ep_is_sca_tile<EP>::value) {
        const int hw = p.scale_hw, units = C::BM * hw * 4;
        if (hw > 0 && tile[0] * 32 + 32 <= p.N) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int u = tid + j * C::THREADS;
                const int f = u / (hw * 4), rem = u - f * hw * 4, r = rem >> 2, q = rem & 3;
                if (u < units && row0 + f < p.M)
                    sca_g[j] = *reinterpret_cast<const uint4*>(p.scale_G + ((size_t)(row0 + f) * hw + r) * p.ldo + tile[0] * 32 + q * 8);
            }
        }
    }
    // element-wise epilogues: same idea for their residual / gate operands and per-column constants (8-wave shapes)
    constexpr bool kEarlyPre = !EP::kTile && !ep_is_sca_tile<EP>::value && C::THREADS >= 512;
    SkinnyPre<C> spre;
    if constexpr (kEarlyPre) {
        const int ncols_e = (TNT == 2) ? (p.N >> 1) : p.N, col_e = tile[0] * 32 + (tid & 31);
        const bool full_e = (row0 + C::BM <= p.M) && (tile[0] * 32 + 32 <= ncols_e);
        if (full_e) skinny_rows_pre<true, C, EP>(p, row0, col_e, ncols_e, tid, spre);
        else skinny_rows_pre<false, C, EP>(p, row0, col_e, ncols_e, tid, spre);
    }
    u32x4_t pf_sink = {0u, 0u, 0u, 0u};
    prefetch_issue(p, (int)(blockIdx.y * gridDim.x + blockIdx.x), (int)(gridDim.x * gridDim.y), tid, C::THREADS, pf_sink);
    HD_STAMP(6);
    LD::template block_finish<C::BM, C::THREADS>(p, row0, smem + C::STATS_OFF, gb, tid, pre);
#pragma unroll
    for (int u = 0; u < UN; ++u) LD::unit_stats(st[u], wm * C::WROWS + (lane >> 3) + 8 * u, smem + C::STATS_OFF);
    HD_STAMP(1);

    char* sA = smem + wave * C::A_WAVE;
    const int a_lane_off = (lane & 31) * LDS_ROW + (lane >> 5) * 16;
    if constexpr (C::HALF) {                                   // rows 16..31 are never staged: keep them zero
        for (int i = lane; i < 16 * LDS_ROW / 16; i += 64) reinterpret_cast<uint4*>(sA + 16 * LDS_ROW)[i] = make_uint4(0, 0, 0, 0);
        __builtin_amdgcn_wave_barrier();
    }
    auto chunk_mma = [&](int d, int cc) {
        if constexpr (CPW > 0) {
            typename LD::ChunkC chunk_c;
            LD::chunk_consts(p, st[0], cc * BK, kq, chunk_c);
#pragma unroll
            for (int u = 0; u < UN; ++u) lds_write_unit<LD::kSplit>(sA + u_off[u], kq, LD::finish_nc(p, st[u], cc * BK, kq, aq[d][u], chunk_c));
        } else {
#pragma unroll
            for (int u = 0; u < UN; ++u) lds_write_unit<LD::kSplit>(sA + u_off[u], kq, LD::finish(p, st[u], cc * BK, kq, aq[d][u]));
        }
        __builtin_amdgcn_wave_barrier();                           // LDS ops of one wave execute in order
#pragma unroll
        for (int ss = 0; ss < 4; ++ss) {
            bf16x8_t a[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                a[mt] = *reinterpret_cast<const bf16x8_t*>(sA + a_lane_off + (mt * 32) * LDS_ROW + ss * 32);
#pragma unroll
            for (int tn = 0; tn < TNT; ++tn)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc[mt][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                        a[mt], __builtin_bit_cast(bf16x8_t, bq[d][ss][tn]), acc[mt][tn], 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
    };
    if constexpr (CPW > 0) {
#pragma unroll
        for (int i = 0; i < CPW; ++i) {
            const int d = i % D, cc = c0 + i;
            chunk_mma(d, cc);
            if (i + D < CPW) { HD_SK_LOAD_B_NC(d, cc + D); HD_SK_FETCH_A_NC(d, cc + D); }
            if (i == 0) HD_STAMP(2);
        }
    } else {
        for (int cb = c0; cb < c_end; cb += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const int cc = cb + d;
                if (cc < c_end) {                                  // wave-uniform
                    chunk_mma(d, cc);
                    HD_SK_LOAD_B(d, cc + D);
                    HD_SK_FETCH_A(d, cc + D);
                    if (cc == c0) HD_STAMP(2);
                }
            }
        }
    }
#undef HD_SK_LOAD_B
#undef HD_SK_FETCH_A
#undef HD_SK_LOAD_B_NC
#undef HD_SK_FETCH_A_NC
    HD_STAMP(3);

    // ---- partial tiles to LDS: red[wk][tn][row in BM][32] ----
    __syncthreads();                                           // staging tiles are dead: reuse as reduction buffer
    HD_STAMP(4);
    float* red = reinterpret_cast<float*>(smem);
    constexpr int TILE_F = C::BM * 32 * TNT;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int tn = 0; tn < TNT; ++tn)
#pragma unroll
            for (int i = 0; i < (C::HALF ? 8 : 16); ++i) {
                const int r = (wm * MT + mt) * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
                red[wk * TILE_F + (tn * C::BM + r) * 32 + (lane & 31)] = acc[mt][tn][i];
            }
    __syncthreads();
    const int ncols = (TNT == 2) ? (p.N >> 1) : p.N;
    const int col = tile[0] * 32 + (tid & 31);

    if constexpr (EP::kTile) {
        // ================= conv1 bias -> depthwise 3x3 -> SimpleGate -> G, pooled =================
        static_assert(TNT == 2, "EpDwGate needs a PAIR tile");
        const int C2 = p.N >> 1;
        if constexpr (!kEarlyDw) load_dw();
        // (1) sum the K-split partials in wave order, add conv1's bias, keep T1 in slice 0
        //     (rows beyond M hold zeros from the masked A loads; their outputs are never stored)
        {
            const float bias_a = dw_b1a, bias_b = dw_b1b;
            for (int e = tid; e < C::BM * 32; e += C::THREADS) {
                float va = bias_a, vb = bias_b;
#pragma unroll
                for (int w = 0; w < WK; ++w) { va += red[w * TILE_F + e]; vb += red[w * TILE_F + C::BM * 32 + e]; }
                red[e] = va; red[C::BM * 32 + e] = vb;
            }
        }
        __syncthreads();
        // (2) depthwise 3x3 (pad 1) on both halves from LDS, SimpleGate, store G (bf16).  One work item =
        //     (channel j, image row): all 3 x (S+2) window values are loaded up front (face side is a
        //     compile-time power of two), so the LDS latency is paid once per row, not once per pixel.
        float* rs = reinterpret_cast<float*>(smem + C::GT_OFF);          // [BM / S][32] row sums
        const int j = tid & 31;
        const int S = p.side, ls = 31 - __builtin_clz(S), HW = p.hw;     // S in {1,2,4,8,16}
        const float (&wa)[9] = dw_wa; const float (&wb)[9] = dw_wb;
        const float ba = dw_ba, bb = dw_bb;
        const float* t1a = red + j;
        const float* t1b = red + C::BM * 32 + j;
        const int nrows_img = C::BM >> ls;                               // image rows in the tile
        for (int rr = tid >> 5; rr < nrows_img; rr += C::THREADS / 32) {
            const int p0 = rr << ls;                                     // first pixel (tile-local row) of this image row
            const int y = (p0 & (HW - 1)) >> ls;
            const bool up = y > 0, dn = y < S - 1;
            const int row = row0 + p0;
            unsigned short* gout = reinterpret_cast<unsigned short*>(p.out) + (size_t)row * p.ldo + col;
            const bool ok = col < ncols;
            const int left = p.M - row;                                  // valid pixels from here on
            float rsum;
            switch (S) {
                case 16: rsum = dw_gate_row<16>(t1a, t1b, p0, up, dn, wa, wb, ba, bb, gout, p.ldo, ok, left); break;
                case 8: rsum = dw_gate_row<8>(t1a, t1b, p0, up, dn, wa, wb, ba, bb, gout, p.ldo, ok, left); break;
                case 4: rsum = dw_gate_row<4>(t1a, t1b, p0, up, dn, wa, wb, ba, bb, gout, p.ldo, ok, left); break;
                case 2: rsum = dw_gate_row<2>(t1a, t1b, p0, up, dn, wa, wb, ba, bb, gout, p.ldo, ok, left); break;
                default: rsum = dw_gate_row<1>(t1a, t1b, p0, up, dn, wa, wb, ba, bb, gout, p.ldo, ok, left); break;
            }
            rs[rr * 32 + j] = rsum;
        }
        __syncthreads();
        // (3) per-face average pool (SCA input): S row sums per face
        const int faces = C::BM / HW;
        for (int idx = tid; idx < faces * 32; idx += C::THREADS) {
            const int f = idx >> 5;
            float sacc = 0.f;
            for (int r = 0; r < S; ++r) sacc += rs[(f * S + r) * 32 + j];
            const int face = row0 / HW + f;
            if (face * HW < p.M && col < ncols) {
                const float pm = sacc / (float)HW;
                p.pooled[(size_t)face * C2 + col] = pm;
                if (p.pooled16) p.pooled16[(size_t)face * C2 + col] = f32_to_bf16_bits(pm);
            }
        }
    } else if constexpr (ep_is_sca_tile<EP>::value) {
        // ================= SCA: s tile -> LDS and S; then scale the faces' rows of G in place =================
        {
            const float bias = col < ncols ? p.bias[col] : 0.f;
            for (int e = tid; e < C::BM * 32; e += C::THREADS) {
                float v = bias;
#pragma unroll
                for (int w = 0; w < WK; ++w) v += red[w * TILE_F + e];
                const int face = row0 + (e >> 5);
                if (face < p.M && col < ncols) reinterpret_cast<float*>(p.out)[(size_t)face * p.ldo + col] = v;
                red[e] = v;                                        // e is owned by this thread in every slice
            }
        }
        __syncthreads();
        const int hw = p.scale_hw;
        if (hw > 0 && tile[0] * 32 + 32 <= ncols) {
            const int units = C::BM * hw * 4;                      // (face, pixel, 8 columns)
            for (int u0 = tid; u0 < units; u0 += C::THREADS * 2) {
                uint4 g[2]; unsigned short* gp[2]; const float* sp[2]; bool ok[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int u = u0 + j * C::THREADS;
                    const int f = u / (hw * 4), rem = u - f * hw * 4, r = rem >> 2, q = rem & 3;
                    const int face = row0 + f;
                    ok[j] = u < units && face < p.M;
                    gp[j] = p.scale_G + ((size_t)(ok[j] ? face : 0) * hw + r) * p.ldo + tile[0] * 32 + q * 8;
                    sp[j] = red + f * 32 + q * 8;
                    g[j] = (u0 == tid) ? sca_g[j]                     // first pass: fetched before the K loop
                                       : (ok[j] ? *reinterpret_cast<const uint4*>(gp[j]) : make_uint4(0, 0, 0, 0));
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (ok[j]) {
                        float v[8];
                        unpack8(g[j], v);
#pragma unroll
                        for (int i = 0; i < 8; ++i) v[i] *= sp[j][i];
                        *reinterpret_cast<uint4*>(gp[j]) = pack8(v);
                    }
                }
            }
        }
    } else {
        // ================= element-wise epilogue, 32 lanes = one row of the tile =================
        const bool full = (row0 + C::BM <= p.M) && (tile[0] * 32 + 32 <= ncols);      // workgroup-uniform
        if constexpr (!kEarlyPre) {
            if (full) skinny_rows_pre<true, C, EP>(p, row0, col, ncols, tid, spre);
            else skinny_rows_pre<false, C, EP>(p, row0, col, ncols, tid, spre);
        }
        if (full) skinny_rows_epilogue<true, C, EP>(p, red, row0, col, ncols, tile[0], tid, spre);
        else skinny_rows_epilogue<false, C, EP>(p, red, row0, col, ncols, tile[0], tid, spre);
    }
    if (p.pf_base) prefetch_drain(pf_sink);
    HD_STAMP(5);
}

// Tile shapes of the tall kernel.
//   T128: 4 waves stacked along M, each 32 rows x 64 cols.     T64: 2x2 waves, 64 rows x 64 cols.
//   T32W: 32 rows x (4 waves x 64 cols): the whole N of a level-0 GEMM in one workgroup.
typedef Cfg<4, 1, 1, 2, false> T128;
typedef Cfg<4, 1, 1, 1, true> T128P;
typedef Cfg<2, 2, 1, 1, false> T64;
typedef Cfg<2, 2, 1, 1, true> T64P;
typedef Cfg<1, 4, 1, 2, false> T32W;
typedef Cfg<1, 4, 1, 1, true> T32WP;

template <class C, class LD, class EP>
inline hipError_t launch_gemm(const GemmP& p, hipStream_t s) {
    const int ncols = C::PAIR ? p.N / 2 : p.N;
    const int smem = C::SMEM + (LD::kGainBiasLds ? 2 * p.Kp * 4 : 0);
    if (smem > 65536) {                                      // above the default dynamic-LDS limit
        static std::atomic<unsigned long long> granted{0};
        { const hipError_t e = grant_dynamic_lds(reinterpret_cast<const void*>(&gemm_kernel<C, LD, EP>), 160 * 1024, granted); if (e != hipSuccess) return e; }
    }
    dim3 grid((p.M + C::BM - 1) / C::BM, (ncols + C::NCOLS - 1) / C::NCOLS, 1);
    hipLaunchKernelGGL((gemm_kernel<C, LD, EP>), grid, dim3(C::THREADS), smem, s, p);
    return hipGetLastError();
}

// deep-prefetch tall kernel: 128-row tiles, 32 (gate) columns, K = 512 or 1024 (see gemm_deep_kernel); false when the shape is
// not one of its own (the caller then uses the mode it would have used otherwise)
template <bool PAIR>
inline bool deep_shape_ok(const GemmP& p) {
    const int ncols = PAIR ? p.N / 2 : p.N;
    // measured at latent 32 (profiles/r03_latent32_kernel_table.txt): pair GEMM 16.5 us against 17.6 us for the skinny kernel at
    // M = 1024 and 16.4 against 19.7 at M = 4096; plain GEMM 12.5 against 14.7 at M = 4096 but 13.8 against 13.0 at M = 1024
    return p.K == p.Kp && (p.Kp == 512 || p.Kp == 1024) && p.M >= (PAIR ? 1024 : 2048) && ncols % 32 == 0 && p.film_face_stride == 0 &&
           ((p.M + 127) / 128) * (ncols / 32) >= 256;
}
#ifndef HD_DEEP_P
#define HD_DEEP_P 4
#endif
template <class LD, class EP, bool PAIR>
inline hipError_t launch_gemm_deep(const GemmP& p, hipStream_t s) {
    typedef Cfg<4, 1, 1, 1, PAIR> C;
    constexpr int P = HD_DEEP_P;                          // chunks in flight per wave (A units + this wave's share of the B fragments)
    const int ncols = PAIR ? p.N / 2 : p.N;
    const int smem = C::SMEM + (LD::kGainBiasLds ? 2 * p.Kp * 4 : 0) + 2 * (PAIR ? 8 : 4) * 1024;     // + two B chunk buffers
    dim3 grid((p.M + C::BM - 1) / C::BM, ncols / 32, 1);
    if constexpr (PAIR) {                                  // eight waves: four row tiles x the two gate halves
        if (p.Kp == 1024) hipLaunchKernelGGL((gemm_deep_pair8_kernel<LD, EP, 16, P>), grid, dim3(512), smem, s, p);
        else hipLaunchKernelGGL((gemm_deep_pair8_kernel<LD, EP, 8, P>), grid, dim3(512), smem, s, p);
        return hipGetLastError();
    }
    if (p.Kp == 1024) hipLaunchKernelGGL((gemm_deep_kernel<C, LD, EP, 16, P>), grid, dim3(C::THREADS), smem, s, p);
    else hipLaunchKernelGGL((gemm_deep_kernel<C, LD, EP, 8, P>), grid, dim3(C::THREADS), smem, s, p);
    return hipGetLastError();
}

template <class C, class LD, class EP, bool NT, int CPW>
inline hipError_t launch_skinny_inst(const GemmP& p, hipStream_t s, int smem) {
    if (smem > 65536) {
        static std::atomic<unsigned long long> granted{0};
        { const hipError_t e = grant_dynamic_lds(reinterpret_cast<const void*>(&gemm_skinny_kernel<C, LD, EP, NT, CPW>), 160 * 1024, granted); if (e != hipSuccess) return e; }
    }
    const int ncols = (C::TNT == 2) ? p.N / 2 : p.N;
    dim3 grid((p.M + C::BM - 1) / C::BM, (ncols + 31) / 32, 1);
    hipLaunchKernelGGL((gemm_skinny_kernel<C, LD, EP, NT, CPW>), grid, dim3(C::THREADS), smem, s, p);
    return hipGetLastError();
}

template <class C, class LD, class EP>
inline hipError_t launch_skinny(const GemmP& p, hipStream_t s) {
    const int smem = C::GB_OFF + (LD::kGainBiasLds ? 2 * p.Kp * 4 : 0);
    if (smem > 160 * 1024 || (p.Kp / 64) % C::WK != 0) return hipErrorInvalidValue;
    const bool nt = p.w_nt != 0;
    // 2 or 4 chunks per wave, K a multiple of 64, whole weight tiles (every GEMM of the denoiser blocks from level 0 to
    // the middle): the straight-line K loop (gemm_skinny_kernel, CPW)
    if constexpr (LD::kStaticK) {
        static const bool no_static = hd_env("HD_NO_STATIC_K") != nullptr;
        const int cpw = (p.Kp / 64) / C::WK;
        const int ncols = (C::TNT == 2) ? p.N / 2 : p.N;
        const bool ok = !no_static && p.K == p.Kp && ncols % 32 == 0 && LD::static_ok(p);
        if (ok && (cpw == 2 || cpw == 4)) {
            if (cpw == 4) return nt ? launch_skinny_inst<C, LD, EP, true, 4>(p, s, smem) : launch_skinny_inst<C, LD, EP, false, 4>(p, s, smem);
            return nt ? launch_skinny_inst<C, LD, EP, true, 2>(p, s, smem) : launch_skinny_inst<C, LD, EP, false, 2>(p, s, smem);
        }
        if constexpr (LD::kStatic8 && C::WK == 8 && C::MT == 1) {
            if (ok && cpw == 8) return nt ? launch_skinny_inst<C, LD, EP, true, 8>(p, s, smem) : launch_skinny_inst<C, LD, EP, false, 8>(p, s, smem);
        }
    }
    return nt ? launch_skinny_inst<C, LD, EP, true, 0>(p, s, smem) : launch_skinny_inst<C, LD, EP, false, 0>(p, s, smem);
}

// Chunks in flight per wave from a register budget: accumulators + D x (B fragments + raw A units) must
// stay inside the 256 VGPRs a wave gets at <= 8 waves per workgroup.
template <int MT, bool PAIR, class LD>
struct ChunkDepth {
    static constexpr int TNT = PAIR ? 2 : 1;
    static constexpr int fixed = MT * TNT * 16 + 48 + MT * 4 * 6;
    static constexpr int per_d = TNT * 16 + MT * 4 * LD::kRawRegs;
    static constexpr int D = (fixed + 3 * per_d <= 236) ? 3 : ((fixed + 2 * per_d <= 236) ? 2 : 1);
};

// Skinny launch, K split over as many waves (<= 8/WM) as keep >= 2 chunks per wave.
template <int WM, int MT, bool PAIR, class LD, class EP>
inline hipError_t launch_skinny_auto(const GemmP& p, hipStream_t s) {
    constexpr int D = ChunkDepth<MT, PAIR, LD>::D;
    const int chunks = p.Kp / 64;
    if constexpr (WM == 1 && MT == 1) {
        // few rows, many columns: 16-row tiles double the workgroups so that every CU streams weights
        static const bool no_half = hd_env("HD_NO_HALF") != nullptr;
        const int tiles = (PAIR ? p.N / 2 : p.N) / 32;
        static const int half_m = hd_env("HD_HALF_M") ? atoi(hd_env("HD_HALF_M")) : 64;
        static const int half_wg = hd_env("HD_HALF_WG") ? atoi(hd_env("HD_HALF_WG")) : 256;
        if (!no_half && p.M <= half_m && p.M % 16 == 0 && ((p.M + 31) / 32) * tiles < half_wg) {
            if (chunks >= 16 && chunks % 8 == 0) return launch_skinny<SkinnyCfg<1, 8, 1, PAIR, D, true>, LD, EP>(p, s);
            if (chunks >= 8 && chunks % 4 == 0) return launch_skinny<SkinnyCfg<1, 4, 1, PAIR, D, true>, LD, EP>(p, s);
        }
    }
    if constexpr (WM <= 1) if (chunks >= 16 && chunks % 8 == 0) return launch_skinny<SkinnyCfg<WM, 8, MT, PAIR, D>, LD, EP>(p, s);
    if constexpr (WM <= 2) if (chunks >= 8 && chunks % 4 == 0) return launch_skinny<SkinnyCfg<WM, 4, MT, PAIR, D>, LD, EP>(p, s);
    if constexpr (WM <= 4) if (chunks >= 4 && chunks % 2 == 0) return launch_skinny<SkinnyCfg<WM, 2, MT, PAIR, D>, LD, EP>(p, s);
    return launch_skinny<SkinnyCfg<WM, 1, MT, PAIR, D>, LD, EP>(p, s);
}

}  // namespace hd

#pragma clang fp contract(fast)
