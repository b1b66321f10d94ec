// hd_wide.hpp — many-row GEMMs of latent 32, level 3 (M = 1024 rows, K = 1024): LayerNorm + FiLM -> conv4 -> SimpleGate and
// bf16 -> conv3 / conv5 -> residual (models/denoiser/conditional_naf.py:120-133), gfx950 only.
//
// The deep-prefetch tall kernel (hd_gemm.hpp: gemm_deep_kernel / gemm_deep_pair8_kernel) spends 0.63 us per 64-deep K chunk at
// these shapes whatever its prefetch depth (tools/deep_bench.hip): a chunk is one workgroup barrier around ~120 KB of LDS traffic for
// 32 MFMAs -- every A fragment read by two waves, every B fragment by four, the FiLM gain / bias by every thread for every unit.
// This form keeps the 128-row x 32-(gate-)column workgroup tile (256 workgroups at M = 1024: one per CU) and changes what
// happens inside it:
//   * 8 waves in two roles, one of each per SIMD: four STAGING waves (global loads HD_WIDE_P stages ahead, LayerNorm transform, LDS
//     stores) and four MFMA waves.  In the tall kernels every wave did both jobs of a chunk behind each other and all waves in
//     step (one barrier per chunk), so the matrix pipe idled during the transform and the VALU during the MFMAs: K loop =
//     ~5 us of LDS traffic and waits + 2 us of MFMA + 3 us of transform = 10 us at M = K = 1024, unmoved by prefetch depth, job order,
//     a second workgroup per CU or the tile shape below (tools/deep_bench.hip what-ifs); with the roles split a stage costs the
//     longer of the two;
//   * stages of 128 k (half the barriers), the 4 MFMA waves = 2 row halves x 2 K halves: a wave holds 64 rows x 32 (x 2 gate halves)
//     accumulators and walks 4 of a stage's 8 k-steps -- per k-step 2 A fragments + 1 (2) B fragments for 2 (4) MFMAs, i.e.
//     0.75-1 KB of LDS reads per MFMA instead of 2; the two K halves meet once, in the epilogue, through LDS, where each wave
//     takes over one 32-row tile (all four waves run the epilogue);
//   * a thread stages a FIXED 8-k column of 8 rows per stage, so it reads its FiLM gain / bias once per stage (16 values),
//     not once per unit, and keeps its rows' statistics in registers;
//   * the statistics merge handles the K / 32 = 32 partials per row with two threads per row in the equal-count form (the
//     loader's fast path stops at 16 partials per row and fell back to Chan's update with its divisions: most of the 3.3 us
//     ahead of the first chunk).
// (Requesting the residual epilogue's operands behind the last K stages moved nothing: 2.3 -> 2.1 us of epilogue, same launch time -- it is stores.)
// Arithmetic and rounding points are the loaders' (LdF32LN_T<false>, LdBF16Plain) and the epilogues' own (tile_epilogue_mfma);
// the accumulation order differs (two K halves), which the parity tests' tolerance covers like any other tile shape.
#pragma once
#include "hd_gemm.hpp"

namespace hd {

#ifdef HD_WIDE_NOB                                                 // what-if (tools/deep_bench): every stage re-reads stage 0's weights (L1 / L2 resident)
#define HD_WIDE_BSTAGE(s) 0
#else
#define HD_WIDE_BSTAGE(s) (s)
#endif
#ifdef HD_WIDE_NOA                                                 // what-if: every stage re-reads stage 0's activations
#define HD_WIDE_ASTAGE(s) 0
#else
#define HD_WIDE_ASTAGE(s) (s)
#endif
typedef float f32x4_wi __attribute__((ext_vector_type(4)));
#ifndef HD_WIDE_P
#define HD_WIDE_P 3
#endif
#ifndef HD_WIDE_PB
#define HD_WIDE_PB 2
#endif
// FORM 0: 128 rows x 128 k per stage, the MFMA waves = 2 row halves (64 rows) x 2 K halves (level 3: 1024 rows, K = 1024: 8 stages of 32 KB);
// FORM 1: 256 rows x 64 k per stage, the MFMA waves = 4 row quarters (64 rows), no K split (level 2: 4096 rows, K = 512: 8 stages of 32 KB);
// FORM 2: 64 rows x 128 k per stage, the MFMA waves = 2 row tiles (32 rows) x 2 K halves (middle level: 256 rows, K = 2048: 16 stages of 16 KB).
template <bool PAIR, int FORM>
struct WideCfg {
    static constexpr int BM = FORM == 1 ? 256 : (FORM == 2 ? 64 : 128), BK = FORM == 1 ? 64 : 128, THREADS = 512, TNT = PAIR ? 2 : 1, P = HD_WIDE_P;   // 4 MFMA waves + 4 staging waves; stages in flight
    static constexpr int UPR = BK / 8, RG = 256 / UPR;           // 16-byte units per staged row; rows covered by the 256 staging threads per pass
    static constexpr int UPT = BM * UPR / 256;                   // passes = units per staging thread and stage (8, 8, 4)
    static constexpr int KSP = FORM == 1 ? 1 : 2, RW = 4 / KSP;  // K halves and row parts of the MFMA waves
    static constexpr int MTW = FORM == 2 ? 1 : 2;                // 32-row tiles per MFMA wave
    static constexpr int TPRW = 256 / BM;                        // threads per row of the statistics merge (2, 1, 4)
    static constexpr int AROW = BK * 2 + 16;                     // bytes per staged row (+ 16 B pad)
    static constexpr int A_BUF = BM * AROW;
    static constexpr int A_OFF = 0, STATS_OFF = 2 * A_BUF, GB_OFF = STATS_OFF + BM * 8;   // + 2 K floats (LayerNorm form)
    static constexpr int PB = HD_WIDE_PB;                        // stages of weight fragments in flight per MFMA wave
};

// 0: not a shape of this kernel; 1: the 128-row form (FORM 0); 2: the 256-row form (FORM 1); 3: the 64-row form (FORM 2)
template <bool PAIR>
inline int wide_shape_ok(const GemmP& p) {
    const int ncols = PAIR ? p.N / 2 : p.N;
    if (p.K != p.Kp || ncols % 32 != 0 || p.film_face_stride != 0 || p.lda != p.K) return 0;
    if (p.Kp == 1024 && p.M % 128 == 0 && p.M >= 1024 && p.M <= 2048 && (p.M / 128) * (ncols / 32) >= 256) return 1;
    static const bool no_tall = hd_env("HD_NO_WIDE_TALL") != nullptr, no_64 = hd_env("HD_NO_WIDE_64") != nullptr;
    if (!no_tall && p.Kp == 512 && p.M % 256 == 0 && p.M >= 4096 && (p.M / 256) * (ncols / 32) >= 256 && (p.M / 256) * (ncols / 32) <= 512) return 2;
    if (!no_64 && p.Kp == 2048 && p.M % 64 == 0 && (p.M / 64) * (ncols / 32) >= 256 && (p.M / 64) * (ncols / 32) <= 512) return 3;
    return 0;
}
inline int wide_form_rows(int form) { return form == 1 ? 128 : (form == 2 ? 256 : 64); }

// the statistics merge takes up to 16 partials per thread: 32 per row in the 128-row form (two threads per row), 16 in the 256-row form
template <bool PAIR>
inline bool wide_stats_ok(const GemmP& p) {
    const int form = wide_shape_ok<PAIR>(p);
    return form != 0 && p.stats_np >= 1 && p.stats_np <= (form == 1 ? 32 : (form == 2 ? 16 : 64)) && p.stats_np * p.stats_cnt == p.K;
}

#ifdef HD_STAMPS     // tools/deep_bench: per-stage stamps of workgroup 0's first staging wave (role 1) and first MFMA wave (role 0), slots 4096 + role * 64 + 4 stage + k
#define HD_WSTAMP(role, s, k) do { if (p.stamps && blockIdx.x == 0 && blockIdx.y == 0 && lane == 0 && pw == 0) p.stamps[4096 + (role) * 64 + 4 * (s) + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define HD_WSTAMP(role, s, k) do { } while (0)
#endif
template <bool LN, class EP, bool PAIR, int FORM, int NST>
__global__ __launch_bounds__(512) void gemm_wide_kernel(const GemmP p) {
    typedef WideCfg<PAIR, FORM> C;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // waves 0..3: MFMA waves (row half rh, K half kh); waves 4..7: staging waves (global -> LayerNorm transform -> LDS).  One of each per SIMD:
    // the matrix pipe and the VALU / LDS-store work of a stage run side by side instead of behind each other
    const bool producer = wave >= 4;
    const int pw = wave & 3, ptid = tid & 255;
    const int rh = pw % C::RW, kh = pw / C::RW;
    const int ksteps_total = p.Kp >> 4;
    int bx = blockIdx.x, by = blockIdx.y;
    if (p.xcd_tile_affine && (gridDim.y & 7) == 0) {               // gemm_deep_kernel's block map: the row groups of a weight tile on one XCD
        const int lin = blockIdx.y * gridDim.x + blockIdx.x, j = lin >> 3;
        bx = j % (int)gridDim.x;
        by = (j / (int)gridDim.x) * 8 + (lin & 7);
    }
    const int row0 = bx * C::BM;
    const int tile0 = by, tile1 = by + (p.N >> 6);                  // pair: the second gate half starts at N/2 = 32 * (N/64)
    HD_STAMP(0);

    // ---- staging roles: A unit (16 B) column q16 of rows r8 + 16 u; B fragments f = wave + 4 j -> (gate half f / 8, k-step f % 8) ----
    const int q16 = ptid % C::UPR, r8 = ptid / C::UPR;
    const unsigned short* Ap = reinterpret_cast<const unsigned short*>(p.A) + (size_t)(row0 + r8) * p.lda + q16 * 8;
    u32x4 raw[C::P][C::UPT];
#define HD_WIDE_LOAD(s)                                                                                            \
    {                                                                                                               \
        _Pragma("unroll") for (int u = 0; u < C::UPT; ++u)                                                          \
            raw[(s) % C::P][u] = *reinterpret_cast<const u32x4*>(Ap + (size_t)(C::RG * u) * p.lda + (HD_WIDE_ASTAGE(s)) * C::BK);    \
    }
    if (producer) {
#pragma unroll
        for (int d = 0; d < C::P; ++d) HD_WIDE_LOAD(d);
    }
    HD_STAMP(1);

    // ---- LayerNorm: (mean, rstd) of the 128 rows from the producer's partials, FiLM row to LDS ----
    float mu[C::UPT], rs[C::UPT];
    float* gb = reinterpret_cast<float*>(smem + C::GB_OFF);
    if constexpr (LN) {
      if (!producer) {                                                 // the MFMA waves have nothing to do yet: they merge the statistics and stage the FiLM row
        const int rl = tid / C::TPRW, part = tid % C::TPRW, np = p.stats_np;
        const float2* sp = p.stats_in + (size_t)(row0 + rl) * np;
        float2 s[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) { const int j = part + C::TPRW * i; s[i] = j < np ? sp[j] : make_float2(0.f, -1.f); }      // M2 < 0 marks "no partial"
        const float* f = LdF32LN_T<false>::film_row(p);
        for (int k = tid * 4; k < p.K; k += 256 * 4) {
            *reinterpret_cast<float4*>(gb + k) = *reinterpret_cast<const float4*>(f + p.film_gain_off + k);
            *reinterpret_cast<float4*>(gb + p.K + k) = *reinterpret_cast<const float4*>(f + p.film_bias_off + k);
        }
        // equal-count partials: mean = sum(mean_i) / np, M2 = sum(M2_i + cnt (mean_i - mean)^2)   (LdF32LN_T::block_finish)
        float sm = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) sm += s[i].y >= 0.f ? s[i].x : 0.f;
        if (C::TPRW >= 2) sm += dpp_mov<0xB1>(sm);                  // lane ^ 1, lane ^ 2: the row's other threads
        if (C::TPRW == 4) sm += dpp_mov<0x4E>(sm);
        const float inv_np = 1.0f / (float)np, cnt = (float)p.stats_cnt;
        const float mean = sm * inv_np;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) { const float d = s[i].x - mean; q += s[i].y >= 0.f ? fmaf(cnt * d, d, s[i].y) : 0.f; }
        if (C::TPRW >= 2) q += dpp_mov<0xB1>(q);
        if (C::TPRW == 4) q += dpp_mov<0x4E>(q);
        const float var = q * (inv_np / cnt);
        if (part == 0) reinterpret_cast<float2*>(smem + C::STATS_OFF)[rl] = make_float2(mean, __frsqrt_rn(var + p.ln_eps));
      }
        __syncthreads();
        if (producer)
#pragma unroll
        for (int u = 0; u < C::UPT; ++u) {
            float2 st = reinterpret_cast<const float2*>(smem + C::STATS_OFF)[r8 + C::RG * u];
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(st.x), "+v"(st.y));      // two registers of their own (LdF32LN_T::unit_stats)
            mu[u] = -st.x * st.y; rs[u] = st.y;
        }
    }
    // stage -> LDS: the LayerNorm transform (two fused multiply-adds per element, round to nearest even) on the way in
    // ls >= 0: the loads of stage ls are requested one unit at a time between the units' transforms: a wave that requests eight units in a row
    // stands at the issue of its vector-memory instructions until the CU's load path has taken them (0.5-0.7 us per stage: this CU takes in
    // 64 KB per stage), and only then starts on the transform (0.7 us)
    auto write_stage = [&](int s, int ls) __attribute__((always_inline)) {
        char* sA = smem + C::A_OFF + (s & 1) * C::A_BUF + r8 * C::AROW + q16 * 16;
        if constexpr (LN) {
            typedef __attribute__((address_space(3))) const float lds_f1;
            typedef float f4v __attribute__((ext_vector_type(4)));
            typedef __attribute__((address_space(3))) const f4v lds_f4;
            lds_f1* gl = (lds_f1*)gb;
            const int k = s * C::BK + q16 * 8;
            const f4v g0 = *(lds_f4*)(gl + k), g1 = *(lds_f4*)(gl + k + 4), b0 = *(lds_f4*)(gl + p.K + k), b1 = *(lds_f4*)(gl + p.K + k + 4);
            const f32x2_t g[4] = {{g0.x, g0.y}, {g0.z, g0.w}, {g1.x, g1.y}, {g1.z, g1.w}}, b[4] = {{b0.x, b0.y}, {b0.z, b0.w}, {b1.x, b1.y}, {b1.z, b1.w}};
#pragma unroll
            for (int u = 0; u < C::UPT; ++u) {
                const u32x4 w = raw[s % C::P][u];
                const f32x2_t r2 = {rs[u], rs[u]}, m2 = {mu[u], mu[u]};
                u32x4 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x2_t x = {__uint_as_float(w[i] << 16), __uint_as_float(w[i] & 0xffff0000u)};
                    o[i] = pack2(__builtin_elementwise_fma(__builtin_elementwise_fma(x, r2, m2), g[i], b[i]));
                }
                *reinterpret_cast<u32x4*>(sA + C::RG * u * C::AROW) = o;
                if (ls >= 0) {
                    raw[ls % C::P][u] = *reinterpret_cast<const u32x4*>(Ap + (size_t)(C::RG * u) * p.lda + (HD_WIDE_ASTAGE(ls)) * C::BK);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < C::UPT; ++u) {
                *reinterpret_cast<u32x4*>(sA + C::RG * u * C::AROW) = raw[s % C::P][u];
                if (ls >= 0) {
                    raw[ls % C::P][u] = *reinterpret_cast<const u32x4*>(Ap + (size_t)(C::RG * u) * p.lda + (HD_WIDE_ASTAGE(ls)) * C::BK);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    };
    // two loops, one per role, with the same barriers: the staging waves' register rings and the MFMA waves' accumulators never
    // live in the same code (in one loop with a branch per role the allocator kept both sets: 218 registers spilled)
    if (producer) {
#ifdef HD_WIDE_PRIO
        __builtin_amdgcn_s_setprio(HD_WIDE_PRIO);                      // the staging wave's instructions ahead of the MFMA wave's on the shared SIMD
#endif
        write_stage(0, -1);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
        for (int s = 0; s < NST; ++s) {
            HD_WSTAMP(1, s, 0);
            HD_WSTAMP(1, s, 1);
            if (s + 1 < NST) write_stage(s + 1, s + C::P < NST ? s + C::P : -1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            HD_WSTAMP(1, s, 2);
            asm volatile("s_barrier" ::: "memory");
            HD_WSTAMP(1, s, 3);
        }
        asm volatile("s_barrier" ::: "memory");                         // the MFMA waves' exchange
        if constexpr (!EP::kTile) return;                               // (the depthwise epilogue below is work for all 512 threads)
    }
#undef HD_WIDE_LOAD
    constexpr int NM = C::KSP == 1 ? 2 : 1;                             // row tiles a MFMA wave finishes: both of its own (no K split) / one after the exchange
    const bool finisher = producer ? false : (C::MTW == 2 || kh == 0);  // one tile per wave and two K halves: K half 0 finishes it
    f32x16_t mine[NM][C::TNT];
    if (!producer) {
    // the MFMA waves take their weight fragments straight from global memory (packed in fragment order: one coalesced 1 KiB load per
    // fragment): through LDS they were a third of the stage's LDS traffic (16 KB stored, 32 KB read per 128 k), and the LDS is what a
    // stage costs once the roles are split.  The two row halves request the same fragments (the second request hits L1 / L2).
    u32x4 breg[C::PB][4][C::TNT];
    const uint4* Wb[C::TNT];
#pragma unroll
    for (int t = 0; t < C::TNT; ++t) Wb[t] = p.W + ((size_t)(t ? tile1 : tile0) * ksteps_total + kh * 4) * 64 + lane;
#define HD_WIDE_BLOAD(s)                                                                                            \
    {                                                                                                               \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                               \
            _Pragma("unroll") for (int t = 0; t < C::TNT; ++t)                                                      \
                breg[(s) % C::PB][j][t] = *reinterpret_cast<const u32x4*>(Wb[t] + (size_t)((HD_WIDE_BSTAGE(s)) * (C::BK / 16) + j) * 64); \
    }
#pragma unroll
    for (int d = 0; d < C::PB; ++d) HD_WIDE_BLOAD(d);
    asm volatile("s_barrier" ::: "memory");                             // stage 0 is in LDS
    HD_STAMP(2);
    f32x16_t acc[C::MTW][C::TNT];
#pragma unroll
    for (int mt = 0; mt < C::MTW; ++mt)
#pragma unroll
        for (int t = 0; t < C::TNT; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][t][i] = 0.f;
    const int a_lane_off = (rh * 32 * C::MTW + (lane & 31)) * C::AROW + (lane >> 5) * 16 + kh * 4 * 32;
#pragma unroll
    for (int s = 0; s < NST; ++s) {
        HD_WSTAMP(0, s, 0);
        const char* sA = smem + C::A_OFF + (s & 1) * C::A_BUF + a_lane_off;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bf16x8_t a0 = *reinterpret_cast<const bf16x8_t*>(sA + j * 32), a1 = *reinterpret_cast<const bf16x8_t*>(sA + (C::MTW == 2 ? 32 : 0) * C::AROW + j * 32);
#pragma unroll
            for (int t = 0; t < C::TNT; ++t) {
                const bf16x8_t b = __builtin_bit_cast(bf16x8_t, breg[s % C::PB][j][t]);
#ifdef HD_WIDE_NOMMA                                                       // what-if (tools/deep_bench): no MFMAs, the fragments are still read
                acc[0][t][0] += __builtin_bit_cast(f32x4_wi, a0)[0] + __builtin_bit_cast(f32x4_wi, b)[0]; acc[C::MTW - 1][t][0] += __builtin_bit_cast(f32x4_wi, a1)[1];
#else
                acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b, acc[0][t], 0, 0, 0);
                if constexpr (C::MTW == 2) acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b, acc[1][t], 0, 0, 0);
#endif
            }
        }
        if (s + C::PB < NST) {
            HD_WIDE_BLOAD(s + C::PB);
            asm volatile("" ::: "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        HD_WSTAMP(0, s, 2);
        asm volatile("s_barrier" ::: "memory");
        HD_WSTAMP(0, s, 3);
    }
#undef HD_WIDE_BLOAD
    HD_STAMP(3); HD_STAMP(4);

    if constexpr (C::KSP == 2) {
        // ---- the two K halves meet: with two row tiles per wave, wave (rh, kh) takes over row tile kh of its row half and gets the partner's partial of it;
        //      with one, K half 1 hands its tile to K half 0 ----
        float* xch = reinterpret_cast<float*>(smem);                   // [wave][TNT][16][64]: the staging buffers are dead (the loop ended with a barrier)
#pragma unroll
        for (int t = 0; t < C::TNT; ++t) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if constexpr (C::MTW == 2) {
                    xch[((wave * C::TNT + t) * 16 + i) * 64 + lane] = kh ? acc[0][t][i] : acc[1][t][i];      // the tile the partner keeps
                    mine[0][t][i] = kh ? acc[1][t][i] : acc[0][t][i];
                } else {
                    if (kh) xch[((wave * C::TNT + t) * 16 + i) * 64 + lane] = acc[0][t][i];
                    mine[0][t][i] = acc[0][t][i];
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const int partner = rh + C::RW * (1 - kh);
        if (C::MTW == 2 || kh == 0) {
#pragma unroll
            for (int t = 0; t < C::TNT; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float o = xch[((partner * C::TNT + t) * 16 + i) * 64 + lane];
                    mine[0][t][i] = kh ? o + mine[0][t][i] : mine[0][t][i] + o;         // K half 0 + K half 1, whoever adds
                }
        }
    } else {
#pragma unroll
        for (int m = 0; m < NM; ++m)
#pragma unroll
            for (int t = 0; t < C::TNT; ++t) mine[m][t] = acc[m][t];
        asm volatile("s_barrier" ::: "memory");                         // (the staging waves' last barrier)
    }
    }
    if constexpr (EP::kTile) {
        // ====== conv1 bias -> depthwise 3x3 -> SimpleGate -> G, pooled on the 128-row tile: gemm_deep_pair8_kernel's tile epilogue
        // (conditional_naf.py:116-119), all 512 threads.  t1[half][128 rows][32] fp32 behind the exchange area.
        constexpr int T1_OFF = C::KSP == 2 ? 32768 : 0;                     // behind the exchange area where there is one
        float* t1 = reinterpret_cast<float*>(smem + T1_OFF);
        float* rs = reinterpret_cast<float*>(smem + T1_OFF + 2 * C::BM * 32 * 4);  // [BM / S][32] row sums (over dead staging / statistics / FiLM rows)
        const int S = p.side, ls = 31 - __builtin_clz(S), HW = p.hw;
        float* wx = rs + (C::BM >> ls) * 32;                                // [2][10][32] taps + bias of both halves
        if (finisher) {
#pragma unroll
            for (int t = 0; t < C::TNT; ++t) {
                const float b1 = p.bias[(t ? tile1 : tile0) * 32 + (lane & 31)];
#pragma unroll
                for (int m = 0; m < NM; ++m)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int r = rh * 32 * C::MTW + (C::MTW == 1 ? 0 : (C::KSP == 2 ? kh : m)) * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
                        t1[(t * C::BM + r) * 32 + (lane & 31)] = mine[m][t][i] + b1;
                    }
            }
        } else if (producer && pw == 0) {                                   // one staging wave fetches the taps: half-wave 0 holds half a, half-wave 1 half b
            const int hb = (tid >> 5) & 1, jj = tid & 31, ce = tile0 * 32 + jj + hb * (p.N >> 1);
#pragma unroll
            for (int t = 0; t < 9; ++t) wx[(hb * 10 + t) * 32 + jj] = p.dw_w[(size_t)t * p.N + ce];
            wx[(hb * 10 + 9) * 32 + jj] = p.dw_b[ce];
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const int C2 = p.N >> 1;
        const int j = tid & 31, col = tile0 * 32 + j;
        float wa[9], wb[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) { wa[t] = wx[t * 32 + j]; wb[t] = wx[(10 + t) * 32 + j]; }
        const float ba = wx[9 * 32 + j], bb = wx[19 * 32 + j];
        const float* t1a = t1 + j;
        const float* t1b = t1 + C::BM * 32 + j;
        const int nrows_img = C::BM >> ls;
        for (int rr = tid >> 5; rr < nrows_img; rr += C::THREADS / 32) {
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
        const int faces = C::BM / HW;
        for (int idx = tid; idx < faces * 32; idx += C::THREADS) {
            const int f = idx >> 5;
            float sacc = 0.f;
            for (int r = 0; r < S; ++r) sacc += rs[(f * S + r) * 32 + j];
            const int face = row0 / HW + f;
            const float pm = sacc / (float)HW;
            p.pooled[(size_t)face * C2 + col] = pm;
            if (p.pooled16) p.pooled16[(size_t)face * C2 + col] = f32_to_bf16_bits(pm);
        }
        HD_STAMP(5);
    } else {
        const int ncols = PAIR ? (p.N >> 1) : p.N;
        const int col = tile0 * 32 + (lane & 31);
        if (finisher)
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            const int rtile = row0 + rh * 32 * C::MTW + (C::MTW == 1 ? 0 : (C::KSP == 2 ? kh : m)) * 32;
            tile_epilogue_mfma<true, PAIR, EP>(p, mine[m][0], mine[m][PAIR ? 1 : 0], rtile + 4 * (lane >> 5), col, ncols, tile0, lane);
        }
        HD_STAMP(5);
    }
}

template <bool LN, class EP, bool PAIR, int FORM, int NST>
inline hipError_t launch_gemm_wide_inst(const GemmP& p, hipStream_t s) {
    typedef WideCfg<PAIR, FORM> C;
    const int ncols = PAIR ? p.N / 2 : p.N;
    int smem = C::GB_OFF + (LN ? 2 * p.Kp * 4 : 0);
    const int epi = 32768 + 2 * C::BM * 32 * 4 + (C::BM / 2) * 32 * 4 + 2 * 10 * 32 * 4;     // exchange area, T1 tiles, row sums (faces of >= 2 x 2), taps: the depthwise epilogue's LDS
    if (EP::kTile && smem < epi) smem = epi;
    static std::atomic<unsigned long long> granted{0};
    { const hipError_t e = grant_dynamic_lds(reinterpret_cast<const void*>(&gemm_wide_kernel<LN, EP, PAIR, FORM, NST>), 160 * 1024, granted); if (e != hipSuccess) return e; }
    hipLaunchKernelGGL((gemm_wide_kernel<LN, EP, PAIR, FORM, NST>), dim3(p.M / C::BM, ncols / 32, 1), dim3(C::THREADS), smem, s, p);
    return hipGetLastError();
}
template <bool LN, class EP, bool PAIR>
inline hipError_t launch_gemm_wide(const GemmP& p, hipStream_t s) {
    const int form = wide_shape_ok<PAIR>(p);
    if (form == 1) return launch_gemm_wide_inst<LN, EP, PAIR, 0, 8>(p, s);          // K / BK stages
    if (form == 2) return launch_gemm_wide_inst<LN, EP, PAIR, 1, 8>(p, s);
    if (form == 3) return launch_gemm_wide_inst<LN, EP, PAIR, 2, 16>(p, s);
    return hipErrorInvalidValue;
}

}  // namespace hd
