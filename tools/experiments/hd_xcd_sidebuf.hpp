// hd_xcd.hpp — a run of ConditionalNAFBlocks of one deep level as ONE launch whose dependencies never leave an XCD
// (gfx950 only).
//
// Faces never interact inside the sampling loop (models/denoiser/conditional_naf.py:108-136 is per sample; the SCA pool
// and LayerNorm2d are per face / per pixel), and at latent 16 a face has 4 pixels at level 3 (C = 1024) and 16 at level 2
// (C = 512).  So the batch is cut into 8 groups of 8 faces, one per XCD (32 CUs behind one L2): the 32 workgroups of a
// group own the 32-column output tiles of every GEMM of the block for the group's rows, and the five dependent phases of
// a block
//     q0  LN + FiLM -> conv1 -> depthwise 3x3 -> SimpleGate -> pooled        (conditional_naf.py:114-119)
//     q1  sca(pooled) ; G <- G * s                                            (:119)
//     q2  conv3 ; y = x + beta * (.) ; LayerNorm partials                    (:120-123)
//     q3  LN + FiLM -> conv4 -> SimpleGate                                    (:126-129)
//     q4  conv5 ; x' = y + gamma * (.) ; LayerNorm partials                  (:130-134)
// are separated by a barrier that only the group takes part in.  What one launch per GEMM pays per phase — a kernel
// boundary, a cold start and the first HBM round trip (5-10 us for <= 4 MB of weights, profiles/r02_kernel_table.txt) —
// becomes one flag line in the XCD's L2 (0.8 us, profiles/r03_xcd_barrier_bench.txt); the next phase's weights are
// requested before the barrier and arrive while it is taken.
//
// Arithmetic is that of the per-GEMM launches (hd_gemm.hpp), operation for operation: the same K split over the waves
// (8 slices of 128 at level 3, 4 at level 2), the same reduction order, the same LayerNorm partial merge, the same
// rounding points -> the results are bit-identical to that path (tests/test_gpu_parity.py compares them).
//
// Hand-off inside a group.  A workgroup's outputs of a phase are stored, every storing wave drains its stores
// (s_waitcnt vmcnt(0)), the workgroup's barrier, then ONE lane stores the workgroup's flag word; a consumer polls the
// group's 32 flags (one 128-byte line) with ONE 32-lane sc1 load and, after its workgroup barrier, reads the payload with
// sc1 loads only (they bypass the CU's L1, which another CU's stores never refresh).  Flags carry an epoch the kernel
// advances itself (launch counter per group), so nothing is zeroed between launches or graph replays.
//   * All 32 workgroups of a group on one XCD (the normal case: blocks b and b + 8 share an XCD under round-robin
//     dispatch; checked at run time with HW_REG_XCC_ID through a start-of-launch handshake): plain payload and flag
//     stores — they stay in the XCD's L2, where the consumers' sc1 loads find them.
//   * Otherwise (any other placement): payload and flag stores are write-through (sc1), the placement-independent form
//     of MI355X_MICROARCH.md "Valid forms" row 1.  Same results, slower hand-offs.
// Every spin is bounded; a timeout raises a host-visible word and the group drains out (the library then reports the
// call as failed and falls back to one launch per GEMM).  All 256 workgroups must be resident (one per CU): nothing else
// may occupy the GPU's CUs for longer than the spin bound.
#pragma once
#include <type_traits>

#include "hd_gemm.hpp"
#include "hd_stage_api.hpp"

#pragma clang fp contract(off)                         // as hd_gemm.hpp: every fused multiply-add is written out

namespace hd {

template <int C_, int HW_>
struct XcdCfg {
    static constexpr int C = C_, HW = HW_;
    static constexpr int R = XS_FACES * HW;              // rows of a group: 32 (level 3) / 128 (level 2)
    static constexpr int NT = C / 32;                    // 32-column tiles of a C-wide output: 32 / 16
    static constexpr int RSPLIT = XS_GROUP_WG / NT;      // row groups inside the XCD: 1 / 2
    static constexpr int RCU = R / RSPLIT;               // rows per workgroup: 32 / 64
    static constexpr int WMW = RCU / 32;                 // 32-row MFMA tiles per workgroup: 1 / 2
    static constexpr int WK = 8 / WMW;                   // K slices (waves per row tile): 8 / 4
    static constexpr int CPW = C / 64 / WK;              // 64-deep chunks per wave: 2 / 2
    static constexpr int KS = C / 16;                    // k-steps of a K = C GEMM
    static constexpr int FCU = RCU / HW;                 // faces per workgroup: 8 / 4
    static constexpr int S = (HW == 4) ? 2 : (HW == 16) ? 4 : (HW == 64) ? 8 : (HW == 256) ? 16 : 1;   // face side
    static constexpr int TPR = XS_THREADS / RCU;         // threads per row in the LayerNorm partial merge: 16 / 8
    static constexpr int NIT = RCU * 32 / XS_THREADS;    // tile elements per thread: 2 / 4
    static constexpr int A_WAVE = 32 * LDS_ROW;          // private staging tile of a wave
    // second half of the next phase's weights: requested behind wave 0's hand-off stores (level 3: 64 KB per workgroup would
    // otherwise sit in the CU's in-order memory queue ahead of them: 216.9 -> 206.2 us for 8 blocks) or before them (level 2:
    // 32 KB, where the extra barrier costs more than the queueing: 89.7 vs 92.4 us for 4 blocks)
    static constexpr bool kStoresFirst = C_ >= 1024;
    // level 3: the weight fragments a wave needs FIRST in the next-but-one phase go into an LDS side buffer a whole phase ahead (below)
    static constexpr bool kSide = C_ >= 1024;
    static_assert(RSPLIT * NT == XS_GROUP_WG && WMW * WK == 8 && CPW == 2 && S * S == HW, "geometry");
    static_assert(RCU / S * 32 == XS_THREADS, "one (channel, image row) item per thread in the depthwise epilogue");
    static_assert(TPR * 4 >= NT, "partials per thread");
};

template <int C, int HW>
struct XLds {
    typedef XcdCfg<C, HW> K;
    char stage[K::WK * K::RCU * 32 * 2 * 4];              // wave-private A staging (8 x 4608 B), then the K-split partial tiles
    float xt[K::RCU * 32], yt[K::RCU * 32];               // this workgroup's tile of the residual stream (x, y), fp32
    unsigned short gt[K::RCU * 32];                       // bf16 gate tile between q0 and q1
    float rs[(K::RCU / K::S) * 32];                       // depthwise row sums
    float pl[XS_FACES * 32];                              // pooled tile / sca tile
    float2 stats[K::RCU];                                 // (mean, rstd) per row
    float gb[2 * C];                                      // FiLM gain | bias
    float dwc[22 * 32];                                   // per-column constants of the fused depthwise epilogue
    XBlockW blk[XS_MAXBLK];
    unsigned base, local, abort, side0;                   // side0: phase whose early set sits in wave 0's slice of `side` (written by wave 1)
    uint4 side[K::kSide ? 8 * 8 * 64 : 1];                // [wave][8 fragments][64 lanes]: 64 KB
};

typedef __attribute__((address_space(1))) unsigned xs_gu32;
typedef unsigned xs_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned xs_u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned xs_xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xfu;
}
// pointers read from the LDS copy of the block table are generic to the compiler: say that they point to global memory
__device__ __forceinline__ uint4 xs_ldg_u4(const uint4* p) {
    const xs_u32x4 v = *reinterpret_cast<__attribute__((address_space(1))) const xs_u32x4*>((unsigned long long)p);
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float xs_ldg_f(const float* p) { return *reinterpret_cast<__attribute__((address_space(1))) const float*>((unsigned long long)p); }
__device__ __forceinline__ float4 xs_ldg_f4(const float* p) {
    typedef float f4v __attribute__((ext_vector_type(4)));
    const f4v v = *reinterpret_cast<__attribute__((address_space(1))) const f4v*>((unsigned long long)p);
    return make_float4(v.x, v.y, v.z, v.w);
}
// LDS-only workgroup barrier: __syncthreads() is also a fence and would drain the weight loads in flight
__device__ __forceinline__ void xs_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ unsigned xs_lds_addr(const void* p) { return (unsigned)(unsigned long long)(const __attribute__((address_space(3))) void*)p; }
// One 1 KiB fragment global -> LDS without a register destination (LDS-DMA; lane l's 16 bytes land at lds_dst + 16 l).  Written as
// asm: the compiler must not know that it writes LDS (it would put s_waitcnt vmcnt(0) in front of every LDS access of the K loop),
// the ordering is the kernel's own (see side_read).  M0 is the compiler's: saved and restored.
__device__ __forceinline__ void xs_dma16(const uint4* gsrc_lane, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc_lane), "s"(lds_dst) : "memory");
}

#ifdef HD_STAMPS
#define HD_XSTAMP(i) do { if (p.stamps && tid == 0) p.stamps[((size_t)ph * 256 + blockIdx.x) * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define HD_XSTAMP(i) do { } while (0)
#endif

template <int C, int HW>
__global__ __launch_bounds__(XS_THREADS) void xcd_stage_kernel(const XStageP p) {
    typedef XcdCfg<C, HW> K;
    __shared__ __attribute__((aligned(16))) XLds<C, HW> L;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int group = blockIdx.x & 7, rank = blockIdx.x >> 3;        // blocks b and b + 8 share an XCD under round-robin dispatch (speed only)
    const int face_g0 = group * XS_FACES;
    if (face_g0 >= p.B) return;                                      // no faces for this group: nobody of the group takes part
    const int ct = rank % K::NT, rsp = rank / K::NT;
    const int M = p.B * HW;
    const int row0 = face_g0 * HW + rsp * K::RCU;                    // first row of this workgroup's tile
    const int face0 = row0 / HW;
    const int wm = wave / K::WK, wk = wave - wm * K::WK;
    const int c0 = wk * K::CPW;
    const int kq = lane & 7;
    const int col = ct * 32 + (tid & 31);                            // epilogue: this thread's column

    // ---- the block table, and flags ----
    {
        const unsigned* src = reinterpret_cast<const unsigned*>(p.blocks);
        unsigned* dst = reinterpret_cast<unsigned*>(L.blk);
        for (int i = tid; i < p.nblocks * (int)(sizeof(XBlockW) / 4); i += XS_THREADS) dst[i] = src[i];
        if (tid == 0) L.abort = __hip_atomic_load((xs_gu32*)p.abort_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // an earlier stage of this call gave up
    }
    xs_lds_barrier();

#ifdef HD_STAMPS
    const size_t act_bytes = p.dbg_no_a ? 0 : (size_t)M * C * 2;    // zero records: the range check drops the loads, the instruction stream stays
#else
    const size_t act_bytes = (size_t)M * C * 2;
#endif
    const __amdgpu_buffer_rsrc_t rs_Xb = __builtin_amdgcn_make_buffer_rsrc(p.Xb, 0, (int)act_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_Yb = __builtin_amdgcn_make_buffer_rsrc(p.Yb, 0, (int)act_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_G = __builtin_amdgcn_make_buffer_rsrc(p.G, 0, (int)act_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_P16 = __builtin_amdgcn_make_buffer_rsrc(p.pooled16, 0, p.B * C * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_sx = __builtin_amdgcn_make_buffer_rsrc(p.sx, 0, M * K::NT * 8, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_sy = __builtin_amdgcn_make_buffer_rsrc(p.sy, 0, M * K::NT * 8, 0x00020000);

    uint4 bq[K::CPW][4][2];                                          // B fragments of this wave's K slice: [chunk][k-step][pair half]
    xs_u32x4 aq[K::CPW][4];                                          // raw A units: rows (lane >> 3) + 8u, 8 k at 8 * kq
    f32x16_t acc[2];

    // one 64-deep chunk of this wave's K slice (d = 0, 1).  A wave stalls at ISSUE once the CU's load queue is full (about
    // 80 KB in flight), so the next phase's weights are requested in two halves with epilogue work in between
    auto load_w_chunk = [&](const uint4* W, auto pair_c, auto d_c) __attribute__((always_inline)) {
        constexpr bool PAIR = decltype(pair_c)::value;
        constexpr int d = decltype(d_c)::value;
#ifdef HD_STAMPS
        if (p.dbg_no_w) return;
#endif
        const uint4* Wl = W + lane;
#pragma unroll
        for (int ss = 0; ss < 4; ++ss) {
            bq[d][ss][0] = xs_ldg_u4(Wl + ((size_t)ct * K::KS + (c0 + d) * 4 + ss) * 64);
            if (PAIR) bq[d][ss][1] = xs_ldg_u4(Wl + ((size_t)(ct + K::NT) * K::KS + (c0 + d) * 4 + ss) * 64);
        }
    };
    auto load_w = [&](const uint4* W, auto pair_c) __attribute__((always_inline)) {
        load_w_chunk(W, pair_c, std::integral_constant<int, 0>());
        load_w_chunk(W, pair_c, std::integral_constant<int, 1>());
    };
    // ---- LDS side buffer (level 3).  A phase streams 64 / 128 KB of weights per workgroup from the fabric (every XCD reads the level's
    // weights; a pair phase is as large as the XCD's L2): requested during the previous phase's epilogue they arrive late in a tail
    // of the workgroups, and every phase waits for its slowest member (what-if with near-source weights: 214 -> 166 us for 8 blocks).
    // The "early set" of a phase -- the 8 fragments a wave needs first: all of a plain phase, the first chunk (both gate halves) of a
    // pair phase -- is therefore requested TWO phases ahead into LDS (there are no registers left for it) and copied to registers right
    // after the K loop; only the second chunk of a pair phase still comes from memory one phase ahead.  Ordering: a slice is read
    // (side_read) after loads that are younger than its DMA have been waited for by the SAME wave (loads retire in order: the A rows
    // of the phase in between), rewritten only behind a workgroup barrier that follows the read; wave 0, which must not hold loads
    // in flight when it drains its hand-off stores, has its slice filled by wave 1 and learns from an LDS word that it has landed.
    const unsigned side_base = xs_lds_addr(L.side);
    auto side_issue = [&](const uint4* W, auto pair_c, int slice) __attribute__((always_inline)) {
        constexpr bool PAIR = decltype(pair_c)::value;
        static_assert(!K::kSide || K::WMW == 1, "a slice is a wave's K slice");
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int d = PAIR ? 0 : (s >> 2), ss = PAIR ? (s >> 1) : (s & 3), half = PAIR ? (s & 1) : 0;
            const uint4* src = W + ((size_t)(ct + half * K::NT) * K::KS + (slice * K::CPW + d) * 4 + ss) * 64 + lane;
            xs_dma16(src, __builtin_amdgcn_readfirstlane(side_base + (unsigned)((slice * 8 + s) * 1024)));
        }
    };
    auto side_read = [&](auto pair_c) __attribute__((always_inline)) {
        constexpr bool PAIR = decltype(pair_c)::value;
        xs_u32x4 v[8];
        const unsigned a = side_base + (unsigned)(wave * 8 * 1024) + (unsigned)lane * 16u;
        asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:1024\n\tds_read_b128 %2, %8 offset:2048\n\tds_read_b128 %3, %8 offset:3072\n\t"
                     "ds_read_b128 %4, %8 offset:4096\n\tds_read_b128 %5, %8 offset:5120\n\tds_read_b128 %6, %8 offset:6144\n\tds_read_b128 %7, %8 offset:7168\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]) : "v"(a) : "memory");
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int d = PAIR ? 0 : (s >> 2), ss = PAIR ? (s >> 1) : (s & 3), half = PAIR ? (s & 1) : 0;
            bq[d][ss][half] = make_uint4(v[s].x, v[s].y, v[s].z, v[s].w);
        }
    };
    // the three points of a phase ph at which weights are requested.  after_k: right behind the K loop (the registers are free): the
    // next phase's early set from LDS / its first chunk from memory; second_pf: behind wave 0's hand-off stores; wave0_w: phase start
    auto after_k = [&](int ph, const uint4* Wn, auto pairn_c, bool have_next) __attribute__((always_inline)) {
        if constexpr (K::kSide) {
            if (!have_next) return;
            if (wave == 1 && lane == 0) *(volatile unsigned*)&L.side0 = (unsigned)ph + 1u;   // wave 1 has used A rows younger than its DMAs for phase ph + 1
            if (wave == 0) {
                for (unsigned spins = 0; *(volatile unsigned*)&L.side0 != (unsigned)ph + 1u; ++spins)
                    if (spins > XS_SPINS) { L.abort = 1u; break; }           // cannot happen while wave 1 runs the same phase
            }
            side_read(pairn_c);
            if (decltype(pairn_c)::value && wave != 0) load_w_chunk(Wn, std::true_type(), std::integral_constant<int, 1>());
        } else {
            if (wave != 0 && have_next) load_w_chunk(Wn, pairn_c, std::integral_constant<int, 0>());
        }
    };
    auto second_pf = [&](const uint4* Wn, auto pairn_c, bool have_next, const uint4* Wnn, auto pairnn_c, bool have_nn) __attribute__((always_inline)) {
        if constexpr (K::kSide) {
            if (have_nn && wave != 0) {
                if (wave == 1) side_issue(Wnn, pairnn_c, 0);
                side_issue(Wnn, pairnn_c, wave);
            }
        } else {
            if (wave != 0 && have_next) load_w_chunk(Wn, pairn_c, std::integral_constant<int, 1>());
        }
    };
    auto wave0_w = [&](const uint4* W, auto pair_c) __attribute__((always_inline)) {
        if constexpr (K::kSide) { if (decltype(pair_c)::value) load_w_chunk(W, std::true_type(), std::integral_constant<int, 1>()); }
        else load_w(W, pair_c);
    };
    // rows of this wave's A sub-tile: tile row (lane >> 3) + 8u of row tile wm
    auto load_a = [&](const __amdgpu_buffer_rsrc_t& rs) __attribute__((always_inline)) {
#pragma unroll
        for (int d = 0; d < K::CPW; ++d)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int gr = row0 + wm * 32 + (lane >> 3) + 8 * u;
                const int r = gr < M ? gr : 0;                       // rows beyond the batch re-read row 0 and are never stored
                aq[d][u] = __builtin_amdgcn_raw_buffer_load_b128(rs, (r * C + (c0 + d) * 64 + 8 * kq) * 2, 0, 16);
            }
    };
    char* sA = L.stage + wave * K::A_WAVE;
    const int a_lane_off = (lane & 31) * LDS_ROW + (lane >> 5) * 16;
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[tn][i] = 0.f;
    };
    // one 64-deep chunk: staged units -> LDS (wave-private), four k-steps of MFMA
    auto chunk_mma = [&](int d, auto pair_c) __attribute__((always_inline)) {
        constexpr bool PAIR = decltype(pair_c)::value;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int ss = 0; ss < 4; ++ss) {
            const bf16x8_t a = *reinterpret_cast<const bf16x8_t*>(sA + a_lane_off + ss * 32);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8_t, bq[d][ss][0]), acc[0], 0, 0, 0);
            if (PAIR) acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8_t, bq[d][ss][1]), acc[1], 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
    };
    // K loop on bf16 rows taken as they are (conv3, conv5 inputs)
    auto gemm_plain = [&](auto pair_c) __attribute__((always_inline)) {
        zero_acc();
#pragma unroll
        for (int d = 0; d < K::CPW; ++d) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int gr = row0 + wm * 32 + (lane >> 3) + 8 * u;
                const xs_u32x4 v = gr < M ? aq[d][u] : (xs_u32x4){0u, 0u, 0u, 0u};
                *reinterpret_cast<xs_u32x4*>(sA + ((lane >> 3) + 8 * u) * LDS_ROW + 16 * kq) = v;
            }
            chunk_mma(d, pair_c);
        }
    };
    // K loop with LayerNorm2d + FiLM applied to the staged rows (utils.py:16-24, conditional_naf.py:114-115,126-127):
    // the arithmetic of LdF32LN_T<false>::finish_nc
    auto gemm_ln = [&]() __attribute__((always_inline)) {
        zero_acc();
        f32x2_t rsv[4], muv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float2 s = L.stats[wm * 32 + (lane >> 3) + 8 * u];
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 3" : "+v"(s.x), "+v"(s.y));      // see LdF32LN_T::unit_stats
            const float mu = -s.x * s.y;
            rsv[u] = (f32x2_t){s.y, s.y}; muv[u] = (f32x2_t){mu, mu};
        }
#pragma unroll
        for (int d = 0; d < K::CPW; ++d) {
            const int k = (c0 + d) * 64 + 8 * kq;
            f32x2_t g[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                g[i] = *reinterpret_cast<const f32x2_t*>(&L.gb[k + 2 * i]);
                b[i] = *reinterpret_cast<const f32x2_t*>(&L.gb[C + k + 2 * i]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const unsigned w[4] = {aq[d][u].x, aq[d][u].y, aq[d][u].z, aq[d][u].w};
                unsigned o[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x2_t x = {__uint_as_float(w[i] << 16), __uint_as_float(w[i] & 0xffff0000u)};
                    o[i] = pack2(__builtin_elementwise_fma(__builtin_elementwise_fma(x, rsv[u], muv[u]), g[i], b[i]));
                }
                *reinterpret_cast<xs_u32x4*>(sA + ((lane >> 3) + 8 * u) * LDS_ROW + 16 * kq) = (xs_u32x4){o[0], o[1], o[2], o[3]};
            }
            chunk_mma(d, std::true_type());
        }
    };
    // K-split partial tiles -> LDS red[wk][tn][row][32] (aliases the staging tiles: barrier first)
    float* red = reinterpret_cast<float*>(L.stage);
    auto to_red = [&](auto pair_c) __attribute__((always_inline)) {
        constexpr int TNT = decltype(pair_c)::value ? 2 : 1;
        constexpr int TILE_F = K::RCU * 32 * TNT;
        xs_lds_barrier();
#pragma unroll
        for (int tn = 0; tn < TNT; ++tn)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int r = wm * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
                red[wk * TILE_F + (tn * K::RCU + r) * 32 + (lane & 31)] = acc[tn][i];
            }
        xs_lds_barrier();
    };
    // LayerNorm partials of the tile rows -> (mean, rstd) per row in LDS: LdF32LN_T::block_issue / block_finish (fast path).
    // The (small) loads go out BEFORE the A rows, the merge runs while those are still arriving (loads return in order).
    float2 ln_ps[4];
    float4 ln_g = make_float4(0.f, 0.f, 0.f, 0.f), ln_b = ln_g;
    auto ln_issue = [&](const __amdgpu_buffer_rsrc_t& rs, int film_bias_off) __attribute__((always_inline)) {
        constexpr int TPR = K::TPR;
        const int rl = tid / TPR, part = tid % TPR, row = row0 + rl;
        const bool rv = row < M;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int j = part + i * TPR;
            ln_ps[i] = make_float2(0.f, -1.f);                                // M2 < 0 marks "no partial"
            if (rv && j < K::NT) {
                const xs_u32x2 raw = __builtin_amdgcn_raw_buffer_load_b64(rs, (row * K::NT + j) * 8, 0, 16);
                ln_ps[i] = make_float2(__uint_as_float(raw.x), __uint_as_float(raw.y));
            }
        }
        // FiLM gain / bias of this LayerNorm: [bias | gain] at film_bias_off (written by an earlier launch: plain loads)
        static_assert(C <= XS_THREADS * 4, "one float4 of gain and bias per thread");
        const int k = tid * 4;
        if (k < C) {
            ln_b = *reinterpret_cast<const float4*>(p.film + film_bias_off + k);
            ln_g = *reinterpret_cast<const float4*>(p.film + film_bias_off + C + k);
        }
    };
    auto ln_finish = [&]() __attribute__((always_inline)) {
        constexpr int TPR = K::TPR;
        const int rl = tid / TPR, part = tid % TPR;
        const bool rv = row0 + rl < M;
        const int k = tid * 4;
        if (k < C) {
            *reinterpret_cast<float4*>(&L.gb[k]) = ln_g;
            *reinterpret_cast<float4*>(&L.gb[C + k]) = ln_b;
        }
        auto row_sum = [](float v) __attribute__((always_inline)) {
            if (TPR > 1) v += dpp_mov<0xB1>(v);
            if (TPR > 2) v += dpp_mov<0x4E>(v);
            if (TPR > 4) v += dpp_mov<0x141>(v);
            if (TPR > 8) v += dpp_mov<0x140>(v);
            return v;
        };
        float sm = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) sm += ln_ps[i].y >= 0.f ? ln_ps[i].x : 0.f;
        const float inv_np = 1.0f / (float)K::NT;
        const float mean = row_sum(sm) * inv_np;
        const float cnt = 32.f;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float d = ln_ps[i].x - mean;
            q += ln_ps[i].y >= 0.f ? fmaf(cnt * d, d, ln_ps[i].y) : 0.f;
        }
        const float var = row_sum(q) * (inv_np / cnt);
        if (part == 0) L.stats[rl] = make_float2(mean, rv ? __frsqrt_rn(var + p.ln_eps) : 0.f);
        xs_lds_barrier();
    };
    // hand-off stores: plain inside one XCD, write-through otherwise (wave-uniform).  ONLY WAVE 0 stores hand-off data: the other
    // waves request the next phase's weights as soon as their K loop has consumed the current ones, and a wave with loads in
    // flight cannot drain its stores without waiting for those loads too (one in-order counter).
    bool local = false;
    auto st128 = [&](const __amdgpu_buffer_rsrc_t& rs, int off, uint4 v) __attribute__((always_inline)) {
        const xs_u32x4 x = {v.x, v.y, v.z, v.w};
        if (local) __builtin_amdgcn_raw_buffer_store_b128(x, rs, off, 0, 0);
        else __builtin_amdgcn_raw_buffer_store_b128(x, rs, off, 0, 16);
    };
    auto st64 = [&](const __amdgpu_buffer_rsrc_t& rs, int off, float2 v) __attribute__((always_inline)) {
        const xs_u32x2 x = {__float_as_uint(v.x), __float_as_uint(v.y)};
        if (local) __builtin_amdgcn_raw_buffer_store_b64(x, rs, off, 0, 0);
        else __builtin_amdgcn_raw_buffer_store_b64(x, rs, off, 0, 16);
    };
    xs_gu32* flags = (xs_gu32*)(p.flags + group * 32);
    unsigned base = 0;
    bool dead = false;
    // end of a phase (wave 0, after the barrier behind the epilogue): its stores are drained, then one lane stores the flag
    auto publish = [&](int ph) __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) {
            if (local) __hip_atomic_store(flags + rank, base + (unsigned)ph + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else __hip_atomic_store(flags + rank, base + (unsigned)ph + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    // wait until all 32 workgroups of the group have published phase ph (wave 0 polls: it has no load in flight)
    auto wait_phase = [&](int ph) __attribute__((always_inline)) {
        if (wave == 0) {
            const unsigned want = base + (unsigned)ph + 1u;
            for (unsigned spins = 0;; ++spins) {
                const unsigned v = lane < XS_GROUP_WG ? __hip_atomic_load(flags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : want;
                const bool inject = p.test_abort > 0 && ph + 1 == p.test_abort && group == 0;
                if (!inject && __all((int)(v - want) >= 0)) break;
                if (spins > XS_SPINS || inject) {
                    if (lane == 0) {
                        L.abort = 1u;
                        __hip_atomic_store((xs_gu32*)p.abort_dev, 0x100u + (unsigned)ph, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store((xs_gu32*)p.tmo, 0x100u + (unsigned)ph, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        xs_lds_barrier();
        dead = L.abort != 0u;
    };
    // the tile row -> LayerNorm partial (mean, M2) of its 32 columns, kept in LDS for wave 0's store pass
    float2* st_out = L.stats;                                        // (mean, rstd) of the rows are dead once the K loop is over

    // ---- weights of the first phase, then the start-of-launch handshake ----
    load_w(L.blk[0].w1, std::true_type());
    if constexpr (K::kSide) {                                        // early set of the second phase (sca of the first block)
        if (tid == 0) L.side0 = 0u;
        if (wave == 1) side_issue(L.blk[0].wsca, std::false_type(), 0);
        if (wave != 0) side_issue(L.blk[0].wsca, std::false_type(), wave);
    }
    if (wave == 0) {
        xs_gu32* gs = (xs_gu32*)(p.gstate + group * 32);
        xs_gu32* hello = (xs_gu32*)(p.hello + group * 32);
        const unsigned n = __hip_atomic_load(gs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned xcc = xs_xcc_id();
        const unsigned mine = ((n + 1u) << 4) | xcc;
        if (lane == 0) __hip_atomic_store(hello + rank, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool same = false;
        for (unsigned spins = 0;; ++spins) {
            const unsigned v = lane < XS_GROUP_WG ? __hip_atomic_load(hello + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : mine;
            if (__all((v >> 4) == (n + 1u))) { same = __all((v & 15u) == xcc); break; }
            if (spins > XS_SPINS) {
                if (lane == 0) {
                    L.abort = 1u;
                    __hip_atomic_store((xs_gu32*)p.abort_dev, 0x80u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store((xs_gu32*)p.tmo, 0x80u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        if (lane == 0) { L.base = n * 64u; L.local = (same && !p.force_global) ? 1u : 0u; }
    }
    // ---- x tile of the residual stream (written by the previous launch) ----
    for (int u = tid; u < K::RCU * 8; u += XS_THREADS) {
        const int r = u >> 3, q4 = u & 7;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row0 + r < M) v = *reinterpret_cast<const float4*>(p.X + (size_t)(row0 + r) * C + ct * 32 + q4 * 4);
        *reinterpret_cast<float4*>(&L.xt[r * 32 + q4 * 4]) = v;
    }
    xs_lds_barrier();
    if (L.abort) return;
    base = L.base; local = L.local != 0u;

    const int P = 5 * p.nblocks;
    const int P_run = (p.phase_limit > 0 && p.phase_limit < P) ? p.phase_limit : P;

    for (int blk = 0; blk < p.nblocks; ++blk) {
        const XBlockW& B = L.blk[blk];
        // ======================= q0: LN + FiLM -> conv1 -> depthwise 3x3 -> SimpleGate -> pooled =======================
        {
            const int ph = 5 * blk;
            if (ph >= P_run) break;
            HD_XSTAMP(0);
            if (ph > 0) { wait_phase(ph - 1); if (dead) return; }
            HD_XSTAMP(1);
            ln_issue(rs_sx, B.film_off);                                // first: the statistics barrier waits for the slowest wave's partials
            if (ph > 0 && wave == 0) wave0_w(B.w1, std::true_type());
            load_a(rs_Xb);
            // per-channel constants of the fused epilogue (weights: plain loads), parked in LDS until the epilogue: 22 values per
            // column (9 + 9 depthwise taps of the two gate halves, their biases, conv1's biases); two loads per thread
            float dwc[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int e = tid + i * XS_THREADS, k = e >> 5, cc = ct * 32 + (e & 31);
                dwc[i] = 0.f;
                if (k < 18) dwc[i] = xs_ldg_f(B.dw_w + (size_t)(k % 9) * 2 * C + cc + (k >= 9 ? C : 0));
                else if (k < 20) dwc[i] = xs_ldg_f(B.dw_b + cc + (k == 19 ? C : 0));
                else if (k < 22) dwc[i] = xs_ldg_f(B.b1 + cc + (k == 21 ? C : 0));
            }
            ln_finish();
            gemm_ln();
            after_k(ph, B.wsca, std::false_type(), ph + 1 < P_run);
            HD_XSTAMP(2);
#pragma unroll
            for (int i = 0; i < 2; ++i) { const int e = tid + i * XS_THREADS; if (e < 22 * 32) L.dwc[e] = dwc[i]; }
            to_red(std::true_type());
            constexpr int TILE_F = K::RCU * 32 * 2;
            float dw_wa[9], dw_wb[9];
            const int jc = tid & 31;
#pragma unroll
            for (int t = 0; t < 9; ++t) { dw_wa[t] = L.dwc[t * 32 + jc]; dw_wb[t] = L.dwc[(9 + t) * 32 + jc]; }
            const float dw_ba = L.dwc[18 * 32 + jc], dw_bb = L.dwc[19 * 32 + jc], b1a = L.dwc[20 * 32 + jc], b1b = L.dwc[21 * 32 + jc];
            // (1) sum the K-split partials in wave order, add conv1's bias, keep T1 in slice 0
            for (int e = tid; e < K::RCU * 32; e += XS_THREADS) {
                float va = b1a, vb = b1b;
#pragma unroll
                for (int w = 0; w < K::WK; ++w) { va += red[w * TILE_F + e]; vb += red[w * TILE_F + K::RCU * 32 + e]; }
                red[e] = va; red[K::RCU * 32 + e] = vb;
            }
            xs_lds_barrier();
            // (2) depthwise 3x3 (pad 1) on both halves, SimpleGate, gate tile (bf16) -> LDS
            {
                const int j = tid & 31, rr = tid >> 5;
                constexpr int ls = (K::S == 2) ? 1 : (K::S == 4) ? 2 : (K::S == 8) ? 3 : (K::S == 16) ? 4 : 0;
                const int p0 = rr << ls;
                const int y = (p0 & (HW - 1)) >> ls;
                const float rsum = dw_gate_row<K::S>(red + j, red + K::RCU * 32 + j, p0, y > 0, y < K::S - 1, dw_wa, dw_wb, dw_ba, dw_bb,
                                                     L.gt + p0 * 32 + j, 32, true, K::S);
                L.rs[rr * 32 + j] = rsum;
            }
            xs_lds_barrier();
            // (3) per-face average pool
            if (tid < K::FCU * 32) {
                const int f = tid >> 5, j = tid & 31;
                float sacc = 0.f;
#pragma unroll
                for (int r = 0; r < K::S; ++r) sacc += L.rs[(f * K::S + r) * 32 + j];
                L.pl[f * 32 + j] = sacc / (float)HW;
            }
            xs_lds_barrier();
            if constexpr (!K::kStoresFirst) { if (wave != 0 && ph + 1 < P_run) load_w_chunk(B.wsca, std::false_type(), std::integral_constant<int, 1>()); }
            HD_XSTAMP(3);
            if (wave == 0) {
                if (lane < K::FCU * 4) {
                    const int f = lane >> 2, q4 = lane & 3;
                    if (face0 + f < p.B) {
                        st128(rs_P16, ((face0 + f) * C + ct * 32 + q4 * 8) * 2, pack8(&L.pl[f * 32 + q4 * 8]));
                        if (p.pooled) {
                            *reinterpret_cast<float4*>(p.pooled + (size_t)(face0 + f) * C + ct * 32 + q4 * 8) = *reinterpret_cast<const float4*>(&L.pl[f * 32 + q4 * 8]);
                            *reinterpret_cast<float4*>(p.pooled + (size_t)(face0 + f) * C + ct * 32 + q4 * 8 + 4) = *reinterpret_cast<const float4*>(&L.pl[f * 32 + q4 * 8 + 4]);
                        }
                    }
                }
                HD_XSTAMP(4);
            }
            if constexpr (K::kStoresFirst) xs_lds_barrier();          // wave 0's stores are in the queue ahead of the weight requests
            if constexpr (K::kStoresFirst) second_pf(B.wsca, std::false_type(), ph + 1 < P_run, B.w3, std::false_type(), ph + 2 < P_run);
            if (wave == 0) publish(ph);
            HD_XSTAMP(5);
        }
        // ======================= q1: s = sca(pooled) ; G <- bf16(G * s) =======================
        {
            const int ph = 5 * blk + 1;
            if (ph >= P_run) break;
            HD_XSTAMP(0);
            wait_phase(ph - 1); if (dead) return;
            if (wave == 0) wave0_w(B.wsca, std::false_type());
            HD_XSTAMP(1);
            const float bsca = xs_ldg_f(B.bsca + col);
            zero_acc();
            if (wm == 0) {
                // rows = this workgroup's faces (<= 8); the other rows of the MFMA tile are zero
                xs_u32x4 pa[K::CPW];
                const int f = lane >> 3;
                const bool fv = f < K::FCU && face0 + f < p.B;
#pragma unroll
                for (int d = 0; d < K::CPW; ++d)
                    pa[d] = __builtin_amdgcn_raw_buffer_load_b128(rs_P16, ((fv ? face0 + f : 0) * C + (c0 + d) * 64 + 8 * kq) * 2, 0, 16);
                for (int i = lane; i < 24 * LDS_ROW / 16; i += 64) reinterpret_cast<uint4*>(sA + 8 * LDS_ROW)[i] = make_uint4(0, 0, 0, 0);
#pragma unroll
                for (int d = 0; d < K::CPW; ++d) {
                    *reinterpret_cast<xs_u32x4*>(sA + f * LDS_ROW + 16 * kq) = fv ? pa[d] : (xs_u32x4){0u, 0u, 0u, 0u};
                    chunk_mma(d, std::false_type());
                }
            }
            after_k(ph, B.w3, std::false_type(), ph + 1 < P_run);
            HD_XSTAMP(2);
            // partial tiles of the 8 face rows -> LDS, summed in wave order
            xs_lds_barrier();
            if (wm == 0) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int r = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
                    if (r < XS_FACES) red[wk * (XS_FACES * 32) + r * 32 + (lane & 31)] = acc[0][i];
                }
            }
            xs_lds_barrier();
            if (tid < K::FCU * 32) {
                float v = bsca;
#pragma unroll
                for (int w = 0; w < K::WK; ++w) v += red[w * (XS_FACES * 32) + tid];
                L.pl[tid] = v;
            }
            xs_lds_barrier();
            if constexpr (!K::kStoresFirst) { if (wave != 0 && ph + 1 < P_run) load_w_chunk(B.w3, std::false_type(), std::integral_constant<int, 1>()); }
            HD_XSTAMP(3);
            if (wave == 0) {
                // G' = bf16(bf16(g) * s[face]) for this workgroup's tile: 16-byte units (row, 8 columns)
                for (int u = lane; u < K::RCU * 4; u += 64) {
                    const int r = u >> 2, q4 = u & 3;
                    if (row0 + r < M) {
                        const uint4 g = *reinterpret_cast<const uint4*>(&L.gt[r * 32 + q4 * 8]);
                        float v[8];
                        unpack8(g, v);
                        const float* sp = &L.pl[(r / HW) * 32 + q4 * 8];
#pragma unroll
                        for (int i = 0; i < 8; ++i) v[i] *= sp[i];
                        st128(rs_G, ((row0 + r) * C + ct * 32 + q4 * 8) * 2, pack8(v));
                    }
                }
                if (p.S) for (int e = lane; e < K::FCU * 32; e += 64) if (face0 + (e >> 5) < p.B) p.S[(size_t)(face0 + (e >> 5)) * C + ct * 32 + (e & 31)] = L.pl[e];
                HD_XSTAMP(4);
            }
            if constexpr (K::kStoresFirst) xs_lds_barrier();          // wave 0's stores are in the queue ahead of the weight requests
            if constexpr (K::kStoresFirst) second_pf(B.w3, std::false_type(), ph + 1 < P_run, B.w4, std::true_type(), ph + 2 < P_run);
            if (wave == 0) publish(ph);
            HD_XSTAMP(5);
        }
        // ======================= q2: conv3 ; y = x + beta * (.) ; LayerNorm partials =======================
        {
            const int ph = 5 * blk + 2;
            if (ph >= P_run) break;
            HD_XSTAMP(0);
            wait_phase(ph - 1); if (dead) return;
            if (wave == 0) wave0_w(B.w3, std::false_type());
            HD_XSTAMP(1);
            load_a(rs_G);
            const float b3 = xs_ldg_f(B.b3 + col), beta = xs_ldg_f(B.beta + col);
            gemm_plain(std::false_type());
            after_k(ph, B.w4, std::true_type(), ph + 1 < P_run);
            HD_XSTAMP(2);
            to_red(std::false_type());
            constexpr int TILE_F = K::RCU * 32;
            float v[K::NIT];
#pragma unroll
            for (int it = 0; it < K::NIT; ++it) {
                const int e = it * XS_THREADS + tid;
                float s = 0.f;
#pragma unroll
                for (int w = 0; w < K::WK; ++w) s += red[w * TILE_F + e];
                v[it] = fmaf(s + b3, beta, L.xt[e]);                      // EpResidF32::store
                if (row0 + (e >> 5) >= M) v[it] = 0.f;
                L.yt[e] = v[it];
            }
            {
                float2 ms[K::NIT];
#pragma unroll
                for (int it = 0; it < K::NIT; ++it) ms[it] = halfwave_mean_m2(v[it]);
                if ((tid & 31) == kStatLane) {
#pragma unroll
                    for (int it = 0; it < K::NIT; ++it) st_out[(it * XS_THREADS + tid) >> 5] = ms[it];
                }
            }
            xs_lds_barrier();
            if constexpr (!K::kStoresFirst) { if (wave != 0 && ph + 1 < P_run) load_w_chunk(B.w4, std::true_type(), std::integral_constant<int, 1>()); }
            HD_XSTAMP(3);
            if (wave == 0) {
                for (int u = lane; u < K::RCU * 4; u += 64) {
                    const int r = u >> 2, q4 = u & 3;
                    if (row0 + r < M) st128(rs_Yb, ((row0 + r) * C + ct * 32 + q4 * 8) * 2, pack8(&L.yt[r * 32 + q4 * 8]));
                }
                for (int r = lane; r < K::RCU; r += 64) if (row0 + r < M) st64(rs_sy, ((row0 + r) * K::NT + ct) * 8, st_out[r]);
                HD_XSTAMP(4);
            }
            if constexpr (K::kStoresFirst) xs_lds_barrier();          // wave 0's stores are in the queue ahead of the weight requests
            if constexpr (K::kStoresFirst) second_pf(B.w4, std::true_type(), ph + 1 < P_run, B.w5, std::false_type(), ph + 2 < P_run);
            if (wave == 0) publish(ph);
            HD_XSTAMP(5);
        }
        // ======================= q3: LN + FiLM -> conv4 -> SimpleGate =======================
        {
            const int ph = 5 * blk + 3;
            if (ph >= P_run) break;
            HD_XSTAMP(0);
            wait_phase(ph - 1); if (dead) return;
            HD_XSTAMP(1);
            ln_issue(rs_sy, B.film_off + 2 * C);                       // first: the statistics barrier waits for the slowest wave's partials
            if (wave == 0) wave0_w(B.w4, std::true_type());
            load_a(rs_Yb);
            const float b4a = xs_ldg_f(B.b4 + col), b4b = xs_ldg_f(B.b4 + col + C);
            ln_finish();
            gemm_ln();
            after_k(ph, B.w5, std::false_type(), ph + 1 < P_run);
            HD_XSTAMP(2);
            to_red(std::true_type());
            constexpr int TILE_F = K::RCU * 32 * 2;
#pragma unroll
            for (int it = 0; it < K::NIT; ++it) {
                const int e = it * XS_THREADS + tid;
                float v1 = 0.f, v2 = 0.f;
#pragma unroll
                for (int w = 0; w < K::WK; ++w) { v1 += red[w * TILE_F + e]; v2 += red[w * TILE_F + K::RCU * 32 + e]; }
                L.gt[e] = f32_to_bf16_bits((v1 + b4a) * (v2 + b4b));
            }
            xs_lds_barrier();
            if constexpr (!K::kStoresFirst) { if (wave != 0 && ph + 1 < P_run) load_w_chunk(B.w5, std::false_type(), std::integral_constant<int, 1>()); }
            HD_XSTAMP(3);
            if (wave == 0) {
                for (int u = lane; u < K::RCU * 4; u += 64) {
                    const int r = u >> 2, q4 = u & 3;
                    if (row0 + r < M) st128(rs_G, ((row0 + r) * C + ct * 32 + q4 * 8) * 2, *reinterpret_cast<const uint4*>(&L.gt[r * 32 + q4 * 8]));
                }
                HD_XSTAMP(4);
            }
            if constexpr (K::kStoresFirst) xs_lds_barrier();          // wave 0's stores are in the queue ahead of the weight requests
            if constexpr (K::kStoresFirst) second_pf(B.w5, std::false_type(), ph + 1 < P_run, L.blk[blk + 1 < p.nblocks ? blk + 1 : blk].w1, std::true_type(), ph + 2 < P_run);
            if (wave == 0) publish(ph);
            HD_XSTAMP(5);
        }
        // ======================= q4: conv5 ; x' = y + gamma * (.) ; LayerNorm partials =======================
        {
            const int ph = 5 * blk + 4;
            if (ph >= P_run) break;
            HD_XSTAMP(0);
            wait_phase(ph - 1); if (dead) return;
            if (wave == 0) wave0_w(B.w5, std::false_type());
            HD_XSTAMP(1);
            load_a(rs_G);
            const float b5 = xs_ldg_f(B.b5 + col), gamma = xs_ldg_f(B.gamma + col);
            const bool last = (ph == P_run - 1);
            const bool gated = last && ph == P - 1 && p.outg16 != nullptr;
            gemm_plain(std::false_type());
            after_k(ph, L.blk[blk + 1 < p.nblocks ? blk + 1 : blk].w1, std::true_type(), !last);
            HD_XSTAMP(2);
            to_red(std::false_type());
            constexpr int TILE_F = K::RCU * 32;
            float v[K::NIT];
#pragma unroll
            for (int it = 0; it < K::NIT; ++it) {
                const int e = it * XS_THREADS + tid;
                float s = 0.f;
#pragma unroll
                for (int w = 0; w < K::WK; ++w) s += red[w * TILE_F + e];
                v[it] = fmaf(s + b5, gamma, L.yt[e]);
                if (row0 + (e >> 5) >= M) v[it] = 0.f;
                L.xt[e] = v[it];
            }
            if (!gated) {
                float2 ms[K::NIT];
#pragma unroll
                for (int it = 0; it < K::NIT; ++it) ms[it] = halfwave_mean_m2(v[it]);
                if ((tid & 31) == kStatLane) {
#pragma unroll
                    for (int it = 0; it < K::NIT; ++it) st_out[(it * XS_THREADS + tid) >> 5] = ms[it];
                }
            }
            xs_lds_barrier();
            if constexpr (!K::kStoresFirst) { if (wave != 0 && !last) load_w_chunk(L.blk[blk + 1].w1, std::true_type(), std::integral_constant<int, 1>()); }
            HD_XSTAMP(3);
            if (wave == 0 && !gated) {
                for (int u = lane; u < K::RCU * 4; u += 64) {
                    const int r = u >> 2, q4 = u & 3;
                    if (row0 + r < M) st128(rs_Xb, ((row0 + r) * C + ct * 32 + q4 * 8) * 2, pack8(&L.xt[r * 32 + q4 * 8]));
                }
                for (int r = lane; r < K::RCU; r += 64) if (row0 + r < M) st64(rs_sx, ((row0 + r) * K::NT + ct) * 8, st_out[r]);
            }
            if (last) {                                                   // exit: what the following launches read (kernel boundary)
                int row0e = row0;                                         // opaque here: the exit addresses are formed now, not hoisted to the kernel's start and spilled
                int tide = tid;
                asm volatile("" : "+s"(row0e), "+v"(tide));
                if (tide < K::RCU * 4) {
                    const int r = tide >> 2, q4 = tide & 3;
                    const int row = row0e + r;
                    if (row < M) {
                        const float* xv = &L.xt[r * 32 + q4 * 8];
                        float* xo = p.X + (size_t)row * C + ct * 32 + q4 * 8;
                        *reinterpret_cast<float4*>(xo) = *reinterpret_cast<const float4*>(xv);
                        *reinterpret_cast<float4*>(xo + 4) = *reinterpret_cast<const float4*>(xv + 4);
                        if (gated) {                                      // f_d * (1 + w_c + w_s) (+ idc term): the HCA conv input (hca.py:28)
                            float gv[8];
                            const float gsr = p.gate_s[row];
                            const size_t o = (size_t)row * C + ct * 32 + q4 * 8;
                            const float* gc = p.gate_c + (size_t)(row / HW) * C + ct * 32 + q4 * 8;
#pragma unroll
                            for (int i = 0; i < 8; ++i) {                 // same association as EpResidF32::store
                                const float a = p.add_src ? p.add_src[o + i] : 0.f;
                                const float g = 1.0f + gc[i] + gsr;
                                gv[i] = (xv[i] + a) * g;
                            }
                            *reinterpret_cast<uint4*>(p.outg16 + o) = pack8(gv);
                        }
                    }
                }
            } else if (wave == 0) {
                HD_XSTAMP(4);
            }
            if constexpr (K::kStoresFirst) xs_lds_barrier();          // wave 0's stores are in the queue ahead of the weight requests
            if constexpr (K::kStoresFirst) second_pf(L.blk[blk + 1 < p.nblocks ? blk + 1 : blk].w1, std::true_type(), !last, L.blk[blk + 1 < p.nblocks ? blk + 1 : blk].wsca, std::false_type(), ph + 2 < P_run);
            if (wave == 0 && !last) publish(ph);
            HD_XSTAMP(5);
        }
    }
    // the group's launch counter: every member has read it (the handshake completed before rank 0 got here)
    if (rank == 0 && tid == 0) {
        xs_gu32* gs = (xs_gu32*)(p.gstate + group * 32);
        __hip_atomic_store(gs, base / 64u + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <int C, int HW>
inline hipError_t launch_xcd_stage(const XStageP& p, hipStream_t s) {
    if (p.B < 1 || p.B > XS_GROUPS * XS_FACES || p.nblocks < 1 || p.nblocks > XS_MAXBLK) return hipErrorInvalidValue;
    hipLaunchKernelGGL((xcd_stage_kernel<C, HW>), dim3(XS_GROUPS * XS_GROUP_WG), dim3(XS_THREADS), 0, s, p);
    return hipGetLastError();
}

}  // namespace hd

#pragma clang fp contract(fast)
