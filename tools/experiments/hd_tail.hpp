// hd_tail.hpp — the middle level of the denoiser as ONE persistent launch (gfx950 only).
//
// At latent 16 the 8 middle ConditionalNAFBlocks (models/denoiser/model.py:195-197,243; block body
// models/denoiser/conditional_naf.py:108-136) run on one pixel per face: M = batch rows (<= 64), C = 2048, and
// every block is five dependent GEMMs (conv1 pair 2048->4096, sca, conv3, conv4 pair, conv5) whose inputs need the
// complete rows of the previous output.  As separate launches each of the 40 GEMMs pays a kernel boundary, a cold
// start and its weight-stream latency (7-12 us each, profiles/r01_kernel_trace_summary.txt).  Here the 40 phases
// are one launch of 256 workgroups (one per CU, 4 row groups of 16 rows x 64 column tiles of 32 channels):
//
//   * waves 4..7 of a workgroup only stream weights: each feeds the private LDS ring of "its" consumer wave with
//     1 KiB MFMA B fragments by LDS-DMA (global_load_lds_dwordx4, non-temporal), as far ahead as the ring allows —
//     across phase boundaries, so the next phase's weights arrive while the activations are being handed over;
//   * waves 0..3 consume: each owns a quarter of K.  A phase starts when the 16 activation tiles of that quarter
//     (written by 16 other workgroups in the previous phase) are flagged; a tile is stored in the lane order of the
//     v_mfma_f32_16x16x32_bf16 A fragment, so one 1 KiB load IS the fragment (no LDS staging); LayerNorm + FiLM
//     (utils.py:16-24, conditional_naf.py:114-115) are applied on the fragment in registers;
//   * wave 0 sums the four K-quarter partial tiles (fixed order: bitwise reproducible), applies the phase's
//     epilogue (depthwise centre tap + SimpleGate, SCA scale, beta/gamma residuals with the residual stream tile
//     held in registers for the whole level, LayerNorm partials) and publishes the 16 x 32 bf16 tile.
//
// Hand-off protocol (MI355X_MICROARCH.md, "Valid forms", row 1): payload stores are write-through (sc1), the storing
// wave drains them (s_waitcnt vmcnt(0)) and then ONE lane stores the workgroup's flag (sc1); a consumer wave polls
// the 16 flags it depends on with sc1 loads and loads the payload with sc1 loads only after its own poll matched.
// No fences, no grid barrier.  Flags carry an epoch (launch counter * 64 + phase + 1) that the kernel itself
// advances, so nothing has to be zeroed between launches or graph replays.  Every spin is bounded; a timeout raises
// a host-visible word and all workgroups drain out (results are then garbage and the library reports it).
// Placement (which XCD a workgroup lands on) only affects speed: the four row groups of a column tile get ids that
// are equal mod 8 so that they share an L2 under round-robin dispatch.
//
// Measured on MI355X (tools/persist_bench.hip, profiles/r02_persist_proto_*): see DESIGN.md §5.
#pragma once
#include "../../hifidiff_amd/csrc/hd_gemm.hpp"

namespace hd {

typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((address_space(1))) unsigned tl_gu32;
typedef unsigned tl_u32x4 __attribute__((ext_vector_type(4)));

constexpr int TL_C = 2048;                 // channels of the level
constexpr int TL_NT = TL_C / 32;           // 64 column tiles
constexpr int TL_RG = 4;                   // row groups of 16 rows
constexpr int TL_WG = TL_NT * TL_RG;       // 256 workgroups
constexpr int TL_KS = TL_C / 32;           // 32-deep k-steps per column group in the packed weights
constexpr int TL_RS = 24;                  // ring slots (1 KiB fragments) per consumer wave
constexpr int TL_RD = 12;                  // LDS-DMA fragments in flight per loader wave
constexpr int TL_SLABS = 3;                // rotating hand-off slabs
constexpr int TL_MAXBLK = 8;
constexpr size_t TL_SLAB_U4 = (size_t)TL_RG * TL_NT * 64;          // uint4 per activation slab (256 KiB)
constexpr size_t TL_STAT_F2 = (size_t)TL_RG * TL_NT * 16;          // float2 per statistics slab

// Weights of one block, packed for v_mfma_f32_16x16x32_bf16: [N/16 column groups][K/32][64 lanes] uint4, lane l holds
// W[k = 32*ks + 8*(l>>4) + j][col = 16*cg + (l&15)], j = 0..7 (pack_weight_frag16_kernel).
struct TailBlockW {
    const uint4 *w1, *wsca, *w3, *w4, *w5;
    const float *b1, *bsca, *b3, *b4, *b5, *beta, *gamma;
    const float *dw_c, *dw_b;              // depthwise centre tap [2C] and bias [2C] (conditional_naf.py:34-42 on a 1x1 map)
    int film_off;                          // offset of the block's 4C FiLM values: [bias_att, gain_att, bias_ffn, gain_ffn]
    int pad_;
};

struct TailP {
    int M;                                 // valid rows (faces), <= 64
    int nblocks;
    const TailBlockW* blocks;              // device array [nblocks]
    // entry (written by the previous launch, row-major)
    const float* X; const unsigned short* Xb; const float2* sx; int sx_np;      // sx: [M][sx_np] partials of 2048 / sx_np channels
    const float* film;                     // FiLM row shared by all faces
    float ln_eps;
    // exit (row-major, what the next launch reads)
    float* Xout; unsigned short* Xout16; float2* stats_out;                     // stats_out: [M][64]
    unsigned short* outg16; const float* gate_c; const float* gate_s; const float* add_src;   // HCA input (hca.py:28, model.py:245-246)
    // hand-off workspace
    uint4* act; float2* stats; unsigned* flags; unsigned* state;               // state[0]: launch counter
    unsigned* tmo;                         // host-visible timeout word
#ifdef HD_STAMPS
    unsigned long long* stamps;            // [phase][workgroup][6]
#endif
};

struct TailLds {
    uint4 ring[4][TL_RS][64];              // 96 KiB
    float red[2][2][4][16 * 32];           // [phase parity][pair half][consumer][row][col]  32 KiB
    float gb[4][2][512];                   // per consumer: FiLM gain / bias of its K quarter  16 KiB
    float2 lnp[2][4][16];                  // [parity][consumer][row] partial (mean, M2) over the consumer's 512 channels
    unsigned long long wptr[5 * TL_MAXBLK];   // weight base of every phase (read by the loader waves without touching vmcnt)
    TailBlockW blk[TL_MAXBLK];             // the blocks' pointers and offsets (one global read per launch)
    unsigned filled[4], consumed[4], red_done[4], ln_done[4], abort, pad_[3];
};

__device__ __forceinline__ unsigned tl_lds_ld(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void tl_lds_st(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
// The loader wave's own LDS words go through asm: behind an LDS-DMA the compiler puts s_waitcnt vmcnt(0) in front of every
// LDS access it can see (the DMA writes LDS), which would drain the ring's in-flight fragments at every poll.
__device__ __forceinline__ unsigned tl_lds_addr(const void* p) { return (unsigned)(unsigned long long)(const __attribute__((address_space(3))) void*)p; }
__device__ __forceinline__ unsigned tl_lds_ld_raw(unsigned addr) { unsigned v; asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(addr) : "memory"); return v; }
__device__ __forceinline__ void tl_lds_st_raw(unsigned addr, unsigned v) { asm volatile("ds_write_b32 %0, %1" :: "v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ unsigned long long tl_lds_ld64_raw(unsigned addr) { unsigned long long v; asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(addr) : "memory"); return v; }
__device__ __forceinline__ void tl_dma(const uint4* src_lane, uint4* lds_slot) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(unsigned long long)src_lane,
                                     (__attribute__((address_space(3))) void*)lds_slot, 16, 0, 2);      // aux 2 = nt
}
__device__ __forceinline__ void tl_timeout(TailLds& L, unsigned* tmo, unsigned code, int lane) {
    tl_lds_st(&L.abort, 1u);
    if (lane == 0) __hip_atomic_store((tl_gu32*)tmo, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// pointers read from the LDS copy of the block table are generic to the compiler: say that they point to global memory
// (flat loads would tie the LDS and the vector-memory counters together)
typedef __attribute__((address_space(1))) const float tl_gcf;
__device__ __forceinline__ float4 tl_ldg4(const float* p) {
    typedef float f4v __attribute__((ext_vector_type(4)));
    const f4v v = *reinterpret_cast<__attribute__((address_space(1))) const f4v*>((unsigned long long)p);
    return make_float4(v.x, v.y, v.z, v.w);
}
constexpr unsigned TL_SPINS = 1u << 21;   // x (>= 0.3 us per poll): over half a second before a wait gives up

// phase q of a block: 0 conv1 (LN, pair, depthwise centre tap + SimpleGate), 1 sca, 2 conv3 (+beta residual, stats),
// 3 conv4 (LN, pair, SimpleGate), 4 conv5 (+gamma residual, stats)
__device__ __forceinline__ const uint4* tl_weights(const TailBlockW& b, int q) {
    return q == 0 ? b.w1 : q == 1 ? b.wsca : q == 2 ? b.w3 : q == 3 ? b.w4 : b.w5;
}

__global__ __launch_bounds__(512) void mid_tail_kernel(const TailP p) {
    __shared__ __attribute__((aligned(16))) TailLds L;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lin = blockIdx.x, xcd = lin & 7, jj = lin >> 3;
    const int ct = (jj >> 2) * 8 + xcd, rg = jj & 3;
    if (tid < 4) { L.filled[tid] = 0u; L.consumed[tid] = 0u; L.red_done[tid] = 0u; L.ln_done[tid] = 0u; }
    if (tid == 0) L.abort = 0u;
    const unsigned launch = __hip_atomic_load((const tl_gu32*)p.state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned epoch0 = launch * 64u + 1u;                       // flag value of phase ph: epoch0 + ph (never 0)
    const int P = 5 * p.nblocks;
    if (tid < p.nblocks) L.blk[tid] = p.blocks[tid];
    if (tid >= 64 && tid < 64 + P) { const int ph = tid - 64; L.wptr[ph] = (unsigned long long)tl_weights(p.blocks[ph / 5], ph % 5); }
    __syncthreads();

    if (wave >= 4) {
        // =============================================== weight loader ===============================================
        const int c = wave - 4;
        unsigned issued = 0; int slot = 0;
        const unsigned a_consumed = tl_lds_addr(&L.consumed[c]), a_filled = tl_lds_addr(&L.filled[c]), a_abort = tl_lds_addr(&L.abort);
        for (int ph = 0; ph < P; ++ph) {
            const int q = ph % 5;
            const uint4* W = (const uint4*)tl_lds_ld64_raw(tl_lds_addr(&L.wptr[ph]));
            const int nf = (q == 0 || q == 3) ? 4 : 2;
            for (int ks = 0; ks < 16; ++ks) {
                for (unsigned spins = 0; (int)(issued + nf - tl_lds_ld_raw(a_consumed)) > TL_RS; ++spins) {
                    if (tl_lds_ld_raw(a_abort)) return;
                    if (spins > TL_SPINS) { tl_lds_st_raw(a_abort, 1u); if (lane == 0) __hip_atomic_store((tl_gu32*)p.tmo, 0x1000u + ph, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); return; }
                    __builtin_amdgcn_s_sleep(1);
                }
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    if (f < nf) {
                        const int cg = (ct + (f >> 1) * TL_NT) * 2 + (f & 1);
                        tl_dma(W + ((size_t)cg * TL_KS + 16 * c + ks) * 64 + lane, &L.ring[c][slot][0]);
                        slot = (slot + 1 == TL_RS) ? 0 : slot + 1;
                    }
                }
                issued += nf;
                if (issued > TL_RD) {
                    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");                  // = TL_RD
                    if (lane == 0) tl_lds_st_raw(a_filled, issued - TL_RD);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) tl_lds_st_raw(a_filled, issued);
        return;
    }

    // ================================================== consumer ==================================================
    const int c = wave;
    const int row = lane & 15, kq = lane >> 4;                         // A fragment: row, 8 k at 8*kq; epilogue: row, 8 columns at 8*kq
    const int grow = 16 * rg + row;                                    // row of the level (face)
    const bool row_ok = grow < p.M;
    const __amdgpu_buffer_rsrc_t act_rs = __builtin_amdgcn_make_buffer_rsrc(p.act, 0, (int)(TL_SLABS * TL_SLAB_U4 * 16), 0x00020000);
    const __amdgpu_buffer_rsrc_t st_rs = __builtin_amdgcn_make_buffer_rsrc(p.stats, 0, (int)(TL_SLABS * TL_STAT_F2 * 8), 0x00020000);
    unsigned taken = 0, avail = 0; int slot = 0;
    // epilogue wave: the workgroup's tile of the residual stream lives in registers for the whole level
    float xs[8], ys[8], gq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { xs[e] = 0.f; ys[e] = 0.f; gq[e] = 0.f; }
    const int col0 = 32 * ct + 8 * kq;                                 // first of this lane's 8 output columns
    if (c == 0 && row_ok) {
        const float4 a = *reinterpret_cast<const float4*>(p.X + (size_t)grow * TL_C + col0);
        const float4 b = *reinterpret_cast<const float4*>(p.X + (size_t)grow * TL_C + col0 + 4);
        xs[0] = a.x; xs[1] = a.y; xs[2] = a.z; xs[3] = a.w; xs[4] = b.x; xs[5] = b.y; xs[6] = b.z; xs[7] = b.w;
    }

    for (int ph = 0; ph < P; ++ph) {
        const int par = ph & 1, blk = ph / 5, q = ph - 5 * blk;
        const TailBlockW& B = L.blk[blk];
        const bool pair = (q == 0 || q == 3), ln = pair;
        const int nf = pair ? 4 : 2;
        const int s_in = (ph + TL_SLABS - 1) % TL_SLABS, s_out = ph % TL_SLABS;
#ifdef HD_STAMPS
        unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
        const bool stamp = p.stamps && tid == 0;
        if (stamp) t0 = __builtin_amdgcn_s_memrealtime();
#endif
        // ---- FiLM gain / bias of this wave's K quarter (constant during the step: no dependency on the hand-off) ----
        float4 fg0 = make_float4(0, 0, 0, 0), fg1 = fg0, fb0 = fg0, fb1 = fg0;   // parked in registers until the hand-off wait is over
        if (ln) {
            const float* f = p.film + B.film_off + (q == 0 ? 0 : 2 * TL_C);      // [bias | gain] of this LayerNorm
            fb0 = tl_ldg4(f + 512 * c + 4 * lane); fb1 = tl_ldg4(f + 512 * c + 256 + 4 * lane);
            fg0 = tl_ldg4(f + TL_C + 512 * c + 4 * lane); fg1 = tl_ldg4(f + TL_C + 512 * c + 256 + 4 * lane);
        }
        // ---- epilogue constants of this phase (wave 0): requested before the wait ----
        float cb[8], cb2[8], cw[8], cw2[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { cb[e] = 0.f; cb2[e] = 0.f; cw[e] = 1.f; cw2[e] = 1.f; }
        if (c == 0) {
            auto ld8 = [&](const float* src, float* dst) {
                const float4 a = tl_ldg4(src + col0), b = tl_ldg4(src + col0 + 4);
                dst[0] = a.x; dst[1] = a.y; dst[2] = a.z; dst[3] = a.w; dst[4] = b.x; dst[5] = b.y; dst[6] = b.z; dst[7] = b.w;
            };
            if (q == 0) {                                               // g = (A1 + wa*acc1) * (A2 + wb*acc2), A = b2 + w*b1 (EpDwGate1)
                float b1a[8], b1b[8], dba[8], dbb[8];
                ld8(B.b1, b1a); ld8(B.b1 + TL_C, b1b); ld8(B.dw_c, cw); ld8(B.dw_c + TL_C, cw2); ld8(B.dw_b, dba); ld8(B.dw_b + TL_C, dbb);
#pragma unroll
                for (int e = 0; e < 8; ++e) { cb[e] = dba[e] + cw[e] * b1a[e]; cb2[e] = dbb[e] + cw2[e] * b1b[e]; }
            } else if (q == 1) { ld8(B.bsca, cb); }
            else if (q == 2) { ld8(B.b3, cb); ld8(B.beta, cw); }
            else if (q == 3) { ld8(B.b4, cb); ld8(B.b4 + TL_C, cb2); }
            else { ld8(B.b5, cb); ld8(B.gamma, cw); }
        }
        // ---- wait for the 16 tiles of this wave's K quarter ----
        if (ph > 0) {
            const tl_gu32* fl = (const tl_gu32*)(p.flags + (size_t)(ph - 1) * TL_WG);
            const unsigned want = epoch0 + (unsigned)(ph - 1);
            for (unsigned spins = 0;; ++spins) {
                unsigned v = want;
                if (lane < 16) {
                    const int t = 16 * c + lane;                        // producer of tile t of row group rg
                    v = __hip_atomic_load(fl + ((((t >> 3) * 4 + rg) << 3) | (t & 7)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (__all(v == want)) break;
                if (tl_lds_ld(&L.abort)) return;
                if (spins > TL_SPINS) { tl_timeout(L, p.tmo, 0x2000u + ph, lane); return; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
#ifdef HD_STAMPS
        if (stamp) t1 = __builtin_amdgcn_s_memrealtime();
#endif
        // ---- A fragments (and, for a LayerNorm phase, the statistics partials of the 16 tiles) ----
        tl_u32x4 a[16];
        float2 sp[4];
        if (ph == 0) {                                                  // entry: row-major bf16 copy of the residual stream
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                a[i] = (tl_u32x4){0u, 0u, 0u, 0u};
                if (row_ok) a[i] = *reinterpret_cast<const tl_u32x4*>(p.Xb + (size_t)grow * TL_C + 32 * (16 * c + i) + 8 * kq);
            }
            // partials of this lane's share of the row: sx_np partials per row, this wave owns a quarter, this lane a quarter of that
            const int per = p.sx_np >> 4;                               // partials per (wave, lane group): 4 (sx_np 64) or 1 (sx_np 16)
            float m = 0.f, m2 = 0.f;
            if (row_ok) {
                const float2* sr = p.sx + (size_t)grow * p.sx_np + (4 * c + kq) * per;
                const float2 v0 = sr[0], v1 = per == 4 ? sr[1] : v0, v2 = per == 4 ? sr[2] : v0, v3 = per == 4 ? sr[3] : v0;
                m = 0.25f * ((v0.x + v1.x) + (v2.x + v3.x));
                const float cnt = (float)(TL_C / p.sx_np);
                const float d0 = v0.x - m, d1 = v1.x - m, d2 = v2.x - m, d3 = v3.x - m;
                m2 = per == 4 ? ((v0.y + v1.y) + (v2.y + v3.y)) + cnt * ((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3)) : v0.y;
            }
            sp[0] = make_float2(m, m2);
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i)
                a[i] = __builtin_amdgcn_raw_buffer_load_b128(act_rs, (int)(((s_in * TL_RG + rg) * TL_NT + 16 * c + i) * 64 + lane) * 16, 0, 16);
            if (ln) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {                           // tiles 16c + 4kq + i, this lane's row
                    const unsigned long long raw = __builtin_bit_cast(unsigned long long,
                        __builtin_amdgcn_raw_buffer_load_b64(st_rs, (int)((((s_in * TL_RG + rg) * TL_NT + 16 * c + 4 * kq + i) * 16 + row) * 8), 0, 16));
                    sp[i] = __builtin_bit_cast(float2, raw);
                }
                float m = 0.25f * ((sp[0].x + sp[1].x) + (sp[2].x + sp[3].x)), m2 = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) { const float d = sp[i].x - m; m2 += sp[i].y + 32.f * d * d; }
                sp[0] = make_float2(m, m2);
            }
        }
        float mu = 0.f, rstd = 0.f;
        if (ln) {
            *reinterpret_cast<float4*>(&L.gb[c][0][4 * lane]) = fg0; *reinterpret_cast<float4*>(&L.gb[c][0][256 + 4 * lane]) = fg1;
            *reinterpret_cast<float4*>(&L.gb[c][1][4 * lane]) = fb0; *reinterpret_cast<float4*>(&L.gb[c][1][256 + 4 * lane]) = fb1;
            // (mean, M2) of this lane group's 128 channels -> the wave's 512 (lane groups kq) -> the row's 2048 (4 consumer waves)
            float m = sp[0].x, m2 = sp[0].y;
            float cnt = 128.f;
#pragma unroll
            for (int o = 16; o <= 32; o <<= 1) {                        // lanes l ^ 16, l ^ 32: equal counts
                const float mo = __shfl_xor(m, o), m2o = __shfl_xor(m2, o);
                const float mm = 0.5f * (m + mo), d = m - mm;
                m2 = m2 + m2o + 2.f * cnt * d * d;
                m = mm; cnt *= 2.f;
            }
            if (kq == 0) L.lnp[par][c][row] = make_float2(m, m2);
            asm volatile("" ::: "memory");
            if (lane == 0) tl_lds_st(&L.ln_done[c], (unsigned)(ph + 1));
            for (unsigned spins = 0;; ++spins) {
                if (tl_lds_ld(&L.ln_done[0]) >= (unsigned)(ph + 1) && tl_lds_ld(&L.ln_done[1]) >= (unsigned)(ph + 1) &&
                    tl_lds_ld(&L.ln_done[2]) >= (unsigned)(ph + 1) && tl_lds_ld(&L.ln_done[3]) >= (unsigned)(ph + 1)) break;
                if (spins > TL_SPINS) { tl_timeout(L, p.tmo, 0x3000u + ph, lane); return; }
                if ((spins & 255) == 255 && tl_lds_ld(&L.abort)) return;
            }
            asm volatile("" ::: "memory");
            const float2 q0 = L.lnp[par][0][row], q1 = L.lnp[par][1][row], q2 = L.lnp[par][2][row], q3 = L.lnp[par][3][row];
            const float mean = 0.25f * ((q0.x + q1.x) + (q2.x + q3.x));
            const float d0 = q0.x - mean, d1 = q1.x - mean, d2 = q2.x - mean, d3 = q3.x - mean;
            const float M2 = ((q0.y + q1.y) + (q2.y + q3.y)) + 512.f * ((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3));
            rstd = __frsqrt_rn(M2 * (1.0f / (float)TL_C) + p.ln_eps);
            mu = -mean * rstd;                                           // x_hat = fma(x, rstd, -mean * rstd)
        }
        // ---- K loop: 16 steps of 32, B fragments from the ring ----
        f32x4_t acc[4];
#pragma unroll
        for (int f = 0; f < 4; ++f) acc[f] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            // ring availability is checked for two k-steps at a time (<= 8 fragments; the loader publishes a fragment once
            // TL_RD = 12 younger ones are in flight, so a demand of more than TL_RS - TL_RD = 12 could never be met)
            if ((ks & 1) == 0) {
                const unsigned need2 = taken + 2 * nf;
                if ((int)(avail - need2) < 0) {
                    for (unsigned spins = 0;; ++spins) {
                        avail = tl_lds_ld(&L.filled[c]);
                        if ((int)(avail - need2) >= 0) break;
                        if (spins > (TL_SPINS << 3)) { tl_timeout(L, p.tmo, 0x4000u + ph, lane); return; }
                        if ((spins & 255) == 255 && tl_lds_ld(&L.abort)) return;
                    }
                    asm volatile("" ::: "memory");
                }
            }
            const unsigned need = taken + nf;
            uint4 bfr[4];
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                if (f < nf) { bfr[f] = L.ring[c][slot][lane]; slot = (slot + 1 == TL_RS) ? 0 : slot + 1; }
            }
            bf16x8_t af;
            if (ln) {
                float v[8];
                unpack8(make_uint4(a[ks].x, a[ks].y, a[ks].z, a[ks].w), v);
                const float4 g0 = *reinterpret_cast<const float4*>(&L.gb[c][0][32 * ks + 8 * kq]), g1 = *reinterpret_cast<const float4*>(&L.gb[c][0][32 * ks + 8 * kq + 4]);
                const float4 b0 = *reinterpret_cast<const float4*>(&L.gb[c][1][32 * ks + 8 * kq]), b1 = *reinterpret_cast<const float4*>(&L.gb[c][1][32 * ks + 8 * kq + 4]);
                v[0] = fmaf(fmaf(v[0], rstd, mu), g0.x, b0.x); v[1] = fmaf(fmaf(v[1], rstd, mu), g0.y, b0.y);
                v[2] = fmaf(fmaf(v[2], rstd, mu), g0.z, b0.z); v[3] = fmaf(fmaf(v[3], rstd, mu), g0.w, b0.w);
                v[4] = fmaf(fmaf(v[4], rstd, mu), g1.x, b1.x); v[5] = fmaf(fmaf(v[5], rstd, mu), g1.y, b1.y);
                v[6] = fmaf(fmaf(v[6], rstd, mu), g1.z, b1.z); v[7] = fmaf(fmaf(v[7], rstd, mu), g1.w, b1.w);
                af = __builtin_bit_cast(bf16x8_t, pack8(v));
            } else {
                af = __builtin_bit_cast(bf16x8_t, a[ks]);
            }
#pragma unroll
            for (int f = 0; f < 4; ++f)
                if (f < nf) acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, __builtin_bit_cast(bf16x8_t, bfr[f]), acc[f], 0, 0, 0);
            taken = need;
            if (ks & 1) {
                asm volatile("" ::: "memory");
                if (lane == 0) tl_lds_st(&L.consumed[c], taken);        // LDS executes a wave's operations in order: the reads above are done
            }
        }
#ifdef HD_STAMPS
        if (stamp) t2 = __builtin_amdgcn_s_memrealtime();
#endif
        // ---- partial tiles: fragment f = (pair half, column half): col = 16 * (f & 1) + (lane & 15), rows 4 * (lane >> 4) + i ----
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            if (f < nf) {
#pragma unroll
                for (int i = 0; i < 4; ++i) L.red[par][f >> 1][c][(4 * kq + i) * 32 + 16 * (f & 1) + row] = acc[f][i];
            }
        }
        asm volatile("" ::: "memory");
        if (lane == 0) tl_lds_st(&L.red_done[c], (unsigned)(ph + 1));
        if (c != 0) continue;

        // ================================ epilogue + publish (wave 0) ================================
        for (unsigned spins = 0;; ++spins) {
            if (tl_lds_ld(&L.red_done[1]) >= (unsigned)(ph + 1) && tl_lds_ld(&L.red_done[2]) >= (unsigned)(ph + 1) &&
                tl_lds_ld(&L.red_done[3]) >= (unsigned)(ph + 1)) break;
            if (spins > (TL_SPINS << 3)) { tl_timeout(L, p.tmo, 0x5000u + ph, lane); return; }
            if ((spins & 255) == 255 && tl_lds_ld(&L.abort)) return;
        }
        asm volatile("" ::: "memory");
#ifdef HD_STAMPS
        if (stamp) t3 = __builtin_amdgcn_s_memrealtime();
#endif
        float v1[8], v2[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { v1[e] = 0.f; v2[e] = 0.f; }
#pragma unroll
        for (int w = 0; w < 4; ++w) {                                    // K quarters in fixed order
            const float4 x0 = *reinterpret_cast<const float4*>(&L.red[par][0][w][row * 32 + 8 * kq]);
            const float4 x1 = *reinterpret_cast<const float4*>(&L.red[par][0][w][row * 32 + 8 * kq + 4]);
            v1[0] += x0.x; v1[1] += x0.y; v1[2] += x0.z; v1[3] += x0.w; v1[4] += x1.x; v1[5] += x1.y; v1[6] += x1.z; v1[7] += x1.w;
            if (pair) {
                const float4 y0 = *reinterpret_cast<const float4*>(&L.red[par][1][w][row * 32 + 8 * kq]);
                const float4 y1 = *reinterpret_cast<const float4*>(&L.red[par][1][w][row * 32 + 8 * kq + 4]);
                v2[0] += y0.x; v2[1] += y0.y; v2[2] += y0.z; v2[3] += y0.w; v2[4] += y1.x; v2[5] += y1.y; v2[6] += y1.z; v2[7] += y1.w;
            }
        }
        float o[8];
        bool stats = false;
        if (q == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float g = fmaf(cw[e], v1[e], cb[e]) * fmaf(cw2[e], v2[e], cb2[e]); o[e] = g; }
        } else if (q == 1) {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = gq[e] * (v1[e] + cb[e]);                     // bf16(G) * sca(G)  (conditional_naf.py:119)
        } else if (q == 2) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { ys[e] = xs[e] + (v1[e] + cb[e]) * cw[e]; o[e] = ys[e]; }   // y = inp + x * beta
            stats = true;
        } else if (q == 3) {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (v1[e] + cb[e]) * (v2[e] + cb2[e]);          // SimpleGate (utils.py:57-60)
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) { xs[e] = ys[e] + (v1[e] + cb[e]) * cw[e]; o[e] = xs[e]; }   // out = y + x * gamma
            stats = true;
        }
        const uint4 ob = pack8(o);
        if (q == 0) unpack8(ob, gq);                                     // the rounded G is what conv3's input is built from
        const bool last = (ph == P - 1);
        if (!last) {
            __builtin_amdgcn_raw_buffer_store_b128((tl_u32x4){ob.x, ob.y, ob.z, ob.w}, act_rs, (int)(((s_out * TL_RG + rg) * TL_NT + ct) * 64 + lane) * 16, 0, 16);
            if (stats) {                                                 // LayerNorm partial of this tile's 32 columns per row
                float s1 = ((o[0] + o[1]) + (o[2] + o[3])) + ((o[4] + o[5]) + (o[6] + o[7]));
                s1 += __shfl_xor(s1, 16); s1 += __shfl_xor(s1, 32);
                const float mean = s1 * (1.0f / 32.0f);
                float s2 = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = o[e] - mean; s2 = fmaf(d, d, s2); }
                s2 += __shfl_xor(s2, 16); s2 += __shfl_xor(s2, 32);
                if (kq == 0) {
                    const unsigned long long raw = __builtin_bit_cast(unsigned long long, make_float2(mean, s2));
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(__attribute__((ext_vector_type(2))) unsigned, raw), st_rs,
                                                          (int)((((s_out * TL_RG + rg) * TL_NT + ct) * 16 + row) * 8), 0, 16);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_store((tl_gu32*)(p.flags + (size_t)ph * TL_WG + lin), epoch0 + (unsigned)ph, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            // ---- exit: what the following launches read, row-major (kernel boundary makes it visible) ----
            float s1 = ((o[0] + o[1]) + (o[2] + o[3])) + ((o[4] + o[5]) + (o[6] + o[7]));
            s1 += __shfl_xor(s1, 16); s1 += __shfl_xor(s1, 32);
            const float mean = s1 * (1.0f / 32.0f);
            float s2 = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = o[e] - mean; s2 = fmaf(d, d, s2); }
            s2 += __shfl_xor(s2, 16); s2 += __shfl_xor(s2, 32);
            if (row_ok) {
                float* xo = p.Xout + (size_t)grow * TL_C + col0;
                *reinterpret_cast<float4*>(xo) = make_float4(o[0], o[1], o[2], o[3]);
                *reinterpret_cast<float4*>(xo + 4) = make_float4(o[4], o[5], o[6], o[7]);
                if (p.Xout16) *reinterpret_cast<uint4*>(p.Xout16 + (size_t)grow * TL_C + col0) = ob;
                if (p.stats_out && kq == 0) p.stats_out[(size_t)grow * TL_NT + ct] = make_float2(mean, s2);
                if (p.outg16) {                                          // f_d * (1 + w_c + w_s) (+ idc term): the HCA conv input
                    float gv[8];
                    const float gs = 1.0f + p.gate_s[grow];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float ad = p.add_src ? p.add_src[(size_t)grow * TL_C + col0 + e] : 0.f;
                        gv[e] = (o[e] + ad) * (gs + p.gate_c[(size_t)grow * TL_C + col0 + e]);
                    }
                    *reinterpret_cast<uint4*>(p.outg16 + (size_t)grow * TL_C + col0) = pack8(gv);
                }
            }
            if (lin == 0 && lane == 0) __hip_atomic_store((tl_gu32*)p.state, launch + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#ifdef HD_STAMPS
        if (stamp) {
            t4 = __builtin_amdgcn_s_memrealtime();
            unsigned long long* sq = p.stamps + ((size_t)ph * TL_WG + lin) * 6;
            sq[0] = t0; sq[1] = t1; sq[2] = t2; sq[3] = t3; sq[4] = t4;
        }
#endif
    }
}

// [N][K] fp32 conv / linear weight (1x1 or centre tap) -> the 16x16x32 B-fragment order above
struct PackF16P { const float* src; uint4* dst; int N, K, KH, KW; };
__global__ void pack_weight_frag16_kernel(const PackF16P p) {
    const int ksteps = p.K >> 5;
    const size_t total = (size_t)(p.N >> 4) * ksteps * 64;
    const int taps = p.KH * p.KW, ctr = (p.KH >> 1) * p.KW + (p.KW >> 1);
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int lane = (int)(e & 63);
        const size_t tk = e >> 6;
        const int ks = (int)(tk % ksteps), cg = (int)(tk / ksteps);
        const int n = cg * 16 + (lane & 15);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = ks * 32 + 8 * (lane >> 4) + j;
            v[j] = p.src[((size_t)n * p.K + k) * taps + ctr];
        }
        p.dst[e] = pack8(v);
    }
}

inline hipError_t launch_mid_tail(const TailP& p, hipStream_t s) {
    if (p.M < 1 || p.M > 64 || p.nblocks < 1 || p.nblocks > TL_MAXBLK) return hipErrorInvalidValue;
    hipLaunchKernelGGL(mid_tail_kernel, dim3(TL_WG), dim3(512), 0, s, p);
    return hipGetLastError();
}

}  // namespace hd
