// hd_mid.hpp — MEASURED NEGATIVE (profiles/r05_mid_stage_bench.txt: 10.2 us per phase against 7.75 us for one launch per GEMM); kept as the
// record of VERDICT r04 item 1, compiled only by tools/mid_stage_bench.hip, not part of the library.
//
// The eight middle-level ConditionalNAFBlocks (C = 2048, one pixel per face) as ONE persistent launch over the whole chip
// (gfx950 only).  models/denoiser/model.py:195-197,243 (middle_blks), models/denoiser/conditional_naf.py:108-136 (the block).
//
// At the middle level a GEMM is [B <= 64 rows] x [2048 x N]: 8-17 MB of weights that every row needs, so the weights must be spread
// over all 256 CUs (each weight byte enters exactly one CU) and every phase ends in a chip-wide exchange of its 64 x 2048 outputs.
// One launch per GEMM pays for that with a kernel boundary and a cold start per phase and with 192-320 KB of operands per CU
// (profiles/r04_kernel_table.txt: 6.5 / 9.7 us per launch, 40 launches).  Here:
//   * Decomposition: 64 clusters of 4 workgroups, the members of a cluster on one XCD (blocks b, b + 8, b + 16, b + 24 of a group of
//     32: checked at run time).  A cluster owns one 32-channel output tile of every GEMM; member m takes K quarter m (k-steps 16 m ..
//     16 m + 15 of 64): 32 / 64 KB of weights (nobody else reads them) + 64 KB of activations per workgroup and phase.
//   * Waves as in hd_xcd2.hpp: 4 compute waves, wave rb owns row block rb (16 faces) -- MFMA 16x16x32 with the weights as the A
//     operand and the activations as B, results already in the lane layout the next GEMM's B operand wants; 4 loader waves stream the
//     weights of the phases ahead into a 128 KB LDS ring by LDS-DMA, independent of the phase structure (weights do not depend on
//     activations, so the ring is full while a wave waits for a hand-off).
//   * Split-K exchange inside the cluster (one XCD, through its L2): member m FINISHES row block m.  Wave rb of member m != rb stores its
//     fp32 partial tile of row block rb (1-4 KB, contiguous), drains, raises its flag and is done with the phase; wave m of member m
//     polls the three flags, adds the four partials in member order (fixed order: reproducible) and runs the epilogue of all 16 rows.
//   * Chip-wide hand-off of a finished tile (16 rows x 32 channels): write-through (sc1) stores of one whole 1 KiB fragment of the
//     consumers' B operand, the storing wave's s_waitcnt vmcnt(0), its own sc1 flag -- one flag per (row block, tile): 256 in all;
//     consumers poll exactly the flags of their producers (the 16 tiles of their K quarter; LayerNorm phases all 64 of the row block,
//     the statistics come from every tile) and read with sc1 loads (MI355X_MICROARCH.md, "Valid forms", row 1, per wave).
//     (First form, profiles/r05_mid_stage_bench.txt: every member finishing 4 rows of every row block -- 1024 flags, 64-byte store runs,
//     every wave polling twice per phase: 12.3 us per phase.)
//   * A face is one pixel: the depthwise 3x3 (pad 1) is its centre tap, the SCA pool is the gate itself (conditional_naf.py:116-119).
// LayerNorm2d statistics (utils.py:16-24): producers emit fp32 (mean, M2) of their 32 channels per row (two-pass), consumers merge
// the 64 partials of a row (equal counts, fixed order).  Rounding points are those of the per-GEMM launches (bf16 MFMA operands, fp32
// everything else); the summation order differs (four K chains of 512 instead of eight slices of 256), so the stage is checked block by
// block against the oracle on its own inputs (tests/test_gpu_parity.py), not bit for bit against the launches.
// Every spin is bounded; giving up raises the abort words (hd_xcd.hpp) and every wave of the workgroup leaves.  All 256 workgroups
// must be resident (one per CU).
#pragma once
#include "../hifidiff_amd/csrc/hd_xcd2.hpp"

namespace hd {

// the middle level (C = 2048, one pixel per face) as one chip-wide persistent launch 
struct MStageP {
    int B, nblocks;
    const XBlockW* blocks;                 // device array [nblocks]; weights in the 16x16x32 packing (pack_weight16_kernel)
    float* X; unsigned short* Xb; const float2* sx;      // entry (standard layouts, sx: [B][64] partials of 32 channels) and exit (X, Xb)
    uint4 *hP, *hG, *hY, *hG2, *hX;        // hand-off, fragment order: [4 row blocks][64 k-steps][64 lanes] uint4 (256 KB each); one buffer per phase of a block
    float2 *hsx, *hsy;                     // hand-off statistics [4 row blocks][64 tiles][16 rows]: (mean, M2) of 32 channels
    uint4* xbuf;                           // split-K exchange inside a cluster: [64 tiles][4 row blocks][4 slots][4 members][4 registers][64 lanes] uint4 (16 MB)
    // introspection copies in the standard layouts (written by the phase a phase_limit stops at; may be NULL)
    unsigned short *dG, *dYb; float *dpooled, *dS;
    const float* film; float ln_eps;
    unsigned short* outg16; const float *gate_c, *gate_s, *add_src;
    unsigned *flags, *xflags, *hello, *gstate;           // [4 row blocks][64 tiles] | [64][4][4 members] | [64][4] | [64][4] words
    unsigned *tmo, *abort_dev; int test_abort;
    int phase_limit, force_global;
#ifdef HD_STAMPS
    unsigned long long* stamps;            // [phase][workgroup][8] of compute wave 0
    int dbg_no_w;                          // timing-only what-if (results are garbage): the loaders count their steps but move no weights
#endif
};


struct MidCfg {
    static constexpr int C = 2048, KS = 64, KSW = 16, NT = 64;          // k-steps of 32, per member, 32-channel tiles
    static constexpr int NCW = 4, NLW = 4, THREADS = 64 * (NCW + NLW);
    static constexpr int RING = 128;                                    // weight ring, 1 KiB fragments: two pair phases / four plain ones
    static constexpr int GBF = 4;                                       // fragments of the member's quarter of a FiLM gain | bias row (512 + 512 floats)
    static constexpr int DWF = 1;                                       // fragment of the per-channel constants of q0
    static constexpr int DL = 8;                                        // LDS-DMA instructions a loader wave keeps in flight
    static constexpr int NP = 64;                                       // statistics partials per row (32 channels each)
};

struct MidLds {
    uint4 ring[MidCfg::RING][64];
    float gb[2 * 512];                                                  // gain [512] | bias [512] of this member's K quarter
    float dwc[256];                                                     // [8][32]: centre taps a / b, depthwise biases a / b, conv1 biases a / b, (2 unused rows)
    XBlockW blk[XS_MAXBLK];
    unsigned landed[MidCfg::NLW], consumed[MidCfg::NCW], gbdone[MidCfg::NCW];
    unsigned base, local, abort, pad_;
};

#ifdef HD_STAMPS
#define HD_MSTAMP(i) do { if (p.stamps && cw == 0 && lane == 0) p.stamps[((size_t)ph * 256 + blockIdx.x) * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define HD_MSTAMP(i) do { } while (0)
#endif

__global__ __launch_bounds__(MidCfg::THREADS) void mid_stage_kernel(const MStageP p) {
    typedef MidCfg K;
    constexpr int C = K::C;
    __shared__ __attribute__((aligned(16))) MidLds L;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xcd = blockIdx.x & 7, j32 = blockIdx.x >> 3;               // blocks b and b + 8 share an XCD under round-robin dispatch (speed only)
    const int ct = xcd * 8 + (j32 >> 2), member = j32 & 3;               // cluster = output tile (32 channels); member = K quarter
    const int M = p.B;

    // ---- block table, LDS words, start-of-launch handshake of the cluster (placement, launch epoch) ----
    {
        const unsigned* src = reinterpret_cast<const unsigned*>(p.blocks);
        unsigned* dst = reinterpret_cast<unsigned*>(L.blk);
        for (int i = tid; i < p.nblocks * (int)(sizeof(XBlockW) / 4); i += K::THREADS) dst[i] = src[i];
        if (tid < K::NLW) L.landed[tid] = 0u;
        if (tid < K::NCW) { L.consumed[tid] = 0u; L.gbdone[tid] = 0u; }
        if (tid == 0) L.abort = __hip_atomic_load((xs_gu32*)p.abort_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // an earlier stage of this call gave up
    }
    if (wave == 0) {
        xs_gu32* gs = (xs_gu32*)(p.gstate + ct * 4);
        xs_gu32* hello = (xs_gu32*)(p.hello + ct * 4);
        const unsigned n = __hip_atomic_load(gs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned xcc = xs_xcc_id();
        const unsigned mine = ((n + 1u) << 4) | xcc;
        if (lane == 0) __hip_atomic_store(hello + member, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool same = false, gaveup = false;
        for (unsigned spins = 0;; ++spins) {
            const unsigned v = lane < 4 ? __hip_atomic_load(hello + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : mine;
            if (__all((v >> 4) == (n + 1u))) { same = __all((v & 15u) == xcc); break; }
            if (spins > XS_SPINS) { gaveup = true; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        if (lane == 0) {
            L.base = n * 64u; L.local = (same && !p.force_global) ? 1u : 0u;
            if (gaveup) {
                L.abort = 1u;
                __hip_atomic_store((xs_gu32*)p.abort_dev, 0x80u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store((xs_gu32*)p.tmo, 0x80u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
    __syncthreads();
    if (L.abort) return;
    const unsigned base = L.base;
    const bool local = L.local != 0u;
    const int P = 5 * p.nblocks;
    const int P_run = (p.phase_limit > 0 && p.phase_limit < P) ? p.phase_limit : P;
    const unsigned a_abort = x2_lds_addr(&L.abort);

    if (wave >= K::NCW) {
        // =========================================== weight loader waves ===========================================
        // Loader lw issues the steps j = lw (mod 4) of every phase (step j: the fragments of k-step 16 member + j, both channel
        // blocks, both gate halves), as far ahead as the ring allows; loader 0 also brings the member's quarter of the FiLM row of a
        // LayerNorm phase (and the per-channel constants of q0) into their own LDS regions.
        const int lw = wave - K::NCW;
        unsigned issued = 0;                                          // DMA instructions of this wave
        unsigned cum = 0;                                             // ring fragments of all earlier steps (every loader counts the same)
        const unsigned a_landed = x2_lds_addr(&L.landed[lw]), a_cons = x2_lds_addr(&L.consumed[0]), a_gbdone = x2_lds_addr(&L.gbdone[0]);
        int ln_seen = 0;
        for (int ph = 0; ph < P_run; ++ph) {
            const int blk = ph / 5, q = ph - 5 * blk;
            const bool pair = (q == 0 || q == 3);
            const int nh = pair ? 2 : 1, nf = 2 * nh;                 // fragments per step
            const uint4* W = x2_weights(L.blk[blk], q);
            if (lw == 0 && pair) {
                // the gain | bias rows: free once every compute wave has finished the transform of the previous LayerNorm phase
                for (unsigned spins = 0;; ++spins) {
                    const unsigned v = lane < K::NCW ? x2_lds_ld(a_gbdone + 4 * lane) : 0xffffffffu;
                    if (__all(v >= (unsigned)ln_seen)) break;
                    if (x2_lds_ldu(a_abort)) return;
                    if (spins > XS_SPINS) { x2_give_up(a_abort, p.abort_dev, p.tmo, 0x700u + (unsigned)ph, lane); return; }
                    __builtin_amdgcn_s_sleep(1);
                }
                const float* f = p.film + L.blk[blk].film_off + (q == 3 ? 2 * C : 0);     // [bias | gain] of this LayerNorm
#pragma unroll
                for (int i = 0; i < K::GBF; ++i) {
                    // LDS image: gain [512] | bias [512] of K quarter `member`; memory: bias [C] | gain [C]
                    const int e = i * 256 + lane * 4;                 // float index in the LDS image
                    const float* src = e < 512 ? f + C + member * 512 + e : f + member * 512 + (e - 512);
                    x2_dma(src, reinterpret_cast<char*>(L.gb) + i * 1024);
                }
                issued += K::GBF;
                if (q == 0) {
                    const XBlockW& B = L.blk[blk];
                    const int k = lane >> 3, c4 = (lane & 7) * 4, ch = ct * 32 + c4;
                    const float* src = B.dw_w + (size_t)4 * 2 * C + ch;                   // centre tap of half a
                    if (k == 1) src = B.dw_w + (size_t)4 * 2 * C + C + ch;
                    else if (k == 2) src = B.dw_b + ch;
                    else if (k == 3) src = B.dw_b + C + ch;
                    else if (k == 4) src = B.b1 + ch;
                    else if (k >= 5) src = B.b1 + C + ch;
                    x2_dma(src, reinterpret_cast<char*>(L.dwc));
                    issued += K::DWF;
                }
            }
            if (pair) ++ln_seen;
            for (int j = 0; j < 16; ++j) {
                if ((j & (K::NLW - 1)) == lw) {
                    const unsigned cum_e = cum + (unsigned)nf;
                    for (unsigned spins = 0;; ++spins) {              // room in the ring: the slowest compute wave has let go of the fragments this step overwrites
                        const unsigned v = lane < K::NCW ? x2_lds_ld(a_cons + 4 * lane) : cum_e;
                        if (__all((int)(cum_e - v) <= K::RING)) break;
                        if (spins == 0) {                             // blocked anyway: everything issued so far lands and is reported
                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                            if (lane == 0) x2_lds_st(a_landed, issued);
                        }
                        if (x2_lds_ldu(a_abort)) return;
                        if (spins > XS_SPINS) { x2_give_up(a_abort, p.abort_dev, p.tmo, 0x700u + (unsigned)ph, lane); return; }
                        __builtin_amdgcn_s_sleep(1);
                    }
#pragma unroll
                    for (int f = 0; f < 4; ++f) {
                        if (f < nf) {
                            const int h = f % nh, cb = f / nh;
                            const int mb = (h ? C / 16 : 0) + ct * 2 + cb, ks = member * 16 + j;
#ifdef HD_STAMPS
                            if (p.dbg_no_w) continue;
#endif
                            x2_dma(W + ((size_t)mb * K::KS + ks) * 64 + lane, &L.ring[(cum + f) % K::RING][0]);
                        }
                    }
                    issued += nf;
                    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(K::DL) : "memory");
                    if (issued > K::DL && lane == 0) x2_lds_st(a_landed, issued - K::DL);
                }
                cum += nf;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) x2_lds_st(a_landed, issued);
        return;
    }

    // ================================================ compute waves ================================================
    const int cw = wave, rb = cw;
    const int n = lane & 15, g = lane >> 4;
    const bool fin = rb == member;                                     // this wave finishes its row block (wave-uniform)
    const int row = rb * 16 + n;                                       // this lane's face
    const bool row_ok = row < M;
    const int rowc = row_ok ? row : 0;
    const int ch0 = ct * 32 + 4 * g;                                   // + 16 * cb + i: this lane's output channels

    constexpr int hbytes = 64 * C * 2;                                 // hand-off buffers: 64 faces
    const __amdgpu_buffer_rsrc_t rs_hP = __builtin_amdgcn_make_buffer_rsrc(p.hP, 0, hbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_hG = __builtin_amdgcn_make_buffer_rsrc(p.hG, 0, hbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_hY = __builtin_amdgcn_make_buffer_rsrc(p.hY, 0, hbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_hG2 = __builtin_amdgcn_make_buffer_rsrc(p.hG2, 0, hbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_hX = __builtin_amdgcn_make_buffer_rsrc(p.hX, 0, hbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_hsx = __builtin_amdgcn_make_buffer_rsrc(p.hsx, 0, 64 * K::NP * 8, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_hsy = __builtin_amdgcn_make_buffer_rsrc(p.hsy, 0, 64 * K::NP * 8, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_Xb = __builtin_amdgcn_make_buffer_rsrc(p.Xb, 0, M * C * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_sx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2*>(p.sx), 0, M * K::NP * 8, 0x00020000);
    // split-K exchange: [tile][row block][slot 4][member][register 4][lane] uint4.  Four slots: a slot is written again four phases later,
    // and any four consecutive phases contain a LayerNorm phase, whose wait covers every tile of the row block
    const __amdgpu_buffer_rsrc_t rs_xb = __builtin_amdgcn_make_buffer_rsrc(p.xbuf, 0, 64 * 4 * 4 * 4 * 4 * 1024, 0x00020000);

    // chip-wide flags [row block][tile]; a consumer's 16 producers (K quarter `member`: tiles 16 member ..) are contiguous
    xs_gu32* fl_rb = (xs_gu32*)(p.flags + rb * 64);
    xs_gu32* my_flag = fl_rb + ct;
    xs_gu32* poll_q = fl_rb + member * 16;
    xs_gu32* xfl = (xs_gu32*)(p.xflags + (ct * 4 + rb) * 4);           // the cluster's exchange flags of this row block, one per member

    bool dead = false;
    unsigned* const abort_dev = p.abort_dev;
    unsigned* const tmo = p.tmo;
#define MID_GIVE_UP(code) do { x2_give_up(a_abort, abort_dev, tmo, (code), lane); dead = true; } while (0)
    // wait until the producers of this wave's rows / k have published phase ph - 1 (all: every tile, for the LayerNorm statistics)
    auto wait_flags = [&](int ph, bool all) __attribute__((always_inline)) {
        const unsigned want = base + (unsigned)ph;
        for (unsigned spins = 0;; ++spins) {
            const int nfl = all ? 64 : 16;
            const unsigned v = lane < nfl ? __hip_atomic_load((all ? fl_rb : poll_q) + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : want;
            const bool ok = (int)(v - want) >= 0;
            const bool inject = p.test_abort > 0 && ph == p.test_abort && ct == 0;
            if (!inject && __all(ok)) break;
            if (x2_lds_ldu(a_abort)) { dead = true; break; }
            if (spins > XS_SPINS || inject) { MID_GIVE_UP(0x100u + (unsigned)(ph - 1)); break; }
            __builtin_amdgcn_s_sleep(1);
        }
    };
    auto publish = [&](int ph) __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(my_flag, base + (unsigned)ph + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    // the finished tile (16 rows x 16 channels of block cb) as fragment ct of the consumers' B operand: lane (n, g) holds
    // k = 16 cb + 4 g .. + 3 of row n -> element 4 (g & 1) .. + 3 of consumer lane n + 16 (2 cb + (g >> 1)); write-through
    auto store_frag = [&](const __amdgpu_buffer_rsrc_t& rs, int cb, const float (&v)[4]) __attribute__((always_inline)) {
        const int lc = n + 16 * (2 * cb + (g >> 1));
        const int off = ((rb * K::KS + ct) * 64 + lc) * 16 + (g & 1) * 8;
        if (row_ok) __builtin_amdgcn_raw_buffer_store_b64((xs_u32x2){pack2(v[0], v[1]), pack2(v[2], v[3])}, rs, off, 0, 16);
    };

    // ---- ring bookkeeping (the same counts as the loaders) ----
    unsigned cum = 0;                                                  // ring fragments of all earlier steps
    unsigned lcnt[K::NLW];                                             // DMA instructions loader l has issued up to the step being read
#pragma unroll
    for (int l = 0; l < K::NLW; ++l) lcnt[l] = 0u;
    unsigned lseen[K::NLW];                                            // last value read of landed[l]
#pragma unroll
    for (int l = 0; l < K::NLW; ++l) lseen[l] = 0u;
    const unsigned a_landed0 = x2_lds_addr(&L.landed[0]), a_cons = x2_lds_addr(&L.consumed[cw]), a_gbdone = x2_lds_addr(&L.gbdone[cw]);
    auto wait_landed = [&](int l, unsigned need, unsigned code) __attribute__((always_inline)) {
        if ((int)(lseen[l] - need) >= 0) return;
        for (unsigned spins = 0;; ++spins) {
            lseen[l] = x2_lds_ldu(a_landed0 + 4 * l);
            if ((int)(lseen[l] - need) >= 0) break;
            if (x2_lds_ldu(a_abort)) { dead = true; break; }
            if (spins > XS_SPINS) { MID_GIVE_UP(code); break; }
            __builtin_amdgcn_s_sleep(1);
        }
    };

    // ---- residual stream in registers: x, y (fp32), the gate value g (bf16-rounded) of the rows this wave finishes ----
    float xv[2][4], yv[2][4], gq[2][4];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        const float4 v = *reinterpret_cast<const float4*>(p.X + (size_t)rowc * C + ch0 + 16 * cb);
        xv[cb][0] = v.x; xv[cb][1] = v.y; xv[cb][2] = v.z; xv[cb][3] = v.w;
#pragma unroll
        for (int i = 0; i < 4; ++i) { yv[cb][i] = 0.f; gq[cb][i] = 0.f; }
    }

    xs_u32x4 araw[16];                                                 // this wave's B operand of the phase: 16 k-steps x (row n, 8 k at 8 g)
    f32x4_t acc[2][2];                                                 // [channel block][gate half]
    float rstd = 0.f, nmr = 0.f;                                       // LayerNorm of this lane's row: x_hat = x * rstd + nmr

    // (mean, rstd) of this lane's row from 64 partials of 32 channels each, 16 per lane group (fixed order)
    auto merge_stats = [&](const float2 (&ps)[16]) __attribute__((always_inline)) {
        float sm = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) sm += ps[i].x;
        const float mean = x2_sum_rows(sm) * (1.0f / 64.0f);
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) { const float d = ps[i].x - mean; q += fmaf(32.f * d, d, ps[i].y); }
        const float var = x2_sum_rows(q) * (1.0f / (float)C);
        rstd = __frsqrt_rn(var + p.ln_eps);
        nmr = -mean * rstd;
    };
    // entry: [row][64] in the standard layout (16-byte loads); inside the stage [row block][tile][16 rows] (a producer's 16 rows are one line)
    auto load_stats_entry = [&](const __amdgpu_buffer_rsrc_t& rs, float2 (&ps)[16]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const xs_u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rs, (rowc * K::NP + g * 16 + 2 * i) * 8, 0, 0);
            ps[2 * i] = make_float2(__uint_as_float(r.x), __uint_as_float(r.y)); ps[2 * i + 1] = make_float2(__uint_as_float(r.z), __uint_as_float(r.w));
        }
    };
    auto load_stats = [&](const __amdgpu_buffer_rsrc_t& rs, float2 (&ps)[16]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const xs_u32x2 r = __builtin_amdgcn_raw_buffer_load_b64(rs, ((rb * K::NP + g * 16) * 16 + n) * 8, i * 128, 16);
            ps[i] = make_float2(__uint_as_float(r.x), __uint_as_float(r.y));
        }
    };

    // LayerNorm2d + FiLM (utils.py:16-24, conditional_naf.py:114-115,126-127) applied to this wave's B operand in place, before the K loop
    // (hd_xcd2.hpp: ln_transform)
    auto ln_transform = [&](int ph) __attribute__((always_inline)) {
        lcnt[0] += K::GBF + ((ph % 5) == 0 ? K::DWF : 0);
        wait_landed(0, lcnt[0], 0x400u + (unsigned)ph);
        asm volatile("" ::: "memory");
        const f32x2_t rs2 = {rstd, rstd}, nm2 = {nmr, nmr};
        f32x4_t gq4[3][4];
        typedef __attribute__((address_space(3))) const f32x4_t lds_f4;
        unsigned gb_base = x2_lds_addr(L.gb) + (unsigned)(g * 32);
        asm volatile("" : "+v"(gb_base));                                 // opaque: the 64 addresses below are base + immediate, not 64 hoisted registers
#pragma unroll
        for (int jj = 0; jj < 18; ++jj) {
            if (jj < 16) {
                const int bi = jj % 3;
                gq4[bi][0] = *(lds_f4*)(size_t)(gb_base + jj * 128); gq4[bi][1] = *(lds_f4*)(size_t)(gb_base + jj * 128 + 16);
                gq4[bi][2] = *(lds_f4*)(size_t)(gb_base + 2048 + jj * 128); gq4[bi][3] = *(lds_f4*)(size_t)(gb_base + 2048 + jj * 128 + 16);
            }
            if (jj >= 2) {
                const int j = jj - 2, bi = j % 3;
                const f32x4_t g0 = gq4[bi][0], g1 = gq4[bi][1], b0 = gq4[bi][2], b1 = gq4[bi][3];
                const f32x2_t gg[4] = {{g0.x, g0.y}, {g0.z, g0.w}, {g1.x, g1.y}, {g1.z, g1.w}};
                const f32x2_t bb[4] = {{b0.x, b0.y}, {b0.z, b0.w}, {b1.x, b1.y}, {b1.z, b1.w}};
                const unsigned w[4] = {araw[j].x, araw[j].y, araw[j].z, araw[j].w};
                unsigned o[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x2_t x = {__uint_as_float(w[i] << 16), __uint_as_float(w[i] & 0xffff0000u)};
                    o[i] = pack2(__builtin_elementwise_fma(__builtin_elementwise_fma(x, rs2, nm2), gg[i], bb[i]));
                }
                araw[j] = (xs_u32x4){o[0], o[1], o[2], o[3]};
                asm volatile("" : "+v"(araw[j]) :: "memory");           // the transform of step j is done before the reads of step j + 3 go out
            } else {
                asm volatile("" ::: "memory");
            }
        }
        asm volatile("" ::: "memory");
        if (lane == 0) x2_lds_st(a_gbdone, (unsigned)(ph / 5) * 2u + ((ph % 5) == 3 ? 2u : 1u));      // the gain | bias rows are free again
    };
    // K loop of one phase (hd_xcd2.hpp: k_loop): NH accumulators per channel block, ring reads PD k-steps ahead of the MFMAs
    auto k_loop = [&](int ph, auto nh_c) __attribute__((always_inline)) {
        constexpr int NH = decltype(nh_c)::value;
        constexpr int NF = 2 * NH;
        constexpr int PD = 2;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int h = 0; h < 2; ++h) acc[cb][h] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        xs_u32x4 wf[PD + 1][2][NH];
        typedef __attribute__((address_space(3))) const xs_u32x4 lds_u4;
        const unsigned ring_lane = x2_lds_addr(&L.ring[0][0]) + (unsigned)lane * 16u;
#pragma unroll
        for (int jj = 0; jj < 16 + PD; ++jj) {
            if (jj < 16) {
                const int l = jj & (K::NLW - 1), bi = jj % (PD + 1);
                lcnt[l] += NF;
                wait_landed(l, lcnt[l], 0x500u + (unsigned)ph);
                asm volatile("" ::: "memory");
                const unsigned step_addr = ring_lane + (((cum + (unsigned)(jj * NF)) % K::RING) << 10);
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int h = 0; h < NH; ++h) wf[bi][cb][h] = *(lds_u4*)(size_t)(step_addr + (cb * NH + h) * 1024);
            }
            if (jj >= PD) {
                const int j = jj - PD, bi = j % (PD + 1);
                const bf16x8_t av = __builtin_bit_cast(bf16x8_t, araw[j]);
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int h = 0; h < NH; ++h)
                        acc[cb][h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf[bi][cb][h]), av, acc[cb][h], 0, 0, 0);
                if ((j & 1) && lane == 0) x2_lds_st(a_cons, cum + (unsigned)((j + 1) * NF));   // steps <= j are in registers: their slots are free
            }
        }
        cum += 16 * NF;
    };
    // split-K exchange inside the cluster.  A wave that does not finish its row block stores its partial tile for the member that does,
    // drains and raises its flag; the finishing wave polls the three flags and adds the four partials in member order (its own from its
    // registers).  Stores stay in the XCD's L2 when the cluster shares one (handshake), write-through otherwise.
    auto exchange = [&](int ph, auto nh_c) __attribute__((always_inline)) {
        constexpr int NH = decltype(nh_c)::value;
        const int slot = ph & 3;
        const unsigned want = base + (unsigned)ph + 1u;
        if (!fin) {
            const int wbase = ((((ct * 4 + rb) * 4 + slot) * 4 + member) * 4 * 64 + lane) * 16;
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int h = 0; h < NH; ++h) {
                    const xs_u32x4 v = {__float_as_uint(acc[cb][h][0]), __float_as_uint(acc[cb][h][1]), __float_as_uint(acc[cb][h][2]), __float_as_uint(acc[cb][h][3])};
                    if (local) __builtin_amdgcn_raw_buffer_store_b128(v, rs_xb, wbase, (cb * NH + h) * 1024, 0);
                    else __builtin_amdgcn_raw_buffer_store_b128(v, rs_xb, wbase, (cb * NH + h) * 1024, 16);
                }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) {
                if (local) __hip_atomic_store(xfl + member, want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                else __hip_atomic_store(xfl + member, want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            return;
        }
        for (unsigned spins = 0;; ++spins) {
            const unsigned v = (lane < 4 && lane != member) ? __hip_atomic_load(xfl + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : want;
            if (__all((int)(v - want) >= 0)) break;
            if (x2_lds_ldu(a_abort)) { dead = true; break; }
            if (spins > XS_SPINS) { MID_GIVE_UP(0x600u + (unsigned)ph); break; }
            __builtin_amdgcn_s_sleep(1);
        }
        if (dead) return;
        xs_u32x4 f[4][2 * NH];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if (m != member) {
                const int rbase = ((((ct * 4 + rb) * 4 + slot) * 4 + m) * 4 * 64 + lane) * 16;
#pragma unroll
                for (int r = 0; r < 2 * NH; ++r) f[m][r] = __builtin_amdgcn_raw_buffer_load_b128(rs_xb, rbase, r * 1024, 16);
            } else {
#pragma unroll
                for (int r = 0; r < 2 * NH; ++r)
                    f[m][r] = (xs_u32x4){__float_as_uint(acc[r / NH][r % NH][0]), __float_as_uint(acc[r / NH][r % NH][1]), __float_as_uint(acc[r / NH][r % NH][2]), __float_as_uint(acc[r / NH][r % NH][3])};
            }
        }
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                const int r = cb * NH + h;
                acc[cb][h][0] = ((__uint_as_float(f[0][r].x) + __uint_as_float(f[1][r].x)) + __uint_as_float(f[2][r].x)) + __uint_as_float(f[3][r].x);
                acc[cb][h][1] = ((__uint_as_float(f[0][r].y) + __uint_as_float(f[1][r].y)) + __uint_as_float(f[2][r].y)) + __uint_as_float(f[3][r].y);
                acc[cb][h][2] = ((__uint_as_float(f[0][r].z) + __uint_as_float(f[1][r].z)) + __uint_as_float(f[2][r].z)) + __uint_as_float(f[3][r].z);
                acc[cb][h][3] = ((__uint_as_float(f[0][r].w) + __uint_as_float(f[1][r].w)) + __uint_as_float(f[2][r].w)) + __uint_as_float(f[3][r].w);
            }
    };

    // activations of a phase from a hand-off buffer (fragment order, sc1): the 16 k-steps of K quarter `member`, 16 KiB per wave
    const int a_frag_off = ((rb * K::KS + member * 16) * 64 + lane) * 16;
    auto load_a = [&](const __amdgpu_buffer_rsrc_t& rs) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 16; ++j) araw[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, a_frag_off, j * 1024, 16);
    };
    // (row, this tile's 32 channels) statistics of the finished rows -> hand-off (two-pass: mean, then M2 about it)
    auto store_stats = [&](const __amdgpu_buffer_rsrc_t& rs, const float (&v)[2][4]) __attribute__((always_inline)) {
        const float s1 = x2_sum_rows(((v[0][0] + v[0][1]) + (v[0][2] + v[0][3])) + ((v[1][0] + v[1][1]) + (v[1][2] + v[1][3])));
        const float mean = s1 * (1.0f / 32.0f);
        float q = 0.f;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int i = 0; i < 4; ++i) { const float d = v[cb][i] - mean; q = fmaf(d, d, q); }
        const float m2 = x2_sum_rows(q);
        if (g == 0) __builtin_amdgcn_raw_buffer_store_b64((xs_u32x2){__float_as_uint(mean), __float_as_uint(m2)}, rs, ((rb * K::NP + ct) * 16 + n) * 8, 0, 16);
    };
    auto col4 = [&](const float* basep, int cb) __attribute__((always_inline)) { return xs_ldg_f4(basep + ch0 + 16 * cb); };

    for (int blk = 0; blk < p.nblocks && !dead; ++blk) {
        const XBlockW& B = L.blk[blk];
        // ======================= q0: LN + FiLM -> conv1 -> depthwise centre tap -> SimpleGate (= pooled) =======================
        {
            const int ph = 5 * blk;
            if (ph >= P_run) break;
            HD_MSTAMP(0);
            float2 ps[16];
            if (ph == 0) {
                // entry: bf16 rows and (mean, M2) partials of 32 channels in the standard layouts, written by the previous launch
                load_stats_entry(rs_sx, ps);
#pragma unroll
                for (int j = 0; j < 16; ++j) araw[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_Xb, (rowc * C + member * 512 + 8 * g) * 2, j * 64, 0);
            } else {
                wait_flags(ph, true); if (dead) break;
                HD_MSTAMP(1);
                load_stats(rs_hsx, ps);
                load_a(rs_hX);
            }
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            merge_stats(ps);
            HD_MSTAMP(7);
            ln_transform(ph); if (dead) break;
            HD_MSTAMP(6);
            k_loop(ph, std::integral_constant<int, 2>()); if (dead) break;
            HD_MSTAMP(2);
            exchange(ph, std::integral_constant<int, 2>()); if (dead) break;
            HD_MSTAMP(3);
            if (fin) {
                // conv1 bias, depthwise 3x3 (pad 1) on a 1 x 1 map = its centre tap, SimpleGate; the pooled mean of one pixel is the gate
                const int cc = 4 * g;
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    float u[2][4];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const float4 wc = *reinterpret_cast<const float4*>(&L.dwc[(0 + h) * 32 + 16 * cb + cc]);
                        const float4 db = *reinterpret_cast<const float4*>(&L.dwc[(2 + h) * 32 + 16 * cb + cc]);
                        const float4 b1 = *reinterpret_cast<const float4*>(&L.dwc[(4 + h) * 32 + 16 * cb + cc]);
                        u[h][0] = fmaf(wc.x, acc[cb][h][0] + b1.x, db.x); u[h][1] = fmaf(wc.y, acc[cb][h][1] + b1.y, db.y);
                        u[h][2] = fmaf(wc.z, acc[cb][h][2] + b1.z, db.z); u[h][3] = fmaf(wc.w, acc[cb][h][3] + b1.w, db.w);
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) gq[cb][i] = bf16_bits_to_f32(f32_to_bf16_bits(u[0][i] * u[1][i]));
                }
                store_frag(rs_hP, 0, gq[0]); store_frag(rs_hP, 1, gq[1]);
                if (ph == P_run - 1 && P_run < P && row_ok) {      // introspection: the gate tile (= pooled) in the standard layouts
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb) {
                        if (p.dG) *reinterpret_cast<uint2*>(p.dG + (size_t)row * C + ch0 + 16 * cb) = make_uint2(pack2(gq[cb][0], gq[cb][1]), pack2(gq[cb][2], gq[cb][3]));
                        if (p.dpooled) *reinterpret_cast<float4*>(p.dpooled + (size_t)row * C + ch0 + 16 * cb) = make_float4(gq[cb][0], gq[cb][1], gq[cb][2], gq[cb][3]);
                    }
                }
                HD_MSTAMP(4);
                publish(ph);
            }
            HD_MSTAMP(5);
        }
        // ======================= q1: s = sca(pooled) ; G <- bf16(G * s) =======================
        {
            const int ph = 5 * blk + 1;
            if (ph >= P_run) break;
            HD_MSTAMP(0);
            wait_flags(ph, false); if (dead) break;
            HD_MSTAMP(1);
            load_a(rs_hP);
            const float4 bs0 = col4(B.bsca, 0), bs1 = col4(B.bsca, 1);
            k_loop(ph, std::integral_constant<int, 1>()); if (dead) break;
            HD_MSTAMP(2);
            exchange(ph, std::integral_constant<int, 1>()); if (dead) break;
            HD_MSTAMP(3);
            if (fin) {
                const float bs[2][4] = {{bs0.x, bs0.y, bs0.z, bs0.w}, {bs1.x, bs1.y, bs1.z, bs1.w}};
                float sv[2][4], gs[2][4];
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int i = 0; i < 4; ++i) { sv[cb][i] = acc[cb][0][i] + bs[cb][i]; gs[cb][i] = gq[cb][i] * sv[cb][i]; }
                store_frag(rs_hG, 0, gs[0]); store_frag(rs_hG, 1, gs[1]);
                if (ph == P_run - 1 && P_run < P && row_ok) {
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb) {
                        if (p.dG) *reinterpret_cast<uint2*>(p.dG + (size_t)row * C + ch0 + 16 * cb) = make_uint2(pack2(gs[cb][0], gs[cb][1]), pack2(gs[cb][2], gs[cb][3]));
                        if (p.dS) *reinterpret_cast<float4*>(p.dS + (size_t)row * C + ch0 + 16 * cb) = make_float4(sv[cb][0], sv[cb][1], sv[cb][2], sv[cb][3]);
                    }
                }
                HD_MSTAMP(4);
                publish(ph);
            }
            HD_MSTAMP(5);
        }
        // ======================= q2: conv3 ; y = x + beta * (.) ; LayerNorm partials =======================
        {
            const int ph = 5 * blk + 2;
            if (ph >= P_run) break;
            HD_MSTAMP(0);
            wait_flags(ph, false); if (dead) break;
            HD_MSTAMP(1);
            load_a(rs_hG);
            const float4 c0 = col4(B.b3, 0), c1 = col4(B.b3, 1), e0 = col4(B.beta, 0), e1 = col4(B.beta, 1);
            k_loop(ph, std::integral_constant<int, 1>()); if (dead) break;
            HD_MSTAMP(2);
            exchange(ph, std::integral_constant<int, 1>()); if (dead) break;
            HD_MSTAMP(3);
            if (fin) {
                const float b3[2][4] = {{c0.x, c0.y, c0.z, c0.w}, {c1.x, c1.y, c1.z, c1.w}}, be[2][4] = {{e0.x, e0.y, e0.z, e0.w}, {e1.x, e1.y, e1.z, e1.w}};
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int i = 0; i < 4; ++i) yv[cb][i] = fmaf(acc[cb][0][i] + b3[cb][i], be[cb][i], xv[cb][i]);
                store_frag(rs_hY, 0, yv[0]); store_frag(rs_hY, 1, yv[1]);
                store_stats(rs_hsy, yv);
                if (ph == P_run - 1 && P_run < P && row_ok && p.dYb) {
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb) *reinterpret_cast<uint2*>(p.dYb + (size_t)row * C + ch0 + 16 * cb) = make_uint2(pack2(yv[cb][0], yv[cb][1]), pack2(yv[cb][2], yv[cb][3]));
                }
                HD_MSTAMP(4);
                publish(ph);
            }
            HD_MSTAMP(5);
        }
        // ======================= q3: LN + FiLM -> conv4 -> SimpleGate =======================
        {
            const int ph = 5 * blk + 3;
            if (ph >= P_run) break;
            HD_MSTAMP(0);
            wait_flags(ph, true); if (dead) break;
            HD_MSTAMP(1);
            float2 ps[16];
            load_stats(rs_hsy, ps);
            load_a(rs_hY);
            const float4 a0 = col4(B.b4, 0), a1 = col4(B.b4, 1), d0 = col4(B.b4 + C, 0), d1 = col4(B.b4 + C, 1);
            asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
            merge_stats(ps);
            HD_MSTAMP(7);
            ln_transform(ph); if (dead) break;
            HD_MSTAMP(6);
            k_loop(ph, std::integral_constant<int, 2>()); if (dead) break;
            HD_MSTAMP(2);
            exchange(ph, std::integral_constant<int, 2>()); if (dead) break;
            HD_MSTAMP(3);
            if (fin) {
                const float b4a[2][4] = {{a0.x, a0.y, a0.z, a0.w}, {a1.x, a1.y, a1.z, a1.w}}, b4b[2][4] = {{d0.x, d0.y, d0.z, d0.w}, {d1.x, d1.y, d1.z, d1.w}};
                float g2[2][4];
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int i = 0; i < 4; ++i) g2[cb][i] = (acc[cb][0][i] + b4a[cb][i]) * (acc[cb][1][i] + b4b[cb][i]);
                store_frag(rs_hG2, 0, g2[0]); store_frag(rs_hG2, 1, g2[1]);
                if (ph == P_run - 1 && P_run < P && row_ok && p.dG) {
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb) *reinterpret_cast<uint2*>(p.dG + (size_t)row * C + ch0 + 16 * cb) = make_uint2(pack2(g2[cb][0], g2[cb][1]), pack2(g2[cb][2], g2[cb][3]));
                }
                HD_MSTAMP(4);
                publish(ph);
            }
            HD_MSTAMP(5);
        }
        // ======================= q4: conv5 ; x' = y + gamma * (.) ; LayerNorm partials =======================
        {
            const int ph = 5 * blk + 4;
            if (ph >= P_run) break;
            HD_MSTAMP(0);
            wait_flags(ph, false); if (dead) break;
            HD_MSTAMP(1);
            load_a(rs_hG2);
            const float4 c0 = col4(B.b5, 0), c1 = col4(B.b5, 1), e0 = col4(B.gamma, 0), e1 = col4(B.gamma, 1);
            k_loop(ph, std::integral_constant<int, 1>()); if (dead) break;
            HD_MSTAMP(2);
            exchange(ph, std::integral_constant<int, 1>()); if (dead) break;
            HD_MSTAMP(3);
            if (fin) {
                const float b5[2][4] = {{c0.x, c0.y, c0.z, c0.w}, {c1.x, c1.y, c1.z, c1.w}}, ga[2][4] = {{e0.x, e0.y, e0.z, e0.w}, {e1.x, e1.y, e1.z, e1.w}};
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int i = 0; i < 4; ++i) xv[cb][i] = fmaf(acc[cb][0][i] + b5[cb][i], ga[cb][i], yv[cb][i]);
                const bool last = (ph == P_run - 1);
                if (!last) {
                    store_frag(rs_hX, 0, xv[0]); store_frag(rs_hX, 1, xv[1]);
                    store_stats(rs_hsx, xv);
                    HD_MSTAMP(4);
                    publish(ph);
                } else if (row_ok) {
                    // exit: what the following launches read (kernel boundary), standard layouts
                    const bool gated = ph == P - 1 && p.outg16 != nullptr;
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb) {
                        const size_t o = (size_t)row * C + ch0 + 16 * cb;
                        *reinterpret_cast<float4*>(p.X + o) = make_float4(xv[cb][0], xv[cb][1], xv[cb][2], xv[cb][3]);
                        if (!gated) {
                            *reinterpret_cast<uint2*>(p.Xb + o) = make_uint2(pack2(xv[cb][0], xv[cb][1]), pack2(xv[cb][2], xv[cb][3]));
                        } else {                                          // f_d * (1 + w_c + w_s) (+ idc term): the HCA conv input (hca.py:28, model.py:245-247)
                            const float gsr = p.gate_s[row];
                            const float4 gc = *reinterpret_cast<const float4*>(p.gate_c + (size_t)row * C + ch0 + 16 * cb);
                            const float gcv[4] = {gc.x, gc.y, gc.z, gc.w};
                            float gv[4];
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const float a = p.add_src ? p.add_src[o + i] : 0.f;
                                gv[i] = (xv[cb][i] + a) * (1.0f + gcv[i] + gsr);
                            }
                            *reinterpret_cast<uint2*>(p.outg16 + o) = make_uint2(pack2(gv[0], gv[1]), pack2(gv[2], gv[3]));
                        }
                    }
                }
            }
            HD_MSTAMP(5);
        }
    }
    // the cluster's launch counter: every member has read it (the handshake completed before anyone got here)
    if (member == 0 && cw == 0 && lane == 0 && !dead) {
        xs_gu32* gs = (xs_gu32*)(p.gstate + ct * 4);
        __hip_atomic_store(gs, base / 64u + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

#undef MID_GIVE_UP

inline hipError_t launch_mid_stage(const MStageP& p, hipStream_t s) {
    if (p.B < 1 || p.B > 64 || p.nblocks < 1 || p.nblocks > XS_MAXBLK) return hipErrorInvalidValue;
    hipLaunchKernelGGL(mid_stage_kernel, dim3(256), dim3(MidCfg::THREADS), 0, s, p);
    return hipGetLastError();
}

}  // namespace hd
