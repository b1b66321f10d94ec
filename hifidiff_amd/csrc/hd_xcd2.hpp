// hd_xcd2.hpp — the XCD-local persistent stage of levels 2 / 3, second form: autonomous waves (gfx950 only).
//
// Same decomposition as hd_xcd.hpp (8 faces per XCD, the 32 workgroups of a group own the 32-column output tiles of every
// GEMM of a ConditionalNAFBlock, models/denoiser/conditional_naf.py:108-136; five dependent phases q0..q4 per block), but
// nothing inside a phase crosses a wave any more, so a phase has no workgroup barrier, no K-split reduction through LDS and
// no single storing wave:
//   * MFMA 16x16x32 with the WEIGHTS as the A operand (16 output channels x 32 k) and the ACTIVATIONS as the B operand
//     (32 k x 16 pixel rows): a wave owns 16 pixel rows x 32 output channels (two accumulators), its result sits with the
//     pixel row on the lane (lane & 15) and four consecutive channels in the registers -- which is what the next GEMM's
//     B operand wants (lane = row, 8 consecutive k), so a wave stores its own tile with one 8-byte store per lane straight
//     into "fragment order" hand-off buffers ([row block][k step][lane][8 bf16]: a consumer's load of a k-step is 1 KiB
//     contiguous) and raises its OWN flag; a consumer wave polls the flags of exactly the waves whose rows / k it reads.
//   * Level 3 (32 rows per XCD group): 2 row blocks x 2 K halves = 4 compute waves; the two K halves of a row block
//     exchange their partial tiles through LDS (pairwise, a flag word each) and each finishes 8 of the 16 rows.
//     Level 2 (64 rows per workgroup): 4 row blocks x the whole K = 4 compute waves, no exchange at all.
//   * Depthwise 3x3 + SimpleGate + pool (conditional_naf.py:116-119) in registers: the pixels of a face are lanes of one
//     16-lane row (quad_perm / row_shr DPP), the SCA input is published per face, the SCA GEMV takes it back replicated per
//     pixel row so that s lands in the lane that holds g.
//   * Weights: 4 loader waves stream the 1 KiB fragments of the phases ahead into an LDS ring by LDS-DMA
//     (global_load_lds_dwordx4), throttled, independent of the phase structure; the FiLM gain / bias row of a LayerNorm
//     and the depthwise constants arrive the same way.  Compute waves never hold a weight load in flight, so they can
//     drain their stores, poll and load activations without queueing behind the weight stream.
//   * LayerNorm + FiLM (conv1, conv4) is applied to the wave's B operand in a pass of its own, as the rows arrive and before
//     the K loop (inside the K loop its gain / bias LDS reads sat in front of every MFMA group); the K loop reads the ring
//     two k-steps ahead of the MFMAs.  Both are LDS-bandwidth bound: this form wins at level 2 and
//     loses to hd_xcd.hpp at level 3, which is why the library runs it at level 2 only.
// LayerNorm2d statistics (utils.py:16-24): producers emit fp32 (mean, M2) partials per (row, 16 channels), consumers
// merge them (equal counts: exact two-sum decomposition, fixed order).  Rounding points are those of hd_xcd.hpp /
// hd_gemm.hpp (bf16 operands, fp32 everything else); summation ORDER differs (one K chain per 512 channels instead of
// eight K slices), so results agree with the per-GEMM launches to accumulation order, not bit for bit:
// tests/test_gpu_parity.py checks every block of every stage against the oracle on its own inputs instead.
//
// Hand-off protocol: MI355X_MICROARCH.md "Valid forms" row 1 per WAVE -- stores, the storing wave's s_waitcnt vmcnt(0),
// its flag store; consumers poll with sc1 loads and read the payload with sc1 loads only.  Inside one XCD (checked at run
// time, HW_REG_XCC_ID handshake) payload and flag stores are plain, otherwise write-through.  Every spin is bounded;
// giving up -- by a compute wave or by a loader wave -- raises the abort words (hd_xcd.hpp) and every wave of the workgroup leaves.
#pragma once
#include "hd_xcd.hpp"

namespace hd {

typedef __attribute__((ext_vector_type(4))) float f32x4_t;

template <int C_, int HW_>
struct X2Cfg {
    static constexpr int C = C_, HW = HW_;
    static constexpr int R = XS_FACES * HW;             // rows of a group: 32 / 128
    static constexpr int NT = C / 32;                   // 32-channel output tiles: 32 / 16
    static constexpr int RSPLIT = XS_GROUP_WG / NT;     // 1 / 2
    static constexpr int RCU = R / RSPLIT;              // rows per workgroup: 32 / 64
    static constexpr int RB = RCU / 16;                 // row blocks per workgroup: 2 / 4
    static constexpr int KSPL = 4 / RB;                 // K halves: 2 / 1
    static constexpr int KS = C / 32;                   // k-steps of a K = C GEMM: 32 / 16
    static constexpr int KSW = KS / KSPL;               // k-steps per wave: 16 / 16
    static constexpr int S = (HW == 4) ? 2 : 4;         // face side
    static constexpr int NCW = 4, NLW = 4, THREADS = 64 * (NCW + NLW);
    static constexpr int RING = 128;                    // weight ring, 1 KiB fragments (level 3: one pair phase; level 2: two)
    static constexpr int GBF = 2 * C * 4 / 1024;        // fragments of a FiLM gain | bias row: 8 / 4
    static constexpr int DWF = 3;                       // fragments of the depthwise constants (704 floats)
    static constexpr int NPE = C / 32, NPS = C / 16;    // statistics partials per row: entry / inside the stage
#ifndef HD_X2_DL
#define HD_X2_DL 8
#endif
#ifndef HD_X2_PD
#define HD_X2_PD 2
#endif
    static constexpr int DL = HD_X2_DL;                 // LDS-DMA instructions a loader wave keeps in flight
    static constexpr int FLAGW = (KSPL == 2) ? 64 : 16; // flags a consumer wave polls: every producer of its row block (level 3: both K halves --
                                                        // the LayerNorm statistics of a row come from all 32 channel tiles)
    static_assert(RB * KSPL == 4 && KSW == 16 && (HW == 4 || HW == 16), "geometry");
};

template <int C, int HW>
struct X2Lds {
    typedef X2Cfg<C, HW> K;
    uint4 ring[K::RING][64];
    float gb[2 * C];                                    // gain [C] | bias [C] of the LayerNorm being applied
    float dwc[K::DWF * 256];                            // [22][32]: 9 + 9 depthwise taps, 2 depthwise biases, 2 conv1 biases of this tile's channels
    float xch[(K::KSPL == 2) ? 4 * 2 * 16 * 32 : 4];    // K-half exchange: [wave][parity][register][foreign lane]
    XBlockW blk[XS_MAXBLK];
    unsigned landed[K::NLW], consumed[K::NCW], gbdone[K::NCW], xflag[4];
    unsigned base, local, abort, pad_;
};

// raw LDS / wait helpers.  The loader waves' own LDS words go through asm: behind an LDS-DMA the compiler puts
// s_waitcnt vmcnt(0) in front of every LDS access it can see (the DMA writes LDS), which would drain the ring.
__device__ __forceinline__ unsigned x2_lds_addr(const void* p) { return (unsigned)(unsigned long long)(const __attribute__((address_space(3))) void*)p; }
__device__ __forceinline__ unsigned x2_lds_ld(unsigned addr) { unsigned v; asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(addr) : "memory"); return v; }
// the same for a control word every lane reads at one address: the value is wave-uniform, so are the branches on it
__device__ __forceinline__ unsigned x2_lds_ldu(unsigned addr) { return __builtin_amdgcn_readfirstlane(x2_lds_ld(addr)); }
__device__ __forceinline__ void x2_lds_st(unsigned addr, unsigned v) { asm volatile("ds_write_b32 %0, %1" :: "v"(addr), "v"(v) : "memory"); }
#ifndef HD_X2_AUX
#define HD_X2_AUX 0                                     // cache policy of the weight DMA (2 = nt: measured, not used)
#endif
__device__ __forceinline__ void x2_dma(const void* src_lane, void* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(unsigned long long)src_lane,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, HD_X2_AUX);
}
// sum over the four lanes that share lane & 15 (the four 16-lane rows of the wave): two VALU lane swaps, no LDS
__device__ __forceinline__ float x2_sum_rows(float v) {
    const unsigned u = __float_as_uint(v);
    const xs_u32x2 a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const float s = __uint_as_float(a.x) + __uint_as_float(a.y);
    const unsigned w = __float_as_uint(s);
    const xs_u32x2 b = __builtin_amdgcn_permlane32_swap(w, w, false, false);
    return __uint_as_float(b.x) + __uint_as_float(b.y);
}
template <int CTRL>
__device__ __forceinline__ float x2_dpp0(float v) {      // DPP move, lanes without a source read 0
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// a wait gave up: the workgroup's abort word (every wave of it leaves), the device word the later stages of the call read, the host-visible word
__device__ __forceinline__ void x2_give_up(unsigned a_abort, unsigned* abort_dev, unsigned* tmo, unsigned code, int lane) {
    if (lane == 0) {
        x2_lds_st(a_abort, 1u);
        __hip_atomic_store((xs_gu32*)abort_dev, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store((xs_gu32*)tmo, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
__device__ __forceinline__ const uint4* x2_weights(const XBlockW& b, int q) { return q == 0 ? b.w1 : q == 1 ? b.wsca : q == 2 ? b.w3 : q == 3 ? b.w4 : b.w5; }

#ifdef HD_STAMPS
#define HD_X2STAMP(i) do { if (p.stamps && cw == 0 && lane == 0) p.stamps[((size_t)ph * 256 + blockIdx.x) * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define HD_X2STAMP(i) do { } while (0)
#endif

template <int C, int HW>
__global__ __launch_bounds__((X2Cfg<C, HW>::THREADS)) void xcd2_stage_kernel(const X2StageP p) {
    typedef X2Cfg<C, HW> K;
    __shared__ __attribute__((aligned(16))) X2Lds<C, HW> L;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int group = blockIdx.x & 7, rank = blockIdx.x >> 3;        // blocks b and b + 8 share an XCD under round-robin dispatch (speed only)
    const int face_g0 = group * XS_FACES;
    if (face_g0 >= p.B) return;                                      // no faces for this group: nobody of the group takes part
    const int ct = rank % K::NT, rsp = rank / K::NT;
    const int M = p.B * HW;

    // ---- block table, LDS words, start-of-launch handshake (placement) ----
    {
        const unsigned* src = reinterpret_cast<const unsigned*>(p.blocks);
        unsigned* dst = reinterpret_cast<unsigned*>(L.blk);
        for (int i = tid; i < p.nblocks * (int)(sizeof(XBlockW) / 4); i += K::THREADS) dst[i] = src[i];
        if (tid < K::NLW) L.landed[tid] = 0u;
        if (tid < K::NCW) { L.consumed[tid] = 0u; L.gbdone[tid] = 0u; L.xflag[tid] = 0u; }
        if (tid == 0) L.abort = __hip_atomic_load((xs_gu32*)p.abort_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // an earlier stage of this call gave up
    }
    if (wave == 0) {
        xs_gu32* gs = (xs_gu32*)(p.gstate + group * 32);
        xs_gu32* hello = (xs_gu32*)(p.hello + group * 32);
        const unsigned n = __hip_atomic_load(gs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned xcc = xs_xcc_id();
        const unsigned mine = ((n + 1u) << 4) | xcc;
        if (lane == 0) __hip_atomic_store(hello + rank, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool same = false, gaveup = false;
        for (unsigned spins = 0;; ++spins) {
            const unsigned v = lane < XS_GROUP_WG ? __hip_atomic_load(hello + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : mine;
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
        // Loader lw issues the steps t = lw (mod NLW) of every phase (step j of a phase: the fragments of k-step j of
        // every K half, both channel blocks, both gate halves), as far ahead as the ring allows; loader 0 also brings the
        // FiLM row of a LayerNorm phase (and the depthwise constants of q0) into their own LDS regions.
        const int lw = wave - K::NCW;
        unsigned issued = 0;                                          // DMA instructions of this wave
        unsigned cum = 0;                                             // ring fragments of all earlier steps (every loader counts the same)
        const unsigned a_landed = x2_lds_addr(&L.landed[lw]), a_cons = x2_lds_addr(&L.consumed[0]), a_gbdone = x2_lds_addr(&L.gbdone[0]);
        int ln_seen = 0;
        // fault injection (hd_set_option "stage_test_abort" = 2000 + ph): loader 0 of group 0 gives up its wait for the gain | bias row of LayerNorm phase ph
        auto loader_inject = [&](int ph) { return p.test_abort == 2000 + ph && group == 0; };
        for (int ph = 0; ph < P_run; ++ph) {
            const int blk = ph / 5, q = ph - 5 * blk;
            const bool pair = (q == 0 || q == 3);
            const int nh = pair ? 2 : 1, nf = 2 * K::KSPL * nh;       // fragments per step
            const uint4* W = x2_weights(L.blk[blk], q);
            if (lw == 0 && pair) {
                // the gain | bias row: free once every compute wave has finished the K loop of the previous LayerNorm phase
                for (unsigned spins = 0;; ++spins) {
                    const unsigned v = lane < K::NCW ? x2_lds_ld(a_gbdone + 4 * lane) : 0xffffffffu;
                    if (!loader_inject(ph) && __all(v >= (unsigned)ln_seen)) break;
                    if (x2_lds_ldu(a_abort)) return;
                    if (spins > XS_SPINS || loader_inject(ph)) { x2_give_up(a_abort, p.abort_dev, p.tmo, 0x700u + (unsigned)ph, lane); return; }
                    __builtin_amdgcn_s_sleep(1);
                }
                const float* f = p.film + L.blk[blk].film_off + (q == 3 ? 2 * C : 0);     // [bias | gain] of this LayerNorm
#pragma unroll
                for (int i = 0; i < K::GBF; ++i) {
                    // LDS image: gain [C] | bias [C]; memory: bias [C] | gain [C]
                    const int e = i * 256 + lane * 4;                 // float index in the LDS image
                    const float* src = e < C ? f + C + e : f + (e - C);
                    x2_dma(src, reinterpret_cast<char*>(L.gb) + i * 1024);
                }
                issued += K::GBF;
                if (q == 0) {
                    const XBlockW& B = L.blk[blk];
#pragma unroll
                    for (int i = 0; i < K::DWF; ++i) {
                        // piece e4 = 4 floats of row k of the [22][32] image: taps of half a (k < 9), of half b (k < 18), depthwise biases, conv1 biases
                        const int e4 = i * 64 + lane, k = e4 >> 3, c4 = (e4 & 7) * 4;
                        const float* src = B.dw_w;
                        if (k < 18) src = B.dw_w + (size_t)(k % 9) * 2 * C + (k >= 9 ? C : 0) + ct * 32 + c4;
                        else if (k < 20) src = B.dw_b + (k == 19 ? C : 0) + ct * 32 + c4;
                        else if (k < 22) src = B.b1 + (k == 21 ? C : 0) + ct * 32 + c4;
                        x2_dma(src, reinterpret_cast<char*>(L.dwc) + i * 1024);
                    }
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
                    for (int f = 0; f < 8; ++f) {
                        if (f < nf) {
                            const int h = f % nh, cb = (f / nh) & 1, kh = f / (2 * nh);
                            const int mb = (h ? C / 16 : 0) + ct * 2 + cb, ks = kh * 16 + j;
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
    const int cw = wave;
    const int rb = cw % K::RB, kh = cw / K::RB;
    const int n = lane & 15, g = lane >> 4;
    const bool own = K::KSPL == 1 || (n >> 3) == kh;                   // rows this wave finishes (level 3: 8 of its 16)
    const int rbg = rsp * K::RB + rb;                                  // row block inside the group
    const int row = face_g0 * HW + rbg * 16 + n;                       // this lane's pixel row
    const bool row_ok = row < M;
    const int rowc = row_ok ? row : 0;
    const int face = rowc / HW;
    const int grb = group * (K::R / 16) + rbg;                         // row block of the whole batch (hand-off buffers are sized for 64 faces)
    const int ch0 = ct * 32 + 4 * g;                                   // + 16 * cb + i: this lane's output channels

    const int hbytes = 64 * HW * C * 2;                                // hand-off buffers: 64 faces
    const __amdgpu_buffer_rsrc_t rs_hX = __builtin_amdgcn_make_buffer_rsrc(p.hX, 0, hbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_hG = __builtin_amdgcn_make_buffer_rsrc(p.hG, 0, hbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_hY = __builtin_amdgcn_make_buffer_rsrc(p.hY, 0, hbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_hsx = __builtin_amdgcn_make_buffer_rsrc(p.hsx, 0, 64 * HW * K::NPS * 8, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_hsy = __builtin_amdgcn_make_buffer_rsrc(p.hsy, 0, 64 * HW * K::NPS * 8, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_P16 = __builtin_amdgcn_make_buffer_rsrc(p.pooled16, 0, p.B * C * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_Xb = __builtin_amdgcn_make_buffer_rsrc(p.Xb, 0, M * C * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_sx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2*>(p.sx), 0, M * K::NPE * 8, 0x00020000);

    // flags: level 3 [rb][ct][kh] (a consumer's 64 producers are contiguous: two lines), level 2 [row block of the group][ct]
    xs_gu32* fl_grp = (xs_gu32*)(p.flags + group * 128);
    xs_gu32* my_flag = fl_grp + (K::KSPL == 2 ? rb * 64 + ct * 2 + kh : rbg * 16 + ct);
    xs_gu32* poll_base = fl_grp + (K::KSPL == 2 ? rb * 64 : rbg * 16);

    bool dead = false;
    unsigned* const abort_dev = p.abort_dev;
    unsigned* const tmo = p.tmo;
#define X2_GIVE_UP(code) do { x2_give_up(a_abort, abort_dev, tmo, (code), lane); dead = true; } while (0)
    // wait until the producers of this wave's rows / k have published phase ph - 1
    auto wait_flags = [&](int ph) __attribute__((always_inline)) {
        const unsigned want = base + (unsigned)ph;
        for (unsigned spins = 0;; ++spins) {
            const unsigned v = lane < K::FLAGW ? __hip_atomic_load(poll_base + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : want;
            const bool inject = p.test_abort > 0 && ph == p.test_abort && group == 0;
            if (!inject && __all((int)(v - want) >= 0)) break;
            if (x2_lds_ldu(a_abort)) { dead = true; break; }
            if (spins > XS_SPINS || inject) { X2_GIVE_UP(0x100u + (unsigned)(ph - 1)); break; }
            __builtin_amdgcn_s_sleep(1);
        }
    };
    auto publish = [&](int ph) __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) {
            if (local) __hip_atomic_store(my_flag, base + (unsigned)ph + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else __hip_atomic_store(my_flag, base + (unsigned)ph + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    auto st64 = [&](const __amdgpu_buffer_rsrc_t& rs, int off, xs_u32x2 v) __attribute__((always_inline)) {
        if (local) __builtin_amdgcn_raw_buffer_store_b64(v, rs, off, 0, 0);
        else __builtin_amdgcn_raw_buffer_store_b64(v, rs, off, 0, 16);
    };
    auto st128 = [&](const __amdgpu_buffer_rsrc_t& rs, int off, xs_u32x4 v) __attribute__((always_inline)) {
        if (local) __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 0);
        else __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 16);
    };
    // this wave's tile (16 rows x 16 channels of block cb) as fragment ct of the consumers' B operand: lane (n, g) holds
    // k = 16 cb + 4 g .. + 3 of row n -> element 4 (g & 1) .. + 3 of consumer lane n + 16 (2 cb + (g >> 1))
    auto store_frag = [&](const __amdgpu_buffer_rsrc_t& rs, int cb, const float (&v)[4]) __attribute__((always_inline)) {
        const int lc = n + 16 * (2 * cb + (g >> 1));
        const int off = ((grb * K::KS + ct) * 64 + lc) * 16 + (g & 1) * 8;
        if (own && row_ok) st64(rs, off, (xs_u32x2){pack2(v[0], v[1]), pack2(v[2], v[3])});
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
            if (spins > XS_SPINS) { X2_GIVE_UP(code); break; }
            __builtin_amdgcn_s_sleep(1);
        }
    };

    // ---- residual stream tile in registers: x, y (fp32), the gate value g (bf16-rounded) ----
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

    // (mean, rstd) of this lane's row from NP partials of CNT channels each, NP / 4 per lane group (fixed order)
    auto merge_stats = [&](const float2* ps, auto np_c, float cnt) __attribute__((always_inline)) {
        constexpr int NPL = decltype(np_c)::value;                     // partials held by this lane
        float sm = 0.f;
#pragma unroll
        for (int i = 0; i < NPL; ++i) sm += ps[i].x;
        const float mean = x2_sum_rows(sm) * (1.0f / (float)(4 * NPL));
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NPL; ++i) { const float d = ps[i].x - mean; q += fmaf(cnt * d, d, ps[i].y); }
        const float var = x2_sum_rows(q) * (1.0f / (float)C);
        rstd = __frsqrt_rn(var + p.ln_eps);
        nmr = -mean * rstd;
    };

    // LayerNorm2d + FiLM (utils.py:16-24, conditional_naf.py:114-115,126-127) applied to this wave's B operand in place, as the rows
    // arrive (loads return in order) and before the K loop: a pass of LDS reads (gain | bias of the lane's 8 k per k-step) and packed
    // VALU only, which the compiler is free to batch -- inside the K loop the same reads sat on the MFMA chain (4.7 us per phase)
    auto ln_transform = [&](int ph) __attribute__((always_inline)) {
        lcnt[0] += K::GBF + ((ph % 5) == 0 ? K::DWF : 0);
        wait_landed(0, lcnt[0], 0x400u + (unsigned)ph);
        asm volatile("" ::: "memory");
        const f32x2_t rs2 = {rstd, rstd}, nm2 = {nmr, nmr};
        // three k-steps of gain | bias reads in flight (the asm statements keep the compiler from batching all 64 reads in front of the
        // arithmetic: 256 registers)
        f32x4_t gq4[3][4];
        typedef __attribute__((address_space(3))) const f32x4_t lds_f4;
        unsigned gb_base = x2_lds_addr(L.gb) + (unsigned)(kh * 2048 + g * 32);
        asm volatile("" : "+v"(gb_base));                                 // opaque: the 64 addresses below are base + immediate, not 64 hoisted registers
#pragma unroll
        for (int jj = 0; jj < 18; ++jj) {
            if (jj < 16) {
                const int bi = jj % 3;
                gq4[bi][0] = *(lds_f4*)(size_t)(gb_base + jj * 128); gq4[bi][1] = *(lds_f4*)(size_t)(gb_base + jj * 128 + 16);
                gq4[bi][2] = *(lds_f4*)(size_t)(gb_base + C * 4 + jj * 128); gq4[bi][3] = *(lds_f4*)(size_t)(gb_base + C * 4 + jj * 128 + 16);
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
        if (lane == 0) x2_lds_st(a_gbdone, (unsigned)(ph / 5) * 2u + ((ph % 5) == 3 ? 2u : 1u));      // the gain | bias row is free again
    };
    // K loop of one phase: NH accumulators per channel block.  Software pipeline, one k-step deep: the weight fragments of step
    // j + 1 are read from the ring before the MFMAs of step j, into the other register set (the loop is fully unrolled: indices are
    // constants).  The waits are the compiler's (exact lgkmcnt / vmcnt counts per use); a ring slot is let go once its step has been used.
    auto k_loop = [&](int ph, auto nh_c) __attribute__((always_inline)) {
        constexpr int NH = decltype(nh_c)::value;
        constexpr int NF = 2 * K::KSPL * NH;
        constexpr int PD = HD_X2_PD;                                   // k-steps of ring reads in flight ahead of the MFMAs
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
                // a step's fragments sit in consecutive slots (cum is a multiple of 32, a step of NF <= 8): one address per step
                const unsigned step_addr = ring_lane + (((cum + (unsigned)(jj * NF)) % K::RING) << 10) + (unsigned)(kh * 2 * NH * 1024);
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

    // level 3: the two K halves of a row block swap the partial tiles of the rows the OTHER one finishes
    auto exchange = [&](int ph, auto nh_c) __attribute__((always_inline)) {
        constexpr int NH = decltype(nh_c)::value;
        if constexpr (K::KSPL == 2) {
            const int par = ph & 1;
            float* mine = &L.xch[((cw * 2 + par) * 16) * 32];
            const float* theirs = &L.xch[(((cw ^ K::RB) * 2 + par) * 16) * 32];
            const int fl = (n & 7) + 8 * g;                            // lane among the 32 of a half
            if (!own) {
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int h = 0; h < NH; ++h)
#pragma unroll
                        for (int i = 0; i < 4; ++i) mine[((cb * 2 + h) * 4 + i) * 32 + fl] = acc[cb][h][i];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) x2_lds_st(x2_lds_addr(&L.xflag[cw]), (unsigned)ph + 1u);
            const unsigned a_their = x2_lds_addr(&L.xflag[cw ^ K::RB]);
            for (unsigned spins = 0;; ++spins) {
                if ((int)(x2_lds_ldu(a_their) - ((unsigned)ph + 1u)) >= 0) break;
                if (x2_lds_ldu(a_abort)) { dead = true; break; }
                if (spins > XS_SPINS) { X2_GIVE_UP(0x600u + (unsigned)ph); break; }
            }
            asm volatile("" ::: "memory");
            if (own) {
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int h = 0; h < NH; ++h)
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[cb][h][i] += theirs[((cb * 2 + h) * 4 + i) * 32 + fl];
            }
        }
    };

    // activations of a phase from a hand-off buffer (fragment order, sc1) -- 16 KiB per wave
    const int a_frag_off = ((grb * K::KS + kh * 16) * 64 + lane) * 16;
    auto load_a = [&](const __amdgpu_buffer_rsrc_t& rs) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 16; ++j) araw[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, a_frag_off, j * 1024, 16);
    };
    // (row, 16 channels) statistics of the finished tile -> hand-off, one 16-byte store per row (both channel blocks)
    auto store_stats = [&](const __amdgpu_buffer_rsrc_t& rs, const float (&v)[2][4]) __attribute__((always_inline)) {
        float2 ms[2];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const float s1 = x2_sum_rows((v[cb][0] + v[cb][1]) + (v[cb][2] + v[cb][3]));
            const float s2 = x2_sum_rows(fmaf(v[cb][0], v[cb][0], v[cb][1] * v[cb][1]) + fmaf(v[cb][2], v[cb][2], v[cb][3] * v[cb][3]));
            const float mean = s1 * (1.0f / 16.0f);
            ms[cb] = make_float2(mean, fmaxf(fmaf(-s1, mean, s2), 0.f));
        }
        if (own && row_ok && g == 0)
            st128(rs, (row * K::NPS + ct * 2) * 8, (xs_u32x4){__float_as_uint(ms[0].x), __float_as_uint(ms[0].y), __float_as_uint(ms[1].x), __float_as_uint(ms[1].y)});
    };
    auto col4 = [&](const float* base, int cb) __attribute__((always_inline)) { return xs_ldg_f4(base + ch0 + 16 * cb); };

    for (int blk = 0; blk < p.nblocks && !dead; ++blk) {
        const XBlockW& B = L.blk[blk];
        // ======================= q0: LN + FiLM -> conv1 -> depthwise 3x3 -> SimpleGate -> pooled =======================
        {
            const int ph = 5 * blk;
            if (ph >= P_run) break;
            HD_X2STAMP(0);
            if (ph == 0) {
                // entry: bf16 rows and (mean, M2) partials of 32 channels in the standard layouts, written by the previous launch
                float2 ps[K::NPE / 4];
#pragma unroll
                for (int i = 0; i < K::NPE / 8; ++i) {
                    const xs_u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rs_sx, (rowc * K::NPE + g * (K::NPE / 4) + 2 * i) * 8, 0, 0);
                    ps[2 * i] = make_float2(__uint_as_float(r.x), __uint_as_float(r.y)); ps[2 * i + 1] = make_float2(__uint_as_float(r.z), __uint_as_float(r.w));
                }
#pragma unroll
                for (int j = 0; j < 16; ++j) araw[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_Xb, (rowc * C + kh * 512 + 8 * g) * 2, j * 64, 0);
                asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                merge_stats(ps, std::integral_constant<int, K::NPE / 4>(), 32.f);
            } else {
                wait_flags(ph); if (dead) break;
                HD_X2STAMP(1);
                float2 ps[K::NPS / 4];
#pragma unroll
                for (int i = 0; i < K::NPS / 8; ++i) {
                    const xs_u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rs_hsx, (rowc * K::NPS + g * (K::NPS / 4) + 2 * i) * 8, 0, 16);
                    ps[2 * i] = make_float2(__uint_as_float(r.x), __uint_as_float(r.y)); ps[2 * i + 1] = make_float2(__uint_as_float(r.z), __uint_as_float(r.w));
                }
                load_a(rs_hX);
                asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                merge_stats(ps, std::integral_constant<int, K::NPS / 4>(), 16.f);
            }
            HD_X2STAMP(7);
            ln_transform(ph); if (dead) break;
            HD_X2STAMP(6);
            k_loop(ph, std::integral_constant<int, 2>()); if (dead) break;
            HD_X2STAMP(2);
            exchange(ph, std::integral_constant<int, 2>()); if (dead) break;
            HD_X2STAMP(3);
            // conv1 bias, depthwise 3x3 (pad 1) over the face's pixels, SimpleGate, pooled mean
            const int cc = 4 * g;                                      // + 16 cb: first of this lane's 4 channels inside the tile
            float pm[2][4];
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                float u[2][4];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float4 b1 = *reinterpret_cast<const float4*>(&L.dwc[(20 + h) * 32 + 16 * cb + cc]);
                    const float4 db = *reinterpret_cast<const float4*>(&L.dwc[(18 + h) * 32 + 16 * cb + cc]);
                    const float t[4] = {acc[cb][h][0] + b1.x, acc[cb][h][1] + b1.y, acc[cb][h][2] + b1.z, acc[cb][h][3] + b1.w};
                    float o[4] = {db.x, db.y, db.z, db.w};
                    if constexpr (HW == 4) {
                        // 2 x 2 face = one quad of lanes: pixel pq = n & 3 reads pixel pq ^ s through tap ((qy - py + 1) * 3 + (qx - px + 1))
                        const int pq = n & 3;
#pragma unroll
                        for (int s = 0; s < 4; ++s) {
                            const int qq = pq ^ s;
                            const int tap = ((qq >> 1) - (pq >> 1) + 1) * 3 + ((qq & 1) - (pq & 1) + 1);
                            const float4 w = *reinterpret_cast<const float4*>(&L.dwc[(h * 9 + tap) * 32 + 16 * cb + cc]);
                            const float ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const float v = s == 0 ? t[i] : s == 1 ? dpp_mov<0xB1>(t[i]) : s == 2 ? dpp_mov<0x4E>(t[i]) : dpp_mov<0x1B>(t[i]);
                                o[i] = fmaf(ww[i], v, o[i]);
                            }
                        }
                    } else {
                        // 4 x 4 face = one 16-lane row: pixel n = 4 y + x; x neighbours masked at the image border, y neighbours
                        // through row_shr:4 / row_shl:4, which read 0 outside the row (= outside the face)
                        const int px = n & 3;
                        float tapw[9][4];
#pragma unroll
                        for (int t9 = 0; t9 < 9; ++t9) {
                            const float4 w = *reinterpret_cast<const float4*>(&L.dwc[(h * 9 + t9) * 32 + 16 * cb + cc]);
                            tapw[t9][0] = w.x; tapw[t9][1] = w.y; tapw[t9][2] = w.z; tapw[t9][3] = w.w;
                        }
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            // (the DPP moves are executed by every lane, THEN masked: inside a conditional the source lanes of the
                            // lanes that take it would be switched off and read as 0)
                            const float l1 = x2_dpp0<0x111>(t[i]), r1 = x2_dpp0<0x101>(t[i]);   // row_shr:1 / row_shl:1 -> value of pixel n - 1 / n + 1
                            const float lft = px > 0 ? l1 : 0.f;
                            const float rgt = px < 3 ? r1 : 0.f;
                            const float xs3[3] = {lft, t[i], rgt};
#pragma unroll
                            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                                for (int dx = 0; dx < 3; ++dx) {
                                    const float v = dy == 0 ? x2_dpp0<0x114>(xs3[dx]) : dy == 2 ? x2_dpp0<0x104>(xs3[dx]) : xs3[dx];   // row above: n - 4, below: n + 4
                                    o[i] = fmaf(tapw[dy * 3 + dx][i], v, o[i]);
                                }
                        }
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) u[h][i] = o[i];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float gv = u[0][i] * u[1][i];
                    gq[cb][i] = bf16_bits_to_f32(f32_to_bf16_bits(gv));
                    float s = gv;
                    if constexpr (HW == 4) { s += dpp_mov<0xB1>(s); s += dpp_mov<0x4E>(s); }
                    else s = row16_sum(s);
                    pm[cb][i] = s * (1.0f / (float)HW);
                }
            }
            if (own && row_ok && (n & (HW - 1)) == 0) {
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) st64(rs_P16, (face * C + ch0 + 16 * cb) * 2, (xs_u32x2){pack2(pm[cb][0], pm[cb][1]), pack2(pm[cb][2], pm[cb][3])});
            }
            if (ph == P_run - 1 && P_run < P && own && row_ok) {      // introspection: the gate tile and the pooled mean in the standard layouts
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    if (p.dG) *reinterpret_cast<uint2*>(p.dG + (size_t)row * C + ch0 + 16 * cb) = make_uint2(pack2(gq[cb][0], gq[cb][1]), pack2(gq[cb][2], gq[cb][3]));
                    if (p.dpooled && (n & (HW - 1)) == 0) *reinterpret_cast<float4*>(p.dpooled + (size_t)face * C + ch0 + 16 * cb) = make_float4(pm[cb][0], pm[cb][1], pm[cb][2], pm[cb][3]);
                }
            }
            HD_X2STAMP(4);
            publish(ph);
            HD_X2STAMP(5);
        }
        // ======================= q1: s = sca(pooled) ; G <- bf16(G * s) =======================
        {
            const int ph = 5 * blk + 1;
            if (ph >= P_run) break;
            HD_X2STAMP(0);
            wait_flags(ph); if (dead) break;
            HD_X2STAMP(1);
            // the pooled vector of this row's face, as if it were the row: s then lands in the lane that holds g
#pragma unroll
            for (int j = 0; j < 16; ++j) araw[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_P16, (face * C + kh * 512 + 8 * g) * 2, j * 64, 16);
            const float4 bs0 = col4(B.bsca, 0), bs1 = col4(B.bsca, 1);
            k_loop(ph, std::integral_constant<int, 1>()); if (dead) break;
            HD_X2STAMP(2);
            exchange(ph, std::integral_constant<int, 1>()); if (dead) break;
            HD_X2STAMP(3);
            const float bs[2][4] = {{bs0.x, bs0.y, bs0.z, bs0.w}, {bs1.x, bs1.y, bs1.z, bs1.w}};
            float sv[2][4], gs[2][4];
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int i = 0; i < 4; ++i) { sv[cb][i] = acc[cb][0][i] + bs[cb][i]; gs[cb][i] = gq[cb][i] * sv[cb][i]; }
            store_frag(rs_hG, 0, gs[0]); store_frag(rs_hG, 1, gs[1]);
            if (ph == P_run - 1 && P_run < P && own && row_ok) {
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    if (p.dG) *reinterpret_cast<uint2*>(p.dG + (size_t)row * C + ch0 + 16 * cb) = make_uint2(pack2(gs[cb][0], gs[cb][1]), pack2(gs[cb][2], gs[cb][3]));
                    if (p.dS && (n & (HW - 1)) == 0) *reinterpret_cast<float4*>(p.dS + (size_t)face * C + ch0 + 16 * cb) = make_float4(sv[cb][0], sv[cb][1], sv[cb][2], sv[cb][3]);
                }
            }
            HD_X2STAMP(4);
            publish(ph);
            HD_X2STAMP(5);
        }
        // ======================= q2: conv3 ; y = x + beta * (.) ; LayerNorm partials =======================
        {
            const int ph = 5 * blk + 2;
            if (ph >= P_run) break;
            HD_X2STAMP(0);
            wait_flags(ph); if (dead) break;
            HD_X2STAMP(1);
            load_a(rs_hG);
            const float4 c0 = col4(B.b3, 0), c1 = col4(B.b3, 1), e0 = col4(B.beta, 0), e1 = col4(B.beta, 1);
            k_loop(ph, std::integral_constant<int, 1>()); if (dead) break;
            HD_X2STAMP(2);
            exchange(ph, std::integral_constant<int, 1>()); if (dead) break;
            HD_X2STAMP(3);
            const float b3[2][4] = {{c0.x, c0.y, c0.z, c0.w}, {c1.x, c1.y, c1.z, c1.w}}, be[2][4] = {{e0.x, e0.y, e0.z, e0.w}, {e1.x, e1.y, e1.z, e1.w}};
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int i = 0; i < 4; ++i) yv[cb][i] = fmaf(acc[cb][0][i] + b3[cb][i], be[cb][i], xv[cb][i]);
            store_frag(rs_hY, 0, yv[0]); store_frag(rs_hY, 1, yv[1]);
            store_stats(rs_hsy, yv);
            if (ph == P_run - 1 && P_run < P && own && row_ok && p.dYb) {
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) *reinterpret_cast<uint2*>(p.dYb + (size_t)row * C + ch0 + 16 * cb) = make_uint2(pack2(yv[cb][0], yv[cb][1]), pack2(yv[cb][2], yv[cb][3]));
            }
            HD_X2STAMP(4);
            publish(ph);
            HD_X2STAMP(5);
        }
        // ======================= q3: LN + FiLM -> conv4 -> SimpleGate =======================
        {
            const int ph = 5 * blk + 3;
            if (ph >= P_run) break;
            HD_X2STAMP(0);
            wait_flags(ph); if (dead) break;
            HD_X2STAMP(1);
            float2 ps[K::NPS / 4];
#pragma unroll
            for (int i = 0; i < K::NPS / 8; ++i) {
                const xs_u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rs_hsy, (rowc * K::NPS + g * (K::NPS / 4) + 2 * i) * 8, 0, 16);
                ps[2 * i] = make_float2(__uint_as_float(r.x), __uint_as_float(r.y)); ps[2 * i + 1] = make_float2(__uint_as_float(r.z), __uint_as_float(r.w));
            }
            load_a(rs_hY);
            const float4 a0 = col4(B.b4, 0), a1 = col4(B.b4, 1), d0 = col4(B.b4 + C, 0), d1 = col4(B.b4 + C, 1);
            asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
            merge_stats(ps, std::integral_constant<int, K::NPS / 4>(), 16.f);
            HD_X2STAMP(7);
            ln_transform(ph); if (dead) break;
            HD_X2STAMP(6);
            k_loop(ph, std::integral_constant<int, 2>()); if (dead) break;
            HD_X2STAMP(2);
            exchange(ph, std::integral_constant<int, 2>()); if (dead) break;
            HD_X2STAMP(3);
            const float b4a[2][4] = {{a0.x, a0.y, a0.z, a0.w}, {a1.x, a1.y, a1.z, a1.w}}, b4b[2][4] = {{d0.x, d0.y, d0.z, d0.w}, {d1.x, d1.y, d1.z, d1.w}};
            float g2[2][4];
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int i = 0; i < 4; ++i) g2[cb][i] = (acc[cb][0][i] + b4a[cb][i]) * (acc[cb][1][i] + b4b[cb][i]);
            store_frag(rs_hG, 0, g2[0]); store_frag(rs_hG, 1, g2[1]);
            if (ph == P_run - 1 && P_run < P && own && row_ok && p.dG) {
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) *reinterpret_cast<uint2*>(p.dG + (size_t)row * C + ch0 + 16 * cb) = make_uint2(pack2(g2[cb][0], g2[cb][1]), pack2(g2[cb][2], g2[cb][3]));
            }
            HD_X2STAMP(4);
            publish(ph);
            HD_X2STAMP(5);
        }
        // ======================= q4: conv5 ; x' = y + gamma * (.) ; LayerNorm partials =======================
        {
            const int ph = 5 * blk + 4;
            if (ph >= P_run) break;
            HD_X2STAMP(0);
            wait_flags(ph); if (dead) break;
            HD_X2STAMP(1);
            load_a(rs_hG);
            const float4 c0 = col4(B.b5, 0), c1 = col4(B.b5, 1), e0 = col4(B.gamma, 0), e1 = col4(B.gamma, 1);
            k_loop(ph, std::integral_constant<int, 1>()); if (dead) break;
            HD_X2STAMP(2);
            exchange(ph, std::integral_constant<int, 1>()); if (dead) break;
            HD_X2STAMP(3);
            const float b5[2][4] = {{c0.x, c0.y, c0.z, c0.w}, {c1.x, c1.y, c1.z, c1.w}}, ga[2][4] = {{e0.x, e0.y, e0.z, e0.w}, {e1.x, e1.y, e1.z, e1.w}};
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int i = 0; i < 4; ++i) xv[cb][i] = fmaf(acc[cb][0][i] + b5[cb][i], ga[cb][i], yv[cb][i]);
            const bool last = (ph == P_run - 1);
            if (!last) {
                store_frag(rs_hX, 0, xv[0]); store_frag(rs_hX, 1, xv[1]);
                store_stats(rs_hsx, xv);
                HD_X2STAMP(4);
                publish(ph);
            } else if (own && row_ok) {
                // exit: what the following launches read (kernel boundary), standard layouts
                const bool gated = ph == P - 1 && p.outg16 != nullptr;
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    const size_t o = (size_t)row * C + ch0 + 16 * cb;
                    *reinterpret_cast<float4*>(p.X + o) = make_float4(xv[cb][0], xv[cb][1], xv[cb][2], xv[cb][3]);
                    if (!gated) {
                        *reinterpret_cast<uint2*>(p.Xb + o) = make_uint2(pack2(xv[cb][0], xv[cb][1]), pack2(xv[cb][2], xv[cb][3]));
                    } else {                                          // f_d * (1 + w_c + w_s) (+ idc term): the HCA conv input (hca.py:28)
                        const float gsr = p.gate_s[row];
                        const float4 gc = *reinterpret_cast<const float4*>(p.gate_c + (size_t)face * C + ch0 + 16 * cb);
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
            HD_X2STAMP(5);
        }
    }
    // a phase_limit that stops inside a block: leave x as it stands (x of the block's input) -- the dumps above are what is inspected
    // the group's launch counter: every member has read it (the handshake completed before anyone got here)
    if (rank == 0 && cw == 0 && lane == 0 && !dead) {
        xs_gu32* gs = (xs_gu32*)(p.gstate + group * 32);
        __hip_atomic_store(gs, base / 64u + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

#undef X2_GIVE_UP

template <int C, int HW>
inline hipError_t launch_xcd2_stage(const X2StageP& p, hipStream_t s) {
    typedef X2Cfg<C, HW> K;
    if (p.B < 1 || p.B > XS_GROUPS * XS_FACES || p.nblocks < 1 || p.nblocks > XS_MAXBLK) return hipErrorInvalidValue;
    hipLaunchKernelGGL((xcd2_stage_kernel<C, HW>), dim3(XS_GROUPS * XS_GROUP_WG), dim3(K::THREADS), 0, s, p);
    return hipGetLastError();
}

}  // namespace hd
