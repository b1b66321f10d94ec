// persist_bench.hip — prototype of the persistent deep-level kernel (tools only): a chain of dependent
// M = 64, K = 2048 GEMM phases (plain N = 2048 and SimpleGate pair N = 4096, the middle level's shapes at
// latent 16, batch 64) run (a) as ONE launch whose 256 workgroups hand the activation tiles over through
// write-through (sc1) stores + per-workgroup flags, and (b) as one launch per phase (kernel boundary).
// Prints us per phase for both, in-kernel stamps for (a), and checks (a) == (b) bit for bit.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o gpurun_out/persist_bench tools/persist_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) unsigned gu32;

constexpr int KDIM = 2048, ROWS = 64, NT = KDIM / 32;        // 64 column tiles of the activation
constexpr int RG = 4, WG = 256;                              // 4 row groups of 16 rows x 64 tiles
constexpr int KSTEPS = KDIM / 16;                            // 128 k-steps of 16
constexpr int NSTAMP = 6;

struct Phase { const uint4* W; int pair; };

typedef __attribute__((address_space(1))) const u32x4 gc_u32x4;
__device__ __forceinline__ uint4 nt_load(const uint4* q) {        // global_load (a pointer read from memory is generic: flat_load otherwise)
    const u32x4 v = __builtin_nontemporal_load((gc_u32x4*)(unsigned long long)q);
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ unsigned pack2(float lo, float hi) {
    unsigned short a = __builtin_bit_cast(unsigned short, (__bf16)lo);
    unsigned short b = __builtin_bit_cast(unsigned short, (__bf16)hi);
    return (unsigned)a | ((unsigned)b << 16);
}

// act: [2 parities][RG][NT][64 lanes] uint4 — tile (rg, ct) is 16 rows x 32 cols bf16 = 1 KiB, lane l holds
// row l >> 2, cols 8 * (l & 3) .. +7.  flags: [P][WG] words, zeroed by the host before every launch.
__global__ __launch_bounds__(512) void persist_chain(const Phase* phases, int p_begin, int p_end, uint4* act, unsigned* flags,
                                                     float scale, unsigned* tmo, unsigned long long* stamps) {
    __shared__ __attribute__((aligned(16))) uint4 stage[8][2][64];              // per wave, 2 slots of one tile
    __shared__ __attribute__((aligned(16))) float red[2][2][8][16 * 32];       // [parity][tn][wave][row][col]
    const int tid = threadIdx.x, lane = tid & 63, wk = tid >> 6;
    const int lin = blockIdx.x, xcd = lin & 7, j = lin >> 3;
    const int ct = (j >> 2) * 8 + xcd, rg = j & 3;
    const __amdgpu_buffer_rsrc_t act_rs = __builtin_amdgcn_make_buffer_rsrc(act, 0, 2 * RG * NT * 1024, 0x00020000);

    uint4 w[2][16];
    auto load_w = [&](const Phase& ph) {
        const uint4* Wl = ph.W + lane;
#pragma unroll
        for (int s = 0; s < 16; ++s) w[0][s] = nt_load(Wl + ((size_t)ct * KSTEPS + 16 * wk + s) * 64);
        if (ph.pair) {
#pragma unroll
            for (int s = 0; s < 16; ++s) w[1][s] = nt_load(Wl + ((size_t)(ct + NT) * KSTEPS + 16 * wk + s) * 64);
        }
    };
    Phase cur = phases[p_begin];
    load_w(cur);
    bool dead = false;

    for (int p = p_begin; p < p_end; ++p) {
        const int par = p & 1;
        unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
        if (stamps && tid == 0) t0 = __builtin_amdgcn_s_memrealtime();
        // ---- wait for the 8 tiles of this wave's K slice (phase p - 1 outputs of row group rg) ----
        if (p > 0 && !dead) {
            const gu32* f = (const gu32*)(flags + (size_t)(p - 1) * WG);
            const unsigned long long tstart = __builtin_amdgcn_s_memrealtime();
            for (;;) {
                unsigned v = 1;
                if (lane < 8) {
                    // producer of tile t = 8 * wk + lane of row group rg: workgroup with ct' = t, rg' = rg
                    const int t = 8 * wk + lane;
                    const int lin_p = (((t >> 3) * 4 + rg) << 3) | (t & 7);
                    v = __hip_atomic_load(f + lin_p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (__all(v != 0)) break;
                if (__builtin_amdgcn_s_memrealtime() - tstart > 2000000ull) {       // 20 ms: give up, flag it
                    if (lane == 0) __hip_atomic_store((gu32*)tmo, (unsigned)(p + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    dead = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
        }
        if (stamps && tid == 0) t1 = __builtin_amdgcn_s_memrealtime();
        // ---- A tiles: sc1 loads (L2-served, never this CU's L1) ----
        u32x4 a[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
            a[i] = __builtin_amdgcn_raw_buffer_load_b128(act_rs, (((par * RG + rg) * NT + 8 * wk + i) * 64 + lane) * 16, 0, 16);
        f32x16_t acc0, acc1;
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
        const int r = lane & 31, kh = lane >> 5;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            uint4* st = stage[wk][i & 1];
            st[lane] = make_uint4(a[i].x, a[i].y, a[i].z, a[i].w);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                uint4 fr = make_uint4(0, 0, 0, 0);
                if (r < 16) fr = st[r * 4 + 2 * h + kh];
                const bf16x8_t af = __builtin_bit_cast(bf16x8_t, fr);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8_t, w[0][2 * i + h]), acc0, 0, 0, 0);
                if (cur.pair) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8_t, w[1][2 * i + h]), acc1, 0, 0, 0);
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (stamps && tid == 0) t2 = __builtin_amdgcn_s_memrealtime();
        // ---- next phase's weights: requested now, they fly during the reduction, the epilogue and the wait ----
        const int was_pair = cur.pair;
        if (p + 1 < p_end) { cur = phases[p + 1]; load_w(cur); }
        // ---- K-split partials -> LDS; rows 0..15 live in acc[0..7] ----
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
            red[par][0][wk][row * 32 + (lane & 31)] = acc0[i];
            if (was_pair) red[par][1][wk][row * 32 + (lane & 31)] = acc1[i];
        }
        __syncthreads();
        if (stamps && tid == 0) t3 = __builtin_amdgcn_s_memrealtime();
        if (wk == 0) {
            float v[8], v2[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { v[e] = 0.f; v2[e] = 0.f; }
#pragma unroll
            for (int ww = 0; ww < 8; ++ww) {
                const float4 x0 = *reinterpret_cast<const float4*>(&red[par][0][ww][lane * 8]);
                const float4 x1 = *reinterpret_cast<const float4*>(&red[par][0][ww][lane * 8 + 4]);
                v[0] += x0.x; v[1] += x0.y; v[2] += x0.z; v[3] += x0.w; v[4] += x1.x; v[5] += x1.y; v[6] += x1.z; v[7] += x1.w;
                if (was_pair) {
                    const float4 y0 = *reinterpret_cast<const float4*>(&red[par][1][ww][lane * 8]);
                    const float4 y1 = *reinterpret_cast<const float4*>(&red[par][1][ww][lane * 8 + 4]);
                    v2[0] += y0.x; v2[1] += y0.y; v2[2] += y0.z; v2[3] += y0.w; v2[4] += y1.x; v2[5] += y1.y; v2[6] += y1.z; v2[7] += y1.w;
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = was_pair ? v[e] * v2[e] * scale * scale * 90.f : v[e] * scale;
            u32x4 o;
            o.x = pack2(v[0], v[1]); o.y = pack2(v[2], v[3]); o.z = pack2(v[4], v[5]); o.w = pack2(v[6], v[7]);
            __builtin_amdgcn_raw_buffer_store_b128(o, act_rs, ((((par ^ 1) * RG + rg) * NT + ct) * 64 + lane) * 16, 0, 16);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_store((gu32*)(flags + (size_t)p * WG + lin), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (stamps && tid == 0) {
            t4 = __builtin_amdgcn_s_memrealtime();
            unsigned long long* sp = stamps + ((size_t)p * WG + lin) * NSTAMP;
            sp[0] = t0; sp[1] = t1; sp[2] = t2; sp[3] = t3; sp[4] = t4;
        }
    }
}


// ------------------------------------------------------------------------------------------------------
// ring_chain: the same chain with the weights decoupled from the dependent phases.  8 waves per workgroup:
// waves 0..3 consume (each a quarter of K, MFMA 16x16x32: A fragment = one 1 KiB tile load, no staging),
// waves 4..7 stream the weight fragments of "their" consumer by LDS-DMA into a private LDS ring, as far ahead
// as the ring allows — across phase boundaries, so the next phase's weights arrive during the hand-off bubble.
// Loader <-> consumer: two LDS words per ring (filled / consumed, monotonic fragment counts).
// Tile layout of the activation hand-off: lane l holds row l & 15, k 8 * (l >> 4) .. +7 (the A fragment itself).
// Weights: [N/16 column groups][K/32 k-steps][64 lanes] uint4 (B fragment of v_mfma_f32_16x16x32_bf16).
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
constexpr int RS = 24;                                       // ring slots (1 KiB fragments) per consumer
constexpr int RD = 12;                                       // LDS-DMA fragments in flight per loader wave
struct RingLds {
    uint4 ring[4][RS][64];
    float red[2][2][4][16 * 32];
    unsigned filled[4], consumed[4], red_done[4], abort;
};
template <bool NT>
__device__ __forceinline__ void dma_frag(const uint4* src_lane, uint4* lds_slot) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(unsigned long long)src_lane,
                                     (__attribute__((address_space(3))) void*)lds_slot, 16, 0, NT ? 2 : 0);
}
#define SPIN_LIMIT 2000000ull
__device__ __forceinline__ unsigned lds_ld(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_st(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
// The loader wave's own LDS words go through asm: behind an LDS-DMA the compiler puts s_waitcnt vmcnt(0) in front of
// every LDS access it can see (the DMA writes LDS), which would drain the ring's in-flight fragments at every poll.
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(unsigned long long)(const __attribute__((address_space(3))) void*)p; }
__device__ __forceinline__ unsigned lds_ld_raw(unsigned addr) { unsigned v; asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(addr) : "memory"); return v; }
__device__ __forceinline__ void lds_st_raw(unsigned addr, unsigned v) { asm volatile("ds_write_b32 %0, %1" :: "v"(addr), "v"(v) : "memory"); }
template <bool NT, int LOADMODE>
__global__ __launch_bounds__(512) void ring_chain(const Phase* __restrict__ phases, int p_begin, int p_end, uint4* act, unsigned* flags,
                                                  float scale, unsigned* tmo, unsigned long long* stamps, int noload, int nslab) {
    __shared__ __attribute__((aligned(16))) RingLds L;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lin = blockIdx.x, xcd = lin & 7, jj = lin >> 3;
    const int ct = (jj >> 2) * 8 + xcd, rg = jj & 3;
    if (tid < 4) { L.filled[tid] = noload ? 0x7fffffffu : 0u; L.consumed[tid] = 0u; L.red_done[tid] = 0u; }
    if (tid == 0) L.abort = 0u;
    __syncthreads();
    if (wave >= 4) {
        // ------------------------------------------------ loader ------------------------------------------------
        if (noload) return;
        const int c = wave - 4;
        unsigned issued = 0; int slot = 0;
        const unsigned a_consumed = lds_addr(&L.consumed[c]), a_filled = lds_addr(&L.filled[c]), a_abort = lds_addr(&L.abort);
        for (int p = p_begin; p < p_end; ++p) {
            const Phase ph = phases[p];
            const int nf = ph.pair ? 4 : 2;
            for (int ks = 0; ks < 16; ++ks) {
                const unsigned long long tstart = __builtin_amdgcn_s_memrealtime();
                while ((int)(issued + nf - lds_ld_raw(a_consumed)) > RS) {
                    if (lds_ld_raw(a_abort)) return;
                    if (__builtin_amdgcn_s_memrealtime() - tstart > SPIN_LIMIT) { lds_st_raw(a_abort, 1u); if (lane == 0) __hip_atomic_store((gu32*)tmo, 1000u + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
                    __builtin_amdgcn_s_sleep(1);
                }
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    if (f < nf) {
                        const int cg = (ct + (f >> 1) * NT) * 2 + (f & 1);
                        dma_frag<NT>(ph.W + ((size_t)cg * 64 + 16 * c + ks) * 64 + lane, &L.ring[c][slot][0]);
                        slot = (slot + 1 == RS) ? 0 : slot + 1;
                    }
                }
                issued += nf;
                if (issued > RD) {
                    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");                 // = RD
                    if (lane == 0) lds_st_raw(a_filled, issued - RD);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) lds_st_raw(a_filled, issued);
        return;
    }
    // -------------------------------------------------- consumer --------------------------------------------------
    const int c = wave;
    const __amdgpu_buffer_rsrc_t act_rs = __builtin_amdgcn_make_buffer_rsrc(act, 0, nslab * RG * NT * 1024, 0x00020000);
    unsigned taken = 0; int slot = 0;
    for (int p = p_begin; p < p_end; ++p) {
        const int par = p & 1, s_in = p % nslab, s_out = (p + 1) % nslab;
        const Phase ph = phases[p];
        const int nf = ph.pair ? 4 : 2;
        unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
        const bool st = stamps && tid == 0;
        if (st) t0 = __builtin_amdgcn_s_memrealtime();
        if (p > 0) {
            const gu32* f = (const gu32*)(flags + (size_t)(p - 1) * WG);
            const unsigned long long tstart = __builtin_amdgcn_s_memrealtime();
            for (;;) {
                unsigned v = 1;
                if (lane < 16) {
                    const int t = 16 * c + lane;
                    const int lin_p = (((t >> 3) * 4 + rg) << 3) | (t & 7);
                    v = __hip_atomic_load(f + lin_p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (__all(v != 0)) break;
                if (lds_ld(&L.abort)) return;
                if (__builtin_amdgcn_s_memrealtime() - tstart > SPIN_LIMIT) { lds_st(&L.abort, 1u); if (lane == 0) __hip_atomic_store((gu32*)tmo, (unsigned)(p + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        if (st) t1 = __builtin_amdgcn_s_memrealtime();
        if (LOADMODE == 2 && p > 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        u32x4 a[16];
#pragma unroll
        for (int i = 0; i < 16; ++i)
            a[i] = __builtin_amdgcn_raw_buffer_load_b128(act_rs, (((s_in * RG + rg) * NT + 16 * c + i) * 64 + lane) * 16, 0, LOADMODE == 0 ? 16 : 0);
        f32x4_t acc[4];
#pragma unroll
        for (int f = 0; f < 4; ++f) acc[f] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const unsigned need = taken + nf;
            for (unsigned spins = 0; (int)(lds_ld(&L.filled[c]) - need) < 0; ++spins) {
                if (spins > (1u << 22)) { lds_st(&L.abort, 1u); if (lane == 0) __hip_atomic_store((gu32*)tmo, 2000u + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
                if ((spins & 63) == 63 && lds_ld(&L.abort)) return;
            }
            asm volatile("" ::: "memory");
            const bf16x8_t af = __builtin_bit_cast(bf16x8_t, a[ks]);
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                if (f < nf) {
                    const uint4 b = L.ring[c][slot][lane];
                    slot = (slot + 1 == RS) ? 0 : slot + 1;
                    acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, __builtin_bit_cast(bf16x8_t, b), acc[f], 0, 0, 0);
                }
            }
            taken = need;
            asm volatile("" ::: "memory");
            if (lane == 0) lds_st(&L.consumed[c], taken);
        }
        if (st) t2 = __builtin_amdgcn_s_memrealtime();
        // partials: tile f = (tn, half): col = 16 * half + (lane & 15), rows 4 * (lane >> 4) + i
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            if (f < nf) {
#pragma unroll
                for (int i = 0; i < 4; ++i) L.red[par][f >> 1][c][(4 * (lane >> 4) + i) * 32 + 16 * (f & 1) + (lane & 15)] = acc[f][i];
            }
        }
        asm volatile("" ::: "memory");
        if (lane == 0) lds_st(&L.red_done[c], (unsigned)(p + 1));
        if (c == 0) {
            const unsigned long long tstart = __builtin_amdgcn_s_memrealtime();
            while (lds_ld(&L.red_done[1]) < (unsigned)(p + 1) || lds_ld(&L.red_done[2]) < (unsigned)(p + 1) || lds_ld(&L.red_done[3]) < (unsigned)(p + 1)) {
                if (lds_ld(&L.abort)) return;
                if (__builtin_amdgcn_s_memrealtime() - tstart > SPIN_LIMIT) { lds_st(&L.abort, 1u); if (lane == 0) __hip_atomic_store((gu32*)tmo, 3000u + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
            }
            asm volatile("" ::: "memory");
            if (st) t3 = __builtin_amdgcn_s_memrealtime();
            const int row = lane & 15, c0 = 8 * (lane >> 4);
            float v[8], v2[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { v[e] = 0.f; v2[e] = 0.f; }
#pragma unroll
            for (int ww = 0; ww < 4; ++ww) {
                const float4 x0 = *reinterpret_cast<const float4*>(&L.red[par][0][ww][row * 32 + c0]);
                const float4 x1 = *reinterpret_cast<const float4*>(&L.red[par][0][ww][row * 32 + c0 + 4]);
                v[0] += x0.x; v[1] += x0.y; v[2] += x0.z; v[3] += x0.w; v[4] += x1.x; v[5] += x1.y; v[6] += x1.z; v[7] += x1.w;
                if (ph.pair) {
                    const float4 y0 = *reinterpret_cast<const float4*>(&L.red[par][1][ww][row * 32 + c0]);
                    const float4 y1 = *reinterpret_cast<const float4*>(&L.red[par][1][ww][row * 32 + c0 + 4]);
                    v2[0] += y0.x; v2[1] += y0.y; v2[2] += y0.z; v2[3] += y0.w; v2[4] += y1.x; v2[5] += y1.y; v2[6] += y1.z; v2[7] += y1.w;
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = ph.pair ? v[e] * v2[e] * scale * scale * 90.f : v[e] * scale;
            u32x4 o;
            o.x = pack2(v[0], v[1]); o.y = pack2(v[2], v[3]); o.z = pack2(v[4], v[5]); o.w = pack2(v[6], v[7]);
            __builtin_amdgcn_raw_buffer_store_b128(o, act_rs, (((s_out * RG + rg) * NT + ct) * 64 + lane) * 16, 0, 16);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_store((gu32*)(flags + (size_t)p * WG + lin), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (st) {
                t4 = __builtin_amdgcn_s_memrealtime();
                unsigned long long* sp = stamps + ((size_t)p * WG + lin) * NSTAMP;
                sp[0] = t0; sp[1] = t1; sp[2] = t2; sp[3] = t3; sp[4] = t4;
            }
        }
    }
}

__global__ void fill_kernel(unsigned* p, size_t n, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        unsigned lo = 0x3c00u + (x & 0x3ffu), hi = 0x3c00u + ((x >> 10) & 0x3ffu);
        p[i] = (lo | ((x >> 20) & 1u) << 15) | ((hi | ((x >> 21) & 1u) << 15) << 16);
    }
}

int main(int argc, char** argv) {
    const int P = argc > 1 ? atoi(argv[1]) : 40;
    const int reps = argc > 2 ? atoi(argv[2]) : 50;
    const char* pat = argc > 3 ? argv[3] : "10010";               // pair pattern of a NAF block: conv1 sca conv3 conv4 conv5
    const int mode = argc > 4 ? atoi(argv[4]) : 0;
    const int noload = argc > 5 ? atoi(argv[5]) : 0;              // modes 4/5: 1 = no weight traffic                // 0: weights prefetched into registers; 1: LDS-DMA ring (nt); 2: ring, no weight traffic; 3: ring, default cache policy; 4: ring, plain A loads from per-phase slabs; 5: ring, acquire fence + plain A loads
    hipStream_t s; CK(hipStreamCreate(&s));
    std::vector<Phase> ph(P);
    const size_t plain_b = (size_t)KDIM * KDIM * 2;
    size_t total = 0;
    const int plen = (int)strlen(pat);
    for (int p = 0; p < P; ++p) {
        ph[p].pair = pat[p % plen] == '1';
        const size_t b = plain_b * (ph[p].pair ? 2 : 1);
        uint4* wp; CK(hipMalloc(&wp, b));
        hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, s, (unsigned*)wp, b / 4, 17u * p + 1);
        ph[p].W = wp; total += b;
    }
    Phase* dph; CK(hipMalloc(&dph, P * sizeof(Phase))); CK(hipMemcpy(dph, ph.data(), P * sizeof(Phase), hipMemcpyHostToDevice));
    const int nslab = (mode >= 4) ? P + 1 : 2;                       // modes 4/5: a slab per phase (no address is read twice in a launch)
    const size_t slab_b = RG * NT * 1024, act_b = (size_t)nslab * slab_b;
    uint4 *act, *act0; CK(hipMalloc(&act, act_b)); CK(hipMalloc(&act0, act_b));
    hipLaunchKernelGGL(fill_kernel, dim3(256), dim3(256), 0, s, (unsigned*)act0, act_b / 4, 4242u);
    unsigned* flags; CK(hipMalloc(&flags, (size_t)P * WG * 4));
    unsigned* tmo; CK(hipMalloc(&tmo, 64)); CK(hipMemset(tmo, 0, 64));
    unsigned long long* stamps; CK(hipMalloc(&stamps, (size_t)P * WG * NSTAMP * 8));
    CK(hipStreamSynchronize(s));
    const float scale = 2.0f;                                      // keeps magnitudes near the inputs' over the chain
    printf("persist_bench: %d phases (pattern %s), weights %.1f MB per pass, mode %d\n", P, pat, total / 1e6, mode);
    auto launch = [&](int p0, int p1, unsigned long long* stp) {
        if (mode == 0) hipLaunchKernelGGL(persist_chain, dim3(WG), dim3(512), 0, s, dph, p0, p1, act, flags, scale, tmo, stp);
        else if (mode == 3) hipLaunchKernelGGL((ring_chain<false, 0>), dim3(WG), dim3(512), 0, s, dph, p0, p1, act, flags, scale, tmo, stp, 0, nslab);
        else if (mode == 4) hipLaunchKernelGGL((ring_chain<true, 1>), dim3(WG), dim3(512), 0, s, dph, p0, p1, act, flags, scale, tmo, stp, noload, nslab);
        else if (mode == 5) hipLaunchKernelGGL((ring_chain<true, 2>), dim3(WG), dim3(512), 0, s, dph, p0, p1, act, flags, scale, tmo, stp, noload, 2);
        else hipLaunchKernelGGL((ring_chain<true, 0>), dim3(WG), dim3(512), 0, s, dph, p0, p1, act, flags, scale, tmo, stp, mode == 2 ? 1 : 0, nslab);
    };

    std::vector<unsigned> ref(slab_b / 4), got(slab_b / 4);
    const size_t fin = (size_t)(P % nslab) * slab_b;              // the last phase's output slab
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto reset = [&]() { CK(hipMemcpyAsync(act, act0, act_b, hipMemcpyDeviceToDevice, s));   /* every slab refilled: a stale read shows */ CK(hipMemsetAsync(flags, 0, (size_t)P * WG * 4, s)); };

    // (b) one launch per phase
    reset();
    for (int p = 0; p < P; ++p) launch(p, p + 1, nullptr);
    CK(hipStreamSynchronize(s));
    CK(hipMemcpy(ref.data(), (char*)act + fin, slab_b, hipMemcpyDeviceToHost));
    {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int p = 0; p < P; ++p) launch(p, p + 1, nullptr);
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0)); CK(hipGraphDestroy(g));
        reset(); CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("(b) graph of %d launches: %.2f us per pass, %.2f us per phase, %.2f TB/s\n", P, ms * 1000 / reps, ms * 1000 / reps / P, total / (ms / reps * 1e-3) / 1e12);
    }
    // (a) one persistent launch
    int bad_runs = 0;
    for (int r = 0; r < 5; ++r) {
        reset();
        launch(0, P, nullptr);
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(got.data(), (char*)act + fin, slab_b, hipMemcpyDeviceToHost));
        size_t bad = 0; for (size_t i = 0; i < got.size(); ++i) bad += got[i] != ref[i];
        if (bad) { ++bad_runs; printf("  persistent run %d: %zu of %zu words differ from the per-launch result\n", r, bad, got.size()); }
    }
    unsigned htmo = 0; CK(hipMemcpy(&htmo, tmo, 4, hipMemcpyDeviceToHost));
    { size_t nz = 0; for (size_t i = 0; i < ref.size(); ++i) nz += (ref[i] & 0x7fff7fffu) != 0; printf("    reference output: %zu of %zu words non-zero, word[5] = %08x\n", nz, ref.size(), ref[5]); }
    printf("(a) persistent vs per-launch: %s; timeout word %u\n", bad_runs ? "MISMATCH" : "bit-identical (5 runs)", htmo);
    if (htmo) { printf("a wait timed out; no timing\n"); return 1; }
    {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        CK(hipMemsetAsync(flags, 0, (size_t)P * WG * 4, s));
        launch(0, P, nullptr);
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0)); CK(hipGraphDestroy(g));
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("(a) persistent launch (+memset node): %.2f us per pass, %.2f us per phase, %.2f TB/s\n", ms * 1000 / reps, ms * 1000 / reps / P, total / (ms / reps * 1e-3) / 1e12);
        // after timing: same data each replay (act ping-pong ends where it started only for even P; compare anyway after reset)
        reset();
        launch(0, P, nullptr);
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(got.data(), (char*)act + fin, slab_b, hipMemcpyDeviceToHost));
        size_t bad = 0; for (size_t i = 0; i < got.size(); ++i) bad += got[i] != ref[i];
        printf("    after timing: %zu words differ\n", bad);
    }
    // stamps
    reset();
    CK(hipMemsetAsync(stamps, 0, (size_t)P * WG * NSTAMP * 8, s));
    launch(0, P, stamps);
    CK(hipStreamSynchronize(s));
    std::vector<unsigned long long> h((size_t)P * WG * NSTAMP);
    CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
    const char* nm[4] = {"wait", "A+mfma", mode ? "red-wait" : "wload+red+sync", "epi+publish"};
    double tot[4] = {0, 0, 0, 0};
    for (int p = 1; p < P; ++p) {
        double med[4];
        for (int k = 0; k < 4; ++k) {
            std::vector<double> d;
            for (int wg = 0; wg < WG; ++wg) { const unsigned long long* t = &h[((size_t)p * WG + wg) * NSTAMP]; d.push_back((double)(t[k + 1] - t[k]) * 0.01); }
            std::sort(d.begin(), d.end()); med[k] = d[d.size() / 2]; tot[k] += med[k];
        }
        if (p <= 10) printf("    phase %2d (%s): wait %.2f  A+mfma %.2f  red %.2f  epi+publish %.2f\n", p, ph[p].pair ? "pair " : "plain", med[0], med[1], med[2], med[3]);
    }
    printf("    median over workgroups, mean over phases 1..%d:", P - 1);
    for (int k = 0; k < 4; ++k) printf(" %s %.2f", nm[k], tot[k] / (P - 1));
    unsigned long long tmin = ~0ull, tmax = 0;
    for (int wg = 0; wg < WG; ++wg) { tmin = std::min(tmin, h[(size_t)wg * NSTAMP]); tmax = std::max(tmax, h[((size_t)(P - 1) * WG + wg) * NSTAMP + 4]); }
    printf("\n    stamped span %.2f us (%.2f per phase)\n", (double)(tmax - tmin) * 0.01, (double)(tmax - tmin) * 0.01 / P);
    return 0;
}
