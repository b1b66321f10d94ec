// xcd_bench.hip — what a phase boundary costs INSIDE one XCD (tools only; VERDICT r02 "next" item 1, first step).
//
// 256 workgroups (one per CU) group themselves by the XCD they landed on (HW_REG_XCC_ID) and run a chain of dependent
// phases.  Every phase: (optionally) stream a slice of a shared weight buffer, wait for the group's previous phase,
// read what all 32 workgroups of the group published, publish an own tile, arrive.  Nothing crosses an XCD: the 32 CUs
// of an XCD share one L2, so the hand-off can use forms that are only valid there:
//
//   mode 0  arrive = atomic add WITHOUT sc1 (executes in the XCD's L2), poll = sc1 load of that counter,
//           payload = plain stores (stay in L2) drained by every storing wave, read back with sc1 loads (L1 bypass)
//   mode 1  arrive = plain store of a per-workgroup flag word (one 128-byte line per XCD holds the 32 flags),
//           poll = ONE 32-lane sc1 load of that line; payload as mode 0
//   mode 2  the placement-independent form for comparison: sc1 (write-through) payload stores, agent-scope atomic
//           add, sc1 poll, sc1 loads (MI355X_MICROARCH.md "Valid forms" row 1)
//
// Every word read back is checked against what its producer must have written in that phase (stale data = an error
// count, reported).  Timing: s_memrealtime per workgroup around the phase loop and around its segments.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o gpurun_out/xcd_bench tools/xcd_bench.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) unsigned gu32;
typedef __attribute__((address_space(1))) const u32x4 gc_u32x4;

constexpr int THREADS = 512;
constexpr int GROUP = 32;                  // workgroups per XCD
constexpr int MAXW = 16;                   // weight loads (16 B each) per thread per phase
constexpr unsigned SPIN_LIMIT = 1u << 22;

struct BenchP {
    int mode, phases;
    int payload_u4;                        // uint4 per workgroup per phase (multiple of THREADS or 0)
    int w_loads;                           // 16-byte weight loads per thread per phase (<= MAXW), 0 = none
    int w_nt;                              // weight loads non-temporal
    int skew;                              // workgroup rank r sleeps (r % 4) * skew units in every phase (uneven load)
    const uint4* W; size_t w_phase_u4;     // weight slice of phase ph starts at ph * w_phase_u4 (wraps at w_total_u4)
    size_t w_total_u4;
    uint4* slab;                           // [2][8][GROUP][payload_u4]
    unsigned* cnt;                         // [8][32] words: word 0 of each 128-byte line is the XCD's arrive counter
    unsigned* flags;                       // [8][32] words: one line per XCD, word r = workgroup r
    unsigned* ticket;                      // [8] rank dispenser
    unsigned* err;                         // [0] mismatching words, [1] timeouts, [2] overflow workgroups
    unsigned* xcc_of;                      // [grid] XCC id each workgroup saw
    unsigned long long* times;             // [grid][6]: total, wait, read, weights, store+drain, arrive (10 ns ticks)
    float* sink;
};

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}
__device__ __forceinline__ unsigned long long now() { return __builtin_amdgcn_s_memrealtime(); }

__global__ __launch_bounds__(THREADS) void xcd_chain(const BenchP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];            // only there to keep one workgroup per CU
    __shared__ unsigned s_rank, s_abort;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned xcc = xcc_id();
    if (tid == 0) {
        s_rank = __hip_atomic_fetch_add((gu32*)(p.ticket + xcc), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_abort = 0;
        p.xcc_of[blockIdx.x] = xcc;
    }
    __syncthreads();
    const unsigned rank = s_rank;
    if (rank >= (unsigned)GROUP) {                                          // more than 32 workgroups on this XCD: not part of a group
        if (tid == 0) atomicAdd(p.err + 2, 1u);
        return;
    }
    gu32* cnt = (gu32*)(p.cnt + xcc * 32);
    gu32* flags = (gu32*)(p.flags + xcc * 32);
    const __amdgpu_buffer_rsrc_t slab_rs = __builtin_amdgcn_make_buffer_rsrc(p.slab, 0, (int)((size_t)2 * 8 * GROUP * p.payload_u4 * 16), 0x00020000);
    const int grp_u4 = GROUP * p.payload_u4;
    unsigned long long t_wait = 0, t_read = 0, t_w = 0, t_store = 0, t_arr = 0;
    unsigned bad = 0;
    float facc = 0.f;
    const unsigned long long t_begin = now();

    for (int ph = 0; ph < p.phases; ++ph) {
        unsigned long long a0 = now();
        // ---- weights of this phase: requested before the wait (they depend on nothing) ----
        u32x4 w[MAXW];
#pragma unroll
        for (int i = 0; i < MAXW; ++i) w[i] = (u32x4){0u, 0u, 0u, 0u};
        if (p.w_loads > 0) {
            size_t base = ((size_t)ph * p.w_phase_u4) % p.w_total_u4 + (size_t)rank * p.w_loads * THREADS;
#pragma unroll
            for (int i = 0; i < MAXW; ++i) {
                if (i < p.w_loads) {
                    gc_u32x4* q = (gc_u32x4*)(unsigned long long)(p.W + base + (size_t)i * THREADS + tid);
                    w[i] = p.w_nt ? __builtin_nontemporal_load(q) : *q;
                }
            }
        }
        if (p.skew > 0) for (unsigned k = 0; k < (rank & 3u) * (unsigned)p.skew; ++k) __builtin_amdgcn_s_sleep(8);
        // ---- wait for phase ph - 1 of the whole group ----
        if (ph > 0) {
            if (wave == 0) {
                const unsigned want = (p.mode == 1) ? (unsigned)ph : (unsigned)(GROUP * ph);
                for (unsigned spins = 0;; ++spins) {
                    bool ok;
                    if (p.mode == 1) {
                        const unsigned v = lane < GROUP ? __hip_atomic_load(flags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : want;
                        ok = __all((int)(v - want) >= 0);
                    } else {
                        const unsigned v = __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ok = (int)(v - want) >= 0;
                    }
                    if (ok) break;
                    if (spins > SPIN_LIMIT) { if (lane == 0) { s_abort = 1; atomicAdd(p.err + 1, 1u); } break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            __syncthreads();
            if (s_abort) return;
        }
        unsigned long long a1 = now();
        // ---- read what the 32 workgroups of the group published in phase ph - 1 ----
        if (ph > 0 && p.payload_u4 > 0) {
            const int slab_off = (((ph - 1) & 1) * 8 + (int)xcc) * grp_u4;
            for (int i0 = 0; i0 < grp_u4; i0 += THREADS * 8) {
                u32x4 v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int i = i0 + j * THREADS + tid;
                    v[j] = (i < grp_u4) ? __builtin_amdgcn_raw_buffer_load_b128(slab_rs, (slab_off + i) * 16, 0, 16) : (u32x4){0u, 0u, 0u, 0u};
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int i = i0 + j * THREADS + tid;
                    if (i < grp_u4) {
                        const unsigned want = ((unsigned)ph << 20) ^ (unsigned)i;       // producer rank is i / payload_u4: part of i
                        bad += (v[j].x != want) + (v[j].y != (want ^ 0x11111111u)) + (v[j].z != (want ^ 0x22222222u)) + (v[j].w != (want ^ 0x33333333u));
                    }
                }
            }
        }
        unsigned long long a2 = now();
        // ---- consume the weights ----
#pragma unroll
        for (int i = 0; i < MAXW; ++i) facc += __uint_as_float((w[i].x ^ w[i].y ^ w[i].z ^ w[i].w) & 0x3fffffffu);
        unsigned long long a3 = now();
        // ---- publish this workgroup's tile of phase ph ----
        if (p.payload_u4 > 0) {
            const int slab_off = ((ph & 1) * 8 + (int)xcc) * grp_u4 + (int)rank * p.payload_u4;
            for (int i = tid; i < p.payload_u4; i += THREADS) {
                const unsigned val = ((unsigned)(ph + 1) << 20) ^ (unsigned)((int)rank * p.payload_u4 + i);
                const u32x4 v = {val, val ^ 0x11111111u, val ^ 0x22222222u, val ^ 0x33333333u};
                if (p.mode == 2) __builtin_amdgcn_raw_buffer_store_b128(v, slab_rs, (slab_off + i) * 16, 0, 16);      // sc1: write-through
                else __builtin_amdgcn_raw_buffer_store_b128(v, slab_rs, (slab_off + i) * 16, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // every storing wave drains its stores
        __syncthreads();
        unsigned long long a4 = now();
        if (tid == 0) {
            if (p.mode == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);        // no sc1: executes in this XCD's L2
            else if (p.mode == 1) __hip_atomic_store(flags + rank, (unsigned)(ph + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // plain store
            else __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        unsigned long long a5 = now();
        t_w += a3 - a2; t_wait += a1 - a0; t_read += a2 - a1; t_store += a4 - a3; t_arr += a5 - a4;
    }
    const unsigned long long t_end = now();
    if (bad) atomicAdd(p.err, bad);
    if (tid == 0) {
        unsigned long long* o = p.times + (size_t)blockIdx.x * 6;
        o[0] = t_end - t_begin; o[1] = t_wait; o[2] = t_read; o[3] = t_w; o[4] = t_store; o[5] = t_arr;
    }
    if (facc == 123.456f) p.sink[0] = facc + smem[0];
}

static void run(const char* label, BenchP p, int reps, hipStream_t st, std::vector<unsigned long long>& h_times, unsigned* d_err) {
    const int grid = 256;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&xcd_chain), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double best_ms = 1e9;
    unsigned h_err[4] = {0, 0, 0, 0};
    std::vector<unsigned> h_xcc(grid);
    for (int r = 0; r < reps; ++r) {
        CK(hipMemsetAsync(p.cnt, 0, 8 * 32 * 4, st)); CK(hipMemsetAsync(p.flags, 0, 8 * 32 * 4, st)); CK(hipMemsetAsync(p.ticket, 0, 8 * 4, st));
        CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL(xcd_chain, dim3(grid), dim3(THREADS), 100 * 1024, st, p);
        CK(hipGetLastError());
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best_ms = std::min(best_ms, (double)ms);
    }
    CK(hipMemcpy(h_err, d_err, sizeof(h_err), hipMemcpyDeviceToHost));
    CK(hipMemcpy(h_times.data(), p.times, (size_t)grid * 6 * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h_xcc.data(), p.xcc_of, grid * 4, hipMemcpyDeviceToHost));
    int per[8] = {0, 0, 0, 0, 0, 0, 0, 0}, mod_ok = 0;
    for (int b = 0; b < grid; ++b) { per[h_xcc[b] & 7]++; mod_ok += (h_xcc[b] == h_xcc[b & 7]); }
    std::vector<double> tot(grid), seg[5];
    for (int b = 0; b < grid; ++b) {
        tot[b] = h_times[(size_t)b * 6] * 0.01 / p.phases;
        for (int k = 0; k < 5; ++k) seg[k].push_back(h_times[(size_t)b * 6 + 1 + k] * 0.01 / p.phases);
    }
    auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    auto mx = [](const std::vector<double>& v) { return *std::max_element(v.begin(), v.end()); };
    printf("%-44s us/phase median %6.3f max %6.3f | wait %5.2f read %5.2f weights %5.2f store+drain %5.2f arrive %5.2f | kernel %8.1f us | "
           "bad words %u timeouts %u overflow %u | per-XCD %d %d %d %d %d %d %d %d, same-XCD-as-(b%%8) %d/256\n",
           label, med(tot), mx(tot), med(seg[0]), med(seg[1]), med(seg[2]), med(seg[3]), med(seg[4]), best_ms * 1e3, h_err[0], h_err[1], h_err[2],
           per[0], per[1], per[2], per[3], per[4], per[5], per[6], per[7], mod_ok);
    CK(hipMemset(d_err, 0, 16));
}

int main(int argc, char** argv) {
    const int phases = argc > 1 ? atoi(argv[1]) : 200;
    const int reps = argc > 2 ? atoi(argv[2]) : 5;
    hipStream_t st; CK(hipStreamCreate(&st));
    BenchP p{};
    p.phases = phases;
    const size_t w_total_u4 = ((size_t)160 << 20) / 16;                     // 160 MB: level 3's weights are 147 MB
    const int max_payload_u4 = 512;                                         // 8 KB per workgroup per phase
    uint4* dW; CK(hipMalloc(&dW, w_total_u4 * 16)); CK(hipMemset(dW, 0x11, w_total_u4 * 16));
    CK(hipMalloc(&p.slab, (size_t)2 * 8 * GROUP * max_payload_u4 * 16)); CK(hipMemset(p.slab, 0, (size_t)2 * 8 * GROUP * max_payload_u4 * 16));
    CK(hipMalloc(&p.cnt, 8 * 32 * 4)); CK(hipMalloc(&p.flags, 8 * 32 * 4)); CK(hipMalloc(&p.ticket, 8 * 4));
    unsigned* d_err; CK(hipMalloc(&d_err, 16)); CK(hipMemset(d_err, 0, 16)); p.err = d_err;
    CK(hipMalloc(&p.xcc_of, 256 * 4)); CK(hipMalloc(&p.times, 256 * 6 * 8)); CK(hipMalloc(&p.sink, 16));
    p.W = dW; p.w_total_u4 = w_total_u4;
    std::vector<unsigned long long> h_times(256 * 6);
    char label[128];
    for (int mode = 0; mode < 3; ++mode) {
        for (int payload_u4 : {0, 128, 256, 512}) {                         // 0 / 2 / 4 / 8 KB per workgroup: the group re-reads 32x that
            for (int wl : {0, 8, 16}) {                                     // 0 / 64 / 128 KB of weights per workgroup per phase
                if (payload_u4 == 0 && wl != 0) continue;
                for (int nt = 0; nt < (wl ? 2 : 1); ++nt) {
                    p.mode = mode; p.payload_u4 = payload_u4; p.w_loads = wl; p.w_nt = nt; p.skew = 0;
                    p.w_phase_u4 = (size_t)GROUP * wl * THREADS;
                    snprintf(label, sizeof(label), "mode %d payload %d KB weights %3d KB%s", mode, payload_u4 * 16 / 1024, wl * THREADS * 16 / 1024, wl ? (nt ? " nt" : " default") : "");
                    run(label, p, reps, st, h_times, d_err);
                }
            }
        }
        // uneven load: ranks sleep 0..3 units before the wait
        p.mode = mode; p.payload_u4 = 128; p.w_loads = 8; p.w_nt = 1; p.skew = 4; p.w_phase_u4 = (size_t)GROUP * 8 * THREADS;
        snprintf(label, sizeof(label), "mode %d payload 2 KB weights 64 KB nt SKEWED", mode);
        run(label, p, reps, st, h_times, d_err);
    }
    return 0;
}
