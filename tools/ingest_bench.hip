// ingest_bench.hip — how fast can the 256 CUs take in weight tiles in steady state (tools only)?
// Every workgroup (512 threads) pulls one 128 KiB tile per round into registers (16 x 1 KiB per wave, the
// skinny GEMM's fragment stream), `red` workgroups of one XCD pull the SAME tile (the row groups of a weight
// tile), rounds run back to back inside one launch (no kernel boundary, no dependency between rounds).
// Reports us per round, GB/s per CU and unique TB/s for red in {1, 2, 4}, nt / default cache policy.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/ib tools/ingest_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <bool NT, int DEPTH>
__global__ __launch_bounds__(512) void ingest(const u32x4* __restrict__ arena, int rounds, int red, int tiles_per_round,
                                              unsigned long long* stamps, unsigned* sink) {
    const int tid = threadIdx.x, lane = tid & 63, wk = tid >> 6;
    const int lin = blockIdx.x, xcd = lin & 7, j = lin >> 3;            // 32 workgroups per XCD
    const int tile = (j / red) * 8 + xcd;                                // `red` consecutive j share a tile
    unsigned acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int r0 = 0; r0 < rounds; r0 += DEPTH) {
        u32x4 w[DEPTH][16];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const u32x4* p = arena + ((size_t)(r0 + d) * tiles_per_round + tile) * (128 * 64) + (16 * wk) * 64 + lane;
#pragma unroll
            for (int s = 0; s < 16; ++s) w[d][s] = NT ? __builtin_nontemporal_load(p + s * 64) : p[s * 64];
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
#pragma unroll
            for (int s = 0; s < 16; ++s) acc ^= w[d][s].x ^ w[d][s].w;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (acc == 0x1234567u) sink[0] = acc;
    if (lane == 0) { atomicMin(&stamps[2 * lin], t0); atomicMax(&stamps[2 * lin + 1], t1); }
}

// variants at red = 4: MODE 1: the sharers start at different quarters of the tile (no two CUs ask for the same line at
// the same time); MODE 2: the same tile every round (L2-resident after round 0); MODE 3: LDS-DMA instead of registers.
template <int MODE, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void ingest_v(const u32x4* __restrict__ arena, int rounds, int red, int tiles_per_round,
                                                       unsigned long long* stamps, unsigned* sink) {
    __shared__ __attribute__((aligned(16))) u32x4 lbuf[MODE == 3 ? 128 * 64 : 1];
    constexpr int PER = 128 / WAVES;                                      // 1 KiB fragments per wave per round
    const int tid = threadIdx.x, lane = tid & 63, wk = tid >> 6;
    const int lin = blockIdx.x, xcd = lin & 7, j = lin >> 3;
    const int tile = (j / red) * 8 + xcd, sharer = j % red;
    unsigned acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int r = 0; r < rounds; ++r) {
        int rr = r;
        if (MODE == 2) { rr = 0; asm volatile("" : "+s"(rr)); }
        const u32x4* base = arena + ((size_t)rr * tiles_per_round + tile) * (128 * 64) + lane;
        if (MODE == 3) {
#pragma unroll
            for (int s = 0; s < PER; ++s) {
                const int f = wk * PER + s;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + f * 64),
                                                 (__attribute__((address_space(3))) void*)&lbuf[f * 64], 16, 0, 2);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            u32x4 w[PER];
#pragma unroll
            for (int s = 0; s < PER; ++s) {
                int f = wk * PER + s;
                if (MODE == 1) f = (f + sharer * (128 / 4)) & 127;
                w[s] = __builtin_nontemporal_load(base + f * 64);
            }
#pragma unroll
            for (int s = 0; s < PER; ++s) acc ^= w[s].x ^ w[s].w;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (MODE == 3) acc = lbuf[tid].x;
    if (acc == 0x1234567u) sink[0] = acc;
    if (lane == 0) { atomicMin(&stamps[2 * lin], t0); atomicMax(&stamps[2 * lin + 1], t1); }
}

// leader / follower: sharer 0 of every tile runs `lag` x 0.64 us ahead of the other sharers (s_sleep at kernel start), so the
// followers find the lines in L2 instead of queueing on the leader's outstanding misses.  LEADER_NT: policy of the leader's loads.
template <bool LEADER_NT, bool FOLLOWER_NT>
__global__ __launch_bounds__(512) void ingest_lf(const u32x4* __restrict__ arena, int rounds, int red, int tiles_per_round,
                                                 unsigned long long* stamps, unsigned* sink, int lag) {
    const int tid = threadIdx.x, lane = tid & 63, wk = tid >> 6;
    const int lin = blockIdx.x, xcd = lin & 7, j = lin >> 3;
    const int tile = (j / red) * 8 + xcd, sharer = j % red;
    unsigned acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (sharer != 0) for (int i = 0; i < lag; ++i) __builtin_amdgcn_s_sleep(24);      // 24 x 64 cycles = 0.64 us at 2.4 GHz
    for (int r = 0; r < rounds; ++r) {
        const u32x4* p = arena + ((size_t)r * tiles_per_round + tile) * (128 * 64) + (16 * wk) * 64 + lane;
        u32x4 w[16];
        if (sharer == 0) {
#pragma unroll
            for (int s = 0; s < 16; ++s) w[s] = LEADER_NT ? __builtin_nontemporal_load(p + s * 64) : p[s * 64];
        } else {
#pragma unroll
            for (int s = 0; s < 16; ++s) w[s] = FOLLOWER_NT ? __builtin_nontemporal_load(p + s * 64) : p[s * 64];
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) acc ^= w[s].x ^ w[s].w;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (acc == 0x1234567u) sink[0] = acc;
    if (lane == 0) { atomicMin(&stamps[2 * lin], t0); atomicMax(&stamps[2 * lin + 1], t1); }
}

__global__ void fill_kernel(unsigned* p, size_t n, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (unsigned)i * 2654435761u + seed;
}

int main() {
    const int rounds = 48;
    const size_t tile_b = 128 * 1024;
    const size_t arena_b = (size_t)rounds * 256 * tile_b;              // red = 1: 256 tiles per round (1.6 GB)
    u32x4* arena; CK(hipMalloc(&arena, arena_b));
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, (unsigned*)arena, arena_b / 4, 7u);
    unsigned long long* stamps; CK(hipMalloc(&stamps, 256 * 16));
    unsigned* sink; CK(hipMalloc(&sink, 64));
    u32x4* flush; CK(hipMalloc(&flush, (size_t)512 << 20));
    CK(hipDeviceSynchronize());
    auto run = [&](const char* name, auto kern, int red, int threads = 512) {
        const int tiles = 256 / red;
        double best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, (unsigned*)flush, ((size_t)512 << 20) / 4, 9u + rep);   // evict the caches
            { std::vector<unsigned long long> init(512); for (int i = 0; i < 256; ++i) { init[2 * i] = ~0ull; init[2 * i + 1] = 0; } CK(hipMemcpy(stamps, init.data(), 512 * 8, hipMemcpyHostToDevice)); }
            hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, arena, rounds, red, tiles, stamps, sink);
            CK(hipDeviceSynchronize());
            std::vector<unsigned long long> h(512);
            CK(hipMemcpy(h.data(), stamps, 512 * 8, hipMemcpyDeviceToHost));
            unsigned long long a = ~0ull, b = 0;
            for (int i = 0; i < 256; ++i) { a = std::min(a, h[2 * i]); b = std::max(b, h[2 * i + 1]); }
            best = std::min(best, (double)(b - a) * 0.01 / rounds);
        }
        printf("%-34s red %d: %6.2f us per round | %6.1f GB/s per CU | unique %5.2f TB/s\n", name, red, best, tile_b / best / 1e3, tiles * tile_b / best / 1e6);
    };
    for (int red : {1, 2, 4}) {
        run("nt,      1 round in flight", ingest<true, 1>, red);
        run("default, 1 round in flight", ingest<false, 1>, red);
        run("nt,      2 rounds in flight", ingest<true, 2>, red);
        run("default, 2 rounds in flight", ingest<false, 2>, red);
    }
    for (int lag : {1, 2, 4, 8}) {
        char nm[64];
        auto runlf = [&](const char* name, auto kern) {
            const int red = 4, tiles = 64;
            double best = 1e9;
            for (int rep = 0; rep < 3; ++rep) {
                hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, (unsigned*)flush, ((size_t)512 << 20) / 4, 9u + rep);
                { std::vector<unsigned long long> init(512); for (int i = 0; i < 256; ++i) { init[2 * i] = ~0ull; init[2 * i + 1] = 0; } CK(hipMemcpy(stamps, init.data(), 512 * 8, hipMemcpyHostToDevice)); }
                hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, arena, rounds, red, tiles, stamps, sink, lag);
                CK(hipDeviceSynchronize());
                std::vector<unsigned long long> h(512);
                CK(hipMemcpy(h.data(), stamps, 512 * 8, hipMemcpyDeviceToHost));
                unsigned long long a = ~0ull, b = 0;
                for (int i = 0; i < 256; ++i) { a = std::min(a, h[2 * i]); b = std::max(b, h[2 * i + 1]); }
                best = std::min(best, ((double)(b - a) * 0.01 - lag * 0.64) / rounds);
            }
            printf("%-34s red 4: %6.2f us per round | %6.1f GB/s per CU | unique %5.2f TB/s\n", name, best, tile_b / best / 1e3, tiles * tile_b / best / 1e6);
        };
        snprintf(nm, sizeof nm, "leader default, fol. default, lag %d", lag); runlf(nm, ingest_lf<false, false>);
        snprintf(nm, sizeof nm, "leader default, fol. nt,      lag %d", lag); runlf(nm, ingest_lf<false, true>);
        snprintf(nm, sizeof nm, "leader nt,      fol. nt,      lag %d", lag); runlf(nm, ingest_lf<true, true>);
    }
    run("nt, sharers rotated by a quarter", ingest_v<1, 8>, 4);
    run("nt, same tile every round (L2)", ingest_v<2, 8>, 4);
    run("nt, same tile every round (L2)", ingest_v<2, 8>, 1);
    run("LDS-DMA nt, 8 waves", ingest_v<3, 8>, 4);
    run("LDS-DMA nt, 4 waves", ingest_v<3, 4>, 4, 256);
    run("LDS-DMA nt, 8 waves", ingest_v<3, 8>, 1);
    run("nt regs, 4 waves x 32", ingest_v<0, 4>, 4, 256);
    run("nt regs, 16 waves x 8", ingest_v<0, 16>, 4, 1024);
    run("nt regs, 16 waves x 8", ingest_v<0, 16>, 1, 1024);
    run("nt regs, 16 waves x 8", ingest_v<0, 16>, 2, 1024);
    run("nt regs, 8 waves x 16 (ingest_v)", ingest_v<0, 8>, 4);
    run("nt regs, 2 waves x 64", ingest_v<0, 2>, 4, 128);
    run("LDS-DMA nt, 16 waves", ingest_v<3, 16>, 4, 1024);
    run("LDS-DMA nt, 2 waves", ingest_v<3, 2>, 4, 128);
    run("LDS-DMA nt, 1 wave", ingest_v<3, 1>, 4, 64);
    return 0;
}
