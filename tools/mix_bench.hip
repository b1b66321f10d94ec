// mix_bench.hip — what one middle-level GEMM launch has to take in, per tiling (tools only).
// The middle level's GEMMs are [64 rows x 2048] x [2048 x N]; with 256 workgroups the per-workgroup operand bytes are
//   rows x columns per workgroup      weights (bf16, sharers on one XCD)      activations (bf16 rows, L2 after the XCD's first read)
//   16 x 32 (what the library runs)   plain 128 KiB / pair 256 KiB, 4 share    64 KiB
//   32 x 16                           plain  64 KiB / pair 128 KiB, 2 share   128 KiB
//   64 x  8 ("unshared weights")      plain  32 KiB / pair  64 KiB, unshared  256 KiB
// This bench issues exactly those loads (16-byte lanes, whole 1 KiB fragments, weights non-temporal and cold: a new arena slice
// per launch; activations from one 256 KiB buffer per launch that every workgroup reads) and nothing else -- no LayerNorm, no
// MFMA, no epilogue -- as (a) back-to-back launches on one stream, timed with events (what a launch costs including its
// boundary and cold start), and (b) rounds inside one launch (steady-state ingest).  If the unshared tiling is not clearly
// faster HERE it cannot be faster as a GEMM.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/mix_bench_bin tools/mix_bench.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// WF / AF: 1 KiB fragments of weights / activations per wave per round (8 waves): WF = 32 -> 256 KiB per workgroup
template <int WF, int AF>
__global__ __launch_bounds__(512) void mix(const u32x4* __restrict__ warena, const u32x4* __restrict__ abuf, int rounds, int red, int tiles_per_round,
                                           size_t round0, unsigned long long* stamps, unsigned* sink) {
    const int tid = threadIdx.x, lane = tid & 63, wk = tid >> 6;
    const int lin = blockIdx.x, xcd = lin & 7, j = lin >> 3;            // 32 workgroups per XCD
    const int tile = (j / red) * 8 + xcd, sharer = j % red;              // `red` workgroups of one XCD share a weight tile
    unsigned acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int r = 0; r < rounds; ++r) {
        const u32x4* wp = warena + ((round0 + r) * tiles_per_round + tile) * (size_t)(WF * 8 * 64) + (size_t)(WF * wk) * 64 + lane;
        // activations: AF * 8 KiB of this round's 256 KiB buffer, the part that belongs to this row group
        const u32x4* ap = abuf + (round0 + r) * (size_t)(256 * 64) + (size_t)((sharer * AF * 8) % 256) * 64 + (size_t)(AF * wk) * 64 + lane;
        u32x4 w[WF], a[AF];
#pragma unroll
        for (int s = 0; s < AF; ++s) a[s] = ap[s * 64];
#pragma unroll
        for (int s = 0; s < WF; ++s) w[s] = __builtin_nontemporal_load(wp + s * 64);
#pragma unroll
        for (int s = 0; s < AF; ++s) acc ^= a[s].x ^ a[s].w;
#pragma unroll
        for (int s = 0; s < WF; ++s) acc ^= w[s].x ^ w[s].w;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (acc == 0x1234567u) sink[0] = acc;
    if (stamps && lane == 0) { atomicMin(&stamps[2 * lin], t0); atomicMax(&stamps[2 * lin + 1], t1); }
}

// XCD-local split-K cluster: 4 workgroups with equal blockIdx.x % 8 (one XCD under round-robin dispatch) share a
// 64-row x 32-column (pair: 2 x 32) tile, each takes K / 4: WF fragments of weights (unshared) + 64 KiB of activations (its K quarter
// of all 64 rows); then the exchange a GEMM would need -- every workgroup stores its fp32 partial tile (8 / 16 KiB, plain stores: the
// sharers sit behind one L2), drains, raises its flag in the cluster's line; wave 0 polls the four flags with sc1 loads; every
// workgroup reads the three foreign partials of ITS quarter of the rows with sc1 loads.  Loads + exchange only, as above.
template <int WF, int PK>                                   // PK: KiB of partial tile per workgroup (8 plain, 16 pair)
__global__ __launch_bounds__(512) void mix_splitk(const u32x4* __restrict__ warena, const u32x4* __restrict__ abuf, u32x4* part, unsigned* flags, int rounds,
                                                  size_t round0, unsigned long long* stamps, unsigned* sink) {
    const int tid = threadIdx.x, lane = tid & 63, wk = tid >> 6;
    const int lin = blockIdx.x, xcd = lin & 7, j = lin >> 3;
    const int cluster = (j >> 2) * 8 + xcd, member = j & 3;            // 64 clusters of 4
    unsigned acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int r = 0; r < rounds; ++r) {
        const size_t rr = round0 + r;
        const u32x4* wp = warena + (rr * 256 + lin) * (size_t)(WF * 8 * 64) + (size_t)(WF * wk) * 64 + lane;
        const u32x4* ap = abuf + rr * (size_t)(256 * 64) + (size_t)(member * 64) * 64 + (size_t)(8 * wk) * 64 + lane;   // K quarter `member` of all rows: 64 KiB
        u32x4 w[WF], a[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) a[s] = ap[s * 64];
#pragma unroll
        for (int s = 0; s < WF; ++s) w[s] = __builtin_nontemporal_load(wp + s * 64);
#pragma unroll
        for (int s = 0; s < 8; ++s) acc ^= a[s].x ^ a[s].w;
#pragma unroll
        for (int s = 0; s < WF; ++s) acc ^= w[s].x ^ w[s].w;
        // partial tile out: PK KiB per workgroup = PK / 8 sixteen-byte stores per lane
        u32x4* mine = part + ((rr & 1) * 256 + lin) * (size_t)(PK * 64);
#pragma unroll
        for (int s = 0; s < PK / 8; ++s) mine[(s * 8 + wk) * 64 + lane] = (u32x4){acc, acc + 1u, acc + 2u, acc + 3u};
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        unsigned* fl = flags + cluster * 32;
        if (tid == 0) __hip_atomic_store(fl + member, (unsigned)rr + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (wk == 0) {
            for (int spins = 0; spins < (1 << 20); ++spins) {
                const unsigned v = lane < 4 ? __hip_atomic_load(fl + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (unsigned)rr + 1u;
                if (__all((int)(v - ((unsigned)rr + 1u)) >= 0)) break;
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        // the three foreign partials of this workgroup's quarter of the rows: 3 x PK / 4 KiB
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(part, 0, 2 * 256 * PK * 1024, 0x00020000);
        if (tid < PK * 16) {                                            // PK / 4 KiB = PK * 16 sixteen-byte units per foreign partial
#pragma unroll
            for (int m = 1; m < 4; ++m) {
                const int other = (j & ~3 | ((member + m) & 3)) * 8 + xcd;
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)((((rr & 1) * 256 + other) * (size_t)(PK * 64) + member * (PK * 16) + tid) * 16), 0, 16);
                acc ^= v.x ^ v.w;
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (acc == 0x1234567u) sink[0] = acc;
    if (stamps && lane == 0) { atomicMin(&stamps[2 * lin], t0); atomicMax(&stamps[2 * lin + 1], t1); }
}

__global__ void fill_kernel(unsigned* p, size_t n, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (unsigned)i * 2654435761u + seed;
}

int main() {
    const int rounds = 40;                                              // 40 launches = the middle level of one diffusion step
    const size_t w_round = (size_t)64 << 20;                            // room per round: 256 workgroups x 256 KiB (largest case, unshared)
    u32x4 *warena, *abuf, *flush;
    CK(hipMalloc(&warena, rounds * w_round));
    CK(hipMalloc(&abuf, (size_t)rounds * 256 * 1024));
    CK(hipMalloc(&flush, (size_t)512 << 20));
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, (unsigned*)warena, rounds * w_round / 4, 7u);
    hipLaunchKernelGGL(fill_kernel, dim3(256), dim3(256), 0, 0, (unsigned*)abuf, (size_t)rounds * 256 * 1024 / 4, 8u);
    unsigned long long* stamps; CK(hipMalloc(&stamps, 256 * 16));
    unsigned* sink; CK(hipMalloc(&sink, 64));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipDeviceSynchronize());
    auto evict = [&](int rep) { hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, st, (unsigned*)flush, ((size_t)512 << 20) / 4, 9u + rep); };
    auto run = [&](const char* name, auto kern, int wkb, int akb, int red) {
        const int tiles = 256 / red;
        double best_launch = 1e9, best_round = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            // (a) one launch per round
            evict(rep);
            CK(hipEventRecord(e0, st));
            for (int r = 0; r < rounds; ++r) hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, st, warena, abuf, 1, red, tiles, (size_t)r, (unsigned long long*)nullptr, sink);
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            best_launch = std::min(best_launch, (double)ms * 1e3 / rounds);
            // (b) all rounds inside one launch
            evict(rep + 3);
            { std::vector<unsigned long long> init(512); for (int i = 0; i < 256; ++i) { init[2 * i] = ~0ull; init[2 * i + 1] = 0; } CK(hipMemcpyAsync(stamps, init.data(), 512 * 8, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st)); }
            hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, st, warena, abuf, rounds, red, tiles, (size_t)0, stamps, sink);
            CK(hipStreamSynchronize(st));
            std::vector<unsigned long long> h(512);
            CK(hipMemcpy(h.data(), stamps, 512 * 8, hipMemcpyDeviceToHost));
            unsigned long long a = ~0ull, b = 0;
            for (int i = 0; i < 256; ++i) { a = std::min(a, h[2 * i]); b = std::max(b, h[2 * i + 1]); }
            best_round = std::min(best_round, (double)(b - a) * 0.01 / rounds);
        }
        printf("%-44s W %3d KiB (x%d share) + A %3d KiB per workgroup: %6.2f us per launch | %6.2f us per round in one launch (%5.1f GB/s per CU)\n", name, wkb, red, akb,
               best_launch, best_round, (wkb + akb) * 1024.0 / best_round / 1e3);
    };
    run("empty launch (1 fragment each)", mix<1, 1>, 8, 8, 1);
    run("pair  16 x 32 (library)", mix<32, 8>, 256, 64, 4);
    run("pair  32 x 16", mix<16, 16>, 128, 128, 2);
    run("pair  64 x  8 (unshared)", mix<8, 32>, 64, 256, 1);
    run("plain 16 x 32 (library)", mix<16, 8>, 128, 64, 4);
    run("plain 32 x 16", mix<8, 16>, 64, 128, 2);
    run("plain 64 x  8 (unshared)", mix<4, 32>, 32, 256, 1);
    // XCD-local split-K clusters: loads + partial exchange
    u32x4* part; CK(hipMalloc(&part, (size_t)2 * 256 * 16 * 1024));
    unsigned* flags; CK(hipMalloc(&flags, 64 * 32 * 4));
    auto run_sk = [&](const char* name, auto kern, int wkb, int pk) {
        double best_launch = 1e9, best_round = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemsetAsync(flags, 0, 64 * 32 * 4, st));
            evict(rep);
            CK(hipEventRecord(e0, st));
            for (int r = 0; r < rounds; ++r) hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, st, warena, abuf, part, flags, 1, (size_t)r, (unsigned long long*)nullptr, sink);
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            best_launch = std::min(best_launch, (double)ms * 1e3 / rounds);
            CK(hipMemsetAsync(flags, 0, 64 * 32 * 4, st));
            evict(rep + 3);
            { std::vector<unsigned long long> init(512); for (int i = 0; i < 256; ++i) { init[2 * i] = ~0ull; init[2 * i + 1] = 0; } CK(hipMemcpyAsync(stamps, init.data(), 512 * 8, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st)); }
            hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, st, warena, abuf, part, flags, rounds, (size_t)0, stamps, sink);
            CK(hipStreamSynchronize(st));
            std::vector<unsigned long long> h(512);
            CK(hipMemcpy(h.data(), stamps, 512 * 8, hipMemcpyDeviceToHost));
            unsigned long long a = ~0ull, b = 0;
            for (int i = 0; i < 256; ++i) { a = std::min(a, h[2 * i]); b = std::max(b, h[2 * i + 1]); }
            best_round = std::min(best_round, (double)(b - a) * 0.01 / rounds);
        }
        printf("%-44s W %3d KiB (unshared) + A  64 KiB + %2d KiB partial out, %2d KiB in: %6.2f us per launch | %6.2f us per round in one launch\n", name, wkb, pk, 3 * pk / 4,
               best_launch, best_round);
    };
    run_sk("pair  split-K x4 cluster (64 x 2x32, K/4)", mix_splitk<8, 16>, 64, 16);
    run_sk("plain split-K x4 cluster (64 x 32, K/4)", mix_splitk<4, 8>, 32, 8);
    return 0;
}
