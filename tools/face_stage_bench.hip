// face_stage_bench.hip — the face-cluster stage kernel of the shallow levels (hifidiff_amd/csrc/hd_face.hpp) on its own, diagnostic
// build with in-kernel stamps (tools only): synthetic weights / activations, batch 64; per block the median over workgroups of
// LayerNorm1 (+ halo wait), conv1, depthwise + gate, pool exchange, sca .. conv5, exit stores + publish; residency per CU.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DHD_STAMPS -o tools/face_stage_bench_bin tools/face_stage_bench.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../hifidiff_amd/csrc/hd_face.hpp"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
using namespace hd;

__global__ void fill_bf16(unsigned short* p, size_t n, unsigned seed, float scale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = f32_to_bf16_bits(((int)(h & 0xffff) - 32768) * (scale / 32768.f));
    }
}
__global__ void fill_f32(float* p, size_t n, unsigned seed, float scale, float offset) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = offset + ((int)(h & 0xffff) - 32768) * (scale / 32768.f);
    }
}
template <class T> T* dmalloc(size_t n) { T* p; CK(hipMalloc(&p, n * sizeof(T))); return p; }

template <int C, int OWN>
void run(int nblocks, int B, int reps, int no_w = 0) {
    typedef FaceCfg<C, OWN> K;
    const int M = B * K::HW;
    std::vector<XBlockW> hb(nblocks);
    unsigned seed = 1;
    for (auto& b : hb) {
        auto w = [&](size_t n, float sc) { unsigned short* p = dmalloc<unsigned short>(n); fill_bf16<<<512, 256>>>(p, n, seed++, sc); return reinterpret_cast<const uint4*>(p); };
        auto f = [&](size_t n, float sc, float off) { float* p = dmalloc<float>(n); fill_f32<<<64, 256>>>(p, n, seed++, sc, off); return (const float*)p; };
        const float ws = 1.0f / sqrtf((float)C);
        b.w1 = w((size_t)2 * C * C, ws); b.wsca = w((size_t)C * C, ws); b.w3 = w((size_t)C * C, ws); b.w4 = w((size_t)2 * C * C, ws); b.w5 = w((size_t)C * C, ws);
        if (no_w) { b.w1 = nullptr; b.wsca = nullptr; b.w3 = nullptr; b.w4 = nullptr; b.w5 = nullptr; }     // face_load_b skips null pointers in the stamps build
        b.b1 = f(2 * C, 0.1f, 0.f); b.bsca = f(C, 0.1f, 1.f); b.b3 = f(C, 0.1f, 0.f); b.b4 = f(2 * C, 0.1f, 0.f); b.b5 = f(C, 0.1f, 0.f);
        b.beta = f(C, 0.2f, 0.f); b.gamma = f(C, 0.2f, 0.f); b.dw_w = f((size_t)9 * 2 * C, 0.3f, 0.f); b.dw_b = f(2 * C, 0.1f, 0.5f);
        b.film_off = (int)(&b - hb.data()) * 4 * C; b.pad_ = 0;
    }
    FStageP p{};
    p.B = B; p.nblocks = nblocks;
    XBlockW* db = dmalloc<XBlockW>(nblocks); CK(hipMemcpy(db, hb.data(), nblocks * sizeof(XBlockW), hipMemcpyHostToDevice)); p.blocks = db;
    p.X = dmalloc<float>((size_t)M * C); fill_f32<<<256, 256>>>(p.X, (size_t)M * C, 77, 1.f, 0.f);
    p.Xb = dmalloc<unsigned short>((size_t)M * C);
    p.pool_part = dmalloc<float>((size_t)64 * 8 * 256);
    float* film = dmalloc<float>((size_t)nblocks * 4 * C); fill_f32<<<64, 256>>>(film, (size_t)nblocks * 4 * C, 80, 0.2f, 1.f); p.film = film; p.ln_eps = 1e-6f;
    unsigned* sync = dmalloc<unsigned>(2 * 64 * 16); CK(hipMemset(sync, 0, 2 * 64 * 16 * 4));
    p.flags = sync; p.gstate = sync + 64 * 16;
    unsigned* tmo_h; CK(hipHostMalloc(reinterpret_cast<void**>(&tmo_h), 64, hipHostMallocMapped)); tmo_h[0] = 0;
    CK(hipHostGetDevicePointer(reinterpret_cast<void**>(&p.tmo), tmo_h, 0));
    { unsigned* ab; CK(hipMalloc(&ab, 256)); CK(hipMemset(ab, 0, 256)); p.abort_dev = ab; }      // every stage launch reads the abort word at entry
    const int grid = 64 * K::CL;
    p.stamps = dmalloc<unsigned long long>((size_t)nblocks * grid * 8); CK(hipMemset(p.stamps, 0, (size_t)nblocks * grid * 8 * 8));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0, st));
        CK((launch_face_stage<C, OWN>(p, st)));
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
        if (tmo_h[0]) { printf("TIMEOUT code 0x%x\n", tmo_h[0]); exit(2); }
    }
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, naf_face_stage_kernel<C, OWN>, K::THREADS, K::SMEM));
    std::vector<unsigned long long> h((size_t)nblocks * grid * 8);
    CK(hipMemcpy(h.data(), p.stamps, h.size() * 8, hipMemcpyDeviceToHost));
    if (no_w) printf("WHAT-IF no weight loads (timing only): ");
    printf("C=%d rows/wg=%d blocks=%d B=%d: %d workgroups of %d threads, %d B dynamic LDS, occupancy API says %d per CU; kernel %.1f us = %.2f us per block\n", C, OWN, nblocks, B, grid, K::THREADS,
           K::SMEM, occ, best * 1e3, best * 1e3 / nblocks);
    auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    const char* nm[6] = {"wait+LN1", "conv1", "dw+gate", "pool xchg", "sca..conv5", "exit+publish"};
    unsigned long long first = ~0ull, lastt = 0, latest_start = 0;
    for (int b = 0; b < grid; ++b) { first = std::min(first, h[(size_t)b * 8]); latest_start = std::max(latest_start, h[(size_t)b * 8]); lastt = std::max(lastt, h[((size_t)(nblocks - 1) * grid + b) * 8 + 6]); }
    printf("  first workgroup starts -> last ends %.2f us; latest workgroup start %.2f us after the first\n", (lastt - first) * 0.01, (latest_start - first) * 0.01);
    for (int blk = 0; blk < nblocks; ++blk) {
        printf("  block %d:", blk);
        for (int k = 0; k < 6; ++k) {
            std::vector<double> d;
            for (int b = 0; b < grid; ++b) { const unsigned long long* s = &h[((size_t)blk * grid + b) * 8]; if (s[k + 1] >= s[k]) d.push_back((s[k + 1] - s[k]) * 0.01); }
            printf(" %s %.2f", nm[k], d.empty() ? -1.0 : med(d));
        }
        printf("\n");
    }
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 20;
    run<128, 32>(2, 64, reps); run<128, 32>(2, 64, reps, 1);
    run<256, 16>(2, 64, reps); run<256, 16>(2, 64, reps, 1);
    run<256, 32>(2, 64, reps);
    return 0;
}
