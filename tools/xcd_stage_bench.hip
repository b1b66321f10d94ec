// xcd_stage_bench.hip — the XCD-local persistent stage kernel (hifidiff_amd/csrc/hd_xcd.hpp) on its own, diagnostic build
// with in-kernel stamps (tools only): synthetic weights and activations of the level's shapes, batch 64, a few warm
// launches, then per phase the median over workgroups of: barrier wait, K loop (A loads + LayerNorm + MFMA), epilogue,
// store drain, publish; and the span of the phase over the whole chip.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DHD_STAMPS -o tools/xcd_stage_bench_bin tools/xcd_stage_bench.hip
//        (-DXS_EXPERIMENT_PASSES [-DXS_PASS_ROWS=64]: `xcd_stage_bench_bin 20 32` times the archived multi-pass form at latent 32, level 3)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#ifdef XS_EXPERIMENT_PASSES          // the multi-pass generalisation tried for latent 32, level 3 (tools/experiments/README.md)
#include "experiments/hd_xcd_passes.hpp"
#else
#include "../hifidiff_amd/csrc/hd_xcd.hpp"
#endif

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
using namespace hd;

__global__ void fill_bf16(unsigned short* p, size_t n, unsigned seed, float scale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        const float v = ((int)(h & 0xffff) - 32768) * (scale / 32768.f);
        p[i] = f32_to_bf16_bits(v);
    }
}
__global__ void fill_f32(float* p, size_t n, unsigned seed, float scale, float offset) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = offset + ((int)(h & 0xffff) - 32768) * (scale / 32768.f);
    }
}
template <class T> T* dmalloc(size_t n) { T* p; CK(hipMalloc(&p, n * sizeof(T))); return p; }

template <int C, int HW>
void run(int nblocks, int B, int reps, int force_global, int no_a = 0, int no_w = 0, int alias_w = 0) {
    const int M = B * HW, NT = C / 32;
    std::vector<XBlockW> hb(nblocks);
    unsigned seed = 1;
    for (auto& b : hb) {
        auto w = [&](size_t n, float sc) { unsigned short* p = dmalloc<unsigned short>(n); fill_bf16<<<512, 256>>>(p, n, seed++, sc); return reinterpret_cast<const uint4*>(p); };
        auto f = [&](size_t n, float sc, float off) { float* p = dmalloc<float>(n); fill_f32<<<64, 256>>>(p, n, seed++, sc, off); return (const float*)p; };
        const float ws = 1.0f / sqrtf((float)C);
        b.w1 = w((size_t)2 * C * C, ws); b.wsca = w((size_t)C * C, ws); b.w3 = w((size_t)C * C, ws); b.w4 = w((size_t)2 * C * C, ws); b.w5 = w((size_t)C * C, ws);
        b.b1 = f(2 * C, 0.1f, 0.f); b.bsca = f(C, 0.1f, 1.f); b.b3 = f(C, 0.1f, 0.f); b.b4 = f(2 * C, 0.1f, 0.f); b.b5 = f(C, 0.1f, 0.f);
        b.beta = f(C, 0.2f, 0.f); b.gamma = f(C, 0.2f, 0.f); b.dw_w = f((size_t)9 * 2 * C, 0.3f, 0.f); b.dw_b = f(2 * C, 0.1f, 0.5f);
        b.film_off = (int)(&b - hb.data()) * 4 * C; b.pad_ = 0;
    }
    if (alias_w)                       // what-if: every block and phase streams the SAME 2 C^2 weights (4 MB at C = 1024: about one XCD L2), values irrelevant
        for (auto& b : hb) { b.w1 = hb[0].w1; b.w4 = hb[0].w1; b.wsca = hb[0].w1; b.w3 = hb[0].w1; b.w5 = hb[0].w1; }
    XStageP p{};
    p.B = B; p.nblocks = nblocks;
    XBlockW* db = dmalloc<XBlockW>(nblocks); CK(hipMemcpy(db, hb.data(), nblocks * sizeof(XBlockW), hipMemcpyHostToDevice)); p.blocks = db;
    p.X = dmalloc<float>((size_t)M * C); fill_f32<<<256, 256>>>(p.X, (size_t)M * C, 77, 1.f, 0.f);
    p.Xb = dmalloc<unsigned short>((size_t)M * C); fill_bf16<<<256, 256>>>(p.Xb, (size_t)M * C, 78, 1.f);
    p.sx = dmalloc<float2>((size_t)M * NT); fill_f32<<<64, 256>>>((float*)p.sx, (size_t)M * NT * 2, 79, 0.1f, 0.5f);
    p.G = dmalloc<unsigned short>((size_t)M * C); p.Yb = dmalloc<unsigned short>((size_t)M * C); p.sy = dmalloc<float2>((size_t)M * NT);
    p.pooled16 = dmalloc<unsigned short>((size_t)B * C); p.pooled = nullptr; p.S = nullptr;
    float* film = dmalloc<float>((size_t)nblocks * 4 * C); fill_f32<<<64, 256>>>(film, (size_t)nblocks * 4 * C, 80, 0.2f, 1.f); p.film = film; p.ln_eps = 1e-6f;
    unsigned* sync = dmalloc<unsigned>(3 * 256); CK(hipMemset(sync, 0, 3 * 256 * 4));
    p.flags = sync; p.hello = sync + 256; p.gstate = sync + 512;
    unsigned* tmo_h; CK(hipHostMalloc(reinterpret_cast<void**>(&tmo_h), 64, hipHostMallocMapped)); tmo_h[0] = 0;
    CK(hipHostGetDevicePointer(reinterpret_cast<void**>(&p.tmo), tmo_h, 0));
    { unsigned* ab; CK(hipMalloc(&ab, 256)); CK(hipMemset(ab, 0, 256)); p.abort_dev = ab; }      // every stage launch reads the abort word at entry
    p.force_global = force_global; p.dbg_no_a = no_a; p.dbg_no_w = no_w;
    const int P = 5 * nblocks;
    p.stamps = dmalloc<unsigned long long>((size_t)P * 256 * 8); CK(hipMemset(p.stamps, 0, (size_t)P * 256 * 8 * 8));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0, st));
        CK((launch_xcd_stage<C, HW>(p, st)));
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
        if (tmo_h[0]) { printf("TIMEOUT code 0x%x\n", tmo_h[0]); exit(2); }
    }
    std::vector<unsigned long long> h((size_t)P * 256 * 8);
    CK(hipMemcpy(h.data(), p.stamps, h.size() * 8, hipMemcpyDeviceToHost));
    if (no_a || no_w || alias_w) printf("WHAT-IF%s%s%s (timing only): ", no_a ? " no activation loads" : "", no_w ? " no weight loads" : "", alias_w ? " all phases read the same 2 C^2 weights (L2 / MALL resident)" : "");
    printf("C=%d HW=%d blocks=%d B=%d %s: kernel %.1f us = %.2f us per block, %.2f us per phase\n", C, HW, nblocks, B, force_global ? "global hand-off" : "local hand-off",
           best * 1e3, best * 1e3 / nblocks, best * 1e3 / P);
    auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    const char* qn[5] = {"q0 conv1+dw", "q1 sca", "q2 conv3", "q3 conv4", "q4 conv5"};
    double acc[5][6] = {};
    int cnt[5] = {};
    for (int ph = 0; ph < P; ++ph) {
        std::vector<double> seg[5];
        unsigned long long lo = ~0ull, hi = 0;
        for (int b = 0; b < 256; ++b) {
            const unsigned long long* s = &h[((size_t)ph * 256 + b) * 8];
            if (!s[0]) continue;
            for (int k = 0; k < 5; ++k) seg[k].push_back((double)(s[k + 1] >= s[k] ? s[k + 1] - s[k] : 0) * 0.01);
            lo = std::min(lo, s[0]); hi = std::max(hi, s[5]);
        }
        if (seg[0].empty()) continue;
        const int q = ph % 5;
        if (ph >= 5 && ph < P - 5) { for (int k = 0; k < 5; ++k) acc[q][k] += med(seg[k]); acc[q][5] += (hi - lo) * 0.01; cnt[q]++; }
        if (ph < 10) printf("  phase %2d %-12s wait %5.2f  kloop %5.2f  epilogue %5.2f  drain %5.2f  publish %5.2f | span %5.2f us\n", ph, qn[q], med(seg[0]), med(seg[1]), med(seg[2]),
                            med(seg[3]), med(seg[4]), (hi - lo) * 0.01);
    }
    for (int q = 0; q < 5; ++q)
        if (cnt[q]) printf("  mean over inner blocks %-12s wait %5.2f  kloop %5.2f  epilogue %5.2f  drain %5.2f  publish %5.2f | span %5.2f us\n", qn[q], acc[q][0] / cnt[q], acc[q][1] / cnt[q],
                           acc[q][2] / cnt[q], acc[q][3] / cnt[q], acc[q][4] / cnt[q], acc[q][5] / cnt[q]);
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 20;
#ifdef XS_EXPERIMENT_PASSES
    if (argc > 2 && atoi(argv[2]) == 32) {          // latent 32, level 3: 128 rows per workgroup in passes of XS_PASS_ROWS (32 or 64) rows
        run<1024, 16>(8, 64, reps, 0);
        run<1024, 16>(2, 64, reps, 0);
        run<1024, 16>(8, 64, reps, 0, 1, 0);
        run<1024, 16>(8, 64, reps, 0, 0, 1);
        return 0;
    }
#endif
    run<1024, 4>(8, 64, reps, 0);
    if (argc > 2 && atoi(argv[2]) == 9) { run<1024, 4>(8, 64, reps, 0, 0, 0, 1); run<1024, 4>(8, 64, reps, 0, 0, 1); return 0; }
    run<512, 16>(4, 64, reps, 0);
    run<1024, 4>(8, 64, reps, 1);
    run<1024, 4>(8, 64, reps, 0, 1, 0);
    run<1024, 4>(8, 64, reps, 0, 0, 1);
    run<1024, 4>(8, 64, reps, 0, 1, 1);
    run<512, 16>(4, 64, reps, 0, 1, 0);
    run<512, 16>(4, 64, reps, 0, 0, 1);
    run<512, 16>(4, 64, reps, 0, 1, 1);
    return 0;
}
