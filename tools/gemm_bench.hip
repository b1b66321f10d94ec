// gemm_bench.hip — microbenchmark of the GEMM kernel variants on the refiner's shapes (tools only).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o gpurun_out/gemm_bench tools/gemm_bench.hip
// Each variant is launched back to back on one stream over a rotation of weight buffers larger than the
// 256 MiB Infinity Cache, so weights come from HBM as in the real step.  XCD=1 uses the library's block -> tile map (row groups of a
// weight tile on one XCD), which is what the denoiser GEMMs run with; CHAIN=1 M C C benchmarks the chain kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define HD_STAMPS 1
#include "../hifidiff_amd/csrc/hd_gemm.hpp"
#include "../hifidiff_amd/csrc/hd_chain.hpp"
#include <algorithm>
using namespace hd;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void fill_kernel(unsigned* p, size_t n, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        // two bf16 in [-1,1): exponent bits around 0x3f00
        unsigned lo = 0x3c00u + (x & 0x3ffu), hi = 0x3c00u + ((x >> 10) & 0x3ffu);
        p[i] = (lo | ((x >> 20) & 1u) << 15) | ((hi | ((x >> 21) & 1u) << 15) << 16);
    }
}
__global__ void fillf_kernel(float* p, size_t n, float v) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v * (float)((i * 7919) % 1000) / 1000.f;
}
// floor: stream the weight bytes only (16 B per lane, everything in flight)
__global__ __launch_bounds__(512) void stream_kernel(const uint4* w, size_t n16, unsigned* sink) {
    unsigned acc = 0;
    const size_t per = (n16 + gridDim.x - 1) / gridDim.x;
    const size_t b = (size_t)blockIdx.x * per, e = b + per < n16 ? b + per : n16;
    for (size_t i = b + threadIdx.x; i < e; i += 256 * 4) {
        uint4 v0 = w[i], v1 = (i + 256 < e) ? w[i + 256] : make_uint4(0,0,0,0), v2 = (i + 512 < e) ? w[i + 512] : make_uint4(0,0,0,0), v3 = (i + 768 < e) ? w[i + 768] : make_uint4(0,0,0,0);
        acc ^= v0.x ^ v1.y ^ v2.z ^ v3.w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

struct Bufs { std::vector<uint4*> W; void* A; float* out; unsigned short* outb; float *bias, *rscale, *resid, *film, *rowscale; float2 *stats_in, *stats_out; };

template <class F>
float time_it(F f, int iters, hipStream_t s) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) f(i);
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(a, s));
    for (int i = 0; i < iters; ++i) f(i);
    CK(hipEventRecord(b, s));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms * 1000.f / iters;
}

static void report_stamps(unsigned long long* dev, int nwg) {
    std::vector<unsigned long long> h((size_t)nwg * 8);
    CK(hipMemcpy(h.data(), dev, h.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull, t5 = 0;
    std::vector<double> d[5];
    for (int w = 0; w < nwg; ++w) {
        const unsigned long long* t = &h[(size_t)w * 8];
        if (t[0] < t0) t0 = t[0];
        if (t[5] > t5) t5 = t[5];
        for (int i = 0; i < 5; ++i) d[i].push_back((double)(t[i + 1] - t[i]) * 0.01);
    }
    double first_last_start = 0;
    for (int w = 0; w < nwg; ++w) first_last_start = std::max(first_last_start, (double)(h[(size_t)w * 8] - t0) * 0.01);
    printf("  | span %.2f us, start skew %.2f;", (double)(t5 - t0) * 0.01, first_last_start);
    const char* nm[5] = {"issue", "1st-chunk", "rest-K", "sync", "epi"};
    for (int i = 0; i < 5; ++i) { std::sort(d[i].begin(), d[i].end()); printf(" %s %.2f/%.2f", nm[i], d[i][d[i].size() / 2], d[i].back()); }
    { std::vector<double> q; for (int w = 0; w < nwg; ++w) if (h[(size_t)w * 8 + 6]) q.push_back((double)(h[(size_t)w * 8 + 6] - h[(size_t)w * 8]) * 0.01);
      if (!q.empty()) { std::sort(q.begin(), q.end()); printf(" | loads-issued@ %.2f/%.2f", q[q.size() / 2], q.back()); } }
    { std::vector<double> q; for (int w = 0; w < nwg; ++w) if (h[(size_t)w * 8 + 7]) q.push_back((double)(h[(size_t)w * 8 + 7] - h[(size_t)w * 8]) * 0.01);
      if (!q.empty()) { std::sort(q.begin(), q.end()); printf(" stats-merged@ %.2f/%.2f", q[q.size() / 2], q.back()); } }
    printf("\n");
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 64, K = argc > 2 ? atoi(argv[2]) : 2048, N = argc > 3 ? atoi(argv[3]) : 2048;
    const int iters = 200;
    const int LDA = K + (getenv("LDA_PAD") ? atoi(getenv("LDA_PAD")) : 0);     // leading dimension of A in elements
    hipStream_t s; CK(hipStreamCreate(&s));
    const size_t wbytes = (size_t)N * K * 2;
    const int nrot = (int)((700ull << 20) / wbytes) + 1;
    Bufs b;
    for (int i = 0; i < nrot; ++i) { uint4* p; CK(hipMalloc(&p, wbytes)); hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, s, (unsigned*)p, wbytes / 4, 17u * i + 1); b.W.push_back(p); }
    CK(hipMalloc(&b.A, (size_t)M * K * 4)); hipLaunchKernelGGL(fillf_kernel, dim3(256), dim3(256), 0, s, (float*)b.A, (size_t)M * K, 1.0f);
    void* Ab; CK(hipMalloc(&Ab, (size_t)M * K * 2)); hipLaunchKernelGGL(fill_kernel, dim3(256), dim3(256), 0, s, (unsigned*)Ab, (size_t)M * K / 2, 99u);
    CK(hipMalloc(&b.out, (size_t)M * N * 4)); CK(hipMalloc(&b.outb, (size_t)M * N * 2));
    CK(hipMalloc(&b.bias, N * 4)); CK(hipMalloc(&b.rscale, N * 4)); CK(hipMalloc(&b.resid, (size_t)M * N * 4));
    CK(hipMalloc(&b.film, 2 * K * 4)); CK(hipMalloc(&b.rowscale, (size_t)M * K * 4)); CK(hipMalloc(&b.stats_in, (size_t)M * (K / 32) * 8)); CK(hipMalloc(&b.stats_out, (size_t)M * (N / 32) * 8));
    hipLaunchKernelGGL(fillf_kernel, dim3(64), dim3(256), 0, s, b.bias, (size_t)N, 0.1f);
    hipLaunchKernelGGL(fillf_kernel, dim3(64), dim3(256), 0, s, b.rscale, (size_t)N, 0.2f);
    hipLaunchKernelGGL(fillf_kernel, dim3(64), dim3(256), 0, s, b.resid, (size_t)M * N, 1.0f);
    hipLaunchKernelGGL(fillf_kernel, dim3(64), dim3(256), 0, s, b.film, (size_t)2 * K, 1.0f);
    hipLaunchKernelGGL(fillf_kernel, dim3(64), dim3(256), 0, s, b.rowscale, (size_t)M * K, 1.0f);
    const int NP = getenv("NP1") ? 1 : K / 32;          // LayerNorm partials per row, as the producers emit them
    std::vector<float2> st((size_t)M * NP, make_float2(0.5f, (float)(K / NP) * 0.08f));
    CK(hipMemcpy(b.stats_in, st.data(), st.size() * 8, hipMemcpyHostToDevice));
    unsigned* sink; CK(hipMalloc(&sink, 64));
    unsigned long long* stamps; CK(hipMalloc(&stamps, 8192 * 64));
    CK(hipStreamSynchronize(s));

    auto base = [&](int i) {
        GemmP p{};
        p.M = M; p.N = N; p.K = K; p.Kp = K; p.nt_total = N / 32; p.W = b.W[getenv("NOROT") ? 0 : i % nrot];
        p.a_scale = 1.f; p.hw = 1; p.ln_eps = 1e-6f; p.shuffle_r = 1; p.stats_np = NP; p.stats_cnt = K / NP;
        p.bias = b.bias; p.rscale = b.rscale; p.resid = b.resid; p.ldr = N; p.out = b.out; p.ldo = N;
        p.xcd_tile_affine = getenv("XCD") ? 1 : 0;      // XCD=1: the library's block -> tile map (row groups of a weight tile on one XCD)
        return p;
    };
    printf("M=%d K=%d N=%d  weights %.1f MB, rotation %d buffers\n", M, K, N, wbytes / 1e6, nrot);
    {
        float us = time_it([&](int i) { hipLaunchKernelGGL(stream_kernel, dim3(256), dim3(256), 0, s, b.W[i % nrot], wbytes / 16, sink); }, iters, s);
        printf("%-44s %8.2f us  %7.1f GB/s\n", "stream floor (256 WG x 256 thr, 4 loads)", us, wbytes / us / 1e3);
        us = time_it([&](int i) { hipLaunchKernelGGL(stream_kernel, dim3(1024), dim3(256), 0, s, b.W[i % nrot], wbytes / 16, sink); }, iters, s);
        printf("%-44s %8.2f us  %7.1f GB/s\n", "stream floor (1024 WG)", us, wbytes / us / 1e3);
        us = time_it([&](int i) { hipLaunchKernelGGL(stream_kernel, dim3(128), dim3(256), 0, s, b.W[i % nrot], wbytes / 16, sink); }, iters, s);
        printf("%-44s %8.2f us  %7.1f GB/s\n", "stream floor (128 WG)", us, wbytes / us / 1e3);
        us = time_it([&](int i) { hipLaunchKernelGGL(stream_kernel, dim3(1), dim3(64), 0, s, b.W[0], (size_t)64, sink); }, iters, s);
        printf("%-44s %8.2f us\n", "empty-ish kernel", us);
    }
#define RUN_SK(name, MT, WAVES, PAIR, G, LD, EP, setup)                                                  \
    {                                                                                                    \
        float us = time_it([&](int i) { GemmP p = base(i); setup; hipError_t e = launch_skinny<SkinnyCfg<1, WAVES, MT, PAIR, G>, LD, EP>(p, s); if (e != hipSuccess) { printf("launch failed %s\n", hipGetErrorString(e)); exit(1);} }, iters, s); \
        printf("%-44s %8.2f us  %7.1f GB/s (weights)", name, us, wbytes / us / 1e3);                    \
        { GemmP p = base(7); setup; p.stamps = stamps; CK(hipMemsetAsync(stamps, 0, 8192 * 64, s));     \
          launch_skinny<SkinnyCfg<1, WAVES, MT, PAIR, G>, LD, EP>(p, s); CK(hipStreamSynchronize(s));      \
          const int nwg = ((M + 32 * MT - 1) / (32 * MT)) * (((PAIR) ? N / 2 : N) / 32);                 \
          report_stamps(stamps, nwg); }                                                                  \
    }
    if (K % 512 == 0 && M <= 1024) {
        // ---- do two streams of graph-replayed launches overlap? (N = 2048 uses 128 workgroups per launch)
        hipStream_t s2; CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
        hipStream_t s1; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
        auto capture = [&](hipStream_t st, int rot0) {
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            for (int i = 0; i < 50; ++i) { GemmP p = base(rot0 + i); p.A = Ab; p.lda = K; (void)launch_skinny<SkinnyCfg<1, 8, 1, false, 2>, LdBF16Plain, EpResidF32>(p, st); }
            CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0)); CK(hipGraphDestroy(g));
            return ge;
        };
        hipGraphExec_t ga = capture(s1, 0), gb = capture(s2, 40);
        hipEvent_t e0, e1, e2; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
        CK(hipGraphLaunch(ga, s1)); CK(hipGraphLaunch(gb, s2)); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, s1)); for (int r = 0; r < 4; ++r) CK(hipGraphLaunch(ga, s1)); CK(hipEventRecord(e1, s1)); CK(hipDeviceSynchronize());
        float one; CK(hipEventElapsedTime(&one, e0, e1));
        CK(hipEventRecord(e0, s1)); CK(hipStreamWaitEvent(s2, e0, 0));
        for (int r = 0; r < 4; ++r) { CK(hipGraphLaunch(ga, s1)); CK(hipGraphLaunch(gb, s2)); }
        CK(hipEventRecord(e2, s2)); CK(hipStreamWaitEvent(s1, e2, 0)); CK(hipEventRecord(e1, s1)); CK(hipDeviceSynchronize());
        float two; CK(hipEventElapsedTime(&two, e0, e1));
        {   // pure kernel-boundary cost: chains of trivial kernels of different shapes in a graph
            auto chain = [&](int wgs, int thr, int lds) {
                hipGraph_t g; hipGraphExec_t ge;
                CK(hipStreamBeginCapture(s1, hipStreamCaptureModeThreadLocal));
                for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(stream_kernel, dim3(wgs), dim3(thr), lds, s1, b.W[0], (size_t)64, sink);
                CK(hipStreamEndCapture(s1, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0)); CK(hipGraphDestroy(g));
                CK(hipGraphLaunch(ge, s1)); CK(hipStreamSynchronize(s1));
                CK(hipEventRecord(e0, s1)); for (int r = 0; r < 4; ++r) CK(hipGraphLaunch(ge, s1)); CK(hipEventRecord(e1, s1)); CK(hipStreamSynchronize(s1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                printf("trivial-kernel chain in a graph: %4d WG x %4d thr, %6d B LDS: %.2f us per launch\n", wgs, thr, lds, ms * 1000 / 400);
            };
            chain(1, 64, 0); chain(256, 256, 0); chain(128, 512, 0); chain(128, 512, 60000); chain(1024, 256, 0);
        }
        printf("graph replay, 200 launches on one stream: %.2f us/launch; 2 x 200 on two streams: %.2f us per pair (%.2fx throughput)\n",
               one * 1000 / 200, two * 1000 / 200, 2 * one / two);

        RUN_SK("skinny bf16plain resid MT1 W8 D4", 1, 8, false, 4, LdBF16Plain, EpResidF32, (p.A = Ab, p.lda = LDA))
        RUN_SK("skinny bf16plain resid MT1 W8 D2", 1, 8, false, 2, LdBF16Plain, EpResidF32, (p.A = Ab, p.lda = LDA))
        RUN_SK("skinny bf16plain resid MT1 W4 D4", 1, 4, false, 4, LdBF16Plain, EpResidF32, (p.A = Ab, p.lda = LDA))
        RUN_SK("skinny bf16plain resid MT2 W8 D4", 2, 8, false, 4, LdBF16Plain, EpResidF32, (p.A = Ab, p.lda = LDA))
        RUN_SK("skinny bf16plain resid MT2 W8 D2", 2, 8, false, 2, LdBF16Plain, EpResidF32, (p.A = Ab, p.lda = LDA))
        RUN_SK("skinny f32plain bias  MT1 W8 D2", 1, 8, false, 2, LdF32Plain, EpBiasF32, (p.A = b.A, p.lda = K))
        RUN_SK("skinny f32plain bias  MT1 W8 D4", 1, 8, false, 4, LdF32Plain, EpBiasF32, (p.A = b.A, p.lda = K))
        RUN_SK("skinny bf16scale resid MT1 W8 D2", 1, 8, false, 2, LdBF16Scale, EpResidF32, (p.A = Ab, p.lda = LDA, p.rowscale = b.rowscale))
        RUN_SK("skinny LN bias MT1 W8 D2", 1, 8, false, 2, LdF32LN, EpBiasF32, (p.A = Ab, p.lda = LDA, p.film = b.film, p.film_gain_off = 0, p.film_bias_off = K, p.stats_in = b.stats_in))
        RUN_SK("skinny LN bias MT1 W8 D4", 1, 8, false, 4, LdF32LN, EpBiasF32, (p.A = Ab, p.lda = LDA, p.film = b.film, p.film_gain_off = 0, p.film_bias_off = K, p.stats_in = b.stats_in))
        RUN_SK("skinny LN gate(pair) MT1 W8 D2", 1, 8, true, 2, LdF32LN, EpGateBF16, (p.A = Ab, p.lda = LDA, p.film = b.film, p.film_gain_off = 0, p.film_bias_off = K, p.stats_in = b.stats_in, p.out = b.outb, p.ldo = N / 2))
        RUN_SK("skinny LN gate(pair) MT2 W8 D1", 2, 8, true, 1, LdF32LN, EpGateBF16, (p.A = Ab, p.lda = LDA, p.film = b.film, p.film_gain_off = 0, p.film_bias_off = K, p.stats_in = b.stats_in, p.out = b.outb, p.ldo = N / 2))
        RUN_SK("skinny LN gate(pair) MT2 W8 D2", 2, 8, true, 2, LdF32LN, EpGateBF16, (p.A = Ab, p.lda = LDA, p.film = b.film, p.film_gain_off = 0, p.film_bias_off = K, p.stats_in = b.stats_in, p.out = b.outb, p.ldo = N / 2))
        RUN_SK("skinny bf16plain gate(pair) MT1 W8 D2 (no LN)", 1, 8, true, 2, LdBF16Plain, EpGateBF16, (p.A = Ab, p.lda = LDA, p.out = b.outb, p.ldo = N / 2))
        RUN_SK("skinny bf16plain bias MT1 W8 D2 (no LN)", 1, 8, false, 2, LdBF16Plain, EpBiasF32, (p.A = Ab, p.lda = LDA))
        // four waves per workgroup (one per SIMD: up to 512 VGPRs each): the whole K slice of a wave in flight
        RUN_SK("skinny LN gate(pair) MT2 W4 D2", 2, 4, true, 2, LdF32LN, EpGateBF16, (p.A = Ab, p.lda = LDA, p.film = b.film, p.film_gain_off = 0, p.film_bias_off = K, p.stats_in = b.stats_in, p.out = b.outb, p.ldo = N / 2))
        RUN_SK("skinny LN gate(pair) MT2 W4 D3", 2, 4, true, 3, LdF32LN, EpGateBF16, (p.A = Ab, p.lda = LDA, p.film = b.film, p.film_gain_off = 0, p.film_bias_off = K, p.stats_in = b.stats_in, p.out = b.outb, p.ldo = N / 2))
        RUN_SK("skinny LN gate(pair) MT2 W4 D4", 2, 4, true, 4, LdF32LN, EpGateBF16, (p.A = Ab, p.lda = LDA, p.film = b.film, p.film_gain_off = 0, p.film_bias_off = K, p.stats_in = b.stats_in, p.out = b.outb, p.ldo = N / 2))
        RUN_SK("skinny LN gate(pair) MT1 W4 D4", 1, 4, true, 4, LdF32LN, EpGateBF16, (p.A = Ab, p.lda = LDA, p.film = b.film, p.film_gain_off = 0, p.film_bias_off = K, p.stats_in = b.stats_in, p.out = b.outb, p.ldo = N / 2))
        RUN_SK("skinny LN gate(pair) MT1 W8 D3", 1, 8, true, 3, LdF32LN, EpGateBF16, (p.A = Ab, p.lda = LDA, p.film = b.film, p.film_gain_off = 0, p.film_bias_off = K, p.stats_in = b.stats_in, p.out = b.outb, p.ldo = N / 2))
        RUN_SK("skinny bf16plain resid MT2 W4 D4", 2, 4, false, 4, LdBF16Plain, EpResidF32, (p.A = Ab, p.lda = LDA))
        RUN_SK("skinny bf16plain resid MT2 W4 D3", 2, 4, false, 3, LdBF16Plain, EpResidF32, (p.A = Ab, p.lda = LDA))
    }
#define RUN_TALL(name, CFG, LD, EP, setup)                                                               \
    {                                                                                                    \
        float us = time_it([&](int i) { GemmP p = base(i); setup; hipError_t e = launch_gemm<CFG, LD, EP>(p, s); if (e != hipSuccess) { printf("launch failed %s\n", hipGetErrorString(e)); exit(1);} }, iters, s); \
        printf("%-44s %8.2f us", name, us);                                                             \
        { GemmP p = base(7); setup; p.stamps = stamps; CK(hipMemsetAsync(stamps, 0, 8192 * 64, s));     \
          (void)launch_gemm<CFG, LD, EP>(p, s); CK(hipStreamSynchronize(s));                             \
          const int nc = CFG::PAIR ? N / 2 : N;                                                          \
          const int nwg = ((M + CFG::BM - 1) / CFG::BM) * ((nc + CFG::NCOLS - 1) / CFG::NCOLS);          \
          report_stamps(stamps, nwg < 8192 ? nwg : 8192); }                                              \
    }
#define RUN_SKH2(name, PAIR, D, LD, EP, setup)                                                                  \
    {                                                                                                    \
        float us = time_it([&](int i) { GemmP p = base(i); setup; hipError_t e = launch_skinny<SkinnyCfg<1, 8, 1, PAIR, D, true>, LD, EP>(p, s); if (e != hipSuccess) { printf("launch failed %s\n", hipGetErrorString(e)); exit(1);} }, iters, s); \
        printf("%-44s %8.2f us", name, us);                                                             \
        { GemmP p = base(7); setup; p.stamps = stamps; CK(hipMemsetAsync(stamps, 0, 8192 * 64, s));     \
          (void)launch_skinny<SkinnyCfg<1, 8, 1, PAIR, D, true>, LD, EP>(p, s); CK(hipStreamSynchronize(s)); \
          const int nwg = ((M + 15) / 16) * (((PAIR) ? N / 2 : N) / 32);                                               \
          report_stamps(stamps, nwg < 8192 ? nwg : 8192); }                                              \
    }
#define RUN_SKH(name, D, LD, EP, setup) RUN_SKH2(name, true, D, LD, EP, setup)
#define RUN_SKW(name, WM, WK, MT, PAIR, D, LD, EP, setup)                                                 \
    {                                                                                                    \
        float us = time_it([&](int i) { GemmP p = base(i); setup; hipError_t e = launch_skinny<SkinnyCfg<WM, WK, MT, PAIR, D>, LD, EP>(p, s); if (e != hipSuccess) { printf("launch failed %s\n", hipGetErrorString(e)); exit(1);} }, iters, s); \
        printf("%-44s %8.2f us", name, us);                                                             \
        { GemmP p = base(7); setup; p.stamps = stamps; CK(hipMemsetAsync(stamps, 0, 8192 * 64, s));     \
          (void)launch_skinny<SkinnyCfg<WM, WK, MT, PAIR, D>, LD, EP>(p, s); CK(hipStreamSynchronize(s)); \
          const int nwg = ((M + 32 * MT * WM - 1) / (32 * MT * WM)) * (((PAIR) ? N / 2 : N) / 32);       \
          report_stamps(stamps, nwg < 8192 ? nwg : 8192); }                                              \
    }
    if (M <= 1024 && K % 512 == 0) {        // deep-level fused conv1 (LN -> conv1 -> depthwise -> gate -> pool), faces of 2x2 / 4x4
        float* pooled; CK(hipMalloc(&pooled, (size_t)M * N * 4));
        float* dww; CK(hipMalloc(&dww, (size_t)N * 9 * 4)); hipLaunchKernelGGL(fillf_kernel, dim3(64), dim3(256), 0, s, dww, (size_t)N * 9, 0.3f);
        RUN_SKW("skinny W8 LN dwgate hw4 (L3 conv1 fused)", 1, 8, 1, true, 2, LdF32LN, EpDwGate, (p.A = Ab, p.lda = LDA, p.film = b.film, p.film_gain_off = 0, p.film_bias_off = K, p.stats_in = b.stats_in, p.out = b.outb, p.ldo = N / 2, p.dw_w = dww, p.dw_b = b.bias, p.pooled = pooled, p.hw = 4, p.side = 2))
        RUN_SKW("skinny W8 LN dwgate hw16 (L2 conv1 fused)", 1, 8, 1, true, 2, LdF32LN, EpDwGate, (p.A = Ab, p.lda = LDA, p.film = b.film, p.film_gain_off = 0, p.film_bias_off = K, p.stats_in = b.stats_in, p.out = b.outb, p.ldo = N / 2, p.dw_w = dww, p.dw_b = b.bias, p.pooled = pooled, p.hw = 16, p.side = 4))
        RUN_SKW("skinny W8 LN gate (no dw)", 1, 8, 1, true, 2, LdF32LN, EpGateBF16, (p.A = Ab, p.lda = LDA, p.film = b.film, p.film_gain_off = 0, p.film_bias_off = K, p.stats_in = b.stats_in, p.out = b.outb, p.ldo = N / 2))
        RUN_SKH("skinny W8 LN gate 16-row tiles D2", 2, LdF32LN, EpGateBF16, (p.A = Ab, p.lda = LDA, p.film = b.film, p.film_gain_off = 0, p.film_bias_off = K, p.stats_in = b.stats_in, p.out = b.outb, p.ldo = N / 2))
        RUN_SKH("skinny W8 LN gate 16-row tiles D3", 3, LdF32LN, EpGateBF16, (p.A = Ab, p.lda = LDA, p.film = b.film, p.film_gain_off = 0, p.film_bias_off = K, p.stats_in = b.stats_in, p.out = b.outb, p.ldo = N / 2))
        RUN_SKH("skinny W8 LN gate 16-row tiles D4", 4, LdF32LN, EpGateBF16, (p.A = Ab, p.lda = LDA, p.film = b.film, p.film_gain_off = 0, p.film_bias_off = K, p.stats_in = b.stats_in, p.out = b.outb, p.ldo = N / 2))
        RUN_SKH2("skinny W8 bf16plain resid 16-row tiles D2", false, 2, LdBF16Plain, EpResidF32, (p.A = Ab, p.lda = LDA, p.stats_out = b.stats_out))
        RUN_SKH2("skinny W8 bf16plain resid 16-row tiles D3", false, 3, LdBF16Plain, EpResidF32, (p.A = Ab, p.lda = LDA, p.stats_out = b.stats_out))
        RUN_SKH2("skinny W8 bf16plain resid 16-row tiles D4", false, 4, LdBF16Plain, EpResidF32, (p.A = Ab, p.lda = LDA, p.stats_out = b.stats_out))
        RUN_SKW("skinny WM2 WK4 LN gate (no dw)", 2, 4, 1, true, 2, LdF32LN, EpGateBF16, (p.A = Ab, p.lda = LDA, p.film = b.film, p.film_gain_off = 0, p.film_bias_off = K, p.stats_in = b.stats_in, p.out = b.outb, p.ldo = N / 2))
        RUN_SKW("skinny WM4 WK2 LN gate (no dw)", 4, 2, 1, true, 2, LdF32LN, EpGateBF16, (p.A = Ab, p.lda = LDA, p.film = b.film, p.film_gain_off = 0, p.film_bias_off = K, p.stats_in = b.stats_in, p.out = b.outb, p.ldo = N / 2))
        RUN_SKW("skinny WM8 WK1 LN gate (no dw)", 8, 1, 1, true, 2, LdF32LN, EpGateBF16, (p.A = Ab, p.lda = LDA, p.film = b.film, p.film_gain_off = 0, p.film_bias_off = K, p.stats_in = b.stats_in, p.out = b.outb, p.ldo = N / 2))
    }
    if (getenv("CHAIN") && (K == 128 || K == 256) && K == N) {
        // ---- the row-local NAF tail as one kernel (hd_chain.hpp): phase stamps ----
        const int C = K, hw = (C == 128) ? 256 : 64;
        ChainP q{};
        float *pooled, *X, *Xo, *bv; unsigned short *G, *Xb; float2* so;
        CK(hipMalloc(&pooled, (size_t)(M / hw) * C * 4)); CK(hipMalloc(&X, (size_t)M * C * 4)); CK(hipMalloc(&Xo, (size_t)M * C * 4));
        CK(hipMalloc(&bv, (size_t)4 * C * 4)); CK(hipMalloc(&G, (size_t)M * C * 2)); CK(hipMalloc(&Xb, (size_t)M * C * 2)); CK(hipMalloc(&so, (size_t)M * (C / 32) * 8));
        hipLaunchKernelGGL(fillf_kernel, dim3(64), dim3(256), 0, s, pooled, (size_t)(M / hw) * C, 0.5f);
        hipLaunchKernelGGL(fillf_kernel, dim3(256), dim3(256), 0, s, X, (size_t)M * C, 1.0f);
        hipLaunchKernelGGL(fillf_kernel, dim3(64), dim3(256), 0, s, bv, (size_t)4 * C, 0.1f);
        hipLaunchKernelGGL(fill_kernel, dim3(256), dim3(256), 0, s, (unsigned*)G, (size_t)M * C / 2, 77u);
        q.M = M; q.hw = hw; q.G = G; q.pooled = pooled; q.X = X;
        q.Wsca = b.W[0]; q.W3 = b.W[1 % nrot]; q.W4 = b.W[2 % nrot]; q.W5 = b.W[3 % nrot];
        q.bsca = bv; q.b3 = bv; q.b4 = bv; q.b5 = bv; q.beta = bv; q.gamma = bv;
        q.film = b.film; q.film_gain_off = 0; q.film_bias_off = C; q.ln_eps = 1e-6f;
        q.Xout = Xo; q.Xout16 = Xb; q.stats_out = so; q.stamps = nullptr;
        auto run = [&](const ChainP& qq) { return C == 128 ? launch_chain<128, 1>(qq, s) : launch_chain<256, 1>(qq, s); };
        float us = time_it([&](int) { (void)run(q); }, iters, s);
        CK(hipMemsetAsync(stamps, 0, 8192 * 64, s)); q.stamps = stamps; (void)run(q); CK(hipStreamSynchronize(s));
        const int nwg = M / 32;
        std::vector<unsigned long long> h((size_t)nwg * 8); CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
        const char* nm[6] = {"sca", "stageA1", "conv3", "LN", "conv4", "conv5+epi"};
        printf("chain C=%d M=%d: %.2f us;", C, M, us);
        unsigned long long t0 = ~0ull, t6 = 0;
        for (int w = 0; w < nwg; ++w) { t0 = std::min(t0, h[(size_t)w * 8]); t6 = std::max(t6, h[(size_t)w * 8 + 6]); }
        printf(" span %.2f;", (double)(t6 - t0) * 0.01);
        for (int i = 0; i < 6; ++i) { std::vector<double> d; for (int w = 0; w < nwg; ++w) d.push_back((double)(h[(size_t)w * 8 + i + 1] - h[(size_t)w * 8 + i]) * 0.01); std::sort(d.begin(), d.end()); printf(" %s %.2f/%.2f", nm[i], d[d.size() / 2], d.back()); }
        { std::vector<double> d; for (int w = 0; w < nwg; ++w) d.push_back((double)(h[(size_t)w * 8] - t0) * 0.01); std::sort(d.begin(), d.end()); printf(" | start skew %.2f/%.2f\n", d[d.size() / 2], d.back()); }
        return 0;
    }
    if (M >= 2048 && N >= 256) {
        float* pooled; CK(hipMalloc(&pooled, (size_t)(M / 16) * N * 4));
        float* dww; CK(hipMalloc(&dww, (size_t)N * 9 * 4)); hipLaunchKernelGGL(fillf_kernel, dim3(64), dim3(256), 0, s, dww, (size_t)N * 9, 0.3f);
        const int hw = (M == 16384) ? 256 : 64, side = (M == 16384) ? 16 : 8;
        if (hw == 256) {
            RUN_SKW("skinny WM8 LN dwgate (L0 conv1 fused)", 8, 1, 1, true, 1, LdF32LN, EpDwGate, (p.A = Ab, p.lda = LDA, p.film = b.film, p.film_gain_off = 0, p.film_bias_off = K, p.stats_in = b.stats_in, p.out = b.outb, p.ldo = N / 2, p.dw_w = dww, p.dw_b = b.bias, p.pooled = pooled, p.hw = hw, p.side = side))
            RUN_SKW("skinny WM8 LN gate (no dw)", 8, 1, 1, true, 1, LdF32LN, EpGateBF16, (p.A = Ab, p.lda = LDA, p.film = b.film, p.film_gain_off = 0, p.film_bias_off = K, p.stats_in = b.stats_in, p.out = b.outb, p.ldo = N / 2, p.hw = hw))
            RUN_SKW("skinny WM8 D2 LN gate (no dw)", 8, 1, 1, true, 2, LdF32LN, EpGateBF16, (p.A = Ab, p.lda = LDA, p.film = b.film, p.film_gain_off = 0, p.film_bias_off = K, p.stats_in = b.stats_in, p.out = b.outb, p.ldo = N / 2, p.hw = hw))
        } else {
            RUN_SKW("skinny WM2 WK2 LN dwgate (L1 conv1 fused)", 2, 2, 1, true, 1, LdF32LN, EpDwGate, (p.A = Ab, p.lda = LDA, p.film = b.film, p.film_gain_off = 0, p.film_bias_off = K, p.stats_in = b.stats_in, p.out = b.outb, p.ldo = N / 2, p.dw_w = dww, p.dw_b = b.bias, p.pooled = pooled, p.hw = hw, p.side = side))
        }
        RUN_SKW("skinny WM4 bf16plain resid +stats", 4, 1, 1, false, 2, LdBF16Plain, EpResidF32, (p.A = Ab, p.lda = LDA, p.stats_out = b.stats_out))
        RUN_SKW("skinny WM8 bf16plain resid +stats", 8, 1, 1, false, 2, LdBF16Plain, EpResidF32, (p.A = Ab, p.lda = LDA, p.stats_out = b.stats_out))
    }
    if (M >= 1024) {
        RUN_TALL("tall T32W bf16plain resid", T32W, LdBF16Plain, EpResidF32, (p.A = Ab, p.lda = LDA, p.stats_out = b.stats_out))
        RUN_TALL("tall T32W bf16plain resid (no stats)", T32W, LdBF16Plain, EpResidF32, (p.A = Ab, p.lda = LDA))
        RUN_TALL("tall T64  bf16plain resid", T64, LdBF16Plain, EpResidF32, (p.A = Ab, p.lda = LDA, p.stats_out = b.stats_out))
        RUN_TALL("tall T128 bf16plain resid", T128, LdBF16Plain, EpResidF32, (p.A = Ab, p.lda = LDA, p.stats_out = b.stats_out))
        RUN_TALL("tall T128 bf16scale resid", T128, LdBF16Scale, EpResidF32, (p.A = Ab, p.lda = LDA, p.rowscale = b.rowscale, p.hw = 256, p.stats_out = b.stats_out))
        RUN_TALL("tall T128 LN bias", T128, LdF32LN, EpBiasF32, (p.A = Ab, p.lda = LDA, p.film = b.film, p.film_gain_off = 0, p.film_bias_off = K, p.stats_in = b.stats_in))
        RUN_TALL("tall T128P LN gate", T128P, LdF32LN, EpGateBF16, (p.A = Ab, p.lda = LDA, p.film = b.film, p.film_gain_off = 0, p.film_bias_off = K, p.stats_in = b.stats_in, p.out = b.outb, p.ldo = N / 2))
    }
    return 0;
}
