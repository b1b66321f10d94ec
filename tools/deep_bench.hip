// deep_bench.hip — the deep-prefetch tall GEMM kernels (hd_gemm.hpp: gemm_deep_kernel, gemm_deep_pair8_kernel) on the latent-32 shapes of
// levels 2 / 3, diagnostic build with in-kernel stamps (tools only): per workgroup, the time to issue the first loads, to the first
// chunk barrier, through the K loop and through the epilogue.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/deep_bench_bin tools/deep_bench.hip ; run: tools/deep_bench_bin
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <cstring>
#include <type_traits>
#define HD_STAMPS 1
#include "../hifidiff_amd/csrc/hd_gemm.hpp"
#include "../hifidiff_amd/csrc/hd_wide.hpp"
using namespace hd;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void fill_u(unsigned* p, size_t n, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        unsigned lo = 0x3c00u + (x & 0x3ffu), hi = 0x3c00u + ((x >> 10) & 0x3ffu);
        p[i] = (lo | ((x >> 20) & 1u) << 15) | ((hi | ((x >> 21) & 1u) << 15) << 16);
    }
}
__global__ void fill_f(float* p, size_t n, float v) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v * (float)((i * 7919) % 1000) / 1000.f;
}
template <class T> T* dm(size_t n) { T* p; CK(hipMalloc(&p, n * sizeof(T))); CK(hipMemset(p, 0, n * sizeof(T))); return p; }

static void report(unsigned long long* dev, int nwg, float us) {
    std::vector<unsigned long long> h((size_t)nwg * 8);
    CK(hipMemcpy(h.data(), dev, h.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull, t5 = 0;
    std::vector<double> d[5], st;
    for (int w = 0; w < nwg; ++w) {
        const unsigned long long* t = &h[(size_t)w * 8];
        t0 = std::min(t0, t[0]); t5 = std::max(t5, t[5]);
        for (int i = 0; i < 5; ++i) d[i].push_back((double)(t[i + 1] - t[i]) * 0.01);
    }
    for (int w = 0; w < nwg; ++w) st.push_back((double)(h[(size_t)w * 8] - t0) * 0.01);
    std::sort(st.begin(), st.end());
    printf("  %.2f us per launch (events) | in-kernel span %.2f, start skew med %.2f max %.2f |", us, (double)(t5 - t0) * 0.01, st[st.size() / 2], st.back());
    const char* nm[5] = {"issue", "to 1st barrier", "K loop", "-", "epilogue"};
    for (int i = 0; i < 5; ++i) { if (i == 3) continue; std::sort(d[i].begin(), d[i].end()); printf(" %s %.2f/%.2f", nm[i], d[i][d[i].size() / 2], d[i].back()); }
    printf(" (median/max over %d workgroups)\n", nwg);
}
// hd_wide.hpp: per-stage stamps of workgroup 0 (HD_WSTAMP)
static void report_wide(unsigned long long* dev) {
    std::vector<unsigned long long> h(128);
    CK(hipMemcpy(h.data(), dev + 4096, 128 * 8, hipMemcpyDeviceToHost));
    if (!h[64]) return;
    const unsigned long long t0 = std::min(h[0], h[64]);
    printf("  workgroup 0, per stage (us from the first stamp): staging wave [start, loads issued, stage s+1 stored, barrier passed] | MFMA wave [start, -, MFMAs + weight requests done, barrier passed]\n");
    for (int s = 0; s < 8; ++s) {
        printf("    stage %d: staging %.2f %.2f %.2f %.2f | MFMA %.2f %.2f %.2f\n", s, (h[64 + 4 * s] - t0) * 0.01, (h[64 + 4 * s + 1] - t0) * 0.01, (h[64 + 4 * s + 2] - t0) * 0.01,
               (h[64 + 4 * s + 3] - t0) * 0.01, (h[4 * s] - t0) * 0.01, (h[4 * s + 2] - t0) * 0.01, (h[4 * s + 3] - t0) * 0.01);
    }
}

template <class LD, class EP, bool PAIR, bool WIDE = false>
void run(const char* name, int M, int K, int N, bool ln, int side = 0) {
    hipStream_t s; CK(hipStreamCreate(&s));
    const size_t wbytes = (size_t)N * K * 2;
    const int nrot = (int)((700ull << 20) / wbytes) + 1;          // weights rotate through more than the Infinity Cache holds
    std::vector<uint4*> W(nrot);
    for (auto& w : W) { w = dm<uint4>(wbytes / 16); fill_u<<<256, 256, 0, s>>>((unsigned*)w, wbytes / 4, 17); }
    unsigned short* A16 = dm<unsigned short>((size_t)M * K); fill_u<<<256, 256, 0, s>>>((unsigned*)A16, (size_t)M * K / 2, 5);
    const int ncols = PAIR ? N / 2 : N;
    float* out = dm<float>((size_t)M * ncols); float* resid = dm<float>((size_t)M * ncols); fill_f<<<64, 256, 0, s>>>(resid, (size_t)M * ncols, 1.f);
    unsigned short* out16 = dm<unsigned short>((size_t)M * ncols);
    float* bias = dm<float>(N); float* rscale = dm<float>(N); fill_f<<<8, 256, 0, s>>>(rscale, N, 1.f);
    float* film = dm<float>(2 * K); fill_f<<<8, 256, 0, s>>>(film, 2 * K, 1.f);
    const int NP = K / 32;
    std::vector<float2> sth((size_t)M * NP, make_float2(0.5f, 32 * 0.08f));
    float2* stats = dm<float2>((size_t)M * NP); CK(hipMemcpy(stats, sth.data(), sth.size() * 8, hipMemcpyHostToDevice));
    float2* stats_out = dm<float2>((size_t)M * (ncols / 32));
    unsigned long long* stamps = dm<unsigned long long>(8192 * 8);
    float* dww = dm<float>((size_t)9 * N); fill_f<<<8, 256, 0, s>>>(dww, (size_t)9 * N, 0.3f);
    float* pooled = dm<float>((size_t)M * N); unsigned short* pooled16 = dm<unsigned short>((size_t)M * N);
    auto base = [&](int i) {
        GemmP p{};
        p.M = M; p.N = N; p.K = K; p.Kp = K; p.nt_total = N / 32; p.W = W[i % nrot]; p.A = A16; p.lda = K;
        p.a_scale = 1.f; p.hw = 1; p.ln_eps = 1e-6f; p.shuffle_r = 1; p.stats_np = NP; p.stats_cnt = 32; p.stats_in = stats;
        p.film = film; p.film_gain_off = 0; p.film_bias_off = K;
        p.bias = bias; p.out = PAIR ? (void*)out16 : (void*)out; p.ldo = ncols;
        if (!PAIR) { p.resid = resid; p.ldr = ncols; p.rscale = rscale; p.stats_out = stats_out; p.out16 = out16; }
        p.xcd_tile_affine = ((size_t)N * K * 2 > (size_t)M * K * 2) ? 1 : 0;
        if (side) { p.side = side; p.hw = side * side; p.dw_w = dww; p.dw_b = bias; p.pooled = pooled; p.pooled16 = pooled16; }
        return p;
    };
    auto launch = [&](const GemmP& q) -> hipError_t {
        if constexpr (WIDE) return launch_gemm_wide<std::is_same<LD, LdF32LN_T<false>>::value, EP, PAIR>(q, s);
        else if (K == 2048) return launch_skinny_auto<1, 2, PAIR, LD, EP>(q, s);
        else return launch_gemm_deep<LD, EP, PAIR>(q, s);
    };
    if (WIDE ? !wide_shape_ok<PAIR>(base(0)) : (K != 2048 && !deep_shape_ok<PAIR>(base(0)))) { printf("%s: shape not taken by this kernel\n", name); return; }
    if constexpr (WIDE) {   // same inputs through the deep kernel first: the two outputs side by side
        const size_t n = (size_t)M * ncols;
        GemmP q = base(0);
        if (K == 2048) { CK((launch_skinny_auto<1, 2, PAIR, LD, EP>(q, s))); }     // middle level: the reference is the 64-row skinny tile the library uses there
        else { CK((launch_gemm_deep<LD, EP, PAIR>(q, s))); }
        CK(hipStreamSynchronize(s));
        std::vector<unsigned short> h16(n); std::vector<float> h32(n);
        CK(hipMemcpy(h16.data(), out16, n * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(h32.data(), out, n * 4, hipMemcpyDeviceToHost));
        CK(hipMemset(out16, 0, n * 2)); CK(hipMemset(out, 0, n * 4));
        CK(launch(q)); CK(hipStreamSynchronize(s));
        std::vector<unsigned short> g16(n); std::vector<float> g32(n);
        CK(hipMemcpy(g16.data(), out16, n * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(g32.data(), out, n * 4, hipMemcpyDeviceToHost));
        double num = 0, den = 0, mx = 0; size_t diff16 = 0;
        auto b2f = [](unsigned short b) { unsigned u = (unsigned)b << 16; float f; memcpy(&f, &u, 4); return f; };
        for (size_t i = 0; i < n; ++i) {
            const double a = PAIR ? b2f(h16[i]) : h32[i], b = PAIR ? b2f(g16[i]) : g32[i];
            num += (a - b) * (a - b); den += a * a; mx = std::max(mx, std::abs(a - b)); diff16 += h16[i] != g16[i];
        }
        printf("  wide vs deep kernel on the same inputs: rel-L2 %.3e, max abs %.3e, bf16 outputs that differ %zu of %zu (|ref| rms %.3e)\n", std::sqrt(num / (den + 1e-30)), mx, diff16, n, std::sqrt(den / n));
    }
    (void)ln;
    for (int i = 0; i < 5; ++i) CK(launch(base(i)));
    CK(hipStreamSynchronize(s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 100;
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i) CK(launch(base(i)));
    CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemset(stamps, 0, 8192 * 8 * 8));
    GemmP p = base(3); p.stamps = stamps;
    CK(launch(p)); CK(hipStreamSynchronize(s));
    printf("%s M=%d K=%d N=%d:\n", name, M, K, N);
    report(stamps, ((M + 127) / 128) * (ncols / 32), ms * 1000.f / iters);
    if constexpr (WIDE) report_wide(stamps);
    for (auto w : W) CK(hipFree(w));
}

int main(int argc, char** argv) {
    if (argc > 1) {     // the wide form (hd_wide.hpp) of the level-3 shapes against the deep kernel
        run<LdF32LN_T<false>, EpGateBF16, true>("LN -> conv4 -> gate (pair8)", 1024, 1024, 2048, true);
        run<LdF32LN_T<false>, EpGateBF16, true, true>("LN -> conv4 -> gate (WIDE)", 1024, 1024, 2048, true);
        run<LdBF16Plain, EpResidF32, false, true>("bf16 -> conv5 -> residual (WIDE)", 1024, 1024, 1024, false);
        run<LdBF16Plain, EpGateBF16, true, true>("WHAT-IF bf16 -> conv4 -> gate (WIDE, no LayerNorm)", 1024, 1024, 2048, false);
        run<LdF32LN_T<false>, EpDwGate, true>("LN -> conv1 -> depthwise -> gate (pair8)", 1024, 1024, 2048, true, 4);
        run<LdF32LN_T<false>, EpDwGate, true, true>("LN -> conv1 -> depthwise -> gate (WIDE)", 1024, 1024, 2048, true, 4);
        run<LdF32LN_T<false>, EpGateBF16, true>("LN -> conv4 -> gate (skinny 64 rows)", 256, 2048, 4096, true);
        run<LdF32LN_T<false>, EpDwGate, true>("LN -> conv1 -> depthwise -> gate (skinny 64 rows)", 256, 2048, 4096, true, 2);
        run<LdF32LN_T<false>, EpGateBF16, true, true>("LN -> conv4 -> gate (WIDE, 64 rows)", 256, 2048, 4096, true);
        run<LdF32LN_T<false>, EpDwGate, true, true>("LN -> conv1 -> depthwise -> gate (WIDE, 64 rows)", 256, 2048, 4096, true, 2);
        run<LdF32LN_T<false>, EpGateBF16, true>("LN -> conv4 -> gate (pair8)", 4096, 512, 1024, true);
        run<LdF32LN_T<false>, EpGateBF16, true, true>("LN -> conv4 -> gate (WIDE, 256 rows)", 4096, 512, 1024, true);
        run<LdBF16Plain, EpResidF32, false>("bf16 -> conv5 -> residual (deep)", 4096, 512, 512, false);
        run<LdBF16Plain, EpResidF32, false, true>("bf16 -> conv5 -> residual (WIDE, 256 rows)", 4096, 512, 512, false);
        run<LdF32LN_T<false>, EpDwGate, true>("LN -> conv1 -> depthwise -> gate (pair8)", 4096, 512, 1024, true, 8);
        run<LdF32LN_T<false>, EpDwGate, true, true>("LN -> conv1 -> depthwise -> gate (WIDE, 256 rows)", 4096, 512, 1024, true, 8);
        return 0;
    }
    run<LdF32LN_T<false>, EpGateBF16, true>("LN -> conv4 -> gate (pair8)", 1024, 1024, 2048, true);
    run<LdF32LN_T<false>, EpGateBF16, true>("LN -> conv4 -> gate (pair8)", 4096, 512, 1024, true);
    run<LdBF16Plain, EpResidF32, false>("bf16 -> conv5 -> residual (deep)", 4096, 512, 512, false);
    // what-if: the pair GEMM without the LayerNorm transform (plain bf16 A): what the transform and its FiLM reads cost inside the K loop
    run<LdBF16Plain, EpGateBF16, true>("WHAT-IF bf16 -> conv4 -> gate (pair8, no LayerNorm)", 1024, 1024, 2048, false);
    run<LdBF16Plain, EpGateBF16, true>("WHAT-IF bf16 -> conv4 -> gate (pair8, no LayerNorm)", 4096, 512, 1024, false);
    run<LdF32LN_T<false>, EpDwGate, true>("LN -> conv1 -> depthwise -> gate (pair8)", 1024, 1024, 2048, true, 4);
    run<LdF32LN_T<false>, EpDwGate, true>("LN -> conv1 -> depthwise -> gate (pair8)", 4096, 512, 1024, true, 8);
    return 0;
}
