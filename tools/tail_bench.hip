// tail_bench.hip — in-kernel stamps of the persistent middle-level kernel (hd_tail.hpp) on synthetic data (tools only).
// Timing and phase shares only: correctness is covered by tests/test_gpu_parity.py through the library.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/tb tools/tail_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define HD_STAMPS 1
#include "experiments/hd_tail.hpp"
using namespace hd;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void fill_bf16(unsigned* p, size_t n, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        unsigned lo = 0x3c00u + (x & 0x3ffu), hi = 0x3c00u + ((x >> 10) & 0x3ffu);
        p[i] = (lo | ((x >> 20) & 1u) << 15) | ((hi | ((x >> 21) & 1u) << 15) << 16);
    }
}
__global__ void fill_f32(float* p, size_t n, float v, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = v * ((float)(x & 0xffff) / 32768.f - 1.f);
    }
}
template <class T> T* dalloc(size_t n) { T* p; CK(hipMalloc(&p, n * sizeof(T))); return p; }

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 50;
    const int nb = 8, P = 5 * nb;
    hipStream_t s; CK(hipStreamCreate(&s));
    std::vector<TailBlockW> hb(nb);
    auto wgt = [&](size_t n, unsigned seed) { uint4* w = dalloc<uint4>(n / 8); hipLaunchKernelGGL(fill_bf16, dim3(1024), dim3(256), 0, s, (unsigned*)w, n / 2, seed); return (const uint4*)w; };
    auto vec = [&](size_t n, float v, unsigned seed) { float* f = dalloc<float>(n); hipLaunchKernelGGL(fill_f32, dim3(64), dim3(256), 0, s, f, n, v, seed); return (const float*)f; };
    const size_t CC = (size_t)TL_C * TL_C;
    for (int j = 0; j < nb; ++j) {
        TailBlockW& t = hb[j];
        t.w1 = wgt(2 * CC, 11 * j + 1); t.wsca = wgt(CC, 11 * j + 2); t.w3 = wgt(CC, 11 * j + 3); t.w4 = wgt(2 * CC, 11 * j + 4); t.w5 = wgt(CC, 11 * j + 5);
        t.b1 = vec(2 * TL_C, 0.1f, 1); t.bsca = vec(TL_C, 0.1f, 2); t.b3 = vec(TL_C, 0.1f, 3); t.b4 = vec(2 * TL_C, 0.1f, 4); t.b5 = vec(TL_C, 0.1f, 5);
        t.beta = vec(TL_C, 0.2f, 6); t.gamma = vec(TL_C, 0.2f, 7); t.dw_c = vec(2 * TL_C, 0.3f, 8); t.dw_b = vec(2 * TL_C, 0.1f, 9);
        t.film_off = j * 4 * TL_C;
    }
    TailP p{};
    p.M = 64; p.nblocks = nb;
    TailBlockW* db = dalloc<TailBlockW>(nb); CK(hipMemcpy(db, hb.data(), nb * sizeof(TailBlockW), hipMemcpyHostToDevice)); p.blocks = db;
    p.X = vec((size_t)64 * TL_C, 1.f, 21);
    unsigned short* xb = dalloc<unsigned short>((size_t)64 * TL_C); hipLaunchKernelGGL(fill_bf16, dim3(64), dim3(256), 0, s, (unsigned*)xb, (size_t)64 * TL_C / 2, 5u); p.Xb = xb;
    std::vector<float2> st((size_t)64 * 64, make_float2(0.01f, 0.5f));
    float2* sx = dalloc<float2>(st.size()); CK(hipMemcpy(sx, st.data(), st.size() * 8, hipMemcpyHostToDevice)); p.sx = sx; p.sx_np = 64;
    p.film = vec((size_t)nb * 4 * TL_C, 1.f, 31); p.ln_eps = 1e-6f;
    p.Xout = dalloc<float>((size_t)64 * TL_C); p.Xout16 = dalloc<unsigned short>((size_t)64 * TL_C); p.stats_out = dalloc<float2>((size_t)64 * 64);
    p.act = dalloc<uint4>(TL_SLABS * TL_SLAB_U4); p.stats = dalloc<float2>(TL_SLABS * TL_STAT_F2);
    p.flags = dalloc<unsigned>((size_t)P * TL_WG); p.state = dalloc<unsigned>(64);
    CK(hipMemset(p.flags, 0, (size_t)P * TL_WG * 4)); CK(hipMemset(p.state, 0, 256));
    unsigned* tmo_h; CK(hipHostMalloc((void**)&tmo_h, 64, hipHostMallocMapped)); tmo_h[0] = 0;
    CK(hipHostGetDevicePointer((void**)&p.tmo, tmo_h, 0));
    unsigned long long* stamps = dalloc<unsigned long long>((size_t)P * TL_WG * 6);
    p.stamps = nullptr;
    CK(hipStreamSynchronize(s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) (void)launch_mid_tail(p, s);
    CK(hipStreamSynchronize(s));
    if (tmo_h[0]) { printf("timeout word 0x%x\n", tmo_h[0]); return 1; }
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < reps; ++i) (void)launch_mid_tail(p, s);
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("mid_tail_kernel, 8 blocks (469.8 MB of weights): %.1f us per launch, %.2f us per phase, %.2f TB/s; timeout word 0x%x\n",
           ms * 1000 / reps, ms * 1000 / reps / P, 469.76e6 / (ms / reps * 1e-3) / 1e12, tmo_h[0]);
    p.stamps = stamps; CK(hipMemset(stamps, 0, (size_t)P * TL_WG * 48));
    (void)launch_mid_tail(p, s); CK(hipStreamSynchronize(s));
    std::vector<unsigned long long> h((size_t)P * TL_WG * 6);
    CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
    const char* qn[5] = {"conv1 (LN, pair)", "sca", "conv3", "conv4 (LN, pair)", "conv5"};
    double tot[5][4] = {};
    for (int ph = 5; ph < P; ++ph) {
        for (int k = 0; k < 4; ++k) {
            std::vector<double> d;
            for (int wg = 0; wg < TL_WG; ++wg) { const unsigned long long* t = &h[((size_t)ph * TL_WG + wg) * 6]; d.push_back((double)(t[k + 1] - t[k]) * 0.01); }
            std::sort(d.begin(), d.end()); tot[ph % 5][k] += d[d.size() / 2];
        }
    }
    printf("median over workgroups, mean over blocks 1..7 (us):   wait-flags   A+LN+mfma   red-wait   epilogue+publish   sum\n");
    double all = 0;
    for (int q = 0; q < 5; ++q) {
        double sum = 0; for (int k = 0; k < 4; ++k) sum += tot[q][k] / (nb - 1);
        printf("  %-18s %38.2f %11.2f %10.2f %18.2f %6.2f\n", qn[q], tot[q][0] / (nb - 1), tot[q][1] / (nb - 1), tot[q][2] / (nb - 1), tot[q][3] / (nb - 1), sum);
        all += sum;
    }
    unsigned long long tmin = ~0ull, tmax = 0;
    for (int wg = 0; wg < TL_WG; ++wg) { tmin = std::min(tmin, h[(size_t)wg * 6]); tmax = std::max(tmax, h[((size_t)(P - 1) * TL_WG + wg) * 6 + 4]); }
    printf("  per block %.2f us; stamped span %.1f us (%.2f per phase)\n", all, (double)(tmax - tmin) * 0.01, (double)(tmax - tmin) * 0.01 / P);
    return 0;
}
