// mid_stage_bench.hip — the chip-wide persistent middle-level stage (tools/hd_mid.hpp; measured negative, not in the library) on its own (tools only):
// synthetic weights and activations of the middle level's shapes (C = 2048, one pixel per face, batch 64, 8 blocks),
//   (1) a correctness run against a naive fp32 reference of the same arithmetic (bf16 operands rounded at the same points, one thread
//       per output element) -- so that what is timed is the real computation, and
//   (2) the diagnostic build's in-kernel stamps: per phase the median over the finishing workgroups (compute wave 0 of member 0) of flag wait, activation loads
//       (+ statistics, transform) + K loop, split-K exchange, epilogue + stores issued, publish (store drain + flag), and the span of the
//       phase over the whole chip.  Gate set by the review: mean <= 5.5 us per phase over the 5-phase block.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DHD_STAMPS -o tools/mid_stage_bench_bin tools/mid_stage_bench.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "hd_mid.hpp"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
using namespace hd;
constexpr int C = 2048;

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
template <class T> T* dmalloc(size_t n) { T* p; CK(hipMalloc(&p, n * sizeof(T))); CK(hipMemset(p, 0, n * sizeof(T))); return p; }

// ---- entry data in the standard layouts: bf16 copy and (mean, M2) partials of 32 channels ----
__global__ void prep_entry(const float* X, unsigned short* Xb, float2* sx, int B) {
    const int row = blockIdx.x, t = threadIdx.x;                       // 64 threads: one tile of 32 channels each
    if (row >= B) return;
    const float* x = X + (size_t)row * C + t * 32;
    float s = 0.f;
    for (int i = 0; i < 32; ++i) { s += x[i]; Xb[(size_t)row * C + t * 32 + i] = f32_to_bf16_bits(x[i]); }
    const float mean = s / 32.f;
    float q = 0.f;
    for (int i = 0; i < 32; ++i) { const float d = x[i] - mean; q += d * d; }
    sx[row * 64 + t] = make_float2(mean, q);
}
// ---- naive reference ----
__device__ float bf16r(float v) { return bf16_bits_to_f32(f32_to_bf16_bits(v)); }
__device__ float wval(const uint4* W, int n, int k) {                 // pack_weight16_kernel's order
    const unsigned short* p = reinterpret_cast<const unsigned short*>(W);
    return bf16_bits_to_f32(p[((((size_t)(n >> 4) * 64 + (k >> 5)) * 64) + (n & 15) + 16 * ((k & 31) >> 3)) * 8 + (k & 7)]);
}
__global__ void ref_ln(const float* X, const float* film_bias, const float* film_gain, float* A, float eps) {      // one block per row
    const int row = blockIdx.x;
    __shared__ float red[256];
    const float* x = X + (size_t)row * C;
    float s = 0.f;
    for (int k = threadIdx.x; k < C; k += 256) s += x[k];
    red[threadIdx.x] = s; __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    const float mean = red[0] / C; __syncthreads();
    float q = 0.f;
    for (int k = threadIdx.x; k < C; k += 256) { const float d = x[k] - mean; q += d * d; }
    red[threadIdx.x] = q; __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    const float rstd = 1.0f / sqrtf(red[0] / C + eps), nmr = -mean * rstd;
    for (int k = threadIdx.x; k < C; k += 256) A[(size_t)row * C + k] = bf16r(fmaf(fmaf(bf16r(x[k]), rstd, nmr), film_gain[k], film_bias[k]));
}
__global__ void ref_gemm(const float* A, const uint4* W, float* out, int N) {      // out[row][n] = sum_k A[row][k] W[n][k]
    const int n = blockIdx.x * blockDim.x + threadIdx.x, row = blockIdx.y;
    if (n >= N) return;
    const float* a = A + (size_t)row * C;
    float s = 0.f;
    for (int k = 0; k < C; ++k) s = fmaf(a[k], wval(W, n, k), s);
    out[(size_t)row * N + n] = s;
}
__global__ void ref_q0(const float* T, const float* b1, const float* dw_w, const float* dw_b, float* gq) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x, row = blockIdx.y;
    const float ua = fmaf(dw_w[(size_t)4 * 2 * C + c], T[(size_t)row * 2 * C + c] + b1[c], dw_b[c]);
    const float ub = fmaf(dw_w[(size_t)4 * 2 * C + C + c], T[(size_t)row * 2 * C + C + c] + b1[C + c], dw_b[C + c]);
    gq[(size_t)row * C + c] = bf16r(ua * ub);
}
__global__ void ref_q1(const float* T, const float* bsca, const float* gq, float* gs) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x, row = blockIdx.y;
    gs[(size_t)row * C + c] = bf16r(gq[(size_t)row * C + c] * (T[(size_t)row * C + c] + bsca[c]));
}
__global__ void ref_resid(const float* T, const float* b, const float* scale, const float* x, float* y) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x, row = blockIdx.y;
    y[(size_t)row * C + c] = fmaf(T[(size_t)row * C + c] + b[c], scale[c], x[(size_t)row * C + c]);
}
__global__ void ref_q3(const float* T, const float* b4, float* g2) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x, row = blockIdx.y;
    g2[(size_t)row * C + c] = bf16r((T[(size_t)row * 2 * C + c] + b4[c]) * (T[(size_t)row * 2 * C + C + c] + b4[C + c]));
}

struct Bench {
    int nblocks, B;
    std::vector<XBlockW> hb;
    MStageP p{};
    float *X0, *film;
    unsigned* tmo_h;
    hipStream_t st;
};

static Bench setup(int nblocks, int B) {
    Bench b; b.nblocks = nblocks; b.B = B;
    b.hb.resize(nblocks);
    unsigned seed = 1;
    for (auto& k : b.hb) {
        auto w = [&](size_t n, float sc) { unsigned short* q = dmalloc<unsigned short>(n); fill_bf16<<<512, 256>>>(q, n, seed++, sc); return reinterpret_cast<const uint4*>(q); };
        auto f = [&](size_t n, float sc, float off) { float* q = dmalloc<float>(n); fill_f32<<<64, 256>>>(q, n, seed++, sc, off); return (const float*)q; };
        const float ws = 1.7f / sqrtf((float)C);
        k.w1 = w((size_t)2 * C * C, ws); k.wsca = w((size_t)C * C, ws); k.w3 = w((size_t)C * C, ws); k.w4 = w((size_t)2 * C * C, ws); k.w5 = w((size_t)C * C, ws);
        k.b1 = f(2 * C, 0.1f, 0.f); k.bsca = f(C, 0.1f, 1.f); k.b3 = f(C, 0.1f, 0.f); k.b4 = f(2 * C, 0.1f, 0.f); k.b5 = f(C, 0.1f, 0.f);
        k.beta = f(C, 0.2f, 0.f); k.gamma = f(C, 0.2f, 0.f); k.dw_w = f((size_t)9 * 2 * C, 0.3f, 0.8f); k.dw_b = f(2 * C, 0.1f, 0.2f);
        k.film_off = (int)(&k - b.hb.data()) * 4 * C; k.pad_ = 0;
    }
    MStageP& p = b.p;
    p.B = B; p.nblocks = nblocks;
    XBlockW* db = dmalloc<XBlockW>(nblocks); CK(hipMemcpy(db, b.hb.data(), nblocks * sizeof(XBlockW), hipMemcpyHostToDevice)); p.blocks = db;
    b.X0 = dmalloc<float>((size_t)64 * C); fill_f32<<<256, 256>>>(b.X0, (size_t)64 * C, 77, 1.f, 0.1f);
    p.X = dmalloc<float>((size_t)64 * C);
    p.Xb = dmalloc<unsigned short>((size_t)64 * C);
    p.sx = dmalloc<float2>((size_t)64 * 64);
    const size_t hn = (size_t)64 * C / 8;
    p.hP = dmalloc<uint4>(hn); p.hG = dmalloc<uint4>(hn); p.hY = dmalloc<uint4>(hn); p.hG2 = dmalloc<uint4>(hn); p.hX = dmalloc<uint4>(hn);
    p.hsx = dmalloc<float2>((size_t)64 * 64); p.hsy = dmalloc<float2>((size_t)64 * 64);
    p.xbuf = dmalloc<uint4>((size_t)64 * 4 * 4 * 4 * 4 * 64);
    b.film = dmalloc<float>((size_t)nblocks * 4 * C); fill_f32<<<64, 256>>>(b.film, (size_t)nblocks * 4 * C, 80, 0.2f, 1.f); p.film = b.film; p.ln_eps = 1e-6f;
    unsigned* sync = dmalloc<unsigned>(1024 + 1024 + 256 + 256 + 64);
    p.flags = sync; p.xflags = sync + 1024; p.hello = sync + 2048; p.gstate = sync + 2304; p.abort_dev = sync + 2560;
    CK(hipHostMalloc(reinterpret_cast<void**>(&b.tmo_h), 64, hipHostMallocMapped)); b.tmo_h[0] = 0;
    CK(hipHostGetDevicePointer(reinterpret_cast<void**>(&p.tmo), b.tmo_h, 0));
    CK(hipStreamCreate(&b.st));
    CK(hipDeviceSynchronize());
    return b;
}
static void reset_x(Bench& b) {
    CK(hipMemcpyAsync(b.p.X, b.X0, (size_t)64 * C * 4, hipMemcpyDeviceToDevice, b.st));
    prep_entry<<<64, 64, 0, b.st>>>(b.p.X, b.p.Xb, const_cast<float2*>(b.p.sx), b.B);
}

static int check(int nblocks, int B, int force_global) {
    Bench b = setup(nblocks, B);
    b.p.force_global = force_global;
    reset_x(b);
    CK(launch_mid_stage(b.p, b.st));
    CK(hipStreamSynchronize(b.st));
    if (b.tmo_h[0]) { printf("TIMEOUT code 0x%x in the correctness run\n", b.tmo_h[0]); return 2; }
    // reference
    float *x = dmalloc<float>((size_t)64 * C), *y = dmalloc<float>((size_t)64 * C), *A = dmalloc<float>((size_t)64 * C), *T = dmalloc<float>((size_t)64 * 2 * C);
    float *gq = dmalloc<float>((size_t)64 * C), *gs = dmalloc<float>((size_t)64 * C);
    CK(hipMemcpy(x, b.X0, (size_t)64 * C * 4, hipMemcpyDeviceToDevice));
    const dim3 gc(C / 256, B), g2c(2 * C / 256, B);
    for (int j = 0; j < nblocks; ++j) {
        const XBlockW& k = b.hb[j];
        const float* f = b.film + k.film_off;
        ref_ln<<<B, 256>>>(x, f, f + C, A, 1e-6f);
        ref_gemm<<<g2c, 256>>>(A, k.w1, T, 2 * C);
        ref_q0<<<gc, 256>>>(T, k.b1, k.dw_w, k.dw_b, gq);
        ref_gemm<<<gc, 256>>>(gq, k.wsca, T, C);
        ref_q1<<<gc, 256>>>(T, k.bsca, gq, gs);
        ref_gemm<<<gc, 256>>>(gs, k.w3, T, C);
        ref_resid<<<gc, 256>>>(T, k.b3, k.beta, x, y);
        ref_ln<<<B, 256>>>(y, f + 2 * C, f + 3 * C, A, 1e-6f);
        ref_gemm<<<g2c, 256>>>(A, k.w4, T, 2 * C);
        ref_q3<<<gc, 256>>>(T, k.b4, gq);
        ref_gemm<<<gc, 256>>>(gq, k.w5, T, C);
        ref_resid<<<gc, 256>>>(T, k.b5, k.gamma, y, x);
    }
    CK(hipDeviceSynchronize());
    std::vector<float> hr((size_t)B * C), hs((size_t)B * C);
    std::vector<unsigned short> hb16((size_t)B * C);
    CK(hipMemcpy(hr.data(), x, hr.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hs.data(), b.p.X, hs.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hb16.data(), b.p.Xb, hb16.size() * 2, hipMemcpyDeviceToHost));
    double num = 0, den = 0, mx = 0, nb = 0; size_t nan = 0;
    for (size_t i = 0; i < hr.size(); ++i) {
        if (!std::isfinite(hs[i])) { ++nan; continue; }
        const double d = (double)hs[i] - hr[i]; num += d * d; den += (double)hr[i] * hr[i]; mx = std::max(mx, std::fabs(d));
        const unsigned xbits = (unsigned)hb16[i] << 16; float xb; memcpy(&xb, &xbits, 4); const double e = (double)xb - hs[i]; nb = std::max(nb, std::fabs(e) / (std::fabs((double)hs[i]) + 1e-3));
    }
    const double rel = std::sqrt(num / std::max(den, 1e-30));
    printf("check: %d blocks, B = %d, %s hand-off: x' rel-L2 vs naive reference %.3e, max abs %.3e (rms %.3f), non-finite %zu, bf16 copy max rel %.2e -> %s\n", nblocks, B,
           force_global ? "global" : "local", rel, mx, std::sqrt(den / hr.size()), nan, nb, (rel < 1e-3 * nblocks && nan == 0 && nb < 5e-3) ? "OK" : "MISMATCH");
    fflush(stdout);
    return (rel < 1e-3 * nblocks && nan == 0) ? 0 : 1;      // bf16 operand roundings decorrelate block by block: the bound grows with the depth
}

static void timing(int nblocks, int B, int reps, int force_global, int no_w) {
    Bench b = setup(nblocks, B);
    MStageP& p = b.p;
    p.force_global = force_global; p.dbg_no_w = no_w;
    const int P = 5 * nblocks;
    p.stamps = dmalloc<unsigned long long>((size_t)P * 256 * 8);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f, sum = 0.f;
    for (int r = 0; r < reps; ++r) {
        reset_x(b);
        CK(hipEventRecord(e0, b.st));
        CK(launch_mid_stage(p, b.st));
        CK(hipEventRecord(e1, b.st));
        CK(hipStreamSynchronize(b.st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms); if (r >= reps / 2) sum += ms;
        if (b.tmo_h[0]) { printf("TIMEOUT code 0x%x (launch %d, %.1f ms)\n", b.tmo_h[0], r, ms); fflush(stdout); exit(2); }
    }
    std::vector<unsigned long long> h((size_t)P * 256 * 8);
    CK(hipMemcpy(h.data(), p.stamps, h.size() * 8, hipMemcpyDeviceToHost));
    if (no_w) printf("WHAT-IF no weight DMA (timing only): ");
    printf("mid stage blocks=%d B=%d %s: kernel best %.1f us, mean of the later half %.1f us = %.2f us per block, %.2f us per phase (gate 5.5)\n", nblocks, B,
           force_global ? "global cluster exchange" : "local cluster exchange", best * 1e3, sum / (reps - reps / 2) * 1e3, best * 1e3 / nblocks, best * 1e3 / P);
    auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    const char* qn[5] = {"q0 conv1+dw", "q1 sca", "q2 conv3", "q3 conv4", "q4 conv5"};
    double acc[5][6] = {}, ln_acc[5][3] = {}, nfs[5] = {};
    int cnt[5] = {};
    for (int ph = 0; ph < P; ++ph) {
        std::vector<double> seg[5], lnseg[3], nf_store;
        unsigned long long lo = ~0ull, hi = 0;
        for (int w = 0; w < 256; ++w) {
            const unsigned long long* s = &h[((size_t)ph * 256 + w) * 8];
            if (!s[0] || !s[5]) continue;
            lo = std::min(lo, s[0]); hi = std::max(hi, s[5]);
            if (((w >> 3) & 3) != 0) { if (s[3] >= s[2]) nf_store.push_back((double)(s[3] - s[2]) * 0.01); continue; }      // wave 0 finishes its row block in member 0 only
            for (int k = 0; k < 5; ++k) seg[k].push_back((double)(s[k + 1] >= s[k] && s[k] ? s[k + 1] - s[k] : 0) * 0.01);
            if (s[6] && s[7] && s[1]) { lnseg[0].push_back((double)(s[7] - s[1]) * 0.01); lnseg[1].push_back((double)(s[6] - s[7]) * 0.01); lnseg[2].push_back((double)(s[2] - s[6]) * 0.01); }
        }
        if (seg[0].empty()) continue;
        const int q = ph % 5;
        if (ph >= 5 && ph < P - 5) {
            for (int k = 0; k < 5; ++k) acc[q][k] += med(seg[k]);
            acc[q][5] += (hi - lo) * 0.01; cnt[q]++; nfs[q] += nf_store.empty() ? 0. : med(nf_store);
            if (!lnseg[0].empty()) for (int k = 0; k < 3; ++k) ln_acc[q][k] += med(lnseg[k]);
        }
        if (ph < 10) printf("  phase %2d %-12s wait %5.2f  loads+kloop %5.2f  exchange %5.2f  epilogue %5.2f  publish %5.2f | span %5.2f us\n", ph, qn[q], med(seg[0]), med(seg[1]),
                            med(seg[2]), med(seg[3]), med(seg[4]), (hi - lo) * 0.01);
    }
    double tot = 0;
    for (int q = 0; q < 5; ++q)
        if (cnt[q]) {
            double s = 0; for (int k = 0; k < 5; ++k) s += acc[q][k] / cnt[q];
            tot += s;
            printf("  mean over inner blocks %-12s wait %5.2f  loads+kloop %5.2f  exchange %5.2f  epilogue %5.2f  publish %5.2f = %5.2f | span %5.2f us | other members: partial store + flag %4.2f\n", qn[q], acc[q][0] / cnt[q],
                   acc[q][1] / cnt[q], acc[q][2] / cnt[q], acc[q][3] / cnt[q], acc[q][4] / cnt[q], s, acc[q][5] / cnt[q], nfs[q] / cnt[q]);
        }
    printf("  sum of the five medians %.2f us per block = %.2f us per phase\n", tot, tot / 5);
    for (int q = 0; q < 5; q += 3)
        if (cnt[q]) printf("  LayerNorm phases %-12s flags -> statistics merged %5.2f  transform %5.2f  K loop %5.2f\n", qn[q], ln_acc[q][0] / cnt[q], ln_acc[q][1] / cnt[q], ln_acc[q][2] / cnt[q]);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 20;
    const int which = argc > 2 ? atoi(argv[2]) : 0;
    int bad = 0;
    if (which == 0 || which == 1) { bad |= check(1, 64, 0); bad |= check(2, 64, 0); bad |= check(2, 37, 1); bad |= check(8, 64, 0); }
    if (bad) { printf("correctness failed: not timing\n"); return 1; }
    if (which == 0 || which == 2) { timing(8, 64, reps, 0, 0); timing(8, 64, reps, 0, 1); }
    if (which == 3) timing(8, 64, reps, 1, 0);
    return 0;
}
