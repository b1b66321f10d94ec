// pkfma_bench.hip — v_pk_fma_f32 with op_sel picking the HIGH register of a pair for both halves, on a pair that an LDS read
// has just delivered, next to a multiply that reads the same pair (the instruction mix of the LayerNorm transform's first use
// of the row statistics).  Compared with plain v_fma_f32 on copies.  (tools only)
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/pkfma_bench_bin tools/pkfma_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(512) void k(unsigned* bad, unsigned* hist, int iters) {
    __shared__ __attribute__((aligned(16))) float2 stats[64];
    __shared__ __attribute__((aligned(16))) float tab[1024];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 1024; i += 512) tab[i] = 1.0f + (float)(i & 15) * 0.0625f;
    unsigned nbad = 0;
    for (int it = 0; it < iters; ++it) {
        __syncthreads();
        if ((tid & 15) == 0) stats[tid >> 4] = make_float2(0.001f * (float)((tid >> 4) + it % 97), 1.0f + 0.01f * (float)((tid >> 4) + it % 89));
        const float xlo = 0.5f + 0.001f * (float)((lane * 7 + it) & 255), xhi = -0.25f + 0.002f * (float)((lane * 3 + it) & 127);
        const unsigned rd = (unsigned)(size_t)(&stats[0]) + (unsigned)(lane >> 3) * 8;   // LDS byte address (low 32 bits of the generic pointer)
        const unsigned tb = (unsigned)(size_t)(&tab[0]) + (lane & 7) * 32;
        float ylo, yhi, mean, rstd;
        asm volatile(
            "s_waitcnt lgkmcnt(0)\n\t"
            "s_barrier\n\t"
            "ds_read2_b64 v[138:141], %4 offset1:8\n\t"
            "ds_read2_b64 v[130:133], %4 offset0:16 offset1:24\n\t"
            "ds_read2_b64 v[168:171], %5 offset1:1\n\t"
            "ds_read2_b64 v[172:175], %5 offset0:2 offset1:3\n\t"
            "ds_read2_b64 v[186:189], %5 offset0:4 offset1:5\n\t"
            "ds_read2_b64 v[190:193], %5 offset0:6 offset1:7\n\t"
            "s_waitcnt lgkmcnt(5)\n\t"
            "v_mul_f32_e64 v148, v139, -v138\n\t"
            "v_mov_b32 v194, %6\n\t"
            "v_mov_b32 v195, %7\n\t"
            "v_pk_fma_f32 v[194:195], v[194:195], v[138:139], v[148:149] op_sel:[0,1,0] op_sel_hi:[1,1,0]\n\t"
            "s_nop 4\n\t"
            "v_mov_b32 %0, v194\n\tv_mov_b32 %1, v195\n\tv_mov_b32 %2, v138\n\tv_mov_b32 %3, v139\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "=v"(ylo), "=v"(yhi), "=v"(mean), "=v"(rstd)
            : "v"(rd), "v"(tb), "v"(xlo), "v"(xhi)
            : "memory", "v130", "v131", "v132", "v133", "v138", "v139", "v140", "v141", "v148", "v149", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175",
              "v186", "v187", "v188", "v189", "v190", "v191", "v192", "v193", "v194", "v195");
        const float2 s = stats[lane >> 3];
        const float mu = -s.x * s.y;
        const float rlo = fmaf(xlo, s.y, mu), rhi = fmaf(xhi, s.y, mu);
        const bool ok = ylo == rlo && yhi == rhi && mean == s.x && rstd == s.y;
        if (!ok) { ++nbad; atomicAdd(&hist[lane], 1u); }
    }
    if (nbad) atomicAdd(bad, nbad);
}

int main() {
    unsigned *bad, *hist; CK(hipMalloc(&bad, 4)); CK(hipMalloc(&hist, 256)); CK(hipMemset(bad, 0, 4)); CK(hipMemset(hist, 0, 256));
    hipLaunchKernelGGL(k, dim3(1024), dim3(512), 0, 0, bad, hist, 4000);
    CK(hipDeviceSynchronize());
    unsigned h, hl[64]; CK(hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hl, hist, 256, hipMemcpyDeviceToHost));
    printf("v_pk_fma_f32 op_sel:[0,1,0] on a pair fresh from LDS: %u mismatches of %u", h, 1024u * 512u * 4000u);
    if (h) { printf("; by lane:"); for (int l = 0; l < 64; ++l) if (hl[l]) printf(" %d:%u", l, hl[l]); }
    printf("\n");
    return 0;
}
