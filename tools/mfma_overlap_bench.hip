// mfma_overlap_bench.hip — may the destination of v_mfma_f32_32x32x16_bf16 overlap its own A / B registers when C is the
// inline constant 0?  (the compiler emitted exactly that: D = v[2:17], A = v[10:13], B = v[6:9])          (tools only)
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/mfma_overlap_bench_bin tools/mfma_overlap_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k(unsigned* bad, int iters) {
    unsigned nbad = 0;
    for (int it = 0; it < iters; ++it) {
        unsigned s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + it * 40503u;
        unsigned a[4], b[4];
        for (int i = 0; i < 4; ++i) {                      // two bf16 in [-2, 2) per dword
            s = s * 1664525u + 1013904223u; a[i] = (0x3f003f00u + ((s >> 8) & 0x007f007fu)) ^ ((s & 1u) << 15) ^ ((s & 2u) << 30);
            s = s * 1664525u + 1013904223u; b[i] = (0x3f003f00u + ((s >> 8) & 0x007f007fu)) ^ ((s & 1u) << 15) ^ ((s & 2u) << 30);
        }
        unsigned diff;
        asm volatile(
            "v_mov_b32 v10, %1\n\tv_mov_b32 v11, %2\n\tv_mov_b32 v12, %3\n\tv_mov_b32 v13, %4\n\t"
            "v_mov_b32 v6, %5\n\tv_mov_b32 v7, %6\n\tv_mov_b32 v8, %7\n\tv_mov_b32 v9, %8\n\t"
            "s_nop 4\n\t"
            "v_mfma_f32_32x32x16_bf16 v[20:35], v[10:13], v[6:9], 0\n\t"      // reference: D apart from A and B
            "v_mfma_f32_32x32x16_bf16 v[2:17], v[10:13], v[6:9], 0\n\t"       // D on top of A and B
            "s_nop 15\n\ts_nop 15\n\t"
            "v_mov_b32 %0, 0\n\t"
            "v_xor_b32 v36, v2, v20\n\tv_or_b32 %0, %0, v36\n\t"  "v_xor_b32 v36, v3, v21\n\tv_or_b32 %0, %0, v36\n\t"
            "v_xor_b32 v36, v4, v22\n\tv_or_b32 %0, %0, v36\n\t"  "v_xor_b32 v36, v5, v23\n\tv_or_b32 %0, %0, v36\n\t"
            "v_xor_b32 v36, v6, v24\n\tv_or_b32 %0, %0, v36\n\t"  "v_xor_b32 v36, v7, v25\n\tv_or_b32 %0, %0, v36\n\t"
            "v_xor_b32 v36, v8, v26\n\tv_or_b32 %0, %0, v36\n\t"  "v_xor_b32 v36, v9, v27\n\tv_or_b32 %0, %0, v36\n\t"
            "v_xor_b32 v36, v10, v28\n\tv_or_b32 %0, %0, v36\n\t" "v_xor_b32 v36, v11, v29\n\tv_or_b32 %0, %0, v36\n\t"
            "v_xor_b32 v36, v12, v30\n\tv_or_b32 %0, %0, v36\n\t" "v_xor_b32 v36, v13, v31\n\tv_or_b32 %0, %0, v36\n\t"
            "v_xor_b32 v36, v14, v32\n\tv_or_b32 %0, %0, v36\n\t" "v_xor_b32 v36, v15, v33\n\tv_or_b32 %0, %0, v36\n\t"
            "v_xor_b32 v36, v16, v34\n\tv_or_b32 %0, %0, v36\n\t" "v_xor_b32 v36, v17, v35\n\tv_or_b32 %0, %0, v36"
            : "=&v"(diff)
            : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3])
            : "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17",
              "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36");
        nbad += diff != 0;
    }
    if (nbad) atomicAdd(bad, nbad);
}

int main() {
    unsigned* bad; CK(hipMalloc(&bad, 4)); CK(hipMemset(bad, 0, 4));
    hipLaunchKernelGGL(k, dim3(2048), dim3(256), 0, 0, bad, 500);
    CK(hipDeviceSynchronize());
    unsigned h; CK(hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost));
    printf("D overlapping A and B, C = 0: %u of %u lane results differ from the non-overlapping form\n", h, 2048u * 256u * 500u);
    return 0;
}
