// vmorder_bench.hip — do vector loads return in issue order? (tools only)
// One load that misses everything (a fresh line of a large buffer), then one that hits (a line this wave has just read),
// then s_waitcnt vmcnt(1): if loads return in order the first register holds the loaded value, never the sentinel.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/vmorder_bench_bin tools/vmorder_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int MODE>   // 0: dword cold, dword hot; 1: dword cold (nt), dwordx4 hot; 2: dword cold, 3 x dwordx4 hot, vmcnt(3); 3 / 4: the hot load with EXEC = 0 / 0xf; 5 / 6: dwordx4 cold, first and last dword checked
__global__ __launch_bounds__(512) void order_kernel(const unsigned* cold, size_t cold_words, const unsigned* hot, unsigned* bad, int iters, unsigned salt) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const size_t nwaves = (size_t)gridDim.x * (blockDim.x >> 6);
    unsigned nbad = 0;
    // warm the hot line
    unsigned h0 = hot[lane];
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(h0));
    for (int it = 0; it < iters; ++it) {
        // a different cold line per wave and iteration, 16 B per lane, spread over the whole buffer
        size_t idx = ((wave + (size_t)it * nwaves) * 2654435761ull + salt) % (cold_words / 256);
        const unsigned* c = cold + idx * 256 + lane * 4;
        const unsigned* h = hot + lane * 4;
        if (MODE == 0) {
            unsigned x = 0xdeadbeefu, y = 0, got;
            asm volatile("global_load_dword %0, %3, off\n\t"
                         "global_load_dword %1, %4, off\n\t"
                         "s_waitcnt vmcnt(1)\n\t"
                         "v_mov_b32 %2, %0\n\t"
                         "s_waitcnt vmcnt(0)"
                         : "+v"(x), "+v"(y), "=v"(got) : "v"(c), "v"(h) : "memory");
            nbad += (got == 0xdeadbeefu) ? 1u : 0u;
            nbad += (got != x) ? 1u : 0u;
        } else {
            typedef unsigned u4 __attribute__((ext_vector_type(4)));
            unsigned x = 0xdeadbeefu, got;
            u4 y = {0, 0, 0, 0}, y2 = {0, 0, 0, 0}, y3 = {0, 0, 0, 0};
            if (MODE == 5 || MODE == 6) {                         // 16 bytes per lane (four passes through the return path): last dword
                unsigned g0, g3;
                if (MODE == 5)
                    asm volatile("v_mov_b32 v40, %4\n\tv_mov_b32 v41, %4\n\tv_mov_b32 v42, %4\n\tv_mov_b32 v43, %4\n\t"
                                 "global_load_dwordx4 v[40:43], %2, off\n\t"
                                 "global_load_dwordx4 v[44:47], %3, off\n\t"
                                 "s_waitcnt vmcnt(1)\n\t"
                                 "v_mov_b32 %0, v40\n\t"
                                 "v_mov_b32 %1, v43\n\t"
                                 "s_waitcnt vmcnt(0)"
                                 : "=v"(g0), "=v"(g3) : "v"(c), "v"(h), "v"(0xdeadbeefu) : "memory", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");
                else
                    asm volatile("v_mov_b32 v40, %4\n\tv_mov_b32 v41, %4\n\tv_mov_b32 v42, %4\n\tv_mov_b32 v43, %4\n\t"
                                 "global_load_dwordx4 v[40:43], %2, off nt\n\t"
                                 "global_load_dwordx4 v[44:47], %3, off\n\t"
                                 "global_load_dwordx4 v[48:51], %3, off offset:1024\n\t"
                                 "global_load_dwordx4 v[52:55], %3, off offset:2048\n\t"
                                 "s_waitcnt vmcnt(3)\n\t"
                                 "v_mov_b32 %0, v40\n\t"
                                 "v_mov_b32 %1, v43\n\t"
                                 "s_waitcnt vmcnt(0)"
                                 : "=v"(g0), "=v"(g3) : "v"(c), "v"(h), "v"(0xdeadbeefu) : "memory", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47",
                                   "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55");
                nbad += (g0 == 0xdeadbeefu) ? 1u : 0u;
                nbad += (g3 == 0xdeadbeefu) ? 1u : 0u;
                continue;
            }
            if (MODE == 3)                                        // the younger load runs with EXEC = 0 (what a predicated load
                asm volatile("global_load_dword %0, %3, off\n\t"     // looks like in a wave where no lane takes it)
                             "s_mov_b64 s[10:11], exec\n\t"
                             "s_mov_b64 exec, 0\n\t"
                             "global_load_dwordx4 %1, %4, off\n\t"
                             "s_mov_b64 exec, s[10:11]\n\t"
                             "s_waitcnt vmcnt(1)\n\t"
                             "v_mov_b32 %2, %0\n\t"
                             "s_waitcnt vmcnt(0)"
                             : "+v"(x), "+v"(y), "=v"(got) : "v"(c), "v"(h) : "memory", "s10", "s11");
            else if (MODE == 4)                                   // ... or with only some lanes active
                asm volatile("global_load_dword %0, %3, off\n\t"
                             "s_mov_b64 s[10:11], exec\n\t"
                             "s_mov_b64 exec, 0xf\n\t"
                             "global_load_dwordx4 %1, %4, off\n\t"
                             "s_mov_b64 exec, s[10:11]\n\t"
                             "s_waitcnt vmcnt(1)\n\t"
                             "v_mov_b32 %2, %0\n\t"
                             "s_waitcnt vmcnt(0)"
                             : "+v"(x), "+v"(y), "=v"(got) : "v"(c), "v"(h) : "memory", "s10", "s11");
            else if (MODE == 1)
                asm volatile("global_load_dword %0, %3, off nt\n\t"
                             "global_load_dwordx4 %1, %4, off\n\t"
                             "s_waitcnt vmcnt(1)\n\t"
                             "v_mov_b32 %2, %0\n\t"
                             "s_waitcnt vmcnt(0)"
                             : "+v"(x), "+v"(y), "=v"(got) : "v"(c), "v"(h) : "memory");
            else
                asm volatile("global_load_dword %0, %5, off\n\t"
                             "global_load_dwordx4 %1, %6, off\n\t"
                             "global_load_dwordx4 %2, %6, off offset:1024\n\t"
                             "global_load_dwordx4 %3, %6, off offset:2048\n\t"
                             "s_waitcnt vmcnt(3)\n\t"
                             "v_mov_b32 %4, %0\n\t"
                             "s_waitcnt vmcnt(0)"
                             : "+v"(x), "+v"(y), "+v"(y2), "+v"(y3), "=v"(got) : "v"(c), "v"(h) : "memory");
            nbad += (got == 0xdeadbeefu) ? 1u : 0u;
            nbad += (got != x) ? 1u : 0u;
        }
    }
    if (nbad) atomicAdd(bad, nbad);
}

int main() {
    const size_t cold_bytes = 3ull << 30;
    unsigned *cold, *hot, *bad;
    CK(hipMalloc(&cold, cold_bytes)); CK(hipMalloc(&hot, 1 << 16)); CK(hipMalloc(&bad, 4));
    CK(hipMemset(cold, 0x11, cold_bytes)); CK(hipMemset(hot, 0x22, 1 << 16));
    for (int mode = 0; mode < 7; ++mode) {
        CK(hipMemset(bad, 0, 4));
        for (int rep = 0; rep < 20; ++rep) {
            if (mode == 0) hipLaunchKernelGGL(order_kernel<0>, dim3(256), dim3(512), 0, 0, cold, cold_bytes / 4, hot, bad, 200, 7919u * rep);
            if (mode == 1) hipLaunchKernelGGL(order_kernel<1>, dim3(256), dim3(512), 0, 0, cold, cold_bytes / 4, hot, bad, 200, 7919u * rep);
            if (mode == 5) hipLaunchKernelGGL(order_kernel<5>, dim3(256), dim3(512), 0, 0, cold, cold_bytes / 4, hot, bad, 200, 7919u * rep);
            if (mode == 6) hipLaunchKernelGGL(order_kernel<6>, dim3(256), dim3(512), 0, 0, cold, cold_bytes / 4, hot, bad, 200, 7919u * rep);
            if (mode == 3) hipLaunchKernelGGL(order_kernel<3>, dim3(256), dim3(512), 0, 0, cold, cold_bytes / 4, hot, bad, 200, 7919u * rep);
            if (mode == 4) hipLaunchKernelGGL(order_kernel<4>, dim3(256), dim3(512), 0, 0, cold, cold_bytes / 4, hot, bad, 200, 7919u * rep);
            if (mode == 2) hipLaunchKernelGGL(order_kernel<2>, dim3(256), dim3(512), 0, 0, cold, cold_bytes / 4, hot, bad, 200, 7919u * rep);
        }
        CK(hipDeviceSynchronize());
        unsigned h; CK(hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost));
        printf("mode %d: %u stale reads after s_waitcnt vmcnt(n) out of %d\n", mode, h, 20 * 200 * 256 * 8 * 64);
    }
    return 0;
}
