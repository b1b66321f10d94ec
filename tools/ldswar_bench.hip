// ldswar_bench.hip — is the data of a 128-bit LDS store safe from an LDS load that follows it and returns into the same
// registers? (tools only)      build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/ldswar_bench_bin tools/ldswar_bench.hip
// Each wave: v[40:43] = a pattern unique per lane; ds_write_b128 (row stride 144 B, as the GEMM staging tiles);
// immediately ds_read2_b64 v[40:43] from a table of 0xAAAAAAAA; wait; read the stored row back and compare.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int MODE>     // 0: store then load into the same registers; 1: VALU writes the last data register right before the store
__global__ __launch_bounds__(512) void war_kernel(unsigned* bad, unsigned* bad_lane_hist, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[8 * 4608 + 4096];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned* table = reinterpret_cast<unsigned*>(smem + 8 * 4608);
    for (int i = tid; i < 1024; i += 512) table[i] = 0xAAAAAAAAu;
    __syncthreads();
    const unsigned wbase = wave * 4608 + (lane >> 3) * 144 + (lane & 7) * 16;      // unit layout of the skinny GEMM
    const unsigned raddr = 8 * 4608 + (lane & 7) * 32;
    unsigned nbad = 0;
    for (int it = 0; it < iters; ++it) {
        const unsigned pat = (unsigned)(blockIdx.x * 131 + it * 7 + tid) * 2654435761u | 1u;
        unsigned r0, r1, r2, r3;
        if (MODE == 0)
            asm volatile("v_mov_b32 v40, %4\n\tv_add_u32 v41, 1, %4\n\tv_add_u32 v42, 2, %4\n\tv_add_u32 v43, 3, %4\n\t"
                         "s_nop 4\n\t"
                         "ds_write_b128 %5, v[40:43]\n\t"
                         "ds_read2_b64 v[40:43], %6 offset1:1\n\t"
                         "s_waitcnt lgkmcnt(0)\n\t"
                         "ds_read_b128 v[44:47], %5\n\t"
                         "s_waitcnt lgkmcnt(0)\n\t"
                         "v_mov_b32 %0, v44\n\tv_mov_b32 %1, v45\n\tv_mov_b32 %2, v46\n\tv_mov_b32 %3, v47"
                         : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(pat), "v"(wbase), "v"(raddr)
                         : "memory", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");
        else if (MODE == 2) {                                     // store, four loads, lgkmcnt(1), use the third load at once
            asm volatile("v_mov_b32 v40, %4\n\tv_add_u32 v41, 1, %4\n\tv_add_u32 v42, 2, %4\n\tv_add_u32 v43, 3, %4\n\t"
                         "v_mov_b32 v52, 0\n\tv_mov_b32 v53, 0\n\tv_mov_b32 v54, 0\n\tv_mov_b32 v55, 0\n\t"
                         "s_nop 4\n\t"
                         "ds_write_b128 %5, v[40:43]\n\t"
                         "ds_read2_b64 v[44:47], %6 offset1:1\n\t"
                         "ds_read2_b64 v[48:51], %6 offset0:2 offset1:3\n\t"
                         "ds_read2_b64 v[52:55], %6 offset0:4 offset1:5\n\t"
                         "ds_read2_b64 v[56:59], %6 offset0:6 offset1:7\n\t"
                         "s_waitcnt lgkmcnt(1)\n\t"
                         "v_mov_b32 %0, v52\n\tv_mov_b32 %1, v53\n\tv_mov_b32 %2, v54\n\tv_mov_b32 %3, v55\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(pat), "v"(wbase), "v"(raddr)
                         : "memory", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59");
            const bool ok2 = r0 == 0xAAAAAAAAu && r1 == 0xAAAAAAAAu && r2 == 0xAAAAAAAAu && r3 == 0xAAAAAAAAu;
            if (!ok2) { ++nbad; atomicAdd(&bad_lane_hist[lane], 1u); }
            continue;
        } else
            asm volatile("v_mov_b32 v40, %4\n\tv_add_u32 v41, 1, %4\n\tv_add_u32 v42, 2, %4\n\tv_mov_b32 v43, 0\n\t"
                         "s_nop 4\n\t"
                         "v_add_u32 v43, 3, %4\n\t"
                         "ds_write_b128 %5, v[40:43]\n\t"
                         "s_waitcnt lgkmcnt(0)\n\t"
                         "ds_read_b128 v[44:47], %5\n\t"
                         "s_waitcnt lgkmcnt(0)\n\t"
                         "v_mov_b32 %0, v44\n\tv_mov_b32 %1, v45\n\tv_mov_b32 %2, v46\n\tv_mov_b32 %3, v47"
                         : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(pat), "v"(wbase), "v"(raddr)
                         : "memory", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");
        const bool ok = r0 == pat && r1 == pat + 1 && r2 == pat + 2 && r3 == pat + 3;
        if (!ok) { ++nbad; atomicAdd(&bad_lane_hist[lane], 1u); }
    }
    if (nbad) atomicAdd(bad, nbad);
}


// mode 3: what the LayerNorm prologue does.  Lanes 0/16/32/48 of every wave write one 8-byte entry each (32 entries per
// workgroup, fresh values every iteration), s_waitcnt lgkmcnt(0), s_barrier; then every wave reads four entries with two
// ds_read2_b64 (8 lanes per entry, as the row statistics) plus four more broadcast reads, resumes on lgkmcnt(5) and uses the
// first read in the next instruction.
__global__ __launch_bounds__(512) void xwave_kernel(unsigned* bad, unsigned* bad_lane_hist, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char xsmem[];
    unsigned char* smem = xsmem;
    const int tid = threadIdx.x, lane = tid & 63;
    unsigned* table = reinterpret_cast<unsigned*>(smem + 70000 + 1024);
    for (int i = tid; i < 1024; i += 512) table[i] = 0xAAAAAAAAu;
    __syncthreads();
    const unsigned wr = 69632 + (unsigned)(tid >> 4) * 8;                 // entry tid / 16, written by lanes with (tid & 15) == 0
    const unsigned rd = 69632 + (unsigned)(lane >> 3) * 8;                // entries lane>>3 (+8, +16, +24)
    const unsigned tb = 70000 + 1024 + (lane & 7) * 32;
    unsigned nbad = 0;
    for (int it = 0; it < iters; ++it) {
        const unsigned base = (unsigned)(blockIdx.x * 977 + it) * 2654435761u;
        const unsigned mine0 = base + (unsigned)(tid >> 4) * 2u, mine1 = mine0 + 1u;
        unsigned a0, a1, b0, b1, c0, c1, d0, d1;
        asm volatile(
            "v_mov_b32 v40, %8\n\tv_mov_b32 v41, %9\n\t"
            "s_mov_b64 s[10:11], exec\n\t"
            "s_mov_b32 exec_lo, 0x00010001\n\t"
            "s_mov_b32 exec_hi, 0x00010001\n\t"
            "ds_write_b64 %10, v[40:41]\n\t"
            "s_mov_b64 exec, s[10:11]\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "s_barrier\n\t"
            "ds_read2_b64 v[42:45], %11 offset1:8\n\t"
            "ds_read2_b64 v[46:49], %11 offset0:16 offset1:24\n\t"
            "ds_read2_b64 v[50:53], %12 offset1:1\n\t"
            "ds_read2_b64 v[54:57], %12 offset0:2 offset1:3\n\t"
            "ds_read2_b64 v[58:61], %12 offset0:4 offset1:5\n\t"
            "ds_read2_b64 v[62:65], %12 offset0:6 offset1:7\n\t"
            "s_waitcnt lgkmcnt(5)\n\t"
            "v_mov_b32 %0, v42\n\tv_mov_b32 %1, v43\n\tv_mov_b32 %2, v44\n\tv_mov_b32 %3, v45\n\t"
            "s_waitcnt lgkmcnt(4)\n\t"
            "v_mov_b32 %4, v46\n\tv_mov_b32 %5, v47\n\tv_mov_b32 %6, v48\n\tv_mov_b32 %7, v49\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "s_barrier"
            : "=v"(a0), "=v"(a1), "=v"(b0), "=v"(b1), "=v"(c0), "=v"(c1), "=v"(d0), "=v"(d1)
            : "v"(mine0), "v"(mine1), "v"(wr), "v"(rd), "v"(tb)
            : "memory", "s10", "s11", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55",
              "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65");
        const unsigned e = (unsigned)(lane >> 3);
        const bool ok = a0 == base + e * 2u && a1 == base + e * 2u + 1u && b0 == base + (e + 8) * 2u && b1 == base + (e + 8) * 2u + 1u &&
                        c0 == base + (e + 16) * 2u && c1 == base + (e + 16) * 2u + 1u && d0 == base + (e + 24) * 2u && d1 == base + (e + 24) * 2u + 1u;
        if (!ok) { ++nbad; atomicAdd(&bad_lane_hist[lane], 1u); }
    }
    if (nbad) atomicAdd(bad, nbad);
}

// mode 4: the address register of LDS reads overwritten by the next VALU instruction (what the compiler emitted right after the
// reads of the row statistics: ds_read2_b64 .., v37 ..; ds_read2_b64 .., v37 ..; v_lshrrev_b32 v37, ..).
template <int GAP>
__global__ __launch_bounds__(512) void addr_war_kernel(unsigned* bad, unsigned* bad_lane_hist, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char xsmem[];
    const int tid = threadIdx.x, lane = tid & 63;
    unsigned* ent = reinterpret_cast<unsigned*>(xsmem + 69632);
    for (int i = tid; i < 64; i += 512) ent[i] = 0x1000u + (unsigned)i;      // entry e = (0x1000 + 2e, 0x1001 + 2e)
    unsigned* table = reinterpret_cast<unsigned*>(xsmem + 70000 + 1024);
    for (int i = tid; i < 1024; i += 512) table[i] = 0xAAAAAAAAu;
    __syncthreads();
    unsigned nbad = 0;
    for (int it = 0; it < iters; ++it) {
        unsigned rd = 69632 + (unsigned)(lane >> 3) * 8;
        const unsigned tb = 70000 + 1024 + (lane & 7) * 32;
        unsigned a0, a1, b0, b1, c0, c1, d0, d1;
        asm volatile(
            "s_barrier\n\t"
            "ds_read2_b64 v[42:45], %8 offset1:8\n\t"
            "ds_read2_b64 v[46:49], %8 offset0:16 offset1:24\n\t"
            ".rept %c10\n\ts_nop 0\n\t.endr\n\t"
            "v_lshrrev_b32 %8, 5, %8\n\t"                         // the address register is dead for the program: reuse it
            "ds_read2_b64 v[50:53], %9 offset1:1\n\t"
            "ds_read2_b64 v[54:57], %9 offset0:2 offset1:3\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_mov_b32 %0, v42\n\tv_mov_b32 %1, v43\n\tv_mov_b32 %2, v44\n\tv_mov_b32 %3, v45\n\t"
            "v_mov_b32 %4, v46\n\tv_mov_b32 %5, v47\n\tv_mov_b32 %6, v48\n\tv_mov_b32 %7, v49"
            : "=v"(a0), "=v"(a1), "=v"(b0), "=v"(b1), "=v"(c0), "=v"(c1), "=v"(d0), "=v"(d1), "+v"(rd)
            : "v"(tb), "n"(GAP)
            : "memory", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57");
        const unsigned e = (unsigned)(lane >> 3);
        const bool ok = a0 == 0x1000u + 2 * e && a1 == 0x1001u + 2 * e && b0 == 0x1000u + 2 * (e + 8) && b1 == 0x1001u + 2 * (e + 8) &&
                        c0 == 0x1000u + 2 * (e + 16) && c1 == 0x1001u + 2 * (e + 16) && d0 == 0x1000u + 2 * (e + 24) && d1 == 0x1001u + 2 * (e + 24);
        if (!ok) { ++nbad; atomicAdd(&bad_lane_hist[lane], 1u); }
    }
    if (nbad) atomicAdd(bad, nbad);
}

int main() {
    unsigned *bad, *hist;
    CK(hipMalloc(&bad, 4)); CK(hipMalloc(&hist, 256));
    for (int mode = 0; mode < 3; ++mode) {
        CK(hipMemset(bad, 0, 4)); CK(hipMemset(hist, 0, 256));
        if (mode == 0) hipLaunchKernelGGL(war_kernel<0>, dim3(1024), dim3(512), 0, 0, bad, hist, 2000);
        else if (mode == 2) hipLaunchKernelGGL(war_kernel<2>, dim3(1024), dim3(512), 0, 0, bad, hist, 2000);
        else hipLaunchKernelGGL(war_kernel<1>, dim3(1024), dim3(512), 0, 0, bad, hist, 2000);
        CK(hipDeviceSynchronize());
        unsigned h, hl[64]; CK(hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hl, hist, 256, hipMemcpyDeviceToHost));
        printf("mode %d: %u corrupted stores of %u", mode, h, 1024u * 512u * 2000u);
        if (h) { printf("; by lane:"); for (int l = 0; l < 64; ++l) if (hl[l]) printf(" %d:%u", l, hl[l]); }
        printf("\n");
    }
    {
        CK(hipMemset(bad, 0, 4)); CK(hipMemset(hist, 0, 256));
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&xwave_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        hipLaunchKernelGGL(xwave_kernel, dim3(1024), dim3(512), 86272, 0, bad, hist, 4000);
        CK(hipDeviceSynchronize());
        unsigned h, hl[64]; CK(hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hl, hist, 256, hipMemcpyDeviceToHost));
        printf("mode 3 (cross-wave write, barrier, ds_read2_b64, lgkmcnt(5), use): %u bad reads of %u", h, 1024u * 512u * 4000u);
        if (h) { printf("; by lane:"); for (int l = 0; l < 64; ++l) if (hl[l]) printf(" %d:%u", l, hl[l]); }
        printf("\n");
    }
    for (int gap = 0; gap < 2; ++gap) {
        CK(hipMemset(bad, 0, 4)); CK(hipMemset(hist, 0, 256));
        if (gap == 0) { CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&addr_war_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                        hipLaunchKernelGGL(addr_war_kernel<0>, dim3(1024), dim3(512), 86272, 0, bad, hist, 4000); }
        else { CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&addr_war_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
               hipLaunchKernelGGL(addr_war_kernel<4>, dim3(1024), dim3(512), 86272, 0, bad, hist, 4000); }
        CK(hipDeviceSynchronize());
        unsigned h, hl[64]; CK(hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hl, hist, 256, hipMemcpyDeviceToHost));
        printf("mode 4 (address VGPR of two ds_read2_b64 overwritten by the next VALU instruction%s): %u bad reads of %u", gap ? ", 4 idle cycles between" : "", h, 1024u * 512u * 4000u);
        if (h) { printf("; by lane:"); for (int l = 0; l < 64; ++l) if (hl[l]) printf(" %d:%u", l, hl[l]); }
        printf("\n");
    }
    return 0;
}
