// Micro-benchmark: what does issuing global_load_lds_dwordx4 (LDS-DMA, 1 KiB per wave instruction) cost, per wave and per CU?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_dma tools/ubench/lds_dma_issue.hip && /tmp/lds_dma
// One workgroup of W waves per CU; every wave issues `per` DMA instructions per round into its own LDS region from an L2-resident
// source (64 KiB per workgroup, re-read every round), waits for them (vmcnt(0)) and repeats.  Reported: cycles per instruction per
// wave (s_memtime around the issue sequence only, and around issue + wait) and bytes per clock per CU.
// Modes: 0 = s_mov m0 + s_nop before every instruction (what gemm256.hip does); 1 = m0 set once per 4 instructions, the LDS
// address advanced through the instruction offset (the scalar base is pre-decremented by the same amount); 2 = mode 0 with the
// scalar-base (saddr) form instead of a 64-bit VGPR address.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__device__ __forceinline__ void dma_v(const void* g, uint32_t m0v) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(m0v) : "memory", "m0");
}
__device__ __forceinline__ void dma_s(uint32_t voff, const char* sb, uint32_t m0v) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sb), "s"(m0v) : "memory", "m0");
}

template <int MODE, int PER>
__global__ __launch_bounds__(512) void k(const char* src, int rounds, unsigned long long* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;
    const char* base = src + (size_t)blockIdx.x * 65536;
    char* my = smem + wid * PER * 1024;
    const uint32_t m0b = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)my;
    unsigned long long t_issue = 0, t_all = 0;
    for (int r = 0; r < rounds; ++r) {
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        if constexpr (MODE == 0) {
#pragma unroll
            for (int i = 0; i < PER; ++i) dma_v(base + ((wid * PER + i) * 1024 + lane * 16) % 65536, m0b + i * 1024);
        } else if constexpr (MODE == 2) {
#pragma unroll
            for (int i = 0; i < PER; ++i) dma_s((uint32_t)(lane * 16), base + ((wid * PER + i) * 1024) % 65536, m0b + i * 1024);
        } else {
#pragma unroll
            for (int i = 0; i < PER; i += 4) {
                const uint32_t m0v = m0b + i * 1024;
                const char* sb = base + ((wid * PER + i) * 1024) % 65536;
                const uint32_t vo = (uint32_t)(lane * 16);
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\t"
                             "global_load_lds_dwordx4 %0, %1\n\t"
                             "global_load_lds_dwordx4 %0, %1 offset:1024\n\t"
                             "global_load_lds_dwordx4 %0, %1 offset:2048\n\t"
                             "global_load_lds_dwordx4 %0, %1 offset:3072" ::"v"(vo), "s"(sb), "s"(m0v) : "memory", "m0");
            }
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t2 = __builtin_amdgcn_s_memtime();
        t_issue += t1 - t0;
        t_all += t2 - t0;
        __builtin_amdgcn_s_barrier();
    }
    if (lane == 0) {
        atomicAdd(&out[0], t_issue);
        atomicAdd(&out[1], t_all);
        atomicAdd(&out[2], 1ull);
    }
    (void)nw;
}

template <int MODE, int PER>
void run(const char* src, unsigned long long* out, int waves, const char* name) {
    const int rounds = 2000;
    hipMemset(out, 0, 64);
    hipFuncSetAttribute((const void*)k<MODE, PER>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<MODE, PER><<<256, 64 * waves, waves * PER * 1024>>>(src, 10, out);
    hipMemset(out, 0, 64);
    hipEventRecord(e0);
    k<MODE, PER><<<256, 64 * waves, waves * PER * 1024>>>(src, rounds, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[3];
    hipMemcpy(h, out, 24, hipMemcpyDeviceToHost);
    const double n = (double)h[2] * rounds * PER;  // instructions issued in total (h[2] waves)
    // s_memtime ticks at 100 MHz on this part: report wall-clock derived numbers too
    const double bytes = (double)256 * waves * PER * 1024.0 * rounds;
    printf("%-28s waves/CU %d per-round %2d: %7.1f GB/s/CU  %6.2f TB/s chip  memtime ticks per instr: issue %.3f, issue+wait %.3f\n", name, waves, PER,
           bytes / 256 / (ms * 1e-3) / 1e9, bytes / (ms * 1e-3) / 1e12, h[0] / n, h[1] / n);
}

int main() {
    char* src;
    unsigned long long* out;
    hipMalloc(&src, 256 * 65536);
    hipMemset(src, 1, 256 * 65536);
    hipMalloc(&out, 64);
    for (int waves : {1, 2, 4, 8}) {
        run<0, 8>(src, out, waves, "vaddr, m0 per instr");
        run<2, 8>(src, out, waves, "saddr, m0 per instr");
        run<1, 8>(src, out, waves, "saddr, m0 per 4 (offset)");
    }
    run<0, 16>(src, out, 4, "vaddr, m0 per instr");
    run<1, 16>(src, out, 4, "saddr, m0 per 4 (offset)");
    run<1, 16>(src, out, 8, "saddr, m0 per 4 (offset)");
    return 0;
}
