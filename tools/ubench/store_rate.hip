// Store-path probe for the GEMM epilogue: every wave writes its 128 x 64 bf16 block of a [M, N] matrix as 16 instructions of
// 16 bytes per lane (8 lanes = one 128-byte line of a row, 8 rows per instruction: the gemm8p.hip epilogue's pattern), 9 tiles per
// workgroup like the K = 768 forward GEMM, with different instruction forms.  Prints ms, TB/s and B/clk/CU at the measured clock.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/store_rate tools/ubench/store_rate.hip && /tmp/store_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

template <int FORM>
__device__ __forceinline__ void st(char* base, uint32_t off, u32x4 v) {
    if constexpr (FORM == 0) asm volatile("global_store_dwordx4 %0, %1, %2" ::"v"(off), "v"(v), "s"(base) : "memory");
    else if constexpr (FORM == 1) asm volatile("global_store_dwordx4 %0, %1, %2 nt" ::"v"(off), "v"(v), "s"(base) : "memory");
    else if constexpr (FORM == 2) asm volatile("global_store_dwordx4 %0, %1, %2 sc0 sc1" ::"v"(off), "v"(v), "s"(base) : "memory");
    else if constexpr (FORM == 3) asm volatile("global_store_dwordx4 %0, %1, %2 sc1" ::"v"(off), "v"(v), "s"(base) : "memory");
    else if constexpr (FORM == 4) asm volatile("global_store_dwordx4 %0, %1, %2 sc0 sc1 nt" ::"v"(off), "v"(v), "s"(base) : "memory");
}

// PATTERN 0: epilogue rows (8 lanes per 128-byte line, 8 rows of the matrix per instruction); 1: 1 KiB contiguous per instruction
template <int FORM, int PATTERN>
__global__ __launch_bounds__(512, 2) void store_kernel(char* C, int64_t ldcb, int tiles_n, int total, int gap) {
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wid >> 2, wc = wid & 3;
    u32x4 v = {(uint32_t)lane, 1u, 2u, 3u};
    for (int t = blockIdx.x; t < total; t += gridDim.x) {
        const int64_t m0 = (int64_t)(t / tiles_n) * 256 + wr * 128, n0 = (int64_t)(t % tiles_n) * 256 + wc * 64;
        char* base = C + m0 * ldcb + n0 * 2;
        for (int g = 0; g < gap; ++g) __builtin_amdgcn_s_sleep(64);  // stands in for the k-loop (~64 x 64 clocks per unit)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            uint32_t off;
            if (PATTERN == 0) off = (uint32_t)(i * 8 + (lane >> 3)) * (uint32_t)ldcb + (lane & 7) * 16;
            else off = (uint32_t)((i * 8 + (lane >> 3)) * 128 + (lane & 7) * 16) ;  // same bytes per wave, packed: wave block = 16 KiB contiguous
            char* b = PATTERN == 0 ? base : C + ((int64_t)t * 8 + wid) * 16384;
            st<FORM>(b, off, v);
        }
    }
}

template <int FORM, int PATTERN>
static int run(const char* name, char* C, int64_t M, int64_t N, int grid, int gap, hipEvent_t e0, hipEvent_t e1, double clk_ghz, int cus) {
    const int tiles_n = (int)(N / 256), total = (int)(M / 256) * tiles_n;
    std::vector<float> ms;
    for (int r = 0; r < 7; ++r) {
        CK(hipEventRecord(e0));
        store_kernel<FORM, PATTERN><<<grid, 512>>>(C, N * 2, tiles_n, total, gap);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float m;
        CK(hipEventElapsedTime(&m, e0, e1));
        ms.push_back(m);
    }
    std::sort(ms.begin(), ms.end());
    const double bytes = (double)M * N * 2;
    const int active = std::min(grid, cus * 2) / 2 > 0 ? std::min((grid + 1) / 2, cus) : 1;
    printf("%-44s grid %4d gap %2d: %7.3f ms  %6.2f TB/s  %5.1f B/clk/CU (%d CUs)\n", name, grid, gap, ms[3], bytes / ms[3] / 1e9, bytes / (ms[3] * 1e-3) / (clk_ghz * 1e9) / active, active);
    fflush(stdout);
    return 0;
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double clk = prop.clockRate / 1e6;
    printf("%s: %d CUs, %.2f GHz nominal\n", prop.name, cus, clk);
    const int64_t M = 131072, N = 2304;
    char* C;
    CK(hipMalloc(&C, M * N * 2 + (1 << 20)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int gap : {0, 4}) {
        run<0, 0>("plain, epilogue rows", C, M, N, 2 * cus, gap, e0, e1, clk, cus);
        run<1, 0>("nt, epilogue rows", C, M, N, 2 * cus, gap, e0, e1, clk, cus);
        run<2, 0>("sc0 sc1, epilogue rows", C, M, N, 2 * cus, gap, e0, e1, clk, cus);
        run<3, 0>("sc1, epilogue rows", C, M, N, 2 * cus, gap, e0, e1, clk, cus);
        run<4, 0>("sc0 sc1 nt, epilogue rows", C, M, N, 2 * cus, gap, e0, e1, clk, cus);
        run<0, 1>("plain, packed 16 KiB per wave", C, M, N, 2 * cus, gap, e0, e1, clk, cus);
        run<1, 1>("nt, packed 16 KiB per wave", C, M, N, 2 * cus, gap, e0, e1, clk, cus);
    }
    // a few CUs only: the per-CU limit without chip-level contention (M shrunk so that the run stays short)
    for (int grid : {2, 16, 64}) {
        run<0, 0>("plain, epilogue rows, few workgroups", C, M / 16, N, grid, 0, e0, e1, clk, cus);
        run<1, 0>("nt, epilogue rows, few workgroups", C, M / 16, N, grid, 0, e0, e1, clk, cus);
    }
    return 0;
}
