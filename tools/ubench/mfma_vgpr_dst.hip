// Micro-benchmark (r05): does an MFMA whose accumulator lives in VGPRs (the score products of the attention kernels: the VALU must read
// them) slow the vector instructions issued beside it, compared with an AGPR accumulator?  One wave per SIMD, 16 x v_mfma_f32_32x32x16_bf16
// per trip with F independent v_fma_f32 in every MFMA gap, operands in registers (no LDS, no memory).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_vgpr_dst tools/ubench/mfma_vgpr_dst.hip && /tmp/mfma_vgpr_dst
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int VDST, int F, int EXPS, int CH = 1, int ASRC = 0>
__global__ __launch_bounds__(256, 1) void k(float* sink, unsigned long long* clk, int iters, float seed) {
    f32x16 acc[4];
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed * (i + 1)); b[i] = (__bf16)(seed * (7 - i)); }
    for (int q = 0; q < 4; ++q) for (int i = 0; i < 16; ++i) acc[q][i] = 0.f;
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = seed * i;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            if constexpr (VDST && ASRC) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[(m / CH) & 3]) : "a"(a), "a"(b));
            else if constexpr (VDST) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[(m / CH) & 3]) : "v"(a), "v"(b));
            else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[(m / CH) & 3]) : "v"(a), "v"(b));
#pragma unroll
            for (int f = 0; f < F; ++f) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[f & 7]) : "v"(seed));
#pragma unroll
            for (int f = 0; f < EXPS; ++f) asm volatile("v_exp_f32 %0, %0" : "+v"(x[(f + 4) & 7]));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0.f;
    for (int q = 0; q < 4; ++q) for (int i = 0; i < 16; ++i) r += acc[q][i];
    for (int i = 0; i < 8; ++i) r += x[i];
    sink[blockIdx.x * 256 + threadIdx.x] = r;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <int VDST, int F, int EXPS, int CH = 1, int ASRC = 0>
void run(float* sink, unsigned long long* clk) {
    const int iters = 20000;
    k<VDST, F, EXPS, CH, ASRC><<<256, 256>>>(sink, clk, 200, 0.001f);
    (void)hipDeviceSynchronize();
    k<VDST, F, EXPS, CH, ASRC><<<256, 256>>>(sink, clk, iters, 0.001f);
    (void)hipDeviceSynchronize();
    unsigned long long h[256];
    (void)hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
    double c = 0;
    for (int i = 0; i < 256; ++i) c += (double)h[i];
    printf("%saccumulator in %s, chains of %d dependent MFMAs, %d v_fma + %d v_exp per MFMA gap: %.1f cycles per MFMA\n", ASRC ? "A and B operands in AGPRs, " : "", VDST ? "VGPRs" : "AGPRs", CH, F, EXPS, c / 256 / iters / 16);
}

int main() {
    float* sink;
    unsigned long long* clk;
    (void)hipMalloc(&sink, 256 * 256 * 4);
    (void)hipMalloc(&clk, 256 * 8);
    run<0, 0, 0>(sink, clk); run<1, 0, 0>(sink, clk);
    run<0, 4, 0>(sink, clk); run<1, 4, 0>(sink, clk);
    run<0, 6, 0>(sink, clk); run<1, 6, 0>(sink, clk);
    run<0, 8, 0>(sink, clk); run<1, 8, 0>(sink, clk);
    run<0, 2, 2>(sink, clk); run<1, 2, 2>(sink, clk);
    run<0, 4, 2>(sink, clk); run<1, 4, 2>(sink, clk);
    run<0, 0, 3>(sink, clk); run<1, 0, 3>(sink, clk);
    run<0, 0, 4>(sink, clk); run<1, 0, 4>(sink, clk);
    // the same with the 16 MFMAs as chains of DEPENDENT ones (k-steps of one product back to back, as the score products issue them)
    run<1, 0, 0, 4>(sink, clk); run<0, 0, 0, 4>(sink, clk); run<1, 0, 0, 2>(sink, clk);
    run<1, 4, 0, 4>(sink, clk); run<1, 2, 2, 4>(sink, clk); run<0, 2, 2, 4>(sink, clk);
    // the score products' form: A (K fragment) and B (Q fragment) in AGPRs, result in VGPRs
    run<1, 0, 0, 4, 1>(sink, clk); run<1, 0, 0, 1, 1>(sink, clk); run<1, 2, 2, 4, 1>(sink, clk);
    return 0;
}
