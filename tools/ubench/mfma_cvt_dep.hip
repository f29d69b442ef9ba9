// Micro-benchmark (r05): what does a v_cvt_pk_bf16_f32 cost beside MFMAs, and what does it cost when the NEXT MFMA reads its result
// (the attention backward packs P / dS into the B operand of the gradient products right in front of them)?  One wave per SIMD, 16 x
// v_mfma_f32_32x32x16_bf16 per trip, accumulators in AGPRs, per MFMA gap: NC packs (+ NM multiplies feeding them when CHAIN).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_cvt_dep tools/ubench/mfma_cvt_dep.hip && /tmp/mfma_cvt_dep
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// DEP 0: the packs write registers no MFMA reads; 1: the packs of gap g write the B operand of the MFMA that ENDS gap g (read at once);
// 2: ... of the MFMA one gap later (software-pipelined by one MFMA).  NM multiplies per gap feed the packs (mul -> cvt chain) when NM > 0.
template <int NC, int NM, int DEP>
__global__ __launch_bounds__(256, 1) void k(float* sink, unsigned long long* clk, int iters, float seed) {
    f32x16 acc[4];
    bf16x8 a;
    u32x4 b[2], junk;
    float x[8], y[8];
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed * (i + 1)); x[i] = seed * (i + 3); y[i] = 1.0f + seed * i; }
    for (int q = 0; q < 4; ++q) for (int i = 0; i < 16; ++i) acc[q][i] = 0.f;
    for (int i = 0; i < 4; ++i) { b[0][i] = 0x3f803f80u + i; b[1][i] = 0x3f813f80u + i; junk[i] = 0; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            // the MFMA that opens gap m reads b[m & 1]
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[m & 3]) : "v"(a), "v"(b[m & 1]));
#pragma unroll
            for (int f = 0; f < NM; ++f) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(x[f & 7]) : "v"(x[f & 7]), "v"(y[f & 7]));
#pragma unroll
            for (int f = 0; f < NC; ++f) {
                if constexpr (DEP == 0) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(junk[f & 3]) : "v"(x[(2 * f) & 7]), "v"(x[(2 * f + 1) & 7]));
                else if constexpr (DEP == 1) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(b[(m + 1) & 1][f & 3]) : "v"(x[(2 * f) & 7]), "v"(x[(2 * f + 1) & 7]));
                else asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(b[m & 1][f & 3]) : "v"(x[(2 * f) & 7]), "v"(x[(2 * f + 1) & 7]));  // read two MFMAs on
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0.f;
    for (int q = 0; q < 4; ++q) for (int i = 0; i < 16; ++i) r += acc[q][i];
    for (int i = 0; i < 8; ++i) r += x[i];
    for (int i = 0; i < 4; ++i) r += (float)junk[i] + (float)b[0][i] + (float)b[1][i];
    sink[blockIdx.x * 256 + threadIdx.x] = r;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <int NC, int NM, int DEP>
void run(float* sink, unsigned long long* clk) {
    const int iters = 20000;
    k<NC, NM, DEP><<<256, 256>>>(sink, clk, 200, 0.001f);
    (void)hipDeviceSynchronize();
    k<NC, NM, DEP><<<256, 256>>>(sink, clk, iters, 0.001f);
    (void)hipDeviceSynchronize();
    unsigned long long h[256];
    (void)hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
    double c = 0;
    for (int i = 0; i < 256; ++i) c += (double)h[i];
    const char* dep = DEP == 0 ? "read by no MFMA" : DEP == 1 ? "read by the NEXT MFMA" : "read by the MFMA after the next";
    printf("%d v_mul + %d v_cvt_pk per MFMA gap, packs %-32s: %.1f cycles per MFMA\n", NM, NC, dep, c / 256 / iters / 16);
}

int main() {
    float* sink;
    unsigned long long* clk;
    (void)hipMalloc(&sink, 256 * 256 * 4);
    (void)hipMalloc(&clk, 256 * 8);
    run<0, 0, 0>(sink, clk);
    run<2, 0, 0>(sink, clk); run<4, 0, 0>(sink, clk); run<6, 0, 0>(sink, clk); run<8, 0, 0>(sink, clk);
    run<4, 0, 1>(sink, clk); run<4, 0, 2>(sink, clk);
    run<2, 2, 0>(sink, clk); run<2, 2, 1>(sink, clk); run<2, 2, 2>(sink, clk);
    run<4, 4, 0>(sink, clk); run<4, 4, 1>(sink, clk); run<4, 4, 2>(sink, clk);
    run<0, 4, 0>(sink, clk); run<0, 6, 0>(sink, clk);
    return 0;
}
