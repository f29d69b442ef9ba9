// Micro-benchmark (r05, verdict item 2): the fused attention backward's MFMA mix per (32-query, 32-key) block pair - 8 score MFMAs whose
// results go to VGPRs (S and dP chains of 4) and 12 accumulating MFMAs (dV^T, dK^T, dQ^T) - with every operand re-read from LDS each
// pair (ds_read_b128, random bf16 data), one wave per SIMD, in two forms of the ACCUMULATING products:
//   mode 0: all twenty as v_mfma_f32_32x32x16_bf16 (the shipped kernel)
//   mode 1: the twelve accumulating ones as 24 x v_mfma_f32_16x16x32_bf16 (same FLOPs, the shape the GEMMs hold 2.1 GHz under)
//   mode 2 / 3: only the accumulating part of mode 0 / 1 (bare comparison of the two shapes beside the same LDS reads)
// Reports time, executed TFLOP/s and the in-kernel clock (s_memtime / s_memrealtime).  Does the 16x16x32 form buy clock here?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_shape_mix tools/ubench/mfma_shape_mix.hip && /tmp/mfma_shape_mix
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ void mfma32_v(f32x16& d, const bf16x8& a, const bf16x8& b) {  // VGPR result, accumulate
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma32_a(f32x16& d, const bf16x8& a, const bf16x8& b) {  // AGPR accumulator
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(d) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma16_a(f32x4& d, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(d) : "v"(a), "v"(b));
}

template <int MODE>
__global__ __launch_bounds__(256, 1) void mix(const uint16_t* __restrict__ src, float* __restrict__ sink, unsigned long long* __restrict__ clk, int iters) {
    __shared__ __attribute__((aligned(16))) char lds[65536];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 65536 / 16; i += 256) reinterpret_cast<uint4*>(lds)[i] = reinterpret_cast<const uint4*>(src)[i + blockIdx.x % 7 * 64];
    __syncthreads();
    f32x16 s, dp, acc32[6];
    f32x4 acc16[24];
#pragma unroll
    for (int i = 0; i < 16; ++i) { s[i] = 0.f; dp[i] = 0.f; }
#pragma unroll
    for (int k = 0; k < 6; ++k)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc32[k][i] = 0.f;
#pragma unroll
    for (int k = 0; k < 24; ++k) acc16[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    const char* base = lds + (tid >> 6) * 16384 + lane * 16;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        const char* p = base + (it & 3) * 512;
        bf16x8 f[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) f[k] = *reinterpret_cast<const bf16x8*>(p + 1024 * (k & 1) + 2048 * (k >> 1));
        if constexpr (MODE < 2) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { mfma32_v(s, f[k], f[k + 4]); mfma32_v(dp, f[k + 4], f[k]); }
#pragma unroll
            for (int i = 0; i < 16; ++i) { s[i] *= 0.001f; dp[i] *= 0.001f; }  // (keeps the chains bounded; 32 VALU per pair, as the exp / mul block)
        }
        if constexpr (MODE == 0 || MODE == 2) {
#pragma unroll
            for (int k = 0; k < 12; ++k) mfma32_a(acc32[k % 6], f[k & 7], f[(k + 3) & 7]);
        } else {
#pragma unroll
            for (int k = 0; k < 24; ++k) mfma16_a(acc16[k], f[k & 7], f[(k + 3) & 7]);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) r += s[i] + dp[i];
#pragma unroll
    for (int k = 0; k < 6; ++k)
#pragma unroll
        for (int i = 0; i < 16; ++i) r += acc32[k][i];
#pragma unroll
    for (int k = 0; k < 24; ++k) r += acc16[k][0] + acc16[k][1] + acc16[k][2] + acc16[k][3];
    sink[blockIdx.x * 256 + tid] = r;
    if (tid == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

int main() {
    const int nb = 256, iters = 40000;
    uint16_t* src;
    float* sink;
    unsigned long long* clk;
    (void)hipMalloc(&src, 1 << 20);
    (void)hipMalloc(&sink, nb * 256 * 4);
    (void)hipMalloc(&clk, nb * 16);
    uint16_t* h = (uint16_t*)malloc(1 << 20);
    srand(1);
    for (int i = 0; i < (1 << 19); ++i) h[i] = (uint16_t)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15));  // random bf16 of magnitude ~0.01-0.03, both signs
    (void)hipMemcpy(src, h, 1 << 20, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const char* names[4] = {"20 x 32x32x16 (shipped mix)", "8 x 32x32x16 + 24 x 16x16x32", "12 x 32x32x16 accumulating only", "24 x 16x16x32 accumulating only"};
    const double macs[4] = {20 * 16384.0, 20 * 16384.0, 12 * 16384.0, 12 * 16384.0};
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 4; ++mode) {
            auto launch = [&](int n) {
                if (mode == 0) mix<0><<<nb, 256>>>(src, sink, clk, n);
                else if (mode == 1) mix<1><<<nb, 256>>>(src, sink, clk, n);
                else if (mode == 2) mix<2><<<nb, 256>>>(src, sink, clk, n);
                else mix<3><<<nb, 256>>>(src, sink, clk, n);
            };
            for (int w = 0; w < 20; ++w) launch(iters);  // (the clock settles under load)
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0);
            launch(iters);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            unsigned long long hc[2 * nb];
            (void)hipMemcpy(hc, clk, sizeof(hc), hipMemcpyDeviceToHost);
            double ghz = 0;
            for (int b = 0; b < nb; ++b) ghz += (double)hc[2 * b] / (double)hc[2 * b + 1] * 0.1;
            ghz /= nb;
            const double flops = 2.0 * macs[mode] * iters * 4.0 * nb;
            printf("mode %d %-34s %8.3f ms  %7.1f TFLOP/s executed  in-kernel clock %.3f GHz  cycles per pair %.1f\n", mode, names[mode], ms, flops / ms / 1e9, ghz,
                   (double)hc[0] / iters);
        }
    return 0;
}
