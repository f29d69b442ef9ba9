// Micro-benchmark: LDS read bandwidth per CU for the two fragment-read shapes of gemm256.hip, with and without MFMAs beside them.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_bw tools/ubench/lds_read_bw.hip && /tmp/lds_bw
// One 512-thread workgroup per CU (8 waves, 2 per SIMD).  Modes: b128 = ds_read_b128 (16 B per lane, conflict-free pattern),
// tr = ds_read_b64_tr_b16 (8 B per lane); "+mfma" issues 16 independent v_mfma_f32_16x16x32_bf16 per 8 (b128) / 16 (tr) reads, the
// ratio of the GEMM k-loop.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ bf16x4 lds_tr16(const void* p) {
    bf16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"((uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p) : "memory");
    return v;
}

template <int MODE, bool WITH_MFMA>
__global__ __launch_bounds__(512) void k(int rounds, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 65536 / 4; i += 512) reinterpret_cast<float*>(smem)[i] = (float)i;
    __syncthreads();
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 keep = {};
    const char* base = smem + wid * 8192;
    for (int r = 0; r < rounds; ++r) {
        bf16x8 f[8];
        if constexpr (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) f[i] = *reinterpret_cast<const bf16x8*>(base + i * 1024 + lane * 16);
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const bf16x4 lo = lds_tr16(base + i * 1024 + lane * 8), hi = lds_tr16(base + i * 1024 + 512 + lane * 8);
                f[i] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        if constexpr (WITH_MFMA) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[i & 7], f[(i + 3) & 7], acc[i], 0, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) keep = keep + f[i];
        }
    }
    float s = (float)keep[0];
    for (int i = 0; i < 16; ++i) s += acc[i][0];
    sink[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MODE, bool WITH_MFMA>
void run(float* sink, const char* name) {
    const int rounds = 20000;
    (void)hipFuncSetAttribute((const void*)k<MODE, WITH_MFMA>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    k<MODE, WITH_MFMA><<<256, 512, 65536>>>(100, sink);
    (void)hipEventRecord(e0);
    k<MODE, WITH_MFMA><<<256, 512, 65536>>>(rounds, sink);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double bytes_cu = 8.0 * 8 * 1024 * rounds;  // per CU: 8 waves x 8 KiB per round
    const double mfma_flop = WITH_MFMA ? 256.0 * 8 * 16 * 16384.0 * rounds : 0.0;
    printf("%-22s %8.1f GB/s per CU (%.0f B/clk at 2.4 GHz)   MFMA %7.1f TFLOP/s\n", name, bytes_cu / (ms * 1e-3) / 1e9,
           bytes_cu / (ms * 1e-3) / 2.4e9, mfma_flop / (ms * 1e-3) / 1e12);
}

int main() {
    float* sink;
    (void)hipMalloc(&sink, 256 * 512 * 4);
    run<0, false>(sink, "ds_read_b128");
    run<1, false>(sink, "ds_read_b64_tr_b16");
    run<0, true>(sink, "ds_read_b128 + mfma");
    run<1, true>(sink, "tr_b16 + mfma");
    return 0;
}
