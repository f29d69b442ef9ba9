// MEASUREMENT KERNEL (not part of the library): the r03 verdict's "full-row GEMM + residual + LayerNorm" built in its plainest form, to put a
// number beside the costing in DESIGN.md section 8.  x = R + A W^T (fp32), y = LayerNorm(x) * gamma (bf16), mean / rstd per row, N = 768 fixed.
//
// One workgroup (8 waves) owns 64 whole rows: wave w computes columns [96 w, 96 w + 96) of all 64 rows (2 x 3 blocks of v_mfma_f32_32x32x16_bf16,
// 96 accumulators), so a row's 768 values sit in the registers of the eight waves and the statistics are two LDS exchanges away - no second
// pass over HBM.  k-steps of 32 through a three-stage LDS ring filled by LDS-DMA (A image 4 KiB, W image 48 KiB per stage: a wave stages
// exactly the 96 W rows it reads, so only the A image needs the workgroup barrier), one counted vmcnt wait + one barrier per k-step.
// The point of the measurement: per k-step a wave issues 6-7 DMA instructions for 12 MFMAs (384 matrix cycles), against 4 per 512 cycles
// in the 256 x 256 ring kernel whose loop is already bound by issuing them.
//
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/ubench/libfullrow.so tools/ubench/fullrow_gemm_ln.hip
//   python tools/fullrow_probe.py
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

namespace {

constexpr int kN = 768, kBM = 64, kBK = 32;
constexpr int kStage = 4096 + 49152;  // A image [64][32] bf16 + W image [768][32] bf16, 64-byte rows, 16-byte chunk ^ ((row >> 2) & 3)
constexpr int kStages = 3;
constexpr int kLds = kStages * kStage;  // 159744

__device__ __forceinline__ void dma16(uint32_t voff, const char* sbase, uint32_t lds) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds) : "memory", "m0");
}
#define WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

__device__ __forceinline__ float half_wave_sum(float v) {  // over the 32 lanes that share lane >> 5
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <int MODE>  // 0: full (x, y, mean, rstd); 1: k-loop only (results folded into one store per lane: timing probe)
__global__ __launch_bounds__(512, 2) void fullrow_kernel(const uint16_t* __restrict__ A, const uint16_t* __restrict__ W,
                                                         const float* __restrict__ R, const float* __restrict__ gamma,
                                                         float* __restrict__ X, uint16_t* __restrict__ Y, float* __restrict__ mean_out,
                                                         float* __restrict__ rstd_out, int64_t M, int64_t K, float eps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, hh = lane >> 5, l31 = lane & 31;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem);
    const int nk = (int)(K / kBK);
    const int tiles = (int)(M / kBM);
    // per-lane source offset of a 16-row DMA piece: LDS position lane = (row lane >> 2, chunk slot lane & 3) holds source chunk slot ^ ((row >> 2) & 3)
    const uint32_t voff = (uint32_t)((lane >> 2) * (int)K * 2 + (((lane & 3) ^ ((lane >> 4) & 3)) << 4));
    // fragment read offsets inside an image (row-dependent swizzle)
    int a_off[2], b_off[3];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        const int row = rb * 32 + l31;
        a_off[rb] = row * 64 + ((hh ^ ((row >> 2) & 3)) << 4);  // kk = 0; kk = 1 flips chunk bit 1 (XOR 32 bytes)
    }
#pragma unroll
    for (int cb = 0; cb < 3; ++cb) {
        const int n = 96 * wid + 32 * cb + l31;
        b_off[cb] = 4096 + n * 64 + ((hh ^ ((n >> 2) & 3)) << 4);
    }

    for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
        const int64_t m0 = (int64_t)t * kBM;
        const char* a_src = reinterpret_cast<const char*>(A + (m0 + 16 * (wid & 3)) * K);  // waves 0-3: 16 rows of the A image each
        const char* w_src = reinterpret_cast<const char*>(W + (int64_t)(96 * wid) * K);    // this wave's own 96 W rows: six 16-row pieces
        auto issue = [&](int kt) {
            const uint32_t st = lds0 + (uint32_t)((kt % kStages) * kStage);
            const int64_t kb = (int64_t)kt * kBK * 2;
            if (wid < 4) dma16(voff, a_src + kb, st + 1024u * wid);
#pragma unroll
            for (int i = 0; i < 6; ++i) dma16(voff, w_src + kb + (int64_t)(16 * i) * K * 2, st + 4096u + (uint32_t)(96 * wid + 16 * i) * 64u);
        };
        f32x16 acc[2][3];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int cb = 0; cb < 3; ++cb)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[rb][cb][v] = 0.f;
        issue(0);
        if (nk > 1) issue(1);
        for (int kt = 0; kt < nk; ++kt) {
            // stage kt has landed for this wave when at most the next stage's pieces are outstanding
            if (kt + 1 < nk) {
                if (wid < 4) WAIT_VM(7);
                else WAIT_VM(6);
            } else {
                WAIT_VM(0);
            }
            __builtin_amdgcn_s_barrier();  // every wave's part of stage kt is visible; every wave is done reading stage kt - 1
            if (kt + 2 < nk) issue(kt + 2);
            const char* st = smem + (kt % kStages) * kStage;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8 fa[2], fb[3];
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) fa[rb] = *reinterpret_cast<const bf16x8*>(st + (a_off[rb] ^ (kk << 5)));
#pragma unroll
                for (int cb = 0; cb < 3; ++cb) fb[cb] = *reinterpret_cast<const bf16x8*>(st + (b_off[cb] ^ (kk << 5)));
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                    for (int cb = 0; cb < 3; ++cb) acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[rb], fb[cb], acc[rb][cb], 0, 0, 0);
            }
        }
        __builtin_amdgcn_s_barrier();  // the ring is free: its first bytes become the statistics exchange

        if constexpr (MODE == 1) {
            float s = 0.f;
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int cb = 0; cb < 3; ++cb)
#pragma unroll
                    for (int v = 0; v < 16; ++v) s += acc[rb][cb][v];
            X[(m0 + l31) * kN + 96 * wid + hh] = s;
            continue;
        }
        // ---- epilogue: x = acc + R; two-pass statistics over the eight waves; stores.  Element (rb, cb, v): row rb*32 + (v/4)*8 + hh*4 + v%4, column 96 wid + 32 cb + l31
        float* red_s = reinterpret_cast<float*>(smem);         // [8 waves][64 rows]
        float* red_q = reinterpret_cast<float*>(smem + 2048);  // [8 waves][64 rows]
        float gm[3];
#pragma unroll
        for (int cb = 0; cb < 3; ++cb) gm[cb] = gamma[96 * wid + 32 * cb + l31];
        // (32-bit offsets from the tile's first element: 96 64-bit addresses per lane would not fit beside the accumulators)
        const float* Rt = R + m0 * kN;
        float* Xt = X + m0 * kN;
        uint16_t* Yt = Y + m0 * kN;
        const int col0 = 96 * wid + l31;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int row = rb * 32 + (v >> 2) * 8 + hh * 4 + (v & 3);
                float s = 0.f;
#pragma unroll
                for (int cb = 0; cb < 3; ++cb) {
                    const int idx = row * kN + col0 + 32 * cb;
                    const float x = acc[rb][cb][v] + Rt[idx];
                    acc[rb][cb][v] = x;
                    Xt[idx] = x;
                    s += x;
                }
                s = half_wave_sum(s);
                if (l31 == 0) red_s[wid * 64 + row] = s;
            }
        __syncthreads();
        float mean[2][16];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int row = rb * 32 + (v >> 2) * 8 + hh * 4 + (v & 3);
                float s = 0.f;
#pragma unroll
                for (int w = 0; w < 8; ++w) s += red_s[w * 64 + row];
                mean[rb][v] = s * (1.0f / kN);
                float q = 0.f;
#pragma unroll
                for (int cb = 0; cb < 3; ++cb) {
                    const float d = acc[rb][cb][v] - mean[rb][v];
                    acc[rb][cb][v] = d;
                    q += d * d;
                }
                q = half_wave_sum(q);
                if (l31 == 0) red_q[wid * 64 + row] = q;
            }
        __syncthreads();
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int row = rb * 32 + (v >> 2) * 8 + hh * 4 + (v & 3);
                float q = 0.f;
#pragma unroll
                for (int w = 0; w < 8; ++w) q += red_q[w * 64 + row];
                const float rstd = 1.0f / sqrtf(q * (1.0f / kN) + eps);
#pragma unroll
                for (int cb = 0; cb < 3; ++cb) {
                    const int idx = row * kN + col0 + 32 * cb;
                    const __bf16 yb = (__bf16)(acc[rb][cb][v] * rstd * gm[cb]);
                    Yt[idx] = __builtin_bit_cast(uint16_t, yb);
                }
                if (wid == 0 && l31 == 0) {
                    mean_out[m0 + row] = mean[rb][v];
                    rstd_out[m0 + row] = rstd;
                }
            }
        __syncthreads();  // the exchange area is ring again
    }
}

}  // namespace

extern "C" int fullrow_gemm_ln(const void* A, const void* W, const float* R, const float* gamma, float* X, void* Y, float* mean, float* rstd,
                               int64_t M, int64_t K, float eps, int mode, int grid, void* stream) {
    if (M % kBM != 0 || K % kBK != 0 || K < 2 * kBK || (K * 2 * 16) >= (int64_t(1) << 31)) return 1;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int tiles = (int)(M / kBM);
    if (grid <= 0 || grid > tiles) grid = tiles < 256 ? tiles : 256;
    static bool once = false;
    if (!once) {
        if (hipFuncSetAttribute((const void*)fullrow_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds) != hipSuccess) return 2;
        if (hipFuncSetAttribute((const void*)fullrow_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds) != hipSuccess) return 2;
        once = true;
    }
    const uint16_t* a = static_cast<const uint16_t*>(A);
    const uint16_t* w = static_cast<const uint16_t*>(W);
    if (mode == 1) fullrow_kernel<1><<<grid, 512, kLds, s>>>(a, w, R, gamma, X, static_cast<uint16_t*>(Y), mean, rstd, M, K, eps);
    else fullrow_kernel<0><<<grid, 512, kLds, s>>>(a, w, R, gamma, X, static_cast<uint16_t*>(Y), mean, rstd, M, K, eps);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}
