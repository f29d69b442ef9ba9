// Micro-benchmark: do the matrix pipe and the VALU of one SIMD overlap across two co-resident waves, and what does a
// softmax-shaped VALU block cost?  One 512-thread workgroup per CU: waves 0-3 play role A, waves 4-7 role B.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/overlap tools/ubench/overlap.hip && /tmp/overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

enum { IDLE = 0, MFMA = 1, VALU = 2, SERIAL = 3, INTERLEAVED = 4, MFMA_IND = 5, MFMA_NOP = 6, MFMA_NOP2 = 7 };

__device__ __forceinline__ f32x16 mm(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }

// 16 MFMAs: two chains of 4 (QK^T shape) + 8 alternating on two accumulators (PV shape)
__device__ __forceinline__ void mfma16(f32x16& s0, f32x16& s1, f32x16& o0, f32x16& o1, bf16x8 a, bf16x8 b) {
#pragma unroll
    for (int i = 0; i < 4; ++i) s0 = mm(a, b, s0);
#pragma unroll
    for (int i = 0; i < 4; ++i) s1 = mm(a, b, s1);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        o0 = mm(a, b, o0);
        o1 = mm(a, b, o1);
    }
}
// softmax-shaped VALU: 32 exp2, 32 adds (4 chains), 16 packs
__device__ __forceinline__ void valu_block(f32x16& x0, f32x16& x1, float& l, bf16x8 (&pf)[4]) {
    float ps[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        x0[i] = __builtin_amdgcn_exp2f(x0[i]);
        x1[i] = __builtin_amdgcn_exp2f(x1[i]);
        ps[i & 3] += x0[i] + x1[i];
    }
    l += (ps[0] + ps[1]) + (ps[2] + ps[3]);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        pf[0][j] = (__bf16)x0[j];
        pf[1][j] = (__bf16)x0[8 + j];
        pf[2][j] = (__bf16)x1[j];
        pf[3][j] = (__bf16)x1[8 + j];
    }
}

__global__ __launch_bounds__(512, 1) void k(int roleA, int roleB, int iters, float* sink, float seed, int prio) {
    const int wid = threadIdx.x >> 6;
    const int role = wid < 4 ? roleA : roleB;
    // prio: 1 = raise waves 0-3, 2 = raise waves 4-7 (s_setprio ignores EXEC, so branch on a scalar)
    if (__builtin_amdgcn_readfirstlane((prio == 1 && wid < 4) || (prio == 2 && wid >= 4))) __builtin_amdgcn_s_setprio(3);
    f32x16 s0, s1, o0, o1;
    bf16x8 a, b, pf[4];
    for (int i = 0; i < 16; ++i) { s0[i] = seed * i; s1[i] = -seed * i; o0[i] = 0; o1[i] = 0; }
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + i); b[i] = (__bf16)(seed - i); }
    for (int q = 0; q < 4; ++q) pf[q] = a;
    float l = 0;
    if (role == MFMA) {
        for (int it = 0; it < iters; ++it) mfma16(s0, s1, o0, o1, a, b);
    } else if (role == MFMA_IND) {  // 16 MFMAs on 4 accumulators round-robin (no back-to-back dependency)
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                s0 = mm(a, b, s0);
                s1 = mm(a, b, s1);
                o0 = mm(a, b, o0);
                o1 = mm(a, b, o1);
            }
        }
    } else if (role >= 6 && role <= 9) {  // scalar padding in every MFMA gap: does the partner's VALU get the vector port then?
#define PADDED(NOP)                                         \
    for (int it = 0; it < iters; ++it) {                    \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {     \
            s0 = mm(a, b, s0);                              \
            asm volatile(NOP);                              \
            s1 = mm(a, b, s1);                              \
            asm volatile(NOP);                              \
            o0 = mm(a, b, o0);                              \
            asm volatile(NOP);                              \
            o1 = mm(a, b, o1);                              \
            asm volatile(NOP);                              \
        }                                                   \
    }
        if (role == 6) { PADDED("s_nop 1") }
        else if (role == 7) { PADDED("s_nop 3") }
        else if (role == 8) { PADDED("s_nop 5") }
        else { PADDED("s_nop 7") }
    } else if (role == VALU) {
        for (int it = 0; it < iters; ++it) {
            valu_block(s0, s1, l, pf);
#pragma unroll
            for (int i = 0; i < 16; ++i) { s0[i] = s0[i] * -0.5f; s1[i] = s1[i] * -0.25f; }  // keep values bounded (32 more VALU)
        }
    } else if (role == SERIAL) {  // what one attention wave does per tile: QK -> softmax -> PV, dependent
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) s0 = mm(a, b, s0);
#pragma unroll
            for (int i = 0; i < 4; ++i) s1 = mm(a, b, s1);
            valu_block(s0, s1, l, pf);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                o0 = mm(a, pf[i], o0);
                o1 = mm(b, pf[i], o1);
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) { s0[i] = -1.0f; s1[i] = -2.0f; }
        }
    }
    else if (role == INTERLEAVED) {  // same work as SERIAL, software-pipelined: tile j's softmax beside tile j+1's QK and tile j-1's PV
        f32x16 t0 = s0, t1 = s1;
        bf16x8 pn[4];
        for (int it = 0; it < iters; ++it) {
            f32x16 n0, n1;
#pragma unroll
            for (int i = 0; i < 16; ++i) { n0[i] = -1.0f; n1[i] = -2.0f; }
#pragma unroll
            for (int i = 0; i < 4; ++i) n0 = mm(a, b, n0);
#pragma unroll
            for (int i = 0; i < 4; ++i) n1 = mm(a, b, n1);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                o0 = mm(a, pf[i], o0);
                o1 = mm(b, pf[i], o1);
            }
            valu_block(t0, t1, l, pn);
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x400, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
            }
            t0 = n0;
            t1 = n1;
#pragma unroll
            for (int q = 0; q < 4; ++q) pf[q] = pn[q];
        }
        s0 = t0;
        s1 = t1;
    }
    if (role != IDLE) {
        float r = l;
        for (int i = 0; i < 16; ++i) r += s0[i] + s1[i] + o0[i] + o1[i];
        for (int q = 0; q < 4; ++q) r += (float)pf[q][0];
        sink[blockIdx.x * 512 + threadIdx.x] = r;
    }
}

int main() {
    float* sink;
    hipMalloc(&sink, 256 * 512 * 4);
    const int iters = 20000;
    const char* names[] = {"idle", "mfma16", "valu", "serial", "interlv", "mfma_ind", "nop1", "nop3", "nop5", "nop7"};
    int cfg[][3] = {{INTERLEAVED, IDLE, 0}, {INTERLEAVED, INTERLEAVED, 0}, {SERIAL, IDLE, 0},
                    {6, IDLE, 0}, {6, VALU, 0}, {7, IDLE, 0}, {7, VALU, 0}, {8, IDLE, 0}, {8, VALU, 0}, {VALU, 8, 0}, {9, IDLE, 0}, {9, VALU, 0},
                    {MFMA, IDLE, 0}, {VALU, IDLE, 0}, {MFMA, VALU, 0}, {VALU, MFMA, 0}, {MFMA, VALU, 1}, {MFMA, VALU, 2}, {VALU, MFMA, 1}, {VALU, MFMA, 2},
                    {MFMA_IND, IDLE, 0}, {MFMA_IND, VALU, 0}, {MFMA_IND, VALU, 2}, {VALU, MFMA_IND, 0}, {SERIAL, SERIAL, 0}, {SERIAL, SERIAL, 2}, {INTERLEAVED, INTERLEAVED, 0}};
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (auto& c : cfg) {
        k<<<256, 512>>>(c[0], c[1], 100, sink, 0.001f, c[2]);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<<<256, 512>>>(c[0], c[1], iters, sink, 0.001f, c[2]);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("A=%-8s B=%-8s prio=%d %8.3f ms  -> %7.1f ns per iteration (x2.1 GHz = %6.0f cycles)\n", names[c[0]], names[c[1]], c[2], ms,
               ms * 1e6 / iters, ms * 1e6 / iters * 2.1);
    }
    return 0;
}
