// Backward of the GLOBAL attention layers as ONE key-parallel kernel that executes each of the five matrix products once
// (S = Q K^T, dP = dO V^T, dV^T += dO^T P, dK^T += Q^T dS, dQ^T += K^T dS^T), head_dim 64, gfx950.
// cm3p_attn_bwd_fused: the same mathematics as cm3p_attn_bwd with window < 0 (attention_bwd.hip, which recomputes S and dP in
// a second, query-parallel kernel: seven products for five), i.e. the backward of F.scaled_dot_product_attention
// (TF:integrations/sdpa_attention.py:153-163) under the key-padding mask of TF:masking_utils.py:168-179, with the inverse of
// apply_rotary_pos_emb (TF:models/modernbert/modeling_modernbert.py:188-219) applied to dq / dk.
//
// Three launches:
//   prep    delta = rowsum(dO o O) and the score offsets, written per 64-row tile as the two 64-float rows the main kernel's
//           accumulators start from (-lse in the score product's units, -delta; -inf / 0 for rows past the sequence and for the
//           padding tiles the main kernel's unrolled ring may touch): 512 contiguous bytes per tile, LDS-DMA friendly.
//   fused   workgroup = 4 waves = 256 keys (64 per wave, ONE wave per SIMD with the whole 512-entry register file, dK^T / dV^T in
//           128 AGPRs) streaming 64-query tiles of Q / dO.  Everything of attn_bwd_dkv3_kernel's hand-placed instruction stream
//           is kept (attention_bwd.hip); new here:
//             - tiles arrive by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write; swizzled by permuting the
//               SOURCE chunks since the DMA writes lane-linearly), three slots, a tile's DMA is in flight for two tile periods;
//               ordering is explicit: a counted s_waitcnt vmcnt in front of the single workgroup barrier per tile (every wave
//               issues exactly five DMAs and two stores per tile, DMAs first: tests/test_kernel_isa.py pins the pattern);
//             - dS leaves the accumulator layout (key on the lane) through a [256 keys][64 queries] bf16 image in LDS, two
//               images, each holding a half-shifted 64-query window ("epoch": second half of tile t-1, first half of tile t) so
//               that the ONE barrier per tile (between steps 1 and 2) both publishes an epoch and retires the one before;
//             - dQ^T of an epoch = K^T dS^T over the workgroup's 256 keys, 16 MFMAs per wave (wave w owns d block w >> 1, query
//               block w & 1), four per pipeline step.  K^T comes from a K image (transposed reads; invisible keys are stored as
//               ZERO rows: nothing is masked in the loop) ONCE: 15 of its 16 fragments stay in AGPRs for the whole kernel (a wave
//               keeps at most 15 LDS operations in flight, so LDS instruction count is a throughput limit at one wave per
//               SIMD); dS^T by transposed reads of the dS image.  The finished bf16 block goes through a wave-private
//               transposition buffer to this key block's slab as whole 64-byte row pieces (16 rows per store instruction);
//             - exponentials carry the hardware clamp (v_exp_f32 ... clamp): p <= 1 also for keys under the padding mask, whose
//               scores the row maximum does not bound, so dS stays finite and meets the zero K row as an exact zero.
//   reduce  dq = scale * inverse_rope(sum of the key blocks' bf16 slabs, fp32, fixed order) -> the q third of dqkv.
//           Two key blocks share a slab (r03): the main kernel runs twice, first over the even key blocks, which STORE their partial,
//           then over the odd ones, which ADD theirs to the same slab with global_atomic_pk_add_bf16 (exactly one add per element,
//           ordered behind the store by the launch boundary: deterministic; the memory side does the add, nothing is fetched).  Half
//           the slabs to write and to reduce.
//           (r03: the same sum done inside the fused kernel by the last workgroup of a (batch, head) to finish - ticket counter,
//           agent-scope fences, bit-identical - was SLOWER, 5.40 vs 4.00 + 0.68 ms at B = 32, S = 4096: a reducing CU has four
//           waves and gets 10 GB/s out of a loaded memory system, and its matrix cores idle meanwhile.  DESIGN.md.)
//
// Workspace (caller-owned, cm3p_attn_bwd_fused_workspace_bytes): the tile statistics and the slabs
// [B * nh][key blocks][64 * tiles + 64 dump rows][64] bf16 (C2: 3.3 GB, C4: 6.5 GB - sized for 288 GB of HBM).
#include <stdlib.h>

#include <type_traits>

#include "attn_common.h"

namespace {

#ifndef CM3P_FABL
#define CM3P_FABL 0  // timing-only ablation builds (tools/ubench/attn_bwd_ablate.sh; results are wrong by construction): 1 no barrier,
#endif               // 2 no tile DMA in the loop, 4 no dS image writes, 8 no dQ operand reads, 16 no slab stores, 32 no dQ MFMAs,
                     // 64 no exponentials, 128 no dS multiplies, 256 no bf16 packs (P / dS fragments a non-zero constant), 512 no Q / dO
                     // row-fragment and statistics reloads, 1024 no transposed Q^T / dO^T fragment reloads, 2048 the twelve accumulating
                     // products of a step (dV^T, dK^T, dQ^T) as pairs of v_mfma_f32_16x16x32_bf16 on four-register accumulators: the same
                     // FLOPs in the other MFMA shape with none of the layout work a port needs (r06, tools/ubench/attn_mfma_shape.sh)

// Wait trace (-DCM3P_FTRACE=1, tools/attn_bwd_trace.py): per wave the cycles spent (0) at the counted vmcnt wait - the DMA of tile t+1 not landed -,
// (1) at the LDS drain + the tile's one barrier, and (2) in all; the stamps are read behind the barrier's own lgkmcnt(0), so they add no wait.
#ifndef CM3P_FTRACE
#define CM3P_FTRACE 0
#endif
#if CM3P_FTRACE
__device__ unsigned long long* g_bwdf_trace = nullptr;
#endif

constexpr int kFStage = 16384 + 512;        // slot: 4 groups of [16 Q rows | 16 dO rows] (128-byte rows) + [2][64] floats
constexpr int kFSlots = 3;
constexpr int kFKimg = kFSlots * kFStage;   // K image [256 keys][128 bytes]
constexpr int kFdS = kFKimg + 32768;        // two dS images [256 keys][64 queries] bf16
constexpr int kFZ = kFdS + 2 * 32768;       // per wave [32 query rows][80 bytes]: the finished dQ block on its way to row-major
constexpr int kFLds = kFZ + 4 * 2560;       // 159232 bytes
constexpr int kFUnroll = 6;                 // lcm(slots, dS images)
constexpr int kFPadTiles = 12;              // statistics tiles past ceil(S / 64) that the ring may DMA (all -inf / 0)

__host__ __device__ inline int64_t fused_stat_floats(int B, int S, int nh) { return (int64_t)B * nh * ((S + 63) / 64 + kFPadTiles) * 128; }
__host__ __device__ inline int fused_slab_rows(int S) { return ((S + 63) / 64) * 64 + 64; }
__host__ __device__ inline int fused_slabs(int S, int G = 2) { return ((S + 255) / 256 + G - 1) / G; }  // key blocks G j .. G j + G - 1 share slab j

__device__ __forceinline__ bf16x8 gload_frag8(const uint16_t* p) { return *reinterpret_cast<const bf16x8*>(p); }

// D (VGPRs) = A (AGPRs) * B (VGPRs) [+ D]
__device__ __forceinline__ void mfma_av0(f32x16& d, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(d) : "a"(a), "v"(b));
}
__device__ __forceinline__ void mfma_ava(f32x16& d, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "a"(a), "v"(b));
}
__device__ __forceinline__ void mfma_vva(f32x16& d, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
}

// ---- timing-only forms of the accumulating products (CM3P_FABL & 2048): two 16x16x32 MFMAs on two quads where the product issues one
// 32x32x16 on a 16-register block.  The accumulators are DECLARED as quads in that build (carving quads out of a 16-register tuple per
// MFMA makes hipcc copy the tuple around every statement: thousands of v_accvgpr_* and scratch).
constexpr bool kQuads = (CM3P_FABL & 2048) != 0;
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ void mfma16_av0(f32x4& d, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(d) : "a"(a), "v"(b));
}
__device__ __forceinline__ void mfma16_ava(f32x4& d, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(d) : "a"(a), "v"(b));
}
__device__ __forceinline__ void mfma16_vva(f32x4& d, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
}

// workgroup barrier that orders LDS traffic only (this wave's LDS-DMA is covered by the counted vmcnt wait in front of it)
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// p = min(2^x, 1) (v_exp_f32 with the clamp bit; NaN -> 0)
template <bool PRE>
__device__ __forceinline__ void exp2c_pair(f32x16& s, int i, float cm) {
    if constexpr ((CM3P_FABL & 64) != 0) return;
    s[i] = __builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(PRE ? s[i] : s[i] * cm), 0.f, 1.f);
    s[i + 1] = __builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(PRE ? s[i + 1] : s[i + 1] * cm), 0.f, 1.f);
}

// ---- prep -----------------------------------------------------------------------------------------------------------------------
// grid (tiles + kFPadTiles, nh, B), 256 threads: four threads per row of the tile
__global__ __launch_bounds__(256) void attn_bwd_prep_kernel(const uint16_t* __restrict__ o_rows, const uint16_t* __restrict__ d_o,
                                                            const float* __restrict__ lse, float* __restrict__ stat_ws, int Smax, int nh,
                                                            float lse_mul, VarLen vl) {
    const int t = blockIdx.x, head = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
    const SeqView sv(vl, b, head, Smax, nh);
    const int S = sv.S;
    const int row = t * 64 + (tid >> 2), part = tid & 3;
    const int64_t ldo = (int64_t)nh * 64;
    float acc = 0.f;
    if (row < S) {
        const uint16_t* po = o_rows + (sv.row0 + row) * ldo + head * 64 + part * 16;
        const uint16_t* pg = d_o + (sv.row0 + row) * ldo + head * 64 + part * 16;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const bf16x8 a = gload_frag8(po + 8 * h), g = gload_frag8(pg + 8 * h);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += (float)a[j] * (float)g[j];
        }
    }
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    if (part == 0) {
        float* st = stat_ws + (((int64_t)b * nh + head) * gridDim.x + t) * 128;
        const bool ok = row < S;
        // (rows whose lse is +inf see no key: -inf, p = 0)
        st[tid >> 2] = ok ? lse[sv.stat0 + row] * lse_mul : kNegInf;
        st[64 + (tid >> 2)] = ok ? -acc : 0.f;
    }
}

// ---- reduce ---------------------------------------------------------------------------------------------------------------------
// One 64-row tile of one (batch, head) by 256 threads: four threads per query row, each owns head dims [8j, 8j+8) and [32+8j, 32+8j+8).
// The slabs are summed in slab order, 8 at a time with all 16 loads of a thread in flight (r06: the batch was 16, but the step's shapes
// have exactly 8 slabs per (batch, head) - 16 key blocks in pairs at S = 4096, 32 in fours at 8192 - so half of every thread's loads
// were repeats of the last slab that were never summed; same sums in the same order).
__device__ __forceinline__ void dq_reduce_tile(const uint16_t* __restrict__ dq_part, uint16_t* __restrict__ dqkv, int t, int head, int b,
                                               int tid, const SeqView& sv, int Smax, int nh, float scale,
                                               const float* __restrict__ rope_cos, const float* __restrict__ rope_sin,
                                               int64_t pos_batch_stride, int G) {
    const int S = sv.S;
    const int row = t * 64 + (tid >> 2), j = tid & 3;
    if (row >= S) return;
    const int nkb = fused_slabs(Smax, G), nkb_b = fused_slabs(S, G);  // (slabs, not key blocks: G key blocks share one)
    const int64_t slab = (int64_t)fused_slab_rows(Smax) * 64;
    const uint16_t* p = dq_part + ((int64_t)b * nh + head) * nkb * slab + (int64_t)row * 64 + 8 * j;
    float lo[8], hi[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) lo[i] = hi[i] = 0.f;
    constexpr int kBatch = 8;
    for (int kb0 = 0; kb0 < nkb_b; kb0 += kBatch) {
        uint4 a[kBatch], c[kBatch];
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            const int kb = min(kb0 + u, nkb_b - 1);  // (past the last block: a repeated load, not summed)
            a[u] = gload16<(CM3P_NT & 8) != 0>(p + kb * slab);
            c[u] = gload16<(CM3P_NT & 8) != 0>(p + kb * slab + 32);
        }
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            if (kb0 + u < nkb_b) {
                lo[0] += bf16lo(a[u].x), lo[1] += bf16hi(a[u].x), lo[2] += bf16lo(a[u].y), lo[3] += bf16hi(a[u].y);
                lo[4] += bf16lo(a[u].z), lo[5] += bf16hi(a[u].z), lo[6] += bf16lo(a[u].w), lo[7] += bf16hi(a[u].w);
                hi[0] += bf16lo(c[u].x), hi[1] += bf16hi(c[u].x), hi[2] += bf16lo(c[u].y), hi[3] += bf16hi(c[u].y);
                hi[4] += bf16lo(c[u].z), hi[5] += bf16hi(c[u].z), hi[6] += bf16lo(c[u].w), hi[7] += bf16hi(c[u].w);
            }
        }
    }
    if (rope_cos) {
        const int64_t prow = sv.pos0(b, pos_batch_stride) + row;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            f32x4 lo4 = {lo[4 * h], lo[4 * h + 1], lo[4 * h + 2], lo[4 * h + 3]};
            f32x4 hi4 = {hi[4 * h], hi[4 * h + 1], hi[4 * h + 2], hi[4 * h + 3]};
            rope_rotate4<true>(lo4, hi4, rope_cos + prow * 32, rope_sin + prow * 32, 8 * j + 4 * h);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                lo[4 * h + r] = lo4[r];
                hi[4 * h + r] = hi4[r];
            }
        }
    }
    uint16_t* drow = dqkv + (sv.row0 + row) * ((int64_t)3 * nh * 64) + head * 64 + 8 * j;
    gstore16<(CM3P_NT & 8) != 0>(drow, uint4{pack_bf16x2(lo[0] * scale, lo[1] * scale), pack_bf16x2(lo[2] * scale, lo[3] * scale),
                                             pack_bf16x2(lo[4] * scale, lo[5] * scale), pack_bf16x2(lo[6] * scale, lo[7] * scale)});
    gstore16<(CM3P_NT & 8) != 0>(drow + 32, uint4{pack_bf16x2(hi[0] * scale, hi[1] * scale), pack_bf16x2(hi[2] * scale, hi[3] * scale),
                                                  pack_bf16x2(hi[4] * scale, hi[5] * scale), pack_bf16x2(hi[6] * scale, hi[7] * scale)});
}

// the reduction as its own launch: grid (tiles, nh, B), 256 threads
__global__ __launch_bounds__(256) void attn_bwd_dq_reduce_kernel(const uint16_t* __restrict__ dq_part, uint16_t* __restrict__ dqkv, int Smax,
                                                                 int nh, float scale, const float* __restrict__ rope_cos,
                                                                 const float* __restrict__ rope_sin, int64_t pos_batch_stride, VarLen vl, int G) {
    const int head = blockIdx.y, b = blockIdx.z;
    const SeqView sv(vl, b, head, Smax, nh);
    dq_reduce_tile(dq_part, dqkv, blockIdx.x, head, b, threadIdx.x, sv, Smax, nh, scale, rope_cos, rope_sin, pos_batch_stride, G);
}

// ---- fused ----------------------------------------------------------------------------------------------------------------------
template <bool PRE, bool ADD>
__global__ __launch_bounds__(256, 1) void attn_bwd_fused_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ d_o,
                                                                const float* __restrict__ stat_ws, uint16_t* __restrict__ dq_part,
                                                                uint16_t* __restrict__ dqkv, const uint8_t* __restrict__ kmask, int Smax,
                                                                int nh, float scale, const float* __restrict__ rope_cos,
                                                                const float* __restrict__ rope_sin, int64_t pos_batch_stride, VarLen vl, int G,
                                                                int phase) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, hh = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);  // (tells the compiler that everything derived from it is wave-uniform)
    const int nkb = (Smax + 255) / 256, NT = (Smax + 63) / 64;
    // this launch's key blocks: G k + phase.  Phase 0 (!ADD) stores its dQ partial to slab k, the phases 1 .. G - 1 (ADD, one launch
    // each, stream-ordered) add theirs to the same slab
    const int nblk = (nkb - phase + G - 1) / G;
    int kblk, head, b;
    decode_block(nblk, nh, kblk, head, b);
    const int slab_idx = kblk;
    kblk = G * kblk + phase;
    const int K0 = kblk * 256;
    const SeqView sv(vl, b, head, Smax, nh);
    const int S = sv.S;
    if (K0 >= S) return;
    const int k0 = K0 + wid * 64;
    const int64_t ld = (int64_t)3 * nh * 64;
    const uint16_t* qbase = qkv + sv.row0 * ld + head * 64;
    const uint16_t* kbase = qbase + nh * 64;
    const uint16_t* vbase = qbase + 2 * nh * 64;
    const int64_t ldo = (int64_t)nh * 64;
    const uint16_t* dobase = d_o + sv.row0 * ldo + head * 64;
    const float cm = scale * kLog2e;
    const int l31 = lane & 31, g4 = lane >> 4, i16 = lane & 15;
    const int n_tiles = (S + 63) / 64;

    // ---- LDS-DMA of a tile: wave w brings rows 16 w .. 16 w + 15 of Q and of dO (four 1-KiB pieces: lane l's 16 bytes land at
    // piece + 16 l, i.e. row l >> 3, chunk slot l & 7, which must hold global chunk (l & 7) ^ swz(row)) and, waves 0 / 1, the
    // tile's two statistics rows (waves 2 / 3 repeat them: every wave issues the same number of vector-memory operations, which
    // is what makes the counted vmcnt waits exact).  One m0 write serves the four pieces: the instruction offset moves the LDS
    // address AND the global address, so piece i's source offset is kept i KiB low (and the scalar base 3 KiB low, offsets >= 0).
    const int prow0 = 16 * wid + (lane >> 3), prow1 = prow0 + 8;
    const int pc0 = (((lane & 7) ^ swz(prow0)) << 4), pc1 = (((lane & 7) ^ swz(prow1)) << 4);
    const int ldb = (int)ld * 2, ldob = (int)ldo * 2;
    const char* q_dma = reinterpret_cast<const char*>(qbase) - 3072;
    const char* do_dma = reinterpret_cast<const char*>(dobase) - 3072;
    const float* stat_bh = stat_ws + ((int64_t)b * nh + head) * (NT + kFPadTiles) * 128;
    const uint32_t stat_l = (uint32_t)(((wid & 1) * 64 + lane) * 4);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const uint32_t m0_tile = __builtin_amdgcn_readfirstlane(lds0 + 4096u * wid);
    const uint32_t m0_stat = __builtin_amdgcn_readfirstlane(lds0 + 16384u + 256u * (wid & 1));
    uint32_t dv0, dv1, dv2, dv3, dvs;  // source offsets of the tile whose DMA is issued next
    auto dma_addr = [&](int t) {
        const int r0 = min(t * 64 + prow0, S - 1), r1 = min(t * 64 + prow1, S - 1);
        dv0 = (uint32_t)(r0 * ldb + (pc0 + 3072));
        dv1 = (uint32_t)(r1 * ldb + (pc1 + 2048));
        dv2 = (uint32_t)(r0 * ldob + (pc0 + 1024));
        dv3 = (uint32_t)(r1 * ldob + pc1);
        dvs = (uint32_t)t * 512u + stat_l;
    };
    auto dma_rows = [&](int slot) {
        const uint32_t m0v = m0_tile + (uint32_t)slot * kFStage;
        CM3P_AUDIT(CM3P_AUD_T0, q_dma + dv0, 16);  // (bases are kept 3 KiB low, source offsets (3 - piece) KiB high, instruction offsets piece KiB)
        CM3P_AUDIT(CM3P_AUD_T0, q_dma + dv1 + 1024, 16);
        CM3P_AUDIT(CM3P_AUD_T1, do_dma + dv2 + 2048, 16);
        CM3P_AUDIT(CM3P_AUD_T1, do_dma + dv3 + 3072, 16);
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, %5\n\t"
                     "global_load_lds_dwordx4 %2, %5 offset:1024\n\t"
                     "global_load_lds_dwordx4 %3, %6 offset:2048\n\t"
                     "global_load_lds_dwordx4 %4, %6 offset:3072" ::"s"(m0v), "v"(dv0), "v"(dv1), "v"(dv2), "v"(dv3), "s"(q_dma), "s"(do_dma)
                     : "memory", "m0");
    };
    auto dma_stat = [&](int slot) {
        const uint32_t m0v = m0_stat + (uint32_t)slot * kFStage;
        CM3P_AUDIT(CM3P_AUD_S0, reinterpret_cast<const char*>(stat_bh) + dvs, 4);
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2" ::"s"(m0v), "v"(dvs), "s"(stat_bh) : "memory", "m0");
    };

    // tiles 0 .. 2 are requested before the wave's own K / V rows: the two round trips overlap
    dma_addr(0);
    dma_rows(0);
    dma_stat(0);
    dma_addr(1);
    dma_rows(1);
    dma_stat(1);
    dma_addr(2);
    dma_rows(2);
    dma_stat(2);

    // ---- the wave's 64 keys: K / V row fragments (B operands of the score products, AGPRs) and its part of the K image
    bf16x8 kf[2][4], vf[2][4];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        const int krow_c = min(k0 + 32 * kb + l31, S - 1);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            kf[kb][s] = gload_frag8(kbase + (int64_t)krow_c * ld + 16 * s + 8 * hh);
            vf[kb][s] = gload_frag8(vbase + (int64_t)krow_c * ld + 16 * s + 8 * hh);
        }
    }
    {
        const uint8_t* km = kmask ? kmask + sv.row0 : nullptr;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = 8 * i + (lane >> 3), c = lane & 7;
            const int krow = k0 + row, rc = min(krow, S - 1);
            uint4 v = *reinterpret_cast<const uint4*>(kbase + (int64_t)rc * ld + c * 8);
            const bool vis = krow < S && (km == nullptr || km[rc] != 0);
            if (!vis) v = uint4{0u, 0u, 0u, 0u};  // a key no query may see: its dS column meets a zero row in the dQ product
            *reinterpret_cast<uint4*>(smem + kFKimg + off_R(64 * wid + row, c)) = v;
        }
    }
    f32x16 dk[2][2], dv[2][2];  // [d block][key block]
    f32x4 dkq[kQuads ? 2 : 1][2][2], dvq[kQuads ? 2 : 1][2][2];  // (timing build only: [d block][key block][quad])
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            if constexpr (kQuads) {
                dkq[db][kb][0] = dkq[db][kb][1] = dvq[db][kb][0] = dvq[db][kb][1] = f32x4{0.f, 0.f, 0.f, 0.f};
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) dk[db][kb][i] = dv[db][kb][i] = 0.f;
            }
        }
#define CM3P_GRAD_MFMA(acc, accq, DB, KB, A, B)                              \
    do {                                                                     \
        if constexpr (kQuads) {                                              \
            accq[DB][KB][0] = mfma16(A, B, accq[DB][KB][0]);                 \
            accq[DB][KB][1] = mfma16(A, B, accq[DB][KB][1]);                 \
        } else {                                                             \
            acc[DB][KB] = mfma32(A, B, acc[DB][KB]);                         \
        }                                                                    \
    } while (0)

    // ---- per-lane LDS byte offsets; everything else is an immediate
    int oR[4], oTlo[2], oThi[2];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) oR[s4] = (l31 >> 4) * 4096 + (l31 & 15) * 128 + (((2 * s4 + hh) ^ swz(l31)) << 4);
#pragma unroll
    for (int db = 0; db < 2; ++db) {
        const int row = 4 * hh + (i16 >> 2), col = 32 * db + 16 * (g4 & 1) + 4 * (i16 & 3);
        oTlo[db] = off_T(row, col);
        oThi[db] = off_T(row + 8, col);
    }
    const int oI = 16384 + 16 * hh;
    // dQ product: wave w owns d block w >> 1 and query block w & 1 of the epoch
    const int dq_db = wid >> 1, dq_qb = wid & 1;
    int oKlo, oKhi, oDlo, oDhi;
    {
        const int row = 4 * hh + (i16 >> 2), cg = 16 * (g4 & 1) + 4 * (i16 & 3);
        oKlo = kFKimg + off_T(row, 32 * dq_db + cg);
        oKhi = kFKimg + off_T(row + 8, 32 * dq_db + cg);
        oDlo = kFdS + off_T(row, 32 * dq_qb + cg);
        oDhi = kFdS + off_T(row + 8, 32 * dq_qb + cg);
    }
    // dS image writes: row = the lane's key, eight possible 16-byte chunks (4 bf16 of one accumulator group = half a chunk)
    int dsW[8];
#pragma unroll
    for (int ci = 0; ci < 8; ++ci) dsW[ci] = kFdS + (64 * wid + l31) * 128 + ((ci ^ swz(l31)) << 4) + 8 * hh;
    // The region bases are part of these registers ON PURPOSE and hidden from the compiler: it would otherwise pull the constants
    // back out, find base + k-step + image offsets beyond the 16-bit DS offset field, and keep dozens of precomputed addresses
    // (spilled) instead of one register plus an immediate.
    asm volatile("" : "+v"(oKlo), "+v"(oKhi), "+v"(oDlo), "+v"(oDhi));
#pragma unroll
    for (int ci = 0; ci < 8; ++ci) asm volatile("" : "+v"(dsW[ci]));

    // register-resident LDS fragments of the current 32-query block
    bf16x8 Qf[4], Gf[4];        // rows of Q / dO (A operands of the score products)
    bf16x8 gT[2][2], qT[2][2];  // [sp][db]: dO^T / Q^T (A operands of the gradient products)
    f32x16 isv, idv;            // score offsets and -delta of the block's 32 rows, in accumulator layout
    bf16x8 dqB0, dqB1;          // dS^T operands of the next two dQ MFMAs
    constexpr int NKA = PRE ? 15 : 11;  // (the !PRE variant needs a few more registers for its score multiplies)
    bf16x8 ktA[NKA];            // K^T operands of k-steps 0 .. NKA-1 of the dQ product (AGPRs: 192 + 60 of the 256)
    bf16x8 ktV[2];              // ... and of the last k-steps, re-read from the K image every tile
    f32x16 dq;                  // dQ^T block of the current epoch
    f32x4 dqq[2];               // (timing build only: its two quads)
    uint4 zr0, zr1;             // the finished epoch's block, row-major: 16 bytes of rows lane >> 2 and 16 + (lane >> 2)
    auto load_init = [&](const char* sq, f32x16& v, int which, int half) {
#pragma unroll
        for (int g = 2 * half; g < 2 * half + 2; ++g) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(sq + oI + 256 * which + 32 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[4 * g + r] = a[r];
        }
    };
    // sq = slot + 8192 * qb (row fragments: Q at +0, dO at +2048 of each 16-row group); sqi = slot + 128 * qb
    auto loadS_all = [&](const char* sq, const char* sqi) {
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            Qf[s4] = ld_frag(sq + oR[s4]);
            Gf[s4] = ld_frag(sq + 2048 + oR[s4]);
        }
        load_init(sqi, isv, 0, 0);
        load_init(sqi, isv, 0, 1);
        load_init(sqi, idv, 1, 0);
        load_init(sqi, idv, 1, 1);
    };
    auto loadG_one = [&](const char* sq, int sp, int db, int which) {  // which: 0 = dO^T, 1 = Q^T
        const char* base = sq + (which == 0 ? 2048 : 0) + 4096 * sp;
        const bf16x8 f = ld_fragT(base + oTlo[db], base + oThi[db]);
        if (which == 0) gT[sp][db] = f;
        else qT[sp][db] = f;
    };
    auto load_dq_operand = [&](bf16x8& bq, int rbuf, int ks) {
        if constexpr ((CM3P_FABL & 8) != 0) return;
        bq = ld_fragT(smem + oDlo + 32768 * rbuf + 2048 * ks, smem + oDhi + 32768 * rbuf + 2048 * ks);
        if (ks >= NKA) ktV[ks & 1] = ld_fragT(smem + oKlo + 2048 * ks, smem + oKhi + 2048 * ks);
    };
    auto dq_mfma = [&](auto ks_c, auto first_c, const bf16x8& bq) {
        constexpr int KS = decltype(ks_c)::value;
        if constexpr ((CM3P_FABL & 32) != 0) return;
        if constexpr (kQuads) {
            if constexpr (decltype(first_c)::value != 0) {
                mfma16_av0(dqq[0], ktA[KS], bq);
                mfma16_av0(dqq[1], ktA[KS], bq);
            } else if constexpr (KS >= NKA) {
                mfma16_vva(dqq[0], ktV[KS & 1], bq);
                mfma16_vva(dqq[1], ktV[KS & 1], bq);
            } else {
                mfma16_ava(dqq[0], ktA[KS], bq);
                mfma16_ava(dqq[1], ktA[KS], bq);
            }
            return;
        }
        if constexpr (decltype(first_c)::value != 0) mfma_av0(dq, ktA[KS], bq);
        else if constexpr (KS >= NKA) mfma_vva(dq, ktV[KS & 1], bq);
        else mfma_ava(dq, ktA[KS], bq);
    };

    // One pipeline step (20 MFMAs): scores of block Y (key block KBY) next to the exponentials of block X, the gradient products of
    // block X (key block KBX), whose dS also goes to image WBUF (query-column chunks WCB ..), and four k-steps KS0 .. KS0+3 of the
    // dQ product of the epoch in image RBUF (FIRST: the epoch's first k-step starts from zero).  Fragment reloads as in
    // attn_bwd_dkv3_kernel; `hook(c)` is called once per chunk (0..15: behind the chunk's MFMA; 16..19: behind the dQ MFMAs).
    auto to_frag = [&](const f32x16& a, int sp) -> bf16x8 {
        if constexpr ((CM3P_FABL & 256) != 0) {  // (timing only: a non-zero constant - zero operands lower the matrix pipe's power)
            (void)a;
            return __builtin_bit_cast(bf16x8, cm3p_u32x4{0x3e003d80u + 0x00010001u * (lane & 63), 0x3d903e10u, 0x3e203da0u + (unsigned)sp, 0x3db03e30u});
        } else {
            return acc_to_frag(a, sp);
        }
    };
    auto step = [&](auto kbx_c, auto kby_c, auto wbuf_c, auto wcb_c, auto rbuf_c, auto ks0_c, auto first_c, auto nrbuf_c, auto nks0_c, f32x16& Xs, f32x16& Xdp,
                    f32x16& Ys, f32x16& Ydp, const char* nS, const char* nSi, const char* nG, auto&& hook) {
        constexpr int KBX = decltype(kbx_c)::value, KBY = decltype(kby_c)::value, WBUF = decltype(wbuf_c)::value;
        constexpr int WCB = decltype(wcb_c)::value, RBUF = decltype(rbuf_c)::value, KS0 = decltype(ks0_c)::value;
        constexpr bool FIRST = decltype(first_c)::value != 0;
        constexpr int NRBUF = decltype(nrbuf_c)::value, NKS0 = decltype(nks0_c)::value;  // the next step's first k-step (-1: none yet)
        char* const dsw = smem + 32768 * WBUF + 4096 * KBX;
#define CM3P_HOOK(C) hook(std::integral_constant<int, (C)>{})
        CM3P_SB();
        mfma_vc(Ys, Qf[0], kf[KBY][0], isv);
        exp2c_pair<PRE>(Xs, 0, cm);
        if constexpr (FIRST) load_dq_operand(dqB0, RBUF, KS0);  // (the epoch was published a moment ago)
        load_dq_operand(dqB1, RBUF, KS0 + 1);
        CM3P_HOOK(0);
        CM3P_SB();
        mfma_vc(Ydp, Gf[0], vf[KBY][0], idv);
        exp2c_pair<PRE>(Xs, 2, cm);
        if constexpr (KBY == 1 && (CM3P_FABL & 512) == 0) {
            Qf[0] = ld_frag(nS + oR[0]);
            Gf[0] = ld_frag(nS + 2048 + oR[0]);
            load_init(nSi, isv, 0, 0);
        }
        CM3P_HOOK(1);
        CM3P_SB();
        mfma_va(Ys, Qf[1], kf[KBY][1]);
        exp2c_pair<PRE>(Xs, 4, cm);
        if constexpr (KBY == 1 && (CM3P_FABL & 512) == 0) {
            load_init(nSi, isv, 0, 1);
            Qf[1] = ld_frag(nS + oR[1]);
        }
        CM3P_HOOK(2);
        CM3P_SB();
        mfma_va(Ydp, Gf[1], vf[KBY][1]);
        exp2c_pair<PRE>(Xs, 6, cm);
        if constexpr (KBY == 1 && (CM3P_FABL & 512) == 0) {
            Gf[1] = ld_frag(nS + 2048 + oR[1]);
            load_init(nSi, idv, 1, 0);
        }
        CM3P_HOOK(3);
        CM3P_SB();
        dq_mfma(std::integral_constant<int, KS0>{}, first_c, dqB0);
        CM3P_HOOK(16);
        CM3P_SB();
        mfma_va(Ys, Qf[2], kf[KBY][2]);
        exp2c_pair<PRE>(Xs, 8, cm);
        load_dq_operand(dqB0, RBUF, KS0 + 2);
        if constexpr (KBY == 1 && (CM3P_FABL & 512) == 0) {
            Qf[2] = ld_frag(nS + oR[2]);
            load_init(nSi, idv, 1, 1);
        }
        CM3P_HOOK(4);
        CM3P_SB();
        mfma_va(Ydp, Gf[2], vf[KBY][2]);
        exp2c_pair<PRE>(Xs, 10, cm);
        if constexpr (KBY == 1 && (CM3P_FABL & 512) == 0) Gf[2] = ld_frag(nS + 2048 + oR[2]);
        CM3P_HOOK(5);
        CM3P_SB();
        mfma_va(Ys, Qf[3], kf[KBY][3]);
        exp2c_pair<PRE>(Xs, 12, cm);
        if constexpr (KBY == 1 && (CM3P_FABL & 512) == 0) Qf[3] = ld_frag(nS + oR[3]);
        CM3P_HOOK(6);
        CM3P_SB();
        mfma_va(Ydp, Gf[3], vf[KBY][3]);
        exp2c_pair<PRE>(Xs, 14, cm);
        if constexpr (KBY == 1 && (CM3P_FABL & 512) == 0) Gf[3] = ld_frag(nS + 2048 + oR[3]);
        CM3P_HOOK(7);
        CM3P_SB();
        dq_mfma(std::integral_constant<int, KS0 + 1>{}, std::integral_constant<int, 0>{}, dqB1);
        CM3P_HOOK(17);
        CM3P_SB();
        // ---- gradient products of X: dV^T += dO^T P, dK^T += Q^T dS; dS -> image
        const bf16x8 pf0 = to_frag(Xs, 0);
        CM3P_GRAD_MFMA(dv, dvq, 0, KBX, gT[0][0], pf0);
        if constexpr (KBX == 1 && (CM3P_FABL & 1024) == 0) loadG_one(nG, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < ((CM3P_FABL & 128) ? 0 : 4); ++i) Xdp[i] *= Xs[i];
        load_dq_operand(dqB1, RBUF, KS0 + 3);
        CM3P_HOOK(8);
        CM3P_SB();
        CM3P_GRAD_MFMA(dv, dvq, 1, KBX, gT[0][1], pf0);
        if constexpr (KBX == 1 && (CM3P_FABL & 1024) == 0) loadG_one(nG, 0, 1, 0);
#pragma unroll
        for (int i = 4; i < ((CM3P_FABL & 128) ? 0 : 8); ++i) Xdp[i] *= Xs[i];
        const bf16x8 ds0 = to_frag(Xdp, 0);
        CM3P_HOOK(9);
        CM3P_SB();
        CM3P_GRAD_MFMA(dk, dkq, 0, KBX, qT[0][0], ds0);
        if constexpr (KBX == 1 && (CM3P_FABL & 1024) == 0) loadG_one(nG, 0, 0, 1);
        const bf16x8 pf1 = to_frag(Xs, 1);
        if constexpr ((CM3P_FABL & 4) == 0) {
            const uint4 w = __builtin_bit_cast(uint4, ds0);
            *reinterpret_cast<uint2*>(dsw + dsW[WCB + 0]) = uint2{w.x, w.y};
            *reinterpret_cast<uint2*>(dsw + dsW[WCB + 1]) = uint2{w.z, w.w};
        }
        CM3P_HOOK(10);
        CM3P_SB();
        CM3P_GRAD_MFMA(dk, dkq, 1, KBX, qT[0][1], ds0);
        if constexpr (KBX == 1 && (CM3P_FABL & 1024) == 0) loadG_one(nG, 0, 1, 1);
#pragma unroll
        for (int i = 8; i < ((CM3P_FABL & 128) ? 0 : 12); ++i) Xdp[i] *= Xs[i];
        CM3P_HOOK(11);
        CM3P_SB();
        dq_mfma(std::integral_constant<int, KS0 + 2>{}, std::integral_constant<int, 0>{}, dqB0);
        CM3P_HOOK(18);
        CM3P_SB();
        CM3P_GRAD_MFMA(dv, dvq, 0, KBX, gT[1][0], pf1);
        if constexpr (KBX == 1 && (CM3P_FABL & 1024) == 0) loadG_one(nG, 1, 0, 0);
#pragma unroll
        for (int i = 12; i < ((CM3P_FABL & 128) ? 0 : 16); ++i) Xdp[i] *= Xs[i];
        const bf16x8 ds1 = to_frag(Xdp, 1);
        if constexpr (NKS0 >= 0) load_dq_operand(dqB0, NRBUF, NKS0);
        CM3P_HOOK(12);
        CM3P_SB();
        CM3P_GRAD_MFMA(dv, dvq, 1, KBX, gT[1][1], pf1);
        if constexpr (KBX == 1 && (CM3P_FABL & 1024) == 0) loadG_one(nG, 1, 1, 0);
        if constexpr ((CM3P_FABL & 4) == 0) {
            const uint4 w = __builtin_bit_cast(uint4, ds1);
            *reinterpret_cast<uint2*>(dsw + dsW[WCB + 2]) = uint2{w.x, w.y};
            *reinterpret_cast<uint2*>(dsw + dsW[WCB + 3]) = uint2{w.z, w.w};
        }
        CM3P_HOOK(13);
        CM3P_SB();
        CM3P_GRAD_MFMA(dk, dkq, 0, KBX, qT[1][0], ds1);
        if constexpr (KBX == 1 && (CM3P_FABL & 1024) == 0) loadG_one(nG, 1, 0, 1);
        CM3P_HOOK(14);
        CM3P_SB();
        CM3P_GRAD_MFMA(dk, dkq, 1, KBX, qT[1][1], ds1);
        if constexpr (KBX == 1 && (CM3P_FABL & 1024) == 0) loadG_one(nG, 1, 1, 1);
        CM3P_HOOK(15);
        CM3P_SB();
        dq_mfma(std::integral_constant<int, KS0 + 3>{}, std::integral_constant<int, 0>{}, dqB1);
        CM3P_HOOK(19);
        CM3P_SB();
#undef CM3P_HOOK
    };

    f32x16 sA, dpA, sB, dpB;
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I4 = std::integral_constant<int, 4>;
    using I8 = std::integral_constant<int, 8>;
    using I12 = std::integral_constant<int, 12>;
    // this key block's slab: rows = query rows, 64 * NT real rows + 64 dump rows
    uint16_t* const slab_u = dq_part + (((int64_t)b * nh + head) * fused_slabs(Smax, G) + slab_idx) * ((int64_t)fused_slab_rows(Smax) * 64) + 32 * dq_db;
    uint16_t* const slab = slab_u + (lane >> 2) * 64 + 8 * (lane & 3);
    uint16_t* srow = slab;
    // ADD: one dword per lane and instruction - lane l adds dword l & 15 of row 4 j + (l >> 4) of the block, j = 0 .. 7, i.e. four
    // whole 64-byte row pieces per instruction (what the memory side's 64-byte atomic requests are made of)
    const uint16_t* srow_u = slab_u;  // wave-uniform row base of the block (scalar registers)
    uint32_t add_off = (uint32_t)((lane >> 4) * 128 + (lane & 15) * 4);
    int zRd = kFZ + 2560 * wid + (lane >> 4) * 80 + (lane & 15) * 4;
    asm volatile("" : "+v"(add_off), "+v"(zRd));
    // the wave's private transposition buffer: written in accumulator layout (lane = query row, 4 head dims per 8-byte store),
    // read back as 16-byte pieces of whole rows (80-byte row pitch: 16-byte aligned reads, 2-way conflicts on the writes)
    int zW = kFZ + 2560 * wid + l31 * 80 + 8 * hh, zR = kFZ + 2560 * wid + (lane >> 2) * 80 + (lane & 3) * 16;
    asm volatile("" : "+v"(zW), "+v"(zR));

    // One tile (ring slot SL, dS image DP = t & 1).  Steps 0 / 1 finish epoch t (second half of tile t-1, first half of tile t) into
    // image DP while the dQ product walks epoch t-1 in image DP ^ 1; the barrier publishes epoch t and tile t+1 (whose DMA each
    // wave has waited for) and retires epoch t-1 and slot SL; steps 2 / 3 start epoch t+1 in image DP ^ 1 and the dQ product of
    // epoch t, store epoch t-1's block and issue the DMA of tile t+3 into slot SL.
#if CM3P_FTRACE
    unsigned long long ft_acc[2] = {0, 0};
#endif
    auto tile = [&](auto slot_c, auto par_c, int t) {
        constexpr int SL = decltype(slot_c)::value, NS = (SL + 1) % kFSlots, DP = decltype(par_c)::value;
        using WA = std::integral_constant<int, DP>;
        using WB = std::integral_constant<int, DP ^ 1>;
        using N1 = std::integral_constant<int, -1>;
        const char* st = smem + SL * kFStage;
        const char* nst = smem + NS * kFStage;
        auto no_hook = [&](auto) {};
        auto pre_hook = [&](auto c) {
            constexpr int C = decltype(c)::value;
            if constexpr (C == 16) dma_addr(t + 3);
        };
        // Behind the barrier: the DMA of tile t+3 first (slot SL was retired a moment ago), then epoch t-1's finished dQ^T block:
        // accumulators -> bf16 -> the wave's transposition buffer -> two 16-byte pieces of whole rows per lane -> two coalesced
        // stores (16 rows x 64 contiguous bytes each).  DMA before stores: vmcnt retires in issue order, and a store's
        // acknowledgement takes far longer than a tile period.
        auto post_hook = [&](auto c) {
            constexpr int C = decltype(c)::value;
            if constexpr (C == 0 && (CM3P_FABL & 2) == 0) dma_rows(SL);
            if constexpr (C == 1 && (CM3P_FABL & 2) == 0) dma_stat(SL);
            if constexpr (C == 2 || C == 3) {
#pragma unroll
                for (int g = 2 * (C - 2); g < 2 * (C - 2) + 2; ++g) {
                    if constexpr (kQuads)
                        *reinterpret_cast<uint2*>(smem + zW + 16 * g) =
                            uint2{pack_bf16x2(dqq[g & 1][0], dqq[g & 1][1]), pack_bf16x2(dqq[g & 1][2], dqq[g & 1][3])};
                    else
                        *reinterpret_cast<uint2*>(smem + zW + 16 * g) =
                            uint2{pack_bf16x2(dq[4 * g], dq[4 * g + 1]), pack_bf16x2(dq[4 * g + 2], dq[4 * g + 3])};
                }
            }
            if constexpr (C == 6) {
                asm volatile("" ::: "memory");  // (the uint2 stores above and these uint4 loads do not alias by type: keep their order)
                if constexpr (ADD) {
                    zr0 = uint4{*reinterpret_cast<const uint32_t*>(smem + zRd), *reinterpret_cast<const uint32_t*>(smem + zRd + 320),
                                *reinterpret_cast<const uint32_t*>(smem + zRd + 640), *reinterpret_cast<const uint32_t*>(smem + zRd + 960)};
                    zr1 = uint4{*reinterpret_cast<const uint32_t*>(smem + zRd + 1280), *reinterpret_cast<const uint32_t*>(smem + zRd + 1600),
                                *reinterpret_cast<const uint32_t*>(smem + zRd + 1920), *reinterpret_cast<const uint32_t*>(smem + zRd + 2240)};
                } else {
                    zr0 = *reinterpret_cast<const uint4*>(smem + zR);
                    zr1 = *reinterpret_cast<const uint4*>(smem + zR + 16 * 80);
                }
            }
            if constexpr (C == 17) {
                // rows of epoch t-1's block: 64 (t-1) - 32 + 32 qb ..; blocks outside the sequence's tiles go to the dump rows
                const int r = 64 * t - 96 + 32 * dq_qb;
                const int rr = (r >= 0 && r < 64 * n_tiles) ? r : 64 * NT + 32 * dq_qb;
                srow = slab + (int64_t)rr * 64;
                srow_u = slab_u + (int64_t)rr * 64;
            }
            if constexpr (C == 9 && (CM3P_FABL & 16) == 0) {
                if constexpr (ADD) {
                    const uint32_t ao = add_off;  // (copies: an asm operand inside a generic lambda does not capture by itself)
                    const uint16_t* su = srow_u;
                    const uint4 a0 = zr0, a1 = zr1;
                    asm volatile("global_atomic_pk_add_bf16 %0, %1, %9\n\t"
                                 "global_atomic_pk_add_bf16 %0, %2, %9 offset:512\n\t"
                                 "global_atomic_pk_add_bf16 %0, %3, %9 offset:1024\n\t"
                                 "global_atomic_pk_add_bf16 %0, %4, %9 offset:1536\n\t"
                                 "global_atomic_pk_add_bf16 %0, %5, %9 offset:2048\n\t"
                                 "global_atomic_pk_add_bf16 %0, %6, %9 offset:2560\n\t"
                                 "global_atomic_pk_add_bf16 %0, %7, %9 offset:3072\n\t"
                                 "global_atomic_pk_add_bf16 %0, %8, %9 offset:3584" ::"v"(ao), "v"(a0.x), "v"(a0.y), "v"(a0.z),
                                 "v"(a0.w), "v"(a1.x), "v"(a1.y), "v"(a1.z), "v"(a1.w), "s"(su)
                                 : "memory");
                } else {
                    gstore16<(CM3P_NT & 2) != 0>(srow, zr0);
                    gstore16<(CM3P_NT & 2) != 0>(srow + 16 * 64, zr1);
                }
            }
        };
        step(I0{}, I1{}, WA{}, I4{}, WB{}, I8{}, I0{}, WB{}, I12{}, sA, dpA, sB, dpB, st + 8192, st + 128, nullptr, no_hook);
        step(I1{}, I0{}, WA{}, I4{}, WB{}, I12{}, I0{}, N1{}, N1{}, sB, dpB, sA, dpA, nullptr, nullptr, st + 8192, pre_hook);
        // Tile t+1 has landed: vmcnt retires in issue order and the only vector-memory operations issued after tile t+1's DMA are
        // tile t-1's two stores, tile t+2's five DMAs and tile t's two stores (every wave issues exactly these, unconditionally).
        // (ADD: eight atomics per tile where the other instance has two stores - 8 + 5 + 8)
#if CM3P_FTRACE
        const unsigned long long ft1 = __builtin_amdgcn_s_memtime();
#endif
        if constexpr (ADD) asm volatile("s_waitcnt vmcnt(21)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
#if CM3P_FTRACE
        const unsigned long long ft2 = __builtin_amdgcn_s_memtime();
#endif
        if constexpr ((CM3P_FABL & 1) == 0) lds_barrier();
#if CM3P_FTRACE
        const unsigned long long ft3 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        ft_acc[0] += ft2 - ft1;
        ft_acc[1] += ft3 - ft2;
#endif
        step(I0{}, I1{}, WB{}, I0{}, WA{}, I0{}, I1{}, WA{}, I4{}, sA, dpA, sB, dpB, nst, nst, nullptr, post_hook);
        step(I1{}, I0{}, WB{}, I0{}, WA{}, I4{}, I0{}, WA{}, I8{}, sB, dpB, sA, dpA, nullptr, nullptr, nst, no_hook);
    };

    // prologue: (tiles 0 .. 2 are in flight since the top of the kernel) fragments of (tile 0, qb0), scores of its first block
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // all three (the counted wait of the loop assumes its steady-state issue order)
    lds_barrier();
    // K^T of the workgroup's 256 keys, the wave's d block: the A operands of all 16 k-steps of the dQ product stay in AGPRs (the
    // K image is little more than a transposition buffer: per-wave LDS throughput is bounded by the 15 operations lgkmcnt lets a wave keep in
    // flight, and these would be 32 more reads per tile)
#pragma unroll
    for (int ks = 0; ks < NKA; ++ks) {
        ktA[ks] = ld_fragT(smem + oKlo + 2048 * ks, smem + oKhi + 2048 * ks);
        asm volatile("" : "+a"(ktA[ks]));
    }
    loadS_all(smem, smem);
#pragma unroll
    for (int sp = 0; sp < 2; ++sp)
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            loadG_one(smem, sp, db, 0);
            loadG_one(smem, sp, db, 1);
        }
    mfma_vc(sA, Qf[0], kf[0][0], isv);
    mfma_vc(dpA, Gf[0], vf[0][0], idv);
#pragma unroll
    for (int s4 = 1; s4 < 4; ++s4) {
        mfma_va(sA, Qf[s4], kf[0][s4]);
        mfma_va(dpA, Gf[s4], vf[0][s4]);
    }
    // (epoch -1 does not exist: the dQ product of tile 0's first two steps runs on whatever the image holds and goes to the dump rows)
    if constexpr (kQuads) {
        dqq[0] = dqq[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) dq[i] = 0.f;
    }
    load_dq_operand(dqB0, 1, 8);
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // (the only place a result is read right behind its asm MFMA chain)
    CM3P_SB();

    // No exit edge inside the unrolled ring (the accumulators never change registers): the tiles past the sequence re-read its last
    // rows with -inf score offsets (p = 0: exact zeros everywhere).  Epoch n_tiles (the second half of the last tile) is stored
    // in tile n_tiles + 1, hence the bound.
#if CM3P_FTRACE
    const unsigned long long ft0 = __builtin_amdgcn_s_memtime();
#endif
    for (int t = 0; t < n_tiles + 2; t += kFUnroll) {
        tile(std::integral_constant<int, 0>{}, I0{}, t);
        tile(std::integral_constant<int, 1>{}, I1{}, t + 1);
        tile(std::integral_constant<int, 2>{}, I0{}, t + 2);
        tile(std::integral_constant<int, 0>{}, I1{}, t + 3);
        tile(std::integral_constant<int, 1>{}, I0{}, t + 4);
        tile(std::integral_constant<int, 2>{}, I1{}, t + 5);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA may outlive the workgroup's LDS allocation
#if CM3P_FTRACE
    if (g_bwdf_trace && lane == 0) {
        unsigned long long* o = g_bwdf_trace + (size_t)blockIdx.x * 16 + wid * 4;  // (1-D grid: decode_block)
        o[0] = ft_acc[0];
        o[1] = ft_acc[1];
        o[2] = __builtin_amdgcn_s_memtime() - ft0;
        o[3] = (unsigned long long)(n_tiles + 2 + kFUnroll - 1) / kFUnroll * kFUnroll;
    }
#endif


    // ---- epilogue: dK = scale * dK^T acc (inverse rotary applied), dV; keys under the padding mask get zeros
    if constexpr (kQuads) {
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    dk[db][kb][i] = dkq[db][kb][(i >> 2) & 1][i & 3];
                    dv[db][kb][i] = dvq[db][kb][(i >> 2) & 1][i & 3];
                }
    }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        const int krow = k0 + 32 * kb + l31;
        if (krow < S) {
            const bool ok = kmask ? kmask[sv.row0 + krow] != 0 : true;
            // PRE: the products were taken with q * scale * log2(e), so dK = ln(2) * accumulator
            const float ks = ok ? (PRE ? 0.69314718055994531f : scale) : 0.f, vs = ok ? 1.f : 0.f;
            uint16_t* dkrow = dqkv + (sv.row0 + krow) * ld + nh * 64 + head * 64;
            uint16_t* dvrow = dkrow + nh * 64;
            if (!ok) {
#pragma unroll
                for (int db = 0; db < 2; ++db)
#pragma unroll
                    for (int i = 0; i < 16; ++i) dk[db][kb][i] = dv[db][kb][i] = 0.f;  // (a masked column may hold inf / NaN)
            }
            if (rope_cos) {
                const int64_t prow = sv.pos0(b, pos_batch_stride) + krow;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 lo4 = {dk[0][kb][4 * g], dk[0][kb][4 * g + 1], dk[0][kb][4 * g + 2], dk[0][kb][4 * g + 3]};
                    f32x4 hi4 = {dk[1][kb][4 * g], dk[1][kb][4 * g + 1], dk[1][kb][4 * g + 2], dk[1][kb][4 * g + 3]};
                    rope_rotate4<true>(lo4, hi4, rope_cos + prow * 32, rope_sin + prow * 32, 8 * g + 4 * hh);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        dk[0][kb][4 * g + r] = lo4[r];
                        dk[1][kb][4 * g + r] = hi4[r];
                    }
                }
            }
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int d = 32 * db + 8 * g + 4 * hh;
                    *reinterpret_cast<uint2*>(dkrow + d) = uint2{pack_bf16x2(dk[db][kb][4 * g] * ks, dk[db][kb][4 * g + 1] * ks),
                                                                 pack_bf16x2(dk[db][kb][4 * g + 2] * ks, dk[db][kb][4 * g + 3] * ks)};
                    *reinterpret_cast<uint2*>(dvrow + d) = uint2{pack_bf16x2(dv[db][kb][4 * g] * vs, dv[db][kb][4 * g + 1] * vs),
                                                                 pack_bf16x2(dv[db][kb][4 * g + 2] * vs, dv[db][kb][4 * g + 3] * vs)};
                }
        }
    }
}

}  // namespace

extern "C" {

// Key blocks per dQ slab.  The ONE place the group size is derived (r04 advisor: the Python side used to parse the override on its own
// and more strictly than this file did; with CM3P_FUSED_SLAB_GROUP="4x" it issued the stage bits of G = 2 to a library running G = 4,
// and two of every four key blocks were never added).  The override counts only when it is exactly "2" or "4"; it is read per call.
int cm3p_attn_bwd_fused_slab_group(int S) {
    const char* e = getenv("CM3P_FUSED_SLAB_GROUP");
    if (e && (e[0] == '2' || e[0] == '4') && e[1] == '\0') return e[0] - '0';
    return (S + 255) / 256 >= 24 ? 4 : 2;
}

int64_t cm3p_attn_bwd_fused_workspace_bytes(int B, int S, int nh) {
    if (B <= 0 || S <= 0 || nh <= 0) return 0;
    const int64_t stats = (fused_stat_floats(B, S, nh) * 4 + 255) / 256 * 256;
    return stats + (int64_t)B * nh * fused_slabs(S) * fused_slab_rows(S) * 128;
}

int cm3p_attn_bwd_fused(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, const uint8_t* key_mask,
                        const int* cu_seqlens, int B, int S, int64_t total, int nh, float scale, const float* cos_tab,
                        const float* sin_tab, int64_t pos_batch_stride, int stages, int q_prescaled, void* workspace,
                        int64_t workspace_bytes, void* stream) {
    CM3P_REQUIRE((cos_tab == nullptr) == (sin_tab == nullptr));
    CM3P_REQUIRE(stages >= 1 && stages <= 255);
    CM3P_REQUIRE(qkv && out && dout && lse && dqkv && workspace && B > 0 && S > 0 && nh > 0 && scale > 0.f);
    CM3P_REQUIRE(cm3p_aligned16(qkv) && cm3p_aligned16(out) && cm3p_aligned16(dout) && cm3p_aligned16(dqkv) && cm3p_aligned16(workspace));
    if (cu_seqlens) CM3P_REQUIRE(total > 0 && key_mask == nullptr && pos_batch_stride == 0);
    else CM3P_REQUIRE(pos_batch_stride == 0 || pos_batch_stride == S);
    CM3P_REQUIRE(workspace_bytes >= cm3p_attn_bwd_fused_workspace_bytes(B, S, nh));
    CM3P_REQUIRE((int64_t)S * 3 * nh * 128 < (int64_t)1 << 31);  // 32-bit DMA source offsets inside one sequence
    hipStream_t s = static_cast<hipStream_t>(stream);
    const VarLen vl{cu_seqlens, cu_seqlens ? total : 0};
    float* stat_ws = static_cast<float*>(workspace);
    uint16_t* dq_part = reinterpret_cast<uint16_t*>(static_cast<char*>(workspace) + (fused_stat_floats(B, S, nh) * 4 + 255) / 256 * 256);
    const int NT = (S + 63) / 64;
    const bool pre = q_prescaled != 0;
    // Key blocks per dQ slab: 2, or 4 from 24 key blocks (S > 5888) on.  Four halve the slabs the reduce pass reads (C4: 6.3 -> 3.1 ms
    // per step) and put three of four launches on the packed-bf16 atomic path (2-8 % slower than the storing one, and one rounding per
    // add): r04 one-call A/B C4 214.6 -> 213.2 ms, C2 (16 key blocks) 168.8 vs 168.8 - hence the length rule.  CM3P_FUSED_SLAB_GROUP=2 / 4
    // overrides (cm3p_attn_bwd_fused_slab_group, which the caller asks too).  The workspace is sized for 2, the larger.
    const int G = cm3p_attn_bwd_fused_slab_group(S);
    // single-launch stage bits must name positions of THIS group size: a caller that derived another G would silently drop key blocks
    CM3P_REQUIRE((stages & (7 * CM3P_ATTN_BWD_FUSED_MAIN_ADD1) & ~(((1 << (G - 1)) - 1) * CM3P_ATTN_BWD_FUSED_MAIN_ADD1)) == 0);
    if (stages & CM3P_ATTN_BWD_FUSED_PREP) {
        // the score accumulators start at -lse * log2(e) (pre: q carries scale * log2 e, the MFMA delivers log2 p) or at
        // -lse / scale (cm * (q.k - lse / scale) = log2 p)
        const float lse_mul = pre ? -kLog2e : -1.0f / scale;
        attn_bwd_prep_kernel<<<dim3(NT + kFPadTiles, nh, B), 256, 0, s>>>((const uint16_t*)out, (const uint16_t*)dout, lse, stat_ws, S, nh,
                                                                          lse_mul, vl);
        if (hipGetLastError() != hipSuccess) return CM3P_ERR_LAUNCH;
    }
    const bool run_even = stages & (CM3P_ATTN_BWD_FUSED_MAIN | CM3P_ATTN_BWD_FUSED_MAIN_EVEN);
    const bool run_odd = stages & (CM3P_ATTN_BWD_FUSED_MAIN | CM3P_ATTN_BWD_FUSED_MAIN_ODD);
    if (run_even || run_odd || (stages & (7 * CM3P_ATTN_BWD_FUSED_MAIN_ADD1))) {
        static Cm3pDevOnce once;  // (per device: common.h)
        const int rc_once = once.run([] {
            return cm3p_set_max_lds({reinterpret_cast<const void*>(&attn_bwd_fused_kernel<true, false>), reinterpret_cast<const void*>(&attn_bwd_fused_kernel<true, true>),
                                     reinterpret_cast<const void*>(&attn_bwd_fused_kernel<false, false>), reinterpret_cast<const void*>(&attn_bwd_fused_kernel<false, true>)},
                                    kFLds);
        });
        if (rc_once != CM3P_OK) return rc_once;
        // G launches: the key blocks G k store their dQ partial to slab k, then the key blocks G k + 1, .. each add theirs to it (1-D
        // grids: decode_block() maps them XCD-aware)
        const int nkb = (S + 255) / 256;
#define CM3P_FUSED_ARGS (const uint16_t*)qkv, (const uint16_t*)dout, stat_ws, dq_part, (uint16_t*)dqkv, key_mask, S, nh, scale, cos_tab, sin_tab, pos_batch_stride, vl, G
        if (run_even) {
            const dim3 grid0(((nkb + G - 1) / G) * nh * B);
            if (pre) attn_bwd_fused_kernel<true, false><<<grid0, 256, kFLds, s>>>(CM3P_FUSED_ARGS, 0);
            else attn_bwd_fused_kernel<false, false><<<grid0, 256, kFLds, s>>>(CM3P_FUSED_ARGS, 0);
            if (hipGetLastError() != hipSuccess) return CM3P_ERR_LAUNCH;
        }
        for (int phase = 1; phase < G && phase < nkb; ++phase) {
            if (!(run_odd || (stages & (CM3P_ATTN_BWD_FUSED_MAIN_ADD1 << (phase - 1))))) continue;
            const dim3 gridp(((nkb - phase + G - 1) / G) * nh * B);
            if (pre) attn_bwd_fused_kernel<true, true><<<gridp, 256, kFLds, s>>>(CM3P_FUSED_ARGS, phase);
            else attn_bwd_fused_kernel<false, true><<<gridp, 256, kFLds, s>>>(CM3P_FUSED_ARGS, phase);
            if (hipGetLastError() != hipSuccess) return CM3P_ERR_LAUNCH;
        }
#undef CM3P_FUSED_ARGS
    }
    if (stages & CM3P_ATTN_BWD_FUSED_REDUCE) {
        attn_bwd_dq_reduce_kernel<<<dim3(NT, nh, B), 256, 0, s>>>(dq_part, (uint16_t*)dqkv, S, nh, scale, cos_tab, sin_tab, pos_batch_stride, vl, G);
    }
    CM3P_LAUNCH_CHECK();
    return CM3P_OK;
}

}  // extern "C"

// timing-only ablation switches this object was built with (0 in every shipped build: cm3p_build_ablation_flags, tests/test_cabi.py)
int cm3p_ablation_flags_attention_bwd_fused() { return (CM3P_FABL) | ((CM3P_FTRACE) << 12); }
#if CM3P_FTRACE
extern "C" int cm3p_debug_set_bwdf_trace(unsigned long long* p) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_bwdf_trace), &p, sizeof(p)) == hipSuccess ? 0 : 1;
}
#endif

#if CM3P_DMA_AUDIT
int cm3p_audit_set_attention_bwd_fused(void* buf) { return cm3p_audit_set_local(buf); }
#endif
