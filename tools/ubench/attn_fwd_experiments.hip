// NOT BUILT - r01 experiments kept for the next round ("Attention: what was measured").
// Variants of the forward kernel: LDS-DMA ring (4/8 waves, 32/64 queries per wave, 3/4 stages) and a software-pipelined
// loop with sched_group_barrier.  All ran within 2.07-2.25 ms per C2 global layer, i.e. no better than the shipped kernel.

// Non-causal flash attention for ModernBERT's global and sliding-window layers, head_dim 64, gfx950.
//
// Replaces F.scaled_dot_product_attention(q, k, v, attn_mask, scale, is_causal=False)
// (TF:integrations/sdpa_attention.py:153-163, called from TF:models/modernbert/modeling_modernbert.py:286-297)
// and the (B,1,S,S) boolean mask the reference materialises for it (TF:masking_utils.py:141-151,168-179):
//     visible(b, q, kv) = key_mask[b, kv] AND (window < 0 OR |q - kv| <= window)
// Nothing S x S is ever stored: the rule is evaluated per score inside the kernels, and sliding-window layers only
// visit the key tiles that intersect the band.
//
// Layout: packed qkv [B, S, 3, nh, 64] bf16 (q, k already rotated), out [B, S, nh, 64] bf16, lse [B, nh, S] fp32.
//
// MFMA formulation (v_mfma_f32_32x32x16_bf16, one wave = 32 queries or 32 keys):
//   forward  S^T = K Q^T (key rows in registers, query on the lane)  ->  softmax statistics are per-lane scalars;
//            O^T += V^T P^T, where P^T is fed to the MFMA straight from the S^T accumulators (no LDS round trip) and
//            V^T comes from ds_read_b64_tr_b16 (hardware transpose) of the row-major V tile.
//   dq       recompute S^T, dP^T = V dO^T, dS^T = P^T o (dP^T - delta); dQ^T += K^T dS^T (K^T by transposed reads).
//   dk, dv   key on the lane: S = Q K^T, dP = dO V^T with -lse/scale and -delta preloaded as the initial accumulators;
//            dV^T += dO^T P and dK^T += Q^T dS consume the accumulators directly as B operands.
// LDS tiles use one swizzled image (128-byte rows) that is bank-conflict free for ds_read_b128 row fragments AND for
// ds_read_b64_tr_b16 transposed reads, so a tile consumed both ways (K in dq; Q and dO in dk/dv) is stored once.
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>

#include "common.h"

namespace {

// Diagnostic build only (-DCM3P_STAMPS, tools/attn_timeline.sh): per-wave s_memtime segment sums of the forward loop.
#ifdef CM3P_STAMPS
__device__ unsigned long long g_seg[16];
#define SEG_INIT() unsigned long long tprev__ = __builtin_amdgcn_s_memtime(), tseg__[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define SEG(k)                                                    \
    do {                                                          \
        const unsigned long long tn__ = __builtin_amdgcn_s_memtime(); \
        tseg__[k] += tn__ - tprev__;                              \
        tprev__ = tn__;                                           \
    } while (0)
#define SEG_FLUSH()                                                                      \
    do {                                                                                 \
        if ((threadIdx.x & 63) == 0) {                                                   \
            for (int k__ = 0; k__ < 8; ++k__) atomicAdd(&g_seg[k__], tseg__[k__]);       \
            atomicAdd(&g_seg[8], 1ull);                                                  \
        }                                                                                \
    } while (0)
#else
#define SEG_INIT()
#define SEG(k)
#define SEG_FLUSH()
#endif

constexpr float kLog2e = 1.4426950408889634f;
constexpr float kNegInf = -__builtin_huge_valf();

// ONE LDS image serves both access shapes (128-byte rows of 64 bf16, 16-byte chunk index XOR swz(row)):
//   row fragments   (ds_read_b128, lane = row, fixed chunk): the 8 even / 8 odd rows of every 16-lane group get 8 different
//                   swz values -> 16 different 16-byte slots of the 256-byte bank row;
//   transposed reads (ds_read_b64_tr_b16, 4 rows x 64 bytes per 32-lane half): rows b, b+1 sit in different halves of the
//                   bank row and bit 2 of swz moves rows b+2, b+3 to the other aligned group of four chunks.
__device__ __forceinline__ int swz(int row) { return (((row >> 1) & 1) << 2) | ((row >> 2) & 3); }
__device__ __forceinline__ int off_R(int row, int c16) { return row * 128 + ((c16 ^ swz(row)) << 4); }
__device__ __forceinline__ int off_T(int row, int col) { return row * 128 + (((col >> 3) ^ swz(row)) << 4) + ((col & 7) << 1); }

// hardware workgroup id -> logical id such that each XCD (n % 8) owns a contiguous range of logical ids (bijective)
__device__ __forceinline__ int xcd_remap(int n, int total) {
    const int q8 = total / 8, r8 = total % 8, xcd = n % 8;
    return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + n / 8;
}

// 1-D grid of nblk * nh * B workgroups -> (block along the sequence, head, batch), XCD-aware: the blocks of one
// (batch, head) are consecutive logical ids, so they run on ONE XCD and share its L2 copy of that head's K / V (or Q / dO)
// instead of pulling it into all eight L2s (which thrashes them and turns every tile load into a fabric access).
__device__ __forceinline__ void decode_block(int nblk, int nh, int& blk, int& head, int& b) {
    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    blk = logical % nblk;
    const int bh = logical / nblk;
    head = bh % nh;
    b = bh / nh;
}

__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// A-operand fragment of X^T (rows = the 64 columns of tile X, k = 16 rows of X starting at krow0) for the product
// X^T * Y where Y comes from accumulators: element j of lane half hh is row krow0 + 8*(j>>2) + 4*hh + (j&3) of X.
// `cblk` selects columns 32*cblk .. 32*cblk+31 of X (the MFMA's 32 output rows).
__device__ __forceinline__ bf16x8 frag_T(const char* tile, int krow0, int cblk, int lane) {
    const int g = lane >> 4, hh = g >> 1, i = lane & 15;
    const int row = krow0 + 4 * hh + (i >> 2);
    const int col = 32 * cblk + 16 * (g & 1) + 4 * (i & 3);
    const bf16x4 lo = lds_read_tr16(tile + off_T(row, col));
    const bf16x4 hi = lds_read_tr16(tile + off_T(row + 8, col));
    return cat_bf16x4(lo, hi);
}

// Row fragment (A or B operand): lane holds X[row0 + (lane&31)][16*s + 8*(lane>>5) + j]
__device__ __forceinline__ bf16x8 frag_R(const char* tile, int row0, int s, int lane) {
    return *reinterpret_cast<const bf16x8*>(tile + off_R(row0 + (lane & 31), 2 * s + (lane >> 5)));
}

// accumulator registers 8*sp .. 8*sp+7 -> bf16 B-operand fragment for k-step sp
__device__ __forceinline__ bf16x8 acc_to_frag(const f32x16& a, int sp) {
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (__bf16)a[8 * sp + j];
    return r;
}

// fragment * c, rounded back to bf16: folds softmax's scale*log2(e) into one MFMA operand so that the accumulator is
// already in exp2 units (saves one VALU op per score)
__device__ __forceinline__ bf16x8 scale_frag(bf16x8 v, float c) {
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (__bf16)((float)v[j] * c);
    return r;
}

__device__ __forceinline__ float max3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

struct TileRegs64 {  // a 64-row x 64-col bf16 tile spread over 256 threads: 2 x 16 bytes each
    uint4 v[2];
};

// rows r0 .. r0+63 of a [*, 64] bf16 matrix with row stride `ld` elements; rows >= limit read as zero
__device__ __forceinline__ void gload64(TileRegs64& t, const uint16_t* base, int64_t ld, int r0, int limit, int tid) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = tid + 256 * i, row = q >> 3, c = q & 7;
        const int r = r0 + row;
        t.v[i] = (r < limit && r >= 0) ? *reinterpret_cast<const uint4*>(base + (int64_t)r * ld + c * 8) : uint4{0u, 0u, 0u, 0u};
    }
}
__device__ __forceinline__ void lstore64_R(char* tile, const TileRegs64& t, int tid) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = tid + 256 * i, row = q >> 3, c = q & 7;
        *reinterpret_cast<uint4*>(tile + off_R(row, c)) = t.v[i];
    }
}

__device__ __forceinline__ float reg_max16(const f32x16& a) {  // 8 x v_max3_f32
    const float m0 = max3(a[0], a[1], a[2]), m1 = max3(a[3], a[4], a[5]), m2 = max3(a[6], a[7], a[8]);
    const float m3 = max3(a[9], a[10], a[11]), m4 = max3(a[12], a[13], a[14]);
    return max3(max3(m0, m1, m2), max3(m3, m4, a[15]), m0);
}

// Visibility of keys (tile key0 + 32*blk + accumulator rows) for this lane's query: invisible scores become -inf.
// Branch-free: [lo, hi] is the lane's window in absolute key positions (whole axis for global layers), maskb holds one
// validity byte per key of the tile (padding and keys >= S are 0).
__device__ __forceinline__ void mask_scores_keyrows(f32x16& s, const uint8_t* maskb, int blk, int key0, int lo, int hi, int hh) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int kl = 32 * blk + 8 * g + 4 * hh;
        const uint32_t mb = *reinterpret_cast<const uint32_t*>(maskb + kl);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = key0 + kl + r;
            const bool ok = (((mb >> (8 * r)) & 0xffu) != 0u) & (key >= lo) & (key <= hi);
            s[4 * g + r] = ok ? s[4 * g + r] : kNegInf;
        }
    }
}

// true when a 64-key tile needs no masking for any query of a 32-row wave block [q0, q0+31]:
// every key valid (flag computed when the tile's mask bytes were staged) and the tile inside every row's window
__device__ __forceinline__ bool tile_unmasked(int all_valid, int key0, int q0, int window) {
    return all_valid && (window < 0 || (key0 >= q0 + 31 - window && key0 + 63 <= q0 + window));
}

// ---------------------------------------------------------------------------------------------------------------
// forward: one workgroup = 4 waves = 128 queries of one (batch, head); K/V tiles of 64 keys, double buffered.
// LDS per stage: K image (8 KiB) + V image (8 KiB) + 64 mask bytes + flag.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kFwdStage = 8192 + 8192 + 64 + 16;  // K image R, V image T, mask bytes, all-valid flag

// QSUB = number of 32-query sub-blocks per wave.  With QSUB = 2 a wave carries two independent softmax chains: the
// MFMAs of one sub-block overlap the VALU work of the other inside a single instruction stream (one wave cannot hide
// its own MFMA -> VALU -> MFMA dependency), and every K / V fragment read from LDS feeds twice the work.
template <int QSUB>
__global__ __launch_bounds__(256, QSUB == 1 ? 2 : 1) void attn_fwd_kernel(const uint16_t* __restrict__ qkv, uint16_t* __restrict__ out,
                                                                        float* __restrict__ lse, const uint8_t* __restrict__ kmask,
                                                                        int S, int nh, int window, float scale) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int QW = 32 * QSUB;  // queries per wave
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, hh = lane >> 5;
    int qblk, head, b;
    decode_block((S + 4 * QW - 1) / (4 * QW), nh, qblk, head, b);
    const int Q0 = qblk * (4 * QW);
    const int q0 = Q0 + wid * QW;
    const int64_t ld = (int64_t)3 * nh * 64;
    const uint16_t* qbase = qkv + (int64_t)b * S * ld + head * 64;
    const uint16_t* kbase = qbase + nh * 64;
    const uint16_t* vbase = qbase + 2 * nh * 64;
    const float c = scale * kLog2e;
    constexpr float kDefer = 6.0f;

    int qrow[QSUB], lo[QSUB], hi[QSUB];
    bf16x8 qf[QSUB][4];  // Q * (scale * log2 e): scores come out of the MFMA in exp2 units
    f32x16 oacc[QSUB][2];
    // Softmax state per query (= per lane): mc_run is the reference point in log2 units that every stored p, l and O is
    // relative to; it is subtracted inside the MFMA (as the initial accumulator) and only moved when a tile maximum
    // exceeds it by more than 2^kDefer ("lazy max": exact in exact arithmetic, the reference point divides out).
    float mc_run[QSUB], l_run[QSUB];
    bool has_ref[QSUB];
#pragma unroll
    for (int u = 0; u < QSUB; ++u) {
        qrow[u] = q0 + 32 * u + (lane & 31);
        lo[u] = window < 0 ? INT_MIN : qrow[u] - window;
        hi[u] = window < 0 ? INT_MAX : qrow[u] + window;
        const int qc = qrow[u] < S ? qrow[u] : S - 1;
#pragma unroll
        for (int s = 0; s < 4; ++s)
            qf[u][s] = scale_frag(*reinterpret_cast<const bf16x8*>(qbase + (int64_t)qc * ld + 16 * s + 8 * hh), c);
#pragma unroll
        for (int i = 0; i < 16; ++i) oacc[u][0][i] = oacc[u][1][i] = 0.f;
        mc_run[u] = 0.f;
        l_run[u] = 0.f;
        has_ref[u] = false;
    }

    const int Q1 = min(S, Q0 + 4 * QW) - 1;
    int klo = 0, khi = S - 1, wlo = 0, whi = S - 1;
    if (window >= 0) {
        klo = max(0, Q0 - window);
        khi = min(S - 1, Q1 + window);
        wlo = max(0, q0 - window);
        whi = min(S - 1, q0 + QW - 1 + window);
    }
    const bool wave_live = q0 < S;
    const int t_lo = klo / 64, t_hi = khi / 64;

    TileRegs64 kr, vr;
    uint8_t mreg = 0;
    auto gload = [&](int t) {
        gload64(kr, kbase, ld, t * 64, S, tid);
        gload64(vr, vbase, ld, t * 64, S, tid);
        if (tid < 64) {
            const int key = t * 64 + tid;
            mreg = key < S ? (kmask ? kmask[(int64_t)b * S + key] : (uint8_t)1) : (uint8_t)0;
        }
    };
    auto lstore = [&](int stage) {
        char* st = smem + stage * kFwdStage;
        lstore64_R(st, kr, tid);
        lstore64_R(st + 8192, vr, tid);
        if (tid < 64) {
            reinterpret_cast<uint8_t*>(st + 16384)[tid] = mreg;
            const unsigned long long valid = __ballot(mreg != 0);
            if (tid == 0) *reinterpret_cast<int*>(st + 16448) = (valid == ~0ull) ? 1 : 0;
        }
    };

    gload(t_lo);
    lstore(0);
    __syncthreads();
    SEG_INIT();

    for (int t = t_lo; t <= t_hi; ++t) {
        const int stage = (t - t_lo) & 1;
        const char* st = smem + stage * kFwdStage;
        const bool more = t < t_hi;
        if (more) gload(t + 1);
        SEG(0);

        const int key0 = t * 64;
        if (wave_live && key0 <= whi && key0 + 63 >= wlo) {
            f32x16 sacc[QSUB][2];
#pragma unroll
            for (int u = 0; u < QSUB; ++u)
#pragma unroll
                for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                    for (int i = 0; i < 16; ++i) sacc[u][blk][i] = -mc_run[u];
            // S^T = K (cQ)^T - reference: each K fragment is read from LDS once and used by every sub-block
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const bf16x8 kf = frag_R(st, 32 * blk, s, lane);
#pragma unroll
                    for (int u = 0; u < QSUB; ++u) sacc[u][blk] = mfma32(kf, qf[u][s], sacc[u][blk]);
                }
            const int all_valid = *reinterpret_cast<const int*>(st + 16448);
            SEG(1);
#pragma unroll
            for (int u = 0; u < QSUB; ++u) {
                if (!tile_unmasked(all_valid, key0, q0 + 32 * u, window)) {
                    const uint8_t* mb = reinterpret_cast<const uint8_t*>(st + 16384);
                    mask_scores_keyrows(sacc[u][0], mb, 0, key0, lo[u], hi[u], hh);
                    mask_scores_keyrows(sacc[u][1], mb, 1, key0, lo[u], hi[u], hh);
                }
                float mt = fmaxf(reg_max16(sacc[u][0]), reg_max16(sacc[u][1]));  // tile max relative to mc_run
                mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
                const bool move = has_ref[u] ? (mt > kDefer) : (mt > kNegInf);
                if (__any(move)) {  // rare after the first tiles: shift the reference point of the rows that need it
                    const float shift = has_ref[u] ? fmaxf(mt, 0.f) : (mt > kNegInf ? mt : 0.f);
                    const float alpha = has_ref[u] ? __builtin_amdgcn_exp2f(-shift) : 1.0f;  // O = l = 0 before the first score
                    has_ref[u] = has_ref[u] || (mt > kNegInf);
                    mc_run[u] += shift;
                    l_run[u] *= alpha;
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        sacc[u][0][i] -= shift;
                        sacc[u][1][i] -= shift;
                        oacc[u][0][i] *= alpha;
                        oacc[u][1][i] *= alpha;
                    }
                }
                float psum = 0.f;
#pragma unroll
                for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const float p = __builtin_amdgcn_exp2f(sacc[u][blk][i]);
                        sacc[u][blk][i] = p;
                        psum += p;
                    }
                l_run[u] += psum;
            }
            SEG(2);
            // O^T += V^T P^T: each V^T fragment (two transposed LDS reads) feeds every sub-block
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const bf16x8 v0 = frag_T(st + 8192, 16 * s, 0, lane);
                const bf16x8 v1 = frag_T(st + 8192, 16 * s, 1, lane);
#pragma unroll
                for (int u = 0; u < QSUB; ++u) {
                    const bf16x8 pf = acc_to_frag(sacc[u][s >> 1], s & 1);
                    oacc[u][0] = mfma32(v0, pf, oacc[u][0]);
                    oacc[u][1] = mfma32(v1, pf, oacc[u][1]);
                }
            }
            SEG(3);
        }
        if (more) lstore(stage ^ 1);
        SEG(4);
        __syncthreads();
        SEG(5);
    }
    SEG_FLUSH();

#pragma unroll
    for (int u = 0; u < QSUB; ++u) {
        const float l_tot = l_run[u] + __shfl_xor(l_run[u], 32, 64);
        const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
        if (qrow[u] < S) {
            uint16_t* orow = out + ((int64_t)b * S + qrow[u]) * nh * 64 + head * 64;
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int dv = 32 * blk + 8 * g + 4 * hh;
                    const uint2 w = {pack_bf16x2(oacc[u][blk][4 * g] * inv, oacc[u][blk][4 * g + 1] * inv),
                                     pack_bf16x2(oacc[u][blk][4 * g + 2] * inv, oacc[u][blk][4 * g + 3] * inv)};
                    *reinterpret_cast<uint2*>(orow + dv) = w;
                }
            if (hh == 0)
                lse[((int64_t)b * nh + head) * S + qrow[u]] =
                    l_tot > 0.f ? (mc_run[u] + __log2f(l_tot)) * 0.69314718055994531f : __builtin_huge_valf();
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// forward, LDS-DMA ring.  Measured on the register-staged kernel above (s_memtime segment sums, tools/attn_timeline.sh):
// a wave spends ~70 % of a tile's wall time issuing and awaiting the NEXT tile's global loads - one tile of prefetch
// (16 KiB per workgroup in flight) is far below what the L2 -> CU path needs to stream at rate (>= 72 KiB per CU in
// flight, MI355X_MICROARCH.md "Indexed rows"), although 94 % of the requests hit L2.  Here K / V tiles go global -> LDS by
// LDS-DMA (global_load_lds_dwordx4: no VGPR staging, no ds_write) into a ring of kRingN stages requested kRingN-1 tiles
// ahead, with one workgroup barrier per tile; the per-key validity of the row lives in LDS as one 64-bit word per tile.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kRingStage = 16384;  // K image (row reads) + V image (transposed reads)

// LDS-DMA of 64 x 16 bytes: lane l's 16 bytes from its own global address land at lds_wave_base + 16 * l.
// Issued as inline assembly on purpose: through the builtin the compiler knows the instruction writes LDS and, unable to
// prove that a later ds_read touches another ring slot, drains EVERY outstanding DMA (s_waitcnt vmcnt(0)) in front of the
// fragment reads of each tile - which serialises the whole prefetch ring.  Ordering is explicit instead: a counted vmcnt
// wait plus the workgroup barrier in front of the first read of a stage.
__device__ __forceinline__ void glds16(const void* gsrc, char* lds_wave_base) {
    const uint32_t m0v = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds_wave_base);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(m0v) : "memory", "m0");
}
// workgroup barrier that orders LDS traffic only (DMA completion is a counted vmcnt wait before it)
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// scores of keys whose validity bit is 0 or that lie outside the lane's window become -inf (bits = the tile's 64 keys)
__device__ __forceinline__ void mask_scores_bits(f32x16& s, unsigned long long bits, int blk, int key0, int lo, int hi, int hh) {
    const uint32_t w = (uint32_t)(bits >> (32 * blk)) >> (4 * hh);
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = key0 + 32 * blk + 8 * g + 4 * hh + r;
            const bool ok = (((w >> (8 * g + r)) & 1u) != 0u) & (key >= lo) & (key <= hi);
            s[4 * g + r] = ok ? s[4 * g + r] : kNegInf;
        }
}

// WAVES waves per workgroup, QSUB sub-blocks of 32 queries per wave (a K / V fragment read from LDS feeds QSUB sub-blocks and
// the wave carries QSUB independent softmax chains), NST ring stages.
template <int WAVES, int QSUB, int NST, int MINB>
__global__ __launch_bounds__(64 * WAVES, MINB) void attn_fwd_ring_kernel(const uint16_t* __restrict__ qkv, uint16_t* __restrict__ out,
                                                                       float* __restrict__ lse, const uint8_t* __restrict__ kmask, int S,
                                                                       int nh, int window, float scale) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int QW = 32 * QSUB;           // queries per wave
    constexpr int QB = QW * WAVES;          // queries per workgroup
    constexpr int PIECES = 16 / WAVES;      // 1-KiB LDS-DMA instructions per wave and stage (8 K pieces + 8 V pieces in all)
    constexpr int DEPTH = NST - 1;
    const int tid = threadIdx.x, lane = tid & 63, hh = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    int qblk, head, b;
    decode_block((S + QB - 1) / QB, nh, qblk, head, b);
    const int Q0 = qblk * QB;
    const int q0 = Q0 + wid * QW;
    const int64_t ld = (int64_t)3 * nh * 64;
    const uint16_t* qbase = qkv + (int64_t)b * S * ld + head * 64;
    const uint16_t* kbase = qbase + nh * 64;
    const uint16_t* vbase = qbase + 2 * nh * 64;
    const float c = scale * kLog2e;
    constexpr float kDefer = 6.0f;

    int qrow[QSUB], lo[QSUB], hi[QSUB];
    bf16x8 qf[QSUB][4];
#pragma unroll
    for (int u = 0; u < QSUB; ++u) {
        qrow[u] = q0 + 32 * u + (lane & 31);
        lo[u] = window < 0 ? INT_MIN : qrow[u] - window;
        hi[u] = window < 0 ? INT_MAX : qrow[u] + window;
        const int qc = qrow[u] < S ? qrow[u] : S - 1;
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[u][s] = *reinterpret_cast<const bf16x8*>(qbase + (int64_t)qc * ld + 16 * s + 8 * hh);
    }

    const int Q1 = min(S, Q0 + QB) - 1;
    int klo = 0, khi = S - 1, wlo = 0, whi = S - 1;
    if (window >= 0) {
        klo = max(0, Q0 - window);
        khi = min(S - 1, Q1 + window);
        wlo = max(0, q0 - window);
        whi = min(S - 1, q0 + QW - 1 + window);
    }
    const bool wave_live = q0 < S;
    const int t_lo = klo / 64, t_hi = khi / 64;

    // validity words of the tiles this workgroup visits: wave w takes tiles t_lo + w, t_lo + w + WAVES, ...
    unsigned long long* lbits = reinterpret_cast<unsigned long long*>(smem + NST * kRingStage);
    for (int t = t_lo + wid; t <= t_hi; t += WAVES) {
        const int key = t * 64 + lane;
        const bool valid = key < S && (kmask == nullptr || kmask[(int64_t)b * S + key] != 0);
        const unsigned long long bits = __ballot(valid);
        if (lane == 0) lbits[t - t_lo] = bits;
    }

    // stage of tile t -> ring slot `slot`; piece i of this wave covers 8 rows; a lane fetches the 16-byte chunk that belongs
    // at its position of the swizzled image.  Rows past the end of the sequence are clamped (their scores are masked).
    auto dma = [&](int t, int slot) {
        char* st = smem + slot * kRingStage;
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            const int piece = wid * PIECES + i;  // 0..7: K image, 8..15: V image
            const int prow = (piece & 7) * 8 + (lane >> 3);
            const int dcol = (((lane & 7) ^ swz(prow)) << 3);
            const int r = min(t * 64 + prow, S - 1);
            glds16((piece < 8 ? kbase : vbase) + (int64_t)r * ld + dcol, st + piece * 1024);
        }
    };
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) dma(t_lo + i, i);  // behind the Q loads: waiting for Q does not wait for these
#pragma unroll
    for (int u = 0; u < QSUB; ++u)
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[u][s] = scale_frag(qf[u][s], c);

    f32x16 oacc[QSUB][2];
    // softmax state per query (= per lane), as in attn_fwd_kernel: reference point mc_run ("lazy max")
    float mc_run[QSUB], l_run[QSUB];
    bool has_ref[QSUB];
#pragma unroll
    for (int u = 0; u < QSUB; ++u) {
#pragma unroll
        for (int i = 0; i < 16; ++i) oacc[u][0][i] = oacc[u][1][i] = 0.f;
        mc_run[u] = l_run[u] = 0.f;
        has_ref[u] = false;
    }

    int slot = 0, slot_next = DEPTH % NST;  // slot of tile t, slot that tile t + DEPTH goes into (= tile t-1's)
    for (int t = t_lo; t <= t_hi; ++t) {
#ifndef AB_NODMA
        // every stage is PIECES instructions per wave, so "all but the youngest DEPTH-1 stages landed" is a counted wait
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES * (DEPTH - 1)) : "memory");
#endif
#ifndef AB_NOBAR
        lds_barrier();  // tile t is in LDS for everyone, and everyone is done reading tile t-1's slot
#endif
#ifndef AB_NODMA
        dma(t + DEPTH, slot_next);  // dummy past t_hi: keeps the count uniform
#endif
        const char* st = smem + slot * kRingStage;
        slot = slot + 1 == NST ? 0 : slot + 1;
        slot_next = slot_next + 1 == NST ? 0 : slot_next + 1;
        const int key0 = t * 64;
        if (wave_live && key0 <= whi && key0 + 63 >= wlo) {
            f32x16 sacc[QSUB][2];
#pragma unroll
            for (int u = 0; u < QSUB; ++u)
#pragma unroll
                for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                    for (int i = 0; i < 16; ++i) sacc[u][blk][i] = -mc_run[u];
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int s = 0; s < 4; ++s) {
#ifdef AB_NOFRAG
                    const bf16x8 kf = qf[0][(s + blk) & 3];
#else
                    const bf16x8 kf = frag_R(st, 32 * blk, s, lane);
#endif
#pragma unroll
                    for (int u = 0; u < QSUB; ++u) sacc[u][blk] = mfma32(kf, qf[u][s], sacc[u][blk]);
                }
#ifndef AB_NOMAX
            const unsigned long long bits = lbits[t - t_lo];
#endif
#pragma unroll
            for (int u = 0; u < QSUB; ++u) {
#ifndef AB_NOMAX
                if (!tile_unmasked(bits == ~0ull, key0, q0 + 32 * u, window)) {
                    mask_scores_bits(sacc[u][0], bits, 0, key0, lo[u], hi[u], hh);
                    mask_scores_bits(sacc[u][1], bits, 1, key0, lo[u], hi[u], hh);
                }
                float mt = fmaxf(reg_max16(sacc[u][0]), reg_max16(sacc[u][1]));  // tile max relative to mc_run
                mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
                const bool move = has_ref[u] ? (mt > kDefer) : (mt > kNegInf);
                if (__any(move)) {  // rare after the first tiles: shift the reference point of the rows that need it
                    const float shift = has_ref[u] ? fmaxf(mt, 0.f) : (mt > kNegInf ? mt : 0.f);
                    const float alpha = has_ref[u] ? __builtin_amdgcn_exp2f(-shift) : 1.0f;  // O = l = 0 before the first score
                    has_ref[u] = has_ref[u] || (mt > kNegInf);
                    mc_run[u] += shift;
                    l_run[u] *= alpha;
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        sacc[u][0][i] -= shift;
                        sacc[u][1][i] -= shift;
                        oacc[u][0][i] *= alpha;
                        oacc[u][1][i] *= alpha;
                    }
                }
#endif
                float ps[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
#ifdef AB_NOEXP
                        const float p = sacc[u][blk][i] * 0.001f;
#else
                        const float p = __builtin_amdgcn_exp2f(sacc[u][blk][i]);
#endif
                        sacc[u][blk][i] = p;
                        ps[i & 3] += p;
                    }
                l_run[u] += (ps[0] + ps[1]) + (ps[2] + ps[3]);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#ifdef AB_NOFRAG
                const bf16x8 v0 = qf[0][s], v1 = qf[0][3 - s];
#else
                const bf16x8 v0 = frag_T(st + 8192, 16 * s, 0, lane);
                const bf16x8 v1 = frag_T(st + 8192, 16 * s, 1, lane);
#endif
#pragma unroll
                for (int u = 0; u < QSUB; ++u) {
                    const bf16x8 pf = acc_to_frag(sacc[u][s >> 1], s & 1);
                    oacc[u][0] = mfma32(v0, pf, oacc[u][0]);
                    oacc[u][1] = mfma32(v1, pf, oacc[u][1]);
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the dummy tail stages

#pragma unroll
    for (int u = 0; u < QSUB; ++u) {
        const float l_tot = l_run[u] + __shfl_xor(l_run[u], 32, 64);
        const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
        if (qrow[u] < S) {
            uint16_t* orow = out + ((int64_t)b * S + qrow[u]) * nh * 64 + head * 64;
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int dv = 32 * blk + 8 * g + 4 * hh;
                    const uint2 w = {pack_bf16x2(oacc[u][blk][4 * g] * inv, oacc[u][blk][4 * g + 1] * inv),
                                     pack_bf16x2(oacc[u][blk][4 * g + 2] * inv, oacc[u][blk][4 * g + 3] * inv)};
                    *reinterpret_cast<uint2*>(orow + dv) = w;
                }
            if (hh == 0)
                lse[((int64_t)b * nh + head) * S + qrow[u]] =
                    l_tot > 0.f ? (mc_run[u] + __log2f(l_tot)) * 0.69314718055994531f : __builtin_huge_valf();
        }
    }
}

template <int WAVES, int QSUB, int NST, int MINB>
int launch_fwd_ring(const void* qkv, void* out, float* lse, const uint8_t* key_mask, int B, int S, int nh, int window, float scale,
                    hipStream_t s) {
    const int T = (S + 63) / 64;
    const size_t lds = NST * kRingStage + (size_t)T * 8 + 16;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)attn_fwd_ring_kernel<WAVES, QSUB, NST, MINB>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                NST * kRingStage + 1024 * 8 + 16) != hipSuccess)
            return CM3P_ERR_LAUNCH;
        attr_set = true;
    }
    constexpr int QB = 32 * QSUB * WAVES;
    const dim3 grid(((S + QB - 1) / QB) * nh * B);
    attn_fwd_ring_kernel<WAVES, QSUB, NST, MINB><<<grid, 64 * WAVES, lds, s>>>((const uint16_t*)qkv, (uint16_t*)out, lse, key_mask, S, nh,
                                                                             window, scale);
    return CM3P_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// forward, global layers, software-pipelined by one tile inside each wave.
// Measured on gfx950 (tools/ubench/overlap.hip): the matrix pipe and the VALU of a SIMD do NOT overlap across two waves
// (16 MFMAs in one wave + a softmax-shaped VALU block in its partner take the SUM of their times), but VALU instructions
// placed between the MFMAs of ONE wave ride in their shadow (an MFMA holds the vector issue port for 8 of its 32 cycles).
// The kernels above run QK^T -> softmax -> PV of a tile back to back, so every tile pays MFMA + VALU.  Here iteration t
// issues, as one branch-free block,
//        MFMA:  S(t+1) = K(t+1) Q^T                O += V(t-1)^T P(t-1)
//        VALU:  P(t) = exp2(S(t)), row sums, bf16 packing          (independent of both MFMA groups)
//        LDS :  fragments of K(t+1), V(t-1)        DMA: ring stage t-1+DEPTH
// and sched_group_barrier spreads the VALU work over the 16 MFMA gaps.  Ring stage s carries {K(s+2), V(s)}, so one
// stage is live per iteration.  Softmax reference point: moved only when a row sum demands it (no per-tile maximum on the
// fast path): p may reach 2^30 relative to `ref` before the exact path (tile maximum, shift, rescale) runs; the first
// tile, masked tiles and rows without a reference also take the exact path.  Everything stored is relative to the same
// reference, which divides out - exact in exact arithmetic, like the lazy maximum above.
// ---------------------------------------------------------------------------------------------------------------
constexpr float kBig = 1073741824.0f;  // 2^30

template <int NST>
__global__ __launch_bounds__(256, 2) void attn_fwd_pipe_kernel(const uint16_t* __restrict__ qkv, uint16_t* __restrict__ out,
                                                               float* __restrict__ lse, const uint8_t* __restrict__ kmask, int S, int nh,
                                                               float scale) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int DEPTH = NST - 1;
    constexpr int PIECES = 4;  // per wave and stage: 2 K pieces + 2 V pieces of 1 KiB
    const int tid = threadIdx.x, lane = tid & 63, hh = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    int qblk, head, b;
    decode_block((S + 127) / 128, nh, qblk, head, b);
    const int q0 = qblk * 128 + wid * 32;
    const int64_t ld = (int64_t)3 * nh * 64;
    const uint16_t* qbase = qkv + (int64_t)b * S * ld + head * 64;
    const uint16_t* kbase = qbase + nh * 64;
    const uint16_t* vbase = qbase + 2 * nh * 64;
    const float c = scale * kLog2e;
    constexpr float kDefer = 6.0f;
    const int T = (S + 63) / 64;
    const bool wave_live = q0 < S;

    const int qrow = q0 + (lane & 31);
    const int qc = qrow < S ? qrow : S - 1;
    bf16x8 qf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(qbase + (int64_t)qc * ld + 16 * s + 8 * hh);

    unsigned long long* lbits = reinterpret_cast<unsigned long long*>(smem + NST * kRingStage);
    for (int t = wid; t < T; t += 4) {
        const int key = t * 64 + lane;
        const bool valid = key < S && (kmask == nullptr || kmask[(int64_t)b * S + key] != 0);
        const unsigned long long bits = __ballot(valid);
        if (lane == 0) lbits[t] = bits;
    }

    // ring stage s = {K(s+2), V(s)}, s = -2 .. T-1, in slot (s+2) % NST.  Waves 0-1 fetch the K image, waves 2-3 the V image
    // (four 1-KiB pieces each), so the source is one wave-uniform base (advanced on the scalar unit) plus a per-lane
    // constant offset: no per-stage address arithmetic on the VALU.  Tiles outside [0, T) are clamped (never consumed);
    // the last tile may be partial, its lanes use offsets clamped to the last row.
    const uint16_t* img_base = wid < 2 ? kbase : vbase;
    const int tile_shift = wid < 2 ? 2 : 0;
    int off_reg[PIECES], off_last[PIECES];
    const int last_rows = S - (T - 1) * 64;  // rows of the last tile
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
        const int prow = ((wid * PIECES + i) & 7) * 8 + (lane >> 3);
        const int dcol = (((lane & 7) ^ swz(prow)) << 3);
        off_reg[i] = prow * (int)ld + dcol;
        off_last[i] = min(prow, last_rows - 1) * (int)ld + dcol;
    }
    auto dma = [&](int s, int slot) {
        char* st = smem + slot * kRingStage + (wid >= 2 ? 8192 : 0) + (wid & 1) * 4096;
        const int tile = min(max(s + tile_shift, 0), T - 1);  // scalar
        const uint16_t* src = img_base + (int64_t)tile * 64 * ld;
        const bool last = tile == T - 1;
#pragma unroll
        for (int i = 0; i < PIECES; ++i) glds16(src + (last ? off_last[i] : off_reg[i]), st + i * 1024);
    };
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) dma(-2 + i, i);
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = scale_frag(qf[s], c);

    f32x16 oacc[2], negref;
    float ref, l_run;
    bool has_ref;
    bf16x8 pfA[4], pfB[4];
    f32x16 sA[2], sB[2];
    int slot, slot_next, next_stage;

    auto stage_sync = [&]() -> const char* {  // the next stage has landed for everyone; its successor may be requested
#ifndef AB_NODMA
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES * (DEPTH - 1)) : "memory");
#endif
#ifndef AB_NOBAR
        lds_barrier();
#endif
        const char* st = smem + slot * kRingStage;
        slot = slot + 1 == NST ? 0 : slot + 1;
        return st;
    };
    auto request = [&]() {
#ifndef AB_NODMA
        dma(next_stage, slot_next);
#endif
        ++next_stage;
        slot_next = slot_next + 1 == NST ? 0 : slot_next + 1;
    };
    auto qk = [&](const char* st, f32x16 (&sc)[2]) {
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            sc[blk] = mfma32(frag_R(st, 32 * blk, 0, lane), qf[0], negref);
#pragma unroll
            for (int s = 1; s < 4; ++s) sc[blk] = mfma32(frag_R(st, 32 * blk, s, lane), qf[s], sc[blk]);
        }
    };
    auto pv = [&](const char* st, bf16x8 (&pf)[4]) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            oacc[0] = mfma32(frag_T(st + 8192, 16 * s, 0, lane), pf[s], oacc[0]);
            oacc[1] = mfma32(frag_T(st + 8192, 16 * s, 1, lane), pf[s], oacc[1]);
        }
    };
    // exact softmax of tile j from scores s_cur (relative to ref); also keeps s_nxt (already computed against the old ref)
    // and O, l consistent when the reference moves
    auto exact_softmax = [&](int j, f32x16 (&s_cur)[2], f32x16 (&s_nxt)[2], bf16x8 (&pf)[4]) {
        const unsigned long long bits = lbits[j];
        if (bits != ~0ull) {
            mask_scores_bits(s_cur[0], bits, 0, j * 64, INT_MIN, INT_MAX, hh);
            mask_scores_bits(s_cur[1], bits, 1, j * 64, INT_MIN, INT_MAX, hh);
        }
        float mt = fmaxf(reg_max16(s_cur[0]), reg_max16(s_cur[1]));
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        const bool move = has_ref ? (mt > kDefer) : (mt > kNegInf);
        if (__any(move)) {
            const float shift = has_ref ? fmaxf(mt, 0.f) : (mt > kNegInf ? mt : 0.f);
            const float alpha = has_ref ? __builtin_amdgcn_exp2f(-shift) : 1.0f;
            has_ref = has_ref || (mt > kNegInf);
            ref += shift;
            l_run *= alpha;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                s_cur[0][i] -= shift;
                s_cur[1][i] -= shift;
                s_nxt[0][i] -= shift;
                s_nxt[1][i] -= shift;
                oacc[0][i] *= alpha;
                oacc[1][i] *= alpha;
                negref[i] = -ref;
            }
        }
        float psum = 0.f;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float pv_ = __builtin_amdgcn_exp2f(s_cur[blk][i]);
                s_cur[blk][i] = pv_;
                psum += pv_;
            }
            pf[2 * blk] = acc_to_frag(s_cur[blk], 0);
            pf[2 * blk + 1] = acc_to_frag(s_cur[blk], 1);
        }
        l_run += psum;
    };
    // exact (unscheduled) iteration for tile j: stage j-1 = {K(j+1), V(j-1)}
    auto iter_exact = [&](int j, f32x16 (&s_cur)[2], f32x16 (&s_nxt)[2], bf16x8 (&pf_cur)[4], bf16x8 (&pf_prv)[4]) {
        const char* st = stage_sync();
        request();
        if (!wave_live) return;
        if (j + 1 < T) qk(st, s_nxt);
        pv(st, pf_prv);
        exact_softmax(j, s_cur, s_nxt, pf_cur);
    };
    // fast iteration: branch-free block; `bad` collects rows whose sum left the safe range (handled by a second, exact pass)
    float bad_sum = 0.f;
    auto iter_fast = [&](f32x16 (&s_cur)[2], f32x16 (&s_nxt)[2], bf16x8 (&pf_cur)[4], bf16x8 (&pf_prv)[4]) {
        const char* st = stage_sync();
        request();
        // four groups of 4 MFMAs, each fed by 16 VGPRs of fragments that are read one group ahead
        bf16x8 k0[4], k1[4], va[2][2], vb[2][2];
#ifdef AB_NOFRAG
#define frag_R(a, b, c, d) qf[(c) & 3]
#define frag_T(a, b, c, d) qf[((b) / 16 + (c)) & 3]
#endif
#pragma unroll
        for (int s = 0; s < 4; ++s) k0[s] = frag_R(st, 0, s, lane);
#pragma unroll
        for (int s = 0; s < 4; ++s) k1[s] = frag_R(st, 32, s, lane);
        s_nxt[0] = mfma32(k0[0], qf[0], negref);
#pragma unroll
        for (int s = 1; s < 4; ++s) s_nxt[0] = mfma32(k0[s], qf[s], s_nxt[0]);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            va[s][0] = frag_T(st + 8192, 16 * s, 0, lane);
            va[s][1] = frag_T(st + 8192, 16 * s, 1, lane);
        }
        s_nxt[1] = mfma32(k1[0], qf[0], negref);
#pragma unroll
        for (int s = 1; s < 4; ++s) s_nxt[1] = mfma32(k1[s], qf[s], s_nxt[1]);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            vb[s][0] = frag_T(st + 8192, 16 * (s + 2), 0, lane);
            vb[s][1] = frag_T(st + 8192, 16 * (s + 2), 1, lane);
        }
#ifdef AB_NOFRAG
#undef frag_R
#undef frag_T
#endif
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            oacc[0] = mfma32(va[s][0], pf_prv[s], oacc[0]);
            oacc[1] = mfma32(va[s][1], pf_prv[s], oacc[1]);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            oacc[0] = mfma32(vb[s][0], pf_prv[s + 2], oacc[0]);
            oacc[1] = mfma32(vb[s][1], pf_prv[s + 2], oacc[1]);
        }
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int h8 = 0; h8 < 2; ++h8) {
                float pr[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) pr[i] = __builtin_amdgcn_exp2f(s_cur[blk][8 * h8 + i]);
                a0 += pr[0] + pr[4];
                a1 += pr[1] + pr[5];
                a2 += pr[2] + pr[6];
                a3 += pr[3] + pr[7];
                const uint4 w = {pack_bf16x2(pr[0], pr[1]), pack_bf16x2(pr[2], pr[3]), pack_bf16x2(pr[4], pr[5]), pack_bf16x2(pr[6], pr[7])};
                pf_cur[2 * blk + h8] = __builtin_bit_cast(bf16x8, w);
            }
        const float psum = (a0 + a1) + (a2 + a3);
        l_run += psum;
        bad_sum = fmaxf(bad_sum, psum);  // NaN-proof enough: inf propagates, NaN needs inf - inf which exp2 cannot produce here
        // schedule: K fragments, a few exps while they land, then every MFMA followed by its share of the VALU work
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
        __builtin_amdgcn_sched_group_barrier(0x400, 4, 0);
#pragma unroll
        for (int grp = 0; grp < 4; ++grp) {
            if (grp == 1 || grp == 2) __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);  // V fragments one group ahead
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x400, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
            }
        }
    };

    int* lflag = reinterpret_cast<int*>(lbits + T);  // workgroup-wide "a row sum overflowed" flag
    // first tile that is not all-valid: fast iterations stop before it
    int jv = T;

    for (int attempt = 0; attempt < 2; ++attempt) {
#pragma unroll
        for (int i = 0; i < 16; ++i) oacc[0][i] = oacc[1][i] = negref[i] = 0.f;
        ref = 0.f;
        l_run = 0.f;
        has_ref = false;
        slot = 0;
        slot_next = DEPTH % NST;
        next_stage = -2 + DEPTH;
        if (attempt == 1) {
            __syncthreads();  // everyone has left the first pass (and drained its requests) before the ring restarts
#pragma unroll
            for (int i = 0; i < DEPTH; ++i) dma(-2 + i, i);
        }
        // prologue: S(0) from stage -2, S(1) from stage -1, exact softmax of tile 0
        {
            const char* st = stage_sync();
            request();
            if (wave_live) qk(st, sA);
            st = stage_sync();
            request();
            if (attempt == 0) {
                jv = T;
                for (int t = T - 1; t >= 1; --t)
                    if (lbits[t] != ~0ull) jv = t;
            }
            if (wave_live) {
                if (T > 1) qk(st, sB);
                exact_softmax(0, sA, sB, pfA);
            }
        }
        // invariant from here: scores of tile j in sB (relative to ref), P(j-1) in pfA, PV(j-1) still to be accumulated
        int j = 1;
        const bool fast_ok = attempt == 0 && wave_live && __all(has_ref);
        if (fast_ok) {
            // Drain every vector-memory operation the COMPILER tracks (spill reloads, Q) before the loop: a value it first
            // uses inside the loop would otherwise get its s_waitcnt vmcnt there, in every iteration, and - vmcnt being
            // in-order - that wait would also drain the untracked LDS-DMA ring each time.
            __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
            {
#pragma unroll
                for (int i = 0; i < DEPTH; ++i) (void)0;
            }
            for (; j + 1 < jv; j += 2) {
                iter_fast(sB, sA, pfB, pfA);
                iter_fast(sA, sB, pfA, pfB);
            }
        }
        for (; j < T; ++j) {
            iter_exact(j, sB, sA, pfB, pfA);
#pragma unroll
            for (int q = 0; q < 2; ++q) sB[q] = sA[q];
#pragma unroll
            for (int q = 0; q < 4; ++q) pfA[q] = pfB[q];
        }
        {
            const char* st = stage_sync();  // stage T-1 holds V(T-1)
            if (wave_live) pv(st, pfA);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // requests past the last stage (dummies)
        if (attempt == 0) {
            // did any row sum leave the safe range (or turn into inf / nan)?  Then redo the workgroup on the exact path.
            if (tid == 0) *lflag = 0;
            __syncthreads();
            if (wave_live && __any(!(bad_sum <= kBig))) {
                if (lane == 0) *lflag = 1;
            }
            __syncthreads();
            if (*lflag == 0) break;
        }
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    if (qrow < S) {
        uint16_t* orow = out + ((int64_t)b * S + qrow) * nh * 64 + head * 64;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int dv = 32 * blk + 8 * g + 4 * hh;
                const uint2 w = {pack_bf16x2(oacc[blk][4 * g] * inv, oacc[blk][4 * g + 1] * inv),
                                 pack_bf16x2(oacc[blk][4 * g + 2] * inv, oacc[blk][4 * g + 3] * inv)};
                *reinterpret_cast<uint2*>(orow + dv) = w;
            }
        if (hh == 0) lse[((int64_t)b * nh + head) * S + qrow] = l_tot > 0.f ? (ref + __log2f(l_tot)) * 0.69314718055994531f : __builtin_huge_valf();
    }
}

// ---------------------------------------------------------------------------------------------------------------
// delta[b, h, q] = sum_d dO[q, d] * O[q, d]
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_delta_kernel(const uint16_t* __restrict__ o, const uint16_t* __restrict__ d_o,
                                                         float* __restrict__ delta, int64_t T, int S, int nh) {
    const int64_t total = T * nh * 8;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total + 7; i += (int64_t)gridDim.x * 256) {
        const bool ok = i < total;
        const int64_t row = ok ? i >> 3 : 0;  // (token, head)
        const int c = (int)(i & 7);
        float s = 0.f;
        if (ok) {
            const uint4 a = *reinterpret_cast<const uint4*>(o + row * 64 + c * 8);
            const uint4 g = *reinterpret_cast<const uint4*>(d_o + row * 64 + c * 8);
            s = bf16lo(a.x) * bf16lo(g.x) + bf16hi(a.x) * bf16hi(g.x) + bf16lo(a.y) * bf16lo(g.y) + bf16hi(a.y) * bf16hi(g.y) +
                bf16lo(a.z) * bf16lo(g.z) + bf16hi(a.z) * bf16hi(g.z) + bf16lo(a.w) * bf16lo(g.w) + bf16hi(a.w) * bf16hi(g.w);
        }
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        s += __shfl_xor(s, 4, 64);
        if (ok && c == 0) {
            const int64_t tok = row / nh;
            const int h = (int)(row % nh);
            delta[((tok / S) * nh + h) * S + tok % S] = s;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// dQ: same geometry as the forward.  LDS per stage: K image (row + transposed reads) + V image + mask bytes.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kDqStage = 2 * 8192 + 64 + 16;  // K image, V image, mask bytes, all-valid flag

__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ d_o,
                                                             const float* __restrict__ lse, const float* __restrict__ delta,
                                                             uint16_t* __restrict__ dqkv, const uint8_t* __restrict__ kmask,
                                                             int S, int nh, int window, float scale,
                                                             const float* __restrict__ rope_cos,
                                                             const float* __restrict__ rope_sin, int64_t pos_batch_stride) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, hh = lane >> 5;
    int qblk, head, b;
    decode_block((S + 127) / 128, nh, qblk, head, b);
    const int Q0 = qblk * 128;
    const int q0 = Q0 + wid * 32;
    const int64_t ld = (int64_t)3 * nh * 64;
    const uint16_t* qbase = qkv + (int64_t)b * S * ld + head * 64;
    const uint16_t* kbase = qbase + nh * 64;
    const uint16_t* vbase = qbase + 2 * nh * 64;
    const int64_t ldo = (int64_t)nh * 64;
    const uint16_t* dobase = d_o + (int64_t)b * S * ldo + head * 64;

    const int qrow = q0 + (lane & 31);
    const int qrow_c = qrow < S ? qrow : S - 1;
    const int lo = window < 0 ? INT_MIN : qrow - window, hi = window < 0 ? INT_MAX : qrow + window;
    const float c = scale * kLog2e;
    bf16x8 qf[4], dof[4];  // qf = Q * (scale * log2 e): K qf^T - lse*log2(e) is log2 of the probability
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        qf[s] = scale_frag(*reinterpret_cast<const bf16x8*>(qbase + (int64_t)qrow_c * ld + 16 * s + 8 * hh), c);
        dof[s] = *reinterpret_cast<const bf16x8*>(dobase + (int64_t)qrow_c * ldo + 16 * s + 8 * hh);
    }
    const int64_t stat = ((int64_t)b * nh + head) * S + qrow_c;
    const float lse2 = lse[stat] * kLog2e;  // +inf for rows with no visible key -> p = 0
    const float dlt = delta[stat];

    const int Q1 = min(S, Q0 + 128) - 1;
    int klo = 0, khi = S - 1, wlo = 0, whi = S - 1;
    if (window >= 0) {
        klo = max(0, Q0 - window);
        khi = min(S - 1, Q1 + window);
        wlo = max(0, q0 - window);
        whi = min(S - 1, q0 + 31 + window);
    }
    const bool wave_live = q0 < S;
    const int t_lo = klo / 64, t_hi = khi / 64;

    f32x16 dq[2], lse_init, dlt_init;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        dq[0][i] = dq[1][i] = 0.f;
        lse_init[i] = -lse2;
        dlt_init[i] = -dlt;
    }

    TileRegs64 kr, vr;
    uint8_t mreg = 0;
    auto gload = [&](int t) {
        gload64(kr, kbase, ld, t * 64, S, tid);
        gload64(vr, vbase, ld, t * 64, S, tid);
        if (tid < 64) {
            const int key = t * 64 + tid;
            mreg = key < S ? (kmask ? kmask[(int64_t)b * S + key] : (uint8_t)1) : (uint8_t)0;
        }
    };
    auto lstore = [&](int stage) {
        char* st = smem + stage * kDqStage;
        lstore64_R(st, kr, tid);
        lstore64_R(st + 8192, vr, tid);
        if (tid < 64) {
            reinterpret_cast<uint8_t*>(st + 16384)[tid] = mreg;
            const unsigned long long valid = __ballot(mreg != 0);
            if (tid == 0) *reinterpret_cast<int*>(st + 16448) = (valid == ~0ull) ? 1 : 0;
        }
    };

    gload(t_lo);
    lstore(0);
    __syncthreads();

    for (int t = t_lo; t <= t_hi; ++t) {
        const int stage = (t - t_lo) & 1;
        const char* st = smem + stage * kDqStage;
        const bool more = t < t_hi;
        if (more) gload(t + 1);
        const int key0 = t * 64;
        if (wave_live && key0 <= whi && key0 + 63 >= wlo) {
            f32x16 sacc[2], dp[2];
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                // row constants as initial accumulators (log2 p, and dP - delta): the first MFMA of each chain reads them
                // from two vectors that stay resident for the whole key sweep (the MFMA's C and D may differ), so no
                // per-tile register fills are needed
                sacc[blk] = mfma32(frag_R(st, 32 * blk, 0, lane), qf[0], lse_init);
                dp[blk] = mfma32(frag_R(st + 8192, 32 * blk, 0, lane), dof[0], dlt_init);
#pragma unroll
                for (int s = 1; s < 4; ++s) {
                    sacc[blk] = mfma32(frag_R(st, 32 * blk, s, lane), qf[s], sacc[blk]);
                    dp[blk] = mfma32(frag_R(st + 8192, 32 * blk, s, lane), dof[s], dp[blk]);
                }
            }
            if (!tile_unmasked(*reinterpret_cast<const int*>(st + 16448), key0, q0, window)) {
                const uint8_t* mb = reinterpret_cast<const uint8_t*>(st + 16384);
                mask_scores_keyrows(sacc[0], mb, 0, key0, lo, hi, hh);
                mask_scores_keyrows(sacc[1], mb, 1, key0, lo, hi, hh);
            }
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float p = __builtin_amdgcn_exp2f(sacc[blk][i]);
                    sacc[blk][i] = p * dp[blk][i];  // dS^T / scale (the scale is applied once, to dQ)
                }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const bf16x8 dsf = acc_to_frag(sacc[s >> 1], s & 1);
                dq[0] = mfma32(frag_T(st, 16 * s, 0, lane), dsf, dq[0]);  // K^T from the same image as the row reads
                dq[1] = mfma32(frag_T(st, 16 * s, 1, lane), dsf, dq[1]);
            }
        }
        if (more) lstore(stage ^ 1);
        __syncthreads();
    }

    if (qrow < S) {
        uint16_t* drow = dqkv + ((int64_t)b * S + qrow) * ld + head * 64;
        if (rope_cos) {  // backward of apply_rotary_pos_emb: dims d / d+32 are the two accumulator blocks of this lane
            const int64_t prow = (int64_t)b * pos_batch_stride + qrow;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 lo4 = {dq[0][4 * g], dq[0][4 * g + 1], dq[0][4 * g + 2], dq[0][4 * g + 3]};
                f32x4 hi4 = {dq[1][4 * g], dq[1][4 * g + 1], dq[1][4 * g + 2], dq[1][4 * g + 3]};
                rope_rotate4<true>(lo4, hi4, rope_cos + prow * 32, rope_sin + prow * 32, 8 * g + 4 * hh);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    dq[0][4 * g + r] = lo4[r];
                    dq[1][4 * g + r] = hi4[r];
                }
            }
        }
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = 32 * blk + 8 * g + 4 * hh;
                const uint2 w = {pack_bf16x2(dq[blk][4 * g] * scale, dq[blk][4 * g + 1] * scale),
                                 pack_bf16x2(dq[blk][4 * g + 2] * scale, dq[blk][4 * g + 3] * scale)};
                *reinterpret_cast<uint2*>(drow + d) = w;
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// dK, dV: one workgroup = 4 waves = 128 keys of one (batch, head); each wave owns 32 keys (key on the lane) and keeps
// dK^T, dV^T (64 x 32 each) in accumulators while the workgroup sweeps query tiles of 64 rows.
// LDS per stage: Q image + dO image (8 KiB each, row and transposed reads) + -lse*log2(e) and -delta (64 floats each).
// ---------------------------------------------------------------------------------------------------------------
constexpr int kDkvStage = 2 * 8192 + 512;  // Q image, dO image, -lse*log2e and -delta rows

__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ d_o,
                                                              const float* __restrict__ lse, const float* __restrict__ delta,
                                                              uint16_t* __restrict__ dqkv, const uint8_t* __restrict__ kmask,
                                                              int S, int nh, int window, float scale,
                                                              const float* __restrict__ rope_cos,
                                                              const float* __restrict__ rope_sin, int64_t pos_batch_stride) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, hh = lane >> 5;
    int kblk, head, b;
    decode_block((S + 127) / 128, nh, kblk, head, b);
    const int K0 = kblk * 128;
    const int k0 = K0 + wid * 32;
    const int64_t ld = (int64_t)3 * nh * 64;
    const uint16_t* qbase = qkv + (int64_t)b * S * ld + head * 64;
    const uint16_t* kbase = qbase + nh * 64;
    const uint16_t* vbase = qbase + 2 * nh * 64;
    const int64_t ldo = (int64_t)nh * 64;
    const uint16_t* dobase = d_o + (int64_t)b * S * ldo + head * 64;
    const float* lse_bh = lse + ((int64_t)b * nh + head) * S;
    const float* dlt_bh = delta + ((int64_t)b * nh + head) * S;

    const int krow = k0 + (lane & 31);
    const int krow_c = krow < S ? krow : S - 1;
    const float c = scale * kLog2e;
    bf16x8 kf[4], vf[4];  // kf = K * (scale * log2 e)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        kf[s] = scale_frag(*reinterpret_cast<const bf16x8*>(kbase + (int64_t)krow_c * ld + 16 * s + 8 * hh), c);
        vf[s] = *reinterpret_cast<const bf16x8*>(vbase + (int64_t)krow_c * ld + 16 * s + 8 * hh);
    }
    const bool key_ok = krow < S && (kmask ? kmask[(int64_t)b * S + krow] != 0 : true);
    const bool keys_all_ok = __all(key_ok);
    const int lo = window < 0 ? INT_MIN : krow - window, hi = window < 0 ? INT_MAX : krow + window;

    const int K1 = min(S, K0 + 128) - 1;
    int qlo = 0, qhi = S - 1, wlo = 0, whi = S - 1;
    if (window >= 0) {
        qlo = max(0, K0 - window);
        qhi = min(S - 1, K1 + window);
        wlo = max(0, k0 - window);
        whi = min(S - 1, k0 + 31 + window);
    }
    const bool wave_live = k0 < S;
    const int t_lo = qlo / 64, t_hi = qhi / 64;

    f32x16 dk[2], dv[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) dk[0][i] = dk[1][i] = dv[0][i] = dv[1][i] = 0.f;

    TileRegs64 qr, gr;
    float sreg = 0.f;
    auto gload = [&](int t) {
        gload64(qr, qbase, ld, t * 64, S, tid);
        gload64(gr, dobase, ldo, t * 64, S, tid);
        if (tid < 128) {
            const int q = t * 64 + (tid & 63);
            if (tid < 64) sreg = q < S ? -lse_bh[q] * kLog2e : kNegInf;  // -lse in log2 units; rows past S contribute p = 0
            else sreg = q < S ? -dlt_bh[q] : 0.f;
        }
    };
    auto lstore = [&](int stage) {
        char* st = smem + stage * kDkvStage;
        lstore64_R(st, qr, tid);
        lstore64_R(st + 8192, gr, tid);
        if (tid < 128) reinterpret_cast<float*>(st + 16384)[tid] = sreg;
    };

    gload(t_lo);
    lstore(0);
    __syncthreads();

    for (int t = t_lo; t <= t_hi; ++t) {
        const int stage = (t - t_lo) & 1;
        const char* st = smem + stage * kDkvStage;
        const bool more = t < t_hi;
        if (more) gload(t + 1);
        const int qt0 = t * 64;
        if (wave_live && qt0 <= whi && qt0 + 63 >= wlo) {
            const float* nlse = reinterpret_cast<const float*>(st + 16384);
            const float* ndlt = nlse + 64;
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {  // two 32-query blocks of the tile
                f32x16 sacc, dp;
                // initial accumulators: row constants -lse*log2(e) and -delta (rows = queries)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(nlse + 32 * qb + 8 * g + 4 * hh);
                    const f32x4 d = *reinterpret_cast<const f32x4*>(ndlt + 32 * qb + 8 * g + 4 * hh);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        sacc[4 * g + r] = a[r];
                        dp[4 * g + r] = d[r];
                    }
                }
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    sacc = mfma32(frag_R(st, 32 * qb, s, lane), kf[s], sacc);           // log2 p: rows q, col key
                    dp = mfma32(frag_R(st + 8192, 32 * qb, s, lane), vf[s], dp);        // dP - delta
                }
                // every (query, key) pair of this 32 x 32 block visible?  (queries past S carry -inf and give p = 0 anyway)
                const int qb0 = qt0 + 32 * qb;
                const bool plain = keys_all_ok && (window < 0 || (qb0 >= k0 + 31 - window && qb0 + 31 <= k0 + window));
                if (plain) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const float p = __builtin_amdgcn_exp2f(sacc[i]);
                        sacc[i] = p;
                        dp[i] = p * dp[i];  // dS / scale (applied once, to dK)
                    }
                } else {
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int i = 4 * g + r;
                            const int q = qb0 + 8 * g + 4 * hh + r;
                            const bool ok = key_ok & (q >= lo) & (q <= hi);
                            const float p = ok ? __builtin_amdgcn_exp2f(sacc[i]) : 0.f;
                            sacc[i] = p;
                            dp[i] = p * dp[i];  // dS / scale (applied once, to dK)
                        }
                }
#pragma unroll
                for (int sp = 0; sp < 2; ++sp) {
                    const bf16x8 pf = acc_to_frag(sacc, sp);
                    const bf16x8 dsf = acc_to_frag(dp, sp);
                    const int r0 = 32 * qb + 16 * sp;
                    dv[0] = mfma32(frag_T(st + 8192, r0, 0, lane), pf, dv[0]);  // dO^T and Q^T from the row-read images
                    dv[1] = mfma32(frag_T(st + 8192, r0, 1, lane), pf, dv[1]);
                    dk[0] = mfma32(frag_T(st, r0, 0, lane), dsf, dk[0]);
                    dk[1] = mfma32(frag_T(st, r0, 1, lane), dsf, dk[1]);
                }
            }
        }
        if (more) lstore(stage ^ 1);
        __syncthreads();
    }

    if (krow < S) {
        uint16_t* dkrow = dqkv + ((int64_t)b * S + krow) * ld + nh * 64 + head * 64;
        uint16_t* dvrow = dkrow + nh * 64;
        if (rope_cos) {
            const int64_t prow = (int64_t)b * pos_batch_stride + krow;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 lo4 = {dk[0][4 * g], dk[0][4 * g + 1], dk[0][4 * g + 2], dk[0][4 * g + 3]};
                f32x4 hi4 = {dk[1][4 * g], dk[1][4 * g + 1], dk[1][4 * g + 2], dk[1][4 * g + 3]};
                rope_rotate4<true>(lo4, hi4, rope_cos + prow * 32, rope_sin + prow * 32, 8 * g + 4 * hh);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    dk[0][4 * g + r] = lo4[r];
                    dk[1][4 * g + r] = hi4[r];
                }
            }
        }
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = 32 * blk + 8 * g + 4 * hh;
                *reinterpret_cast<uint2*>(dkrow + d) =
                    uint2{pack_bf16x2(dk[blk][4 * g] * scale, dk[blk][4 * g + 1] * scale),
                          pack_bf16x2(dk[blk][4 * g + 2] * scale, dk[blk][4 * g + 3] * scale)};
                *reinterpret_cast<uint2*>(dvrow + d) =
                    uint2{pack_bf16x2(dv[blk][4 * g], dv[blk][4 * g + 1]), pack_bf16x2(dv[blk][4 * g + 2], dv[blk][4 * g + 3])};
            }
    }
}

}  // namespace

extern "C" {

int cm3p_attn_fwd(const void* qkv, void* out, float* lse, const uint8_t* key_mask, int B, int S, int nh, int window,
                  float scale, void* stream) {
    CM3P_REQUIRE(qkv && out && lse && B > 0 && S > 0 && nh > 0 && scale > 0.f);
    CM3P_REQUIRE(cm3p_aligned16(qkv) && cm3p_aligned16(out));
    hipStream_t s = static_cast<hipStream_t>(stream);
    // QSUB = 1 (32 queries per wave, 3 waves per SIMD) measured faster than QSUB = 2 (64 per wave, compiler-scheduled):
    // 2.29 ms vs 3.49 ms per C2 global layer.  The two-chain variant needs a hand-placed MFMA/VALU interleave to pay.
    // CM3P_ATTN_FWD selects the forward variant while tuning: "a" 4 waves x 32 q, 3-stage ring, 3 WG/CU (default);
    // "b" 4 waves x 64 q, 4-stage ring, 2 WG/CU; "c" 8 waves x 32 q, 4-stage ring; "d" 4 x 32 q, 4-stage ring, 2 WG/CU; "regs"
    static const char* variant = getenv("CM3P_ATTN_FWD");
    const char v = variant ? variant[0] : 'a';
    if (v != 'r' && (S + 63) / 64 <= 1024) {
        int rc;
        if (v == 'p' && window < 0) {
            constexpr int NSTP = 4;
            const int T = (S + 63) / 64;
            static bool attr_set = false;
            if (!attr_set) {
                if (hipFuncSetAttribute((const void*)attn_fwd_pipe_kernel<NSTP>, hipFuncAttributeMaxDynamicSharedMemorySize, NSTP * kRingStage + 1024 * 8 + 32) != hipSuccess) return CM3P_ERR_LAUNCH;
                attr_set = true;
            }
            const dim3 grid(((S + 127) / 128) * nh * B);
            attn_fwd_pipe_kernel<NSTP><<<grid, 256, NSTP * kRingStage + (size_t)T * 8 + 32, s>>>((const uint16_t*)qkv, (uint16_t*)out, lse, key_mask, S, nh, scale);
            rc = CM3P_OK;
        } else if (v == 'b') rc = launch_fwd_ring<4, 2, 4, 2>(qkv, out, lse, key_mask, B, S, nh, window, scale, s);
        else if (v == 'c') rc = launch_fwd_ring<8, 1, 4, 1>(qkv, out, lse, key_mask, B, S, nh, window, scale, s);
        else if (v == 'd') rc = launch_fwd_ring<4, 1, 4, 2>(qkv, out, lse, key_mask, B, S, nh, window, scale, s);
        else rc = launch_fwd_ring<4, 1, 3, 3>(qkv, out, lse, key_mask, B, S, nh, window, scale, s);
        if (rc != CM3P_OK) return rc;
        CM3P_LAUNCH_CHECK();
        return CM3P_OK;
    }
    const dim3 grid(((S + 127) / 128) * nh * B);
    attn_fwd_kernel<1><<<grid, 256, 2 * kFwdStage, s>>>((const uint16_t*)qkv, (uint16_t*)out, lse, key_mask, S, nh, window, scale);
    CM3P_LAUNCH_CHECK();
#ifdef CM3P_STAMPS
    {
        unsigned long long h[16];
        hipDeviceSynchronize();
        hipMemcpyFromSymbol(h, HIP_SYMBOL(g_seg), sizeof(h));
        const double n = (double)h[8] * ((S + 63) / 64);
        fprintf(stderr, "[stamps] old fwd window=%d waves=%llu  cycles per wave-tile:", window, h[8]);
        for (int k = 0; k < 8; ++k) fprintf(stderr, " seg%d=%.0f", k, h[k] / n);
        fprintf(stderr, "\n");
        unsigned long long z[16] = {0};
        hipMemcpyToSymbol(HIP_SYMBOL(g_seg), z, sizeof(z));
    }
#endif
    return CM3P_OK;
}

int cm3p_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse, float* delta, void* dqkv,
                  const uint8_t* key_mask, int B, int S, int nh, int window, float scale, const float* cos_tab,
                  const float* sin_tab, int64_t pos_batch_stride, void* stream) {
    CM3P_REQUIRE((cos_tab == nullptr) == (sin_tab == nullptr));
    CM3P_REQUIRE(pos_batch_stride == 0 || pos_batch_stride == S);
    CM3P_REQUIRE(qkv && out && dout && lse && delta && dqkv && B > 0 && S > 0 && nh > 0 && scale > 0.f);
    CM3P_REQUIRE(cm3p_aligned16(qkv) && cm3p_aligned16(out) && cm3p_aligned16(dout) && cm3p_aligned16(dqkv));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t T = (int64_t)B * S;
    int64_t blocks = (T * nh * 8 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    attn_delta_kernel<<<(int)blocks, 256, 0, s>>>((const uint16_t*)out, (const uint16_t*)dout, delta, T, S, nh);
    CM3P_LAUNCH_CHECK();
    const dim3 grid(((S + 127) / 128) * nh * B);
    attn_bwd_dq_kernel<<<grid, 256, 2 * kDqStage, s>>>((const uint16_t*)qkv, (const uint16_t*)dout, lse, delta, (uint16_t*)dqkv,
                                                       key_mask, S, nh, window, scale, cos_tab, sin_tab, pos_batch_stride);
    CM3P_LAUNCH_CHECK();
    attn_bwd_dkv_kernel<<<grid, 256, 2 * kDkvStage, s>>>((const uint16_t*)qkv, (const uint16_t*)dout, lse, delta, (uint16_t*)dqkv,
                                                         key_mask, S, nh, window, scale, cos_tab, sin_tab, pos_batch_stride);
    CM3P_LAUNCH_CHECK();
    return CM3P_OK;
}

}  // extern "C"
