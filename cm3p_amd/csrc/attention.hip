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
#include <stdlib.h>

#include "attn_common.h"

#ifndef CM3P_BABL
#define CM3P_BABL 0  // timing-only ablations of attn_bwd_dkv_kernel: 1 no products, 2 no epilogue, 4 no tile DMA
#endif

namespace {

// a * c + b * s with a FIXED rounding sequence: the product b * s is rounded on its own (the empty asm keeps the compiler from folding it
// into the addition), then one fma.  Left to the compiler, `a * c + b * s` becomes fma(a, c, b * s) in one template instance and
// fma(b, s, a * c) in another (r04: the masked and the unmasked instance of the dQ sweep then disagreed in the last bit of two
// elements out of 10^5, which is what the packed-equals-padded test compares).
__device__ __forceinline__ float rot_fma(float a, float c, float b, float s) {
    float t = b * s;
    asm volatile("" : "+v"(t));
    return __builtin_fmaf(a, c, t);
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

// The same for the LDS-DMA staged kernels: one validity DWORD per key (the mask byte zero-extended; MASK == false: no mask was
// given), keys past the sequence are cut by `hi` (the caller clamps it to S - 1).
template <bool MASK>
__device__ __forceinline__ void mask_scores_keyrows_d(f32x16& s, const uint32_t* maskd, int blk, int key0, int lo, int hi, int hh) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int kl = 32 * blk + 8 * g + 4 * hh;
        uint4 mb = uint4{1u, 1u, 1u, 1u};
        if constexpr (MASK) mb = *reinterpret_cast<const uint4*>(maskd + kl);
        const uint32_t m[4] = {mb.x, mb.y, mb.z, mb.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = key0 + kl + r;
            const bool ok = (m[r] != 0u) & (key >= lo) & (key <= hi);
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
// PRE: the q third already holds q * scale * log2(e) (cm3p_qkv_gemm_rope's q_scale): the MFMA delivers scores in exp2 units and the
// reference point rides in as the initial accumulator.  !PRE: plain q; scale * log2(e) and the reference point are applied by one
// fp32 v_fma per score - q is never re-rounded to bf16.
// BAND: sliding-window layers (window >= 0); the global instance (window = -1 at compile time) carries no band arithmetic and
// shows up as its own row in a profile.
template <int QSUB, bool PRE, bool BAND>
__global__ __launch_bounds__(256, QSUB == 1 ? 2 : 1) void attn_fwd_kernel(const uint16_t* __restrict__ qkv, uint16_t* __restrict__ out,
                                                                        float* __restrict__ lse, const uint8_t* __restrict__ kmask,
                                                                        int Smax, int nh, int window_arg, float scale, VarLen vl) {
    const int window = BAND ? window_arg : -1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int QW = 32 * QSUB;  // queries per wave
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, hh = lane >> 5;
    int qblk, head, b;
    if (BAND) decode_block_band((Smax + 4 * QW - 1) / (4 * QW), nh, qblk, head, b);
    else decode_block((Smax + 4 * QW - 1) / (4 * QW), nh, qblk, head, b);
    const int Q0 = qblk * (4 * QW);
    const SeqView sv(vl, b, head, Smax, nh);
    const int S = sv.S;
    if (Q0 >= S) return;  // (unpadded batches: the grid is sized for the longest sequence)
    const int q0 = Q0 + wid * QW;
    const int64_t ld = (int64_t)3 * nh * 64;
    const uint16_t* qbase = qkv + sv.row0 * ld + head * 64;
    const uint16_t* kbase = qbase + nh * 64;
    const uint16_t* vbase = qbase + 2 * nh * 64;
    const float c = scale * kLog2e;
    constexpr float kDefer = 6.0f;

    int qrow[QSUB], lo[QSUB], hi[QSUB];
    bf16x8 qf[QSUB][4];
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
            qf[u][s] = *reinterpret_cast<const bf16x8*>(qbase + (int64_t)qc * ld + 16 * s + 8 * hh);
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
    uint8_t mraw = 1;
    bool mok = false;
    // (nothing here consumes a loaded value - see attn_bwd_dkv_kernel; the mask byte is judged in lstore())
    auto gload = [&](int t) {
        gload64(kr, kbase, ld, t * 64, S, tid);
        gload64(vr, vbase, ld, t * 64, S, tid);
        const int key = t * 64 + (tid & 63);
        mok = key < S;
        if (kmask) mraw = kmask[sv.row0 + min(max(key, 0), S - 1)];
    };
    auto lstore = [&](int stage) {
        char* st = smem + stage * kFwdStage;
        lstore64_R(st, kr, tid);
        lstore64_R(st + 8192, vr, tid);
        if (tid < 64) {
            const uint8_t mreg = mok ? mraw : (uint8_t)0;
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
        const char* st = smem + stage * kFwdStage;
        const bool more = t < t_hi;
        if (more) gload(t + 1);

        const int key0 = t * 64;
        if (wave_live && key0 <= whi && key0 + 63 >= wlo) {
            f32x16 sacc[QSUB][2];
#pragma unroll
            for (int u = 0; u < QSUB; ++u)
#pragma unroll
                for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                    for (int i = 0; i < 16; ++i) sacc[u][blk][i] = PRE ? -mc_run[u] : 0.f;
            // S^T = K Q^T (- reference): each K fragment is read from LDS once and used by every sub-block
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const bf16x8 kf = frag_R(st, 32 * blk, s, lane);
#pragma unroll
                    for (int u = 0; u < QSUB; ++u) sacc[u][blk] = mfma32(kf, qf[u][s], sacc[u][blk]);
                }
            const int all_valid = *reinterpret_cast<const int*>(st + 16448);
#pragma unroll
            for (int u = 0; u < QSUB; ++u) {
                if constexpr (!PRE) {
#pragma unroll
                    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                        for (int i = 0; i < 16; ++i) sacc[u][blk][i] = __builtin_fmaf(sacc[u][blk][i], c, -mc_run[u]);
                }
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
        }
        if (more) lstore(stage ^ 1);
        __syncthreads();
    }

#pragma unroll
    for (int u = 0; u < QSUB; ++u) {
        const float l_tot = l_run[u] + __shfl_xor(l_run[u], 32, 64);
        const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
        // (the loop's last barrier is behind every wave: the stages are free and serve as the waves' transposition buffers)
        store_rows32(smem + 4608 * wid, oacc[u][0], oacc[u][1], inv, out + (sv.row0 + q0 + 32 * u) * nh * 64 + head * 64, (int64_t)nh * 64,
                     S - (q0 + 32 * u), lane);
        if (qrow[u] < S && hh == 0)
            lse[sv.stat0 + qrow[u]] = l_tot > 0.f ? (mc_run[u] + __log2f(l_tot)) * 0.69314718055994531f : __builtin_huge_valf();
    }
}

// ---------------------------------------------------------------------------------------------------------------
// dQ (and delta[b, h, q] = sum_d dO[q, d] * O[q, d], which it computes for its own rows and publishes for the dK/dV kernel):
// same geometry as the forward.  LDS per ring slot: K image (row + transposed reads) + V image + one validity dword per key; tiles
// arrive by LDS-DMA (TileDma, attn_common.h), the result leaves as whole rows (store_rows32).
// ---------------------------------------------------------------------------------------------------------------
// Phase timestamps of the dQ sweep (trace builds only: -DCM3P_BAND_TRACE=1, tools/band_trace.py): thread 0 of every workgroup writes the
// 100 MHz wall clock at entry (0), after the prologue's requests (1), when tile i is ready (2 + 2 i) and swept (3 + 2 i), after the
// last barrier (14) and after the stores are issued (15).
#ifndef CM3P_BAND_TRACE
#define CM3P_BAND_TRACE 0
#endif
#if CM3P_BAND_TRACE
__device__ unsigned long long* g_band_trace = nullptr;
#define BAND_T(k)                                                                                          \
    do {                                                                                                   \
        if (g_band_trace && threadIdx.x == 0) g_band_trace[(size_t)blockIdx.x * 16 + (k)] = wall_clock64(); \
    } while (0)
#else
#define BAND_T(k) \
    do {          \
    } while (0)
#endif
constexpr int kDqStage = 2 * 8192 + 256;  // K image, V image, one validity dword per key
constexpr int kDqSlots = 4;               // LDS-DMA ring: tile t+3 is requested while tile t is consumed

template <bool PRE, bool MASK>
__device__ __forceinline__ void band_dq_block(char* smem, int qblk, int head, int b, const uint16_t* __restrict__ qkv,
                                              const uint16_t* __restrict__ d_o, const uint16_t* __restrict__ o_rows,
                                              const float* __restrict__ lse, float* __restrict__ delta, uint16_t* __restrict__ dqkv,
                                              const uint8_t* __restrict__ kmask, int Smax, int nh, int window, float scale,
                                              const float* __restrict__ rope_cos, const float* __restrict__ rope_sin,
                                              int64_t pos_batch_stride, VarLen vl) {
    const int tid = threadIdx.x, lane = tid & 63, hh = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Q0 = qblk * 128;
    const SeqView sv(vl, b, head, Smax, nh);
    const int S = sv.S;
    if (Q0 >= S) return;
    BAND_T(0);
    const int q0 = Q0 + wid * 32;
    const int64_t ld = (int64_t)3 * nh * 64;
    const uint16_t* qbase = qkv + sv.row0 * ld + head * 64;
    const uint16_t* kbase = qbase + nh * 64;
    const uint16_t* vbase = qbase + 2 * nh * 64;
    const int64_t ldo = (int64_t)nh * 64;
    const uint16_t* dobase = d_o + sv.row0 * ldo + head * 64;

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

    // K / V tiles (and the keys' validity bytes) by LDS-DMA into a four-slot ring: three tiles in flight per workgroup instead of
    // one - these kernels see at most five tiles, their time is the memory round trips, and the bytes in flight per CU set the
    // bandwidth they reach.  Every wave issues the same operations per tile (ND), which is what the counted vmcnt wait relies on.
    constexpr int ND = MASK ? 5 : 4;
    const TileDma dma(wid, lane);
    const int ldb = (int)ld * 2;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const uint32_t m0_k = __builtin_amdgcn_readfirstlane(lds0 + 2048u * wid);
    const uint8_t* km = MASK ? kmask + sv.row0 : nullptr;
    auto dma_tile = [&](int t) {
        const uint32_t slot = (uint32_t)((t - t_lo) & (kDqSlots - 1)) * kDqStage;
        dma.rows(m0_k + slot, kbase, ldb, t * 64, S);
        dma.rows(m0_k + slot + 8192u, vbase, ldb, t * 64, S, CM3P_AUD_T1);
        if constexpr (MASK) dma_ubyte64(lds0 + slot + 16384u, km, (uint32_t)min(t * 64 + lane, S - 1));
    };
    for (int t = t_lo; t <= min(t_hi, t_lo + 2); ++t) dma_tile(t);
    // (the tiles are on their way before the wave's own fragments are requested: the two round trips overlap)
    const int qrow = q0 + (lane & 31);
    const int qrow_c = qrow < S ? qrow : S - 1;
    const int lo = window < 0 ? INT_MIN : qrow - window, hi = window < 0 ? S - 1 : min(qrow + window, S - 1);  // (keys past S: cut here)
    const float c = scale * kLog2e;
    bf16x8 qf[4], dof[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        qf[s] = *reinterpret_cast<const bf16x8*>(qbase + (int64_t)qrow_c * ld + 16 * s + 8 * hh);
        dof[s] = *reinterpret_cast<const bf16x8*>(dobase + (int64_t)qrow_c * ldo + 16 * s + 8 * hh);
    }
    const int64_t stat = sv.stat0 + qrow_c;
    // PRE: scores arrive in exp2 units, the accumulator starts at -lse * log2(e); !PRE: it starts at -lse / scale and the
    // exponent is c * accumulator (fp32 multiply per score).  +inf lse (no visible key) -> p = 0 either way.
    const float lse2 = PRE ? lse[stat] * kLog2e : lse[stat] / scale;
    // delta[q] = sum_d dO[q, d] O[q, d]: this lane holds half of its query's dO row already; the other half sits 32 lanes
    // away.  Written out for the dK/dV kernel, which runs after this one (no separate delta launch, one less pass over dO).
    float dlt = 0.f;
    {
        const uint16_t* obase = o_rows + sv.row0 * ldo + head * 64;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const bf16x8 of = *reinterpret_cast<const bf16x8*>(obase + (int64_t)qrow_c * ldo + 16 * s + 8 * hh);
#pragma unroll
            for (int j = 0; j < 8; ++j) dlt += (float)of[j] * (float)dof[s][j];
        }
        dlt += __shfl_xor(dlt, 32, 64);
        if (hh == 0 && qrow < S) delta[stat] = dlt;
    }

    // the rotary rows of the epilogue are requested now: their round trip hides behind the key sweep instead of ending the kernel
    f32x4 rcs[4], rsn[4];
    if (rope_cos) {
        const int64_t prow = sv.pos0(b, pos_batch_stride) + qrow_c;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            rcs[g] = *reinterpret_cast<const f32x4*>(rope_cos + prow * 32 + 8 * g + 4 * hh);
            rsn[g] = *reinterpret_cast<const f32x4*>(rope_sin + prow * 32 + 8 * g + 4 * hh);
        }
    }
    f32x16 dq[2], lse_init, dlt_init;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        dq[0][i] = dq[1][i] = 0.f;
        lse_init[i] = -lse2;
        dlt_init[i] = -dlt;
    }


    BAND_T(1);
    for (int t = t_lo; t <= t_hi; ++t) {
        const char* st = smem + ((t - t_lo) & (kDqSlots - 1)) * kDqStage;
        dma_wait_barrier(ND * min(t_hi - t, 2));  // tile t has landed in every wave; the slot of tile t-1 is free
        BAND_T(2 + 2 * (t - t_lo));
        if (t + 3 <= t_hi) dma_tile(t + 3);
        const int key0 = t * 64;
        if (wave_live && key0 <= whi && key0 + 63 >= wlo) {
            const uint32_t* mb = reinterpret_cast<const uint32_t*>(st + 16384);
            int all_valid = key0 + 63 < S;
            if constexpr (MASK) all_valid = all_valid && __all(mb[lane] != 0u);
            const bool unmasked = tile_unmasked(all_valid, key0, q0, window);
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                // (a 64-key tile that touches the wave's band may still hold a 32-key block that lies outside it: a quarter of the
                //  blocks of a +-64 window)
                if (key0 + 32 * blk > whi || key0 + 32 * blk + 31 < wlo) continue;
                // row constants as initial accumulators (log2 p, and dP - delta): the first MFMA of each chain reads them
                // from two vectors that stay resident for the whole key sweep (the MFMA's C and D may differ), so no
                // per-tile register fills are needed
                f32x16 sacc = mfma32(frag_R(st, 32 * blk, 0, lane), qf[0], lse_init);
                f32x16 dp = mfma32(frag_R(st + 8192, 32 * blk, 0, lane), dof[0], dlt_init);
#pragma unroll
                for (int s = 1; s < 4; ++s) {
                    sacc = mfma32(frag_R(st, 32 * blk, s, lane), qf[s], sacc);
                    dp = mfma32(frag_R(st + 8192, 32 * blk, s, lane), dof[s], dp);
                }
                if (!unmasked) mask_scores_keyrows_d<MASK>(sacc, mb, blk, key0, lo, hi, hh);
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float p = __builtin_amdgcn_exp2f(PRE ? sacc[i] : sacc[i] * c);
                    sacc[i] = p * dp[i];  // dS^T / scale (the scale is applied once, to dQ)
                }
#pragma unroll
                for (int sp = 0; sp < 2; ++sp) {
                    const bf16x8 dsf = acc_to_frag(sacc, sp);
                    dq[0] = mfma32(frag_T(st, 32 * blk + 16 * sp, 0, lane), dsf, dq[0]);  // K^T from the same image as the row reads
                    dq[1] = mfma32(frag_T(st, 32 * blk + 16 * sp, 1, lane), dsf, dq[1]);
                }
            }
        }
        BAND_T(3 + 2 * (t - t_lo));
    }

    if (rope_cos) {  // backward of apply_rotary_pos_emb (the transposed rotation): dims d / d+32 are the lane's two accumulator blocks
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float a0 = dq[0][4 * g + r], b0 = dq[1][4 * g + r];
                dq[0][4 * g + r] = rot_fma(a0, rcs[g][r], b0, rsn[g][r]);
                dq[1][4 * g + r] = rot_fma(b0, rcs[g][r], -a0, rsn[g][r]);
            }
    }
    lds_only_barrier();  // every wave is done with the ring: its slots become the waves' transposition buffers
    BAND_T(14);
    store_rows32(smem + 4608 * wid, dq[0], dq[1], scale, dqkv + (sv.row0 + q0) * ld + head * 64, ld, S - q0, lane);
    BAND_T(15);
}

template <bool PRE, bool MASK>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ d_o,
                                                             const uint16_t* __restrict__ o_rows, const float* __restrict__ lse,
                                                             float* __restrict__ delta,
                                                             uint16_t* __restrict__ dqkv, const uint8_t* __restrict__ kmask,
                                                             int Smax, int nh, int window, float scale,
                                                             const float* __restrict__ rope_cos,
                                                             const float* __restrict__ rope_sin, int64_t pos_batch_stride, VarLen vl) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int qblk, head, b;
    if (window >= 0) decode_block_band((Smax + 127) / 128, nh, qblk, head, b);
    else decode_block((Smax + 127) / 128, nh, qblk, head, b);
    band_dq_block<PRE, MASK>(smem, qblk, head, b, qkv, d_o, o_rows, lse, delta, dqkv, kmask, Smax, nh, window, scale, rope_cos, rope_sin,
                                    pos_batch_stride, vl);
}

// ---------------------------------------------------------------------------------------------------------------
// dK, dV: one workgroup = 4 waves = 128 keys of one (batch, head); each wave owns 32 keys (key on the lane) and keeps
// dK^T, dV^T (64 x 32 each) in accumulators while the workgroup sweeps query tiles of 64 rows.
// LDS per ring slot: Q image + dO image (8 KiB each, row and transposed reads) + the rows' lse and delta (raw, 64 floats each:
// the accumulator init multiplies / negates them); tiles arrive by LDS-DMA, dK / dV leave as whole rows (store_rows32).
// ---------------------------------------------------------------------------------------------------------------
constexpr int kDkvStage = 2 * 8192 + 512;  // Q image, dO image, the rows' lse and delta (raw, 64 floats each)
constexpr int kDkvSlots = 4;               // LDS-DMA ring, as in the dq kernel

template <bool PRE>
__device__ __forceinline__ void band_dkv_block(char* smem, int kblk, int head, int b, const uint16_t* __restrict__ qkv,
                                               const uint16_t* __restrict__ d_o, const float* __restrict__ lse,
                                               const float* __restrict__ delta, uint16_t* __restrict__ dqkv,
                                               const uint8_t* __restrict__ kmask, int Smax, int nh, int window, float scale,
                                               const float* __restrict__ rope_cos, const float* __restrict__ rope_sin,
                                               int64_t pos_batch_stride, VarLen vl) {
    const int tid = threadIdx.x, lane = tid & 63, hh = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int K0 = kblk * 128;
    const SeqView sv(vl, b, head, Smax, nh);
    const int S = sv.S;
    if (K0 >= S) return;
    const int k0 = K0 + wid * 32;
    const int64_t ld = (int64_t)3 * nh * 64;
    const uint16_t* qbase = qkv + sv.row0 * ld + head * 64;
    const uint16_t* kbase = qbase + nh * 64;
    const uint16_t* vbase = qbase + 2 * nh * 64;
    const int64_t ldo = (int64_t)nh * 64;
    const uint16_t* dobase = d_o + sv.row0 * ldo + head * 64;
    const float* lse_bh = lse + sv.stat0;
    const float* dlt_bh = delta + sv.stat0;

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

    // Q / dO tiles and the rows' lse / delta by LDS-DMA into a four-slot ring (see attn_bwd_dq_kernel): four operations per wave
    // and tile, six in wave 0, which also brings the two statistics rows (and, in the sequence's last tile, overwrites the rows
    // past the end - read from row S - 1 - with lse = +inf, delta = 0 before the barrier publishes them: p = 0 there)
    const int ND = wid == 0 ? 6 : 4;
    const TileDma dma(wid, lane);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const uint32_t m0_q = __builtin_amdgcn_readfirstlane(lds0 + 2048u * wid);
    auto dma_tile = [&](int t) {
        const uint32_t slot = (uint32_t)((t - t_lo) & (kDkvSlots - 1)) * kDkvStage;
        dma.rows(m0_q + slot, qbase, (int)ld * 2, t * 64, S);
        dma.rows(m0_q + slot + 8192u, dobase, (int)ldo * 2, t * 64, S, CM3P_AUD_T1);
        if (wid == 0) {
            const uint32_t so = (uint32_t)(min(t * 64 + lane, S - 1) * 4);
            dma_dword64(lds0 + slot + 16384u, lse_bh, so);
            dma_dword64(lds0 + slot + 16384u + 256u, dlt_bh, so, CM3P_AUD_S1);
        }
    };
    for (int t = t_lo; t <= min(t_hi, (CM3P_BABL & 4) ? t_lo - 1 : t_lo + 2); ++t) dma_tile(t);
    // (the tiles are on their way before the wave's own K / V fragments are requested: the two round trips overlap)
    const int krow = k0 + (lane & 31);
    const int krow_c = krow < S ? krow : S - 1;
    const float c = scale * kLog2e;
    bf16x8 kf[4], vf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        kf[s] = *reinterpret_cast<const bf16x8*>(kbase + (int64_t)krow_c * ld + 16 * s + 8 * hh);
        vf[s] = *reinterpret_cast<const bf16x8*>(vbase + (int64_t)krow_c * ld + 16 * s + 8 * hh);
    }
    const bool key_ok = krow < S && (kmask ? kmask[sv.row0 + krow] != 0 : true);
    const bool keys_all_ok = __all(key_ok);
    const int lo = window < 0 ? INT_MIN : krow - window, hi = window < 0 ? INT_MAX : krow + window;

    // the rotary rows of the epilogue are requested now: their round trip hides behind the query sweep
    f32x4 rcs[4], rsn[4];
    if (rope_cos) {
        const int64_t prow = sv.pos0(b, pos_batch_stride) + krow_c;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            rcs[g] = *reinterpret_cast<const f32x4*>(rope_cos + prow * 32 + 8 * g + 4 * hh);
            rsn[g] = *reinterpret_cast<const f32x4*>(rope_sin + prow * 32 + 8 * g + 4 * hh);
        }
    }
    f32x16 dk[2], dv[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) dk[0][i] = dk[1][i] = dv[0][i] = dv[1][i] = 0.f;

    // the score accumulators start at -lse * log2(e) (PRE) or -lse / scale, dP's at -delta; rows past the sequence at -inf (p = 0)
    const float lse_mul = PRE ? -kLog2e : -1.0f / scale;

    for (int t = t_lo; t <= t_hi; ++t) {
        char* st = smem + ((t - t_lo) & (kDkvSlots - 1)) * kDkvStage;
        const int qt0 = t * 64;
        dma_wait(ND * min(t_hi - t, 2));  // this wave's part of tile t has landed
        if (wid == 0 && qt0 + lane >= S) {
            reinterpret_cast<float*>(st + 16384)[lane] = __builtin_huge_valf();
            reinterpret_cast<float*>(st + 16384 + 256)[lane] = 0.f;
        }
        lds_only_barrier();  // ... everyone's has; the slot of tile t-1 is free
        if (!(CM3P_BABL & 4) && t + 3 <= t_hi) dma_tile(t + 3);
        if (!(CM3P_BABL & 1) && wave_live && qt0 <= whi && qt0 + 63 >= wlo) {
            const float* nlse = reinterpret_cast<const float*>(st + 16384);
            const float* ndlt = nlse + 64;
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {  // two 32-query blocks of the tile
                // (a 64-row tile that touches the wave's band may still hold a 32-row block that lies outside it: a quarter of the
                //  blocks of a +-64 window)
                if (qt0 + 32 * qb > whi || qt0 + 32 * qb + 31 < wlo) continue;
                f32x16 sacc, dp;
                // initial accumulators: row constants -lse*log2(e) and -delta (rows = queries)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(nlse + 32 * qb + 8 * g + 4 * hh);
                    const f32x4 d = *reinterpret_cast<const f32x4*>(ndlt + 32 * qb + 8 * g + 4 * hh);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        sacc[4 * g + r] = a[r] * lse_mul;
                        dp[4 * g + r] = -d[r];
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
                        const float p = __builtin_amdgcn_exp2f(PRE ? sacc[i] : sacc[i] * c);
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
                            const float p = ok ? __builtin_amdgcn_exp2f(PRE ? sacc[i] : sacc[i] * c) : 0.f;
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
    }

    if (rope_cos) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float a0 = dk[0][4 * g + r], b0 = dk[1][4 * g + r];
                dk[0][4 * g + r] = rot_fma(a0, rcs[g][r], b0, rsn[g][r]);
                dk[1][4 * g + r] = rot_fma(b0, rcs[g][r], -a0, rsn[g][r]);
            }
    }
    lds_only_barrier();  // every wave is done with the ring: its slots become the waves' transposition buffers
    if (!(CM3P_BABL & 2)) {
        // PRE: the products were taken with q * scale * log2(e), so dK = ln(2) * accumulator
        uint16_t* dk0 = dqkv + (sv.row0 + k0) * ld + nh * 64 + head * 64;
        store_rows32(smem + 4608 * wid, dk[0], dk[1], PRE ? 0.69314718055994531f : scale, dk0, ld, S - k0, lane);
        store_rows32(smem + 4608 * wid, dv[0], dv[1], 1.0f, dk0 + nh * 64, ld, S - k0, lane);
    }
}

template <bool PRE>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ d_o,
                                                              const float* __restrict__ lse, const float* __restrict__ delta,
                                                              uint16_t* __restrict__ dqkv, const uint8_t* __restrict__ kmask,
                                                              int Smax, int nh, int window, float scale,
                                                              const float* __restrict__ rope_cos,
                                                              const float* __restrict__ rope_sin, int64_t pos_batch_stride, VarLen vl) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int kblk, head, b;
    if (window >= 0) decode_block_band((Smax + 127) / 128, nh, kblk, head, b);
    else decode_block((Smax + 127) / 128, nh, kblk, head, b);
    band_dkv_block<PRE>(smem, kblk, head, b, qkv, d_o, lse, delta, dqkv, kmask, Smax, nh, window, scale, rope_cos, rope_sin, pos_batch_stride, vl);
}


// ---------------------------------------------------------------------------------------------------------------
// output_attentions: the probabilities themselves, [B, nh, S, S] fp32 - what the reference returns when a caller asks for
// attention weights (TF then runs eager_attention_forward, TF:models/modernbert/modeling_modernbert.py:133-170: softmax of
// scale * q k^T + additive mask in fp32).  An inspection path, not a hot one: p[q, k] = exp(scale * q.k - lse[q]) from the lse the
// forward stored, plain fp32 FMAs, 64 x 64 tiles.  Invisible keys give exact zeros; a row with no visible key is uniform
// 1 / S, as the eager path's finite additive mask (finfo.min on every key) leaves it.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_probs_kernel(const uint16_t* __restrict__ qkv, const float* __restrict__ lse,
                                                         const uint8_t* __restrict__ kmask, float* __restrict__ probs, int S, int nh,
                                                         int window, float q_mul) {
    __shared__ float ks[64][65];
    const int qb = blockIdx.x, head = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
    const int qrow = qb * 64 + (tid >> 2), part = tid & 3;
    const int64_t ld = (int64_t)3 * nh * 64;
    const uint16_t* base = qkv + (int64_t)b * S * ld + head * 64;
    float q[64];
    const int qc = min(qrow, S - 1);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(base + (int64_t)qc * ld + 8 * c);
#pragma unroll
        for (int j = 0; j < 8; ++j) q[8 * c + j] = (float)v[j] * q_mul;  // q_mul: log2(e) * scale for plain q, 1 when q is prescaled
    }
    const float l = lse[((int64_t)b * nh + head) * S + qc];
    const bool dead = l == __builtin_huge_valf();
    const float l2 = l * kLog2e;
    float* prow = probs + (((int64_t)b * nh + head) * S + qc) * S;
    for (int k0 = 0; k0 < S; k0 += 64) {
        __syncthreads();
        for (int i = tid; i < 64 * 8; i += 256) {
            const int r = i >> 3, c = i & 7, kr = min(k0 + r, S - 1);
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(base + nh * 64 + (int64_t)kr * ld + 8 * c);
#pragma unroll
            for (int j = 0; j < 8; ++j) ks[r][8 * c + j] = (float)v[j];
        }
        __syncthreads();
        if (qrow < S) {
#pragma unroll 4
            for (int kk = 0; kk < 16; ++kk) {
                const int kl = part * 16 + kk, key = k0 + kl;
                if (key >= S) break;
                float acc = 0.f;
#pragma unroll
                for (int d = 0; d < 64; ++d) acc = __builtin_fmaf(q[d], ks[kl][d], acc);
                const bool vis = (kmask == nullptr || kmask[(int64_t)b * S + key] != 0) && (window < 0 || abs(qrow - key) <= window);
                prow[key] = dead ? 1.0f / (float)S : (vis ? __builtin_amdgcn_exp2f(acc - l2) : 0.f);
            }
        }
    }
}

}  // namespace

// attention_fwd.hip: the global layers' forward as one software-pipelined stream per wave (r05; pre-scaled q only)
int cm3p_launch_attn_fwd_global(const void* qkv, void* out, float* lse, const uint8_t* key_mask, int B, int S, int nh, const int* cu_seqlens,
                                int64_t total, hipStream_t s);
// CM3P_ATTN_FWD_IMPL=wave3 keeps the global layers on attn_fwd_kernel (three compiler-scheduled waves per SIMD): the A/B partner and
// the independent implementation the pipelined kernel is cross-checked against in the tests
static bool fwd_pipelined() {  // (read per call: the tests run both in one process)
    const char* e = getenv("CM3P_ATTN_FWD_IMPL");
    return !(e && e[0] == 'w');
}

// the routing rule, also exported (cm3p_attn_fwd_impl): (TileDma::rows: 32-bit row * pitch source offsets, attn_common.h)
static bool fwd_takes_pipelined(int S, int nh, int window, int pre) {
    return window < 0 && pre && fwd_pipelined() && (int64_t)S * 3 * nh * 128 < (int64_t(1) << 31);
}

static int launch_attn_fwd(const void* qkv, void* out, float* lse, const uint8_t* key_mask, int B, int S, int nh, int window,
                           float scale, VarLen vl, int pre, hipStream_t s) {
    if (fwd_takes_pipelined(S, nh, window, pre))
        return cm3p_launch_attn_fwd_global(qkv, out, lse, key_mask, B, S, nh, vl.cu, vl.total, s);
    // QSUB = 1 (32 queries per wave, 3 waves per SIMD) measured faster than QSUB = 2 (64 per wave, compiler-scheduled):
    // 2.29 ms vs 3.49 ms per C2 global layer.  The two-chain variant needs a hand-placed MFMA/VALU interleave to pay.
    const dim3 grid(((S + 127) / 128) * nh * B);  // 1-D: decode_block() maps it XCD-aware
#define CM3P_FWD_ARGS (const uint16_t*)qkv, (uint16_t*)out, lse, key_mask, S, nh, window, scale, vl
    if (window >= 0) {
        if (pre) attn_fwd_kernel<1, true, true><<<grid, 256, 2 * kFwdStage, s>>>(CM3P_FWD_ARGS);
        else attn_fwd_kernel<1, false, true><<<grid, 256, 2 * kFwdStage, s>>>(CM3P_FWD_ARGS);
    } else {
        if (pre) attn_fwd_kernel<1, true, false><<<grid, 256, 2 * kFwdStage, s>>>(CM3P_FWD_ARGS);
        else attn_fwd_kernel<1, false, false><<<grid, 256, 2 * kFwdStage, s>>>(CM3P_FWD_ARGS);
    }
#undef CM3P_FWD_ARGS
    return CM3P_OK;
}

// attention_bwd.hip: the global-layer (window < 0) backward kernels
int cm3p_launch_attn_bwd_global(const void* qkv, const void* out, const void* dout, const float* lse, float* delta, void* dqkv,
                                const uint8_t* key_mask, int B, int S, int nh, float scale, const float* cos_tab, const float* sin_tab,
                                int64_t pos_batch_stride, const int* cu_seqlens, int64_t total, int stages, int pre, hipStream_t s);

// Kernel experiments only: CM3P_ATTN_BAND_EVERYWHERE=1 routes global layers through the band kernels of this file as well.
static bool band_kernels_everywhere() {
    static const bool v = [] { const char* e = getenv("CM3P_ATTN_BAND_EVERYWHERE"); return e && e[0] == '1'; }();
    return v;
}

// stages: CM3P_ATTN_BWD_DQ (dq and delta) | CM3P_ATTN_BWD_DKV (dk, dv; reads the delta the dq stage wrote)
static int launch_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse, float* delta, void* dqkv,
                           const uint8_t* key_mask, int B, int S, int nh, int window, float scale, const float* cos_tab,
                           const float* sin_tab, int64_t pos_batch_stride, VarLen vl, int stages, int pre, hipStream_t s) {
    if (window < 0 && !band_kernels_everywhere())
        return cm3p_launch_attn_bwd_global(qkv, out, dout, lse, delta, dqkv, key_mask, B, S, nh, scale, cos_tab, sin_tab, pos_batch_stride,
                                           vl.cu, vl.total, stages, pre, s);
    const dim3 grid(((S + 127) / 128) * nh * B);  // 1-D: decode_block() maps it XCD-aware
    if (stages & CM3P_ATTN_BWD_DQ) {
#define CM3P_DQ_ARGS (const uint16_t*)qkv, (const uint16_t*)dout, (const uint16_t*)out, lse, delta, (uint16_t*)dqkv, key_mask, S, nh, window, scale, cos_tab, sin_tab, pos_batch_stride, vl
        static Cm3pDevOnce once;  // (per device: common.h)
        const int rc_once = once.run([] {
            return cm3p_set_max_lds({reinterpret_cast<const void*>(&attn_bwd_dq_kernel<true, true>), reinterpret_cast<const void*>(&attn_bwd_dq_kernel<true, false>),
                                     reinterpret_cast<const void*>(&attn_bwd_dq_kernel<false, true>), reinterpret_cast<const void*>(&attn_bwd_dq_kernel<false, false>)},
                                    kDqSlots * kDqStage);
        });
        if (rc_once != CM3P_OK) return rc_once;
        if (pre && key_mask) attn_bwd_dq_kernel<true, true><<<grid, 256, kDqSlots * kDqStage, s>>>(CM3P_DQ_ARGS);
        else if (pre) attn_bwd_dq_kernel<true, false><<<grid, 256, kDqSlots * kDqStage, s>>>(CM3P_DQ_ARGS);
        else if (key_mask) attn_bwd_dq_kernel<false, true><<<grid, 256, kDqSlots * kDqStage, s>>>(CM3P_DQ_ARGS);
        else attn_bwd_dq_kernel<false, false><<<grid, 256, kDqSlots * kDqStage, s>>>(CM3P_DQ_ARGS);
#undef CM3P_DQ_ARGS
        if (hipGetLastError() != hipSuccess) return CM3P_ERR_LAUNCH;
    }
    if (stages & CM3P_ATTN_BWD_DKV) {
#define CM3P_DKV_ARGS (const uint16_t*)qkv, (const uint16_t*)dout, lse, delta, (uint16_t*)dqkv, key_mask, S, nh, window, scale, cos_tab, sin_tab, pos_batch_stride, vl
        static Cm3pDevOnce once2;
        const int rc_once = once2.run([] {
            return cm3p_set_max_lds({reinterpret_cast<const void*>(&attn_bwd_dkv_kernel<true>), reinterpret_cast<const void*>(&attn_bwd_dkv_kernel<false>)},
                                    kDkvSlots * kDkvStage);
        });
        if (rc_once != CM3P_OK) return rc_once;
        if (pre) attn_bwd_dkv_kernel<true><<<grid, 256, kDkvSlots * kDkvStage, s>>>(CM3P_DKV_ARGS);
        else attn_bwd_dkv_kernel<false><<<grid, 256, kDkvSlots * kDkvStage, s>>>(CM3P_DKV_ARGS);
#undef CM3P_DKV_ARGS
    }
    return CM3P_OK;
}

extern "C" {

int cm3p_attn_fwd(const void* qkv, void* out, float* lse, const uint8_t* key_mask, int B, int S, int nh, int window,
                  float scale, int q_prescaled, void* stream) {
    CM3P_REQUIRE(qkv && out && lse && B > 0 && S > 0 && nh > 0 && scale > 0.f);
    CM3P_REQUIRE(cm3p_aligned16(qkv) && cm3p_aligned16(out));
    const int rc = launch_attn_fwd(qkv, out, lse, key_mask, B, S, nh, window, scale, VarLen{nullptr, 0}, q_prescaled != 0,
                                   static_cast<hipStream_t>(stream));
    if (rc != CM3P_OK) return rc;
    CM3P_LAUNCH_CHECK();
    return CM3P_OK;
}

int cm3p_attn_fwd_impl(int S, int nh, int window, int q_prescaled) { return fwd_takes_pipelined(S, nh, window, q_prescaled != 0) ? 1 : 0; }

int cm3p_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse, float* delta, void* dqkv,
                  const uint8_t* key_mask, int B, int S, int nh, int window, float scale, const float* cos_tab,
                  const float* sin_tab, int64_t pos_batch_stride, int stages, int q_prescaled, void* stream) {
    CM3P_REQUIRE((cos_tab == nullptr) == (sin_tab == nullptr));
    CM3P_REQUIRE(stages >= 1 && stages <= 3);
    CM3P_REQUIRE(pos_batch_stride == 0 || pos_batch_stride == S);
    CM3P_REQUIRE(qkv && out && dout && lse && delta && dqkv && B > 0 && S > 0 && nh > 0 && scale > 0.f);
    CM3P_REQUIRE(cm3p_aligned16(qkv) && cm3p_aligned16(out) && cm3p_aligned16(dout) && cm3p_aligned16(dqkv));
    CM3P_REQUIRE((int64_t)S * 3 * nh * 128 < (int64_t(1) << 31));  // TileDma::rows: 32-bit row * pitch source offsets (attn_common.h)
    const int rc = launch_attn_bwd(qkv, out, dout, lse, delta, dqkv, key_mask, B, S, nh, window, scale, cos_tab, sin_tab,
                                   pos_batch_stride, VarLen{nullptr, 0}, stages, q_prescaled != 0, static_cast<hipStream_t>(stream));
    if (rc != CM3P_OK) return rc;
    CM3P_LAUNCH_CHECK();
    return CM3P_OK;
}

int cm3p_attn_fwd_varlen(const void* qkv, void* out, float* lse, const int* cu_seqlens, int B, int max_seqlen, int64_t total,
                         int nh, int window, float scale, int q_prescaled, void* stream) {
    CM3P_REQUIRE(qkv && out && lse && cu_seqlens && B > 0 && max_seqlen > 0 && total > 0 && nh > 0 && scale > 0.f);
    CM3P_REQUIRE(cm3p_aligned16(qkv) && cm3p_aligned16(out));
    const int rc = launch_attn_fwd(qkv, out, lse, nullptr, B, max_seqlen, nh, window, scale, VarLen{cu_seqlens, total}, q_prescaled != 0,
                                   static_cast<hipStream_t>(stream));
    if (rc != CM3P_OK) return rc;
    CM3P_LAUNCH_CHECK();
    return CM3P_OK;
}

int cm3p_attn_bwd_varlen(const void* qkv, const void* out, const void* dout, const float* lse, float* delta, void* dqkv,
                         const int* cu_seqlens, int B, int max_seqlen, int64_t total, int nh, int window, float scale,
                         const float* cos_tab, const float* sin_tab, int stages, int q_prescaled, void* stream) {
    CM3P_REQUIRE((cos_tab == nullptr) == (sin_tab == nullptr));
    CM3P_REQUIRE(stages >= 1 && stages <= 3);
    CM3P_REQUIRE(qkv && out && dout && lse && delta && dqkv && cu_seqlens && B > 0 && max_seqlen > 0 && total > 0 && nh > 0 && scale > 0.f);
    CM3P_REQUIRE(cm3p_aligned16(qkv) && cm3p_aligned16(out) && cm3p_aligned16(dout) && cm3p_aligned16(dqkv));
    CM3P_REQUIRE((int64_t)max_seqlen * 3 * nh * 128 < (int64_t(1) << 31));  // TileDma::rows: 32-bit row * pitch source offsets
    const int rc = launch_attn_bwd(qkv, out, dout, lse, delta, dqkv, nullptr, B, max_seqlen, nh, window, scale, cos_tab, sin_tab, 0,
                                   VarLen{cu_seqlens, total}, stages, q_prescaled != 0, static_cast<hipStream_t>(stream));
    if (rc != CM3P_OK) return rc;
    CM3P_LAUNCH_CHECK();
    return CM3P_OK;
}

int cm3p_attn_probs(const void* qkv, const float* lse, const uint8_t* key_mask, float* probs, int B, int S, int nh, int window, float scale,
                    int q_prescaled, void* stream) {
    CM3P_REQUIRE(qkv && lse && probs && B > 0 && S > 0 && nh > 0 && scale > 0.f && cm3p_aligned16(qkv));
    attn_probs_kernel<<<dim3((S + 63) / 64, nh, B), 256, 0, static_cast<hipStream_t>(stream)>>>(
        static_cast<const uint16_t*>(qkv), lse, key_mask, probs, S, nh, window, q_prescaled ? 1.0f : scale * kLog2e);
    CM3P_LAUNCH_CHECK();
    return CM3P_OK;
}

}  // extern "C"

// timing-only ablation switches this object was built with (0 in every shipped build: cm3p_build_ablation_flags, tests/test_cabi.py)
int cm3p_ablation_flags_attention() { return (CM3P_BABL); }
#if CM3P_BAND_TRACE
extern "C" int cm3p_debug_set_band_trace(void* buf) {
    unsigned long long* p = static_cast<unsigned long long*>(buf);
    return hipMemcpyToSymbol(HIP_SYMBOL(g_band_trace), &p, sizeof(p)) == hipSuccess ? 0 : 1;
}
#endif
#if CM3P_DMA_AUDIT
int cm3p_audit_set_attention(void* buf) { return cm3p_audit_set_local(buf); }
#endif
