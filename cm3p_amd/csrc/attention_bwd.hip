// Backward of the GLOBAL attention layers (window < 0), head_dim 64, gfx950: the kernels of cm3p_attn_bwd / cm3p_attn_bwd_varlen
// when no sliding window is set (the sliding-window layers keep the band kernels of attention.hip).
//
// Replaces the backward of F.scaled_dot_product_attention (TF:integrations/sdpa_attention.py:153-163) under the key-padding
// mask of TF:masking_utils.py:168-179; the inverse of apply_rotary_pos_emb (TF:models/modernbert/modeling_modernbert.py:188-219)
// is applied to dq / dk in the epilogues.
//
// Structure (both kernels, details at each): ONE wave per SIMD with the whole 512-entry register file, 64 rows of the stationary
// operand per wave (256 per 4-wave workgroup), 64-row tiles of the streamed operands staged global -> registers -> LDS into a
// four-slot ring (two staging register sets: a tile's loads fly for two tile periods; one workgroup barrier per tile; no load
// is consumed before its LDS store; loads are branch-free with clamped rows), and ONE hand-placed instruction stream per wave
// (sched_barrier between chunks) that software-pipelines the tile's four 32 x 32 score blocks: score MFMAs of block n+1 next to
// the exponentials of block n, then block n's gradient MFMAs next to its products / bf16 packs.  Score MFMAs are inline asm with
// VGPR results (the VALU reads them) and AGPR-resident stationary operands; gradient MFMAs are compiler MFMAs with AGPR
// accumulators.  LDS fragments are register-resident per 32-row block and reloaded a full step before their next use.
//
//   dkv  key on the lane: S = Q K^T and dP = dO V^T with the score offset (-lse) and -delta preloaded as initial accumulators;
//        dV^T += dO^T P and dK^T += Q^T dS take the score accumulators directly as B operands.  Nothing is masked in the loop: a
//        padded key only pollutes its own column of dK^T / dV^T, which the epilogue writes as zeros.
//   dq   query on the lane: S^T = K Q^T, dP^T = V dO^T - delta, dS^T = P^T o dP^T, dQ^T += K^T dS^T (K^T by transposed reads of
//        the same LDS image); also computes delta = rowsum(dO o O) for its rows and publishes it for the dkv kernel.  Invisible
//        keys are staged with a zero K row, so nothing is masked in the loop.
//
// Softmax scale: q_prescaled (cm3p_hip.h) - either q already carries scale * log2(e) (folded in before q's single bf16 rounding
// by the Wqkv GEMM's epilogue) or the kernels multiply the fp32 score accumulators; q is never re-rounded to bf16.
// Measured (C2 global layer, B 32 x 12 heads x 4096^2): dkv 3.96 -> 2.92 ms, dq 2.84 -> 2.48 ms against the 32-row-per-wave,
// two-waves-per-SIMD kernels they replace (attention.hip, which still serves the sliding-window layers).
#include <stdlib.h>

#include <type_traits>

#include "attn_common.h"

namespace {

constexpr int kBwdStage = 2 * 8192 + 512;  // two 64 x 64 bf16 images + 128 floats (dkv: the score offsets and -delta of the 64 rows)

__device__ __forceinline__ bf16x8 gload_frag(const uint16_t* p) { return *reinterpret_cast<const bf16x8*>(p); }

// -----------------------------------------------------------------------------------------------------------------------------
// dK, dV, hand-scheduled: ONE wave per SIMD with the whole 512-entry register file, wave w owns keys K0 + 64 w .. + 63 (two
// 32-key column blocks kb), workgroup = 4 waves = 256 keys.
//
// With the 512-entry budget hipcc selects the AGPR form of every MFMA it generates, and the VALU cannot read AGPRs.  So the
// products whose results the VALU consumes (S and dP) are issued as inline-asm MFMAs with VGPR destinations (their B operands,
// the wave's K / V fragments, live in AGPRs: 64 registers the VGPR file does not have to hold), while the accumulating products
// (dV^T, dK^T: 128 AGPRs) stay compiler MFMAs.  An asm MFMA's result is first read one pipeline step (hundreds of cycles) after
// it was issued, behind a sched_barrier, which covers the MFMA -> VALU hazard the compiler does not pad for an asm statement.
//
// Schedule: the tile's four 32 x 32 blocks u = (qb, kb) are software pipelined in ONE instruction stream, every instruction
// placed by hand at chunk granularity (sched_barrier(0) between chunks):
//     step k:  [8 score MFMAs of block k+1, each followed by two (mul, exp) pairs of block k]      <- 16 exponentials
//              [8 gradient MFMAs of block k, each preceded by the products / bf16 packs it needs]
// LDS fragments are register-resident per 32-query block and shared by its two key blocks; a fragment register is reloaded for
// the NEXT query block right behind the last MFMA that read it, a full step before its next use, so no MFMA waits for LDS.
// Tiles are staged global -> registers -> LDS into a four-slot ring (one barrier per tile); the loop is unrolled over the ring
// so every LDS address is a per-lane base plus an immediate.
// -----------------------------------------------------------------------------------------------------------------------------
constexpr int kSlots3 = 4;

#ifndef CM3P_ABL
#define CM3P_ABL 0  // timing-only ablation builds (tools/ubench/attn_bwd_ablate.sh): 1 no barrier, 2 no tile staging, 4 no fragment reloads, 8 no exponentials
#endif
#if CM3P_ABL & 16  // timing only: every score MFMA starts from the row constants (no accumulate chain)
#define SC_ACC(d, a, b, c) mfma_vc(d, a, b, c)
#else
#define SC_ACC(d, a, b, c) mfma_va(d, a, b)
#endif

// PRE: scores arrive in exp2 units (q carries scale * log2 e); !PRE: one fp32 multiply per score
template <bool PRE>
__device__ __forceinline__ void exp2_pair(f32x16& s, int i, float cm) {
    if constexpr ((CM3P_ABL & 8) != 0) return;
    s[i] = __builtin_amdgcn_exp2f(PRE ? s[i] : s[i] * cm);
    s[i + 1] = __builtin_amdgcn_exp2f(PRE ? s[i + 1] : s[i + 1] * cm);
}

template <bool PRE>
__global__ __launch_bounds__(256, 1) void attn_bwd_dkv3_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ d_o,
                                                               const float* __restrict__ lse, const float* __restrict__ delta,
                                                               uint16_t* __restrict__ dqkv, const uint8_t* __restrict__ kmask,
                                                               int Smax, int nh, float scale, const float* __restrict__ rope_cos,
                                                               const float* __restrict__ rope_sin, int64_t pos_batch_stride, VarLen vl) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, hh = lane >> 5;
    int kblk, head, b;
    decode_block((Smax + 255) / 256, nh, kblk, head, b);
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
    const float* lse_bh = lse + sv.stat0;
    const float* dlt_bh = delta + sv.stat0;
    const float cm = scale * kLog2e;
    // the score accumulators start at -lse * log2(e) (PRE: q carries scale * log2 e, the MFMA delivers log2 p) or at -lse / scale
    // (!PRE: cm * (q.k - lse / scale) = log2 p)
    const float lse_mul = PRE ? -kLog2e : -1.0f / scale;

    bf16x8 kf[2][4], vf[2][4];  // B operands of the asm MFMAs ("a" constraint: they live in AGPRs)
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        const int krow_c = min(k0 + 32 * kb + (lane & 31), S - 1);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            kf[kb][s] = gload_frag(kbase + (int64_t)krow_c * ld + 16 * s + 8 * hh);
            vf[kb][s] = gload_frag(vbase + (int64_t)krow_c * ld + 16 * s + 8 * hh);
        }
    }
    f32x16 dk[2][2], dv[2][2];  // [d block][key block]
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 16; ++i) dk[db][kb][i] = dv[db][kb][i] = 0.f;

    // per-lane LDS byte offsets inside a slot; everything else is an immediate
    const int l31 = lane & 31, g4 = lane >> 4, i16 = lane & 15;
    int oR[4], oTlo[2], oThi[2];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) oR[s4] = off_R(l31, 2 * s4 + hh);
#pragma unroll
    for (int db = 0; db < 2; ++db) {
        const int row = 4 * hh + (i16 >> 2), col = 32 * db + 16 * (g4 & 1) + 4 * (i16 & 3);
        oTlo[db] = off_T(row, col);
        oThi[db] = off_T(row + 8, col);
    }
    const int oI = 16384 + 16 * hh;

    const int n_tiles = (S + 63) / 64;
    // Staging registers: TWO sets (tiles of even / odd index), so a tile's global loads are in flight for two tile periods before
    // their first use in the LDS stores.  Plain named scalars on purpose: a struct or an indexed array ends up in scratch.
    uint4 aq0, aq1, ag0, ag1, bq0, bq1, bg0, bg1;
    float araw = 0.f, braw = 0.f;  // raw lse (threads with tid & 64 == 0) or delta of row t * 64 + (tid & 63)
    bool aok = false, bok = false;
    const int srow = tid >> 3, schunk = (tid & 7) * 8;
    const float* stat_src = (tid & 64) ? dlt_bh : lse_bh;
    const int oW0 = off_R(srow, tid & 7), oW1 = off_R(srow + 32, tid & 7);
    // Branch-free, and NOTHING in a load block consumes a loaded value: the first use of every staging register is in the store
    // block (an arithmetic instruction on a fresh load makes the compiler wait for the whole memory round trip right there,
    // every tile: measured 30 % of this kernel).  Tiles past the end reload the last rows with a -inf score offset (p = 0).
#define CM3P_GLOAD(P, t_)                                                                          \
    do {                                                                                            \
        const int t__ = (t_);                                                                       \
        const int r0 = min(t__ * 64 + srow, S - 1), r1 = min(t__ * 64 + 32 + srow, S - 1);           \
        P##q0 = *reinterpret_cast<const uint4*>(qbase + (int64_t)r0 * ld + schunk);                 \
        P##q1 = *reinterpret_cast<const uint4*>(qbase + (int64_t)r1 * ld + schunk);                 \
        P##g0 = *reinterpret_cast<const uint4*>(dobase + (int64_t)r0 * ldo + schunk);               \
        P##g1 = *reinterpret_cast<const uint4*>(dobase + (int64_t)r1 * ldo + schunk);               \
        const int q__ = t__ * 64 + (tid & 63);                                                      \
        P##raw = stat_src[min(q__, S - 1)];                                                         \
        P##ok = q__ < S;                                                                            \
    } while (0)
    // -lse / scale (rows past S, and rows whose lse is +inf because they see no key, start at -inf: p = 0) and -delta
#define CM3P_LSTORE(P, st_)                                                                                          \
    do {                                                                                                              \
        char* st__ = (st_);                                                                                           \
        *reinterpret_cast<uint4*>(st__ + oW0) = P##q0;                                                                \
        *reinterpret_cast<uint4*>(st__ + oW1) = P##q1;                                                                \
        *reinterpret_cast<uint4*>(st__ + 8192 + oW0) = P##g0;                                                         \
        *reinterpret_cast<uint4*>(st__ + 8192 + oW1) = P##g1;                                                         \
        const float sreg__ = (tid & 64) ? (P##ok ? -P##raw : 0.f) : (P##ok ? P##raw * lse_mul : kNegInf);             \
        reinterpret_cast<float*>(st__ + 16384)[tid & 127] = sreg__; /* (threads 128..255 repeat the same values) */   \
    } while (0)

    // The same two blocks cut into five pieces each, so that a tile's staging rides in the gaps of the MFMA stream (one piece per
    // chunk of a step) instead of sitting in front of it: piece I of the stores / of the loads.
#define CM3P_LSTORE_PIECE(P, st_, I)                                                                                     \
    do {                                                                                                                  \
        char* st__ = (st_);                                                                                               \
        if constexpr ((I) == 0) *reinterpret_cast<uint4*>(st__ + oW0) = P##q0;                                            \
        if constexpr ((I) == 1) *reinterpret_cast<uint4*>(st__ + oW1) = P##q1;                                            \
        if constexpr ((I) == 2) *reinterpret_cast<uint4*>(st__ + 8192 + oW0) = P##g0;                                     \
        if constexpr ((I) == 3) *reinterpret_cast<uint4*>(st__ + 8192 + oW1) = P##g1;                                     \
        if constexpr ((I) == 4) {                                                                                         \
            const float sreg__ = (tid & 64) ? (P##ok ? -P##raw : 0.f) : (P##ok ? P##raw * lse_mul : kNegInf);             \
            reinterpret_cast<float*>(st__ + 16384)[tid & 127] = sreg__;                                                   \
        }                                                                                                                 \
    } while (0)
#define CM3P_GLOAD_PIECE(P, t_, I)                                                                                       \
    do {                                                                                                                  \
        const int t__ = (t_);                                                                                             \
        const int r0 = min(t__ * 64 + srow, S - 1), r1 = min(t__ * 64 + 32 + srow, S - 1);                                 \
        if constexpr ((I) == 0) P##q0 = *reinterpret_cast<const uint4*>(qbase + (int64_t)r0 * ld + schunk);               \
        if constexpr ((I) == 1) P##q1 = *reinterpret_cast<const uint4*>(qbase + (int64_t)r1 * ld + schunk);               \
        if constexpr ((I) == 2) P##g0 = *reinterpret_cast<const uint4*>(dobase + (int64_t)r0 * ldo + schunk);             \
        if constexpr ((I) == 3) P##g1 = *reinterpret_cast<const uint4*>(dobase + (int64_t)r1 * ldo + schunk);             \
        if constexpr ((I) == 4) {                                                                                         \
            const int q__ = t__ * 64 + (tid & 63);                                                                        \
            P##raw = stat_src[min(q__, S - 1)];                                                                           \
            P##ok = q__ < S;                                                                                              \
        }                                                                                                                 \
    } while (0)

    // register-resident LDS fragments of the current 32-query block
    bf16x8 Qf[4], Gf[4];     // rows of Q / dO (A operands of the score products)
    bf16x8 gT[2][2], qT[2][2];  // [sp][db]: dO^T / Q^T (A operands of the gradient products)
    f32x16 isv, idv;         // -lse / scale and -delta of the block's 32 rows, in accumulator layout
    auto load_init = [&](const char* sq, f32x16& v, int which, int half) {  // `half` of the 16 values: two 16-byte reads
#pragma unroll
        for (int g = 2 * half; g < 2 * half + 2; ++g) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(sq + oI + 256 * which + 32 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[4 * g + r] = a[r];
        }
    };
    // sq = slot base + 4096 * qb (row fragments), 128 * qb extra for the init floats is folded in by the caller via sqi
    auto loadS_all = [&](const char* sq, const char* sqi) {
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            Qf[s4] = ld_frag(sq + oR[s4]);
            Gf[s4] = ld_frag(sq + 8192 + oR[s4]);
        }
        load_init(sqi, isv, 0, 0);
        load_init(sqi, isv, 0, 1);
        load_init(sqi, idv, 1, 0);
        load_init(sqi, idv, 1, 1);
    };
    auto loadG_one = [&](const char* sq, int sp, int db, int which) {  // which: 0 = dO^T, 1 = Q^T
        const char* base = sq + (which == 0 ? 8192 : 0) + 2048 * sp;
        const bf16x8 f = ld_fragT(base + oTlo[db], base + oThi[db]);
        if (which == 0) gT[sp][db] = f;
        else qT[sp][db] = f;
    };

    // One pipeline step: scores of block Y (key block KBY; its query block's fragments are resident) next to the exponentials
    // of block X, then the gradient products of block X (key block KBX).  KBY == 1: Y is the last user of the resident row
    // fragments, which are reloaded for the next query block from `nS` / `nSi` right behind the MFMAs that read them.
    // KBX == 1: X is the last user of the resident transposed fragments, reloaded from `nG` the same way.
    // `hook(chunk)` is called once in each of the 16 chunks (integral_constant 0..15), behind the chunk's MFMA: the tile's staging
    // pieces ride there.
    auto step = [&](auto kbx_c, auto kby_c, f32x16& Xs, f32x16& Xdp, f32x16& Ys, f32x16& Ydp, const char* nS, const char* nSi,
                    const char* nG, auto&& hook) {
        constexpr int KBX = decltype(kbx_c)::value, KBY = decltype(kby_c)::value;
#define CM3P_HOOK(C) hook(std::integral_constant<int, (C)>{})
        CM3P_SB();
        mfma_vc(Ys, Qf[0], kf[KBY][0], isv);
        exp2_pair<PRE>(Xs, 0, cm);
        CM3P_HOOK(0);
        CM3P_SB();
        mfma_vc(Ydp, Gf[0], vf[KBY][0], idv);
        exp2_pair<PRE>(Xs, 2, cm);
        if constexpr (KBY == 1 && !(CM3P_ABL & 4)) {
            Qf[0] = ld_frag(nS + oR[0]);
            Gf[0] = ld_frag(nS + 8192 + oR[0]);
            load_init(nSi, isv, 0, 0);
        }
        CM3P_HOOK(1);
        CM3P_SB();
        SC_ACC(Ys, Qf[1], kf[KBY][1], isv);
        exp2_pair<PRE>(Xs, 4, cm);
        if constexpr (KBY == 1 && !(CM3P_ABL & 4)) {
            load_init(nSi, isv, 0, 1);
            Qf[1] = ld_frag(nS + oR[1]);
        }
        CM3P_HOOK(2);
        CM3P_SB();
        SC_ACC(Ydp, Gf[1], vf[KBY][1], idv);
        exp2_pair<PRE>(Xs, 6, cm);
        if constexpr (KBY == 1 && !(CM3P_ABL & 4)) {
            Gf[1] = ld_frag(nS + 8192 + oR[1]);
            load_init(nSi, idv, 1, 0);
        }
        CM3P_HOOK(3);
        CM3P_SB();
        SC_ACC(Ys, Qf[2], kf[KBY][2], isv);
        exp2_pair<PRE>(Xs, 8, cm);
        if constexpr (KBY == 1 && !(CM3P_ABL & 4)) {
            Qf[2] = ld_frag(nS + oR[2]);
            load_init(nSi, idv, 1, 1);
        }
        CM3P_HOOK(4);
        CM3P_SB();
        SC_ACC(Ydp, Gf[2], vf[KBY][2], idv);
        exp2_pair<PRE>(Xs, 10, cm);
        if constexpr (KBY == 1 && !(CM3P_ABL & 4)) Gf[2] = ld_frag(nS + 8192 + oR[2]);
        CM3P_HOOK(5);
        CM3P_SB();
        SC_ACC(Ys, Qf[3], kf[KBY][3], isv);
        exp2_pair<PRE>(Xs, 12, cm);
        if constexpr (KBY == 1 && !(CM3P_ABL & 4)) Qf[3] = ld_frag(nS + oR[3]);
        CM3P_HOOK(6);
        CM3P_SB();
        SC_ACC(Ydp, Gf[3], vf[KBY][3], idv);
        exp2_pair<PRE>(Xs, 14, cm);
        if constexpr (KBY == 1 && !(CM3P_ABL & 4)) Gf[3] = ld_frag(nS + 8192 + oR[3]);
        CM3P_HOOK(7);
        CM3P_SB();
        // ---- gradient products of X: dV^T += dO^T P, dK^T += Q^T dS
#if CM3P_ABL & 32
        const bf16x8 pf0 = Qf[0], ds0 = Gf[0], pf1 = Qf[1], ds1 = Gf[1];
#define CM3P_VALU(x)
#else
#define CM3P_VALU(x) x
        const bf16x8 pf0 = acc_to_frag(Xs, 0);
#endif
        dv[0][KBX] = mfma32(gT[0][0], pf0, dv[0][KBX]);
        if constexpr (KBX == 1 && !(CM3P_ABL & 4)) loadG_one(nG, 0, 0, 0);
        CM3P_VALU(_Pragma("unroll") for (int i = 0; i < 4; ++i) Xdp[i] *= Xs[i];)
        CM3P_HOOK(8);
        CM3P_SB();
        dv[1][KBX] = mfma32(gT[0][1], pf0, dv[1][KBX]);
        if constexpr (KBX == 1 && !(CM3P_ABL & 4)) loadG_one(nG, 0, 1, 0);
        CM3P_VALU(_Pragma("unroll") for (int i = 4; i < 8; ++i) Xdp[i] *= Xs[i];)
        CM3P_VALU(const bf16x8 ds0 = acc_to_frag(Xdp, 0);)
        CM3P_HOOK(9);
        CM3P_SB();
        dk[0][KBX] = mfma32(qT[0][0], ds0, dk[0][KBX]);
        if constexpr (KBX == 1 && !(CM3P_ABL & 4)) loadG_one(nG, 0, 0, 1);
        CM3P_VALU(const bf16x8 pf1 = acc_to_frag(Xs, 1);)
        CM3P_HOOK(10);
        CM3P_SB();
        dk[1][KBX] = mfma32(qT[0][1], ds0, dk[1][KBX]);
        if constexpr (KBX == 1 && !(CM3P_ABL & 4)) loadG_one(nG, 0, 1, 1);
        CM3P_VALU(_Pragma("unroll") for (int i = 8; i < 12; ++i) Xdp[i] *= Xs[i];)
        CM3P_HOOK(11);
        CM3P_SB();
        dv[0][KBX] = mfma32(gT[1][0], pf1, dv[0][KBX]);
        if constexpr (KBX == 1 && !(CM3P_ABL & 4)) loadG_one(nG, 1, 0, 0);
        CM3P_VALU(_Pragma("unroll") for (int i = 12; i < 16; ++i) Xdp[i] *= Xs[i];)
        CM3P_VALU(const bf16x8 ds1 = acc_to_frag(Xdp, 1);)
        CM3P_HOOK(12);
        CM3P_SB();
        dv[1][KBX] = mfma32(gT[1][1], pf1, dv[1][KBX]);
        if constexpr (KBX == 1 && !(CM3P_ABL & 4)) loadG_one(nG, 1, 1, 0);
        CM3P_HOOK(13);
        CM3P_SB();
        dk[0][KBX] = mfma32(qT[1][0], ds1, dk[0][KBX]);
        if constexpr (KBX == 1 && !(CM3P_ABL & 4)) loadG_one(nG, 1, 0, 1);
        CM3P_HOOK(14);
        CM3P_SB();
        dk[1][KBX] = mfma32(qT[1][1], ds1, dk[1][KBX]);
        if constexpr (KBX == 1 && !(CM3P_ABL & 4)) loadG_one(nG, 1, 1, 1);
        CM3P_HOOK(15);
        CM3P_SB();
#undef CM3P_HOOK
    };

    f32x16 sA, dpA, sB, dpB;
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    // one tile (ring slot SL): publish tile t+1, sync, start the loads of tile t+2, then the four steps
    auto tile = [&](auto slot_c, int t) {
        constexpr int SL = decltype(slot_c)::value, NS = (SL + 1) % kSlots3;
        const char* st = smem + SL * kBwdStage;
        char* nst = smem + NS * kBwdStage;
        // Tile t+1 sits in the staging set of ITS parity since tile t-1 and is stored to its slot (whose last readers finished
        // tile t-3) piece by piece in the score chunks of step 1; the barrier between steps 1 and 2 publishes it just before step 2
        // starts reading it; the loads of tile t+3 then take the set over, piece by piece in the gradient chunks of step 2.
        auto no_hook = [&](auto) {};
        auto store_hook = [&](auto c) {
            constexpr int C = decltype(c)::value;
            if constexpr (!(CM3P_ABL & 2) && C < 5) {
                if constexpr (SL & 1) CM3P_LSTORE_PIECE(a, nst, C);
                else CM3P_LSTORE_PIECE(b, nst, C);
            }
        };
        auto load_hook = [&](auto c) {
            constexpr int C = decltype(c)::value;
            if constexpr (!(CM3P_ABL & 2) && C >= 8 && C < 13) {
                if constexpr (SL & 1) CM3P_GLOAD_PIECE(a, t + 3, C - 8);
                else CM3P_GLOAD_PIECE(b, t + 3, C - 8);
            }
        };
        step(I0{}, I1{}, sA, dpA, sB, dpB, st + 4096, st + 128, nullptr, no_hook);     // X = (qb0, kb0), Y = (qb0, kb1); row fragments -> qb1
        step(I1{}, I0{}, sB, dpB, sA, dpA, nullptr, nullptr, st + 4096, store_hook);  // X = (qb0, kb1), Y = (qb1, kb0); transposed -> qb1
        if constexpr (!(CM3P_ABL & 1)) __syncthreads();
        step(I0{}, I1{}, sA, dpA, sB, dpB, nst, nst, nullptr, load_hook);              // X = (qb1, kb0), Y = (qb1, kb1); row fragments -> next tile
        step(I1{}, I0{}, sB, dpB, sA, dpA, nullptr, nullptr, nst, no_hook);            // X = (qb1, kb1), Y = next tile's (qb0, kb0)
    };

    // prologue: tile 0 in slot 0, tile 1 in flight, fragments of (tile 0, qb0), scores of its first block
    CM3P_GLOAD(a, 0);
    CM3P_LSTORE(a, smem);
    CM3P_GLOAD(b, 1);  // odd tiles: set b
    CM3P_GLOAD(a, 2);  // even tiles: set a
    __syncthreads();
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
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // (the only place a result is read right behind its asm MFMA chain)
    CM3P_SB();

    // (tiles past the last one - when the tile count is not a multiple of the ring size - re-read the last rows with a -inf
    //  score offset and add exact zeros: no exit edge inside the unrolled ring, so the accumulators never change registers)
    for (int t = 0; t < n_tiles; t += kSlots3) {
        tile(std::integral_constant<int, 0>{}, t);
        tile(std::integral_constant<int, 1>{}, t + 1);
        tile(std::integral_constant<int, 2>{}, t + 2);
        tile(std::integral_constant<int, 3>{}, t + 3);
    }

    // ---- epilogue: dK = scale * dK^T acc (inverse rotary applied), dV; keys under the padding mask get zeros
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        const int krow = k0 + 32 * kb + (lane & 31);
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


// -----------------------------------------------------------------------------------------------------------------------------
// dQ and delta, hand-scheduled like attn_bwd_dkv3_kernel: ONE wave per SIMD, wave w owns queries Q0 + 64 w .. + 63 (two 32-query
// column blocks qb), workgroup = 4 waves = 256 queries; the blocks u = (kb, qb) of a 64-key tile are software pipelined in one
// hand-placed instruction stream.
//   scores    S^T = K Q^T (asm MFMA from zero, VGPR result; -lse log2 e is added by the v_fma that also applies scale log2 e)
//             dP^T = V dO^T - delta (asm MFMA, initial accumulator = -delta of the lane's query), Q / dO fragments in AGPRs
//   gradient  dQ^T += K^T dS^T (compiler MFMAs, AGPR accumulators), K^T by transposed reads of the K image
// K / V row fragments and K^T fragments are register-resident per 32-key block, shared by its two query blocks and reloaded for
// the next key block right behind the last MFMA that read them.  Keys under the padding mask (and rows past the end of the
// sequence) are staged with a ZERO K row, so nothing is masked in the loop.
// -----------------------------------------------------------------------------------------------------------------------------
constexpr int kDq3Stage = 2 * 8192;

__device__ __forceinline__ void exp2_fma_pair(f32x16& s, int i, float cm, float nl) {
    s[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[i], cm, nl));
    s[i + 1] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[i + 1], cm, nl));
}

__global__ __launch_bounds__(256, 1) void attn_bwd_dq3_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ d_o,
                                                              const uint16_t* __restrict__ o_rows, const float* __restrict__ lse,
                                                              float* __restrict__ delta, uint16_t* __restrict__ dqkv,
                                                              const uint8_t* __restrict__ kmask, int Smax, int nh, float scale,
                                                              const float* __restrict__ rope_cos, const float* __restrict__ rope_sin,
                                                              int64_t pos_batch_stride, VarLen vl, int pre) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, hh = lane >> 5;
    int qblk, head, b;
    decode_block((Smax + 255) / 256, nh, qblk, head, b);
    const int Q0 = qblk * 256;
    const SeqView sv(vl, b, head, Smax, nh);
    const int S = sv.S;
    if (Q0 >= S) return;
    const int q0 = Q0 + wid * 64;
    const int64_t ld = (int64_t)3 * nh * 64;
    const uint16_t* qbase = qkv + sv.row0 * ld + head * 64;
    const uint16_t* kbase = qbase + nh * 64;
    const uint16_t* vbase = qbase + 2 * nh * 64;
    const int64_t ldo = (int64_t)nh * 64;
    const uint16_t* dobase = d_o + sv.row0 * ldo + head * 64;
    const uint16_t* obase = o_rows + sv.row0 * ldo + head * 64;
    const float cm = pre ? 1.0f : scale * kLog2e;  // (q carries scale * log2 e when pre: the v_fma then only adds -lse * log2 e)

    bf16x8 qf[2][4], dof[2][4];  // B operands of the asm MFMAs ("a" constraint: they live in AGPRs)
    f32x16 dlt_init[2];
    float nl2[2];  // -lse * log2(e) of the lane's queries (-inf for rows with no visible key: p = 0)
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int qrow = q0 + 32 * qb + (lane & 31);
        const int qrow_c = min(qrow, S - 1);
        float dlt = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            qf[qb][s] = gload_frag(qbase + (int64_t)qrow_c * ld + 16 * s + 8 * hh);
            dof[qb][s] = gload_frag(dobase + (int64_t)qrow_c * ldo + 16 * s + 8 * hh);
            const bf16x8 of = gload_frag(obase + (int64_t)qrow_c * ldo + 16 * s + 8 * hh);
#pragma unroll
            for (int j = 0; j < 8; ++j) dlt += (float)of[j] * (float)dof[qb][s][j];
        }
        // delta[q] = sum_d dO[q, d] O[q, d]: the other half of the row sits 32 lanes away.  Published for the dkv kernel.
        dlt += __shfl_xor(dlt, 32, 64);
        const int64_t stat = sv.stat0 + qrow_c;
        if (hh == 0 && qrow < S) delta[stat] = dlt;
        nl2[qb] = -lse[stat] * kLog2e;
#pragma unroll
        for (int i = 0; i < 16; ++i) dlt_init[qb][i] = -dlt;
        // dO fed the delta sum on the VALU, so the compiler holds it in VGPRs and would copy it into an AGPR right in front of
        // every asm MFMA (v_accvgpr_write -> MFMA operand: a hazard nobody pads for an asm statement - stale operands, NaN).
        // Re-define both operand sets HERE as AGPR values; from now on only "a" operands read them.
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            asm volatile("" : "+a"(qf[qb][s]));
            asm volatile("" : "+a"(dof[qb][s]));
        }
    }
    f32x16 dq[2][2];  // [d block][query block]
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
            for (int i = 0; i < 16; ++i) dq[db][qb][i] = 0.f;

    // per-lane LDS byte offsets inside a slot; everything else is an immediate
    const int l31 = lane & 31, g4 = lane >> 4, i16 = lane & 15;
    int oR[4], oTlo[2], oThi[2];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) oR[s4] = off_R(l31, 2 * s4 + hh);
#pragma unroll
    for (int db = 0; db < 2; ++db) {
        const int row = 4 * hh + (i16 >> 2), col = 32 * db + 16 * (g4 & 1) + 4 * (i16 & 3);
        oTlo[db] = off_T(row, col);
        oThi[db] = off_T(row + 8, col);
    }

    const int n_tiles = (S + 63) / 64;
    // two staging sets (even / odd tiles), plain named scalars (see attn_bwd_dkv3_kernel)
    uint4 ak0, ak1, av0, av1, bk0, bk1, bv0, bv1;
    uint8_t am0 = 0, am1 = 0, bm0 = 0, bm1 = 0;
    bool aok0 = false, aok1 = false, bok0 = false, bok1 = false;
    const int srow = tid >> 3, schunk = (tid & 7) * 8;
    const int oW0 = off_R(srow, tid & 7), oW1 = off_R(srow + 32, tid & 7);
    const uint8_t* km = kmask ? kmask + sv.row0 : nullptr;
#define CM3P_GLOADQ(P, t_)                                                                     \
    do {                                                                                        \
        const int t__ = (t_);                                                                   \
        const int r0 = min(t__ * 64 + srow, S - 1), r1 = min(t__ * 64 + 32 + srow, S - 1);       \
        P##k0 = *reinterpret_cast<const uint4*>(kbase + (int64_t)r0 * ld + schunk);             \
        P##k1 = *reinterpret_cast<const uint4*>(kbase + (int64_t)r1 * ld + schunk);             \
        P##v0 = *reinterpret_cast<const uint4*>(vbase + (int64_t)r0 * ld + schunk);             \
        P##v1 = *reinterpret_cast<const uint4*>(vbase + (int64_t)r1 * ld + schunk);             \
        if (km) {                                                                               \
            P##m0 = km[r0];                                                                     \
            P##m1 = km[r1];                                                                     \
        }                                                                                       \
        P##ok0 = t__ * 64 + srow < S;                                                           \
        P##ok1 = t__ * 64 + 32 + srow < S;                                                      \
    } while (0)
    // the K rows of keys that no query may see are stored as zeros
#define CM3P_LSTOREQ(P, st_)                                                                    \
    do {                                                                                        \
        char* st__ = (st_);                                                                     \
        const bool z0 = !(P##ok0 && (km == nullptr || P##m0 != 0)), z1 = !(P##ok1 && (km == nullptr || P##m1 != 0)); \
        *reinterpret_cast<uint4*>(st__ + oW0) = z0 ? uint4{0u, 0u, 0u, 0u} : P##k0;             \
        *reinterpret_cast<uint4*>(st__ + oW1) = z1 ? uint4{0u, 0u, 0u, 0u} : P##k1;             \
        *reinterpret_cast<uint4*>(st__ + 8192 + oW0) = P##v0;                                   \
        *reinterpret_cast<uint4*>(st__ + 8192 + oW1) = P##v1;                                   \
    } while (0)

    // register-resident LDS fragments of the current 32-key block
    bf16x8 Kf[4], Vf[4];  // rows of K / V (A operands of the score products)
    bf16x8 kT[2][2];      // [sp][db]: K^T (A operands of the gradient product)
    auto loadT_one = [&](const char* sq, int sp, int db) {
        const char* base = sq + 2048 * sp;
        kT[sp][db] = ld_fragT(base + oTlo[db], base + oThi[db]);
    };

    // One pipeline step: scores of block Y (query block QBY; its key block's fragments are resident) next to the exponentials of
    // block X, then the gradient product of block X (query block QBX).  QBY == 1: Y is the last user of the resident row
    // fragments, reloaded for the next key block from `nS`.  QBX == 1: X is the last user of the K^T fragments, reloaded from `nG`.
    auto step = [&](auto qbx_c, auto qby_c, f32x16& Xs, f32x16& Xdp, f32x16& Ys, f32x16& Ydp, const char* nS, const char* nG) {
        constexpr int QBX = decltype(qbx_c)::value, QBY = decltype(qby_c)::value;
        const float nl = nl2[QBX];
        CM3P_SB();
        mfma_v0(Ys, Kf[0], qf[QBY][0]);
        exp2_fma_pair(Xs, 0, cm, nl);
        CM3P_SB();
        mfma_vc(Ydp, Vf[0], dof[QBY][0], dlt_init[QBY]);
        exp2_fma_pair(Xs, 2, cm, nl);
        if constexpr (QBY == 1) {
            Kf[0] = ld_frag(nS + oR[0]);
            Vf[0] = ld_frag(nS + 8192 + oR[0]);
        }
        CM3P_SB();
        mfma_va(Ys, Kf[1], qf[QBY][1]);
        exp2_fma_pair(Xs, 4, cm, nl);
        if constexpr (QBY == 1) Kf[1] = ld_frag(nS + oR[1]);
        CM3P_SB();
        mfma_va(Ydp, Vf[1], dof[QBY][1]);
        exp2_fma_pair(Xs, 6, cm, nl);
        if constexpr (QBY == 1) Vf[1] = ld_frag(nS + 8192 + oR[1]);
        CM3P_SB();
        mfma_va(Ys, Kf[2], qf[QBY][2]);
        exp2_fma_pair(Xs, 8, cm, nl);
        if constexpr (QBY == 1) Kf[2] = ld_frag(nS + oR[2]);
        CM3P_SB();
        mfma_va(Ydp, Vf[2], dof[QBY][2]);
        exp2_fma_pair(Xs, 10, cm, nl);
        if constexpr (QBY == 1) Vf[2] = ld_frag(nS + 8192 + oR[2]);
        CM3P_SB();
        mfma_va(Ys, Kf[3], qf[QBY][3]);
        exp2_fma_pair(Xs, 12, cm, nl);
        if constexpr (QBY == 1) Kf[3] = ld_frag(nS + oR[3]);
        CM3P_SB();
        mfma_va(Ydp, Vf[3], dof[QBY][3]);
        exp2_fma_pair(Xs, 14, cm, nl);
        if constexpr (QBY == 1) Vf[3] = ld_frag(nS + 8192 + oR[3]);
        CM3P_SB();
        // ---- gradient product of X: dQ^T += K^T dS^T with dS^T = P^T o (dP^T - delta) (the 1/sqrt(d) scale is applied once, to dQ)
#pragma unroll
        for (int i = 0; i < 8; ++i) Xdp[i] *= Xs[i];
        const bf16x8 ds0 = acc_to_frag(Xdp, 0);
        dq[0][QBX] = mfma32(kT[0][0], ds0, dq[0][QBX]);
        if constexpr (QBX == 1) loadT_one(nG, 0, 0);
#pragma unroll
        for (int i = 8; i < 12; ++i) Xdp[i] *= Xs[i];
        CM3P_SB();
        dq[1][QBX] = mfma32(kT[0][1], ds0, dq[1][QBX]);
        if constexpr (QBX == 1) loadT_one(nG, 0, 1);
#pragma unroll
        for (int i = 12; i < 16; ++i) Xdp[i] *= Xs[i];
        const bf16x8 ds1 = acc_to_frag(Xdp, 1);
        CM3P_SB();
        dq[0][QBX] = mfma32(kT[1][0], ds1, dq[0][QBX]);
        if constexpr (QBX == 1) loadT_one(nG, 1, 0);
        CM3P_SB();
        dq[1][QBX] = mfma32(kT[1][1], ds1, dq[1][QBX]);
        if constexpr (QBX == 1) loadT_one(nG, 1, 1);
        CM3P_SB();
    };

    f32x16 sA, dpA, sB, dpB;
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    auto tile = [&](auto slot_c, int t) {
        constexpr int SL = decltype(slot_c)::value, NS = (SL + 1) % kSlots3;
        const char* st = smem + SL * kDq3Stage;
        char* nst = smem + NS * kDq3Stage;
        // tile t+1 sits in the staging set of its parity since the top of tile t-1; tile t+3 takes the set over
        if constexpr (SL & 1) CM3P_LSTOREQ(a, nst);
        else CM3P_LSTOREQ(b, nst);
        __syncthreads();
        if constexpr (SL & 1) CM3P_GLOADQ(a, t + 3);
        else CM3P_GLOADQ(b, t + 3);
        step(I0{}, I1{}, sA, dpA, sB, dpB, st + 4096, nullptr);  // X = (kb0, qb0), Y = (kb0, qb1); row fragments -> kb1
        step(I1{}, I0{}, sB, dpB, sA, dpA, nullptr, st + 4096);  // X = (kb0, qb1), Y = (kb1, qb0); K^T fragments -> kb1
        step(I0{}, I1{}, sA, dpA, sB, dpB, nst, nullptr);        // X = (kb1, qb0), Y = (kb1, qb1); row fragments -> next tile
        step(I1{}, I0{}, sB, dpB, sA, dpA, nullptr, nst);        // X = (kb1, qb1), Y = next tile's (kb0, qb0)
    };

    // prologue: tile 0 in slot 0, tiles 1 and 2 in flight, fragments of (tile 0, kb0), scores of its first block
    CM3P_GLOADQ(a, 0);
    CM3P_LSTOREQ(a, smem);
    CM3P_GLOADQ(b, 1);
    CM3P_GLOADQ(a, 2);
    __syncthreads();
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
        Kf[s4] = ld_frag(smem + oR[s4]);
        Vf[s4] = ld_frag(smem + 8192 + oR[s4]);
    }
#pragma unroll
    for (int sp = 0; sp < 2; ++sp)
#pragma unroll
        for (int db = 0; db < 2; ++db) loadT_one(smem, sp, db);
    mfma_v0(sA, Kf[0], qf[0][0]);
    mfma_vc(dpA, Vf[0], dof[0][0], dlt_init[0]);
#pragma unroll
    for (int s4 = 1; s4 < 4; ++s4) {
        mfma_va(sA, Kf[s4], qf[0][s4]);
        mfma_va(dpA, Vf[s4], dof[0][s4]);
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // (the only place a result is read right behind its asm MFMA chain)
    CM3P_SB();

    // (tiles past the last one stage zero K rows and add exact zeros)
    for (int t = 0; t < n_tiles; t += kSlots3) {
        tile(std::integral_constant<int, 0>{}, t);
        tile(std::integral_constant<int, 1>{}, t + 1);
        tile(std::integral_constant<int, 2>{}, t + 2);
        tile(std::integral_constant<int, 3>{}, t + 3);
    }

#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int qrow = q0 + 32 * qb + (lane & 31);
        if (qrow < S) {
            uint16_t* drow = dqkv + (sv.row0 + qrow) * ld + head * 64;
            if (rope_cos) {  // backward of apply_rotary_pos_emb: dims d / d+32 are the two accumulator blocks of this lane
                const int64_t prow = sv.pos0(b, pos_batch_stride) + qrow;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 lo4 = {dq[0][qb][4 * g], dq[0][qb][4 * g + 1], dq[0][qb][4 * g + 2], dq[0][qb][4 * g + 3]};
                    f32x4 hi4 = {dq[1][qb][4 * g], dq[1][qb][4 * g + 1], dq[1][qb][4 * g + 2], dq[1][qb][4 * g + 3]};
                    rope_rotate4<true>(lo4, hi4, rope_cos + prow * 32, rope_sin + prow * 32, 8 * g + 4 * hh);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        dq[0][qb][4 * g + r] = lo4[r];
                        dq[1][qb][4 * g + r] = hi4[r];
                    }
                }
            }
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int d = 32 * db + 8 * g + 4 * hh;
                    *reinterpret_cast<uint2*>(drow + d) =
                        uint2{pack_bf16x2(dq[db][qb][4 * g] * scale, dq[db][qb][4 * g + 1] * scale),
                              pack_bf16x2(dq[db][qb][4 * g + 2] * scale, dq[db][qb][4 * g + 3] * scale)};
                }
        }
    }
}

}  // namespace

// Launcher used by attention.hip's cm3p_attn_bwd / cm3p_attn_bwd_varlen for window < 0.
int cm3p_launch_attn_bwd_global(const void* qkv, const void* out, const void* dout, const float* lse, float* delta, void* dqkv,
                                const uint8_t* key_mask, int B, int S, int nh, float scale, const float* cos_tab, const float* sin_tab,
                                int64_t pos_batch_stride, const int* cu_seqlens, int64_t total, int stages, int pre, hipStream_t s) {
    const VarLen vl{cu_seqlens, total};
    const dim3 grid(((S + 255) / 256) * nh * B);  // 1-D: decode_block() maps it XCD-aware
    if (stages & CM3P_ATTN_BWD_DQ) {
        attn_bwd_dq3_kernel<<<grid, 256, kSlots3 * kDq3Stage, s>>>((const uint16_t*)qkv, (const uint16_t*)dout, (const uint16_t*)out, lse, delta,
                                                                   (uint16_t*)dqkv, key_mask, S, nh, scale, cos_tab, sin_tab, pos_batch_stride, vl, pre);
        if (hipGetLastError() != hipSuccess) return CM3P_ERR_LAUNCH;
    }
    if (stages & CM3P_ATTN_BWD_DKV) {
#define CM3P_DKV3_ARGS (const uint16_t*)qkv, (const uint16_t*)dout, lse, delta, (uint16_t*)dqkv, key_mask, S, nh, scale, cos_tab, sin_tab, pos_batch_stride, vl
        if (pre) attn_bwd_dkv3_kernel<true><<<grid, 256, kSlots3 * kBwdStage, s>>>(CM3P_DKV3_ARGS);
        else attn_bwd_dkv3_kernel<false><<<grid, 256, kSlots3 * kBwdStage, s>>>(CM3P_DKV3_ARGS);
#undef CM3P_DKV3_ARGS
    }
    return CM3P_OK;
}

// timing-only ablation switches this object was built with (0 in every shipped build: cm3p_build_ablation_flags, tests/test_cabi.py)
int cm3p_ablation_flags_attention_bwd() { return (CM3P_ABL); }
