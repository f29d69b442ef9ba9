// Forward of the GLOBAL attention layers (window < 0), head_dim 64, gfx950: the kernel behind cm3p_attn_fwd /
// cm3p_attn_fwd_varlen when no sliding window is set (the sliding-window layers keep the band kernel of attention.hip).
//
// Replaces F.scaled_dot_product_attention(q, k, v, attn_mask, scale, is_causal=False) (TF:integrations/sdpa_attention.py:153-163,
// called from TF:models/modernbert/modeling_modernbert.py:286-297) under the key-padding mask of TF:masking_utils.py:168-179.
//
// Hand-scheduled like the backward kernels of attention_bwd.hip: ONE wave per SIMD with the whole 512-entry register file, wave w
// owns queries Q0 + 64 w .. + 63 as two 32-query column blocks qb (query = MFMA column = lane & 31), workgroup = 4 waves = 256
// queries, 64-key tiles staged global -> registers -> LDS into a four-slot ring (two staging sets: loads fly for two tile periods;
// one barrier per tile).
//   scores   S^T = K Q^T as inline-asm MFMAs with VGPR results (the VALU reads them; Q fragments live in AGPRs), from zero - or from
//            a per-key bias row (0 / -inf) when keys can be invisible (padding mask, ragged length, packed batches)
//   softmax  per lane = per query, in fp32 on the raw scores: p = exp2(s * scale*log2e - ref) (one v_fma + one v_exp per score;
//            nothing is pre-scaled and re-rounded to bf16).  `ref` is a lazily moved reference point: it follows the running
//            maximum only when a block's maximum exceeds it by more than 2^6, which is exact in exact arithmetic (it divides out)
//            and takes the accumulator rescale off the common path.
//   output   O^T += V^T P^T: compiler MFMAs with AGPR accumulators, P^T straight from the score registers (bf16 packs), V^T by
//            transposed LDS reads.
// The four 32 x 32 blocks u = (kb, qb) of a tile are software pipelined in ONE hand-placed instruction stream (sched_barrier
// between chunks): while block X's exponentials run on the VALU, the score MFMAs of the next block and the output MFMAs of X are
// issued in the gaps, and the running-maximum check of the next block rides behind X's last output MFMAs.  At head_dim 64 the
// kernel is VALU-issue co-limited (70 vector instructions per 8 MFMAs), which bounds it near 60 % of the matrix peak.
#include <type_traits>

#include "attn_common.h"

namespace {

constexpr int kFwd3Slots = 4;
constexpr int kFwd3Stage = 2 * 8192 + 256;  // K image, V image, 64 bias floats (0 visible / -inf invisible key)
#ifndef CM3P_FWD3_DEFER
#define CM3P_FWD3_DEFER 6.0f
#endif
constexpr float kDefer3 = CM3P_FWD3_DEFER;

template <bool MASKED>
__global__ __launch_bounds__(256, 1) void attn_fwd3_kernel(const uint16_t* __restrict__ qkv, uint16_t* __restrict__ out,
                                                           float* __restrict__ lse, const uint8_t* __restrict__ kmask, int Smax, int nh,
                                                           float scale, VarLen vl) {
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
    const uint8_t* km = kmask ? kmask + sv.row0 : nullptr;
    const float cm = scale * kLog2e;

    bf16x8 qf[2][4];  // B operands of the asm MFMAs ("a" constraint: they live in AGPRs)
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int qrow_c = min(q0 + 32 * qb + (lane & 31), S - 1);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            qf[qb][s] = *reinterpret_cast<const bf16x8*>(qbase + (int64_t)qrow_c * ld + 16 * s + 8 * hh);
            asm volatile("" : "+a"(qf[qb][s]));  // defined as an AGPR value here: never re-materialised in front of an MFMA
        }
    }
    f32x16 o[2][2];  // [d block][query block]
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
            for (int i = 0; i < 16; ++i) o[db][qb][i] = 0.f;
    // Softmax state per query (= per lane): every stored p, l and O is relative to the reference point mc_run (log2 units)
    float mc_run[2] = {0.f, 0.f}, l_run[2] = {0.f, 0.f};
    bool has_ref[2] = {false, false};

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
    // two staging sets (tiles of even / odd index), plain named scalars (a struct or an indexed array ends up in scratch); nothing
    // in a load block consumes a loaded value (see attn_bwd_dkv3_kernel)
    uint4 ak0, ak1, av0, av1, bk0, bk1, bv0, bv1;
    uint8_t amb = 1, bmb = 1;
    bool akok = false, bkok = false;
    const int srow = tid >> 3, schunk = (tid & 7) * 8;
    const int oW0 = off_R(srow, tid & 7), oW1 = off_R(srow + 32, tid & 7);
#define CM3P_GLOADF(P, t_)                                                                 \
    do {                                                                                    \
        const int t__ = (t_);                                                               \
        const int r0 = min(t__ * 64 + srow, S - 1), r1 = min(t__ * 64 + 32 + srow, S - 1);   \
        P##k0 = *reinterpret_cast<const uint4*>(kbase + (int64_t)r0 * ld + schunk);         \
        P##k1 = *reinterpret_cast<const uint4*>(kbase + (int64_t)r1 * ld + schunk);         \
        P##v0 = *reinterpret_cast<const uint4*>(vbase + (int64_t)r0 * ld + schunk);         \
        P##v1 = *reinterpret_cast<const uint4*>(vbase + (int64_t)r1 * ld + schunk);         \
        if constexpr (MASKED) {                                                             \
            const int key__ = t__ * 64 + (tid & 63);                                        \
            P##kok = key__ < S;                                                             \
            if (km) P##mb = km[min(key__, S - 1)];                                          \
        }                                                                                   \
    } while (0)
#define CM3P_LSTOREF(P, st_)                                                                                        \
    do {                                                                                                             \
        char* st__ = (st_);                                                                                          \
        *reinterpret_cast<uint4*>(st__ + oW0) = P##k0;                                                               \
        *reinterpret_cast<uint4*>(st__ + oW1) = P##k1;                                                               \
        *reinterpret_cast<uint4*>(st__ + 8192 + oW0) = P##v0;                                                        \
        *reinterpret_cast<uint4*>(st__ + 8192 + oW1) = P##v1;                                                        \
        if constexpr (MASKED) /* (threads 64..255 repeat the same values) */                                         \
            reinterpret_cast<float*>(st__ + 16384)[tid & 63] = (P##kok && P##mb != 0) ? 0.f : kNegInf;                \
    } while (0)

    // register-resident LDS fragments of the current 32-key block
    bf16x8 Kf[4];      // rows of K (A operands of the score product)
    bf16x8 vT[2][2];   // [sp][db]: V^T (A operands of the output product)
    f32x16 biasv;      // MASKED: 0 / -inf per key row, in accumulator layout
    auto loadT_one = [&](const char* sq, int sp, int db) {
        const char* base = sq + 8192 + 2048 * sp;
        vT[sp][db] = ld_fragT(base + oTlo[db], base + oThi[db]);
    };
    auto load_bias = [&](const char* sqi) {  // sqi = slot base + 128 * kb
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(sqi + oI + 32 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r) biasv[4 * g + r] = a[r];
        }
    };

    // exponentials of score pair (2 i, 2 i + 1) of block X, in place, and their row sum
    auto grp = [&](f32x16& Xs, int i, float nm, float& lsum) {
        const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(Xs[2 * i], cm, nm));
        const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(Xs[2 * i + 1], cm, nm));
        Xs[2 * i] = p0;
        Xs[2 * i + 1] = p1;
        lsum += p0;
        lsum += p1;
        asm volatile("" : "+v"(lsum));  // pins the partial sum to this chunk (the IR-level sinking pass ignores sched_barrier)
    };
    // running-maximum check of a freshly scored block (query block QB): the reference point moves only when the block maximum
    // exceeds it by more than 2^kDefer3 (or no reference exists yet) - rare after the first tile
    auto decide = [&](auto qb_c, float mt) {
        constexpr int QB = decltype(qb_c)::value;
#ifdef CM3P_FWD3_SHFL
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
#else
        // The other 16 key rows sit 32 lanes away.  v_permlane32_swap exchanges the upper half of its first operand with the
        // lower half of its second IN PLACE: the two operands must be different registers (handed the same register it returns
        // the other half's value twice), so the second one is an explicit copy the compiler cannot fold back.
        float mt2 = mt;
        asm volatile("" : "+v"(mt2));
        const auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, mt), __builtin_bit_cast(unsigned, mt2), false, false);
        mt = fmaxf(__builtin_bit_cast(float, sw[0]), __builtin_bit_cast(float, sw[1]));
#endif
        const float mts = __builtin_fmaf(mt, cm, -mc_run[QB]);  // block maximum relative to the reference, log2 units
        const bool seen = mt > kNegInf;
        const bool move = has_ref[QB] ? (mts > kDefer3) : seen;
        if (__any(move)) {
            // the accumulators are re-defined here by a statement that cannot be speculated: without it the compiler hoists the 32
            // AGPR -> VGPR copies of the rescale above the branch, into every step of the common path
            asm volatile("" : "+a"(o[0][QB]), "+a"(o[1][QB]));
            const float shift = has_ref[QB] ? fmaxf(mts, 0.f) : (seen ? mts : 0.f);
            const float alpha = has_ref[QB] ? __builtin_amdgcn_exp2f(-shift) : 1.0f;  // O = l = 0 before the first score
            has_ref[QB] = has_ref[QB] || seen;
            mc_run[QB] += shift;
            l_run[QB] *= alpha;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                o[0][QB][i] *= alpha;
                o[1][QB][i] *= alpha;
            }
        }
    };
    auto max8 = [&](const f32x16& Ys, int h, float m) {  // 4 x v_max3_f32 over half of the block's scores
#pragma unroll
        for (int i = 0; i < 4; ++i) m = max3(m, Ys[8 * h + 2 * i], Ys[8 * h + 2 * i + 1]);
        return m;
    };

    // One pipeline step.  X (query block QBX): exponentials + output products; Y (query block QBY): scores now, maximum check at
    // the end.  QBY == 1: Y is the last user of the resident K rows (and bias), reloaded for the next key block from `nS` / `nSi`
    // right behind the MFMAs that read them; QBX == 1: X is the last user of the V^T fragments, reloaded from `nG` the same way.
    auto step = [&](auto qbx_c, auto qby_c, f32x16& Xs, f32x16& Ys, const char* nS, const char* nSi, const char* nG) {
        constexpr int QBX = decltype(qbx_c)::value, QBY = decltype(qby_c)::value;
        const float nm = -mc_run[QBX];
        float lsum = 0.f;
        CM3P_SB();
        if constexpr (MASKED) mfma_vc(Ys, Kf[0], qf[QBY][0], biasv);
        else mfma_v0(Ys, Kf[0], qf[QBY][0]);
        grp(Xs, 0, nm, lsum);
        CM3P_SB();
        mfma_va(Ys, Kf[1], qf[QBY][1]);
        if constexpr (QBY == 1) {
            Kf[0] = ld_frag(nS + oR[0]);
            if constexpr (MASKED) load_bias(nSi);
        }
        grp(Xs, 1, nm, lsum);
        CM3P_SB();
        mfma_va(Ys, Kf[2], qf[QBY][2]);
        if constexpr (QBY == 1) Kf[1] = ld_frag(nS + oR[1]);
        grp(Xs, 2, nm, lsum);
        CM3P_SB();
        mfma_va(Ys, Kf[3], qf[QBY][3]);
        if constexpr (QBY == 1) Kf[2] = ld_frag(nS + oR[2]);
        grp(Xs, 3, nm, lsum);
        const bf16x8 pf0 = acc_to_frag(Xs, 0);
        CM3P_SB();
        o[0][QBX] = mfma32(vT[0][0], pf0, o[0][QBX]);
        if constexpr (QBY == 1) Kf[3] = ld_frag(nS + oR[3]);
        if constexpr (QBX == 1) loadT_one(nG, 0, 0);
        grp(Xs, 4, nm, lsum);
        CM3P_SB();
        o[1][QBX] = mfma32(vT[0][1], pf0, o[1][QBX]);
        if constexpr (QBX == 1) loadT_one(nG, 0, 1);
        grp(Xs, 5, nm, lsum);
        CM3P_SB();
        grp(Xs, 6, nm, lsum);
        CM3P_SB();
        grp(Xs, 7, nm, lsum);
        const bf16x8 pf1 = acc_to_frag(Xs, 1);
        l_run[QBX] += lsum;
        CM3P_SB();
        o[0][QBX] = mfma32(vT[1][0], pf1, o[0][QBX]);
        if constexpr (QBX == 1) loadT_one(nG, 1, 0);
        float mt = max8(Ys, 0, kNegInf);
        CM3P_SB();
        o[1][QBX] = mfma32(vT[1][1], pf1, o[1][QBX]);
        if constexpr (QBX == 1) loadT_one(nG, 1, 1);
        mt = max8(Ys, 1, mt);
        CM3P_SB();
        decide(qby_c, mt);
        CM3P_SB();
    };

    f32x16 sA, sB;
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    auto tile = [&](auto slot_c, int t) {
        constexpr int SL = decltype(slot_c)::value, NS = (SL + 1) % kFwd3Slots;
        const char* st = smem + SL * kFwd3Stage;
        char* nst = smem + NS * kFwd3Stage;
        // tile t+1 sits in the staging set of its parity since the top of tile t-1; tile t+3 takes the set over
        if constexpr (SL & 1) CM3P_LSTOREF(a, nst);
        else CM3P_LSTOREF(b, nst);
        __syncthreads();
        if constexpr (SL & 1) CM3P_GLOADF(a, t + 3);
        else CM3P_GLOADF(b, t + 3);
        step(I0{}, I1{}, sA, sB, st + 4096, st + 128, nullptr);  // X = (kb0, qb0), Y = (kb0, qb1); K rows -> kb1
        step(I1{}, I0{}, sB, sA, nullptr, nullptr, st + 4096);   // X = (kb0, qb1), Y = (kb1, qb0); V^T -> kb1
        step(I0{}, I1{}, sA, sB, nst, nst, nullptr);             // X = (kb1, qb0), Y = (kb1, qb1); K rows -> next tile
        step(I1{}, I0{}, sB, sA, nullptr, nullptr, nst);         // X = (kb1, qb1), Y = next tile's (kb0, qb0)
    };

    // prologue: tile 0 in slot 0, tiles 1 and 2 in flight, fragments of (tile 0, kb0), scores and maximum check of its first block
    CM3P_GLOADF(a, 0);
    CM3P_LSTOREF(a, smem);
    CM3P_GLOADF(b, 1);
    CM3P_GLOADF(a, 2);
    __syncthreads();
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) Kf[s4] = ld_frag(smem + oR[s4]);
    if constexpr (MASKED) load_bias(smem);
#pragma unroll
    for (int sp = 0; sp < 2; ++sp)
#pragma unroll
        for (int db = 0; db < 2; ++db) loadT_one(smem, sp, db);
    if constexpr (MASKED) mfma_vc(sA, Kf[0], qf[0][0], biasv);
    else mfma_v0(sA, Kf[0], qf[0][0]);
#pragma unroll
    for (int s4 = 1; s4 < 4; ++s4) mfma_va(sA, Kf[s4], qf[0][s4]);
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // (the only place a result is read right behind its asm MFMA chain)
    CM3P_SB();
    decide(I0{}, max8(sA, 1, max8(sA, 0, kNegInf)));
    CM3P_SB();

    // (tiles past the last one - when the tile count is not a multiple of the ring size - exist only in the MASKED instance, where
    //  their keys carry a -inf bias: p = 0; the host picks the unmasked instance only for S % 256 == 0)
    for (int t = 0; t < n_tiles; t += kFwd3Slots) {
        tile(std::integral_constant<int, 0>{}, t);
        tile(std::integral_constant<int, 1>{}, t + 1);
        tile(std::integral_constant<int, 2>{}, t + 2);
        tile(std::integral_constant<int, 3>{}, t + 3);
    }

#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int qrow = q0 + 32 * qb + (lane & 31);
        const float l_tot = l_run[qb] + __shfl_xor(l_run[qb], 32, 64);  // the other 16 key rows of every block sit 32 lanes away
        const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;  // rows with no visible key: exact zeros, lse = +inf
        if (qrow < S) {
            uint16_t* orow = out + (sv.row0 + qrow) * nh * 64 + head * 64;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int d = 32 * db + 8 * g + 4 * hh;
                    *reinterpret_cast<uint2*>(orow + d) = uint2{pack_bf16x2(o[db][qb][4 * g] * inv, o[db][qb][4 * g + 1] * inv),
                                                                pack_bf16x2(o[db][qb][4 * g + 2] * inv, o[db][qb][4 * g + 3] * inv)};
                }
            if (hh == 0) lse[sv.stat0 + qrow] = l_tot > 0.f ? (mc_run[qb] + __log2f(l_tot)) * 0.69314718055994531f : __builtin_huge_valf();
        }
    }
}

}  // namespace

// Launcher used by attention.hip's cm3p_attn_fwd / cm3p_attn_fwd_varlen for window < 0.
int cm3p_launch_attn_fwd_global(const void* qkv, void* out, float* lse, const uint8_t* key_mask, int B, int S, int nh, float scale,
                                const int* cu_seqlens, int64_t total, hipStream_t s) {
    const VarLen vl{cu_seqlens, total};
    const dim3 grid(((S + 255) / 256) * nh * B);  // 1-D: decode_block() maps it XCD-aware
    const size_t lds = kFwd3Slots * kFwd3Stage;
    // every key visible to every query and the tile count a multiple of the ring: no bias rows, scores start from zero
    if (key_mask == nullptr && cu_seqlens == nullptr && S % 256 == 0)
        attn_fwd3_kernel<false><<<grid, 256, lds, s>>>((const uint16_t*)qkv, (uint16_t*)out, lse, key_mask, S, nh, scale, vl);
    else
        attn_fwd3_kernel<true><<<grid, 256, lds, s>>>((const uint16_t*)qkv, (uint16_t*)out, lse, key_mask, S, nh, scale, vl);
    return CM3P_OK;
}
