// Forward of the GLOBAL attention layers as ONE software-pipelined instruction stream per wave, head_dim 64, gfx950 (r05).
// Same mathematics and the same MFMA formulation as attn_fwd_kernel (attention.hip: S^T = K Q^T with the query on the lane, the
// lazily moved reference point riding in as the score product's initial accumulator, O^T += V^T P^T straight from the accumulators),
// i.e. F.scaled_dot_product_attention (TF:integrations/sdpa_attention.py:153-163, called from
// TF:models/modernbert/modeling_modernbert.py:286-297) under the key-padding mask of TF:masking_utils.py:168-179.
//
// What is different is WHO overlaps the matrix pipe with the softmax arithmetic.  attn_fwd_kernel leaves it to three co-resident
// waves per SIMD, each of which runs QK^T -> max -> exp -> PV phase by phase; measured, a SIMD then delivers MFMA time PLUS VALU
// time (1375 cycles per 16 MFMAs, 40 % matrix-busy).  Here a workgroup is 4 waves, ONE per SIMD with the whole 512-entry register
// file, and a wave owns U = 4 blocks of 32 queries (128 queries; the workgroup 512).  Its stream is a chain of "periods", one per
// (key tile, query block) = sub-block j; period j issues 16 MFMAs - the 8 of PV(j-1), then the 8 of QK^T(j+1) - and between
// them, a few instructions per MFMA gap and placed by hand (sched_barrier chunks), the softmax arithmetic of sub-block j:
//     gaps 0-3   the tail of sub-block j-1 (eight row-sum adds, two packs), 16 v_max3 over the 32 scores of the lane; then the
//                wave-uniform decision whether any row's reference point must move - compare, four of the adds, then the branch (rare:
//                the cold block behind it rescales O, l, the scores and rewrites the 16-register -ref operand, per query row)
//     gaps 4-15  32 v_exp, 24 v_add (two partial row sums per query block), 16 v_cvt_pk, three gaps per 8-score chunk
// so a sub-block's scores are produced a period before they are needed and consumed a period after, and no MFMA result is read
// before 2 further MFMAs have been issued (hipcc pads no hazard for the inline-asm MFMAs that deliver the scores to VGPRs).
// Per 16 MFMAs the stream carries 97 VALU instructions (the old kernel: 155): no v_mov re-splats -ref (it is a persistent C
// operand), no address arithmetic (tiles arrive by LDS-DMA with scalar bases, fragments are read with per-lane offsets + immediates),
// no cross-lane traffic (both half-waves of a query decide and sum on their own; they meet in the cold block and in the epilogue).
// Every K / V fragment is read from LDS once per tile and feeds all four query blocks: 24 LDS reads and 4 (5) DMAs per 64 MFMAs.
//
// K / V tiles (64 keys): LDS-DMA (TileDma, attn_common.h) into a ring of 4 slots, three tiles ahead; ONE workgroup barrier per
// tile behind ONE counted s_waitcnt vmcnt (every wave issues exactly 4 - with a key mask 5 - DMAs per tile, unconditionally, rows
// clamped to the sequence: tests/test_kernel_isa.py pins the pattern).  Key validity (padding mask, keys past the sequence) is
// a 64-bit wave-uniform word per tile.  The sweep is TWO loops: the tiles in front of the first key some lane may not see run
// the stream above with no masking code at all (a not-taken branch per block costs 4 %), the rest - up to the tile of the last
// visible key; tiles behind it are not visited - a copy with one branch per block in front of three VALU instructions per score.
#include <stdlib.h>

#include <type_traits>

#include "attn_common.h"

namespace {

#ifndef CM3P_GABL
#define CM3P_GABL 0  // timing-only ablation builds (results wrong by construction): 1 no exponentials, 2 no max, 4 no row sums, 8 no tile DMA in the loop, 16 no barrier, 32 max and decision kept but no branch,
                    // 64 no K / V fragment reloads in the loop, 128 no bf16 packs, 256 the score products start from the constant 0,
                    // 512 the output product as pairs of v_mfma_f32_16x16x32_bf16 (same FLOPs; tools/ubench/attn_mfma_shape.sh)
#endif

// Cycle trace (trace builds only: -DCM3P_GTRACE=1, tools/attn_fwd_trace.py): every wave accumulates, in scalar registers, the shader cycles
// (s_memtime) it spends in six regions of the sweep and writes the sums at its end - region 0: the top of a tile (counted wait, barrier,
// validity word), 1: gaps 0-3 of a period (tail, maximum, decision), 2: gaps 4-7 (second half of PV), 3: gaps 8-11, 4: gaps 12-15 (QK^T of
// the next sub-block), 5: prologue + epilogue.  A stamp waits for the wave's outstanding LDS reads (s_memtime returns through lgkmcnt), so the
// build runs a few per cent slower than the product and its split, not its total, is the information.
#ifndef CM3P_GTRACE
#define CM3P_GTRACE 0
#endif
#if CM3P_GTRACE
__device__ unsigned long long* g_fwd_trace = nullptr;
#define GT_STAMP(k)                                            \
    do {                                                       \
        const unsigned long long now__ = __builtin_amdgcn_s_memtime(); \
        gt_acc[k] += now__ - gt_prev;                          \
        gt_prev = now__;                                       \
    } while (0)
#else
#define GT_STAMP(k) \
    do {            \
    } while (0)
#endif
// (CM3P_GTRACE == 2: the regions are gap 0, gap 1, gap 2, gap 3 up to the branch, the branch itself, and everything else;
//  CM3P_GTRACE == 3: no stamp inside the sweep - three per wave in all -, for the in-kernel clock: cycle total / s_memrealtime)
#define GT1(k)                                \
    do {                                      \
        if constexpr (CM3P_GTRACE == 1) GT_STAMP(k); \
    } while (0)
#define GT2(k)                                \
    do {                                      \
        if constexpr (CM3P_GTRACE == 2) GT_STAMP(k); \
    } while (0)

constexpr int kGSlots = 4;
constexpr int kGStageNoMask = 16384;          // K image (row fragments) + V image (transposed reads)
constexpr int kGStageMask = 16384 + 4 * 256;  // + one validity dword per key, a private copy per wave (the MASK instance adds 32 bytes behind the ring)
constexpr float kGDefer = 6.0f;

#define CM3P_IC(n) std::integral_constant<int, (n)> {}

// The score product delivers to VGPRs (the VALU consumes it) from AGPR operands; the output product accumulates in AGPRs.  Both as
// inline assembly: left to hipcc, the output accumulators of a function whose cold block touches them on the VALU get a VGPR home for
// part of the loop and are copied back and forth every tile (80 v_accvgpr_* per tile in the first build of this kernel).
// D (VGPRs) = A (AGPRs) * B (AGPRs) + C (VGPRs); D never overlaps an input
__device__ __forceinline__ void mfma_s0(f32x16& d, const bf16x8& a, const bf16x8& b, const f32x16& c) {
    if constexpr ((CM3P_GABL & 256) != 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(d) : "a"(a), "a"(b));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(d) : "a"(a), "a"(b), "v"(c));
}
__device__ __forceinline__ void mfma_s(f32x16& d, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "a"(a), "a"(b));
}
// D (AGPRs) += A (VGPRs) * B (VGPRs)
__device__ __forceinline__ void mfma_o(f32x16& d, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(d) : "v"(a), "v"(b));
}
// Timing only (CM3P_GABL & 512; r06, the shape question of the r05 verdict): the output product's FLOPs as TWO v_mfma_f32_16x16x32_bf16
// on two four-register accumulators - what it would issue in that shape, with none of the layout work a port needs (results wrong by
// construction).  The accumulators are declared as quads in that build: carving four registers out of a 16-register tuple per MFMA made
// hipcc copy the tuple around every statement (6874 v_accvgpr_* and scratch in the first try).
__device__ __forceinline__ void mfma_o4(f32x4& lo, f32x4& hi, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %2, %3, %0\n\tv_mfma_f32_16x16x32_bf16 %1, %2, %3, %1" : "+a"(lo), "+a"(hi) : "v"(a), "v"(b));
}
constexpr bool kOQuads = (CM3P_GABL & 512) != 0;
#define CM3P_MFMA_O(u, db, a, b)                                                  \
    do {                                                                          \
        if constexpr (kOQuads) mfma_o4(oq[u][db][0], oq[u][db][1], a, b);         \
        else mfma_o(oacc[u][db], a, b);                                           \
    } while (0)
// Pins a value where it is computed.  The cold blocks (mask, reference move) split a period into several basic blocks, and hipcc's
// sinking pass moves pure VALU work across sched_barriers into the block of its first use - the exponentials of a sub-block ended
// up in front of the PV MFMAs of the NEXT period, all 16 of a chunk pair in a row.  A volatile asm statement cannot be crossed.
#define CM3P_PIN(x) asm volatile("" ::"v"(x))  // (input only: an asm OUTPUT would cost a wait state in front of the next reader)
__device__ __forceinline__ float vmax3(float a, float b, float c) {
    float r = max3(a, b, c);
    CM3P_PIN(r);
    return r;
}

template <int U, bool MASK>
__global__ __launch_bounds__(256, 1) void attn_fwd_g_kernel(const uint16_t* __restrict__ qkv, uint16_t* __restrict__ out, float* __restrict__ lse,
                                                            const uint8_t* __restrict__ kmask, int Smax, int nh, VarLen vl) {
    static_assert(U == 2 || U == 4, "two S buffers alternate with the parity of the query block");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int QW = 32 * U, QB = 4 * QW;
    constexpr int STG = MASK ? kGStageMask : kGStageNoMask;
    const int tid = threadIdx.x, lane = tid & 63, hh = lane >> 5, l31 = lane & 31, g4 = lane >> 4, i16 = lane & 15;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    int qblk, head, b;
    decode_block((Smax + QB - 1) / QB, nh, qblk, head, b);
    const int Q0 = qblk * QB;
    const SeqView sv(vl, b, head, Smax, nh);
    const int S = sv.S;
    if (Q0 >= S) return;  // (unpadded batches: the grid is sized for the longest sequence)
    const int q0 = Q0 + wid * QW;
    const int64_t ld = (int64_t)3 * nh * 64;
    const uint16_t* qbase = qkv + sv.row0 * ld + head * 64;
    const uint16_t* kbase = qbase + nh * 64;
    const uint16_t* vbase = qbase + 2 * nh * 64;
    const int NT = (S + 63) / 64;
#if CM3P_GTRACE
    unsigned long long gt_acc[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long gt_prev = __builtin_amdgcn_s_memtime();
    const unsigned long long gt_rt0 = __builtin_amdgcn_s_memrealtime();  // (100 MHz: with the cycle total, the clock this wave ran at)
#endif
    // The sweep's two parts (see the loops below): [0, t_fast) tiles whose keys are all visible, t_fast a multiple of the ring; [t_fast,
    // t_end) the rest up to the tile of the last visible key.  With a key mask the workgroup scans its batch row's S bytes once.
    int first_bad = S % 64 ? S / 64 : NT, t_end = NT;
    if constexpr (MASK) {
        const uint8_t* kmr = kmask + sv.row0;
        int fz = INT_MAX, lnz = -1;  // first zero byte, last non-zero byte of this thread's share (16 consecutive bytes per 4 KiB of the row;
        for (int base = 0; base < S; base += 4096) {  // sixteen independent byte loads per trip: one round trip, not sixteen)
            uint8_t v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = kmr[min(base + tid * 16 + j, S - 1)];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int i = base + tid * 16 + j;
                if (i < S) {
                    fz = (v[j] == 0 && i < fz) ? i : fz;
                    lnz = v[j] != 0 ? i : lnz;
                }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            fz = min(fz, __shfl_xor(fz, o, 64));
            lnz = max(lnz, __shfl_xor(lnz, o, 64));
        }
        int* red = reinterpret_cast<int*>(smem + kGSlots * kGStageMask);  // 8 ints behind the ring
        if (lane == 0) {
            red[wid] = fz;
            red[4 + wid] = lnz;
        }
        __syncthreads();
        fz = min(min(red[0], red[1]), min(red[2], red[3]));
        lnz = max(max(red[4], red[5]), max(red[6], red[7]));
        first_bad = min(first_bad, fz / 64);  // (INT_MAX / 64 is far past NT)
        t_end = lnz < 0 ? 0 : lnz / 64 + 1;
    }
    const int t_fast = min(first_bad, t_end) & ~3;

    // ---- tile DMA: wave w brings rows 16 w .. 16 w + 15 of K and of V (two 1-KiB pieces each) and, with a mask, its own copy of the
    // tile's 64 validity bytes as dwords.  Rows are clamped to the sequence, so tiles past it (the ring runs three ahead) read valid memory.
    const TileDma dma(wid, lane);
    const int ldb = (int)ld * 2;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const uint32_t m0_k = __builtin_amdgcn_readfirstlane(lds0 + 2048u * wid);
    const uint32_t m0_m = __builtin_amdgcn_readfirstlane(lds0 + 16384u + 256u * wid);
    const uint8_t* km = MASK ? kmask + sv.row0 : nullptr;
    auto dma_k = [&](int t) {
        if constexpr ((CM3P_GABL & 8) != 0) return;
        dma.rows(m0_k + (uint32_t)(t & (kGSlots - 1)) * STG, kbase, ldb, t * 64, S);
    };
    auto dma_v = [&](int t) {
        if constexpr ((CM3P_GABL & 8) != 0) return;
        const uint32_t slot = (uint32_t)(t & (kGSlots - 1)) * STG;
        dma.rows(m0_k + slot + 8192u, vbase, ldb, t * 64, S, CM3P_AUD_T1);
        if constexpr (MASK) dma_ubyte64(m0_m + slot, km, (uint32_t)min(t * 64 + lane, S - 1));
    };
    auto dma_tile_now = [&](int t) {  // (prologue: not subject to the ablation switch)
        const uint32_t slot = (uint32_t)(t & (kGSlots - 1)) * STG;
        dma.rows(m0_k + slot, kbase, ldb, t * 64, S);
        dma.rows(m0_k + slot + 8192u, vbase, ldb, t * 64, S, CM3P_AUD_T1);
        if constexpr (MASK) dma_ubyte64(m0_m + slot, km, (uint32_t)min(t * 64 + lane, S - 1));
    };
    dma_tile_now(0);
    dma_tile_now(1);
    dma_tile_now(2);

    // ---- the wave's queries: B operands of the score product (AGPRs)
    bf16x8 qf[U][4];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int qc = min(q0 + 32 * u + l31, S - 1);
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[u][s] = *reinterpret_cast<const bf16x8*>(qbase + (int64_t)qc * ld + 16 * s + 8 * hh);
    }

    // ---- per-lane LDS byte offsets (slot base and k-step / block strides are added as scalars / immediates)
    int oR[4], oTlo[2], oThi[2];
#pragma unroll
    for (int s = 0; s < 4; ++s) oR[s] = l31 * 128 + (((2 * s + hh) ^ swz(l31)) << 4);
#pragma unroll
    for (int db = 0; db < 2; ++db) {
        const int row = 4 * hh + (i16 >> 2), col = 32 * db + 16 * (g4 & 1) + 4 * (i16 & 3);
        oTlo[db] = 8192 + off_T(row, col);
        oThi[db] = 8192 + off_T(row + 8, col);
    }

    // (opaque to the compiler from here on: it would otherwise re-associate offset + slot + k-step constants into dozens of live addresses)
#pragma unroll
    for (int s = 0; s < 4; ++s) asm volatile("" : "+v"(oR[s]));
    asm volatile("" : "+v"(oTlo[0]), "+v"(oTlo[1]), "+v"(oThi[0]), "+v"(oThi[1]));

    // ---- state.  Softmax state per query = per lane, one copy per half-wave (a half holds 32 of a tile's 64 keys): ref is the
    // reference point in log2 units that every p, l and O is relative to, negref its negative splat over the 16 registers of an
    // MFMA C operand, thr the amount a tile maximum may exceed it before it is moved (-inf until the row has seen a visible key:
    // the first one sets the reference exactly), lA / lB two partial row sums.
    f32x16 oacc[kOQuads ? 1 : U][2], negref[U];  // (the timing build keeps its accumulators in oq and never touches oacc in the sweep)
    f32x4 oq[kOQuads ? U : 1][2][2];  // (timing build only: CM3P_MFMA_O)
    float lA[U], lB[U], ref[U], thr[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if constexpr (kOQuads) oq[u][0][0] = oq[u][0][1] = oq[u][1][0] = oq[u][1][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            negref[u][i] = 0.f;
            if constexpr (!kOQuads) oacc[u][0][i] = oacc[u][1][i] = 0.f;
        }
        lA[u] = lB[u] = ref[u] = 0.f;
        thr[u] = kNegInf;
    }
    f32x16 SA[2], SB[2];  // scores of the sub-blocks with even / odd query block
    cm3p_u32x4 Pw[4];     // P^T of the last finished sub-block, bf16 pairs: chunk c = keys 16 c .. 16 c + 15 of the tile
#pragma unroll
    for (int i = 0; i < 16; ++i) SB[0][i] = SB[1][i] = 0.f;  // (period 0 runs the tail of "sub-block -1": adds zeros)
#pragma unroll
    for (int c = 0; c < 4; ++c) Pw[c] = cm3p_u32x4{0u, 0u, 0u, 0u};  // (... and its PV: adds zero)
    if constexpr ((CM3P_GABL & 128) != 0) {  // (timing only: the packs are gone, P is a constant - NOT zero: a zero operand lowers the matrix pipe's power and raises the clock)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int i = 0; i < 4; ++i) Pw[c][i] = 0x3e003d80u + 0x00010001u * ((lane * 37u + c * 11u + i * 5u) & 127u);
    }
    bf16x8 Kf[2][4], Vf[4][2];

    auto load_k = [&](const char* st, auto blk_c, auto s_c) {
        constexpr int BLK = decltype(blk_c)::value, SS = decltype(s_c)::value;
        Kf[BLK][SS] = ld_frag(st + 4096 * BLK + oR[SS]);
    };
    auto load_v = [&](const char* st, auto s_c, auto d_c) {
        constexpr int SS = decltype(s_c)::value, D = decltype(d_c)::value;
        Vf[SS][D] = ld_fragT(st + 2048 * SS + oTlo[D], st + 2048 * SS + oThi[D]);
    };

    // the Q rows have landed (and, vmcnt being in order, the first three tiles of this wave); after the barrier: of every wave.
    // Through the builtin, so that hipcc KNOWS its Q loads are complete: with an asm wait it placed its own s_waitcnt vmcnt(11 .. 0) in
    // front of the first use of each Q fragment inside the loop - every trip, draining the DMA ring (vmcnt retires in order).
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), expcnt / lgkmcnt unconstrained (gfx9 encoding)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    {
        const char* st = smem;
        load_k(st, CM3P_IC(0), CM3P_IC(0)); load_k(st, CM3P_IC(0), CM3P_IC(1)); load_k(st, CM3P_IC(0), CM3P_IC(2)); load_k(st, CM3P_IC(0), CM3P_IC(3));
        load_k(st, CM3P_IC(1), CM3P_IC(0)); load_k(st, CM3P_IC(1), CM3P_IC(1)); load_k(st, CM3P_IC(1), CM3P_IC(2)); load_k(st, CM3P_IC(1), CM3P_IC(3));
        load_v(st, CM3P_IC(0), CM3P_IC(0)); load_v(st, CM3P_IC(0), CM3P_IC(1)); load_v(st, CM3P_IC(1), CM3P_IC(0)); load_v(st, CM3P_IC(1), CM3P_IC(1));
        load_v(st, CM3P_IC(2), CM3P_IC(0)); load_v(st, CM3P_IC(2), CM3P_IC(1)); load_v(st, CM3P_IC(3), CM3P_IC(0)); load_v(st, CM3P_IC(3), CM3P_IC(1));
    }
    // QK^T of sub-block 0
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
        mfma_s0(SA[blk], Kf[blk][0], qf[0][0], negref[0]);
        mfma_s(SA[blk], Kf[blk][1], qf[0][1]);
        mfma_s(SA[blk], Kf[blk][2], qf[0][2]);
        mfma_s(SA[blk], Kf[blk][3], qf[0][3]);
    }

    // ---- pieces of a period -----------------------------------------------------------------------------------------------------
    // chunk c of a score buffer = registers 8 (c & 1) .. + 7 of block c >> 1 = keys 16 c .. 16 c + 15 (this half-wave's 8 of them)
    auto E = [&](f32x16 (&Sx)[2], int c, int i) {  // one exponential, in place
        if constexpr ((CM3P_GABL & 1) != 0) return;
        float x = __builtin_amdgcn_exp2f(Sx[c >> 1][8 * (c & 1) + i]);
        CM3P_PIN(x);
        Sx[c >> 1][8 * (c & 1) + i] = x;
    };
    auto A = [&](f32x16 (&Sx)[2], int c, int i, float& l) {  // one row-sum add
        if constexpr ((CM3P_GABL & 4) != 0) return;
        l += Sx[c >> 1][8 * (c & 1) + i];
        CM3P_PIN(l);
    };
    auto C2 = [&](f32x16 (&Sx)[2], int c, int h) {  // two packs: elements 4 h .. 4 h + 3 of chunk c
        if constexpr ((CM3P_GABL & 128) != 0) return;
        const f32x16& b = Sx[c >> 1];
        const int r = 8 * (c & 1) + 4 * h;
        uint32_t w0 = pack_bf16x2(b[r], b[r + 1]), w1 = pack_bf16x2(b[r + 2], b[r + 3]);
        CM3P_PIN(w0);
        CM3P_PIN(w1);
        Pw[c][2 * h] = w0;
        Pw[c][2 * h + 1] = w1;
    };
    // keys this lane may not see -> -inf (cold: tiles with padding or past the sequence).  w: validity bits of the block's 32 keys as this
    // half-wave numbers them (bit 8 g + r = accumulator register 4 g + r)
    auto mask_blk = [&](f32x16& s, uint32_t w) {
        asm volatile("" : "+v"(w));  // (keeps the sixteen bit tests inside the cold block: hipcc hoisted them into every period)
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = ((w >> (8 * (i >> 2) + (i & 3))) & 1u) ? s[i] : kNegInf;
    };

    // One period: softmax of sub-block (tile t, query block UU) in Sc, PV of the sub-block before it, QK^T of the one after it into So.
    //   stV   slot of tile t (UU == 0: the V fragments are replaced by tile t's behind the MFMAs that read tile t-1's)
    //   stKn  slot of tile t+1 (UU == U-2: the K fragments are replaced behind the last MFMAs that read tile t's)
    auto period = [&](auto u_c, auto slot_c, auto fast_c, f32x16 (&Sc)[2], f32x16 (&So)[2], int tile_bad, uint32_t w0, uint32_t w1, int t) {
        constexpr bool FAST = decltype(fast_c)::value != 0;  // (the sweep's first part: every key of every tile visible, no masking code at all)
        constexpr int UU = decltype(u_c)::value, UP = (UU + U - 1) % U, UN = (UU + 1) % U;
        constexpr int SLOT = decltype(slot_c)::value;
        constexpr bool RV = UU == 0, RK = UU == U - 2;
        const char* const stV = smem + SLOT * STG;                        // (compile-time slot: every fragment read is register + immediate)
        const char* const stKn = smem + ((SLOT + 1) & (kGSlots - 1)) * STG;
        float ma, mb;
        // gap 0
        CM3P_SB();
        CM3P_MFMA_O(UP, 0, Vf[0][0], __builtin_bit_cast(bf16x8, Pw[0]));
        if constexpr (RV && (CM3P_GABL & 64) == 0) load_v(stV, CM3P_IC(0), CM3P_IC(0));
        A(So, 3, 6, lA[UP]);  // tail of the sub-block before: its last chunk's row sums fill the light gaps 0-3
        A(So, 3, 7, lB[UP]);
        C2(So, 3, 1);
        if constexpr (!FAST)
            if (__builtin_expect(tile_bad != 0, 0)) mask_blk(Sc[0], w0);
        if constexpr ((CM3P_GABL & 2) == 0) {
            ma = vmax3(Sc[0][0], Sc[0][1], Sc[0][2]);
            mb = vmax3(Sc[0][3], Sc[0][4], Sc[0][5]);
            ma = vmax3(ma, Sc[0][6], Sc[0][7]);
            mb = vmax3(mb, Sc[0][8], Sc[0][9]);
            ma = vmax3(ma, Sc[0][10], Sc[0][11]);
        }
        GT2(0);
        // gap 1
        CM3P_SB();
        CM3P_MFMA_O(UP, 1, Vf[0][1], __builtin_bit_cast(bf16x8, Pw[0]));
        if constexpr (RV && (CM3P_GABL & 64) == 0) load_v(stV, CM3P_IC(0), CM3P_IC(1));
        A(So, 3, 0, lA[UP]);
        A(So, 3, 1, lB[UP]);
        if constexpr (!FAST)
            if (__builtin_expect(tile_bad != 0, 0)) mask_blk(Sc[1], w1);  // (two MFMAs behind the last one of this block's score chain)
        if constexpr ((CM3P_GABL & 2) == 0) {
            mb = vmax3(mb, Sc[0][12], Sc[0][13]);
            ma = vmax3(ma, Sc[0][14], Sc[0][15]);
            mb = vmax3(mb, Sc[1][0], Sc[1][1]);
            ma = vmax3(ma, Sc[1][2], Sc[1][3]);
        }
        GT2(1);
        // gap 2
        CM3P_SB();
        CM3P_MFMA_O(UP, 0, Vf[1][0], __builtin_bit_cast(bf16x8, Pw[1]));
        if constexpr (RV && (CM3P_GABL & 64) == 0) load_v(stV, CM3P_IC(1), CM3P_IC(0));
        if constexpr ((CM3P_GABL & 2) == 0) {
            mb = vmax3(mb, Sc[1][4], Sc[1][5]);
            ma = vmax3(ma, Sc[1][6], Sc[1][7]);
            mb = vmax3(mb, Sc[1][8], Sc[1][9]);
            ma = vmax3(ma, Sc[1][10], Sc[1][11]);
            mb = vmax3(mb, Sc[1][12], Sc[1][13]);
        }
        GT2(2);
        // gap 3: the last of the maximum, the wave-uniform decision, THEN four row-sum adds, then the branch: a branch right behind the
        // compare that feeds it waits out the VALU -> scalar round trip with nothing to issue
        CM3P_SB();
        CM3P_MFMA_O(UP, 1, Vf[1][1], __builtin_bit_cast(bf16x8, Pw[1]));
        if constexpr (RV && (CM3P_GABL & 64) == 0) load_v(stV, CM3P_IC(1), CM3P_IC(1));
        if constexpr ((CM3P_GABL & 2) == 0) {
            ma = vmax3(ma, Sc[1][14], Sc[1][15]);
            const float mt = vmax3(ma, mb, mb);  // this half-wave's tile maximum, relative to the reference point
            const unsigned long long any_move = __ballot(mt > thr[UU]);
            asm volatile("" ::"s"(any_move));
            CM3P_SB();  // (the compare stays in front of the adds: pure VALU work floats across a pin, not across this)
            A(So, 3, 2, lA[UP]);
            A(So, 3, 3, lB[UP]);
            A(So, 3, 4, lA[UP]);
            A(So, 3, 5, lB[UP]);
            CM3P_SB();
            GT2(3);
            if (((CM3P_GABL & 32) == 0) && __builtin_expect(any_move != 0ull, 0)) {  // rare after the first tiles: move the reference point (both halves of a query by the same amount)
                // Per QUERY: only a row whose own maximum asks for it moves (both half-waves of a query see the same mf and thr).  A row's
                // result must not depend on what the other lanes of its wave hold - the rows of a padded batch that lie past a
                // sequence's end are other data than the clamped rows of the packed batch, and a move they trigger must not shift
                // the valid rows' reference (packed == padded bit for bit: test_attention_varlen_equals_padded_on_valid_rows).
                const float mf = fmaxf(mt, __shfl_xor(mt, 32, 64));
                const bool had = thr[UU] > 0.f;
                const bool move = mf > thr[UU];  // (had: mf > 2^kGDefer above the reference; !had: the row's first visible key)
                const float shift = move ? mf : 0.f;
                const float alpha = (move && had) ? __builtin_amdgcn_exp2f(-shift) : 1.0f;  // O = l = 0 before the first visible key
                thr[UU] = move ? kGDefer : thr[UU];
                ref[UU] += shift;
                lA[UU] *= alpha;
                lB[UU] *= alpha;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    Sc[0][i] -= shift;
                    Sc[1][i] -= shift;
                }
                // O *= alpha and the -ref splat, one register at a time through one temporary: written as plain C++ the block
                // wants 64 VGPRs at once, and what the allocator spills to make room is reloaded in the hot path
                const float nr = -ref[UU];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    if constexpr (kOQuads) {  // (timing build: the rare block only rewrites the -ref splat)
                        negref[UU][i] = nr;
                        continue;
                    }
                    constexpr int UO = kOQuads ? 0 : UU;
                    float o0 = oacc[UO][0][i], o1 = oacc[UO][1][i], nv = negref[UU][i], tmp;
                    asm volatile("v_accvgpr_read_b32 %3, %0\n\tv_mov_b32 %2, %5\n\tv_mul_f32 %3, %3, %4\n\tv_accvgpr_write_b32 %0, %3\n\t"
                                 "v_accvgpr_read_b32 %3, %1\n\ts_nop 0\n\tv_mul_f32 %3, %3, %4\n\tv_accvgpr_write_b32 %1, %3"
                                 : "+a"(o0), "+a"(o1), "=&v"(nv), "=&v"(tmp)
                                 : "v"(alpha), "v"(nr));
                    oacc[UO][0][i] = o0;
                    oacc[UO][1][i] = o1;
                    negref[UU][i] = nv;
                }
            }
        } else {
            A(So, 3, 2, lA[UP]);
            A(So, 3, 3, lB[UP]);
            A(So, 3, 4, lA[UP]);
            A(So, 3, 5, lB[UP]);
        }
        GT1(1);
        GT2(4);
        // gaps 4-6: chunk 0
        CM3P_SB();
        CM3P_MFMA_O(UP, 0, Vf[2][0], __builtin_bit_cast(bf16x8, Pw[2]));
        if constexpr (RV && (CM3P_GABL & 64) == 0) load_v(stV, CM3P_IC(2), CM3P_IC(0));
        E(Sc, 0, 0); E(Sc, 0, 1); E(Sc, 0, 2);
        CM3P_SB();
        CM3P_MFMA_O(UP, 1, Vf[2][1], __builtin_bit_cast(bf16x8, Pw[2]));
        if constexpr (RV && (CM3P_GABL & 64) == 0) load_v(stV, CM3P_IC(2), CM3P_IC(1));
        E(Sc, 0, 3); E(Sc, 0, 4); E(Sc, 0, 5);
        A(Sc, 0, 0, lA[UU]); A(Sc, 0, 1, lB[UU]); A(Sc, 0, 2, lA[UU]);
        if constexpr (UU == (U == 2 ? 0 : 1)) dma_k(t + 3);
        CM3P_SB();
        CM3P_MFMA_O(UP, 0, Vf[3][0], __builtin_bit_cast(bf16x8, Pw[3]));
        if constexpr (RV && (CM3P_GABL & 64) == 0) load_v(stV, CM3P_IC(3), CM3P_IC(0));
        E(Sc, 0, 6); E(Sc, 0, 7);
        A(Sc, 0, 3, lB[UU]); A(Sc, 0, 4, lA[UU]); A(Sc, 0, 5, lB[UU]);
        C2(Sc, 0, 0);
        // gaps 7-9: chunk 1 (gap 7 is the last PV MFMA: Pw[3] of the sub-block before is free from here on)
        CM3P_SB();
        CM3P_MFMA_O(UP, 1, Vf[3][1], __builtin_bit_cast(bf16x8, Pw[3]));
        if constexpr (RV && (CM3P_GABL & 64) == 0) load_v(stV, CM3P_IC(3), CM3P_IC(1));
        E(Sc, 1, 0); E(Sc, 1, 1); E(Sc, 1, 2);
        A(Sc, 0, 6, lA[UU]); A(Sc, 0, 7, lB[UU]);
        C2(Sc, 0, 1);
        GT1(2);
        CM3P_SB();
        mfma_s0(So[0], Kf[0][0], qf[UN][0], negref[UN]);
        if constexpr (RK && (CM3P_GABL & 64) == 0) load_k(stKn, CM3P_IC(0), CM3P_IC(0));
        E(Sc, 1, 3); E(Sc, 1, 4); E(Sc, 1, 5);
        A(Sc, 1, 0, lA[UU]); A(Sc, 1, 1, lB[UU]); A(Sc, 1, 2, lA[UU]);
        CM3P_SB();
        mfma_s(So[0], Kf[0][1], qf[UN][1]);
        if constexpr (RK && (CM3P_GABL & 64) == 0) load_k(stKn, CM3P_IC(0), CM3P_IC(1));
        E(Sc, 1, 6); E(Sc, 1, 7);
        A(Sc, 1, 3, lB[UU]); A(Sc, 1, 4, lA[UU]); A(Sc, 1, 5, lB[UU]);
        C2(Sc, 1, 0);
        // gaps 10-12: chunk 2
        CM3P_SB();
        mfma_s(So[0], Kf[0][2], qf[UN][2]);
        if constexpr (RK && (CM3P_GABL & 64) == 0) load_k(stKn, CM3P_IC(0), CM3P_IC(2));
        E(Sc, 2, 0); E(Sc, 2, 1); E(Sc, 2, 2);
        A(Sc, 1, 6, lA[UU]); A(Sc, 1, 7, lB[UU]);
        C2(Sc, 1, 1);
        CM3P_SB();
        mfma_s(So[0], Kf[0][3], qf[UN][3]);
        if constexpr (RK && (CM3P_GABL & 64) == 0) load_k(stKn, CM3P_IC(0), CM3P_IC(3));
        E(Sc, 2, 3); E(Sc, 2, 4); E(Sc, 2, 5);
        A(Sc, 2, 0, lA[UU]); A(Sc, 2, 1, lB[UU]); A(Sc, 2, 2, lA[UU]);
        if constexpr (UU == U - 1) dma_v(t + 3);
        GT1(3);
        CM3P_SB();
        mfma_s0(So[1], Kf[1][0], qf[UN][0], negref[UN]);
        if constexpr (RK && (CM3P_GABL & 64) == 0) load_k(stKn, CM3P_IC(1), CM3P_IC(0));
        E(Sc, 2, 6); E(Sc, 2, 7);
        A(Sc, 2, 3, lB[UU]); A(Sc, 2, 4, lA[UU]); A(Sc, 2, 5, lB[UU]);
        C2(Sc, 2, 0);
        // gaps 13-15: chunk 3 (its last two adds and packs run in gap 0 of the next period)
        CM3P_SB();
        mfma_s(So[1], Kf[1][1], qf[UN][1]);
        if constexpr (RK && (CM3P_GABL & 64) == 0) load_k(stKn, CM3P_IC(1), CM3P_IC(1));
        E(Sc, 3, 0); E(Sc, 3, 1); E(Sc, 3, 2);
        A(Sc, 2, 6, lA[UU]); A(Sc, 2, 7, lB[UU]);
        C2(Sc, 2, 1);
        CM3P_SB();
        mfma_s(So[1], Kf[1][2], qf[UN][2]);
        if constexpr (RK && (CM3P_GABL & 64) == 0) load_k(stKn, CM3P_IC(1), CM3P_IC(2));
        E(Sc, 3, 3); E(Sc, 3, 4); E(Sc, 3, 5);
        CM3P_SB();
        mfma_s(So[1], Kf[1][3], qf[UN][3]);
        if constexpr (RK && (CM3P_GABL & 64) == 0) load_k(stKn, CM3P_IC(1), CM3P_IC(3));
        E(Sc, 3, 6); E(Sc, 3, 7);
        C2(Sc, 3, 0);
        CM3P_SB();
        GT1(4);
        GT2(5);
    };

    // ---- the key sweep, four tiles (= the ring) per trip so that a tile's slot is a compile-time constant ---------------------------
    auto tile = [&](auto slot_c, auto fast_c, int t) {
        constexpr int SLOT = decltype(slot_c)::value;
        constexpr bool FAST = decltype(fast_c)::value != 0;
        // tiles <= t + 1 of this wave have landed (only tile t + 2 may be in flight); behind the barrier: of every wave, and every
        // wave has finished with the slot of tile t - 1, which the DMAs of tile t + 3 overwrite
        if constexpr (MASK) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        if constexpr ((CM3P_GABL & 16) == 0) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        int tile_bad = 0;
        uint32_t w0 = ~0u, w1 = ~0u;
        if constexpr (!FAST) {
            // validity of the tile's keys: wave-uniform 64 bits (bit k = key 64 t + k may be seen)
            unsigned long long valid = ~0ull;
            if constexpr (MASK) valid = __ballot(*reinterpret_cast<const uint32_t*>(smem + SLOT * STG + 16384 + 256 * wid + 4 * lane) != 0u);
            const int rem = S - t * 64;
            if (rem < 64) valid = rem > 0 ? valid & ((1ull << rem) - 1ull) : 0ull;
            // (a scalar: as a bool the compiler re-derived the branch condition through a v_cndmask / v_cmp pair in every period)
            tile_bad = __builtin_amdgcn_readfirstlane(valid == ~0ull ? 0 : 1);
            const unsigned long long vsh = valid >> (4 * hh);
            w0 = (uint32_t)vsh, w1 = (uint32_t)(vsh >> 32);
        }
        GT1(0);
        GT2(5);
        if constexpr (U == 4) {
            period(CM3P_IC(0), slot_c, fast_c, SA, SB, tile_bad, w0, w1, t);
            period(CM3P_IC(1), slot_c, fast_c, SB, SA, tile_bad, w0, w1, t);
            period(CM3P_IC(2), slot_c, fast_c, SA, SB, tile_bad, w0, w1, t);
            period(CM3P_IC(3), slot_c, fast_c, SB, SA, tile_bad, w0, w1, t);
        } else {
            period(CM3P_IC(0), slot_c, fast_c, SA, SB, tile_bad, w0, w1, t);
            period(CM3P_IC(1), slot_c, fast_c, SB, SA, tile_bad, w0, w1, t);
        }
    };
    // Two loops, one after the other: the tiles in front of the first key some lane may not see run a stream WITHOUT masking code (a
    // not-taken branch in front of each block's maximum costs 4 %: two per period; timing-only build without them 3.00 -> 2.88 ms at
    // C4), the rest - from the ring-aligned tile of the first invisible key to the tile of the last visible one - the stream with one
    // branch per block.  Tiles behind the last visible key are not visited at all: every score of theirs would be -inf, p = 0.
    // (Both copies inside ONE loop behind an if / else per tile made hipcc spill 1500 registers: the copies must not meet in a phi.)
    GT_STAMP(5);
    int t = 0;
    for (; t < t_fast; t += 4) {
        tile(CM3P_IC(0), CM3P_IC(1), t);
        tile(CM3P_IC(1), CM3P_IC(1), t + 1);
        tile(CM3P_IC(2), CM3P_IC(1), t + 2);
        tile(CM3P_IC(3), CM3P_IC(1), t + 3);
    }
    // (tiles past t_end - the trip count is rounded up to the ring - have no visible key: every score of theirs becomes -inf)
    for (; t < t_end; t += 4) {
        tile(CM3P_IC(0), CM3P_IC(0), t);
        tile(CM3P_IC(1), CM3P_IC(0), t + 1);
        tile(CM3P_IC(2), CM3P_IC(0), t + 2);
        tile(CM3P_IC(3), CM3P_IC(0), t + 3);
    }

    // ---- drain: the tail and the PV of the last sub-block (query block U-1; its scores are in SB)
    A(SB, 3, 6, lA[U - 1]);
    A(SB, 3, 7, lB[U - 1]);
    C2(SB, 3, 1);
    A(SB, 3, 0, lA[U - 1]);
    A(SB, 3, 1, lB[U - 1]);
    A(SB, 3, 2, lA[U - 1]);
    A(SB, 3, 3, lB[U - 1]);
    A(SB, 3, 4, lA[U - 1]);
    A(SB, 3, 5, lB[U - 1]);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        CM3P_MFMA_O(U - 1, 0, Vf[s][0], __builtin_bit_cast(bf16x8, Pw[s]));
        CM3P_MFMA_O(U - 1, 1, Vf[s][1], __builtin_bit_cast(bf16x8, Pw[s]));
    }
    // no DMA may be in flight when the ring becomes the transposition buffers (or when the workgroup's LDS is handed on)
    dma_wait_barrier(0);

#pragma unroll
    for (int u = 0; u < U; ++u) {
        const float lh = lA[u] + lB[u];
        const float l_tot = lh + __shfl_xor(lh, 32, 64);
        const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
        const int qrow = q0 + 32 * u + l31;
        if constexpr (kOQuads) {
            f32x16 t0, t1;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                t0[i] = oq[u][0][(i >> 2) & 1][i & 3];
                t1[i] = oq[u][1][(i >> 2) & 1][i & 3];
            }
            store_rows32(smem + 4608 * wid, t0, t1, inv, out + (sv.row0 + q0 + 32 * u) * nh * 64 + head * 64, (int64_t)nh * 64, S - (q0 + 32 * u), lane);
        } else {
            store_rows32(smem + 4608 * wid, oacc[kOQuads ? 0 : u][0], oacc[kOQuads ? 0 : u][1], inv, out + (sv.row0 + q0 + 32 * u) * nh * 64 + head * 64,
                         (int64_t)nh * 64, S - (q0 + 32 * u), lane);
        }
        if (qrow < S && hh == 0)
            lse[sv.stat0 + qrow] = l_tot > 0.f ? (ref[u] + __log2f(l_tot)) * 0.69314718055994531f : __builtin_huge_valf();
    }
#if CM3P_GTRACE
    GT_STAMP(5);
    if (g_fwd_trace && lane == 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) g_fwd_trace[((size_t)blockIdx.x * 4 + wid) * 8 + k] = gt_acc[k];
        g_fwd_trace[((size_t)blockIdx.x * 4 + wid) * 8 + 6] = (unsigned long long)((t_end + 3) & ~3);
        g_fwd_trace[((size_t)blockIdx.x * 4 + wid) * 8 + 7] = __builtin_amdgcn_s_memrealtime() - gt_rt0;
    }
#endif
}

}  // namespace

// attention.hip: launch_attn_fwd routes the global layers (window < 0) with pre-scaled q here
int cm3p_launch_attn_fwd_global(const void* qkv, void* out, float* lse, const uint8_t* key_mask, int B, int S, int nh, const int* cu_seqlens,
                                int64_t total, hipStream_t s) {
#ifndef CM3P_FWD_U
#define CM3P_FWD_U 4
#endif
    constexpr int U = CM3P_FWD_U;
    const VarLen vl{cu_seqlens, total};
    const dim3 grid(((S + 128 * U - 1) / (128 * U)) * nh * B);  // 1-D: decode_block() maps it XCD-aware
    static Cm3pDevOnce once;  // (per device: common.h)
    const int rc_once = once.run([] {
        return cm3p_set_max_lds({reinterpret_cast<const void*>(&attn_fwd_g_kernel<U, true>), reinterpret_cast<const void*>(&attn_fwd_g_kernel<U, false>)},
                                kGSlots * kGStageMask + 32);
    });
    if (rc_once != CM3P_OK) return rc_once;
    if (key_mask)
        attn_fwd_g_kernel<U, true><<<grid, 256, kGSlots * kGStageMask + 32, s>>>((const uint16_t*)qkv, (uint16_t*)out, lse, key_mask, S, nh, vl);
    else
        attn_fwd_g_kernel<U, false><<<grid, 256, kGSlots * kGStageNoMask, s>>>((const uint16_t*)qkv, (uint16_t*)out, lse, nullptr, S, nh, vl);
    if (hipGetLastError() != hipSuccess) return CM3P_ERR_LAUNCH;
    return CM3P_OK;
}

int cm3p_ablation_flags_attention_fwd() { return (CM3P_GABL) | ((CM3P_GTRACE) << 8); }
#if CM3P_GTRACE
extern "C" int cm3p_debug_set_fwd_trace(void* buf) {
    unsigned long long* p = static_cast<unsigned long long*>(buf);
    return hipMemcpyToSymbol(HIP_SYMBOL(g_fwd_trace), &p, sizeof(p)) == hipSuccess ? 0 : 1;
}
#endif
#if CM3P_DMA_AUDIT
int cm3p_audit_set_attention_fwd(void* buf) { return cm3p_audit_set_local(buf); }
#endif
