// Large-tile bf16 MFMA GEMM: 256 x 256 x 64 per 512-thread workgroup, operands streamed global -> LDS by LDS-DMA
// (global_load_lds_dwordx4, no VGPR staging, no ds_write), double-buffered, one barrier per k-step.
//
// Same contract as gemm.hip (C[m,n] = sum_k A(m,k) B(n,k) (+R); a_kc / b_kc layouts; epilogues; split-K) for the
// shapes that dominate the step: K a multiple of 64.  The 128 x 128 kernel in gemm.hip keeps every other shape.
// Why a second kernel: at 128 x 128 a workgroup moves 1/64 byte per flop from L2 - more than the chip's L2 can feed at
// MFMA speed; 256 x 256 halves that, and LDS-DMA frees ~64 VGPRs and all staging instructions.
//
// LDS image per stage (64 KiB): A then B, 32 KiB each.
//   k-contiguous operand: [256 rows][64 k] bf16, 128-byte rows, 16-byte chunk index XOR (row & 7).
//   k-strided operand:    [64 k][256 idx] bf16, 512-byte rows, 32-byte segment index XOR f(k), f(k) = (k&3) | ((k>>3)&1)<<2.
// LDS-DMA writes lane-linearly (wave base + 16 * lane), so the swizzle is applied to each lane's SOURCE address and
// again when fragments are read (both sides or neither).
// 8 waves as 2 (m) x 4 (n); each wave owns 128 x 64 of C = 8 x 4 tiles of v_mfma_f32_16x16x32_bf16 (128 accumulator VGPRs).
#include "common.h"

namespace {

constexpr int TM = 256, TN = 256, TK = 64;
// Wave layout: 2 waves along m x kWN along n.  kWN = 4 (shipped): eight waves of 128 x 64, two per SIMD.  kWN = 2 builds four
// waves of 128 x 128 (one per SIMD, 256 accumulator AGPRs; a k-step then reads 128 KiB of fragments from LDS instead of 192 KiB):
// correct, but 10-30 % slower in a one-call A/B - a lone wave per SIMD cannot hide its own LDS-DMA issue and read latency.
#ifndef CM3P_G256_WN
#define CM3P_G256_WN 4
#endif
constexpr int kWN = CM3P_G256_WN;
constexpr int kThreads = 128 * kWN;      // 512 or 256
constexpr int kNJ = 16 / kWN;            // 16-column MFMA tiles per wave: 4 or 8
constexpr int kWaveN = TN / kWN;         // 64 or 128
#ifndef CM3P_G256_ABL
#define CM3P_G256_ABL 0  // timing-only probes (results invalid): 1 no LDS-DMA of the B operand in the steady state, 2 no B fragment reads
#endif
constexpr int kPieces = 16 / kWN;        // 1-KiB LDS-DMA pieces per wave, operand and k-tile: 4 or 8
constexpr int kItems = 2048 / kThreads;  // 16-byte store items per thread and epilogue pass: 4 or 8
constexpr int kOperandBytes = TM * TK * 2;  // 32 KiB
constexpr int kStage = 2 * kOperandBytes;   // 64 KiB

__device__ __forceinline__ int kc_off(int r, int c) { return r * 128 + ((c ^ (r & 7)) << 4); }
__device__ __forceinline__ int ksf(int k) { return (k & 3) | (((k >> 3) & 1) << 2); }
__device__ __forceinline__ int ks_off(int k, int idx) { return k * 512 + (((idx >> 4) ^ ksf(k)) << 5) + ((idx & 15) << 1); }

// LDS-DMA of 64 x 16 bytes (lane l's 16 bytes land at lds_wave_base + 16 * l), issued as inline assembly on purpose.
// Through __builtin_amdgcn_global_load_lds the compiler knows the instruction writes LDS; unable to prove that the fragment
// reads of the CURRENT stage touch the other half of the buffer, it puts s_waitcnt vmcnt(0) in front of them - i.e. it
// drains the NEXT stage's loads right after they were issued, and the double buffer overlaps nothing (rocprof: the k-loop
// sat at 57 % MFMA-busy).  Ordering is explicit instead: the counted/zero vmcnt wait and the barrier that end a k-step.
__device__ __forceinline__ void glds16(const void* gsrc, char* lds_wave_base) {
    const uint32_t m0v = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds_wave_base);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(m0v) : "memory", "m0");
}

// Per-thread source pointers for the 4 one-KiB pieces this wave stages per operand per k-tile.
template <bool KC>
struct Stager {
    const uint16_t* src[kPieces];
    int64_t kstep;

    __device__ __forceinline__ void init(const uint16_t* base, int64_t ld, int64_t idx0, int64_t extent, int64_t kbeg, int wid,
                                         int lane) {
#pragma unroll
        for (int i = 0; i < kPieces; ++i) {
            const int q = wid * kPieces + i;  // piece index 0..31 inside the operand image
            if constexpr (KC) {
                const int r = q * 8 + (lane >> 3);            // row inside the tile
                const int c = (lane & 7) ^ (r & 7);           // which 16-byte chunk of the row lands at this lane's slot
                int64_t row = idx0 + r;
                if (row > extent - 1) row = extent - 1;       // clamp: rows past the edge are never stored
                src[i] = base + row * ld + kbeg + c * 8;
            } else {
                const int k = q * 2 + (lane >> 5);
                const int p = lane & 31;
                const int seg = (p >> 1) ^ ksf(k);
                int64_t col = idx0 + (seg * 2 + (p & 1)) * 8;
                if (col > extent - 8) col = extent - 8;       // clamp (extent % 8 == 0)
                src[i] = base + (kbeg + k) * ld + col;
            }
        }
        kstep = KC ? TK : TK * ld;
#pragma unroll
        for (int i = 0; i < kPieces; ++i) src[i] -= (i & 3) * 512;  // (elements) compensates the instruction offset of piece i
    }
    // one m0 write for the wave's four consecutive 1-KiB pieces: the instruction offset moves the LDS address (and the global
    // address, which is why src[i] is kept i KiB low - see init)
    __device__ __forceinline__ void issue(char* image, int wid, int audit_id = CM3P_AUD_A) {
#pragma unroll
        for (int i = 0; i < kPieces; ++i) CM3P_AUDIT(audit_id, src[i] + (i & 3) * 512, 16);  // (src[i] is kept (i & 3) KiB low, see init)
#pragma unroll
        for (int g4 = 0; g4 < kPieces / 4; ++g4) {
            const uint32_t m0v = __builtin_amdgcn_readfirstlane(
                (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(image + (wid * kPieces + g4 * 4) * 1024));
            asm volatile("s_mov_b32 m0, %4\n\ts_nop 0\n\t"
                         "global_load_lds_dwordx4 %0, off\n\t"
                         "global_load_lds_dwordx4 %1, off offset:1024\n\t"
                         "global_load_lds_dwordx4 %2, off offset:2048\n\t"
                         "global_load_lds_dwordx4 %3, off offset:3072" ::"v"(src[g4 * 4]), "v"(src[g4 * 4 + 1]), "v"(src[g4 * 4 + 2]),
                         "v"(src[g4 * 4 + 3]), "s"(m0v)
                         : "memory", "m0");
        }
#pragma unroll
        for (int i = 0; i < kPieces; ++i) src[i] += kstep;
    }
};

template <bool KC>
__device__ __forceinline__ bf16x8 frag(const char* image, int idx0, int kk, int lane) {
    if constexpr (KC) {
        const int r = idx0 + (lane & 15);
        return *reinterpret_cast<const bf16x8*>(image + kc_off(r, kk * 4 + (lane >> 4)));
    } else {
        const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
        const int k = kk * 32 + 8 * g + q;
        const bf16x4 lo = lds_read_tr16(image + ks_off(k, idx0 + 4 * p));
        const bf16x4 hi = lds_read_tr16(image + ks_off(k + 4, idx0 + 4 * p));
        return cat_bf16x4(lo, hi);
    }
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() would also drain vmcnt, i.e. wait for every global
// store already issued - in the epilogue that serialises the passes on HBM write latency.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

template <bool A_KC, bool B_KC, int EPI>
__global__ __launch_bounds__(kThreads, kWN == 4 ? 2 : 1) void gemm256_kernel(const uint16_t* __restrict__ A, const uint16_t* __restrict__ B,
                                                         void* __restrict__ Cv, const float* R, int64_t M, int64_t N, int64_t K,
                                                         int64_t lda, int64_t ldb, int64_t ldc, int tiles_n, int ntiles, int total,
                                                         int64_t kchunk, int64_t c_split_stride, RopeArgs rope) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / kWN, wn = wid % kWN;

    // Persistent workgroups: work item v = (k-split, tile); block b takes v = b, b + grid, ...  The XCD-aware (bijective)
    // order gives the blocks of one XCD (b % 8 equal) neighbouring tiles in every round, so operand panels are re-read
    // from that XCD's L2.  Speed only: any order is correct.
    const int q8 = total / 8, r8 = total % 8;
    auto decode = [&](int v, int64_t& m0, int64_t& n0, int& z) {
        const int xcd = v % 8;
        const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + v / 8;
        z = swz / ntiles;
        const int t = swz - z * ntiles;
        m0 = (int64_t)(t / tiles_n) * TM;
        n0 = (int64_t)(t % tiles_n) * TN;
    };

    int v = blockIdx.x;
    int64_t m0, n0;
    int z;
    decode(v, m0, n0, z);
    int64_t kbeg = (int64_t)z * kchunk;
    int nk = (int)((min(K, kbeg + kchunk) - kbeg) / TK);  // K % 64 == 0 on this path

    Stager<A_KC> sa;
    Stager<B_KC> sb;
    sa.init(A, lda, m0, M, kbeg, wid, lane);
    sb.init(B, ldb, n0, N, kbeg, wid, lane);
    int stage = 0;
    sa.issue(smem, wid);
    sb.issue(smem + kOperandBytes, wid, CM3P_AUD_B);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    while (true) {
        f32x4 acc[8][kNJ];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < kNJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        const int vn = v + gridDim.x;
        const bool has_next = vn < total;
        int64_t m0n = 0, n0n = 0, kbegn = 0;
        int zn = 0, nkn = 0;

        for (int kt = 0; kt < nk; ++kt) {
            char* cur = smem + stage * kStage;
            char* nxt = smem + (stage ^ 1) * kStage;
            // Steady state: the next k-tile of this work item.  In the forward / input-gradient instances the two waves of a SIMD
            // (w and w + 4) issue it at different points of the k-step - waves 0-3 before their first MFMA group, waves 4-7 after
            // it: an LDS-DMA instruction holds its wave's issue port for 40-70 cycles (tools/ubench/lds_dma_issue.hip), and with
            // all eight waves issuing at the top of the step the matrix pipe idles for the whole burst (one-call A/B: dgrad -4 %,
            // forward -2 %).  Issuing later than that (after group 1 or 2) costs 5-10 %: with two 64 KiB stages those loads no
            // longer land before the barrier that ends the step.  The weight-gradient instance gained nothing and issues at the top.
            const bool steady = kt + 1 < nk;
            auto issue_at = [&](int g) {
                if (steady && g == ((A_KC && kWN == 4 && wm == 1) ? 0 : -1)) {
                    sa.issue(nxt, wid);
                    if constexpr (!(CM3P_G256_ABL & 1)) sb.issue(nxt + kOperandBytes, wid, CM3P_AUD_B);
                }
            };
            if (!steady && has_next) {
                // cross-tile prefetch: the first k-tile of the NEXT work item streams in under this item's last MFMAs and
                // epilogue, so the next item starts without a load bubble
                decode(vn, m0n, n0n, zn);
                kbegn = (int64_t)zn * kchunk;
                nkn = (int)((min(K, kbegn + kchunk) - kbegn) / TK);
                sa.init(A, lda, m0n, M, kbegn, wid, lane);
                sb.init(B, ldb, n0n, N, kbegn, wid, lane);
                sa.issue(nxt, wid);
                sb.issue(nxt + kOperandBytes, wid, CM3P_AUD_B);
            }
            issue_at(-1);
            const char* ia = cur;
            const char* ib = cur + kOperandBytes;
            if constexpr (A_KC) {
                // Forward and input-gradient GEMMs: the k-step's 64 MFMAs run as four groups of 16 (g = 2 kk + half) and the
                // fragments of group g + 1 are requested before the MFMAs of group g are issued (left alone, the compiler sinks each
                // group's ds_reads next to their use), so only the first group's LDS latency is exposed: one-call A/B dgrad -8 %,
                // forward -1..3 %.  The weight-gradient GEMM (both operands read through ds_read_b64_tr_b16, twice the read
                // instructions) lost 2-4 % with the same schedule and keeps the plain loop below.
                bf16x8 fbq[2][kNJ], faq[2][4];
#pragma unroll
                for (int j = 0; j < kNJ; ++j) fbq[0][j] = frag<B_KC>(ib, wn * kWaveN + j * 16, 0, lane);
#pragma unroll
                for (int i = 0; i < 4; ++i) faq[0][i] = frag<A_KC>(ia, wm * 128 + i * 16, 0, lane);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int kk = g >> 1, half = g & 1;
                    if (g < 3) {
                        const int kn = (g + 1) >> 1, hn = (g + 1) & 1;
                        if (hn == 0 && !(CM3P_G256_ABL & 2)) {
#pragma unroll
                            for (int j = 0; j < kNJ; ++j) fbq[kn & 1][j] = frag<B_KC>(ib, wn * kWaveN + j * 16, kn, lane);
                        }
#pragma unroll
                        for (int i = 0; i < 4; ++i) faq[(g + 1) & 1][i] = frag<A_KC>(ia, wm * 128 + (hn * 4 + i) * 16, kn, lane);
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < kNJ; ++j)
                            acc[half * 4 + i][j] =
                                __builtin_amdgcn_mfma_f32_16x16x32_bf16(fbq[kk & 1][j], faq[g & 1][i], acc[half * 4 + i][j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);  // the next group's reads stay above these MFMAs
                    issue_at(g);
                }
            } else {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    bf16x8 fb[kNJ];
#pragma unroll
                    for (int j = 0; j < kNJ; ++j) fb[j] = frag<B_KC>(ib, wn * kWaveN + j * 16, kk, lane);
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        bf16x8 fa[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) fa[i] = frag<A_KC>(ia, wm * 128 + (half * 4 + i) * 16, kk, lane);
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int j = 0; j < kNJ; ++j)
                                acc[half * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[half * 4 + i][j], 0, 0, 0);
                        issue_at(kk * 2 + half);
                    }
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's LDS-DMA for the other stage has landed
            __syncthreads();                                   // everyone's has, and everyone is done reading `cur`
            stage ^= 1;
        }

        // ---- epilogue through LDS: fragment-shaped accumulators -> whole rows -> 16-byte coalesced stores -----------
        // (a lane owns 4 consecutive n of one m; storing that directly issues 32 partial-line stores per lane and is
        //  store-issue bound.)  `stage` now holds the next item's first k-tile (if any); the other 64 KiB are free.
        char* ebuf = smem + (stage ^ 1) * kStage;
        if constexpr (EPI == CM3P_EPI_BF16 || EPI == CM3P_EPI_BF16_ROPE) {
            constexpr int kRow = TN * 2 + 16;  // padded row pitch (bytes): 16 rows of one column land on different banks
            uint16_t* C = static_cast<uint16_t*>(Cv);
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {  // 64 rows per pass: waves with wm == pass/2, accumulator rows i in [4*(pass&1), +4)
                if (wm == (pass >> 1)) {
#pragma unroll
                    for (int i4 = 0; i4 < 4; ++i4)
#pragma unroll
                        for (int j = 0; j < kNJ; ++j) {
                            const f32x4 a = acc[(pass & 1) * 4 + i4][j];
                            const int r = i4 * 16 + (lane & 15), cidx = wn * kWaveN + j * 16 + 4 * (lane >> 4);
                            *reinterpret_cast<uint2*>(ebuf + r * kRow + cidx * 2) = uint2{pack_bf16x2(a.x, a.y), pack_bf16x2(a.z, a.w)};
                        }
                }
                lds_barrier();
                bool rotated = false;
                if constexpr (EPI == CM3P_EPI_BF16_ROPE) {
                    // Rotary embedding at store time, on whole staged rows (apply_rotary_pos_emb on the bf16 projection, as the
                    // reference's autocast path does): a work item is (row, head, 8 dims d..d+7 < 32) = the chunk pair (d, d+32);
                    // one table read serves both.  Doing it in the accumulators instead cost 33 % of this GEMM (eight serialised
                    // gather round trips per wave and ~50 spilled VGPRs).
                    if (n0 < rope.ncols) {  // tiles are head-aligned: a tile lies entirely inside or outside the rotated columns
                        rotated = true;
#pragma unroll
                        for (int u = 0; u < kItems / 2; ++u) {
                            const int id = tid + kThreads * u, r = id >> 4, hd = (id >> 2) & 3, dc = id & 3;
                            const int64_t m = m0 + pass * 64 + r, n = n0 + hd * 64 + dc * 8;
                            if (m < M && n < N) {
                                const int64_t prow = rope.per_batch ? m : (int64_t)((uint32_t)m % (uint32_t)rope.S);  // M < 2^31 (checked by the caller)
                                const float* cr = rope.cos + prow * 32 + dc * 8;
                                const float* sr = rope.sin + prow * 32 + dc * 8;
                                const f32x4 c0 = *reinterpret_cast<const f32x4*>(cr), c1 = *reinterpret_cast<const f32x4*>(cr + 4);
                                const f32x4 s0 = *reinterpret_cast<const f32x4*>(sr), s1 = *reinterpret_cast<const f32x4*>(sr + 4);
                                const uint4 xa = *reinterpret_cast<const uint4*>(ebuf + r * kRow + (hd * 8 + dc) * 16);
                                const uint4 xb = *reinterpret_cast<const uint4*>(ebuf + r * kRow + (hd * 8 + dc + 4) * 16);
                                const uint32_t wa[4] = {xa.x, xa.y, xa.z, xa.w}, wb[4] = {xb.x, xb.y, xb.z, xb.w};
                                const float cs[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
                                const float sn[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
                                uint32_t oa[4], ob[4];
                                const float qs = n < rope.q_cols ? rope.q_scale : 1.f;  // q columns: softmax scale folded in before the rounding
#pragma unroll
                                for (int t = 0; t < 4; ++t) {
                                    const float a0 = bf16lo(wa[t]), a1 = bf16hi(wa[t]), b0 = bf16lo(wb[t]), b1 = bf16hi(wb[t]);
                                    oa[t] = pack_bf16x2(qs * (a0 * cs[2 * t] - b0 * sn[2 * t]), qs * (a1 * cs[2 * t + 1] - b1 * sn[2 * t + 1]));
                                    ob[t] = pack_bf16x2(qs * (b0 * cs[2 * t] + a0 * sn[2 * t]), qs * (b1 * cs[2 * t + 1] + a1 * sn[2 * t + 1]));
                                }
                                *reinterpret_cast<uint4*>(C + m * ldc + n) = uint4{oa[0], oa[1], oa[2], oa[3]};
                                *reinterpret_cast<uint4*>(C + m * ldc + n + 32) = uint4{ob[0], ob[1], ob[2], ob[3]};
                            }
                        }
                    }
                }
                if (!rotated) {
#pragma unroll
                    for (int u = 0; u < kItems; ++u) {
                        const int id = tid + kThreads * u, r = id >> 5, ch = id & 31;
                        const int64_t m = m0 + pass * 64 + r, n = n0 + ch * 8;
                        if (m < M && n < N) *reinterpret_cast<uint4*>(C + m * ldc + n) = *reinterpret_cast<const uint4*>(ebuf + r * kRow + ch * 16);
                    }
                }
                lds_barrier();
            }
        } else {
            constexpr int kRow = TN * 4 + 16;
            float* C = static_cast<float*>(Cv) + (int64_t)z * c_split_stride;
            // residual rows of pass p+1 are requested before pass p's LDS round trip (16 VGPRs): a pass then waits for loads that
            // have had a whole pass to arrive instead of issuing them and stalling on HBM latency eight times per tile
            f32x4 rres[kItems];
            auto load_resid = [&](int pass) {
                if constexpr (EPI == CM3P_EPI_F32_RESID) {
#pragma unroll
                    for (int u = 0; u < kItems; ++u) {
                        const int id = tid + kThreads * u, r = id >> 6, ch = id & 63;
                        const int64_t m = m0 + pass * 32 + r, n = n0 + ch * 4;
                        rres[u] = (m < M && n < N) ? *reinterpret_cast<const f32x4*>(R + m * ldc + n) : f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                }
                if constexpr (EPI == CM3P_EPI_F32_BIAS) {  // a thread keeps its four columns in every pass: one read per tile
                    if (pass == 0) {
#pragma unroll
                        for (int u = 0; u < kItems; ++u) {
                            const int64_t n = n0 + ((tid + kThreads * u) & 63) * 4;
                            rres[u] = n < N ? *reinterpret_cast<const f32x4*>(R + n) : f32x4{0.f, 0.f, 0.f, 0.f};
                        }
                    }
                }
            };
            load_resid(0);
#pragma unroll
            for (int pass = 0; pass < 8; ++pass) {  // 32 rows per pass: waves with wm == pass/4, accumulator rows 2*(pass&3), +1
                if (wm == (pass >> 2)) {
#pragma unroll
                    for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
                        for (int j = 0; j < kNJ; ++j) {
                            const f32x4 a = acc[(pass & 3) * 2 + i2][j];  // pass loop fully unrolled: static register index
                            const int r = i2 * 16 + (lane & 15), cidx = wn * kWaveN + j * 16 + 4 * (lane >> 4);
                            *reinterpret_cast<f32x4*>(ebuf + r * kRow + cidx * 4) = a;
                        }
                }
                lds_barrier();
                f32x4 rcur[kItems];
#pragma unroll
                for (int u = 0; u < kItems; ++u) rcur[u] = rres[u];
                if (pass < 7) load_resid(pass + 1);
#pragma unroll
                for (int u = 0; u < kItems; ++u) {
                    const int id = tid + kThreads * u, r = id >> 6, ch = id & 63;
                    const int64_t m = m0 + pass * 32 + r, n = n0 + ch * 4;
                    if (m < M && n < N) {
                        f32x4 a = *reinterpret_cast<const f32x4*>(ebuf + r * kRow + ch * 16);
                        if constexpr (EPI == CM3P_EPI_F32_RESID || EPI == CM3P_EPI_F32_BIAS) a += rcur[u];
                        *reinterpret_cast<f32x4*>(C + m * ldc + n) = a;
                    }
                }
                lds_barrier();
            }
        }

        if (!has_next) break;
        v = vn;
        m0 = m0n;
        n0 = n0n;
        z = zn;
        nk = nkn;
    }
}

template <bool A_KC, bool B_KC>
int launch256(const uint16_t* a, const uint16_t* b, void* C, const float* R, int64_t M, int64_t N, int64_t K, int64_t lda,
              int64_t ldb, int64_t ldc, int epi, int splits, int64_t kchunk, int64_t c_split_stride, hipStream_t s, RopeArgs rope) {
    const int tiles_m = (int)((M + TM - 1) / TM), tiles_n = (int)((N + TN - 1) / TN);
    const int ntiles = tiles_m * tiles_n, total = ntiles * splits;
    const int num_cu = cm3p_num_cu();
    const dim3 grid(total < num_cu ? total : num_cu);  // one persistent 512-thread workgroup per CU
    const size_t lds = 2 * kStage;
#define CM3P_G256(E)                                                                                                     \
    {                                                                                                                    \
        static Cm3pDevOnce once;                                                                                         \
        const int rc_once = once.run([&] {                                                                               \
            return hipFuncSetAttribute((const void*)gemm256_kernel<A_KC, B_KC, E>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                       (int)lds) == hipSuccess                                                           \
                       ? CM3P_OK                                                                                         \
                       : CM3P_ERR_LAUNCH;                                                                                \
        });                                                                                                              \
        if (rc_once != CM3P_OK) return rc_once;                                                                          \
        gemm256_kernel<A_KC, B_KC, E><<<grid, kThreads, lds, s>>>(a, b, C, R, M, N, K, lda, ldb, ldc, tiles_n, ntiles, total, kchunk, c_split_stride, rope); \
    }
    switch (epi) {
        case CM3P_EPI_BF16: CM3P_G256(CM3P_EPI_BF16) break;
        case CM3P_EPI_F32: CM3P_G256(CM3P_EPI_F32) break;
        case CM3P_EPI_F32_RESID: CM3P_G256(CM3P_EPI_F32_RESID) break;
        case CM3P_EPI_BF16_ROPE:
            if constexpr (A_KC && B_KC) {
                CM3P_G256(CM3P_EPI_BF16_ROPE)
                break;
            }
            return CM3P_ERR_INVALID;
        case CM3P_EPI_F32_BIAS:
            if constexpr (A_KC && B_KC) {
                CM3P_G256(CM3P_EPI_F32_BIAS)
                break;
            }
            return CM3P_ERR_INVALID;
        default: return CM3P_ERR_INVALID;
    }
#undef CM3P_G256
    return CM3P_OK;
}

}  // namespace

// Internal entry used by cm3p_gemm_bf16 (gemm.hip) when the shape qualifies; not part of the public header.
int cm3p_gemm256_dispatch(const void* A, const void* B, void* C, const float* R, int64_t M, int64_t N, int64_t K, int64_t lda,
                          int64_t ldb, int64_t ldc, int a_kc, int b_kc, int epi, int splits, int64_t kchunk,
                          int64_t c_split_stride, hipStream_t s, RopeArgs rope) {
    const uint16_t* a = static_cast<const uint16_t*>(A);
    const uint16_t* b = static_cast<const uint16_t*>(B);
    if (a_kc && b_kc) return launch256<true, true>(a, b, C, R, M, N, K, lda, ldb, ldc, epi, splits, kchunk, c_split_stride, s, rope);
    if (a_kc) return launch256<true, false>(a, b, C, R, M, N, K, lda, ldb, ldc, epi, splits, kchunk, c_split_stride, s, rope);
    if (b_kc) return launch256<false, true>(a, b, C, R, M, N, K, lda, ldb, ldc, epi, splits, kchunk, c_split_stride, s, rope);
    return launch256<false, false>(a, b, C, R, M, N, K, lda, ldb, ldc, epi, splits, kchunk, c_split_stride, s, rope);
}

// timing-only ablation switches this object was built with (0 in every shipped build: cm3p_build_ablation_flags, tests/test_cabi.py)
int cm3p_ablation_flags_gemm256() { return (CM3P_G256_ABL) | ((CM3P_G256_WN != 4) << 8); }
#if CM3P_DMA_AUDIT
int cm3p_audit_set_gemm256(void* buf) { return cm3p_audit_set_local(buf); }
#endif
