// 128 x 256 x 32 bf16 MFMA GEMM with TWO resident workgroups per CU (r04; the r03 verdict's item 5: "build, not cost").
//
// gemm8p.hip's workgroup owns its CU (160 KiB of LDS, 2 x 248 VGPRs per SIMD), so while it stores a tile nothing else runs there: at the
// step's K = 768 forward shapes the epilogue is a quarter (bf16) to a half (fp32 + residual) of the kernel.  Here a workgroup is four
// waves and 72 KiB, two of them share a CU, and one's store phase overlaps the other's k-loop on the same SIMDs.
//
//   tile      128 rows x 256 columns per workgroup; wave w owns columns 64 w .. 64 w + 63 of all 128 rows: the 128 x 64 wave tile,
//             the fragment layout, the MFMA (v_mfma_f32_16x16x32_bf16, B fragment first) and the accumulation order of gemm8p.hip,
//             so the results are bit-identical to it.
//   operands  both k-contiguous (the forward orientation and the input gradient through W^T): A [M, lda], B [N, ldb].
//   ring      three stages of one 32-deep k-step: A image [128 rows][64 bytes] (8 KiB) + B image [256 rows][64 bytes] (16 KiB).
//             16-byte chunk c of row r sits at chunk position c ^ f(r >> 2 & 3), f = {0, 3, 2, 1}: a ds_read_b128 fragment read (lane
//             = row of a 16-row group, chunk lane >> 4) is then conflict free in each of its four 16-lane groups.  LDS-DMA writes
//             lane-linearly, so the permutation is applied to each lane's SOURCE chunk.
//             A wave stages the two A pieces of its 32 rows and the four B pieces of ITS OWN 64 columns: the B image is wave-private
//             (only the A image needs the workgroup barrier).
//   k-loop    position p (a k-step of some work item) lives in slot p % 3; per position: `s_waitcnt vmcnt(6)` (position p landed,
//             p + 1 may be in flight), one barrier (everybody's A pieces of p are there AND everybody is done reading p - 1), the
//             LDS-DMA of position p + 2 into the slot p - 1 has left, 12 ds_read_b128, 32 MFMAs.  The stream of positions runs on
//             into the next work item: the ring never drains.
//   epilogue  per wave, 16 rows at a time through 4 KiB of LDS - the wave's own B region of the slot it has just finished with
//             (nobody else reads it, and its next overwrite is this wave's own DMA one position later): no extra LDS, no barrier.
#include <stdlib.h>

#include "common.h"

#ifndef CM3P_G4W_SCHED
#define CM3P_G4W_SCHED 0  // how a k-step's fragment reads and MFMAs are placed (one-call A/B, tools/ubench/g4w_variants.sh)
#endif

namespace {

constexpr int kStage4 = 8192 + 16384;  // A image + B image of one k-step
constexpr int kLds4 = 3 * kStage4;     // 72 KiB: two workgroups per CU

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4w;

// one 1-KiB LDS-DMA piece: 16 rows x 64 bytes; global address = sbase + voff, LDS address = lds + 16 * lane
template <int AUD>
__device__ __forceinline__ void glds4(uint32_t voff, const char* sbase, uint32_t lds) {
    CM3P_AUDIT(AUD, sbase + voff, 16);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds) : "memory", "m0");
}

__device__ __forceinline__ int swz4(int row) { return (4 - ((row >> 2) & 3)) & 3; }

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm4w_kernel(const uint16_t* __restrict__ A, const uint16_t* __restrict__ B, void* __restrict__ Cv,
                                                        const float* R, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc,
                                                        int tiles_n, int total, RopeArgs rope) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem);
    const int nk = (int)(K / 32);

    // work item v -> tile, XCD-aware bijective order (consecutive items of an XCD share the A row panel); speed only
    const int q8 = total / 8, r8 = total % 8;
    auto decode = [&](int v, int64_t& m0, int64_t& n0) {
        const int xcd = v % 8;
        const int t = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + v / 8;
        m0 = (int64_t)(t / tiles_n) * 128;
        n0 = (int64_t)(t % tiles_n) * 256;
    };

    // ---- staging stream: lane l of a piece brings row l >> 2, chunk position l & 3 = source chunk (l & 3) ^ swz(row)
    const int prow = lane >> 2;
    const int pch = ((lane & 3) ^ swz4(prow)) << 4;
    const uint32_t voff_a = (uint32_t)(prow * (int)lda * 2 + pch), voff_b = (uint32_t)(prow * (int)ldb * 2 + pch);
    const char* sa[2];  // this wave's two A pieces (rows 32 w + 16 i ..) and four B pieces (rows 64 w + 16 j ..) at the stream's k
    const char* sb[4];
    int sv = blockIdx.x, sk = 0;
    bool sdone = false;
    auto stream_setup = [&](int v) {
        int64_t m0, n0;
        decode(v, m0, n0);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int64_t r = m0 + 32 * wid + 16 * i;
            if (r > M - 16) r = M - 16;  // (M % 16 == 0: a piece is inside or outside as a whole; what is loaded for the outside is never stored)
            sa[i] = reinterpret_cast<const char*>(A + r * lda);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int64_t r = n0 + 64 * wid + 16 * j;
            if (r > N - 16) r = N - 16;
            sb[j] = reinterpret_cast<const char*>(B + r * ldb);
        }
    };
    int spos = 0;  // positions issued so far
    auto stream_issue = [&]() {
        // (past this workgroup's last k-step the last one is staged again - never read - so that the vmcnt pattern stays fixed)
        const uint32_t slot = lds0 + (uint32_t)(spos % 3) * kStage4;
        glds4<CM3P_AUD_A>(voff_a, sa[0], slot + (2 * wid) * 1024);
        glds4<CM3P_AUD_A>(voff_a, sa[1], slot + (2 * wid + 1) * 1024);
#pragma unroll
        for (int j = 0; j < 4; ++j) glds4<CM3P_AUD_B>(voff_b, sb[j], slot + 8192 + (4 * wid + j) * 1024);
        ++spos;
        if (sdone) return;
        if (++sk < nk) {
#pragma unroll
            for (int i = 0; i < 2; ++i) sa[i] += 64;
#pragma unroll
            for (int j = 0; j < 4; ++j) sb[j] += 64;
        } else {
            sk = 0;
            sv += gridDim.x;
            if (sv < total) stream_setup(sv);
            else sdone = true;
        }
    };

    // fragment reads: lane = row (lane & 15) of a 16-row group, chunk lane >> 4
    const int frow = lane & 15;
    const int fa_off = frow * 64 + (((lane >> 4) ^ swz4(frow)) << 4);             // + 1024 * (16-row group) inside the A image
    const int fb_off = 8192 + (64 * wid + frow) * 64 + (((lane >> 4) ^ swz4(frow)) << 4);  // + 1024 * (16-column group)

    f32x4 acc[8][4];

    auto epilogue = [&](int64_t m0, int64_t n0, char* ebuf) {
        const int64_t nw = n0 + wid * 64;
        const bool full = m0 + 128 <= M && nw + 64 <= N;
        if constexpr (EPI == CM3P_EPI_BF16 || EPI == CM3P_EPI_BF16_ROPE) {
            char* Cb = reinterpret_cast<char*>(static_cast<uint16_t*>(Cv) + m0 * ldc + nw);
            const uint32_t ldcb = (uint32_t)ldc * 2;
            const bool rotate = (EPI == CM3P_EPI_BF16_ROPE) && nw < rope.ncols;
            const float qs = (EPI == CM3P_EPI_BF16_ROPE && nw < rope.q_cols) ? rope.q_scale : 1.f;
            uint32_t prow0 = 0;
            if constexpr (EPI == CM3P_EPI_BF16_ROPE) prow0 = rope.per_batch ? 0u : (uint32_t)m0 % (uint32_t)rope.S;
#pragma unroll
            for (int i4 = 0; i4 < 8; ++i4) {
                char* eb = ebuf + (i4 & 1) * 2048;
#pragma unroll
                for (int j4 = 0; j4 < 4; ++j4) {
                    const f32x4 a = acc[i4][j4];
                    const int row = lane & 15, s8 = (j4 * 4 + (lane >> 4)) ^ ((row & 7) << 1);
                    *reinterpret_cast<uint2*>(eb + row * 128 + s8 * 8) = uint2{pack_bf16x2(a.x, a.y), pack_bf16x2(a.z, a.w)};
                }
                asm volatile("" ::: "memory");
                if (rotate) {  // (gemm8p.hip's rotation at store time, to the bit)
                    const int row = lane >> 2, dc = lane & 3;
                    char* pa = eb + row * 128 + ((dc ^ (row & 7)) << 4);
                    char* pb = eb + row * 128 + (((dc + 4) ^ (row & 7)) << 4);
                    const u32x4w xa = *reinterpret_cast<const u32x4w*>(pa);
                    const u32x4w xb = *reinterpret_cast<const u32x4w*>(pb);
                    if (full || m0 + i4 * 16 + row < M) {
                        int64_t prw;
                        if (rope.per_batch) prw = m0 + i4 * 16 + row;
                        else {
                            uint32_t x = prow0 + i4 * 16 + row;
                            if (rope.S >= 128) x = x >= (uint32_t)rope.S ? x - (uint32_t)rope.S : x;
                            else x %= (uint32_t)rope.S;
                            prw = x;
                        }
                        const float* cr = rope.cos + prw * 32 + dc * 8;
                        const float* sr = rope.sin + prw * 32 + dc * 8;
                        const f32x4 c0 = *reinterpret_cast<const f32x4*>(cr), c1 = *reinterpret_cast<const f32x4*>(cr + 4);
                        const f32x4 s0 = *reinterpret_cast<const f32x4*>(sr), s1 = *reinterpret_cast<const f32x4*>(sr + 4);
                        const float cs[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
                        const float sn[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
                        u32x4w oa, ob;
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const float a0 = bf16lo(xa[t]), a1 = bf16hi(xa[t]), b0 = bf16lo(xb[t]), b1 = bf16hi(xb[t]);
                            oa[t] = pack_bf16x2(qs * (a0 * cs[2 * t] - b0 * sn[2 * t]), qs * (a1 * cs[2 * t + 1] - b1 * sn[2 * t + 1]));
                            ob[t] = pack_bf16x2(qs * (b0 * cs[2 * t] + a0 * sn[2 * t]), qs * (b1 * cs[2 * t + 1] + a1 * sn[2 * t + 1]));
                        }
                        *reinterpret_cast<u32x4w*>(pa) = oa;
                        *reinterpret_cast<u32x4w*>(pb) = ob;
                    }
                    asm volatile("" ::: "memory");
                }
                const int r0 = lane >> 3, r1 = 8 + (lane >> 3), ch = lane & 7;
                const u32x4w x0 = *reinterpret_cast<const u32x4w*>(eb + r0 * 128 + ((ch ^ (r0 & 7)) << 4));
                const u32x4w x1 = *reinterpret_cast<const u32x4w*>(eb + r1 * 128 + ((ch ^ (r1 & 7)) << 4));
                asm volatile("" ::: "memory");
                const bool col_ok = full || nw + ch * 8 < N;
                if (full || (col_ok && m0 + i4 * 16 + r0 < M)) gstore16<(CM3P_NT & 1) != 0>(Cb + (uint32_t)(i4 * 16 + r0) * ldcb + ch * 16, x0);
                if (full || (col_ok && m0 + i4 * 16 + r1 < M)) gstore16<(CM3P_NT & 1) != 0>(Cb + (uint32_t)(i4 * 16 + r1) * ldcb + ch * 16, x1);
            }
        } else {  // fp32, fp32 + residual
            char* Cb = reinterpret_cast<char*>(static_cast<float*>(Cv) + m0 * ldc + nw);
            const char* Rb = reinterpret_cast<const char*>(R + m0 * ldc + nw);
            const uint32_t ldcb = (uint32_t)ldc * 4;
            const int lrow = lane >> 4, lch = lane & 15;
            const uint32_t loff = (uint32_t)lrow * ldcb + lch * 16;
            const bool col_ok = full || nw + lch * 4 < N;
            f32x4 rnext[4];
            auto load_r = [&](int i4) {
                if constexpr (EPI == CM3P_EPI_F32_RESID) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int row = i4 * 16 + u * 4;
                        rnext[u] = (full || (col_ok && m0 + row + lrow < M)) ? *reinterpret_cast<const f32x4*>(Rb + loff + (uint32_t)row * ldcb)
                                                                             : f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                }
            };
            load_r(0);
#pragma unroll
            for (int i4 = 0; i4 < 8; ++i4) {
#pragma unroll
                for (int j4 = 0; j4 < 4; ++j4) {
                    const int row = lane & 15, ch = j4 * 4 + (lane >> 4);
                    *reinterpret_cast<f32x4*>(ebuf + row * 256 + ((ch ^ row) << 4)) = acc[i4][j4];
                }
                asm volatile("" ::: "memory");
                f32x4 x[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int row = u * 4 + lrow;
                    x[u] = *reinterpret_cast<const f32x4*>(ebuf + row * 256 + ((lch ^ row) << 4));
                }
                asm volatile("" ::: "memory");
                if constexpr (EPI == CM3P_EPI_F32_RESID) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) x[u] += rnext[u];
                }
                if (i4 < 7) load_r(i4 + 1);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int row = i4 * 16 + u * 4;
                    if (full || (col_ok && m0 + row + lrow < M)) gstore16f<(CM3P_NT & 32) != 0>(Cb + loff + (uint32_t)row * ldcb, x[u]);
                }
            }
        }
    };

    // prologue: positions 0 and 1
    stream_setup(sv);
    stream_issue();
    stream_issue();
    int pos = 0;          // position being consumed
    bool tail = false;    // the previous work item's stores are still in the queue behind position pos + 1's pieces (see below)
    for (int v = blockIdx.x; v < total; v += gridDim.x) {
        int64_t m0, n0;
        decode(v, m0, n0);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int kt = 0; kt < nk; ++kt, ++pos) {
            const char* st = smem + (pos % 3) * kStage4;
            // this wave's six pieces of position `pos` have landed when at most the operations issued after them are outstanding: the six
            // pieces of position pos + 1 and, in the first k-step behind a FULL tile's epilogue, that epilogue's stores (16 / 32, issued
            // unconditionally there; an edge tile's count is not known: the plain wait then also covers its stores - slower, never wrong)
            if (tail) {
                if constexpr (EPI == CM3P_EPI_BF16 || EPI == CM3P_EPI_BF16_ROPE) asm volatile("s_waitcnt vmcnt(22)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(38)" ::: "memory");
                tail = false;
            } else {
                asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();                     // ... everybody's have, and everybody is done reading position pos - 1
            stream_issue();                                   // position pos + 2 -> the slot of position pos - 1
#if CM3P_G4W_SCHED == 0
            bf16x8 fa[8], fb[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) fb[nt] = *reinterpret_cast<const bf16x8*>(st + fb_off + nt * 1024);
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) fa[mt] = *reinterpret_cast<const bf16x8*>(st + fa_off + mt * 1024);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int mt = 0; mt < 8; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[nt], fa[mt], acc[mt][nt], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
#elif CM3P_G4W_SCHED == 1  // the compiler places reads and MFMAs (its own counted lgkmcnt waits)
            bf16x8 fa[8], fb[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) fb[nt] = *reinterpret_cast<const bf16x8*>(st + fb_off + nt * 1024);
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) fa[mt] = *reinterpret_cast<const bf16x8*>(st + fa_off + mt * 1024);
#pragma unroll
            for (int mt = 0; mt < 8; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[nt], fa[mt], acc[mt][nt], 0, 0, 0);
#else  // two halves: the A fragments of rows 64 .. 127 are requested before the MFMAs of rows 0 .. 63 are issued
            bf16x8 fa[8], fb[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) fb[nt] = *reinterpret_cast<const bf16x8*>(st + fb_off + nt * 1024);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) fa[mt] = *reinterpret_cast<const bf16x8*>(st + fa_off + mt * 1024);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 4; mt < 8; ++mt) fa[mt] = *reinterpret_cast<const bf16x8*>(st + fa_off + mt * 1024);
            asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[nt], fa[mt], acc[mt][nt], 0, 0, 0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 4; mt < 8; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[nt], fa[mt], acc[mt][nt], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
#endif
        }
        // the wave's own B region of the slot it has just read (see the header): its next writer is this wave's DMA of position pos + 2
        epilogue(m0, n0, smem + ((pos - 1) % 3) * kStage4 + 8192 + wid * 4096);
        tail = m0 + 128 <= M && n0 + wid * 64 + 64 <= N;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA may outlive the workgroup's LDS allocation
}

}  // namespace

// Internal entry used by gemm.hip (development switch CM3P_GEMM_IMPL=4w, tools/gemm_ab.py); CM3P_ERR_INVALID = not covered.
int cm3p_gemm4w_dispatch(const void* A, const void* B, void* C, const float* R, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb,
                         int64_t ldc, int epi, hipStream_t s, RopeArgs rope) {
    if (M % 16 != 0 || N % 16 != 0 || K % 32 != 0 || K < 64) return CM3P_ERR_INVALID;
    if (lda * 2 * 16 >= (int64_t(1) << 31) || ldb * 2 * 16 >= (int64_t(1) << 31)) return CM3P_ERR_INVALID;  // 32-bit lane offsets
    const int tiles_m = (int)((M + 127) / 128), tiles_n = (int)((N + 255) / 256);
    const int total = tiles_m * tiles_n;
    static int num_cu = 0;
    if (num_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) num_cu = prop.multiProcessorCount;
        if (num_cu <= 0) num_cu = 256;
    }
    const int grid = total < 2 * num_cu ? total : 2 * num_cu;
    const uint16_t* a = static_cast<const uint16_t*>(A);
    const uint16_t* b = static_cast<const uint16_t*>(B);
#define CM3P_G4W(E)                                                                                                           \
    {                                                                                                                         \
        static bool attr_set = false;                                                                                         \
        if (!attr_set) {                                                                                                      \
            if (hipFuncSetAttribute((const void*)gemm4w_kernel<E>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds4) != hipSuccess) return CM3P_ERR_LAUNCH; \
            attr_set = true;                                                                                                  \
        }                                                                                                                     \
        gemm4w_kernel<E><<<grid, 256, kLds4, s>>>(a, b, C, R, M, N, K, lda, ldb, ldc, tiles_n, total, rope);                  \
    }
    switch (epi) {
        case CM3P_EPI_BF16: CM3P_G4W(CM3P_EPI_BF16) break;
        case CM3P_EPI_F32: CM3P_G4W(CM3P_EPI_F32) break;
        case CM3P_EPI_F32_RESID: CM3P_G4W(CM3P_EPI_F32_RESID) break;
        case CM3P_EPI_BF16_ROPE: CM3P_G4W(CM3P_EPI_BF16_ROPE) break;
        default: return CM3P_ERR_INVALID;
    }
#undef CM3P_G4W
    return CM3P_OK;
}

#if CM3P_DMA_AUDIT
int cm3p_audit_set_gemm4w(void* buf) { return cm3p_audit_set_local(buf); }
#endif
