// NOT BUILT - r01 experiment.  Correct (passes the GEMM parity tests) but 10-25 % slower than gemm256.hip in all four
// schedule modes when A/B-ed in one session: the limiter is LDS-DMA issue cost (100-250 cycles per instruction per wave, s_memtime stamps)
// and load latency, not the phase alignment of the two waves of a SIMD.

// 256 x 256 bf16 MFMA GEMM, "staggered" schedule (same contract as gemm256.hip / gemm.hip).
//
// Why: rocprofv3 counters on gemm256.hip (r01): a wave is parked at s_waitcnt / s_barrier 30-40 % of its life, the matrix
// pipe is 44-54 % busy.  Both waves of a SIMD (w and w+4) pass the same barrier, read their fragments from LDS at the same
// time and then compute at the same time - nobody computes while the other one waits.  Here the two halves of the
// workgroup run ONE SLOT apart for the whole launch:
//
//     slot        s            s+1           s+2
//     waves 0-3   C(h)         C(h+1)        C(h+2)        C(h) = 32 MFMAs on half-stage h (K = 32) of the current tile
//     waves 4-7   C(h-1)       C(h)          C(h+1)
//
// so one wave of every SIMD is issuing MFMAs while its partner reads fragments, waits for a barrier or stores results.
// The epilogue is per wave (its 128 x 64 accumulators are transposed through a private 4 KiB LDS patch, no workgroup
// barrier inside) and takes kEpiSlots ordinary slots; a half moves straight on to the next tile, so the other half's MFMAs
// also cover the epilogue, and the prefetch ring never drains between tiles.
//
// LDS: ring of 4 half-stages (A half 16 KiB + B half 16 KiB each) = 128 KiB + 8 x 4 KiB epilogue patches = 160 KiB.
// Half-stage H is consumed by waves 0-3 in some slot t and by waves 4-7 in slot t+1; it is requested as soon as half-stage
// H-4 (same ring slot) has been consumed by both, i.e. up to three half-stages ahead, by LDS-DMA issued as inline assembly
// (see gemm256.hip for why), and awaited with a counted vmcnt in front of the barrier that ends the slot before t.
//   k-contiguous operand half: [256 rows][32 k] bf16 = 64-byte rows, 16-byte chunk c of row r at position c ^ f(r),
//       f(r) = {0,2,3,1}[(r >> 2) & 3]: conflict-free for ds_read_b128's lane groups ({0-3,12-15,20-27}, ...).
//   k-strided operand half:    [32 k][256 idx] bf16 = 512-byte rows, 32-byte segment XOR as in gemm256.hip, ds_read_b64_tr_b16.
#include <stdio.h>

#include <type_traits>

#include "common.h"

namespace {

// Diagnostic build only (-DCM3P_STAMPS): per-wave s_memtime segment sums of the slot loop, printed by the dispatcher.
#ifdef CM3P_STAMPS
__device__ unsigned long long g_seg[16];
#define SEG_INIT() unsigned long long tprev__ = __builtin_amdgcn_s_memtime(), tseg__[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define SEG(k)                                                        \
    do {                                                              \
        const unsigned long long tn__ = __builtin_amdgcn_s_memtime(); \
        tseg__[k] += tn__ - tprev__;                                  \
        tprev__ = tn__;                                               \
    } while (0)
#define SEG_FLUSH()                                                                \
    do {                                                                           \
        if ((threadIdx.x & 63) == 0) {                                             \
            for (int k__ = 0; k__ < 8; ++k__) atomicAdd(&g_seg[k__], tseg__[k__]); \
            atomicAdd(&g_seg[8], 1ull);                                            \
        }                                                                          \
    } while (0)
#else
#define SEG_INIT()
#define SEG(k)
#define SEG_FLUSH()
#endif

constexpr int TM = 256, TN = 256, HK = 32;
constexpr int kHalfOp = TM * HK * 2;   // 16 KiB: one operand, one half-stage
constexpr int kSlotBytes = 2 * kHalfOp;  // 32 KiB
constexpr int kRing = 4;
constexpr int kPatch = 4096;           // per-wave epilogue patch
constexpr int kLds = kRing * kSlotBytes + 8 * kPatch;
constexpr int kEpiSlots = 2;           // the epilogue of a wave is spread over this many slots (4 accumulator rows-of-16 each)
constexpr int kDmaPerHs = 4;
#ifndef GEMM256S_OPPOSITE
#define GEMM256S_OPPOSITE 0
#endif
#ifndef GEMM256S_LATEREQ
#define GEMM256S_LATEREQ 0
#endif
constexpr bool kOpposite = GEMM256S_OPPOSITE;  // halves read / multiply in opposite order inside a slot (else: same order, one half-stage apart)
constexpr bool kLateReq = GEMM256S_LATEREQ;    // waves 4-7 request at the end of the slot instead of its start
           // LDS-DMA instructions per wave and half-stage (2 A pieces + 2 B pieces)

__device__ __forceinline__ int kc_f(int r) { return (0x78 >> (((r >> 2) & 3) << 1)) & 3; }
__device__ __forceinline__ int kc_off(int r, int c) { return r * 64 + ((c ^ kc_f(r)) << 4); }
__device__ __forceinline__ int ksf(int k) { return (k & 3) | (((k >> 3) & 1) << 2); }
__device__ __forceinline__ int ks_off(int k, int idx) { return k * 512 + (((idx >> 4) ^ ksf(k)) << 5) + ((idx & 15) << 1); }

__device__ __forceinline__ void glds16(const void* gsrc, char* lds_wave_base) {
    const uint32_t m0v = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds_wave_base);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(m0v) : "memory", "m0");
}

// wait until at most n vector-memory operations of this wave are outstanding (n is wave-uniform; vmcnt takes an immediate)
__device__ __forceinline__ void wait_vm_at_most(int n) {
    if (n >= 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else if (n >= 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (n >= 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (n >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (n >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
__device__ __forceinline__ void slot_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// Source pointers of the two 1-KiB pieces this wave stages per operand and half-stage.
template <bool KC>
struct Stager {
    const uint16_t* src[2];
    int64_t hstep;
    __device__ __forceinline__ void init(const uint16_t* base, int64_t ld, int64_t idx0, int64_t extent, int64_t kbeg, int wid, int lane) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = wid * 2 + i;  // piece 0..15 of the half image
            if constexpr (KC) {
                const int r = q * 16 + (lane >> 2);
                const int c = (lane & 3) ^ kc_f(r);
                int64_t row = idx0 + r;
                if (row > extent - 1) row = extent - 1;  // rows past the edge are never stored
                src[i] = base + row * ld + kbeg + c * 8;
            } else {
                const int k = q * 2 + (lane >> 5);
                const int p = lane & 31;
                const int seg = (p >> 1) ^ ksf(k);
                int64_t col = idx0 + (seg * 2 + (p & 1)) * 8;
                if (col > extent - 8) col = extent - 8;
                src[i] = base + (kbeg + k) * ld + col;
            }
        }
        hstep = KC ? HK : HK * ld;
    }
    __device__ __forceinline__ void issue(char* image, int wid) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            glds16(src[i], image + (wid * 2 + i) * 1024);
            src[i] += hstep;
        }
    }
};

template <bool KC>
__device__ __forceinline__ bf16x8 frag(const char* image, int idx0, int lane) {
    if constexpr (KC) {
        return *reinterpret_cast<const bf16x8*>(image + kc_off(idx0 + (lane & 15), lane >> 4));
    } else {
        const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
        const int k = 8 * g + q;
        const bf16x4 lo = lds_read_tr16(image + ks_off(k, idx0 + 4 * p));
        const bf16x4 hi = lds_read_tr16(image + ks_off(k + 4, idx0 + 4 * p));
        return cat_bf16x4(lo, hi);
    }
}

// Position of one half of the workgroup in the launch's work: item v (or past the end), local step u of that item.
struct Cursor {
    int v;        // work item (k-split, tile), >= total: finished
    int u;        // step inside the item: [0, nh) compute half-stages, [nh, nh + kEpiSlots) epilogue
    int nh;       // half-stages of this item
    int64_t m0, n0, kbeg;
    int z;
};

template <bool A_KC, bool B_KC, int EPI>
__global__ __launch_bounds__(512, 1) void gemm256s_kernel(const uint16_t* __restrict__ A, const uint16_t* __restrict__ B,
                                                          void* __restrict__ Cv, const float* R, int64_t M, int64_t N, int64_t K,
                                                          int64_t lda, int64_t ldb, int64_t ldc, int tiles_n, int ntiles, int total,
                                                          int64_t kchunk, int64_t c_split_stride) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 2, wn = wid & 3;  // wm is also the half (group): waves 4-7 run one slot behind waves 0-3
    char* patch = smem + kRing * kSlotBytes + wid * kPatch;

    const int q8 = total / 8, r8 = total % 8;
    auto decode = [&](Cursor& c) {
        if (c.v >= total) {
            c.nh = 0;
            return;
        }
        const int xcd = c.v % 8;
        const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + c.v / 8;
        c.z = swz / ntiles;
        const int t = swz - c.z * ntiles;
        c.m0 = (int64_t)(t / tiles_n) * TM;
        c.n0 = (int64_t)(t % tiles_n) * TN;
        c.kbeg = (int64_t)c.z * kchunk;
        c.nh = (int)((min(K, c.kbeg + kchunk) - c.kbeg) / HK);  // K % 64 == 0 on this path
    };
    auto advance = [&](Cursor& c) {
        if (++c.u >= c.nh + kEpiSlots) {
            c.v += gridDim.x;
            c.u = 0;
            decode(c);
        }
    };

    // The loader walks the same item sequence one half-stage at a time (all waves run it identically).
    Cursor ld_cur{(int)blockIdx.x, 0, 0, 0, 0, 0, 0};
    decode(ld_cur);
    Stager<A_KC> sa;
    Stager<B_KC> sb;
    if (ld_cur.v < total) {
        sa.init(A, lda, ld_cur.m0, M, ld_cur.kbeg, wid, lane);
        sb.init(B, ldb, ld_cur.n0, N, ld_cur.kbeg, wid, lane);
    }
    int issued = 0;  // half-stages requested so far (= global index of the next one).  Every half-stage is kDmaPerHs DMA
                     // instructions per wave, and counted waits consider DMA only: other vector-memory traffic (residual loads,
                     // stores) can only make a wait longer than needed, never shorter.
    auto request_next = [&]() -> bool {  // request half-stage `issued` if there is one
        if (ld_cur.v >= total) return false;
        char* slot = smem + (issued & (kRing - 1)) * kSlotBytes;
        sa.issue(slot, wid);
        sb.issue(slot + kHalfOp, wid);
        ++issued;
        if (++ld_cur.u >= ld_cur.nh) {
            ld_cur.v += gridDim.x;
            ld_cur.u = 0;
            decode(ld_cur);
            if (ld_cur.v < total) {
                sa.init(A, lda, ld_cur.m0, M, ld_cur.kbeg, wid, lane);
                sb.init(B, ldb, ld_cur.n0, N, ld_cur.kbeg, wid, lane);
            }
        }
        return true;
    };
    // wait until half-stage h (already requested) has landed: at most the DMA of the younger half-stages may be outstanding
    auto await_hs = [&](int h) { wait_vm_at_most(kDmaPerHs * (issued - h - 1)); };

    // Both halves walk the same sequence of steps; `lead` is where waves 0-3 are, `lag` where waves 4-7 are (one slot
    // behind).  Every wave tracks both (wave-uniform scalars) so that all waves take the same request / wait decisions.
    Cursor lead{(int)blockIdx.x, 0, 0, 0, 0, 0, 0};
    decode(lead);
    Cursor lag = lead;
    int lead_consumed = 0, lag_consumed = 0;  // half-stages read so far by waves 0-3 / waves 4-7

    f32x4 acc[8][4];

    // Fragments of one half-stage (48 VGPRs) are read in one burst and consumed by 32 MFMAs.  The two halves do this in
    // OPPOSITE order inside a slot: waves 0-3 read half-stage h and then multiply it; waves 4-7 first multiply the fragments
    // they read in the previous slot (half-stage h-1) and then read h.  So in every slot one wave of each SIMD is in its
    // LDS burst while its partner owns the matrix pipe, and vice versa.
    bf16x8 fb[4], fa[8];
    auto read_frags = [&](const char* slot) {
        int ln = lane;
        asm volatile("" : "+v"(ln));  // keep the fragment-address arithmetic inside the slot: hoisted out of the loop, the
                                      // address registers of all slot bodies would be live everywhere (spills)
        const char* ia = slot;
        const char* ib = slot + kHalfOp;
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = frag<B_KC>(ib, wn * 64 + j * 16, ln);
#pragma unroll
        for (int i = 0; i < 8; ++i) fa[i] = frag<A_KC>(ia, wm * 128 + i * 16, ln);
    };
    // `fresh`: first half-stage of a tile - the MFMAs start from C = 0 (no accumulator clearing pass after the epilogue)
    auto mfma_frags = [&](auto fresh_c) {
        constexpr bool fresh = decltype(fresh_c)::value;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], fresh ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[i][j], 0, 0, 0);
    };

    // same-order mode: read (compiler-interleaved) and multiply one half-stage
    auto compute = [&](const char* slot, auto fresh_c) {
        constexpr bool fresh = decltype(fresh_c)::value;
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const char* ia = slot;
        const char* ib = slot + kHalfOp;
        bf16x8 gb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) gb[j] = frag<B_KC>(ib, wn * 64 + j * 16, ln);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            bf16x8 ga[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) ga[i] = frag<A_KC>(ia, wm * 128 + (half * 4 + i) * 16, ln);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[half * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        gb[j], ga[i], fresh ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[half * 4 + i][j], 0, 0, 0);
        }
    };

    // One epilogue slot: accumulator row-groups i0 .. i0+3 (16 rows x 64 columns each) of this wave, transposed through its
    // private patch so that a store instruction writes whole rows.  No workgroup barrier: only this wave touches the patch.
    auto epilogue = [&](const Cursor c, auto part_c) {
        constexpr int part = decltype(part_c)::value;
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int64_t mb = c.m0 + wm * 128, nb = c.n0 + wn * 64;
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
            const int i = part * 4 + ii;  // compile-time after unrolling `part` below
            if constexpr (EPI == CM3P_EPI_BF16) {
                // patch: 16 rows x 128 B; 8-byte unit (row r, 4 columns c4) at chunk (c4 >> 1) ^ ((r >> 1) & 7)
                const int r = ln & 15;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c4 = j * 4 + (ln >> 4);
                    const f32x4 a = acc[i][j];
                    *reinterpret_cast<uint2*>(patch + r * 128 + ((((c4 >> 1) ^ ((r >> 1) & 7))) << 4) + ((c4 & 1) << 3)) =
                        uint2{pack_bf16x2(a.x, a.y), pack_bf16x2(a.z, a.w)};
                }
                uint16_t* C = static_cast<uint16_t*>(Cv);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int rr = h * 8 + (ln >> 3), ch = ln & 7;
                    const uint4 v = *reinterpret_cast<const uint4*>(patch + rr * 128 + ((ch ^ ((rr >> 1) & 7)) << 4));
                    const int64_t m = mb + i * 16 + rr, n = nb + ch * 8;
                    if (m < M && n < N) *reinterpret_cast<uint4*>(C + m * ldc + n) = v;
                }
            } else {
                // patch: 16 rows x 256 B; 16-byte chunk c of row r at position c ^ r
                const int r = ln & 15;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c4 = j * 4 + (ln >> 4);
                    *reinterpret_cast<f32x4*>(patch + r * 256 + ((c4 ^ r) << 4)) = acc[i][j];
                }
                float* C = static_cast<float*>(Cv) + (int64_t)c.z * c_split_stride;
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    const int rr = h * 4 + (ln >> 4), ch = ln & 15;
                    f32x4 v = *reinterpret_cast<const f32x4*>(patch + rr * 256 + ((ch ^ rr) << 4));
                    const int64_t m = mb + i * 16 + rr, n = nb + ch * 4;
                    if (m < M && n < N) {
                        if constexpr (EPI == CM3P_EPI_F32_RESID) v += *reinterpret_cast<const f32x4*>(R + m * ldc + n);
                        *reinterpret_cast<f32x4*>(C + m * ldc + n) = v;
                    }
                }
            }
        }
    };

    // ---- slots -------------------------------------------------------------------------------------------------------
    // prologue: fill the ring, make half-stage 0 visible
#pragma unroll
    for (int i = 0; i < kRing; ++i) request_next();
    if (issued > 0) await_hs(0);
    slot_barrier();

    bool first = true;
    SEG_INIT();
    while (true) {
        const bool lead_act = lead.v < total;
        const bool lag_act = !first && lag.v < total;
        if (!lead_act && !lag_act) break;

        // Ring: a slot is free once BOTH halves have read what was in it.  Opposite-order mode: both read half-stage h in
        // the same slot; same-order mode: waves 4-7 read it one slot later.
        const int freed = kOpposite ? lead_consumed : lag_consumed;
        // (issuing an LDS-DMA instruction costs its wave 100-250 cycles - s_memtime stamps)
        if (!kLateReq || wm == 0) {
            while (issued < freed + kRing && request_next()) {
            }
        }
        SEG(0);
        const bool lead_computes = lead_act && lead.u < lead.nh;
        const char* slot = smem + (lead_consumed & (kRing - 1)) * kSlotBytes;  // the half-stage waves 0-3 read in this slot

        Cursor mine;  // by value, field by field: a reference picked at run time would force both cursors into memory
        mine.v = wm == 0 ? lead.v : lag.v;
        mine.u = wm == 0 ? lead.u : lag.u;
        mine.nh = wm == 0 ? lead.nh : lag.nh;
        mine.m0 = wm == 0 ? lead.m0 : lag.m0;
        mine.n0 = wm == 0 ? lead.n0 : lag.n0;
        mine.kbeg = 0;
        mine.z = wm == 0 ? lead.z : lag.z;
        const bool my_act = wm == 0 ? lead_act : lag_act;
        // one code path for both halves (the accumulators must not be written in two branches: the merge would double them)
        if constexpr (kOpposite) {
            if (wm == 0 && lead_computes) read_frags(slot);  // leading half: read, then multiply
            __builtin_amdgcn_sched_barrier(0);
            SEG(1);
            if (my_act) {
                if (mine.u == 0) mfma_frags(std::true_type{});
                else if (mine.u < mine.nh) mfma_frags(std::false_type{});
                else if (mine.u == mine.nh) epilogue(mine, std::integral_constant<int, 0>{});
                else epilogue(mine, std::integral_constant<int, 1>{});
            }
            __builtin_amdgcn_sched_barrier(0);
            SEG(my_act && mine.u >= mine.nh ? 3 : 2);
            if (wm == 1 && lead_computes) read_frags(slot);  // lagging half: multiply what it read a slot ago, then read
        } else {
            SEG(1);
            if (my_act) {
                const char* myslot = smem + ((wm == 0 ? lead_consumed : lag_consumed) & (kRing - 1)) * kSlotBytes;
                if (mine.u == 0) compute(myslot, std::true_type{});
                else if (mine.u < mine.nh) compute(myslot, std::false_type{});
                else if (mine.u == mine.nh) epilogue(mine, std::integral_constant<int, 0>{});
                else epilogue(mine, std::integral_constant<int, 1>{});
            }
            SEG(my_act && mine.u >= mine.nh ? 3 : 2);
        }
        if (kLateReq && wm == 1) {
            while (issued < freed + kRing && request_next()) {
            }
        }
        SEG(4);
        if (lag_act) {
            if (lag.u < lag.nh) ++lag_consumed;
            advance(lag);
        }
        if (lead_act) {
            if (lead.u < lead.nh) ++lead_consumed;
            advance(lead);
        }
        first = false;

        // what the leading half reads in the next slot must have landed (what the lagging half reads landed a slot earlier)
        if (lead_consumed < issued) await_hs(lead_consumed);
        SEG(5);
        slot_barrier();
        SEG(6);
    }
    SEG_FLUSH();
}

template <bool A_KC, bool B_KC>
int launch256s(const uint16_t* a, const uint16_t* b, void* C, const float* R, int64_t M, int64_t N, int64_t K, int64_t lda,
               int64_t ldb, int64_t ldc, int epi, int splits, int64_t kchunk, int64_t c_split_stride, hipStream_t s) {
    const int tiles_m = (int)((M + TM - 1) / TM), tiles_n = (int)((N + TN - 1) / TN);
    const int ntiles = tiles_m * tiles_n, total = ntiles * splits;
    static int num_cu = 0;
    if (num_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) num_cu = prop.multiProcessorCount;
        if (num_cu <= 0) num_cu = 256;
    }
    const dim3 grid(total < num_cu ? total : num_cu);
#define CM3P_G256S(E)                                                                                                    \
    {                                                                                                                    \
        static bool attr_set = false;                                                                                    \
        if (!attr_set) {                                                                                                 \
            if (hipFuncSetAttribute((const void*)gemm256s_kernel<A_KC, B_KC, E>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                    kLds) != hipSuccess)                                                                 \
                return CM3P_ERR_LAUNCH;                                                                                  \
            attr_set = true;                                                                                             \
        }                                                                                                                \
        gemm256s_kernel<A_KC, B_KC, E><<<grid, 512, kLds, s>>>(a, b, C, R, M, N, K, lda, ldb, ldc, tiles_n, ntiles, total, kchunk, c_split_stride); \
    }
    switch (epi) {
        case CM3P_EPI_BF16: CM3P_G256S(CM3P_EPI_BF16) break;
        case CM3P_EPI_F32: CM3P_G256S(CM3P_EPI_F32) break;
        case CM3P_EPI_F32_RESID: CM3P_G256S(CM3P_EPI_F32_RESID) break;
        default: return CM3P_ERR_INVALID;
    }
#undef CM3P_G256S
#ifdef CM3P_STAMPS
    {
        unsigned long long h[16];
        (void)hipDeviceSynchronize();
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_seg), sizeof(h));
        const double n = (double)h[8];
        fprintf(stderr, "[stamps] gemm256s M=%ld N=%ld K=%ld epi=%d waves=%llu  kcycles per wave: request=%.1f lead-read=%.1f mfma=%.1f epilogue=%.1f lag-read=%.1f await=%.1f barrier=%.1f\n",
                (long)M, (long)N, (long)K, epi, h[8], h[0] / n / 1e3, h[1] / n / 1e3, h[2] / n / 1e3, h[3] / n / 1e3, h[4] / n / 1e3, h[5] / n / 1e3, h[6] / n / 1e3);
        unsigned long long z[16] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_seg), z, sizeof(z));
    }
#endif
    return CM3P_OK;
}

}  // namespace

// Internal entry used by cm3p_gemm_bf16 (gemm.hip); not part of the public header.
int cm3p_gemm256s_dispatch(const void* A, const void* B, void* C, const float* R, int64_t M, int64_t N, int64_t K, int64_t lda,
                           int64_t ldb, int64_t ldc, int a_kc, int b_kc, int epi, int splits, int64_t kchunk,
                           int64_t c_split_stride, hipStream_t s) {
    const uint16_t* a = static_cast<const uint16_t*>(A);
    const uint16_t* b = static_cast<const uint16_t*>(B);
    if (a_kc && b_kc) return launch256s<true, true>(a, b, C, R, M, N, K, lda, ldb, ldc, epi, splits, kchunk, c_split_stride, s);
    if (a_kc) return launch256s<true, false>(a, b, C, R, M, N, K, lda, ldb, ldc, epi, splits, kchunk, c_split_stride, s);
    if (b_kc) return launch256s<false, true>(a, b, C, R, M, N, K, lda, ldb, ldc, epi, splits, kchunk, c_split_stride, s);
    return launch256s<false, false>(a, b, C, R, M, N, K, lda, ldb, ldc, epi, splits, kchunk, c_split_stride, s);
}
