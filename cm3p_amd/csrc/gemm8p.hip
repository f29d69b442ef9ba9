// 256 x 256 x 64 bf16 MFMA GEMM on a half-tile ring: the "8-phase" schedule of cdna_hip_programming.md (256^2 8-phase template),
// re-derived for this library's contract (persistent workgroups over work items, the operand layouts and epilogues of gemm256.hip).
//
// What differs from gemm256.hip (one 64 KiB stage per k-tile, `s_waitcnt vmcnt(0)` + barrier every k-step):
//   * LDS holds a ring of eight 16 KiB HALF-tiles: {A-lo, A-hi, B-lo, B-hi} x two k-tile parities.  A k-tile is computed in
//     four phases of 16 MFMAs (one quadrant of the wave's 128 x 64 output each); a phase first requests the fragments it needs
//     (12 / 4 / 8 / 0 ds_read_b128) and issues the LDS-DMA of ONE half-tile of a later k-tile, then passes a barrier and runs
//     its MFMAs.  Three half-tiles are always in flight across the barriers behind ONE counted wait per k-tile
//     (`s_waitcnt vmcnt(6)`, never 0 inside the loop).
//   * The two waves of a SIMD (w and w + 4: wave rows 0 and 1) run one barrier apart: while one is in its MFMA cluster the other
//     reads fragments and issues DMA.
//   * The ring does not drain between work items: the half-tile stream simply continues with the next item's k-tiles, and the
//     epilogue goes through a wave-private 4 KiB LDS buffer with no workgroup barrier.
//
// Half-tile contents (so that a wave's output stays the contiguous 128 x 64 block of gemm256.hip):
//   A-lo = tile rows {0..63, 128..191}, A-hi = {64..127, 192..255}: wave row wr reads LDS rows wr*64.. of each = tile rows wr*128 + {0..63 | 64..127}
//   B-lo = tile cols {g*64 + 0..31}, B-hi = {g*64 + 32..63}, g = 0..3: wave column wc reads LDS rows wc*32.. of each.
// LDS image of a half-tile: [128 rows][64 k] bf16, 128-byte rows, 16-byte chunk index XOR (row & 7) (conflict-free ds_read_b128);
// LDS-DMA writes lane-linearly, so the XOR is applied to each lane's SOURCE chunk.
//
// Synchronisation (p = phase, two barriers per phase; group 1 = waves 4-7 runs one barrier behind group 0):
//   RAW  LDS-DMA data is ordered for a ds_read only by the issuing waves' counted vmcnt wait followed by a barrier the reader has
//        passed.  The wait sits in phase 4 (before its first barrier) and covers the whole NEXT k-tile; reads start in phase 1.
//   WAR  a slot is restaged >= 2 phases after its last ds_read (A-lo: read p1, restaged p3; B-hi: p2 -> p4; A-hi: p3 -> p1 of the
//        next k-tile), except B-lo (read p1, restaged p2): its four reads are issued first and retired by `lgkmcnt(8)` before the
//        first barrier of p1.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

#ifndef CM3P_G8P_ABL
#define CM3P_G8P_ABL 0  // timing-only probes (results invalid; tools/ubench/gemm8p_ablate.sh): 1 no global stores / residual loads in the
                        // epilogue, 2 no epilogue at all, 4 accumulators not zeroed, 8 no counted vmcnt wait in phase 4,
                        // 16 every second workgroup of an XCD starts CM3P_G8P_DELAY x ~0.5 us late (de-phasing probe; results valid),
                        // 32 no B-lo fragment reads in phase 1 (stale registers: what moving them out of the longest phase could buy),
                        // 64 no LDS-DMA in the k-loop (stale tiles: the cost of issuing the operand stream),
                        // 128 the RoPE instance does not read its cos / sin tables (what their traffic costs)
#endif
#ifndef CM3P_G8P_DELAY
#define CM3P_G8P_DELAY 10
#endif

namespace {

constexpr int kHalf = 16384;                 // one half-tile
constexpr int kRing = 8 * kHalf;             // 128 KiB
constexpr int kEpiWave = 4096;               // wave-private epilogue buffer
constexpr int kLds8p = kRing + 8 * kEpiWave;  // 160 KiB
enum { kAL = 0, kAH = 1, kBL = 2, kBH = 3 };
__host__ __device__ constexpr int slot_off(int par, int kind) { return (par * 4 + kind) * kHalf; }

// one 1-KiB LDS-DMA piece: global address = sbase (SGPR pair) + voff (VGPR, bytes), LDS address = ldsw + IMM + 16 * lane
template <int IMM, int AUD = CM3P_AUD_A>
__device__ __forceinline__ void glds_s(uint32_t voff, const char* sbase, uint32_t ldsw) {
    CM3P_AUDIT(AUD, sbase + voff, 16);
    if constexpr (CM3P_G8P_ABL & 64) {
        asm volatile("" ::"v"(voff), "s"(sbase), "s"(ldsw));
        return;
    }
    asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(ldsw), "n"(IMM)
                 : "memory", "m0", "scc");
}

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;  // (HIP's uint4 is a struct: arrays of it land in scratch)


#if CM3P_G8P_ABL & 1
#define G8P_GLOBAL(stmt) asm volatile("" ::: "memory")
#else
#define G8P_GLOBAL(stmt) stmt
#endif
// Compiler-only fence between the two halves of a cross-lane exchange through the wave's LDS buffer.  The hardware executes a wave's
// LDS instructions in order, but to the compiler every lane is a thread of its own: without a fence it deletes a lane's stores that
// the same lane never reads back (dead-store elimination: that happened on edge tiles, where the readers are masked) and may move
// the next block's stores above this block's loads.
#define G8P_LANE_XCHG_FENCE() asm volatile("" ::: "memory")
#define G8P_WAIT_LGKM(n) asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory")
#define G8P_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

// One GEMM operand (A: rows of the output tile, B: its columns) as the staging stream and the fragment reader see it.
//   KC  k-contiguous: element (idx, k) at X[idx * ld + k].  Half-tile image [128 idx][64 k], 128-byte rows, 16-byte chunk ^ (idx & 7).
//   !KC k-strided:    element (idx, k) at X[k * ld + idx].  Half-tile image [64 k][128 idx], 256-byte rows, 32-byte segment index
//       XOR f(k), f(k) = (k & 3) | ((k >> 3) & 1) << 2, read with ds_read_b64_tr_b16 (the layout rule of gemm256.hip at half width).
// Which 128 of the tile's 256 idx a half-tile holds: position j of half h is tile idx (j / G) * 2G + h * G + j % G with G = 64 (A) or
// 32 (B), so that wave row wr / wave column wc owns a contiguous 128 x 64 block of the output.
template <bool KC, bool IS_A>
struct Operand {
    static constexpr int G = IS_A ? 64 : 32;
    const char* p[2][2];  // [half][piece]: wave-uniform source base of this wave's two 1-KiB pieces (KS: the same for both halves)
    uint32_t voff[2];     // [half]: per-lane byte offset added to the base (KC: the same for both halves)
    int64_t kstep;        // bytes from one k-tile to the next
    int fbase;            // per-lane LDS offset of the fragment reads

    __device__ __forceinline__ void init(int64_t ld, int wid, int lane) {
        const int wx = IS_A ? (wid >> 2) : (wid & 3);
        if constexpr (KC) {
            kstep = 128;
            voff[0] = voff[1] = (uint32_t)((lane >> 3) * (int)ld * 2 + (((lane & 7) ^ (lane >> 3)) << 4));
            fbase = (wx * G + (lane & 15)) * 128 + (((lane >> 4) ^ (lane & 7)) << 4);
        } else {
            kstep = 128 * ld;
            const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
            fbase = (8 * g + q) * 256 + 8 * pp + (((wx * (G / 16)) ^ (q | ((g & 1) << 2))) << 5);
        }
    }
    // item = (idx0, kbeg); KC: clamp whole 8-row pieces, KS: clamp each lane's 8 idx (extent % 8 == 0: a piece / a lane's 16 bytes
    // is inside or outside as a whole; what is loaded for the outside is never stored)
    __device__ __forceinline__ void setup(const uint16_t* X, int64_t ld, int64_t idx0, int64_t extent, int64_t kbeg, int wid, int lane) {
        if constexpr (KC) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int j = 16 * wid + 8 * i;
                    int64_t idx = idx0 + (j / G) * 2 * G + h * G + j % G;
                    if (idx > extent - 8) idx = extent - 8;
                    p[h][i] = reinterpret_cast<const char*>(X + idx * ld + kbeg);
                }
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) p[0][i] = p[1][i] = reinterpret_cast<const char*>(X + (kbeg + 8 * wid + 4 * i) * ld + idx0);
            // lane l brings LDS bytes [16 l, 16 l + 16) of the piece: k row l >> 4, 16-byte chunk c = l & 15 = (segment' << 1) | half
            const int kl = lane >> 4, c = lane & 15;
            const int seg = (c >> 1) ^ (kl | ((wid & 1) << 2));  // f(k) for k = 8 wid + 4 i + kl
            const int j = seg * 16 + (c & 1) * 8;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                int64_t idx = idx0 + (j / G) * 2 * G + h * G + j % G;
                if (idx > extent - 8) idx = extent - 8;
                voff[h] = (uint32_t)(kl * (int)ld * 2 + (int)(idx - idx0) * 2);
            }
        }
    }
    __device__ __forceinline__ void advance() {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 2; ++i) p[h][i] += kstep;
    }
    template <int SLOT>
    __device__ __forceinline__ void stage(int h, uint32_t lds) const {
        glds_s<SLOT, IS_A ? CM3P_AUD_A : CM3P_AUD_B>(voff[h], p[h][0], lds);
        glds_s<SLOT + 1024, IS_A ? CM3P_AUD_A : CM3P_AUD_B>(voff[h], p[h][1], lds);
    }
    // fragment t (16 idx) of this wave's share of a half-tile, k sub-step kk (32 deep); `off` = parity offset + slot offset
    __device__ __forceinline__ bf16x8 frag(const char* smem, int off, int t, int kk) const {
        if constexpr (KC) {
            return *reinterpret_cast<const bf16x8*>(smem + (fbase ^ (kk ? 64 : 0)) + off + t * 2048);
        } else {
            const char* a = smem + (fbase ^ (t << 5)) + off + kk * 8192;
            return cat_bf16x4(lds_read_tr16(a), lds_read_tr16(a + 1024));
        }
    }
};

// REBAL: the B-lo fragments of k-tile t+1 are read in phase 4 of k-tile t (which has no fragment reads of its own) instead of in
// phase 1 of t+1, the longest memory section (12 reads + DMA + the B-lo retire wait): r03 probe, -7 % on the long-K shapes with those
// reads gone.  The two B register sets then swap roles every k-tile, so the k-loop is unrolled by two and every work item must
// have an even number of k-tiles (the dispatcher sends other shapes to the plain instance).
template <bool A_KC, bool B_KC, int EPI, bool REBAL>
__global__ __launch_bounds__(512, 2) void gemm8p_kernel(const uint16_t* __restrict__ A, const uint16_t* __restrict__ B, void* __restrict__ Cv,
                                                        const float* R, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb,
                                                        int64_t ldc, int tiles_n, int ntiles, int total, int64_t kchunk,
                                                        int64_t c_split_stride, RopeArgs rope, BatchArgs bt, int batched) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, lane_ = lane;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3;
    const uint32_t ldsw = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem) + wid * 2048;

    // work item v = (k-split z, tile) in the XCD-aware bijective order of gemm256.hip (speed only)
    const int q8 = total / 8, r8 = total % 8;
    auto decode = [&](int v, int64_t& m0, int64_t& n0, int& z) {
        const int xcd = v % 8;
        const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + v / 8;
        z = swz / ntiles;
        const int t = swz - z * ntiles;
        m0 = (int64_t)(t / tiles_n) * 256;
        n0 = (int64_t)(t % tiles_n) * 256;
    };

    // ---- staging stream: runs two k-tiles ahead of the MFMAs and straight on into the next work item
    Operand<A_KC, true> oa;
    Operand<B_KC, false> ob;
    oa.init(lda, wid, lane);
    ob.init(ldb, wid, lane);
    const char* sah[2];  // A-hi one k-tile behind (it is staged in phase 1 of the following k-tile)
    uint32_t vah;
    int sv = blockIdx.x, skt = 0, snk = 0;
    bool sdone = false;  // the stream has passed this workgroup's last k-tile: the last k-tile is re-staged (never read; keeps the vmcnt pattern fixed)
    auto stream_setup = [&](int v) {
        int64_t m0, n0;
        int z;
        decode(v, m0, n0, z);
        // z: the k-split of a plain GEMM, or (batched) the matrix of a strided batch
        const int64_t kbeg = batched ? 0 : (int64_t)z * kchunk;
        snk = (int)((min(K, kbeg + kchunk) - kbeg) / 64);
        oa.setup(A + (batched ? z * bt.a_stride : 0), lda, m0, M, kbeg, wid, lane);
        ob.setup(B + (batched ? z * bt.b_stride : 0), ldb, n0, N, kbeg, wid, lane);
    };
    auto stream_advance = [&]() {
        sah[0] = oa.p[1][0];
        sah[1] = oa.p[1][1];
        vah = oa.voff[1];
        if (sdone) return;
        ++skt;
        if (skt < snk) {
            oa.advance();
            ob.advance();
        } else {
            skt = 0;
            sv += gridDim.x;
            if (sv < total) stream_setup(sv);
            else sdone = true;
        }
    };

    f32x4 acc[8][4];

    if constexpr (CM3P_G8P_ABL & 16) {
        if ((blockIdx.x >> 3) & 1)
            for (int i = 0; i < CM3P_G8P_DELAY * 8; ++i) __builtin_amdgcn_s_sleep(2);  // 2 x 64 cycles x 8 ~ 0.5 us per unit
    }
    // prologue: stream k-tiles 0 (all four halves) and 1 (B-lo, A-lo, B-hi); A-hi of k-tile 1 follows in phase 1 of k-tile 0
    stream_setup(sv);
    oa.template stage<slot_off(0, kAL)>(0, ldsw);
    ob.template stage<slot_off(0, kBL)>(0, ldsw);
    ob.template stage<slot_off(0, kBH)>(1, ldsw);
    oa.template stage<slot_off(0, kAH)>(1, ldsw);
    stream_advance();
    ob.template stage<slot_off(1, kBL)>(0, ldsw);
    oa.template stage<slot_off(1, kAL)>(0, ldsw);
    ob.template stage<slot_off(1, kBH)>(1, ldsw);
    stream_advance();  // sah = A-hi of k-tile 1, state = k-tile 2
    G8P_WAIT_VM(6);
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();  // the stagger: group 1 runs one barrier behind

    // FIRST: the first k-tile of a work item starts its accumulators from the MFMA's zero operand instead of 128 v_mov per tile.
    // The MFMAs are inline assembly with the accumulator tied to itself ("+v"): with several inlined copies of the k-tile body
    // (first / steady, and the role-swapped pair of REBAL) the compiler otherwise gives every copy its own accumulator registers,
    // moves them with D != C MFMAs and spills inside the clusters.  hipcc pads no hazards for an asm statement: operands come
    // from ds_reads behind an explicit lgkmcnt(0), an accumulator is re-read as C eight MFMAs later (same vDst: forwarded by the
    // hardware), and the epilogue reads the results behind explicit s_nops (matrix-result -> VALU / LDS read: up to 18 wait states).
    auto mma = [&](auto Fc, auto Ic, auto Jc, const bf16x8 (&fa)[4][2], const bf16x8 (&fb)[2][2]) {
        constexpr int I = decltype(Ic)::value, J = decltype(Jc)::value;
        constexpr bool FIRST = decltype(Fc)::value && !(CM3P_G8P_ABL & 4);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    f32x4& c = acc[I * 4 + mt][J * 2 + nt];
                    if constexpr (FIRST) {
                        if (kk == 0) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(c) : "v"(fb[nt][kk]), "v"(fa[mt][kk]));
                        else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(fb[nt][kk]), "v"(fa[mt][kk]));
                    } else {
                        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(fb[nt][kk]), "v"(fa[mt][kk]));
                    }
                }
        __builtin_amdgcn_s_setprio(0);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    // (one code path for both k-tile parities - the parity is a run-time 64 KiB offset: accumulators written in two branches
    //  that merge are duplicated by the compiler and spilled)
    // fbl / fbh: the two B register sets.  !REBAL: B-lo / B-hi of this k-tile, both read here.  REBAL: fbl arrives holding this
    // k-tile's B-lo (read during the previous k-tile) and fbh leaves holding the next k-tile's.
    auto ktile = [&](int par, auto Fc, bf16x8 (&fbl)[2][2], bf16x8 (&fbh)[2][2]) {
        const int pofs = par << 16;                                         // this k-tile's four slots
        const uint32_t lds_p = ldsw + pofs, lds_q = ldsw + (pofs ^ 65536);  // LDS-DMA bases: this parity / the other one
        bf16x8 fa[4][2];
        // ---- phase 1: quadrant (A-lo, B-lo)
        if constexpr (!REBAL) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    if constexpr (CM3P_G8P_ABL & 32) asm volatile("" : "=v"(fbl[nt][kk]));
                    else fbl[nt][kk] = ob.frag(smem, pofs + slot_off(0, kBL), nt, kk);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) fa[mt][kk] = oa.frag(smem, pofs + slot_off(0, kAL), mt, kk);
        __builtin_amdgcn_sched_barrier(0);
        glds_s<slot_off(0, kAH)>(vah, sah[0], lds_q);
        glds_s<slot_off(0, kAH) + 1024>(vah, sah[1], lds_q);
        // !REBAL: the B-lo reads (issued first) are done, so B-lo may be restaged in the next phase: at most as many LDS operations
        // outstanding as were issued behind them (8 ds_read_b128 or 16 ds_read_b64_tr_b16; the counter holds 15).  REBAL: B-lo was
        // read three phases ago.
        if constexpr (!REBAL) {
            if constexpr (A_KC) G8P_WAIT_LGKM(8);
            else G8P_WAIT_LGKM(15);
        }
        __builtin_amdgcn_s_barrier();
        G8P_WAIT_LGKM(0);
        __builtin_amdgcn_sched_barrier(0);
        mma(Fc, I0{}, I0{}, fa, fbl);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        // ---- phase 2: (A-lo, B-hi)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) fbh[nt][kk] = ob.frag(smem, pofs + slot_off(0, kBH), nt, kk);
        __builtin_amdgcn_sched_barrier(0);
        ob.template stage<slot_off(0, kBL)>(0, lds_p);
        __builtin_amdgcn_s_barrier();
        G8P_WAIT_LGKM(0);
        __builtin_amdgcn_sched_barrier(0);
        mma(Fc, I0{}, I1{}, fa, fbh);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        // ---- phase 3: (A-hi, B-hi)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) fa[mt][kk] = oa.frag(smem, pofs + slot_off(0, kAH), mt, kk);
        __builtin_amdgcn_sched_barrier(0);
        oa.template stage<slot_off(0, kAL)>(0, lds_p);
        __builtin_amdgcn_s_barrier();
        G8P_WAIT_LGKM(0);
        __builtin_amdgcn_sched_barrier(0);
        mma(Fc, I1{}, I1{}, fa, fbh);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        // ---- phase 4: (A-hi, B-lo); the counted wait retires every half-tile of the next k-tile
        if constexpr (REBAL) {
            // B-lo of the NEXT k-tile (other parity; it landed before this k-tile began: retired by the previous k-tile's counted
            // wait) into the set B-hi has just left; its slot is restaged two phases after the next k-tile starts
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) fbh[nt][kk] = ob.frag(smem, (pofs ^ 65536) + slot_off(0, kBL), nt, kk);
            __builtin_amdgcn_sched_barrier(0);
        }
        ob.template stage<slot_off(0, kBH)>(1, lds_p);
        stream_advance();
        if constexpr (!(CM3P_G8P_ABL & 8)) G8P_WAIT_VM(6);
        __builtin_amdgcn_s_barrier();
        if constexpr (REBAL) G8P_WAIT_LGKM(0);
        __builtin_amdgcn_sched_barrier(0);
        mma(Fc, I1{}, I0{}, fa, fbl);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
    };

    char* ebuf = smem + kRing + wid * kEpiWave;

    // ---- epilogue: per wave, 16 rows at a time through the wave's own 4 KiB of LDS (fragment layout -> whole rows -> 16-byte
    // stores of full 128-byte lines), no workgroup barrier.  FULL: the tile lies inside the matrix, nothing is guarded.
    // Global addresses are a wave-uniform tile origin plus a 32-bit lane offset.
    auto epilogue = [&](auto Fc, int64_t m0, int64_t n0, int z) {
        constexpr bool FULL = decltype(Fc)::value;
        // the lane id is laundered so that the epilogue's address arithmetic is not hoisted out of the work-item loop and kept
        // alive (in registers the k-loop needs) across it
        int lane = lane_;
        asm volatile("s_nop 15\n\ts_nop 7" : "+v"(lane));  // (and the asm MFMAs' last results are 24 wait states old before anything reads them)
        const int64_t mw = m0 + wr * 128, nw = n0 + wc * 64;
        if constexpr (CM3P_G8P_ABL & 2) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(acc[i][j]));
        } else if constexpr (EPI == CM3P_EPI_BF16 || EPI == CM3P_EPI_BF16_ROPE) {
            char* Cb = reinterpret_cast<char*>(static_cast<uint16_t*>(Cv) + mw * ldc + nw);
            const uint32_t ldcb = (uint32_t)ldc * 2;
            const bool rotate = (EPI == CM3P_EPI_BF16_ROPE) && nw < rope.ncols;  // a wave's 64 columns are one head
            const float qs = (EPI == CM3P_EPI_BF16_ROPE && nw < rope.q_cols) ? rope.q_scale : 1.f;
            uint32_t prow0 = 0;
            if constexpr (EPI == CM3P_EPI_BF16_ROPE) prow0 = rope.per_batch ? 0u : (uint32_t)mw % (uint32_t)rope.S;  // M < 2^31 (checked by the caller)
            // rotary tables of the lane's row (row = lane >> 2 of each 16-row step, dims 8 dc .. 8 dc + 7): the rows of step i4 + 1 are
            // requested before step i4 rotates (r05: requested where they were used, each of the eight steps of this exposed epilogue
            // waited out an L2 round trip - the RoPE instance took 473 us where the plain one takes 291)
            f32x4 tc0, tc1, ts0, ts1;
            auto tab_load = [&](int i4, f32x4& c0, f32x4& c1, f32x4& s0, f32x4& s1) {
                const int row = lane >> 2, dc = lane & 3;
                int64_t prow;
                if (rope.per_batch) prow = mw + i4 * 16 + row;
                else {
                    uint32_t x = prow0 + i4 * 16 + row;
                    if (rope.S >= 256) x = x >= (uint32_t)rope.S ? x - (uint32_t)rope.S : x;
                    else x %= (uint32_t)rope.S;
                    prow = x;
                }
                if (!FULL && mw + i4 * 16 + row >= M) prow = 0;  // (rows past the matrix: any table row, the result is not stored)
                const float* cr = rope.cos + prow * 32 + dc * 8;
                const float* sr = rope.sin + prow * 32 + dc * 8;
                if constexpr (CM3P_G8P_ABL & 128) {
                    asm volatile("" : "=v"(c0), "=v"(c1), "=v"(s0), "=v"(s1) : "v"(cr), "v"(sr));
                } else {
                    c0 = *reinterpret_cast<const f32x4*>(cr), c1 = *reinterpret_cast<const f32x4*>(cr + 4);
                    s0 = *reinterpret_cast<const f32x4*>(sr), s1 = *reinterpret_cast<const f32x4*>(sr + 4);
                }
            };
            if constexpr (EPI == CM3P_EPI_BF16_ROPE) {
                if (rotate) tab_load(0, tc0, tc1, ts0, ts1);
            }
#pragma unroll
            for (int i4 = 0; i4 < 8; ++i4) {
                char* eb = ebuf + (i4 & 1) * 2048;
#pragma unroll
                for (int j4 = 0; j4 < 4; ++j4) {
                    const f32x4 a = acc[i4][j4];
                    const int row = lane & 15, s8 = (j4 * 4 + (lane >> 4)) ^ ((row & 7) << 1);
                    *reinterpret_cast<uint2*>(eb + row * 128 + s8 * 8) = uint2{pack_bf16x2(a.x, a.y), pack_bf16x2(a.z, a.w)};
                }
                G8P_LANE_XCHG_FENCE();
                if (rotate) {
                    // rotary embedding at store time on the bf16-rounded projection (what the reference's autocast path computes):
                    // a lane takes (row, dims d..d+7 and d+32..d+39) of the wave's head - one table read serves both chunks - and
                    // writes the rotated pair back IN PLACE (each chunk of the staged rows belongs to exactly one lane); the rows
                    // then leave through the common path below as whole 128-byte lines.  (Storing the two chunks straight from
                    // here left two half-line stores per row and the L2 fetched every half-written line: 1.6 GB of traffic for
                    // 0.8 GB, r03 PMC; rotating every pair twice so that a lane owns one chunk doubled the exposed VALU work and
                    // was slower still.)
                    const int row = lane >> 2, dc = lane & 3;
                    char* pa = eb + row * 128 + ((dc ^ (row & 7)) << 4);
                    char* pb = eb + row * 128 + (((dc + 4) ^ (row & 7)) << 4);
                    const u32x4 xa = *reinterpret_cast<const u32x4*>(pa);
                    const u32x4 xb = *reinterpret_cast<const u32x4*>(pb);
                    const f32x4 c0 = tc0, c1 = tc1, s0 = ts0, s1 = ts1;
                    if (i4 < 7) tab_load(i4 + 1, tc0, tc1, ts0, ts1);
                    {
                        const float cs[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
                        const float sn[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
                        u32x4 oa, ob;
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const float a0 = bf16lo(xa[t]), a1 = bf16hi(xa[t]), b0 = bf16lo(xb[t]), b1 = bf16hi(xb[t]);
                            oa[t] = pack_bf16x2(qs * (a0 * cs[2 * t] - b0 * sn[2 * t]), qs * (a1 * cs[2 * t + 1] - b1 * sn[2 * t + 1]));
                            ob[t] = pack_bf16x2(qs * (b0 * cs[2 * t] + a0 * sn[2 * t]), qs * (b1 * cs[2 * t + 1] + a1 * sn[2 * t + 1]));
                        }
                        *reinterpret_cast<u32x4*>(pa) = oa;
                        *reinterpret_cast<u32x4*>(pb) = ob;
                    }
                    G8P_LANE_XCHG_FENCE();
                }
                {
                    const int r0 = lane >> 3, r1 = 8 + (lane >> 3), ch = lane & 7;
                    const u32x4 x0 = *reinterpret_cast<const u32x4*>(eb + r0 * 128 + ((ch ^ (r0 & 7)) << 4));
                    const u32x4 x1 = *reinterpret_cast<const u32x4*>(eb + r1 * 128 + ((ch ^ (r1 & 7)) << 4));
                    G8P_LANE_XCHG_FENCE();
                    const bool col_ok = FULL || nw + ch * 8 < N;
                    if (FULL || (col_ok && mw + i4 * 16 + r0 < M)) G8P_GLOBAL(gstore16<(CM3P_NT & 1) != 0>(Cb + (uint32_t)(i4 * 16 + r0) * ldcb + ch * 16, x0));
                    if (FULL || (col_ok && mw + i4 * 16 + r1 < M)) G8P_GLOBAL(gstore16<(CM3P_NT & 1) != 0>(Cb + (uint32_t)(i4 * 16 + r1) * ldcb + ch * 16, x1));
                    if constexpr (CM3P_G8P_ABL & 1) asm volatile("" ::"v"(x0), "v"(x1));
                }
            }
        } else if constexpr (EPI == CM3P_EPI_BF16_GEGLU) {
            // The Wi projection with GeGLU in its store phase (forward-only calls: nothing keeps h and g).  The weight rows were
            // interleaved on the host so that a wave's 64 columns are [h_j .. h_j+31 | g_j .. g_j+31]: the rows are staged as bf16
            // (the rounding the unfused path applies when it stores h and g), a lane takes 8 h and the 8 g of the same columns,
            // and 4 lanes write a row's 64 bytes of gelu_erf(h) * g.  C is [M, N / 2] bf16, ldc its row pitch.
            char* Cb = reinterpret_cast<char*>(static_cast<uint16_t*>(Cv) + mw * ldc + nw / 2);
            const uint32_t ldcb = (uint32_t)ldc * 2;
#pragma unroll
            for (int i4 = 0; i4 < 8; ++i4) {
                char* eb = ebuf + (i4 & 1) * 2048;
#pragma unroll
                for (int j4 = 0; j4 < 4; ++j4) {
                    const f32x4 a = acc[i4][j4];
                    const int row = lane & 15, s8 = (j4 * 4 + (lane >> 4)) ^ ((row & 7) << 1);
                    *reinterpret_cast<uint2*>(eb + row * 128 + s8 * 8) = uint2{pack_bf16x2(a.x, a.y), pack_bf16x2(a.z, a.w)};
                }
                G8P_LANE_XCHG_FENCE();
                const int row = lane >> 2, dc = lane & 3;
                const u32x4 xh = *reinterpret_cast<const u32x4*>(eb + row * 128 + ((dc ^ (row & 7)) << 4));
                const u32x4 xg = *reinterpret_cast<const u32x4*>(eb + row * 128 + (((dc + 4) ^ (row & 7)) << 4));
                G8P_LANE_XCHG_FENCE();
                u32x4 y;
#pragma unroll
                for (int t = 0; t < 4; ++t) {  // (gelu_erf2: common.h, the function cm3p_geglu_fwd evaluates - same bits)
                    const f32x2 gl = gelu_erf2(f32x2{bf16lo(xh[t]), bf16hi(xh[t])});
                    y[t] = pack_bf16x2(gl.x * bf16lo(xg[t]), gl.y * bf16hi(xg[t]));
                }
                if (FULL || (mw + i4 * 16 + row < M && nw + dc * 8 < N))
                    G8P_GLOBAL(gstore16<(CM3P_NT & 1) != 0>(Cb + (uint32_t)(i4 * 16 + row) * ldcb + dc * 16, y));
                if constexpr (CM3P_G8P_ABL & 1) asm volatile("" ::"v"(y));
            }
        } else if constexpr (EPI == CM3P_EPI_BF16_AXPBY) {
            // C_z = bf16(alpha * acc + beta * R_z), R bf16 with C's layout (the Newton-Schulz polynomial steps of the Muon update):
            // the fp32 rows of the exchange below, combined in fp32 and rounded once (the arithmetic of gemm.hip's small-tile kernel)
            char* Cb = reinterpret_cast<char*>(static_cast<uint16_t*>(Cv) + (int64_t)z * bt.c_stride + mw * ldc + nw);
            const char* Rb = reinterpret_cast<const char*>(bt.Rb + (int64_t)z * bt.r_stride + mw * ldc + nw);
            const uint32_t ldcb = (uint32_t)ldc * 2;
            const int lrow = lane >> 4, lch = lane & 15;
            const uint32_t loff = (uint32_t)lrow * ldcb + lch * 8;
            const bool col_ok = FULL || nw + lch * 4 < N;
#pragma unroll
            for (int i4 = 0; i4 < 8; ++i4) {
                uint2 r[4];
                if (bt.Rb) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int row = i4 * 16 + u * 4;
                        r[u] = (FULL || (col_ok && mw + row + lrow < M)) ? *reinterpret_cast<const uint2*>(Rb + loff + (uint32_t)row * ldcb) : uint2{0u, 0u};
                    }
                }
#pragma unroll
                for (int j4 = 0; j4 < 4; ++j4) {
                    const int row = lane & 15, ch = j4 * 4 + (lane >> 4);
                    *reinterpret_cast<f32x4*>(ebuf + row * 256 + ((ch ^ row) << 4)) = acc[i4][j4];
                }
                G8P_LANE_XCHG_FENCE();
                f32x4 x[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int row = u * 4 + lrow;
                    x[u] = *reinterpret_cast<const f32x4*>(ebuf + row * 256 + ((lch ^ row) << 4));
                }
                G8P_LANE_XCHG_FENCE();
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    f32x4 v = x[u];
                    v *= bt.alpha;
                    if (bt.Rb) v += bt.beta * f32x4{bf16lo(r[u].x), bf16hi(r[u].x), bf16lo(r[u].y), bf16hi(r[u].y)};
                    const int row = i4 * 16 + u * 4;
                    if (FULL || (col_ok && mw + row + lrow < M))
                        G8P_GLOBAL(*reinterpret_cast<uint2*>(Cb + loff + (uint32_t)row * ldcb) = (uint2{pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w)}));
                    if constexpr (CM3P_G8P_ABL & 1) asm volatile("" ::"v"(v));
                }
            }
        } else {
            char* Cb = reinterpret_cast<char*>(static_cast<float*>(Cv) + (int64_t)z * c_split_stride + mw * ldc + nw);
            const char* Rb = reinterpret_cast<const char*>(R + mw * ldc + nw);
            const uint32_t ldcb = (uint32_t)ldc * 4;
            const int lrow = lane >> 4, lch = lane & 15;
            const uint32_t loff = (uint32_t)lrow * ldcb + lch * 16;
            const bool col_ok = FULL || nw + lch * 4 < N;
            f32x4 rnext[4];
            auto load_r = [&](int i4) {
                if constexpr (EPI == CM3P_EPI_F32_RESID && !(CM3P_G8P_ABL & 1)) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int row = i4 * 16 + u * 4;
                        rnext[u] = (FULL || (col_ok && mw + row + lrow < M)) ? gload16f<(CM3P_NT & 64) != 0>(Rb + loff + (uint32_t)row * ldcb)
                                                                             : f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                }
            };
            f32x4 bias = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (EPI == CM3P_EPI_F32_BIAS) {
                if (col_ok) bias = *reinterpret_cast<const f32x4*>(R + nw + lch * 4);
            }
            load_r(0);
#pragma unroll
            for (int i4 = 0; i4 < 8; ++i4) {
#pragma unroll
                for (int j4 = 0; j4 < 4; ++j4) {
                    const int row = lane & 15, ch = j4 * 4 + (lane >> 4);
                    *reinterpret_cast<f32x4*>(ebuf + row * 256 + ((ch ^ row) << 4)) = acc[i4][j4];
                }
                G8P_LANE_XCHG_FENCE();
                f32x4 x[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int row = u * 4 + lrow;
                    x[u] = *reinterpret_cast<const f32x4*>(ebuf + row * 256 + ((lch ^ row) << 4));
                }
                G8P_LANE_XCHG_FENCE();
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if constexpr (EPI == CM3P_EPI_F32_RESID && !(CM3P_G8P_ABL & 1)) x[u] += rnext[u];
                    if constexpr (EPI == CM3P_EPI_F32_BIAS) x[u] += bias;
                }
                if (i4 < 7) load_r(i4 + 1);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int row = i4 * 16 + u * 4;
                    if (FULL || (col_ok && mw + row + lrow < M)) G8P_GLOBAL(gstore16f<(CM3P_NT & 32) != 0>(Cb + loff + (uint32_t)row * ldcb, x[u]));
                    if constexpr (CM3P_G8P_ABL & 1) asm volatile("" ::"v"(x[u]));
                }
            }
        }
    };

    int par = 0;
    bf16x8 fbP[2][2], fbQ[2][2];
    if constexpr (REBAL) {  // B-lo of the very first k-tile (the prologue's wait and barrier(s) are behind us)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) fbP[nt][kk] = ob.frag(smem, slot_off(0, kBL), nt, kk);
    }
    for (int v = blockIdx.x; v < total; v += gridDim.x) {
        int64_t m0, n0;
        int z;
        decode(v, m0, n0, z);
        const int64_t kbeg = batched ? 0 : (int64_t)z * kchunk;
        const int nk = (int)((min(K, kbeg + kchunk) - kbeg) / 64);
        if constexpr (REBAL) {  // nk even (checked by the dispatcher): the register sets swap roles every k-tile
            ktile(par, std::true_type{}, fbP, fbQ);
            ktile(par ^ 1, std::false_type{}, fbQ, fbP);
            for (int kt = 2; kt < nk; kt += 2) {
                ktile(par, std::false_type{}, fbP, fbQ);
                ktile(par ^ 1, std::false_type{}, fbQ, fbP);
            }
        } else {
            ktile(par, std::true_type{}, fbP, fbQ);
            par ^= 1;
            for (int kt = 1; kt < nk; ++kt) {
                ktile(par, std::false_type{}, fbP, fbQ);
                par ^= 1;
            }
        }
        if (m0 + 256 <= M && n0 + 256 <= N) epilogue(std::true_type{}, m0, n0, z);
        else epilogue(std::false_type{}, m0, n0, z);
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();  // matches group 1's extra barrier
    G8P_WAIT_VM(0);                             // no LDS-DMA may outlive the workgroup's LDS allocation
}

template <bool A_KC, bool B_KC, bool REBAL>
int launch8p(const uint16_t* a, const uint16_t* b, void* C, const float* R, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb,
             int64_t ldc, int epi, int splits, int64_t kchunk, int64_t c_split_stride, hipStream_t s, RopeArgs rope, BatchArgs bt,
             int batch) {
    const int tiles_m = (int)((M + 255) / 256), tiles_n = (int)((N + 255) / 256);
    const int ntiles = tiles_m * tiles_n, total = ntiles * (batch > 0 ? batch : splits);
    const int num_cu = cm3p_num_cu();
    // cm3p_gemm8p_set_grid / CM3P_G8P_GRID: fewer workgroups leave CUs to a kernel on another stream (tools/overlap_ab.py), more than CUs
    // are dispatched as CUs come free (tools/blocker_probe.py: what to run beside RCCL).  Any value is safe: a workgroup of this kernel
    // never waits for another one (no instance has an inter-workgroup hand-off: split-K partials go to a separate reduce launch, REBAL only
    // reorders a workgroup's own LDS reads), so surplus workgroups simply start when resident ones exit.
    const int set = cm3p_gemm8p_get_grid();
    const int want = set > 0 ? set : num_cu;
    const dim3 grid(total < want ? total : want);
#define CM3P_G8P(E)                                                                                                               \
    {                                                                                                                             \
        static Cm3pDevOnce once;                                                                                                  \
        const int rc_once = once.run([&] {                                                                                        \
            return hipFuncSetAttribute((const void*)gemm8p_kernel<A_KC, B_KC, E, REBAL>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds8p) == hipSuccess \
                       ? CM3P_OK                                                                                                  \
                       : CM3P_ERR_LAUNCH;                                                                                         \
        });                                                                                                                       \
        if (rc_once != CM3P_OK) return rc_once;                                                                                   \
        gemm8p_kernel<A_KC, B_KC, E, REBAL><<<grid, 512, kLds8p, s>>>(a, b, C, R, M, N, K, lda, ldb, ldc, tiles_n, ntiles, total, kchunk, c_split_stride, rope, bt, batch > 0 ? 1 : 0); \
    }
    switch (epi) {
        case CM3P_EPI_BF16: CM3P_G8P(CM3P_EPI_BF16) break;
        case CM3P_EPI_F32: CM3P_G8P(CM3P_EPI_F32) break;
        case CM3P_EPI_F32_RESID: CM3P_G8P(CM3P_EPI_F32_RESID) break;
        case CM3P_EPI_BF16_ROPE:
            if constexpr (A_KC && B_KC) {
                CM3P_G8P(CM3P_EPI_BF16_ROPE)
                break;
            }
            return CM3P_ERR_INVALID;
        case CM3P_EPI_F32_BIAS:
            if constexpr (A_KC && B_KC) {
                CM3P_G8P(CM3P_EPI_F32_BIAS)
                break;
            }
            return CM3P_ERR_INVALID;
        case CM3P_EPI_BF16_GEGLU:
            if constexpr (A_KC && B_KC) {
                CM3P_G8P(CM3P_EPI_BF16_GEGLU)
                break;
            }
            return CM3P_ERR_INVALID;
        case CM3P_EPI_BF16_AXPBY:
            if constexpr (A_KC || !B_KC) {  // (the Muon step's three layouts)
                CM3P_G8P(CM3P_EPI_BF16_AXPBY)
                break;
            }
            return CM3P_ERR_INVALID;
        default: return CM3P_ERR_INVALID;
    }
#undef CM3P_G8P
    return CM3P_OK;
}

}  // namespace

// Internal entry used by gemm.hip; returns CM3P_ERR_INVALID for what this kernel does not cover (the caller then falls back).
int cm3p_gemm8p_dispatch(const void* A, const void* B, void* C, const float* R, int64_t M, int64_t N, int64_t K, int64_t lda,
                         int64_t ldb, int64_t ldc, int a_kc, int b_kc, int epi, int splits, int64_t kchunk, int64_t c_split_stride,
                         hipStream_t s, RopeArgs rope, BatchArgs bt, int batch) {
    // batch > 0: a strided batch of `batch` matrices (the work item's z is the matrix instead of the k-split), bf16 a x + b y epilogue
    if ((batch > 0) != (epi == CM3P_EPI_BF16_AXPBY) || (batch > 0 && (splits != 1 || kchunk != K))) return CM3P_ERR_INVALID;
    if (M % 8 != 0 || N % 8 != 0 || K % 64 != 0 || kchunk % 64 != 0) return CM3P_ERR_INVALID;
    if (lda * 2 * 8 >= (int64_t(1) << 31) || ldb * 2 * 8 >= (int64_t(1) << 31)) return CM3P_ERR_INVALID;  // 32-bit lane offsets
    const uint16_t* a = static_cast<const uint16_t*>(A);
    const uint16_t* b = static_cast<const uint16_t*>(B);
    // every work item's k-tile count even (the last k-split may be shorter): the instance whose B register sets swap roles
    const int64_t last = K - (int64_t)(splits - 1) * kchunk;
    static const bool no_rebal = [] { const char* e = getenv("CM3P_G8P_REBAL"); return e && e[0] == '0'; }();
    const bool rebal = !no_rebal && (kchunk / 64) % 2 == 0 && (splits == 1 ? (K / 64) % 2 == 0 : (last > 0 && (last / 64) % 2 == 0));
#define CM3P_G8P_GO(AK, BK)                                                                                                      \
    return rebal ? launch8p<AK, BK, true>(a, b, C, R, M, N, K, lda, ldb, ldc, epi, splits, kchunk, c_split_stride, s, rope, bt, batch)  \
                 : launch8p<AK, BK, false>(a, b, C, R, M, N, K, lda, ldb, ldc, epi, splits, kchunk, c_split_stride, s, rope, bt, batch)
    if (a_kc && b_kc) CM3P_G8P_GO(true, true);
    if (a_kc) CM3P_G8P_GO(true, false);
    if (b_kc) CM3P_G8P_GO(false, true);
    CM3P_G8P_GO(false, false);
#undef CM3P_G8P_GO
}

// timing-only ablation switches this object was built with (0 in every shipped build: cm3p_build_ablation_flags, tests/test_cabi.py)
static std::atomic<int> g_g8p_grid{-1};  // -1: not set yet, the environment is asked once
extern "C" int cm3p_gemm8p_get_grid(void) {
    int g = g_g8p_grid.load(std::memory_order_relaxed);
    if (g < 0) {
        const char* e = getenv("CM3P_G8P_GRID");
        g = e ? atoi(e) : 0;
        if (g < 0) g = 0;
        g_g8p_grid.store(g, std::memory_order_relaxed);
    }
    return g;
}
extern "C" int cm3p_gemm8p_set_grid(int workgroups) {
    if (workgroups < 0) return CM3P_ERR_INVALID;
    g_g8p_grid.store(workgroups, std::memory_order_relaxed);
    return CM3P_OK;
}

int cm3p_ablation_flags_gemm8p() { return (CM3P_G8P_ABL); }
#if CM3P_DMA_AUDIT
int cm3p_audit_set_gemm8p(void* buf) { return cm3p_audit_set_local(buf); }
#endif
