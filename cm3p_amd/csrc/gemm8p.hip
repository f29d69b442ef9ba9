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
#include <type_traits>

#include "common.h"

namespace {

constexpr int kHalf = 16384;                 // one half-tile
constexpr int kRing = 8 * kHalf;             // 128 KiB
constexpr int kEpiWave = 4096;               // wave-private epilogue buffer
constexpr int kLds8p = kRing + 8 * kEpiWave;  // 160 KiB
enum { kAL = 0, kAH = 1, kBL = 2, kBH = 3 };
__host__ __device__ constexpr int slot_off(int par, int kind) { return (par * 4 + kind) * kHalf; }

// one 1-KiB LDS-DMA piece: global address = sbase (SGPR pair) + voff (VGPR, bytes), LDS address = ldsw + IMM + 16 * lane
template <int IMM>
__device__ __forceinline__ void glds_s(uint32_t voff, const char* sbase, uint32_t ldsw) {
    asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(ldsw), "n"(IMM)
                 : "memory", "m0", "scc");
}

#define G8P_WAIT_LGKM(n) asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory")
#define G8P_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm8p_kernel(const uint16_t* __restrict__ A, const uint16_t* __restrict__ B, void* __restrict__ Cv,
                                                        const float* R, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb,
                                                        int64_t ldc, int tiles_n, int ntiles, int total, int64_t kchunk,
                                                        int64_t c_split_stride, RopeArgs rope) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
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

    // ---- staging stream: scalar (wave-uniform) source bases of this wave's 8 pieces of one k-tile; lanes add a constant offset
    const char* sa[2][2];  // [A-lo | A-hi][piece]
    const char* sb[2][2];
    const char* sah[2];    // A-hi one k-tile behind (it is staged in phase 1 of the following k-tile)
    int sv = blockIdx.x, skt = 0, snk = 0;
    bool sdone = false;  // the stream has passed this workgroup's last k-tile: the last k-tile is re-staged (never read; keeps the vmcnt pattern fixed)
    auto stream_setup = [&](int v) {
        int64_t m0, n0;
        int z;
        decode(v, m0, n0, z);
        const int64_t kbeg = (int64_t)z * kchunk;
        snk = (int)((min(K, kbeg + kchunk) - kbeg) / 64);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                int64_t row = m0 + wr * 128 + h * 64 + wc * 16 + 8 * i;
                if (row > M - 8) row = M - 8;  // rows past the edge are never stored (M % 8 == 0 on this path)
                sa[h][i] = reinterpret_cast<const char*>(A + row * lda + kbeg);
                int64_t col = n0 + (wid >> 1) * 64 + h * 32 + (wid & 1) * 16 + 8 * i;
                if (col > N - 8) col = N - 8;
                sb[h][i] = reinterpret_cast<const char*>(B + col * ldb + kbeg);
            }
    };
    auto stream_advance = [&]() {
        sah[0] = sa[1][0];
        sah[1] = sa[1][1];
        if (sdone) return;
        ++skt;
        if (skt < snk) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    sa[h][i] += 128;
                    sb[h][i] += 128;
                }
        } else {
            skt = 0;
            sv += gridDim.x;
            if (sv < total) stream_setup(sv);
            else sdone = true;
        }
    };
    const uint32_t voffa = (uint32_t)((lane >> 3) * (int)lda * 2 + (((lane & 7) ^ (lane >> 3)) << 4));
    const uint32_t voffb = (uint32_t)((lane >> 3) * (int)ldb * 2 + (((lane & 7) ^ (lane >> 3)) << 4));

    // ---- fragment addresses: LDS row r' = base row + (lane & 15), 16-byte chunk kk*4 + (lane >> 4), XOR (r' & 7)
    const int xk0 = (((lane >> 4) ^ (lane & 7)) << 4);
    const int fa_base = (wr * 64 + (lane & 15)) * 128 + xk0;  // kk = 1: the chunk index gains 4, i.e. the byte offset ^ 64
    const int fb_base = (wc * 32 + (lane & 15)) * 128 + xk0;
    // (one code path for both k-tile parities - the parity is a run-time 64 KiB offset: accumulators written in two branches
    //  that merge are duplicated by the compiler and spilled)
    auto frag_at = [&](int base, int off, int kk) -> bf16x8 {
        return *reinterpret_cast<const bf16x8*>(smem + (base ^ (kk ? 64 : 0)) + off);
    };

    f32x4 acc[8][4];

    // prologue: stream k-tiles 0 (all four halves) and 1 (B-lo, A-lo, B-hi); A-hi of k-tile 1 follows in phase 1 of k-tile 0
    stream_setup(sv);
    glds_s<slot_off(0, kAL)>(voffa, sa[0][0], ldsw);
    glds_s<slot_off(0, kAL) + 1024>(voffa, sa[0][1], ldsw);
    glds_s<slot_off(0, kBL)>(voffb, sb[0][0], ldsw);
    glds_s<slot_off(0, kBL) + 1024>(voffb, sb[0][1], ldsw);
    glds_s<slot_off(0, kBH)>(voffb, sb[1][0], ldsw);
    glds_s<slot_off(0, kBH) + 1024>(voffb, sb[1][1], ldsw);
    glds_s<slot_off(0, kAH)>(voffa, sa[1][0], ldsw);
    glds_s<slot_off(0, kAH) + 1024>(voffa, sa[1][1], ldsw);
    stream_advance();
    glds_s<slot_off(1, kBL)>(voffb, sb[0][0], ldsw);
    glds_s<slot_off(1, kBL) + 1024>(voffb, sb[0][1], ldsw);
    glds_s<slot_off(1, kAL)>(voffa, sa[0][0], ldsw);
    glds_s<slot_off(1, kAL) + 1024>(voffa, sa[0][1], ldsw);
    glds_s<slot_off(1, kBH)>(voffb, sb[1][0], ldsw);
    glds_s<slot_off(1, kBH) + 1024>(voffb, sb[1][1], ldsw);
    stream_advance();  // sah = A-hi of k-tile 1, state = k-tile 2
    G8P_WAIT_VM(6);
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();  // the stagger: group 1 runs one barrier behind

    auto mma = [&](auto Ic, auto Jc, const bf16x8 (&fa)[4][2], const bf16x8 (&fb)[2][2]) {
        constexpr int I = decltype(Ic)::value, J = decltype(Jc)::value;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    acc[I * 4 + mt][J * 2 + nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[nt][kk], fa[mt][kk], acc[I * 4 + mt][J * 2 + nt], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    auto ktile = [&](int par) {
        const int pofs = par << 16;                            // this k-tile's four slots
        const uint32_t lds_p = ldsw + pofs, lds_q = ldsw + (pofs ^ 65536);  // LDS-DMA bases: this parity / the other one
        const int fa_p = fa_base + pofs, fb_p = fb_base + pofs;
        bf16x8 fa[4][2], fbl[2][2], fbh[2][2];
        // ---- phase 1: quadrant (A-lo, B-lo)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) fbl[nt][kk] = frag_at(fb_p, slot_off(0, kBL) + nt * 2048, kk);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) fa[mt][kk] = frag_at(fa_p, slot_off(0, kAL) + mt * 2048, kk);
        __builtin_amdgcn_sched_barrier(0);
        glds_s<slot_off(0, kAH)>(voffa, sah[0], lds_q);
        glds_s<slot_off(0, kAH) + 1024>(voffa, sah[1], lds_q);
        G8P_WAIT_LGKM(8);  // the four B-lo reads (issued first) are done: B-lo may be restaged in the next phase
        __builtin_amdgcn_s_barrier();
        G8P_WAIT_LGKM(0);
        __builtin_amdgcn_sched_barrier(0);
        mma(I0{}, I0{}, fa, fbl);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        // ---- phase 2: (A-lo, B-hi)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) fbh[nt][kk] = frag_at(fb_p, slot_off(0, kBH) + nt * 2048, kk);
        __builtin_amdgcn_sched_barrier(0);
        glds_s<slot_off(0, kBL)>(voffb, sb[0][0], lds_p);
        glds_s<slot_off(0, kBL) + 1024>(voffb, sb[0][1], lds_p);
        __builtin_amdgcn_s_barrier();
        G8P_WAIT_LGKM(0);
        __builtin_amdgcn_sched_barrier(0);
        mma(I0{}, I1{}, fa, fbh);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        // ---- phase 3: (A-hi, B-hi)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) fa[mt][kk] = frag_at(fa_p, slot_off(0, kAH) + mt * 2048, kk);
        __builtin_amdgcn_sched_barrier(0);
        glds_s<slot_off(0, kAL)>(voffa, sa[0][0], lds_p);
        glds_s<slot_off(0, kAL) + 1024>(voffa, sa[0][1], lds_p);
        __builtin_amdgcn_s_barrier();
        G8P_WAIT_LGKM(0);
        __builtin_amdgcn_sched_barrier(0);
        mma(I1{}, I1{}, fa, fbh);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        // ---- phase 4: (A-hi, B-lo); the counted wait retires every half-tile of the next k-tile
        glds_s<slot_off(0, kBH)>(voffb, sb[1][0], lds_p);
        glds_s<slot_off(0, kBH) + 1024>(voffb, sb[1][1], lds_p);
        stream_advance();
        G8P_WAIT_VM(6);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        mma(I1{}, I0{}, fa, fbl);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
    };

    char* ebuf = smem + kRing + wid * kEpiWave;
    int par = 0;
    for (int v = blockIdx.x; v < total; v += gridDim.x) {
        int64_t m0, n0;
        int z;
        decode(v, m0, n0, z);
        const int64_t kbeg = (int64_t)z * kchunk;
        const int nk = (int)((min(K, kbeg + kchunk) - kbeg) / 64);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int kt = 0; kt < nk; ++kt) {
            ktile(par);
            par ^= 1;
        }

        // ---- epilogue: per wave, 16 rows at a time through the wave's own 4 KiB of LDS (fragment layout -> whole rows), no barrier
        const int64_t mw = m0 + wr * 128, nw = n0 + wc * 64;
        if constexpr (EPI == CM3P_EPI_BF16 || EPI == CM3P_EPI_BF16_ROPE) {
            uint16_t* C = static_cast<uint16_t*>(Cv);
            const bool rotate = (EPI == CM3P_EPI_BF16_ROPE) && nw < rope.ncols;  // a wave's 64 columns are one head
#pragma unroll
            for (int i4 = 0; i4 < 8; ++i4) {
                char* eb = ebuf + (i4 & 1) * 2048;
#pragma unroll
                for (int j4 = 0; j4 < 4; ++j4) {
                    const f32x4 a = acc[i4][j4];
                    const int row = lane & 15, s8 = (j4 * 4 + (lane >> 4)) ^ ((row & 7) << 1);
                    *reinterpret_cast<uint2*>(eb + row * 128 + s8 * 8) = uint2{pack_bf16x2(a.x, a.y), pack_bf16x2(a.z, a.w)};
                }
                if (rotate) {
                    const int row = lane >> 2, dc = lane & 3;
                    const int64_t m = mw + i4 * 16 + row;
                    const uint4 xa = *reinterpret_cast<const uint4*>(eb + row * 128 + ((dc ^ (row & 7)) << 4));
                    const uint4 xb = *reinterpret_cast<const uint4*>(eb + row * 128 + (((dc + 4) ^ (row & 7)) << 4));
                    if (m < M) {
                        const int64_t prow = rope.per_batch ? m : (int64_t)((uint32_t)m % (uint32_t)rope.S);
                        const float* cr = rope.cos + prow * 32 + dc * 8;
                        const float* sr = rope.sin + prow * 32 + dc * 8;
                        const f32x4 c0 = *reinterpret_cast<const f32x4*>(cr), c1 = *reinterpret_cast<const f32x4*>(cr + 4);
                        const f32x4 s0 = *reinterpret_cast<const f32x4*>(sr), s1 = *reinterpret_cast<const f32x4*>(sr + 4);
                        const uint32_t wa[4] = {xa.x, xa.y, xa.z, xa.w}, wb[4] = {xb.x, xb.y, xb.z, xb.w};
                        const float cs[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
                        const float sn[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
                        uint32_t oa[4], ob[4];
                        const float qs = nw < rope.q_cols ? rope.q_scale : 1.f;
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const float a0 = bf16lo(wa[t]), a1 = bf16hi(wa[t]), b0 = bf16lo(wb[t]), b1 = bf16hi(wb[t]);
                            oa[t] = pack_bf16x2(qs * (a0 * cs[2 * t] - b0 * sn[2 * t]), qs * (a1 * cs[2 * t + 1] - b1 * sn[2 * t + 1]));
                            ob[t] = pack_bf16x2(qs * (b0 * cs[2 * t] + a0 * sn[2 * t]), qs * (b1 * cs[2 * t + 1] + a1 * sn[2 * t + 1]));
                        }
                        uint16_t* dst = C + m * ldc + nw + dc * 8;
                        *reinterpret_cast<uint4*>(dst) = uint4{oa[0], oa[1], oa[2], oa[3]};
                        *reinterpret_cast<uint4*>(dst + 32) = uint4{ob[0], ob[1], ob[2], ob[3]};
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int row = u * 8 + (lane >> 3), ch = lane & 7;
                        const uint4 x = *reinterpret_cast<const uint4*>(eb + row * 128 + ((ch ^ (row & 7)) << 4));
                        const int64_t m = mw + i4 * 16 + row, n = nw + ch * 8;
                        if (m < M && n < N) *reinterpret_cast<uint4*>(C + m * ldc + n) = x;
                    }
                }
            }
        } else {
            float* C = static_cast<float*>(Cv) + (int64_t)z * c_split_stride;
            f32x4 rnext[4];
            auto load_r = [&](int i4) {
                if constexpr (EPI == CM3P_EPI_F32_RESID) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int row = u * 4 + (lane >> 4), ch = lane & 15;
                        const int64_t m = mw + i4 * 16 + row, n = nw + ch * 4;
                        rnext[u] = (m < M && n < N) ? *reinterpret_cast<const f32x4*>(R + m * ldc + n) : f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                }
            };
            f32x4 bias = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (EPI == CM3P_EPI_F32_BIAS) {
                const int64_t n = nw + (lane & 15) * 4;
                if (n < N) bias = *reinterpret_cast<const f32x4*>(R + n);
            }
            load_r(0);
#pragma unroll
            for (int i4 = 0; i4 < 8; ++i4) {
#pragma unroll
                for (int j4 = 0; j4 < 4; ++j4) {
                    const int row = lane & 15, ch = j4 * 4 + (lane >> 4);
                    *reinterpret_cast<f32x4*>(ebuf + row * 256 + ((ch ^ row) << 4)) = acc[i4][j4];
                }
                f32x4 rcur[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) rcur[u] = rnext[u];
                if (i4 < 7) load_r(i4 + 1);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int row = u * 4 + (lane >> 4), ch = lane & 15;
                    f32x4 x = *reinterpret_cast<const f32x4*>(ebuf + row * 256 + ((ch ^ row) << 4));
                    const int64_t m = mw + i4 * 16 + row, n = nw + ch * 4;
                    if constexpr (EPI == CM3P_EPI_F32_RESID) x += rcur[u];
                    if constexpr (EPI == CM3P_EPI_F32_BIAS) x += bias;
                    if (m < M && n < N) *reinterpret_cast<f32x4*>(C + m * ldc + n) = x;
                }
            }
        }
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();  // matches group 1's extra barrier
    G8P_WAIT_VM(0);                             // no LDS-DMA may outlive the workgroup's LDS allocation
}

int launch8p(const uint16_t* a, const uint16_t* b, void* C, const float* R, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb,
             int64_t ldc, int epi, int splits, int64_t kchunk, int64_t c_split_stride, hipStream_t s, RopeArgs rope) {
    const int tiles_m = (int)((M + 255) / 256), tiles_n = (int)((N + 255) / 256);
    const int ntiles = tiles_m * tiles_n, total = ntiles * splits;
    static int num_cu = 0;
    if (num_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) num_cu = prop.multiProcessorCount;
        if (num_cu <= 0) num_cu = 256;
    }
    const dim3 grid(total < num_cu ? total : num_cu);
#define CM3P_G8P(E)                                                                                                               \
    {                                                                                                                             \
        static bool attr_set = false;                                                                                             \
        if (!attr_set) {                                                                                                          \
            if (hipFuncSetAttribute((const void*)gemm8p_kernel<E>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds8p) != hipSuccess) \
                return CM3P_ERR_LAUNCH;                                                                                           \
            attr_set = true;                                                                                                      \
        }                                                                                                                         \
        gemm8p_kernel<E><<<grid, 512, kLds8p, s>>>(a, b, C, R, M, N, K, lda, ldb, ldc, tiles_n, ntiles, total, kchunk, c_split_stride, rope); \
    }
    switch (epi) {
        case CM3P_EPI_BF16: CM3P_G8P(CM3P_EPI_BF16) break;
        case CM3P_EPI_F32: CM3P_G8P(CM3P_EPI_F32) break;
        case CM3P_EPI_F32_RESID: CM3P_G8P(CM3P_EPI_F32_RESID) break;
        case CM3P_EPI_BF16_ROPE: CM3P_G8P(CM3P_EPI_BF16_ROPE) break;
        case CM3P_EPI_F32_BIAS: CM3P_G8P(CM3P_EPI_F32_BIAS) break;
        default: return CM3P_ERR_INVALID;
    }
#undef CM3P_G8P
    return CM3P_OK;
}

}  // namespace

// Internal entry used by gemm.hip; returns CM3P_ERR_INVALID for what this kernel does not cover (the caller then falls back).
int cm3p_gemm8p_dispatch(const void* A, const void* B, void* C, const float* R, int64_t M, int64_t N, int64_t K, int64_t lda,
                         int64_t ldb, int64_t ldc, int a_kc, int b_kc, int epi, int splits, int64_t kchunk, int64_t c_split_stride,
                         hipStream_t s, RopeArgs rope) {
    if (!(a_kc && b_kc)) return CM3P_ERR_INVALID;
    if (M % 8 != 0 || N % 8 != 0 || K % 64 != 0 || kchunk % 64 != 0) return CM3P_ERR_INVALID;
    if (lda * 2 * 8 >= (int64_t(1) << 31) || ldb * 2 * 8 >= (int64_t(1) << 31)) return CM3P_ERR_INVALID;
    return launch8p(static_cast<const uint16_t*>(A), static_cast<const uint16_t*>(B), C, R, M, N, K, lda, ldb, ldc, epi, splits, kchunk,
                    c_split_stride, s, rope);
}
