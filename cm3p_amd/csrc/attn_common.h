// Device helpers shared by the attention kernels (attention.hip: forward and the sliding-window backward;
// attention_bwd.hip: the global-layer backward).  See attention.hip's header comment for the MFMA formulation.
#pragma once
#include <limits.h>

#include "common.h"

namespace {

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

// hardware workgroup id -> logical id such that each XCD (workgroup n runs on XCD n % 8) owns a contiguous range of logical
// ids (bijective for any grid size)
__device__ __forceinline__ int xcd_remap(int n, int total) {
    const int q8 = total / 8, r8 = total % 8, xcd = n % 8;
    return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + n / 8;
}
// 1-D grid of nblk * nh * B workgroups -> (block along the sequence, head, batch), XCD-aware: the blocks of one
// (batch, head) are consecutive logical ids, so they run on ONE XCD and share its L2 copy of that head's K / V (or Q / dO)
// instead of pulling it into all eight L2s.  Speed only (measured: forward 2.20 -> 2.10 ms, backward 6.60 -> 6.42 ms per C2
// global layer); any order is correct.
__device__ __forceinline__ void decode_block(int nblk, int nh, int& blk, int& head, int& b) {
    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    blk = logical % nblk;
    const int bh = logical / nblk;
    head = bh % nh;
    b = bh / nh;
}

// The same grid for the SLIDING-WINDOW kernels: head fastest, then the block along the sequence.  A band workgroup reads each row
// of its head once (plus a halo), 128 bytes out of a 4608-byte token row; with the twelve heads of a row block on neighbouring
// workgroups the whole token row is consumed while its DRAM page is open, and a row block's neighbours (the halo) are still on the
// same XCD.  Speed only (r04: see DESIGN section 4); any order is correct.
#ifndef CM3P_BAND_ORDER
#define CM3P_BAND_ORDER 1  // 0: the global kernels' order (A/B builds)
#endif
__device__ __forceinline__ void decode_block_band(int nblk, int nh, int& blk, int& head, int& b) {
    if constexpr (CM3P_BAND_ORDER == 0) {
        decode_block(nblk, nh, blk, head, b);
    } else {
        const int logical = xcd_remap(blockIdx.x, gridDim.x);
        head = logical % nh;
        const int rb = logical / nh;
        blk = rb % nblk;
        b = rb / nblk;
    }
}

// Unpadded ("varlen") batches: sequences are packed back to back, sequence b occupies rows cu[b] .. cu[b+1]-1 of the
// [total, ...] tensors (the layout the reference's flash_attention_2 path builds with _unpad_cm3p_input,
// ref:cm3p/modeling_cm3p.py:65-134) and the per-row statistics are [nh, total].  cu == nullptr: padded [B, S, ...] tensors.
struct VarLen {
    const int* cu;
    int64_t total;
};
struct SeqView {
    int64_t row0;   // first row of this sequence in the token-major tensors
    int64_t stat0;  // index of its row 0 in lse / delta for this head
    int S;          // its length
    bool packed;
    __device__ __forceinline__ SeqView(const VarLen& vl, int b, int head, int Smax, int nh) {
        packed = vl.cu != nullptr;
        if (packed) {
            row0 = vl.cu[b];
            S = vl.cu[b + 1] - vl.cu[b];
            stat0 = (int64_t)head * vl.total + row0;
        } else {
            row0 = (int64_t)b * Smax;
            S = Smax;
            stat0 = ((int64_t)b * nh + head) * Smax;
        }
    }
    // first row of the rotary tables for this sequence: packed tables are per token; padded ones per batch row or shared
    __device__ __forceinline__ int64_t pos0(int b, int64_t pos_batch_stride) const { return packed ? row0 : (int64_t)b * pos_batch_stride; }
};

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


// ---- hand-scheduled kernels (attention_bwd.hip, attention_fwd.hip) --------------------------------------------------------------
// With the 512-entry register budget of one wave per SIMD hipcc selects the AGPR form of every MFMA it generates, and the VALU
// cannot read AGPRs.  Products whose results the VALU consumes are therefore issued as inline-asm MFMAs with VGPR destinations;
// their stationary B operands live in AGPRs ("a").  hipcc pads no hazards for an asm statement: a result is first read one
// pipeline step (hundreds of cycles) after it was issued, behind a sched_barrier, and an AGPR operand must never be
// (re)materialised right in front of the MFMA (tests/test_kernel_isa.py checks the generated loops).
#define CM3P_SB() __builtin_amdgcn_sched_barrier(0)

// D (VGPRs) = A (VGPRs) * B (AGPRs) + C (VGPRs); D never overlaps an input
__device__ __forceinline__ void mfma_vc(f32x16& d, const bf16x8& a, const bf16x8& b, const f32x16& c) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "a"(b), "v"(c));
}
// D += A * B (same registers)
__device__ __forceinline__ void mfma_va(f32x16& d, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "a"(b));
}
__device__ __forceinline__ bf16x8 ld_frag(const char* p) { return *reinterpret_cast<const bf16x8*>(p); }
__device__ __forceinline__ bf16x8 ld_fragT(const char* lo, const char* hi) { return cat_bf16x4(lds_read_tr16(lo), lds_read_tr16(hi)); }
// D (VGPRs) = A (VGPRs) * B (AGPRs), from zero
__device__ __forceinline__ void mfma_v0(f32x16& d, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(d) : "v"(a), "a"(b));
}


// A 32 x 64 block held as two accumulator blocks (lane = row l31, half hh; register i of block blk = column
// 32 blk + 8 (i >> 2) + 4 hh + (i & 3)) -> bf16 rows of a row-major matrix, through a wave-private LDS buffer of 32 x 144 bytes.
// Stored straight from the accumulators a lane owns 4 consecutive columns of one row: eight 8-byte stores per lane, each
// instruction touching 32 different lines.  From here: four 16-byte stores per lane, 8 whole 128-byte rows per instruction.
// (The uint2 stores and uint4 loads do not alias by type, hence the compiler barrier; one wave's LDS operations execute in order.)
__device__ __forceinline__ void store_rows32(char* buf, const f32x16& b0, const f32x16& b1, float mul, uint16_t* dst_row0, int64_t ld_elems,
                                             int nrows_valid, int lane) {
    const int l31 = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        *reinterpret_cast<uint2*>(buf + l31 * 144 + (8 * g + 4 * hh) * 2) =
            uint2{pack_bf16x2(b0[4 * g] * mul, b0[4 * g + 1] * mul), pack_bf16x2(b0[4 * g + 2] * mul, b0[4 * g + 3] * mul)};
        *reinterpret_cast<uint2*>(buf + l31 * 144 + (32 + 8 * g + 4 * hh) * 2) =
            uint2{pack_bf16x2(b1[4 * g] * mul, b1[4 * g + 1] * mul), pack_bf16x2(b1[4 * g + 2] * mul, b1[4 * g + 3] * mul)};
    }
    asm volatile("" ::: "memory");
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (lane >> 3) + 8 * i;
        const uint4 v = *reinterpret_cast<const uint4*>(buf + row * 144 + (lane & 7) * 16);
        if (row < nrows_valid) gstore16<(CM3P_NT & 4) != 0>(dst_row0 + row * ld_elems + (lane & 7) * 8, v);
    }
    asm volatile("" ::: "memory");
}

// ---- LDS-DMA staging of 64-row tiles by a four-wave workgroup (the sliding-window kernels of attention.hip) ---------------------
// A [*, 64] bf16 matrix tile (64 rows x 128 bytes) goes to LDS as the swizzled image off_R / off_T read: wave w brings rows
// 16 w .. 16 w + 15 as two 1-KiB pieces (global_load_lds_dwordx4: lane l's 16 bytes land at piece + 16 l = row l >> 3, chunk slot
// l & 7, which must hold global chunk (l & 7) ^ swz(row): the swizzle is applied to the SOURCE address).  One m0 write serves
// both pieces: the instruction offset moves the LDS address AND the global address, so the scalar base is kept 1 KiB low and
// the first piece's offset 1 KiB high.  Issued as inline assembly on purpose (gemm256.hip: through the builtin the compiler drains
// every outstanding DMA before the next LDS read); ordering is the caller's counted s_waitcnt vmcnt + workgroup barrier.
struct TileDma {
    int prow0, prow1, pc0, pc1;  // the lane's two rows inside the tile and its source chunk byte offsets
    __device__ __forceinline__ TileDma(int wid, int lane) {
        prow0 = 16 * wid + (lane >> 3);
        prow1 = prow0 + 8;
        pc0 = ((lane & 7) ^ swz(prow0)) << 4;
        pc1 = ((lane & 7) ^ swz(prow1)) << 4;
    }
    // rows r0 + prow (clamped to limit - 1) of the matrix at `base` (row stride ldb BYTES) -> LDS address m0v (+ 2 KiB per wave
    // already included by the caller)
    // (audit_id: which operand this is to the bounds audit, common.h)
    __device__ __forceinline__ void rows(uint32_t m0v, const void* base, int ldb, int r0, int limit, int audit_id = CM3P_AUD_T0) const {
        const uint32_t v0 = (uint32_t)(min(r0 + prow0, limit - 1) * ldb + pc0 + 1024);
        const uint32_t v1 = (uint32_t)(min(r0 + prow1, limit - 1) * ldb + pc1);
        const char* b = static_cast<const char*>(base) - 1024;
        CM3P_AUDIT(audit_id, b + v0, 16);          // first piece: base + row * pitch + chunk
        CM3P_AUDIT(audit_id, b + v1 + 1024, 16);   // second piece: the instruction offset moves the global address too
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\tglobal_load_lds_dwordx4 %2, %3 offset:1024" ::"s"(m0v), "v"(v0),
                     "v"(v1), "s"(b)
                     : "memory", "m0");
    }
};
// 64 consecutive floats / bytes (one per lane, index clamped by the caller) -> 64 dwords at LDS address m0v (bytes zero-extended)
__device__ __forceinline__ void dma_dword64(uint32_t m0v, const void* base, uint32_t byte_off, int audit_id = CM3P_AUD_S0) {
    CM3P_AUDIT(audit_id, static_cast<const char*>(base) + byte_off, 4);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2" ::"s"(m0v), "v"(byte_off), "s"(base) : "memory", "m0");
}
__device__ __forceinline__ void dma_ubyte64(uint32_t m0v, const void* base, uint32_t byte_off, int audit_id = CM3P_AUD_S0) {
    CM3P_AUDIT(audit_id, static_cast<const char*>(base) + byte_off, 1);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_ubyte %1, %2" ::"s"(m0v), "v"(byte_off), "s"(base) : "memory", "m0");
}
// wait until at most n of this wave's vector-memory operations are outstanding (n in {0, 4, 5, 6, 8, 10, 12}: the immediates the
// ring kernels need)
__device__ __forceinline__ void dma_wait(int n) {
    if (n == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (n == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (n == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if (n == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (n == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (n == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
}
// workgroup barrier that orders LDS traffic only
__device__ __forceinline__ void lds_only_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void dma_wait_barrier(int n) {
    dma_wait(n);
    lds_only_barrier();
}

}  // namespace
