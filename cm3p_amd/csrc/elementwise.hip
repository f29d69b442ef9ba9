// HBM-bound elementwise / small-reduction kernels of the encoder: casts, RoPE, GeGLU, GELU, pooling.
// 16-byte accesses per lane, grid capped at 2048 blocks with grid-stride loops (256 CUs x 8 blocks).
//
// Math restated from:
//   TF:models/modernbert/modeling_modernbert.py:141-163,188-219  rotary tables and apply_rotary_pos_emb (fp32 rotate, half-split)
//   TF:models/modernbert/modeling_modernbert.py:89-91             GeGLU with exact-erf GELU
//   ref:cm3p/modeling_cm3p.py:385-396,631-642                     cls / masked-mean pooling
#include "common.h"

namespace {

inline int ew_grid(int64_t items, int per_block = 256) {
    int64_t blocks = (items + per_block - 1) / per_block;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

// (the GELU itself: gelu_erf2 / gelu_cdf_pdf2 in common.h)
__device__ __forceinline__ void unpack8(const uint4 w, float (&f)[8]) {
    f[0] = bf16lo(w.x); f[1] = bf16hi(w.x); f[2] = bf16lo(w.y); f[3] = bf16hi(w.y);
    f[4] = bf16lo(w.z); f[5] = bf16hi(w.z); f[6] = bf16lo(w.w); f[7] = bf16hi(w.w);
}
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
    return uint4{pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7])};
}

__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ x, uint16_t* __restrict__ y, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
        reinterpret_cast<uint2*>(y)[i] = uint2{pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w)};
    }
}

// bf16 copy of an fp32 [rows, cols] matrix AND its transpose [cols, rows] in one pass: 64 x 64 tiles through LDS (each element is
// rounded once; both outputs hold the same bf16 values).  rows, cols multiples of 8.
__global__ __launch_bounds__(256) void cast_f32_bf16_t_kernel(const float* __restrict__ x, uint16_t* __restrict__ y, uint16_t* __restrict__ yt,
                                                              int rows, int cols, int tiles_c) {
    __shared__ uint16_t tile[64][66];  // 66: a column walk touches a different bank pair per row
    const int r0 = (blockIdx.x / tiles_c) * 64, c0 = (blockIdx.x % tiles_c) * 64;
    const int tid = threadIdx.x;
#pragma unroll
    for (int it = 0; it < 4; ++it) {  // 64 rows x 16 float4
        const int r = it * 16 + (tid >> 4), c = (tid & 15) * 4;
        if (r0 + r < rows && c0 + c < cols) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(x + (int64_t)(r0 + r) * cols + c0 + c);
            const uint32_t lo = pack_bf16x2(v.x, v.y), hi = pack_bf16x2(v.z, v.w);
            *reinterpret_cast<uint2*>(y + (int64_t)(r0 + r) * cols + c0 + c) = uint2{lo, hi};
            tile[r][c] = (uint16_t)lo;
            tile[r][c + 1] = (uint16_t)(lo >> 16);
            tile[r][c + 2] = (uint16_t)hi;
            tile[r][c + 3] = (uint16_t)(hi >> 16);
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 2; ++it) {  // 64 transposed rows x 8 chunks of 8 elements
        const int c = it * 32 + (tid >> 3), r = (tid & 7) * 8;
        if (c0 + c < cols && r0 + r < rows) {
            uint32_t w[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) w[k] = (uint32_t)tile[r + 2 * k][c] | ((uint32_t)tile[r + 2 * k + 1][c] << 16);
            *reinterpret_cast<uint4*>(yt + (int64_t)(c0 + c) * rows + r0 + r) = uint4{w[0], w[1], w[2], w[3]};
        }
    }
}

// The same for MANY matrices in one launch (a tower's Wqkv / Wo / Wi / Wo2 of every layer at the start of a training forward: r03 cast
// each of them in a launch of its own, 8 us apiece for a few us of work).  table[6 i ..]: source, bf16 copy, transposed copy (device
// addresses), rows, cols, first block of matrix i; block b belongs to the last matrix whose first block is <= b.
__global__ __launch_bounds__(256) void cast_f32_bf16_t_multi_kernel(const int64_t* __restrict__ table, int n) {
    __shared__ uint16_t tile[64][66];
    int lo = 0, hi = n - 1;  // (wave-uniform bisection over at most a few hundred entries)
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[6 * mid + 5] <= (int64_t)blockIdx.x) lo = mid;
        else hi = mid - 1;
    }
    const float* x = reinterpret_cast<const float*>(table[6 * lo]);
    uint16_t* y = reinterpret_cast<uint16_t*>(table[6 * lo + 1]);
    uint16_t* yt = reinterpret_cast<uint16_t*>(table[6 * lo + 2]);
    const int rows = (int)table[6 * lo + 3], cols = (int)table[6 * lo + 4];
    const int blk = (int)((int64_t)blockIdx.x - table[6 * lo + 5]);
    const int tiles_c = (cols + 63) / 64;
    const int r0 = (blk / tiles_c) * 64, c0 = (blk % tiles_c) * 64;
    const int tid = threadIdx.x;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int r = it * 16 + (tid >> 4), c = (tid & 15) * 4;
        if (r0 + r < rows && c0 + c < cols) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(x + (int64_t)(r0 + r) * cols + c0 + c);
            const uint32_t lo2 = pack_bf16x2(v.x, v.y), hi2 = pack_bf16x2(v.z, v.w);
            *reinterpret_cast<uint2*>(y + (int64_t)(r0 + r) * cols + c0 + c) = uint2{lo2, hi2};
            tile[r][c] = (uint16_t)lo2;
            tile[r][c + 1] = (uint16_t)(lo2 >> 16);
            tile[r][c + 2] = (uint16_t)hi2;
            tile[r][c + 3] = (uint16_t)(hi2 >> 16);
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int c = it * 32 + (tid >> 3), r = (tid & 7) * 8;
        if (c0 + c < cols && r0 + r < rows) {
            uint32_t w[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) w[k] = (uint32_t)tile[r + 2 * k][c] | ((uint32_t)tile[r + 2 * k + 1][c] << 16);
            *reinterpret_cast<uint4*>(yt + (int64_t)(c0 + c) * rows + r0 + r) = uint4{w[0], w[1], w[2], w[3]};
        }
    }
}

template <bool B_BF16>
__global__ __launch_bounds__(256) void add_f32_kernel(const float* __restrict__ a, const void* __restrict__ b, float* y32,
                                                      uint16_t* __restrict__ y16, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        f32x4 v = reinterpret_cast<const f32x4*>(a)[i];
        if constexpr (B_BF16) {
            const uint2 w = reinterpret_cast<const uint2*>(b)[i];
            v += f32x4{bf16lo(w.x), bf16hi(w.x), bf16lo(w.y), bf16hi(w.y)};
        } else {
            v += reinterpret_cast<const f32x4*>(b)[i];
        }
        if (y32) reinterpret_cast<f32x4*>(y32)[i] = v;
        if (y16) reinterpret_cast<uint2*>(y16)[i] = uint2{pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w)};
    }
}

// ---- RoPE ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rope_table_kernel(const int64_t* __restrict__ pos, const float* __restrict__ inv_freq,
                                                         float* __restrict__ cos_out, float* __restrict__ sin_out, int64_t n,
                                                         int half) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n * half; i += (int64_t)gridDim.x * 256) {
        const int64_t p = i / half;
        const int j = (int)(i % half);
        const float ang = (float)pos[p] * inv_freq[j];  // one fp32 product, as the reference's (d/2,1)@(1,S) matmul
        cos_out[i] = cosf(ang);
        sin_out[i] = sinf(ang);
    }
}

// One thread rotates 8 (j, j+32) pairs of one head: two 16-byte loads, two 16-byte stores.
template <bool INVERSE>
__global__ __launch_bounds__(256) void rope_apply_kernel(uint16_t* __restrict__ qkv, const float* __restrict__ cos_tab,
                                                         const float* __restrict__ sin_tab, int64_t T, int S, int nh,
                                                         int64_t pos_batch_stride) {
    const int64_t per_tok = (int64_t)2 * nh * 4;
    const int64_t total = T * per_tok;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t t = i / per_tok;
        const int r = (int)(i % per_tok);
        const int part = r / (nh * 4);  // 0 = q, 1 = k
        const int h = (r / 4) % nh;
        const int c = r & 3;
        const int64_t b = t / S, s = t % S;
        const int64_t prow = (pos_batch_stride ? b * pos_batch_stride : 0) + s;
        uint16_t* base = qkv + (t * 3 + part) * (int64_t)nh * 64 + h * 64 + c * 8;
        float x1[8], x2[8], cs[8], sn[8];
        unpack8(*reinterpret_cast<const uint4*>(base), x1);
        unpack8(*reinterpret_cast<const uint4*>(base + 32), x2);
        const f32x4* cp = reinterpret_cast<const f32x4*>(cos_tab + prow * 32 + c * 8);
        const f32x4* sp = reinterpret_cast<const f32x4*>(sin_tab + prow * 32 + c * 8);
        const f32x4 c0 = cp[0], c1 = cp[1], s0 = sp[0], s1 = sp[1];
        cs[0] = c0.x; cs[1] = c0.y; cs[2] = c0.z; cs[3] = c0.w; cs[4] = c1.x; cs[5] = c1.y; cs[6] = c1.z; cs[7] = c1.w;
        sn[0] = s0.x; sn[1] = s0.y; sn[2] = s0.z; sn[3] = s0.w; sn[4] = s1.x; sn[5] = s1.y; sn[6] = s1.z; sn[7] = s1.w;
        float y1[8], y2[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if constexpr (!INVERSE) {
                y1[j] = x1[j] * cs[j] - x2[j] * sn[j];
                y2[j] = x2[j] * cs[j] + x1[j] * sn[j];
            } else {
                y1[j] = x1[j] * cs[j] + x2[j] * sn[j];
                y2[j] = x2[j] * cs[j] - x1[j] * sn[j];
            }
        }
        *reinterpret_cast<uint4*>(base) = pack8(y1);
        *reinterpret_cast<uint4*>(base + 32) = pack8(y2);
    }
}

// ---- GeGLU / GELU -------------------------------------------------------------------------------------------
#ifndef CM3P_GEGLU_X2
#define CM3P_GEGLU_X2 3  // extra items per lane and trip - 3: a wave walks 4 KiB of every stream per trip (0: 1 KiB, r01-r04)
#endif
// The lane's items of one trip: a wave's 64 NI consecutive items i = t * c8 + c, lane l taking base + l, base + 64 + l, ...: every load / store
// of a trip is one contiguous KiB per wave and a wave streams NI KiB of every row segment before it moves on (r05: with 1 KiB per stream
// and trip the GeGLU kernels' three to five streams ran at 5.0 TB/s where LayerNorm's 3-KiB rows reach 6.2; 4 KiB: backward 296 -> 275 us).
// (t, c) advances by the stride's quotient and remainder: no 64-bit division per item.
template <int NI>
struct GegluWalk {
    int64_t dq, t0;
    int dr, c0, c8;
    __device__ __forceinline__ GegluWalk(int I) {
        c8 = I / 8;
        const int64_t stride = (int64_t)gridDim.x * 256 * NI, first = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * (64 * NI) + (threadIdx.x & 63);
        dq = stride / c8;
        dr = (int)(stride % c8);
        t0 = first / c8;
        c0 = (int)(first % c8);
    }
    // -> false when the walk is over; else the NI items of this trip (items past the end repeat item 0 and are flagged)
    __device__ __forceinline__ bool next(int64_t T, int64_t (&tt)[NI], int (&cc)[NI], bool (&ok)[NI]) {
        if (t0 >= T) return false;
        if (c0 >= c8) {
            c0 -= c8;
            if (++t0 >= T) return false;
        }
        tt[0] = t0, cc[0] = c0, ok[0] = true;
#pragma unroll
        for (int k = 1; k < NI; ++k) {
            tt[k] = tt[k - 1];
            cc[k] = cc[k - 1] + 64;
            while (cc[k] >= c8) {
                cc[k] -= c8;
                ++tt[k];
            }
            ok[k] = tt[k] < T;
            if (!ok[k]) tt[k] = t0, cc[k] = c0;
        }
        t0 += dq;
        c0 += dr;
        return true;
    }
};

__global__ __launch_bounds__(256) void geglu_fwd_kernel(const uint16_t* __restrict__ h, uint16_t* __restrict__ g, int64_t T, int I) {
    constexpr int NI = CM3P_GEGLU_X2 + 1;
    GegluWalk<NI> walk(I);
    int64_t tt[NI];
    int cc[NI];
    bool ok[NI];
    while (walk.next(T, tt, cc, ok)) {
        uint4 ha[NI], hb[NI];
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            ha[k] = gload16<(CM3P_NT & 128) != 0>(h + tt[k] * 2 * I + cc[k] * 8);
            hb[k] = gload16<(CM3P_NT & 128) != 0>(h + tt[k] * 2 * I + I + cc[k] * 8);
        }
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            float a[8], b[8], y[8];
            unpack8(ha[k], a);
            unpack8(hb[k], b);
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                const f32x2 gl = gelu_erf2(f32x2{a[j], a[j + 1]});
                y[j] = gl.x * b[j];
                y[j + 1] = gl.y * b[j + 1];
            }
            if (ok[k]) gstore16<(CM3P_NT & 16) != 0>(g + tt[k] * I + cc[k] * 8, pack8(y));
        }
    }
}

// one item of the GeGLU backward: 8 columns of one row
__device__ __forceinline__ void geglu_bwd_item(const uint4 ha, const uint4 hb, const uint4 hd, uint4& oa, uint4& ob) {
    float a[8], b[8], d[8], da[8], db[8];
    unpack8(ha, a);
    unpack8(hb, b);
    unpack8(hd, d);
#pragma unroll
    for (int j = 0; j < 8; j += 2) {  // gelu'(a) = Phi(a) + a phi(a), gelu(a) = a Phi(a): one Phi for both
        const f32x2 av = {a[j], a[j + 1]};
        f32x2 cdf, pdf;
        gelu_cdf_pdf2(av, cdf, pdf);
        const f32x2 gr = av * pdf + cdf, gl = av * cdf;
        da[j] = d[j] * b[j] * gr.x;
        da[j + 1] = d[j + 1] * b[j + 1] * gr.y;
        db[j] = d[j] * gl.x;
        db[j + 1] = d[j + 1] * gl.y;
    }
    oa = pack8(da);
    ob = pack8(db);
}

__global__ __launch_bounds__(256) void geglu_bwd_kernel(const uint16_t* __restrict__ dg, const uint16_t* __restrict__ h,
                                                        uint16_t* __restrict__ dh, int64_t T, int I) {
    constexpr int NI = CM3P_GEGLU_X2 + 1;
    GegluWalk<NI> walk(I);
    int64_t tt[NI];
    int cc[NI];
    bool ok[NI];
    while (walk.next(T, tt, cc, ok)) {
        uint4 ha[NI], hb[NI], hd[NI];
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            ha[k] = gload16<(CM3P_NT & 128) != 0>(h + tt[k] * 2 * I + cc[k] * 8);
            hb[k] = gload16<(CM3P_NT & 128) != 0>(h + tt[k] * 2 * I + I + cc[k] * 8);
            hd[k] = gload16<(CM3P_NT & 128) != 0>(dg + tt[k] * I + cc[k] * 8);
        }
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            uint4 oa, ob;
            geglu_bwd_item(ha[k], hb[k], hd[k], oa, ob);
            if (ok[k]) {
                gstore16<(CM3P_NT & 16) != 0>(dh + tt[k] * 2 * I + cc[k] * 8, oa);
                gstore16<(CM3P_NT & 16) != 0>(dh + tt[k] * 2 * I + I + cc[k] * 8, ob);
            }
        }
    }
}

__global__ __launch_bounds__(256) void gelu_fwd_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        float a[8], o[8];
        unpack8(reinterpret_cast<const uint4*>(x)[i], a);
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
            const f32x2 gl = gelu_erf2(f32x2{a[j], a[j + 1]});
            o[j] = gl.x;
            o[j + 1] = gl.y;
        }
        reinterpret_cast<uint4*>(y)[i] = pack8(o);
    }
}

__global__ __launch_bounds__(256) void gelu_bwd_kernel(const uint16_t* __restrict__ dy, const uint16_t* __restrict__ x,
                                                       uint16_t* __restrict__ dx, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        float a[8], d[8], o[8];
        unpack8(reinterpret_cast<const uint4*>(x)[i], a);
        unpack8(reinterpret_cast<const uint4*>(dy)[i], d);
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
            const f32x2 gr = gelu_erf_grad2(f32x2{a[j], a[j + 1]});
            o[j] = d[j] * gr.x;
            o[j + 1] = d[j + 1] * gr.y;
        }
        reinterpret_cast<uint4*>(dx)[i] = pack8(o);
    }
}

// ---- pooling ------------------------------------------------------------------------------------------------
constexpr int kPoolChunk = 128;

// partial[b, c, :] = sum over the 128 rows of chunk c of h[b, s, :] * m[b, s]
__global__ __launch_bounds__(256) void pool_partial_kernel(const float* __restrict__ h, const int64_t* __restrict__ mask,
                                                           float* __restrict__ partial, int S, int H, int nchunks) {
    const int b = blockIdx.x / nchunks, c = blockIdx.x % nchunks;
    const int s0 = c * kPoolChunk, s1 = min(S, s0 + kPoolChunk);
    for (int col = threadIdx.x * 4; col < H; col += 1024) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int s = s0; s < s1; ++s) {
            const float m = mask ? (float)mask[(int64_t)b * S + s] : 1.0f;
            acc += *reinterpret_cast<const f32x4*>(h + ((int64_t)b * S + s) * H + col) * m;
        }
        *reinterpret_cast<f32x4*>(partial + ((int64_t)b * nchunks + c) * H + col) = acc;
    }
}

__global__ __launch_bounds__(256) void pool_final_kernel(const float* __restrict__ partial, const int64_t* __restrict__ mask,
                                                         float* __restrict__ pooled, float* __restrict__ count, int S, int H,
                                                         int nchunks) {
    __shared__ float red[4];
    const int b = blockIdx.x;
    float cnt = 0.f;
    if (mask) {
        for (int s = threadIdx.x; s < S; s += 256) cnt += (float)mask[(int64_t)b * S + s];
        cnt = wave_sum(cnt);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = cnt;
        __syncthreads();
        cnt = (red[0] + red[1]) + (red[2] + red[3]);
    } else {
        cnt = (float)S;
    }
    const float inv = 1.0f / fmaxf(cnt, 1e-9f);
    if (threadIdx.x == 0 && count) count[b] = cnt;
    for (int col = threadIdx.x; col < H; col += 256) {
        float s = 0.f;
        for (int c = 0; c < nchunks; ++c) s += partial[((int64_t)b * nchunks + c) * H + col];
        pooled[(int64_t)b * H + col] = mask ? s * inv : s / (float)S;
    }
}

__global__ __launch_bounds__(256) void pool_cls_kernel(const float* __restrict__ h, float* __restrict__ pooled, int S, int H) {
    const int b = blockIdx.x;
    for (int col = threadIdx.x; col < H; col += 256) pooled[(int64_t)b * H + col] = h[(int64_t)b * S * H + col];
}

__global__ __launch_bounds__(256) void pool_bwd_kernel(const float* __restrict__ dpooled, const int64_t* __restrict__ mask,
                                                       const float* __restrict__ count, float* __restrict__ dh, int Bn, int S,
                                                       int H, int cls) {
    const int h4 = H / 4;
    const int64_t total = (int64_t)Bn * S * h4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / h4;
        const int col = (int)(i % h4) * 4;
        const int64_t b = row / S, s = row % S;
        float scale;
        if (cls) scale = (s == 0) ? 1.0f : 0.0f;
        else if (mask) scale = (float)mask[row] / fmaxf(count[b], 1e-9f);
        else scale = 1.0f / (float)S;
        const f32x4 g = *reinterpret_cast<const f32x4*>(dpooled + b * H + col);
        *reinterpret_cast<f32x4*>(dh + row * H + col) = g * scale;
    }
}

// dst row i = src row idx[i] (GATHER) or dst row idx[i] = src row i (scatter); rows of H fp32 values, H % 4 == 0
template <bool GATHER>
__global__ __launch_bounds__(256) void move_rows_kernel(const float* __restrict__ src, const int64_t* __restrict__ idx,
                                                        float* __restrict__ dst, int64_t n, int h4) {
    const int64_t total = n * h4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / h4;
        const int c = (int)(i - r * h4);
        const int64_t other = idx[r];
        if constexpr (GATHER) reinterpret_cast<f32x4*>(dst)[i] = reinterpret_cast<const f32x4*>(src)[other * h4 + c];
        else reinterpret_cast<f32x4*>(dst)[other * h4 + c] = reinterpret_cast<const f32x4*>(src)[i];
    }
}

}  // namespace

extern "C" {

int cm3p_abi_version(void) { return CM3P_ABI_VERSION; }

int cm3p_cast_f32_bf16(const float* x, void* y, int64_t n, void* stream) {
    CM3P_REQUIRE(x && y && n >= 0 && n % 4 == 0);
    if (n == 0) return CM3P_OK;
    cast_f32_bf16_kernel<<<ew_grid(n / 4), 256, 0, static_cast<hipStream_t>(stream)>>>(x, (uint16_t*)y, n / 4);
    CM3P_LAUNCH_CHECK();
    return CM3P_OK;
}

int cm3p_cast_f32_bf16_t(const float* x, void* y, void* y_t, int64_t rows, int64_t cols, void* stream) {
    CM3P_REQUIRE(x && y && y_t && rows > 0 && cols > 0 && rows % 8 == 0 && cols % 8 == 0 && rows < (1 << 24) && cols < (1 << 24));
    CM3P_REQUIRE(cm3p_aligned16(x) && cm3p_aligned16(y) && cm3p_aligned16(y_t));
    const int tiles_r = (int)((rows + 63) / 64), tiles_c = (int)((cols + 63) / 64);
    cast_f32_bf16_t_kernel<<<tiles_r * tiles_c, 256, 0, static_cast<hipStream_t>(stream)>>>(x, (uint16_t*)y, (uint16_t*)y_t, (int)rows, (int)cols, tiles_c);
    CM3P_LAUNCH_CHECK();
    return CM3P_OK;
}

int cm3p_cast_f32_bf16_t_multi(const int64_t* table, int n, int64_t total_blocks, void* stream) {
    CM3P_REQUIRE(table && n > 0 && total_blocks > 0 && total_blocks < (int64_t(1) << 31));
    cast_f32_bf16_t_multi_kernel<<<(int)total_blocks, 256, 0, static_cast<hipStream_t>(stream)>>>(table, n);
    CM3P_LAUNCH_CHECK();
    return CM3P_OK;
}

int cm3p_add_f32(const float* a, const void* b, int b_dtype, float* y_f32, void* y_bf16, int64_t n, void* stream) {
    CM3P_REQUIRE(a && b && (y_f32 || y_bf16) && n >= 0 && n % 4 == 0);
    if (n == 0) return CM3P_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (b_dtype == CM3P_BF16) add_f32_kernel<true><<<ew_grid(n / 4), 256, 0, s>>>(a, b, y_f32, (uint16_t*)y_bf16, n / 4);
    else add_f32_kernel<false><<<ew_grid(n / 4), 256, 0, s>>>(a, b, y_f32, (uint16_t*)y_bf16, n / 4);
    CM3P_LAUNCH_CHECK();
    return CM3P_OK;
}

int cm3p_gather_rows_f32(const float* src, const int64_t* idx, float* dst, int64_t n, int H, void* stream) {
    CM3P_REQUIRE(src && idx && dst && n >= 0 && H > 0 && H % 4 == 0 && cm3p_aligned16(src) && cm3p_aligned16(dst));
    if (n == 0) return CM3P_OK;
    move_rows_kernel<true><<<ew_grid(n * (H / 4)), 256, 0, static_cast<hipStream_t>(stream)>>>(src, idx, dst, n, H / 4);
    CM3P_LAUNCH_CHECK();
    return CM3P_OK;
}

int cm3p_scatter_rows_f32(const float* src, const int64_t* idx, float* dst, int64_t n, int H, void* stream) {
    CM3P_REQUIRE(src && idx && dst && n >= 0 && H > 0 && H % 4 == 0 && cm3p_aligned16(src) && cm3p_aligned16(dst));
    if (n == 0) return CM3P_OK;
    move_rows_kernel<false><<<ew_grid(n * (H / 4)), 256, 0, static_cast<hipStream_t>(stream)>>>(src, idx, dst, n, H / 4);
    CM3P_LAUNCH_CHECK();
    return CM3P_OK;
}

int cm3p_rope_table(const int64_t* position_ids, int64_t n_pos, const float* inv_freq, int half_dim, float* cos_out,
                    float* sin_out, void* stream) {
    CM3P_REQUIRE(position_ids && inv_freq && cos_out && sin_out && n_pos > 0 && half_dim > 0);
    rope_table_kernel<<<ew_grid(n_pos * half_dim), 256, 0, static_cast<hipStream_t>(stream)>>>(position_ids, inv_freq, cos_out,
                                                                                              sin_out, n_pos, half_dim);
    CM3P_LAUNCH_CHECK();
    return CM3P_OK;
}

int cm3p_rope_apply(void* qkv, const float* cos_tab, const float* sin_tab, int B, int S, int nh, int64_t pos_batch_stride,
                    int inverse, void* stream) {
    CM3P_REQUIRE(qkv && cos_tab && sin_tab && B > 0 && S > 0 && nh > 0);
    CM3P_REQUIRE(pos_batch_stride == 0 || pos_batch_stride == S);
    const int64_t T = (int64_t)B * S;
    const int grid = ew_grid(T * 2 * nh * 4);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (inverse) rope_apply_kernel<true><<<grid, 256, 0, s>>>((uint16_t*)qkv, cos_tab, sin_tab, T, S, nh, pos_batch_stride);
    else rope_apply_kernel<false><<<grid, 256, 0, s>>>((uint16_t*)qkv, cos_tab, sin_tab, T, S, nh, pos_batch_stride);
    CM3P_LAUNCH_CHECK();
    return CM3P_OK;
}

int cm3p_geglu_fwd(const void* h, void* g, int64_t T, int I, void* stream) {
    CM3P_REQUIRE(h && g && T >= 0 && I > 0 && I % 8 == 0);
    if (T == 0) return CM3P_OK;
    geglu_fwd_kernel<<<ew_grid(T * (I / 8)), 256, 0, static_cast<hipStream_t>(stream)>>>((const uint16_t*)h, (uint16_t*)g, T, I);
    CM3P_LAUNCH_CHECK();
    return CM3P_OK;
}

int cm3p_geglu_bwd(const void* dg, const void* h, void* dh, int64_t T, int I, void* stream) {
    CM3P_REQUIRE(dg && h && dh && T >= 0 && I > 0 && I % 8 == 0);
    if (T == 0) return CM3P_OK;
    geglu_bwd_kernel<<<ew_grid(T * (I / 8)), 256, 0, static_cast<hipStream_t>(stream)>>>((const uint16_t*)dg, (const uint16_t*)h,
                                                                                        (uint16_t*)dh, T, I);
    CM3P_LAUNCH_CHECK();
    return CM3P_OK;
}

int cm3p_gelu_fwd(const void* x, void* y, int64_t n, void* stream) {
    CM3P_REQUIRE(x && y && n >= 0 && n % 8 == 0);
    if (n == 0) return CM3P_OK;
    gelu_fwd_kernel<<<ew_grid(n / 8), 256, 0, static_cast<hipStream_t>(stream)>>>((const uint16_t*)x, (uint16_t*)y, n / 8);
    CM3P_LAUNCH_CHECK();
    return CM3P_OK;
}

int cm3p_gelu_bwd(const void* dy, const void* x, void* dx, int64_t n, void* stream) {
    CM3P_REQUIRE(dy && x && dx && n >= 0 && n % 8 == 0);
    if (n == 0) return CM3P_OK;
    gelu_bwd_kernel<<<ew_grid(n / 8), 256, 0, static_cast<hipStream_t>(stream)>>>((const uint16_t*)dy, (const uint16_t*)x,
                                                                                 (uint16_t*)dx, n / 8);
    CM3P_LAUNCH_CHECK();
    return CM3P_OK;
}

int cm3p_pool_chunks(int S) { return (S + kPoolChunk - 1) / kPoolChunk; }

int cm3p_pool_fwd(const float* h, const int64_t* mask, float* pooled, float* partial, float* count, int Bn, int S, int H,
                  int cls, void* stream) {
    CM3P_REQUIRE(h && pooled && Bn > 0 && S > 0 && H > 0 && H % 4 == 0);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (cls) {
        pool_cls_kernel<<<Bn, 256, 0, s>>>(h, pooled, S, H);
    } else {
        CM3P_REQUIRE(partial);
        const int nch = cm3p_pool_chunks(S);
        pool_partial_kernel<<<Bn * nch, 256, 0, s>>>(h, mask, partial, S, H, nch);
        CM3P_LAUNCH_CHECK();
        pool_final_kernel<<<Bn, 256, 0, s>>>(partial, mask, pooled, count, S, H, nch);
    }
    CM3P_LAUNCH_CHECK();
    return CM3P_OK;
}

int cm3p_pool_bwd(const float* dpooled, const int64_t* mask, const float* count, float* dh, int Bn, int S, int H, int cls,
                  void* stream) {
    CM3P_REQUIRE(dpooled && dh && Bn > 0 && S > 0 && H > 0 && H % 4 == 0);
    CM3P_REQUIRE(cls || !mask || count);
    pool_bwd_kernel<<<ew_grid((int64_t)Bn * S * (H / 4)), 256, 0, static_cast<hipStream_t>(stream)>>>(dpooled, mask, count, dh,
                                                                                                      Bn, S, H, cls);
    CM3P_LAUNCH_CHECK();
    return CM3P_OK;
}

}  // extern "C"
