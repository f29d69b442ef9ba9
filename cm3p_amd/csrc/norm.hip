// Bias-free LayerNorm (forward / backward), the fused token-embedding gather + LayerNorm, and the small
// reductions that go with them.  All of it is HBM-bound: one wave owns a row, 16-byte accesses, fp32 statistics.
//
// Math restated from the reference's encoder (third-party, not vendored):
//   TF:models/modernbert/modeling_modernbert.py:61,64-71   embeddings: LayerNorm(tok_embeddings(ids)), no bias
//   TF:models/modernbert/modeling_modernbert.py:309-314,420 attn_norm / mlp_norm / final_norm = nn.LayerNorm(H, eps, bias=False)
//   ref:cm3p/modeling_cm3p.py:592,603-605                   embedding lookup and the audio-embedding scatter
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int kMaxChunks = 8;  // 8 * 64 lanes * 4 floats = H <= 2048

// A row of H values held by one wave: NC = ceil(H / 256) chunks of 64 lanes x 4 floats.  NC is a template parameter so
// that H = 768 costs 3 chunks of registers and instructions, not the worst-case 8 (occupancy is what hides HBM latency).
template <int NC>
struct RowRegs {
    f32x4 v[NC];
};

// load a row of H values (fp32 or bf16 source) into registers, lane owns columns c*256 + lane*4 .. +3
template <bool SRC_BF16, int NC>
__device__ __forceinline__ void load_row(RowRegs<NC>& r, const void* src, int H, int lane) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int col = c * 256 + lane * 4;
        if (col < H) {
            // (CM3P_NT bit 128: read-once streams as non-temporal loads - an r04 probe)
            if constexpr (SRC_BF16) {
                const uint2 w = gload8<(CM3P_NT & 128) != 0>(static_cast<const uint16_t*>(src) + col);
                r.v[c] = f32x4{bf16lo(w.x), bf16hi(w.x), bf16lo(w.y), bf16hi(w.y)};
            } else {
                r.v[c] = gload16f<(CM3P_NT & 128) != 0>(static_cast<const float*>(src) + col);
            }
        } else {
            r.v[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
}

template <int NC>
__device__ __forceinline__ void row_stats(const RowRegs<NC>& r, int H, int lane, float eps, float& mean, float& rstd) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) s += (r.v[c].x + r.v[c].y) + (r.v[c].z + r.v[c].w);
    mean = wave_sum(s) / (float)H;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int col = c * 256 + lane * 4;
        if (col < H) {
            const f32x4 d = r.v[c] - mean;
            q += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
        }
    }
    const float var = wave_sum(q) / (float)H;
    rstd = 1.0f / sqrtf(var + eps);
}

template <int NC>
__device__ __forceinline__ void store_norm(const RowRegs<NC>& r, const float* __restrict__ w, float mean, float rstd,
                                           float* y32, uint16_t* y16, int H, int lane) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int col = c * 256 + lane * 4;
        if (col < H) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(w + col);
            const f32x4 y = (r.v[c] - mean) * rstd * g;
            if (y32) gstore16f<(CM3P_NT & 16) != 0>(y32 + col, y);
            if (y16) gstore8<(CM3P_NT & 16) != 0>(y16 + col, uint2{pack_bf16x2(y.x, y.y), pack_bf16x2(y.z, y.w)});
        }
    }
}

template <bool X_BF16, int NC>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const void* __restrict__ x, const float* __restrict__ w,
                                                            float* __restrict__ y32, uint16_t* __restrict__ y16,
                                                            float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                            int64_t rows, int H, float eps, int reverse) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    for (int64_t it = wave; it < rows; it += nwaves) {
        // reverse: the last rows first - what the kernel before this one wrote last is what the Infinity Cache still holds
        const int64_t row = reverse ? rows - 1 - it : it;
        RowRegs<NC> r;
        const void* src = X_BF16 ? (const void*)(static_cast<const uint16_t*>(x) + row * H)
                                 : (const void*)(static_cast<const float*>(x) + row * H);
        load_row<X_BF16, NC>(r, src, H, lane);
        float mean, rstd;
        row_stats(r, H, lane, eps, mean, rstd);
        store_norm(r, w, mean, rstd, y32 ? y32 + row * H : nullptr, y16 ? y16 + row * H : nullptr, H, lane);
        if (lane == 0) {
            if (mean_out) mean_out[row] = mean;
            if (rstd_out) rstd_out[row] = rstd;
        }
    }
}

// Backward.  xhat = (x - mean) * rstd, g = dy * w:
//   dx = rstd * (g - mean_H(g) - xhat * mean_H(g * xhat)) [+ dres]      dw[col] = sum_rows dy * xhat
// Each wave walks rows grid-stride and keeps its dw partial in registers; one partial row per block.
template <bool DY_BF16, int NC>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const void* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ w, const float* __restrict__ mean_in,
                                                            const float* __restrict__ rstd_in, const float* dres,
                                                            float* dx32, uint16_t* __restrict__ dx16,
                                                            float* __restrict__ dw_partial, int64_t rows, int H) {
    extern __shared__ __attribute__((aligned(16))) float dw_lds[];  // [4][H]
    const int lane = threadIdx.x & 63;
    const int wid = threadIdx.x >> 6;
    const int64_t wave = (int64_t)blockIdx.x * 4 + wid;
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    RowRegs<NC> dw;
#pragma unroll
    for (int c = 0; c < NC; ++c) dw.v[c] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int64_t row = wave; row < rows; row += nwaves) {
        RowRegs<NC> xr, gr;
        load_row<false, NC>(xr, x + row * H, H, lane);
        const void* dsrc = DY_BF16 ? (const void*)(static_cast<const uint16_t*>(dy) + row * H)
                                   : (const void*)(static_cast<const float*>(dy) + row * H);
        load_row<DY_BF16, NC>(gr, dsrc, H, lane);
        // (requested with the row, not behind the two wave reductions: one exposed round trip per row less, 0.314 -> 0.305 ms at C2)
        RowRegs<NC> rr;
        if (dres) load_row<false, NC>(rr, dres + row * H, H, lane);
        const float mean = mean_in[row], rstd = rstd_in[row];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int col = c * 256 + lane * 4;
            if (col < H) {
                const f32x4 xhat = (xr.v[c] - mean) * rstd;
                const f32x4 wv = *reinterpret_cast<const f32x4*>(w + col);
                dw.v[c] += gr.v[c] * xhat;
                const f32x4 g = gr.v[c] * wv;
                xr.v[c] = xhat;
                gr.v[c] = g;
                s1 += (g.x + g.y) + (g.z + g.w);
                const f32x4 gx = g * xhat;
                s2 += (gx.x + gx.y) + (gx.z + gx.w);
            }
        }
        s1 = wave_sum(s1) / (float)H;
        s2 = wave_sum(s2) / (float)H;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int col = c * 256 + lane * 4;
            if (col < H) {
                f32x4 d = (gr.v[c] - s1 - xr.v[c] * s2) * rstd;
                if (dres) d += rr.v[c];
                if (dx32) gstore16f<(CM3P_NT & 16) != 0>(dx32 + row * H + col, d);
                if (dx16) gstore8<(CM3P_NT & 16) != 0>(dx16 + row * H + col, uint2{pack_bf16x2(d.x, d.y), pack_bf16x2(d.z, d.w)});
            }
        }
    }
    // block-level reduction of the four waves' dw partials, fixed order -> deterministic
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int col = c * 256 + lane * 4;
        if (col < H) *reinterpret_cast<f32x4*>(dw_lds + wid * H + col) = dw.v[c];
    }
    __syncthreads();
    for (int col = threadIdx.x; col < H; col += 256) {
        dw_partial[(int64_t)blockIdx.x * H + col] = (dw_lds[col] + dw_lds[H + col]) + (dw_lds[2 * H + col] + dw_lds[3 * H + col]);
    }
}

// out[col] = sum_b partial[b][col], fixed summation order.  Block = 32 columns x 32 row slices (1024 threads: the 24 workgroups
// this launch has at H = 768 sit between two kernels of the backward chain, so its time is the depth of its dependent load rounds -
// 4 rounds of 8 loads per thread at 1024 partial rows, 16 in the 8-slice form of r01-r03: 17.6 -> ~6 us per launch);
// slices are combined through LDS in slice order (deterministic).
constexpr int kColsumSlices = 32;
__global__ __launch_bounds__(1024) void colsum_kernel(const float* __restrict__ partial, float* __restrict__ out, int nblk, int H) {
    __shared__ float red[kColsumSlices][33];
    const int cx = threadIdx.x & 31, slice = threadIdx.x >> 5;
    const int col = blockIdx.x * 32 + cx;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (col < H) {
        const int per = (nblk + kColsumSlices - 1) / kColsumSlices;
        const int b0 = slice * per, b1 = min(nblk, b0 + per);
        int b = b0;
        for (; b + 8 <= b1; b += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u] += partial[(int64_t)(b + u) * H + col];
        }
        for (; b < b1; ++b) acc[0] += partial[(int64_t)b * H + col];
    }
    red[slice][cx] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    __syncthreads();
    if (slice == 0 && col < H) {
        float s4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < kColsumSlices; i += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) s4[u] += red[i + u][cx];
        }
        out[col] = (s4[0] + s4[1]) + (s4[2] + s4[3]);
    }
}

// ---- embedding gather + LayerNorm ---------------------------------------------------------------------------
// Source row of token t: override_rows[slot[t]] when slot && slot[t] >= 0 (audio placeholder,
// ref:cm3p/modeling_cm3p.py:603-605), else table[ids[t]].
template <bool TAB_BF16, bool OVR_BF16, int NC>
__device__ __forceinline__ void load_embed_row(RowRegs<NC>& r, const int64_t* ids, const void* table, const int32_t* slot,
                                               const void* ovr, int64_t t, int H, int lane, int64_t vocab) {
    const int s = slot ? slot[t] : -1;
    if (s >= 0) {
        const void* src = OVR_BF16 ? (const void*)(static_cast<const uint16_t*>(ovr) + (int64_t)s * H)
                                   : (const void*)(static_cast<const float*>(ovr) + (int64_t)s * H);
        load_row<OVR_BF16, NC>(r, src, H, lane);
    } else {
        const int64_t id = ids[t];
        if ((uint64_t)id >= (uint64_t)vocab) {  // nn.Embedding raises a device-side assert here; a kernel must not fault: zero row
#pragma unroll
            for (int c = 0; c < NC; ++c) r.v[c] = f32x4{0.f, 0.f, 0.f, 0.f};
            return;
        }
        const void* src = TAB_BF16 ? (const void*)(static_cast<const uint16_t*>(table) + id * H)
                                   : (const void*)(static_cast<const float*>(table) + id * H);
        load_row<TAB_BF16, NC>(r, src, H, lane);
    }
}

template <bool TAB_BF16, bool OVR_BF16, int NC>
__global__ __launch_bounds__(256) void embed_ln_fwd_kernel(const int64_t* __restrict__ ids, const void* __restrict__ table,
                                                           const int32_t* __restrict__ slot, const void* __restrict__ ovr,
                                                           const float* __restrict__ w, float* __restrict__ y32,
                                                           uint16_t* __restrict__ y16, float* __restrict__ mean_out,
                                                           float* __restrict__ rstd_out, int64_t T, int H, float eps, int64_t vocab) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    for (int64_t t = wave; t < T; t += nwaves) {
        RowRegs<NC> r;
        load_embed_row<TAB_BF16, OVR_BF16, NC>(r, ids, table, slot, ovr, t, H, lane, vocab);
        float mean, rstd;
        row_stats(r, H, lane, eps, mean, rstd);
        store_norm(r, w, mean, rstd, y32 ? y32 + t * H : nullptr, y16 ? y16 + t * H : nullptr, H, lane);
        if (lane == 0) {
            mean_out[t] = mean;
            rstd_out[t] = rstd;
        }
    }
}

// Backward of the gather + LayerNorm: LayerNorm backward on the re-gathered row, then the row gradient is
// scattered: atomically added into d_table[ids[t]] (several tokens share a row) or stored to d_override[slot[t]]
// (each audio row is used exactly once).
template <bool TAB_BF16, bool OVR_BF16, int NC>
__global__ __launch_bounds__(256) void embed_ln_bwd_kernel(const float* __restrict__ dy, const int64_t* __restrict__ ids,
                                                           const void* __restrict__ table, const int32_t* __restrict__ slot,
                                                           const void* __restrict__ ovr, const float* __restrict__ w,
                                                           const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                           float* __restrict__ d_table, float* __restrict__ d_ovr,
                                                           float* __restrict__ dw_partial, int64_t T, int H,
                                                           int64_t padding_idx, int64_t vocab) {
    extern __shared__ __attribute__((aligned(16))) float dw_lds[];
    const int lane = threadIdx.x & 63;
    const int wid = threadIdx.x >> 6;
    const int64_t wave = (int64_t)blockIdx.x * 4 + wid;
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    RowRegs<NC> dw;
#pragma unroll
    for (int c = 0; c < NC; ++c) dw.v[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int64_t t = wave; t < T; t += nwaves) {
        RowRegs<NC> xr, gr;
        load_embed_row<TAB_BF16, OVR_BF16, NC>(xr, ids, table, slot, ovr, t, H, lane, vocab);
        load_row<false, NC>(gr, dy + t * H, H, lane);
        const float mean = mean_in[t], rstd = rstd_in[t];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int col = c * 256 + lane * 4;
            if (col < H) {
                const f32x4 xhat = (xr.v[c] - mean) * rstd;
                const f32x4 wv = *reinterpret_cast<const f32x4*>(w + col);
                dw.v[c] += gr.v[c] * xhat;
                const f32x4 g = gr.v[c] * wv;
                xr.v[c] = xhat;
                gr.v[c] = g;
                s1 += (g.x + g.y) + (g.z + g.w);
                const f32x4 gx = g * xhat;
                s2 += (gx.x + gx.y) + (gx.z + gx.w);
            }
        }
        s1 = wave_sum(s1) / (float)H;
        s2 = wave_sum(s2) / (float)H;
        const int s = slot ? slot[t] : -1;
        // nn.Embedding(padding_idx=pad_token_id) gives the padding row no gradient (TF:...modeling_modernbert.py:60)
        const int64_t id = ids[t];
        float* dst = s >= 0 ? (d_ovr ? d_ovr + (int64_t)s * H : nullptr)
                            : ((d_table && id != padding_idx && (uint64_t)id < (uint64_t)vocab) ? d_table + id * H : nullptr);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int col = c * 256 + lane * 4;
            if (col < H && dst) {
                const f32x4 d = (gr.v[c] - s1 - xr.v[c] * s2) * rstd;
                if (s >= 0) {
                    *reinterpret_cast<f32x4*>(dst + col) = d;
                } else {
                    atomicAdd(dst + col + 0, d.x);
                    atomicAdd(dst + col + 1, d.y);
                    atomicAdd(dst + col + 2, d.z);
                    atomicAdd(dst + col + 3, d.w);
                }
            }
        }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int col = c * 256 + lane * 4;
        if (col < H) *reinterpret_cast<f32x4*>(dw_lds + wid * H + col) = dw.v[c];
    }
    __syncthreads();
    for (int col = threadIdx.x; col < H; col += 256) {
        dw_partial[(int64_t)blockIdx.x * H + col] = (dw_lds[col] + dw_lds[H + col]) + (dw_lds[2 * H + col] + dw_lds[3 * H + col]);
    }
}

// ---- the same backward without atomics: tokens visited in id order ------------------------------------------------------------------
// `order` lists the tokens sorted by id (stable: ascending token index inside an id); `run_of[p]` numbers the runs of the sorted
// sequence, a new run starting at every change of id AND at every multiple of kEmbChunk (so that a run never leaves its chunk).
// One wave per chunk: it re-gathers each token's row, takes the LayerNorm backward and adds the row gradient to a register
// accumulator, which is written to run_rows[run] whenever the run ends.  embed_run_sum_kernel then gives every vocabulary row the sum
// of its runs in run order.  Every sum has a fixed order: the embedding gradient is reproducible bit for bit (the atomic kernel's
// is not), and tokens that share an id - most of a beatmap - no longer serialise on one row of d_table.
constexpr int kEmbChunk = 64;

template <bool TAB_BF16, bool OVR_BF16, int NC>
__global__ __launch_bounds__(256) void embed_ln_bwd_sorted_kernel(const float* __restrict__ dy, const int64_t* __restrict__ ids,
                                                                  const int64_t* __restrict__ order, const int32_t* __restrict__ run_of,
                                                                  const void* __restrict__ table, const int32_t* __restrict__ slot,
                                                                  const void* __restrict__ ovr, const float* __restrict__ w,
                                                                  const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                                  float* __restrict__ d_ovr, float* __restrict__ run_rows,
                                                                  int64_t* __restrict__ run_ids, float* __restrict__ dw_partial,
                                                                  int64_t T, int H, int64_t padding_idx, int64_t vocab) {
    extern __shared__ __attribute__((aligned(16))) float dw_lds[];
    const int lane = threadIdx.x & 63;
    const int wid = threadIdx.x >> 6;
    const int64_t chunk = (int64_t)blockIdx.x * 4 + wid;
    RowRegs<NC> dw, acc;
#pragma unroll
    for (int c = 0; c < NC; ++c) dw.v[c] = acc.v[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int64_t p0 = chunk * kEmbChunk, p1 = min(T, p0 + kEmbChunk);
    int cur_run = -1;
    int64_t cur_id = -1;
    auto flush = [&]() {
        if (cur_run < 0) return;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int col = c * 256 + lane * 4;
            if (col < H) *reinterpret_cast<f32x4*>(run_rows + (int64_t)cur_run * H + col) = acc.v[c];
            acc.v[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (lane == 0) run_ids[cur_run] = cur_id;
    };
    for (int64_t p = p0; p < p1; ++p) {
        const int64_t t = order[p];
        const int run = run_of[p];
        if (run != cur_run) {
            flush();
            cur_run = run;
            cur_id = ids[t];
        }
        RowRegs<NC> xr, gr;
        load_embed_row<TAB_BF16, OVR_BF16, NC>(xr, ids, table, slot, ovr, t, H, lane, vocab);
        load_row<false, NC>(gr, dy + t * H, H, lane);
        const float mean = mean_in[t], rstd = rstd_in[t];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int col = c * 256 + lane * 4;
            if (col < H) {
                const f32x4 xhat = (xr.v[c] - mean) * rstd;
                const f32x4 wv = *reinterpret_cast<const f32x4*>(w + col);
                dw.v[c] += gr.v[c] * xhat;
                const f32x4 g = gr.v[c] * wv;
                xr.v[c] = xhat;
                gr.v[c] = g;
                s1 += (g.x + g.y) + (g.z + g.w);
                const f32x4 gx = g * xhat;
                s2 += (gx.x + gx.y) + (gx.z + gx.w);
            }
        }
        s1 = wave_sum(s1) / (float)H;
        s2 = wave_sum(s2) / (float)H;
        const int sl = slot ? slot[t] : -1;
        const int64_t id = ids[t];
        // audio placeholders: the row gradient belongs to the audio embedding, not to the table; padding row / ids outside the table: none
        const bool to_table = sl < 0 && id != padding_idx && (uint64_t)id < (uint64_t)vocab;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int col = c * 256 + lane * 4;
            if (col < H) {
                const f32x4 d = (gr.v[c] - s1 - xr.v[c] * s2) * rstd;
                if (sl >= 0) {
                    if (d_ovr) *reinterpret_cast<f32x4*>(d_ovr + (int64_t)sl * H + col) = d;
                } else if (to_table) {
                    acc.v[c] += d;
                }
            }
        }
    }
    flush();
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int col = c * 256 + lane * 4;
        if (col < H) *reinterpret_cast<f32x4*>(dw_lds + wid * H + col) = dw.v[c];
    }
    __syncthreads();
    for (int col = threadIdx.x; col < H; col += 256) {
        dw_partial[(int64_t)blockIdx.x * H + col] = (dw_lds[col] + dw_lds[H + col]) + (dw_lds[2 * H + col] + dw_lds[3 * H + col]);
    }
}

// d_table[v] = sum of the runs whose id is v, in run order (zero when there is none): one wave per vocabulary row.  The run ids
// ascend (the tokens were sorted), so the first run of v is found by bisection over run_ids[0 .. n_runs).
template <int NC>
__global__ __launch_bounds__(256) void embed_run_sum_kernel(const float* __restrict__ run_rows, const int64_t* __restrict__ run_ids,
                                                            const int32_t* __restrict__ run_of, float* __restrict__ d_table, int64_t T,
                                                            int H, int64_t vocab) {
    const int lane = threadIdx.x & 63;
    const int64_t v = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (v >= vocab) return;
    const int n_runs = run_of[T - 1] + 1;
    int lo = 0, hi = n_runs;  // first run with id >= v
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (run_ids[mid] < v) lo = mid + 1;
        else hi = mid;
    }
    RowRegs<NC> acc;
#pragma unroll
    for (int c = 0; c < NC; ++c) acc.v[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int r = lo; r < n_runs && run_ids[r] == v; ++r) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int col = c * 256 + lane * 4;
            if (col < H) acc.v[c] += *reinterpret_cast<const f32x4*>(run_rows + (int64_t)r * H + col);
        }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int col = c * 256 + lane * 4;
        if (col < H) *reinterpret_cast<f32x4*>(d_table + v * H + col) = acc.v[c];
    }
}

// slot[t] = (ids[t] == audio_id) ? number of audio placeholders before t in row-major (b, s) order : -1.
// Single-block exclusive scan (T is at most a few hundred thousand); count[0] = total placeholders.
__global__ __launch_bounds__(1024) void audio_slots_kernel(const int64_t* __restrict__ ids, int64_t T, int64_t audio_id,
                                                           int32_t* __restrict__ slot, int32_t* __restrict__ count) {
    __shared__ int wave_tot[16];
    __shared__ int carry;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < T; base += 1024) {
        const int64_t t = base + threadIdx.x;
        const int flag = (t < T && ids[t] == audio_id) ? 1 : 0;
        const unsigned long long ball = __ballot(flag);
        const int before = __popcll(ball & ((1ull << lane) - 1ull));
        if (lane == 0) wave_tot[wid] = __popcll(ball);
        __syncthreads();
        int off = carry;
        for (int i = 0; i < wid; ++i) off += wave_tot[i];
        if (t < T) slot[t] = flag ? off + before : -1;
        __syncthreads();
        if (threadIdx.x == 0) {
            int tot = 0;
            for (int i = 0; i < 16; ++i) tot += wave_tot[i];
            carry += tot;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) count[0] = carry;
}

// ---- token order for the embedding backward: a stable counting sort of the ids -----------------------------------------------
// What cm3p_embed_ln_bwd_sorted needs from the ids alone: the tokens in ascending-id order, ties in token order (`order`), and the
// run number of every sorted position (`run_of`: a new run where the id changes and at every multiple of kEmbChunk).  Keys are
// clamp(id, -1, vocab) + 1 in [0, vocab + 2): ids outside the table share the two end buckets.  Token ids are small integers
// (beatmap vocabulary 3167), so this is a counting sort in four launches, nothing read by the host:
//   1. per block of kOrdBlock consecutive tokens, a histogram of its keys -> cnt[key][block];
//   2. per key the exclusive prefix of cnt over the blocks and the key's total, then an exclusive scan of the totals over the keys:
//      key start + prefix = the first sorted position of each (key, block) pair;
//   3. each block hands its tokens their positions in token order (one wave walks the block 64 tokens at a time; within a wave
//      the lanes that share a key find each other with one ballot per key bit), writing order[] and the sorted keys;
//   4. run starts from the sorted keys and their running count (two launches over blocks of 1024 positions).
constexpr int kOrdBlock = 1024;     // tokens per block of the histogram / scatter kernels
constexpr int kOrdMaxKeys = 12288;  // LDS: one int per key

__global__ __launch_bounds__(256) void token_hist_kernel(const int64_t* __restrict__ ids, int64_t T, int64_t vocab, int nblk,
                                                         int32_t* __restrict__ cnt) {
    extern __shared__ int hist[];
    const int nkeys = (int)vocab + 2;
    for (int k = threadIdx.x; k < nkeys; k += 256) hist[k] = 0;
    __syncthreads();
    const int64_t t0 = (int64_t)blockIdx.x * kOrdBlock;
    for (int i = threadIdx.x; i < kOrdBlock; i += 256) {
        const int64_t t = t0 + i;
        if (t < T) {
            const int64_t id = ids[t];
            const int key = (int)(id < 0 ? -1 : (id > vocab ? vocab : id)) + 1;
            atomicAdd(&hist[key], 1);
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < nkeys; k += 256) cnt[(int64_t)k * nblk + blockIdx.x] = hist[k];
}

// one wave per key: cnt[key][0 .. nblk) -> its exclusive prefix over the blocks (in place), key_total[key] = the key's token count
__global__ __launch_bounds__(256) void token_key_prefix_kernel(int32_t* __restrict__ cnt, int32_t* __restrict__ key_total, int nkeys, int nblk) {
    const int lane = threadIdx.x & 63;
    const int key = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (key >= nkeys) return;
    int32_t* row = cnt + (int64_t)key * nblk;
    int carry = 0;
    for (int b0 = 0; b0 < nblk; b0 += 64) {
        const int b = b0 + lane;
        const int c = b < nblk ? row[b] : 0;
        int incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int up = __shfl_up(incl, o, 64);
            if (lane >= o) incl += up;
        }
        if (b < nblk) row[b] = carry + incl - c;
        carry += __shfl(incl, 63, 64);
    }
    if (lane == 0) key_total[key] = carry;
}

// in-place exclusive scan of n ints (n = number of keys: a few thousand) by ONE block of 1024 threads, coalesced tiles of 1024
__global__ __launch_bounds__(1024) void exclusive_scan_kernel(int32_t* __restrict__ v, int n) {
    __shared__ int wave_tot[16];
    __shared__ int carry_s;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int i = base + threadIdx.x;
        const int c = i < n ? v[i] : 0;
        int incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int up = __shfl_up(incl, o, 64);
            if (lane >= o) incl += up;
        }
        if (lane == 63) wave_tot[wid] = incl;
        __syncthreads();
        int off = carry_s + incl - c;
        for (int w = 0; w < wid; ++w) off += wave_tot[w];
        if (i < n) v[i] = off;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = off + c;
        __syncthreads();
    }
}

__global__ __launch_bounds__(64) void token_scatter_kernel(const int64_t* __restrict__ ids, int64_t T, int64_t vocab, int nblk,
                                                           const int32_t* __restrict__ first, const int32_t* __restrict__ key_start,
                                                           int64_t* __restrict__ order, int32_t* __restrict__ sorted_key) {
    extern __shared__ int next[];  // next free sorted position of each key, for this block's tokens
    const int lane = threadIdx.x;
    const int nkeys = (int)vocab + 2;
    for (int k = lane; k < nkeys; k += 64) next[k] = key_start[k] + first[(int64_t)k * nblk + blockIdx.x];
    __syncthreads();
    const int64_t t0 = (int64_t)blockIdx.x * kOrdBlock;
    for (int c = 0; c < kOrdBlock; c += 64) {
        const int64_t t = t0 + c + lane;
        const bool live = t < T;
        int key = -1;
        if (live) {
            const int64_t id = ids[t];
            key = (int)(id < 0 ? -1 : (id > vocab ? vocab : id)) + 1;
        }
        // the lanes that share a key, found without a loop over the keys: AND over the key's bits of "lanes whose bit equals mine"
        // (14 ballots cover kOrdMaxKeys); the lowest such lane leads.  Lanes of a key, lowest token first, take consecutive positions.
        unsigned long long same = __ballot(live);
#pragma unroll
        for (int b = 0; b < 14; ++b) {
            const unsigned long long set = __ballot((key >> b) & 1);
            same &= ((key >> b) & 1) ? set : ~set;
        }
        // (a wave's LDS operations execute in order: every lane's read of next[key] precedes the leader's update; volatile keeps the
        //  compiler from carrying a value of next[] across chunks)
        if (live) {
            volatile int* nk = next + key;
            const int pos = *nk + __popcll(same & ((1ull << lane) - 1ull));
            order[pos] = t;
            sorted_key[pos] = key;
            if ((same & ((1ull << lane) - 1ull)) == 0) *nk = pos + __popcll(same);
        }
    }
}

// run_of[p] = (number of run starts at positions <= p) - 1; a run starts at p = 0, where the sorted key changes and at every multiple of
// `chunk`.  Two launches over blocks of 1024 positions: count the starts per block, then number them behind the earlier blocks' sum.
__device__ __forceinline__ bool run_starts(const int32_t* __restrict__ sorted_key, int64_t p, int chunk) {
    return p == 0 || sorted_key[p] != sorted_key[p - 1] || p % chunk == 0;
}
__global__ __launch_bounds__(1024) void token_run_count_kernel(const int32_t* __restrict__ sorted_key, int64_t T, int chunk,
                                                               int32_t* __restrict__ block_starts) {
    __shared__ int wave_tot[16];
    const int64_t p = (int64_t)blockIdx.x * 1024 + threadIdx.x;
    const unsigned long long ball = __ballot(p < T && run_starts(sorted_key, p, chunk));
    if ((threadIdx.x & 63) == 0) wave_tot[threadIdx.x >> 6] = __popcll(ball);
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0;
        for (int i = 0; i < 16; ++i) tot += wave_tot[i];
        block_starts[blockIdx.x] = tot;
    }
}
__global__ __launch_bounds__(1024) void token_run_number_kernel(const int32_t* __restrict__ sorted_key, int64_t T, int chunk,
                                                                const int32_t* __restrict__ block_starts, int32_t* __restrict__ run_of) {
    __shared__ int wave_tot[16];
    __shared__ int before_s;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (wid == 0) {  // starts in the earlier blocks
        int acc = 0;
        for (int b = lane; b < (int)blockIdx.x; b += 64) acc += block_starts[b];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        if (lane == 0) before_s = acc;
    }
    const int64_t p = (int64_t)blockIdx.x * 1024 + threadIdx.x;
    const bool st = p < T && run_starts(sorted_key, p, chunk);
    const unsigned long long ball = __ballot(st);
    if (lane == 0) wave_tot[wid] = __popcll(ball);
    __syncthreads();
    int run = before_s + __popcll(ball & ((2ull << lane) - 1ull)) - 1;  // starts at positions <= p, minus one
    for (int w = 0; w < wid; ++w) run += wave_tot[w];
    if (p < T) run_of[p] = run;
}

// run MACRO(NC) with NC = the smallest supported chunk count covering H
#define CM3P_NC_SWITCH(H, MACRO)      \
    {                                 \
        const int nc__ = ((H) + 255) / 256; \
        if (nc__ <= 1) { MACRO(1) }   \
        else if (nc__ == 2) { MACRO(2) } \
        else if (nc__ == 3) { MACRO(3) } \
        else if (nc__ == 4) { MACRO(4) } \
        else { MACRO(8) }             \
    }

inline int ln_grid(int64_t rows, int cap = 2048) {
    int64_t blocks = (rows + 3) / 4;
    if (blocks > cap) blocks = cap;  // 256 CUs x 8 blocks, grid-stride the rest
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}
// (r03 sweep at C2: 768 and 1024 workgroups 0.29 ms, 512 0.335, 1280 - a fifth workgroup per CU that 98 VGPRs do not admit - 0.36, 2048 0.32)
inline int ln_bwd_grid(int64_t rows) {
    if (const char* e = getenv("CM3P_LN_BWD_CAP")) {  // development switch (tools/overlap_ab.py)
        const int g = atoi(e);
        if (g > 0) return ln_grid(rows, g);
    }
    return ln_grid(rows, 1024);
}

}  // namespace

extern "C" {

int cm3p_layernorm_fwd(const void* x, int x_dtype, const float* weight, float* y_f32, void* y_bf16, float* mean, float* rstd,
                       int64_t rows, int H, float eps, void* stream) {
    CM3P_REQUIRE(x && weight && (y_f32 || y_bf16) && rows >= 0 && H > 0 && H % 4 == 0 && H <= 2048);
    CM3P_REQUIRE(x_dtype == CM3P_F32 || x_dtype == CM3P_BF16);
    if (rows == 0) return CM3P_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const char* env_r = getenv("CM3P_LN_REVERSE");  // development probe (tools/mall_order_probe.py): row order of the sweep
    const int reverse = env_r && env_r[0] == '1';
#define CM3P_LN_FWD(NC)                                                                                                      \
    if (x_dtype == CM3P_BF16)                                                                                                \
        layernorm_fwd_kernel<true, NC><<<ln_grid(rows), 256, 0, s>>>(x, weight, y_f32, (uint16_t*)y_bf16, mean, rstd, rows, H, eps, reverse); \
    else                                                                                                                     \
        layernorm_fwd_kernel<false, NC><<<ln_grid(rows), 256, 0, s>>>(x, weight, y_f32, (uint16_t*)y_bf16, mean, rstd, rows, H, eps, reverse);
    CM3P_NC_SWITCH(H, CM3P_LN_FWD)
#undef CM3P_LN_FWD
    CM3P_LAUNCH_CHECK();
    return CM3P_OK;
}

int cm3p_layernorm_bwd_blocks(int64_t rows) { return ln_bwd_grid(rows); }

int cm3p_layernorm_bwd(const void* dy, int dy_dtype, const float* x, const float* weight, const float* mean, const float* rstd,
                       const float* dres, float* dx_f32, void* dx_bf16, float* dw_partial, float* dw, int64_t rows, int H,
                       void* stream) {
    CM3P_REQUIRE(dy && x && weight && mean && rstd && dw_partial && dw && (dx_f32 || dx_bf16));
    CM3P_REQUIRE(rows > 0 && H > 0 && H % 4 == 0 && H <= 2048);
    CM3P_REQUIRE(dy_dtype == CM3P_F32 || dy_dtype == CM3P_BF16);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int grid = ln_bwd_grid(rows);
    const size_t lds = (size_t)4 * H * sizeof(float);
#define CM3P_LN_BWD(NC)                                                                                                      \
    if (dy_dtype == CM3P_BF16)                                                                                               \
        layernorm_bwd_kernel<true, NC><<<grid, 256, lds, s>>>(dy, x, weight, mean, rstd, dres, dx_f32, (uint16_t*)dx_bf16, dw_partial, rows, H); \
    else                                                                                                                     \
        layernorm_bwd_kernel<false, NC><<<grid, 256, lds, s>>>(dy, x, weight, mean, rstd, dres, dx_f32, (uint16_t*)dx_bf16, dw_partial, rows, H);
    CM3P_NC_SWITCH(H, CM3P_LN_BWD)
#undef CM3P_LN_BWD
    CM3P_LAUNCH_CHECK();
    colsum_kernel<<<(H + 31) / 32, 1024, 0, s>>>(dw_partial, dw, grid, H);
    CM3P_LAUNCH_CHECK();
    return CM3P_OK;
}

int cm3p_embed_ln_fwd(const int64_t* ids, const void* table, int table_dtype, const int32_t* slot, const void* override_rows,
                      int override_dtype, const float* weight, float* y_f32, void* y_bf16, float* mean, float* rstd, int64_t T,
                      int H, float eps, int64_t vocab, void* stream) {
    CM3P_REQUIRE(ids && table && weight && mean && rstd && (y_f32 || y_bf16) && T >= 0 && H > 0 && H % 4 == 0 && H <= 2048 && vocab > 0);
    CM3P_REQUIRE((slot == nullptr) == (override_rows == nullptr));
    if (T == 0) return CM3P_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int grid = ln_grid(T);
    const bool tb = table_dtype == CM3P_BF16, ob = override_dtype == CM3P_BF16;
#define CM3P_EMB_FWD_NC(NC) \
    embed_ln_fwd_kernel<TB_, OB_, NC><<<grid, 256, 0, s>>>(ids, table, slot, override_rows, weight, y_f32, (uint16_t*)y_bf16, mean, rstd, T, H, eps, vocab);
#define CM3P_EMB_FWD(TB, OB)               \
    do {                                   \
        constexpr bool TB_ = TB, OB_ = OB; \
        CM3P_NC_SWITCH(H, CM3P_EMB_FWD_NC) \
    } while (0)
    if (tb && ob) CM3P_EMB_FWD(true, true);
    else if (tb) CM3P_EMB_FWD(true, false);
    else if (ob) CM3P_EMB_FWD(false, true);
    else CM3P_EMB_FWD(false, false);
#undef CM3P_EMB_FWD
#undef CM3P_EMB_FWD_NC
    CM3P_LAUNCH_CHECK();
    return CM3P_OK;
}

int cm3p_embed_ln_bwd(const float* dy, const int64_t* ids, const void* table, int table_dtype, const int32_t* slot,
                      const void* override_rows, int override_dtype, const float* weight, const float* mean, const float* rstd,
                      float* d_table, float* d_override, float* dw_partial, float* dw, int64_t T, int H, int64_t padding_idx,
                      int64_t vocab, void* stream) {
    CM3P_REQUIRE(dy && ids && table && weight && mean && rstd && dw_partial && dw && T > 0 && H > 0 && H % 4 == 0 && H <= 2048);
    CM3P_REQUIRE((slot == nullptr) == (override_rows == nullptr));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int grid = ln_bwd_grid(T);
    const size_t lds = (size_t)4 * H * sizeof(float);
    const bool tb = table_dtype == CM3P_BF16, ob = override_dtype == CM3P_BF16;
#define CM3P_EMB_BWD_NC(NC)                                                                                              \
    embed_ln_bwd_kernel<TB_, OB_, NC><<<grid, 256, lds, s>>>(dy, ids, table, slot, override_rows, weight, mean, rstd, d_table, \
                                                             d_override, dw_partial, T, H, padding_idx, vocab);
#define CM3P_EMB_BWD(TB, OB)               \
    do {                                   \
        constexpr bool TB_ = TB, OB_ = OB; \
        CM3P_NC_SWITCH(H, CM3P_EMB_BWD_NC) \
    } while (0)
    if (tb && ob) CM3P_EMB_BWD(true, true);
    else if (tb) CM3P_EMB_BWD(true, false);
    else if (ob) CM3P_EMB_BWD(false, true);
    else CM3P_EMB_BWD(false, false);
#undef CM3P_EMB_BWD
#undef CM3P_EMB_BWD_NC
    CM3P_LAUNCH_CHECK();
    colsum_kernel<<<(H + 31) / 32, 1024, 0, s>>>(dw_partial, dw, grid, H);
    CM3P_LAUNCH_CHECK();
    return CM3P_OK;
}

int cm3p_embed_ln_bwd_sorted_chunk(void) { return kEmbChunk; }

int cm3p_embed_ln_bwd_sorted(const float* dy, const int64_t* ids, const int64_t* order, const int32_t* run_of, const void* table,
                             int table_dtype, const int32_t* slot, const void* override_rows, int override_dtype, const float* weight,
                             const float* mean, const float* rstd, float* d_table, float* d_override, float* run_rows, int64_t* run_ids,
                             float* dw_partial, float* dw, int64_t T, int H, int64_t padding_idx, int64_t vocab, void* stream) {
    CM3P_REQUIRE(dy && ids && order && run_of && table && weight && mean && rstd && d_table && run_rows && run_ids && dw_partial && dw);
    CM3P_REQUIRE(T > 0 && H > 0 && H % 4 == 0 && H <= 2048 && vocab > 0);
    CM3P_REQUIRE((slot == nullptr) == (override_rows == nullptr));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t chunks = (T + kEmbChunk - 1) / kEmbChunk;
    const int grid = (int)((chunks + 3) / 4);
    const size_t lds = (size_t)4 * H * sizeof(float);
    const bool tb = table_dtype == CM3P_BF16, ob = override_dtype == CM3P_BF16;
#define CM3P_EMB_SORT_NC(NC)                                                                                                       \
    embed_ln_bwd_sorted_kernel<TB_, OB_, NC><<<grid, 256, lds, s>>>(dy, ids, order, run_of, table, slot, override_rows, weight, mean, \
                                                                    rstd, d_override, run_rows, run_ids, dw_partial, T, H, padding_idx, vocab);
#define CM3P_EMB_SORT(TB, OB)               \
    do {                                    \
        constexpr bool TB_ = TB, OB_ = OB;  \
        CM3P_NC_SWITCH(H, CM3P_EMB_SORT_NC) \
    } while (0)
    if (tb && ob) CM3P_EMB_SORT(true, true);
    else if (tb) CM3P_EMB_SORT(true, false);
    else if (ob) CM3P_EMB_SORT(false, true);
    else CM3P_EMB_SORT(false, false);
#undef CM3P_EMB_SORT
#undef CM3P_EMB_SORT_NC
    CM3P_LAUNCH_CHECK();
#define CM3P_EMB_SUM_NC(NC) embed_run_sum_kernel<NC><<<(int)((vocab + 3) / 4), 256, 0, s>>>(run_rows, run_ids, run_of, d_table, T, H, vocab);
    CM3P_NC_SWITCH(H, CM3P_EMB_SUM_NC)
#undef CM3P_EMB_SUM_NC
    CM3P_LAUNCH_CHECK();
    colsum_kernel<<<(H + 31) / 32, 1024, 0, s>>>(dw_partial, dw, grid, H);
    CM3P_LAUNCH_CHECK();
    return CM3P_OK;
}

int64_t cm3p_token_order_workspace_ints(int64_t T, int64_t vocab) {
    if (T <= 0 || vocab <= 0 || vocab + 2 > kOrdMaxKeys) return 0;  // 0: not covered (the caller sorts by other means)
    const int64_t nblk = (T + kOrdBlock - 1) / kOrdBlock;
    return (vocab + 2) * nblk + (vocab + 2) + T + (T + 1023) / 1024;
}

int cm3p_token_order(const int64_t* ids, int64_t T, int64_t vocab, int64_t* order, int32_t* run_of, int32_t* workspace, void* stream) {
    CM3P_REQUIRE(ids && order && run_of && workspace && T > 0 && vocab > 0 && vocab + 2 <= kOrdMaxKeys && T < (int64_t(1) << 31));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int nblk = (int)((T + kOrdBlock - 1) / kOrdBlock), nkeys = (int)vocab + 2, nrb = (int)((T + 1023) / 1024);
    int32_t* cnt = workspace;                               // [nkeys][nblk]
    int32_t* key_start = cnt + (int64_t)nkeys * nblk;       // [nkeys]
    int32_t* sorted_key = key_start + nkeys;                // [T]
    int32_t* block_starts = sorted_key + T;                 // [nrb]
    const size_t lds = (size_t)nkeys * sizeof(int);
    token_hist_kernel<<<nblk, 256, lds, s>>>(ids, T, vocab, nblk, cnt);
    CM3P_LAUNCH_CHECK();
    token_key_prefix_kernel<<<(nkeys + 3) / 4, 256, 0, s>>>(cnt, key_start, nkeys, nblk);
    CM3P_LAUNCH_CHECK();
    exclusive_scan_kernel<<<1, 1024, 0, s>>>(key_start, nkeys);
    CM3P_LAUNCH_CHECK();
    token_scatter_kernel<<<nblk, 64, lds, s>>>(ids, T, vocab, nblk, cnt, key_start, order, sorted_key);
    CM3P_LAUNCH_CHECK();
    token_run_count_kernel<<<nrb, 1024, 0, s>>>(sorted_key, T, kEmbChunk, block_starts);
    CM3P_LAUNCH_CHECK();
    token_run_number_kernel<<<nrb, 1024, 0, s>>>(sorted_key, T, kEmbChunk, block_starts, run_of);
    CM3P_LAUNCH_CHECK();
    return CM3P_OK;
}

int cm3p_audio_slots(const int64_t* ids, int64_t T, int64_t audio_token_id, int32_t* slot, int32_t* count, void* stream) {
    CM3P_REQUIRE(ids && slot && count && T > 0);
    audio_slots_kernel<<<1, 1024, 0, static_cast<hipStream_t>(stream)>>>(ids, T, audio_token_id, slot, count);
    CM3P_LAUNCH_CHECK();
    return CM3P_OK;
}

}  // extern "C"
