// Attention for head sizes other than 64 (16 and 32; 64 is instantiated as a cross-check of the MFMA kernels).
//
// Every tower of the reference's default configuration has head_dim 64 (768 / 12, 256 / 4, 512 / 8:
// ref:configs/model/default.yaml:16-19,88-91) and that is what attention.hip / attention_bwd*.hip are built for.  BASELINE.json's
// configs[0], the reference's own tiny test configuration (hidden 64, 4 heads: head_dim 16), used to raise NotImplementedError on the
// GPU.  These kernels make such configurations RUN with the same semantics - they are plain fp32 loops, one thread per query (or key)
// row, no matrix cores - so that a user of the reference can point any of its configurations at the library; they are not on the
// measured path.
//
// Same contract as attention.hip (TF:integrations/sdpa_attention.py:153-163 with the mask rule of TF:masking_utils.py:141-151,168-179):
//     visible(b, q, kv) = key_mask[b, kv] AND (window < 0 OR |q - kv| <= window);  rows with no visible key: exact zeros, lse = +inf.
// Layout: qkv [B, S, 3, nh, D] bf16 (q, k already rotated: cm3p_rope_apply_generic), out [B, S, nh, D] bf16, lse / delta [B, nh, S] fp32.
#include "common.h"

namespace {

constexpr float kLog2eG = 1.4426950408889634f;

template <int D>
__device__ __forceinline__ void load_row_f32(float (&dst)[D], const uint16_t* src) {
#pragma unroll
    for (int c = 0; c < D / 8; ++c) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(src + 8 * c);
#pragma unroll
        for (int j = 0; j < 8; ++j) dst[8 * c + j] = (float)v[j];
    }
}
template <int D>
__device__ __forceinline__ void store_row_bf16(uint16_t* dst, const float (&src)[D], float mul) {
#pragma unroll
    for (int c = 0; c < D / 8; ++c) {
        const uint4 w = uint4{pack_bf16x2(src[8 * c] * mul, src[8 * c + 1] * mul), pack_bf16x2(src[8 * c + 2] * mul, src[8 * c + 3] * mul),
                              pack_bf16x2(src[8 * c + 4] * mul, src[8 * c + 5] * mul), pack_bf16x2(src[8 * c + 6] * mul, src[8 * c + 7] * mul)};
        *reinterpret_cast<uint4*>(dst + 8 * c) = w;
    }
}

// one workgroup = 64 queries of one (batch, head), one thread per query; key tiles of 64 rows staged in LDS as fp32
template <int D>
__global__ __launch_bounds__(64) void attn_gen_fwd_kernel(const uint16_t* __restrict__ qkv, uint16_t* __restrict__ out, float* __restrict__ lse,
                                                          const uint8_t* __restrict__ kmask, int S, int nh, int window, float scale) {
    __shared__ float Ks[64][D + 1], Vs[64][D + 1];
    __shared__ int Ms[64];
    const int tid = threadIdx.x, head = blockIdx.y, b = blockIdx.z;
    const int Q0 = blockIdx.x * 64, q = Q0 + tid;
    const int64_t ld = (int64_t)3 * nh * D;
    const uint16_t* base = qkv + (int64_t)b * S * ld + head * D;
    const float c = scale * kLog2eG;
    float qv[D], o[D];
    load_row_f32<D>(qv, base + (int64_t)min(q, S - 1) * ld);
#pragma unroll
    for (int d = 0; d < D; ++d) o[d] = 0.f;
    float m = -__builtin_huge_valf(), l = 0.f;
    const int klo = window < 0 ? 0 : max(0, Q0 - window), khi = window < 0 ? S - 1 : min(S - 1, Q0 + 63 + window);
    for (int t = klo / 64; t <= khi / 64; ++t) {
        const int key = t * 64 + tid;
        {
            float kr[D], vr[D];
            const int kc = min(key, S - 1);
            load_row_f32<D>(kr, base + (int64_t)kc * ld + nh * D);
            load_row_f32<D>(vr, base + (int64_t)kc * ld + 2 * nh * D);
#pragma unroll
            for (int d = 0; d < D; ++d) {
                Ks[tid][d] = kr[d];
                Vs[tid][d] = vr[d];
            }
            Ms[tid] = key < S && (kmask ? kmask[(int64_t)b * S + key] != 0 : true);
        }
        __syncthreads();
        for (int k = 0; k < 64; ++k) {
            if (!Ms[k]) continue;  // (uniform)
            const int kk = t * 64 + k;
            if (window >= 0 && (kk < q - window || kk > q + window)) continue;
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < D; ++d) s = __builtin_fmaf(qv[d], Ks[k][d], s);
            s *= c;
            if (s > m) {  // new reference point: rescale what has been summed
                const float a = __builtin_amdgcn_exp2f(m - s);  // (m = -inf: 0)
                l *= a;
#pragma unroll
                for (int d = 0; d < D; ++d) o[d] *= a;
                m = s;
            }
            const float p = __builtin_amdgcn_exp2f(s - m);
            l += p;
#pragma unroll
            for (int d = 0; d < D; ++d) o[d] = __builtin_fmaf(p, Vs[k][d], o[d]);
        }
        __syncthreads();
    }
    if (q < S) {
        const float inv = l > 0.f ? 1.0f / l : 0.f;
        store_row_bf16<D>(out + ((int64_t)b * S + q) * nh * D + head * D, o, inv);
        lse[((int64_t)b * nh + head) * S + q] = l > 0.f ? (m + __log2f(l)) * 0.69314718055994531f : __builtin_huge_valf();
    }
}

// dq (gradient w.r.t. the rotated q) and delta[q] = sum_d dO[q, d] O[q, d]; one thread per query row
template <int D>
__global__ __launch_bounds__(64) void attn_gen_dq_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ o_rows,
                                                         const uint16_t* __restrict__ d_o, const float* __restrict__ lse,
                                                         float* __restrict__ delta, uint16_t* __restrict__ dqkv, const uint8_t* __restrict__ kmask,
                                                         int S, int nh, int window, float scale) {
    __shared__ float Ks[64][D + 1], Vs[64][D + 1];
    __shared__ int Ms[64];
    const int tid = threadIdx.x, head = blockIdx.y, b = blockIdx.z;
    const int Q0 = blockIdx.x * 64, q = Q0 + tid, qc = min(q, S - 1);
    const int64_t ld = (int64_t)3 * nh * D, ldo = (int64_t)nh * D;
    const uint16_t* base = qkv + (int64_t)b * S * ld + head * D;
    const float c = scale * kLog2eG;
    float qv[D], dov[D], dq[D];
    load_row_f32<D>(qv, base + (int64_t)qc * ld);
    load_row_f32<D>(dov, d_o + ((int64_t)b * S + qc) * ldo + head * D);
    float dlt = 0.f;
    {
        float ov[D];
        load_row_f32<D>(ov, o_rows + ((int64_t)b * S + qc) * ldo + head * D);
#pragma unroll
        for (int d = 0; d < D; ++d) dlt = __builtin_fmaf(ov[d], dov[d], dlt);
    }
    const int64_t stat = ((int64_t)b * nh + head) * S + qc;
    const float lse2 = lse[stat] * kLog2eG;  // +inf (no visible key): p = 0 everywhere
    if (q < S) delta[stat] = dlt;
#pragma unroll
    for (int d = 0; d < D; ++d) dq[d] = 0.f;
    const int klo = window < 0 ? 0 : max(0, Q0 - window), khi = window < 0 ? S - 1 : min(S - 1, Q0 + 63 + window);
    for (int t = klo / 64; t <= khi / 64; ++t) {
        const int key = t * 64 + tid;
        {
            float kr[D], vr[D];
            const int kc = min(key, S - 1);
            load_row_f32<D>(kr, base + (int64_t)kc * ld + nh * D);
            load_row_f32<D>(vr, base + (int64_t)kc * ld + 2 * nh * D);
#pragma unroll
            for (int d = 0; d < D; ++d) {
                Ks[tid][d] = kr[d];
                Vs[tid][d] = vr[d];
            }
            Ms[tid] = key < S && (kmask ? kmask[(int64_t)b * S + key] != 0 : true);
        }
        __syncthreads();
        for (int k = 0; k < 64; ++k) {
            if (!Ms[k]) continue;
            const int kk = t * 64 + k;
            if (window >= 0 && (kk < q - window || kk > q + window)) continue;
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                s = __builtin_fmaf(qv[d], Ks[k][d], s);
                dp = __builtin_fmaf(dov[d], Vs[k][d], dp);
            }
            const float p = __builtin_amdgcn_exp2f(s * c - lse2);
            const float ds = p * (dp - dlt);
#pragma unroll
            for (int d = 0; d < D; ++d) dq[d] = __builtin_fmaf(ds, Ks[k][d], dq[d]);
        }
        __syncthreads();
    }
    if (q < S) store_row_bf16<D>(dqkv + ((int64_t)b * S + q) * ld + head * D, dq, scale);
}

// dk, dv: one thread per key row; query tiles of 64 rows (q, dO, lse, delta) staged in LDS
template <int D>
__global__ __launch_bounds__(64) void attn_gen_dkv_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ d_o,
                                                          const float* __restrict__ lse, const float* __restrict__ delta,
                                                          uint16_t* __restrict__ dqkv, const uint8_t* __restrict__ kmask, int S, int nh, int window,
                                                          float scale) {
    __shared__ float Qs[64][D + 1], Gs[64][D + 1];
    __shared__ float Ls[64], Ds[64];
    const int tid = threadIdx.x, head = blockIdx.y, b = blockIdx.z;
    const int K0 = blockIdx.x * 64, key = K0 + tid, kc = min(key, S - 1);
    const int64_t ld = (int64_t)3 * nh * D, ldo = (int64_t)nh * D;
    const uint16_t* base = qkv + (int64_t)b * S * ld + head * D;
    const float c = scale * kLog2eG;
    float kv[D], vv[D], dk[D], dv[D];
    load_row_f32<D>(kv, base + (int64_t)kc * ld + nh * D);
    load_row_f32<D>(vv, base + (int64_t)kc * ld + 2 * nh * D);
#pragma unroll
    for (int d = 0; d < D; ++d) dk[d] = dv[d] = 0.f;
    const bool key_ok = key < S && (kmask ? kmask[(int64_t)b * S + key] != 0 : true);
    const int qlo = window < 0 ? 0 : max(0, K0 - window), qhi = window < 0 ? S - 1 : min(S - 1, K0 + 63 + window);
    for (int t = qlo / 64; t <= qhi / 64; ++t) {
        const int qr = t * 64 + tid;
        {
            float a[D], g[D];
            const int qc = min(qr, S - 1);
            load_row_f32<D>(a, base + (int64_t)qc * ld);
            load_row_f32<D>(g, d_o + ((int64_t)b * S + qc) * ldo + head * D);
#pragma unroll
            for (int d = 0; d < D; ++d) {
                Qs[tid][d] = a[d];
                Gs[tid][d] = g[d];
            }
            const int64_t stat = ((int64_t)b * nh + head) * S + qc;
            Ls[tid] = qr < S ? lse[stat] * kLog2eG : __builtin_huge_valf();  // rows past the sequence: p = 0
            Ds[tid] = qr < S ? delta[stat] : 0.f;
        }
        __syncthreads();
        if (key_ok) {
            for (int i = 0; i < 64; ++i) {
                const int qq = t * 64 + i;
                if (window >= 0 && (qq < key - window || qq > key + window)) continue;
                float s = 0.f, dp = 0.f;
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    s = __builtin_fmaf(Qs[i][d], kv[d], s);
                    dp = __builtin_fmaf(Gs[i][d], vv[d], dp);
                }
                const float p = __builtin_amdgcn_exp2f(s * c - Ls[i]);
                const float ds = p * (dp - Ds[i]);
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    dv[d] = __builtin_fmaf(p, Gs[i][d], dv[d]);
                    dk[d] = __builtin_fmaf(ds, Qs[i][d], dk[d]);
                }
            }
        }
        __syncthreads();
    }
    if (key < S) {
        uint16_t* dst = dqkv + ((int64_t)b * S + key) * ld + nh * D + head * D;
        store_row_bf16<D>(dst, dk, scale);
        store_row_bf16<D>(dst + nh * D, dv, 1.0f);
    }
}

// apply_rotary_pos_emb (TF:models/modernbert/modeling_modernbert.py:188-219) in place on the q and k thirds of a packed qkv
// [T, 3, nh, D]: dims (j, j + D / 2) are a pair (rotate_half convention), fp32 arithmetic on the bf16 values, one rounding.
template <bool INVERSE>
__global__ __launch_bounds__(256) void rope_gen_kernel(uint16_t* __restrict__ qkv, const float* __restrict__ cos_tab,
                                                       const float* __restrict__ sin_tab, int64_t T, int S, int nh, int D,
                                                       int64_t pos_batch_stride) {
    const int half = D / 2;
    const int64_t per_tok = (int64_t)2 * nh * half, total = T * per_tok;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t t = i / per_tok;
        const int r = (int)(i % per_tok);
        const int part = r / (nh * half), h = (r / half) % nh, j = r % half;
        const int64_t b = t / S, s = t % S;
        const int64_t prow = (pos_batch_stride ? b * pos_batch_stride : 0) + s;
        const float cs = cos_tab[prow * half + j], sn = sin_tab[prow * half + j];
        uint16_t* p = qkv + t * 3 * nh * D + (int64_t)part * nh * D + h * D + j;
        const float x1 = bf16_bits_to_f32(p[0]), x2 = bf16_bits_to_f32(p[half]);
        float y1, y2;
        if constexpr (!INVERSE) {
            y1 = x1 * cs - x2 * sn;
            y2 = x2 * cs + x1 * sn;
        } else {
            y1 = x1 * cs + x2 * sn;
            y2 = x2 * cs - x1 * sn;
        }
        p[0] = f32_to_bf16_bits(y1);
        p[half] = f32_to_bf16_bits(y2);
    }
}

}  // namespace

extern "C" {

int cm3p_attn_generic_supported(int head_dim) { return head_dim == 16 || head_dim == 32 || head_dim == 64; }

int cm3p_attn_fwd_generic(const void* qkv, void* out, float* lse, const uint8_t* key_mask, int B, int S, int nh, int head_dim, int window,
                          float scale, void* stream) {
    CM3P_REQUIRE(qkv && out && lse && B > 0 && S > 0 && nh > 0 && scale > 0.f && cm3p_attn_generic_supported(head_dim));
    CM3P_REQUIRE(cm3p_aligned16(qkv) && cm3p_aligned16(out));
    const dim3 grid((S + 63) / 64, nh, B);
    hipStream_t s = static_cast<hipStream_t>(stream);
#define CM3P_GEN_FWD(DD) attn_gen_fwd_kernel<DD><<<grid, 64, 0, s>>>((const uint16_t*)qkv, (uint16_t*)out, lse, key_mask, S, nh, window, scale)
    if (head_dim == 16) CM3P_GEN_FWD(16);
    else if (head_dim == 32) CM3P_GEN_FWD(32);
    else CM3P_GEN_FWD(64);
#undef CM3P_GEN_FWD
    CM3P_LAUNCH_CHECK();
    return CM3P_OK;
}

int cm3p_attn_bwd_generic(const void* qkv, const void* out, const void* dout, const float* lse, float* delta, void* dqkv,
                          const uint8_t* key_mask, int B, int S, int nh, int head_dim, int window, float scale, void* stream) {
    CM3P_REQUIRE(qkv && out && dout && lse && delta && dqkv && B > 0 && S > 0 && nh > 0 && scale > 0.f && cm3p_attn_generic_supported(head_dim));
    CM3P_REQUIRE(cm3p_aligned16(qkv) && cm3p_aligned16(out) && cm3p_aligned16(dout) && cm3p_aligned16(dqkv));
    const dim3 grid((S + 63) / 64, nh, B);
    hipStream_t s = static_cast<hipStream_t>(stream);
#define CM3P_GEN_BWD(DD)                                                                                                                   \
    attn_gen_dq_kernel<DD><<<grid, 64, 0, s>>>((const uint16_t*)qkv, (const uint16_t*)out, (const uint16_t*)dout, lse, delta, (uint16_t*)dqkv, \
                                               key_mask, S, nh, window, scale);                                                            \
    attn_gen_dkv_kernel<DD><<<grid, 64, 0, s>>>((const uint16_t*)qkv, (const uint16_t*)dout, lse, delta, (uint16_t*)dqkv, key_mask, S, nh, window, scale)
    if (head_dim == 16) { CM3P_GEN_BWD(16); }
    else if (head_dim == 32) { CM3P_GEN_BWD(32); }
    else { CM3P_GEN_BWD(64); }
#undef CM3P_GEN_BWD
    CM3P_LAUNCH_CHECK();
    return CM3P_OK;
}

int cm3p_rope_apply_generic(void* qkv, const float* cos_tab, const float* sin_tab, int B, int S, int nh, int head_dim, int64_t pos_batch_stride,
                            int inverse, void* stream) {
    CM3P_REQUIRE(qkv && cos_tab && sin_tab && B > 0 && S > 0 && nh > 0 && head_dim > 0 && head_dim % 2 == 0);
    CM3P_REQUIRE(pos_batch_stride == 0 || pos_batch_stride == S);
    const int64_t T = (int64_t)B * S, n = T * 2 * nh * (head_dim / 2);
    int64_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (inverse) rope_gen_kernel<true><<<(int)blocks, 256, 0, s>>>((uint16_t*)qkv, cos_tab, sin_tab, T, S, nh, head_dim, pos_batch_stride);
    else rope_gen_kernel<false><<<(int)blocks, 256, 0, s>>>((uint16_t*)qkv, cos_tab, sin_tab, T, S, nh, head_dim, pos_batch_stride);
    CM3P_LAUNCH_CHECK();
    return CM3P_OK;
}

}  // extern "C"
