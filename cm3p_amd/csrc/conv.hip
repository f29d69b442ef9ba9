// Audio front end of the beatmap tower: the two Whisper-style Conv1d(k=3) layers as im2col + bf16 MFMA GEMM, with the
// bias + exact-erf GELU applied in a separate fused pass.  Restates CM3PAudioEncoder.forward
// (ref:cm3p/modeling_cm3p.py:488-489,501-504): gelu(conv1(x)), gelu(conv2(.)) with padding 1, stride 1 / 2, then
// permute(0,2,1) - here the activations are token-major [B, T, C] from the start, so the permute disappears.
//
// Patch layout matches the Conv1d weight [C_out, C_in, 3] flattened to [C_out, C_in*3]: column c*3 + kk holds
// input channel c at time t*stride + kk - 1 (zero outside [0, T_in)).
#include "common.h"

namespace {

inline int cv_grid(int64_t items) {
    int64_t blocks = (items + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

// (the GELU: gelu_erf2 / gelu_erf_grad2 in common.h, two values per call)
__device__ __forceinline__ float gelu_erf_grad(float x) { return gelu_erf_grad2(f32x2{x, x}).x; }

// x: [B, C, T_in] fp32 (channel-major, the model input) -> patches [B*T_out, C*3] bf16
__global__ __launch_bounds__(256) void im2col_cm_kernel(const float* __restrict__ x, uint16_t* __restrict__ p, int B, int C,
                                                        int T_in, int T_out, int stride) {
    const int64_t total = (int64_t)B * T_out * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        // t fastest across lanes: coalesced reads along time
        const int t = (int)(i % T_out);
        const int c = (int)((i / T_out) % C);
        const int b = (int)(i / ((int64_t)T_out * C));
        const float* src = x + ((int64_t)b * C + c) * T_in;
        uint16_t* dst = p + ((int64_t)b * T_out + t) * (C * 3) + c * 3;
#pragma unroll
        for (int kk = 0; kk < 3; ++kk) {
            const int ti = t * stride + kk - 1;
            dst[kk] = f32_to_bf16_bits((ti >= 0 && ti < T_in) ? src[ti] : 0.f);
        }
    }
}

// a: [B, T_in, C] bf16 (token-major) -> patches [B*T_out, C*3] bf16
__global__ __launch_bounds__(256) void im2col_tm_kernel(const uint16_t* __restrict__ a, uint16_t* __restrict__ p, int B, int C,
                                                        int T_in, int T_out, int stride) {
    const int64_t total = (int64_t)B * T_out * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C);  // c fastest: coalesced reads along channels
        const int t = (int)((i / C) % T_out);
        const int b = (int)(i / ((int64_t)T_out * C));
        uint16_t* dst = p + ((int64_t)b * T_out + t) * (C * 3) + c * 3;
#pragma unroll
        for (int kk = 0; kk < 3; ++kk) {
            const int ti = t * stride + kk - 1;
            dst[kk] = (ti >= 0 && ti < T_in) ? a[((int64_t)b * T_in + ti) * C + c] : (uint16_t)0;
        }
    }
}

// transpose of im2col_tm: da[b, ti, c] = sum over (t, kk) with t*stride + kk - 1 == ti of dp[b, t, c*3 + kk]
__global__ __launch_bounds__(256) void col2im_tm_kernel(const uint16_t* __restrict__ dp, uint16_t* __restrict__ da, int B, int C,
                                                        int T_in, int T_out, int stride) {
    const int64_t total = (int64_t)B * T_in * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        const int ti = (int)((i / C) % T_in);
        const int b = (int)(i / ((int64_t)T_in * C));
        float s = 0.f;
#pragma unroll
        for (int kk = 0; kk < 3; ++kk) {
            const int num = ti + 1 - kk;
            if (num >= 0 && num % stride == 0) {
                const int t = num / stride;
                if (t < T_out) s += bf16_bits_to_f32(dp[((int64_t)b * T_out + t) * (C * 3) + c * 3 + kk]);
            }
        }
        da[i] = f32_to_bf16_bits(s);
    }
}

// a = gelu(z + bias); z fp32 [R, C]; writes bf16 and/or fp32
__global__ __launch_bounds__(256) void bias_gelu_fwd_kernel(const float* __restrict__ z, const float* __restrict__ bias,
                                                            uint16_t* __restrict__ a16, float* __restrict__ a32, int64_t R, int C) {
    const int c4 = C / 4;
    const int64_t total = R * c4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int col = (int)(i % c4) * 4;
        const f32x4 v = reinterpret_cast<const f32x4*>(z)[i] + *reinterpret_cast<const f32x4*>(bias + col);
        const f32x2 g01 = gelu_erf2(f32x2{v.x, v.y}), g23 = gelu_erf2(f32x2{v.z, v.w});
        const f32x4 g = {g01.x, g01.y, g23.x, g23.y};
        if (a32) reinterpret_cast<f32x4*>(a32)[i] = g;
        if (a16) reinterpret_cast<uint2*>(a16)[i] = uint2{pack_bf16x2(g.x, g.y), pack_bf16x2(g.z, g.w)};
    }
}

// dz = da * gelu'(z + bias) -> bf16; per-block partial column sums of dz for the bias gradient
template <bool DA_BF16>
__global__ __launch_bounds__(256) void bias_gelu_bwd_kernel(const void* __restrict__ da, const float* __restrict__ z,
                                                            const float* __restrict__ bias, uint16_t* __restrict__ dz,
                                                            float* __restrict__ db_partial, int64_t R, int C) {
    // block handles rows r = blockIdx.x, + gridDim.x, ...; thread handles columns tid, tid + 256, ...
    for (int col = threadIdx.x; col < C; col += 256) {
        const float bv = bias[col];
        float acc = 0.f;
        for (int64_t r = blockIdx.x; r < R; r += gridDim.x) {
            const int64_t i = r * C + col;
            const float g = DA_BF16 ? bf16_bits_to_f32(static_cast<const uint16_t*>(da)[i]) : static_cast<const float*>(da)[i];
            const float d = g * gelu_erf_grad(z[i] + bv);
            const uint16_t q = f32_to_bf16_bits(d);
            dz[i] = q;
            acc += bf16_bits_to_f32(q);  // the bias gradient sums exactly what the GEMMs downstream will see
        }
        db_partial[(int64_t)blockIdx.x * C + col] = acc;
    }
}

// Same, four columns per thread (16-byte reads of z and fp32 da, 8-byte reads of bf16 da, 8-byte stores) and two rows in flight:
// the one-element-per-thread version above ran at ~1.5 TB/s.  C % 4 == 0, C <= 4096.
template <bool DA_BF16>
__global__ __launch_bounds__(256) void bias_gelu_bwd_vec_kernel(const void* __restrict__ da, const float* __restrict__ z,
                                                                const float* __restrict__ bias, uint16_t* __restrict__ dz,
                                                                float* __restrict__ db_partial, int64_t R, int C) {
    constexpr int MAXQ = 4;
    const int quads = C / 4;
    f32x4 acc[MAXQ], bv[MAXQ];
#pragma unroll
    for (int k = 0; k < MAXQ; ++k) {
        const int qd = threadIdx.x + 256 * k;
        acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        bv[k] = qd < quads ? *reinterpret_cast<const f32x4*>(bias + 4 * qd) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    auto load_da = [&](int64_t i) -> f32x4 {
        if constexpr (DA_BF16) {
            const uint2 w = *reinterpret_cast<const uint2*>(static_cast<const uint16_t*>(da) + i);
            return f32x4{bf16_bits_to_f32((uint16_t)(w.x & 0xffffu)), bf16_bits_to_f32((uint16_t)(w.x >> 16)),
                         bf16_bits_to_f32((uint16_t)(w.y & 0xffffu)), bf16_bits_to_f32((uint16_t)(w.y >> 16))};
        } else {
            return *reinterpret_cast<const f32x4*>(static_cast<const float*>(da) + i);
        }
    };
    auto one = [&](int64_t i, f32x4 g, f32x4 zz, int k) {
        uint16_t qb[4];
#pragma unroll
        for (int j = 0; j < 4; j += 2) {
            const f32x2 gr = gelu_erf_grad2(f32x2{zz[j] + bv[k][j], zz[j + 1] + bv[k][j + 1]});
            qb[j] = f32_to_bf16_bits(g[j] * gr.x);
            qb[j + 1] = f32_to_bf16_bits(g[j + 1] * gr.y);
            acc[k][j] += bf16_bits_to_f32(qb[j]);  // the bias gradient sums exactly what the GEMMs downstream will see
            acc[k][j + 1] += bf16_bits_to_f32(qb[j + 1]);
        }
        *reinterpret_cast<uint2*>(dz + i) = uint2{(uint32_t)qb[0] | ((uint32_t)qb[1] << 16), (uint32_t)qb[2] | ((uint32_t)qb[3] << 16)};
    };
    const int64_t step = gridDim.x;
    int64_t r = blockIdx.x;
    for (; r + step < R; r += 2 * step) {
#pragma unroll
        for (int k = 0; k < MAXQ; ++k) {
            const int qd = threadIdx.x + 256 * k;
            if (qd < quads) {
                const int64_t i0 = r * C + 4 * qd, i1 = (r + step) * C + 4 * qd;
                const f32x4 g0 = load_da(i0), g1 = load_da(i1);
                const f32x4 z0 = *reinterpret_cast<const f32x4*>(z + i0), z1 = *reinterpret_cast<const f32x4*>(z + i1);
                one(i0, g0, z0, k);
                one(i1, g1, z1, k);
            }
        }
    }
    for (; r < R; r += step) {
#pragma unroll
        for (int k = 0; k < MAXQ; ++k) {
            const int qd = threadIdx.x + 256 * k;
            if (qd < quads) {
                const int64_t i0 = r * C + 4 * qd;
                one(i0, load_da(i0), *reinterpret_cast<const f32x4*>(z + i0), k);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < MAXQ; ++k) {
        const int qd = threadIdx.x + 256 * k;
        if (qd < quads) *reinterpret_cast<f32x4*>(db_partial + (int64_t)blockIdx.x * C + 4 * qd) = acc[k];
    }
}

__global__ __launch_bounds__(256) void colsum2_kernel(const float* __restrict__ partial, float* __restrict__ out, int nblk, int C) {
    // 64 columns per workgroup; wave g sums blocks [g*per, (g+1)*per) with four loads in flight, then the four waves' sums are
    // added in a fixed order (a single thread walking all nblk partials was latency-bound: 0.6 ms for 2048 partials)
    __shared__ float red[4][64];
    const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + c;
    const int per = (nblk + 3) / 4;
    const int b0 = g * per, b1 = b0 + per < nblk ? b0 + per : nblk;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (col < C) {
        int b = b0;
        for (; b + 3 < b1; b += 4) {
            const float v0 = partial[(int64_t)b * C + col], v1 = partial[(int64_t)(b + 1) * C + col];
            const float v2 = partial[(int64_t)(b + 2) * C + col], v3 = partial[(int64_t)(b + 3) * C + col];
            s0 += v0;
            s1 += v1;
            s2 += v2;
            s3 += v3;
        }
        for (; b < b1; ++b) s0 += partial[(int64_t)b * C + col];
    }
    red[g][c] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (g == 0 && col < C) out[col] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}

}  // namespace

extern "C" {

int cm3p_im2col_k3(const void* x, int x_token_major, void* patches, int B, int C, int T_in, int T_out, int stride, void* stream) {
    CM3P_REQUIRE(x && patches && B > 0 && C > 0 && T_in > 0 && T_out > 0 && (stride == 1 || stride == 2));
    CM3P_REQUIRE(T_out == (T_in + 2 - 3) / stride + 1);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int grid = cv_grid((int64_t)B * T_out * C);
    if (x_token_major) im2col_tm_kernel<<<grid, 256, 0, s>>>((const uint16_t*)x, (uint16_t*)patches, B, C, T_in, T_out, stride);
    else im2col_cm_kernel<<<grid, 256, 0, s>>>((const float*)x, (uint16_t*)patches, B, C, T_in, T_out, stride);
    CM3P_LAUNCH_CHECK();
    return CM3P_OK;
}

int cm3p_col2im_k3(const void* dpatches, void* dx, int B, int C, int T_in, int T_out, int stride, void* stream) {
    CM3P_REQUIRE(dpatches && dx && B > 0 && C > 0 && T_in > 0 && T_out > 0 && (stride == 1 || stride == 2));
    col2im_tm_kernel<<<cv_grid((int64_t)B * T_in * C), 256, 0, static_cast<hipStream_t>(stream)>>>(
        (const uint16_t*)dpatches, (uint16_t*)dx, B, C, T_in, T_out, stride);
    CM3P_LAUNCH_CHECK();
    return CM3P_OK;
}

int cm3p_bias_gelu_fwd(const float* z, const float* bias, void* a_bf16, float* a_f32, int64_t R, int C, void* stream) {
    CM3P_REQUIRE(z && bias && (a_bf16 || a_f32) && R > 0 && C > 0 && C % 4 == 0);
    bias_gelu_fwd_kernel<<<cv_grid(R * (C / 4)), 256, 0, static_cast<hipStream_t>(stream)>>>(z, bias, (uint16_t*)a_bf16, a_f32, R, C);
    CM3P_LAUNCH_CHECK();
    return CM3P_OK;
}

int cm3p_bias_gelu_bwd_blocks(int64_t R) { return (int)(R < 512 ? (R < 1 ? 1 : R) : 512); }

int cm3p_bias_gelu_bwd(const void* da, int da_dtype, const float* z, const float* bias, void* dz_bf16, float* db_partial,
                       float* dbias, int64_t R, int C, void* stream) {
    CM3P_REQUIRE(da && z && bias && dz_bf16 && db_partial && dbias && R > 0 && C > 0);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int grid = cm3p_bias_gelu_bwd_blocks(R);
    const bool vec = C % 4 == 0 && C <= 4096 && cm3p_aligned16(z) && cm3p_aligned16(bias) && cm3p_aligned16(db_partial) &&
                     cm3p_aligned16(da) && (reinterpret_cast<uintptr_t>(dz_bf16) & 7) == 0;
    if (vec && da_dtype == CM3P_BF16) bias_gelu_bwd_vec_kernel<true><<<grid, 256, 0, s>>>(da, z, bias, (uint16_t*)dz_bf16, db_partial, R, C);
    else if (vec) bias_gelu_bwd_vec_kernel<false><<<grid, 256, 0, s>>>(da, z, bias, (uint16_t*)dz_bf16, db_partial, R, C);
    else if (da_dtype == CM3P_BF16) bias_gelu_bwd_kernel<true><<<grid, 256, 0, s>>>(da, z, bias, (uint16_t*)dz_bf16, db_partial, R, C);
    else bias_gelu_bwd_kernel<false><<<grid, 256, 0, s>>>(da, z, bias, (uint16_t*)dz_bf16, db_partial, R, C);
    CM3P_LAUNCH_CHECK();
    colsum2_kernel<<<(C + 63) / 64, 256, 0, s>>>(db_partial, dbias, grid, C);
    CM3P_LAUNCH_CHECK();
    return CM3P_OK;
}

}  // extern "C"
