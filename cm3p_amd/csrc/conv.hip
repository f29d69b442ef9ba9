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

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    return cdf + x * 0.39894228040143267794f * __expf(-0.5f * x * x);
}

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
        const f32x4 g = {gelu_erf(v.x), gelu_erf(v.y), gelu_erf(v.z), gelu_erf(v.w)};
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

__global__ __launch_bounds__(256) void colsum2_kernel(const float* __restrict__ partial, float* __restrict__ out, int nblk, int C) {
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= C) return;
    float s = 0.f;
    for (int b = 0; b < nblk; ++b) s += partial[(int64_t)b * C + col];
    out[col] = s;
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
    if (da_dtype == CM3P_BF16) bias_gelu_bwd_kernel<true><<<grid, 256, 0, s>>>(da, z, bias, (uint16_t*)dz_bf16, db_partial, R, C);
    else bias_gelu_bwd_kernel<false><<<grid, 256, 0, s>>>(da, z, bias, (uint16_t*)dz_bf16, db_partial, R, C);
    CM3P_LAUNCH_CHECK();
    colsum2_kernel<<<(C + 255) / 256, 256, 0, s>>>(db_partial, dbias, grid, C);
    CM3P_LAUNCH_CHECK();
    return CM3P_OK;
}

}  // extern "C"
