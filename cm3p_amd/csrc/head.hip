// Contrastive head in fp32: projections, L2 normalisation, scaled similarity logits and the symmetric cross-entropy
// (ref:cm3p/modeling_cm3p.py:27-62, 958-985).  The problems are tiny (batch x 512 x 768); the kernels favour fixed
// summation order (bitwise reproducible) over throughput.
#include "common.h"

namespace {

// C[m, n] (+)= alpha * sum_k A[m*a_rs + k*a_cs] * B[n*b_rs + k*b_cs];  16 x 16 outputs per block, k tiled by 16 via LDS
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                       float* __restrict__ C, int M, int N, int K, int64_t a_rs, int64_t a_cs,
                                                       int64_t b_rs, int64_t b_cs, int64_t ldc, float alpha, int accumulate) {
    __shared__ float sa[16][17], sb[16][17];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int m0 = blockIdx.y * 16, n0 = blockIdx.x * 16;
    float acc = 0.f;
    for (int k0 = 0; k0 < K; k0 += 16) {
        // thread (ty, tx) loads A[m0+ty][k0+tx] and B[n0+ty][k0+tx]
        const int k = k0 + tx;
        sa[ty][tx] = (m0 + ty < M && k < K) ? A[(int64_t)(m0 + ty) * a_rs + (int64_t)k * a_cs] : 0.f;
        sb[ty][tx] = (n0 + ty < N && k < K) ? B[(int64_t)(n0 + ty) * b_rs + (int64_t)k * b_cs] : 0.f;
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) acc = fmaf(sa[ty][kk], sb[tx][kk], acc);
        __syncthreads();
    }
    const int m = m0 + ty, n = n0 + tx;
    if (m < M && n < N) {
        float* c = C + (int64_t)m * ldc + n;
        *c = accumulate ? *c + alpha * acc : alpha * acc;
    }
}

__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                         float* __restrict__ norm, int rows, int D) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float s = 0.f;
    for (int c = lane; c < D; c += 64) {
        const float v = x[(int64_t)row * D + c];
        s += v * v;
    }
    const float nrm = sqrtf(wave_sum(s));
    for (int c = lane; c < D; c += 64) y[(int64_t)row * D + c] = x[(int64_t)row * D + c] / nrm;
    if (lane == 0) norm[row] = nrm;
}

// y = x / n  ->  dx = (dy - y * <y, dy>) / n
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                         const float* __restrict__ norm, float* __restrict__ dx, int rows, int D) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float s = 0.f;
    for (int c = lane; c < D; c += 64) s += y[(int64_t)row * D + c] * dy[(int64_t)row * D + c];
    s = wave_sum(s);
    const float inv = 1.0f / norm[row];
    for (int c = lane; c < D; c += 64) dx[(int64_t)row * D + c] = (dy[(int64_t)row * D + c] - y[(int64_t)row * D + c] * s) * inv;
}

__device__ __forceinline__ float block_reduce(float v, float* red, bool is_max) {
    v = is_max ? wave_max(v) : wave_sum(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    const float r = is_max ? fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) : (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    return r;
}

__global__ __launch_bounds__(256) void cross_entropy_kernel(const float* __restrict__ logits, int cols, int64_t row_stride,
                                                            int64_t col_stride, const int64_t* __restrict__ row_offset,
                                                            const int64_t* __restrict__ target, float grad_scale,
                                                            float* __restrict__ loss_rows, float* dlogits) {
    __shared__ float red[4];
    const int r = blockIdx.x;
    const int64_t base = row_offset ? row_offset[r] : (int64_t)r * row_stride;
    const float* x = logits + base;
    float mx = -__builtin_huge_valf();
    for (int c = threadIdx.x; c < cols; c += 256) mx = fmaxf(mx, x[(int64_t)c * col_stride]);
    mx = block_reduce(mx, red, true);
    float se = 0.f;
    for (int c = threadIdx.x; c < cols; c += 256) se += expf(x[(int64_t)c * col_stride] - mx);
    se = block_reduce(se, red, false);
    const float lse = mx + logf(se);
    const int64_t t = target[r];
    if (threadIdx.x == 0) loss_rows[r] = lse - x[t * col_stride];
    if (dlogits) {
        float* d = dlogits + base;
        for (int c = threadIdx.x; c < cols; c += 256) {
            const float p = expf(x[(int64_t)c * col_stride] - lse);
            d[(int64_t)c * col_stride] += grad_scale * (p - (c == t ? 1.0f : 0.0f));
        }
    }
}

// Masked-LM cross entropy (nn.functional.cross_entropy(ignore_index=...) as used by ForMaskedLMLoss,
// TF:loss/loss_utils.py:32-46,74-91): rows whose target equals ignore_index contribute nothing.  dlogits (same layout,
// fully written) = grad_scale * (*inv_count) * (softmax - onehot), zero for ignored rows and for padded columns.
__global__ __launch_bounds__(256) void cross_entropy_masked_kernel(const float* __restrict__ logits, int cols, int64_t row_stride,
                                                                   const int64_t* __restrict__ target, int64_t ignore_index,
                                                                   float grad_scale, const float* __restrict__ inv_count,
                                                                   float* __restrict__ loss_rows, float* __restrict__ dlogits) {
    __shared__ float red[4];
    const int r = blockIdx.x;
    const float* x = logits + (int64_t)r * row_stride;
    float* d = dlogits ? dlogits + (int64_t)r * row_stride : nullptr;
    const int64_t t = target[r];
    if (t == ignore_index) {  // block-uniform
        if (threadIdx.x == 0) loss_rows[r] = 0.f;
        if (d)
            for (int c = threadIdx.x; c < row_stride; c += 256) d[c] = 0.f;
        return;
    }
    float mx = -__builtin_huge_valf();
    for (int c = threadIdx.x; c < cols; c += 256) mx = fmaxf(mx, x[c]);
    mx = block_reduce(mx, red, true);
    float se = 0.f;
    for (int c = threadIdx.x; c < cols; c += 256) se += expf(x[c] - mx);
    se = block_reduce(se, red, false);
    const float lse = mx + logf(se);
    if (threadIdx.x == 0) loss_rows[r] = lse - x[t];
    if (d) {
        const float g = grad_scale * inv_count[0];
        for (int c = threadIdx.x; c < row_stride; c += 256) d[c] = c < cols ? g * (expf(x[c] - lse) - (c == t ? 1.0f : 0.0f)) : 0.f;
    }
}

// inv_count[0] = 1 / max(#targets != ignore_index, 1)   (the "mean" reduction's denominator)
__global__ __launch_bounds__(256) void inv_valid_count_kernel(const int64_t* __restrict__ target, int64_t n, int64_t ignore_index,
                                                              float* __restrict__ inv_count) {
    __shared__ float red[4];
    float c = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 256) c += target[i] != ignore_index ? 1.f : 0.f;
    c = block_reduce(c, red, false);
    if (threadIdx.x == 0) inv_count[0] = 1.0f / fmaxf(c, 1.0f);
}

// x[r, c] += bias[c]
__global__ __launch_bounds__(256) void add_bias_kernel(float* __restrict__ x, const float* __restrict__ bias, int64_t rows, int cols4) {
    const int64_t total = rows * cols4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % cols4);
        reinterpret_cast<f32x4*>(x)[i] += reinterpret_cast<const f32x4*>(bias)[c];
    }
}

__global__ __launch_bounds__(256) void add_bias_scalar_kernel(float* __restrict__ x, const float* __restrict__ bias, int64_t rows, int cols) {
    const int64_t total = rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) x[i] += bias[i % cols];
}

// partial[b, c] = sum over rows r = b, b + nblk, ... of x[r, c]; then out[c] = sum_b partial[b, c] (fixed order)
__global__ __launch_bounds__(256) void colsum_rows_kernel(const float* __restrict__ x, float* __restrict__ partial, int64_t rows, int cols) {
    for (int c = threadIdx.x; c < cols; c += 256) {
        float acc = 0.f;
        for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) acc += x[r * cols + c];
        partial[(int64_t)blockIdx.x * cols + c] = acc;
    }
}
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ partial, float* __restrict__ out, int nblk, int cols) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    float s = 0.f;
    for (int b = 0; b < nblk; ++b) s += partial[(int64_t)b * cols + c];
    out[c] = s;
}

__global__ void first_zero_index_kernel(const int64_t* __restrict__ classes, int B, int V, int64_t* __restrict__ idx) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int64_t r = 0;
    for (int v = 0; v < V; ++v)
        if (classes[(int64_t)b * V + v] == 0) {
            r = v;
            break;
        }
    idx[b] = r;
}

// y[i] = x[i] * exp(*log_scale)
__global__ __launch_bounds__(256) void scale_exp_kernel(const float* __restrict__ x, const float* __restrict__ log_scale,
                                                        float* __restrict__ y, int64_t n) {
    const float s = expf(*log_scale);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) y[i] = x[i] * s;
}

// y[i] = x[i] * (*scale)
__global__ __launch_bounds__(256) void scale_by_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                       float* __restrict__ y, int64_t n) {
    const float s = *scale;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) y[i] = x[i] * s;
}

// out[0] = sum_i a[i] * b[i], single block, fixed order
__global__ __launch_bounds__(256) void dot_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                                                  int64_t n) {
    __shared__ float red[4];
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 256) s += a[i] * b[i];
    s = block_reduce(s, red, false);
    if (threadIdx.x == 0) out[0] = s;
}

// out[0] = scale * sum_i x[i]
__global__ __launch_bounds__(256) void sum_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t n, float scale,
                                                  int accumulate) {
    __shared__ float red[4];
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 256) s += x[i];
    s = block_reduce(s, red, false);
    if (threadIdx.x == 0) out[0] = accumulate ? out[0] + scale * s : scale * s;
}

// Mean element-wise losses of the classifier variant (ref:cm3p/modeling_cm3p.py:1196-1218): kind 0 = MSELoss, kind 1 =
// BCEWithLogitsLoss.  out[0] = mean_i loss(x_i, y_i); dx[i] = d mean / d x_i.  One workgroup, fixed order: n is batch * labels.
__global__ __launch_bounds__(256) void pointwise_loss_kernel(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ out,
                                                             float* __restrict__ dx, int64_t n, int kind) {
    __shared__ float red[4];
    const float inv = 1.0f / (float)n;
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 256) {
        const float a = x[i], t = y[i];
        float l, g;
        if (kind == 0) {
            const float d = a - t;
            l = d * d;
            g = 2.f * d;
        } else {
            // max(a, 0) - a t + log1p(exp(-|a|)): torch's numerically stable form (ATen binary_cross_entropy_with_logits)
            l = fmaxf(a, 0.f) - a * t + log1pf(expf(-fabsf(a)));
            g = 1.f / (1.f + expf(-a)) - t;
        }
        s += l;
        if (dx) dx[i] = g * inv;
    }
    s = block_reduce(s, red, false);
    if (threadIdx.x == 0) out[0] = s * inv;
}

}  // namespace

extern "C" {

int cm3p_pointwise_loss(const float* x, const float* y, float* out, float* dx, int64_t n, int kind, void* stream) {
    CM3P_REQUIRE(x && y && out && n > 0 && (kind == 0 || kind == 1));
    pointwise_loss_kernel<<<1, 256, 0, static_cast<hipStream_t>(stream)>>>(x, y, out, dx, n, kind);
    CM3P_LAUNCH_CHECK();
    return CM3P_OK;
}

int cm3p_gemm_f32(const float* A, const float* B, float* C, int M, int N, int K, int64_t a_rs, int64_t a_cs, int64_t b_rs,
                  int64_t b_cs, int64_t ldc, float alpha, int accumulate, void* stream) {
    CM3P_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0 && ldc >= N);
    const dim3 grid((N + 15) / 16, (M + 15) / 16);
    gemm_f32_kernel<<<grid, 256, 0, static_cast<hipStream_t>(stream)>>>(A, B, C, M, N, K, a_rs, a_cs, b_rs, b_cs, ldc, alpha, accumulate);
    CM3P_LAUNCH_CHECK();
    return CM3P_OK;
}

int cm3p_l2norm_fwd(const float* x, float* y, float* norm, int rows, int D, void* stream) {
    CM3P_REQUIRE(x && y && norm && rows > 0 && D > 0);
    l2norm_fwd_kernel<<<(rows + 3) / 4, 256, 0, static_cast<hipStream_t>(stream)>>>(x, y, norm, rows, D);
    CM3P_LAUNCH_CHECK();
    return CM3P_OK;
}

int cm3p_l2norm_bwd(const float* dy, const float* y, const float* norm, float* dx, int rows, int D, void* stream) {
    CM3P_REQUIRE(dy && y && norm && dx && rows > 0 && D > 0);
    l2norm_bwd_kernel<<<(rows + 3) / 4, 256, 0, static_cast<hipStream_t>(stream)>>>(dy, y, norm, dx, rows, D);
    CM3P_LAUNCH_CHECK();
    return CM3P_OK;
}

int cm3p_cross_entropy(const float* logits, int rows, int cols, int64_t row_stride, int64_t col_stride,
                       const int64_t* row_offset, const int64_t* target, float grad_scale, float* loss_rows, float* dlogits,
                       void* stream) {
    CM3P_REQUIRE(logits && target && loss_rows && rows > 0 && cols > 0);
    cross_entropy_kernel<<<rows, 256, 0, static_cast<hipStream_t>(stream)>>>(logits, cols, row_stride, col_stride, row_offset,
                                                                            target, grad_scale, loss_rows, dlogits);
    CM3P_LAUNCH_CHECK();
    return CM3P_OK;
}

int cm3p_cross_entropy_masked(const float* logits, int64_t rows, int cols, int64_t row_stride, const int64_t* target,
                              int64_t ignore_index, float grad_scale, const float* inv_count, float* loss_rows, float* dlogits,
                              void* stream) {
    CM3P_REQUIRE(logits && target && loss_rows && rows > 0 && cols > 0 && row_stride >= cols && (!dlogits || inv_count));
    cross_entropy_masked_kernel<<<(unsigned)rows, 256, 0, static_cast<hipStream_t>(stream)>>>(logits, cols, row_stride, target, ignore_index,
                                                                                             grad_scale, inv_count, loss_rows, dlogits);
    CM3P_LAUNCH_CHECK();
    return CM3P_OK;
}

int cm3p_inv_valid_count(const int64_t* target, int64_t n, int64_t ignore_index, float* inv_count, void* stream) {
    CM3P_REQUIRE(target && inv_count && n > 0);
    inv_valid_count_kernel<<<1, 256, 0, static_cast<hipStream_t>(stream)>>>(target, n, ignore_index, inv_count);
    CM3P_LAUNCH_CHECK();
    return CM3P_OK;
}

int cm3p_add_bias_f32(float* x, const float* bias, int64_t rows, int cols, void* stream) {
    CM3P_REQUIRE(x && bias && rows > 0 && cols > 0);
    if (cols % 4 != 0 || !cm3p_aligned16(x) || !cm3p_aligned16(bias)) {  // small classifier heads (any number of labels)
        int64_t blocks = (rows * cols + 255) / 256;
        if (blocks > 4096) blocks = 4096;
        add_bias_scalar_kernel<<<(int)blocks, 256, 0, static_cast<hipStream_t>(stream)>>>(x, bias, rows, cols);
        CM3P_LAUNCH_CHECK();
        return CM3P_OK;
    }
    int64_t blocks = (rows * (cols / 4) + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    add_bias_kernel<<<(int)blocks, 256, 0, static_cast<hipStream_t>(stream)>>>(x, bias, rows, cols / 4);
    CM3P_LAUNCH_CHECK();
    return CM3P_OK;
}

int cm3p_colsum_blocks(int64_t rows) { return (int)(rows < 512 ? (rows < 1 ? 1 : rows) : 512); }

int cm3p_colsum_f32(const float* x, float* partial, float* out, int64_t rows, int cols, void* stream) {
    CM3P_REQUIRE(x && partial && out && rows > 0 && cols > 0);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int nblk = cm3p_colsum_blocks(rows);
    colsum_rows_kernel<<<nblk, 256, 0, s>>>(x, partial, rows, cols);
    CM3P_LAUNCH_CHECK();
    colsum_final_kernel<<<(cols + 255) / 256, 256, 0, s>>>(partial, out, nblk, cols);
    CM3P_LAUNCH_CHECK();
    return CM3P_OK;
}

int cm3p_first_zero_index(const int64_t* classes, int B, int V, int64_t* idx, void* stream) {
    CM3P_REQUIRE(classes && idx && B > 0 && V > 0);
    first_zero_index_kernel<<<(B + 63) / 64, 64, 0, static_cast<hipStream_t>(stream)>>>(classes, B, V, idx);
    CM3P_LAUNCH_CHECK();
    return CM3P_OK;
}

int cm3p_scale_exp(const float* x, const float* log_scale, float* y, int64_t n, void* stream) {
    CM3P_REQUIRE(x && log_scale && y && n > 0);
    int64_t blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    scale_exp_kernel<<<(int)blocks, 256, 0, static_cast<hipStream_t>(stream)>>>(x, log_scale, y, n);
    CM3P_LAUNCH_CHECK();
    return CM3P_OK;
}

int cm3p_scale_by(const float* x, const float* scale, float* y, int64_t n, void* stream) {
    CM3P_REQUIRE(x && scale && y && n > 0);
    int64_t blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    scale_by_kernel<<<(int)blocks, 256, 0, static_cast<hipStream_t>(stream)>>>(x, scale, y, n);
    CM3P_LAUNCH_CHECK();
    return CM3P_OK;
}

int cm3p_dot_f32(const float* a, const float* b, float* out, int64_t n, void* stream) {
    CM3P_REQUIRE(a && b && out && n > 0);
    dot_kernel<<<1, 256, 0, static_cast<hipStream_t>(stream)>>>(a, b, out, n);
    CM3P_LAUNCH_CHECK();
    return CM3P_OK;
}

int cm3p_sum_f32(const float* x, float* out, int64_t n, float scale, int accumulate, void* stream) {
    CM3P_REQUIRE(x && out && n > 0);
    sum_kernel<<<1, 256, 0, static_cast<hipStream_t>(stream)>>>(x, out, n, scale, accumulate);
    CM3P_LAUNCH_CHECK();
    return CM3P_OK;
}

}  // extern "C"
