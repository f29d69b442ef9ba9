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
    if (threadIdx.x == 0) loss_rows[r] = (uint64_t)t < (uint64_t)cols ? lse - x[t * col_stride] : 0.f;  // (a target outside the row is never read)
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
    if (t == ignore_index || (uint64_t)t >= (uint64_t)cols) {  // block-uniform; a label outside [0, cols) is ignored too, never read
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

// Masked-LM loss in two kernels that never materialise an fp32 gradient of the logits (1.66 GB at the C2 token count):
//   stats:   per labelled row, loss = lse - x[target] and lse; ignored rows cost nothing (no read of their logits).
//   dlogits: dl[r, c] = bf16(s * (exp(x[r, c] - lse[r]) - [c == target[r]])) for labelled rows, zero elsewhere and in the pad
//            columns, s = scale_a[0] * scale_b[0] (the incoming loss gradient times 1 / #labelled, both device scalars), written
//            in the bf16 layout the decoder's dgrad / wgrad GEMMs read; partial[blk, c] = the workgroup's column sums of the
//            unrounded values (the decoder bias gradient), reduced afterwards in a fixed order.
__global__ __launch_bounds__(256) void ce_masked_stats_kernel(const float* __restrict__ logits, int cols, int64_t row_stride,
                                                              const int64_t* __restrict__ target, int64_t ignore_index,
                                                              float* __restrict__ loss_rows, float* __restrict__ lse_rows) {
    __shared__ float red[4];
    const int r = blockIdx.x;
    const int64_t t = target[r];
    if (t == ignore_index || (uint64_t)t >= (uint64_t)cols) {  // block-uniform; a label outside [0, cols) is ignored too, never read
        if (threadIdx.x == 0) {
            loss_rows[r] = 0.f;
            lse_rows[r] = 0.f;
        }
        return;
    }
    const float* x = logits + (int64_t)r * row_stride;
    float mx = -__builtin_huge_valf();
    for (int c = threadIdx.x; c < cols; c += 256) mx = fmaxf(mx, x[c]);
    mx = block_reduce(mx, red, true);
    float se = 0.f;
    for (int c = threadIdx.x; c < cols; c += 256) se += expf(x[c] - mx);
    se = block_reduce(se, red, false);
    const float lse = mx + logf(se);
    if (threadIdx.x == 0) {
        loss_rows[r] = lse - x[t];
        lse_rows[r] = lse;
    }
}

constexpr int kDlRows = 64;   // rows per workgroup
constexpr int kDlQuads = 4;   // 4-column groups per thread: a workgroup covers a window of 4 * 256 * 4 = 4096 columns (grid.y windows)

__global__ __launch_bounds__(256) void ce_masked_dlogits_kernel(const float* __restrict__ logits, int cols, int64_t pitch, int64_t rows,
                                                                const int64_t* __restrict__ target, int64_t ignore_index,
                                                                const float* __restrict__ lse_rows, const float* __restrict__ scale_a,
                                                                const float* __restrict__ scale_b, uint16_t* __restrict__ dl,
                                                                float* __restrict__ partial) {
    const float s = scale_a[0] * scale_b[0];
    const int quads = (int)(pitch / 4);
    const int qbase = blockIdx.y * 256 * kDlQuads;  // this workgroup's window of 4-column groups (4096 columns per window)
    f32x4 acc[kDlQuads];
#pragma unroll
    for (int q = 0; q < kDlQuads; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int64_t r0 = (int64_t)blockIdx.x * kDlRows;
    const int64_t r1 = r0 + kDlRows < rows ? r0 + kDlRows : rows;
    // the row loop is serial: fetch its 64 targets and lse values in one go instead of one dependent load per row
    __shared__ int64_t tg[kDlRows];
    __shared__ float ls[kDlRows];
    if (threadIdx.x < kDlRows && r0 + threadIdx.x < rows) {
        tg[threadIdx.x] = target[r0 + threadIdx.x];
        ls[threadIdx.x] = lse_rows[r0 + threadIdx.x];
    }
    __syncthreads();
    for (int64_t r = r0; r < r1; ++r) {
        const int64_t t = tg[r - r0];  // workgroup-uniform
        uint16_t* drow = dl + r * pitch;
        if (t == ignore_index || (uint64_t)t >= (uint64_t)cols) {
#pragma unroll
            for (int q = 0; q < kDlQuads; ++q) {
                const int c4 = qbase + threadIdx.x + 256 * q;
                if (c4 < quads) *reinterpret_cast<uint2*>(drow + 4 * c4) = uint2{0u, 0u};
            }
            continue;
        }
        const float* x = logits + r * pitch;
        const float lse = ls[r - r0];
#pragma unroll
        for (int q = 0; q < kDlQuads; ++q) {
            const int c4 = qbase + threadIdx.x + 256 * q;
            if (c4 < quads) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(x + 4 * c4);
                f32x4 g;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c = 4 * c4 + j;
                    g[j] = c < cols ? s * (expf(v[j] - lse) - (c == t ? 1.0f : 0.0f)) : 0.f;
                }
                acc[q] += g;
                *reinterpret_cast<uint2*>(drow + 4 * c4) = uint2{pack_bf16x2(g[0], g[1]), pack_bf16x2(g[2], g[3])};
            }
        }
    }
#pragma unroll
    for (int q = 0; q < kDlQuads; ++q) {
        const int c4 = qbase + threadIdx.x + 256 * q;
        if (c4 < quads) *reinterpret_cast<f32x4*>(partial + (int64_t)blockIdx.x * pitch + 4 * c4) = acc[q];
    }
}

// One 1024-thread workgroup, fixed summation order: thread t takes elements t, t + 1024, ... in four interleaved partial sums
// (four loads in flight per thread - a single load per iteration made these latency-bound: 0.2 ms for 128 K elements).
__device__ __forceinline__ float block_reduce16(float v, float* red) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = 0.f;
    if (threadIdx.x == 0)
        for (int w = 0; w < 16; ++w) r += red[w];
    return r;  // valid in thread 0
}

// inv_count[0] = 1 / max(#targets != ignore_index, 1)   (the "mean" reduction's denominator)
__global__ __launch_bounds__(1024) void inv_valid_count_kernel(const int64_t* __restrict__ target, int64_t n, int64_t ignore_index,
                                                               float* __restrict__ inv_count) {
    __shared__ float red[16];
    int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    int64_t i = threadIdx.x;
    for (; i + 3072 < n; i += 4096) {
        const int64_t t0 = target[i], t1 = target[i + 1024], t2 = target[i + 2048], t3 = target[i + 3072];
        c0 += t0 != ignore_index;
        c1 += t1 != ignore_index;
        c2 += t2 != ignore_index;
        c3 += t3 != ignore_index;
    }
    for (; i < n; i += 1024) c0 += target[i] != ignore_index;
    const float c = block_reduce16((float)((c0 + c1) + (c2 + c3)), red);  // counts: exact in fp32 up to 2^24 per workgroup lane sum
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
// four columns per thread, two rows in flight (cols % 4 == 0, cols <= 4096); same fixed per-column order of additions as above
__global__ __launch_bounds__(256) void colsum_rows_vec_kernel(const float* __restrict__ x, float* __restrict__ partial, int64_t rows, int cols) {
    constexpr int MAXQ = 4;
    const int quads = cols / 4;
    f32x4 acc[MAXQ];
#pragma unroll
    for (int k = 0; k < MAXQ; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int64_t step = gridDim.x;
    int64_t r = blockIdx.x;
    for (; r + step < rows; r += 2 * step) {
#pragma unroll
        for (int k = 0; k < MAXQ; ++k) {
            const int qd = threadIdx.x + 256 * k;
            if (qd < quads) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(x + r * cols + 4 * qd);
                const f32x4 b = *reinterpret_cast<const f32x4*>(x + (r + step) * cols + 4 * qd);
                acc[k] += a;
                acc[k] += b;
            }
        }
    }
    for (; r < rows; r += step) {
#pragma unroll
        for (int k = 0; k < MAXQ; ++k) {
            const int qd = threadIdx.x + 256 * k;
            if (qd < quads) acc[k] += *reinterpret_cast<const f32x4*>(x + r * cols + 4 * qd);
        }
    }
#pragma unroll
    for (int k = 0; k < MAXQ; ++k) {
        const int qd = threadIdx.x + 256 * k;
        if (qd < quads) *reinterpret_cast<f32x4*>(partial + (int64_t)blockIdx.x * cols + 4 * qd) = acc[k];
    }
}
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ partial, float* __restrict__ out, int nblk, int C) {
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
__global__ __launch_bounds__(1024) void sum_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t n, float scale,
                                                   int accumulate) {
    __shared__ float red[16];
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int64_t i = threadIdx.x;
    for (; i + 3072 < n; i += 4096) {
        const float a = x[i], b = x[i + 1024], c = x[i + 2048], d = x[i + 3072];
        s0 += a;
        s1 += b;
        s2 += c;
        s3 += d;
    }
    for (; i < n; i += 1024) s0 += x[i];
    const float s = block_reduce16((s0 + s1) + (s2 + s3), red);
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

int cm3p_ce_masked_stats(const float* logits, int64_t rows, int cols, int64_t row_stride, const int64_t* target, int64_t ignore_index,
                         float* loss_rows, float* lse_rows, void* stream) {
    CM3P_REQUIRE(logits && target && loss_rows && lse_rows && rows > 0 && cols > 0 && row_stride >= cols);
    ce_masked_stats_kernel<<<(unsigned)rows, 256, 0, static_cast<hipStream_t>(stream)>>>(logits, cols, row_stride, target, ignore_index,
                                                                                        loss_rows, lse_rows);
    CM3P_LAUNCH_CHECK();
    return CM3P_OK;
}

int cm3p_ce_masked_dlogits_blocks(int64_t rows) { return (int)((rows + kDlRows - 1) / kDlRows); }

int cm3p_ce_masked_dlogits_bf16(const float* logits, int64_t rows, int cols, int64_t row_stride, const int64_t* target, int64_t ignore_index,
                                const float* lse_rows, const float* scale_a, const float* scale_b, void* dlogits_bf16, float* partial,
                                float* colsum, void* stream) {
    CM3P_REQUIRE(logits && target && lse_rows && scale_a && scale_b && dlogits_bf16 && partial && colsum && rows > 0 && cols > 0);
    CM3P_REQUIRE(row_stride >= cols && row_stride % 4 == 0 && cm3p_aligned16(logits) &&
                 cm3p_aligned16(partial) && (reinterpret_cast<uintptr_t>(dlogits_bf16) & 7) == 0);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int nblk = cm3p_ce_masked_dlogits_blocks(rows);
    const dim3 dgrid(nblk, (unsigned)((row_stride + 4 * 256 * kDlQuads - 1) / (4 * 256 * kDlQuads)));
    ce_masked_dlogits_kernel<<<dgrid, 256, 0, s>>>(logits, cols, row_stride, rows, target, ignore_index, lse_rows, scale_a, scale_b,
                                                  (uint16_t*)dlogits_bf16, partial);
    CM3P_LAUNCH_CHECK();
    colsum_final_kernel<<<(int)((row_stride + 63) / 64), 256, 0, s>>>(partial, colsum, nblk, (int)row_stride);
    CM3P_LAUNCH_CHECK();
    return CM3P_OK;
}

int cm3p_inv_valid_count(const int64_t* target, int64_t n, int64_t ignore_index, float* inv_count, void* stream) {
    CM3P_REQUIRE(target && inv_count && n > 0);
    inv_valid_count_kernel<<<1, 1024, 0, static_cast<hipStream_t>(stream)>>>(target, n, ignore_index, inv_count);
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
    if (cols % 4 == 0 && cols <= 4096 && cm3p_aligned16(x) && cm3p_aligned16(partial)) colsum_rows_vec_kernel<<<nblk, 256, 0, s>>>(x, partial, rows, cols);
    else colsum_rows_kernel<<<nblk, 256, 0, s>>>(x, partial, rows, cols);
    CM3P_LAUNCH_CHECK();
    colsum_final_kernel<<<(cols + 63) / 64, 256, 0, s>>>(partial, out, nblk, cols);
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
    sum_kernel<<<1, 1024, 0, static_cast<hipStream_t>(stream)>>>(x, out, n, scale, accumulate);
    CM3P_LAUNCH_CHECK();
    return CM3P_OK;
}

}  // extern "C"
