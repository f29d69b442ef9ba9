// Optimizer step of the reference's Muon (ref:utils/muon_utils.py:138-203): the element-wise stages around the
// Newton-Schulz GEMMs (which are cm3p_gemm_bf16_batched in gemm.hip) and the AdamW branch for the other parameters.
//
// Same-shaped 2-D weights are processed as one strided batch: the parameters themselves stay where torch allocated them
// (device tables of pointers), the bf16 Newton-Schulz iterate lives in one workspace [n_mat, rows_p, cols_p] whose
// extents are rounded up to 8 (zero padding is invariant under the iteration, so odd shapes need no special GEMM).
// Everything here is HBM-bound streaming; reductions are two-stage with a fixed order, so every rank of a data-parallel
// job computes bit-identical updates from bit-identical (all-reduced) gradients.
#include "common.h"

namespace {

constexpr int kThreads = 256;

__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) red[wid] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < kThreads / 64; ++w) t += red[w];  // fixed order
    return t;
}

// buf = momentum * buf + g;  u = nesterov ? g + momentum * buf : g (sic, :163-164);  X = bf16(u);  partial[mat][blk] = sum float(X)^2
// (ref:utils/muon_utils.py:160-164 and the first two lines of zeropower_via_newtonschulz5, :46-47)
template <bool VEC>
__global__ __launch_bounds__(kThreads) void muon_momentum_kernel(const int64_t* __restrict__ g_ptrs, const int64_t* __restrict__ buf_ptrs,
                                                                 uint16_t* __restrict__ X, float* __restrict__ partials, int rows,
                                                                 int cols, int ldx, int64_t x_stride, float momentum, int nesterov) {
    __shared__ float red[kThreads / 64];
    const int mat = blockIdx.y;
    const float* g = reinterpret_cast<const float*>(g_ptrs[mat]);
    float* buf = reinterpret_cast<float*>(buf_ptrs[mat]);
    uint16_t* x = X + (int64_t)mat * x_stride;
    const int64_t numel = (int64_t)rows * cols;
    float ss = 0.f;
    if constexpr (VEC) {
        const int64_t n4 = numel >> 2;
        const int64_t per = (n4 + gridDim.x - 1) / gridDim.x;
        const int64_t beg = per * blockIdx.x, end = min(n4, beg + per);
        for (int64_t i = beg + threadIdx.x; i < end; i += kThreads) {
            const f32x4 gv = reinterpret_cast<const f32x4*>(g)[i];
            f32x4 bv = reinterpret_cast<const f32x4*>(buf)[i];
            f32x4 u;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bv[j] = __fadd_rn(__fmul_rn(bv[j], momentum), gv[j]);  // buf.mul_(momentum).add_(g): two roundings
                u[j] = nesterov ? fmaf(momentum, bv[j], gv[j]) : gv[j];  // g.add(buf, alpha=momentum)
            }
            reinterpret_cast<f32x4*>(buf)[i] = bv;
            const uint32_t w0 = pack_bf16x2(u.x, u.y), w1 = pack_bf16x2(u.z, u.w);
            const float a = bf16lo(w0), b = bf16hi(w0), c = bf16lo(w1), d = bf16hi(w1);
            ss += a * a + b * b + c * c + d * d;
            const int64_t e = i << 2;
            const int64_t r = e / cols, cc = e - r * cols;  // cols % 4 == 0: the four elements share a row
            *reinterpret_cast<uint2*>(x + r * ldx + cc) = uint2{w0, w1};
        }
    } else {
        const int64_t per = (numel + gridDim.x - 1) / gridDim.x;
        const int64_t beg = per * blockIdx.x, end = min(numel, beg + per);
        for (int64_t e = beg + threadIdx.x; e < end; e += kThreads) {
            const float gv = g[e];
            const float bv = __fadd_rn(__fmul_rn(buf[e], momentum), gv);
            buf[e] = bv;
            const float u = nesterov ? fmaf(momentum, bv, gv) : gv;
            const uint16_t w = f32_to_bf16_bits(u);
            const float a = bf16_bits_to_f32(w);
            ss += a * a;
            const int64_t r = e / cols, cc = e - r * cols;
            x[r * ldx + cc] = w;
        }
    }
    const float t = block_sum(ss, red);
    if (threadIdx.x == 0) partials[(int64_t)mat * gridDim.x + blockIdx.x] = t;
}

// X /= (bf16(||X||) + eps), all in the reference's bf16 arithmetic (ref:utils/muon_utils.py:47: the norm of a bf16
// tensor is a bf16 scalar, the sum with eps is rounded to bf16 again, the quotient is rounded to bf16).
// Runs over the padded image (padding is zero and stays zero).
__global__ __launch_bounds__(kThreads) void muon_normalize_kernel(uint16_t* __restrict__ X, const float* __restrict__ partials, int nparts,
                                                                  int64_t x_stride, float eps) {
    const int mat = blockIdx.y;
    float total = 0.f;
    for (int i = 0; i < nparts; ++i) total += partials[(int64_t)mat * nparts + i];  // same order in every thread
    const float norm = bf16_bits_to_f32(f32_to_bf16_bits(sqrtf(total)));
    const float denom = bf16_bits_to_f32(f32_to_bf16_bits(norm + eps));
    uint16_t* x = X + (int64_t)mat * x_stride;
    const int64_t n8 = x_stride >> 3;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n8; i += (int64_t)gridDim.x * kThreads) {
        uint4 v = reinterpret_cast<uint4*>(x)[i];
        uint32_t* w = reinterpret_cast<uint32_t*>(&v);
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = pack_bf16x2(bf16lo(w[j]) / denom, bf16hi(w[j]) / denom);
        reinterpret_cast<uint4*>(x)[i] = v;
    }
}

// p += -lr * float(bf16(float(X) * shape_scale))   (ref:utils/muon_utils.py:173-176)
__global__ __launch_bounds__(kThreads) void muon_apply_kernel(const int64_t* __restrict__ p_ptrs, const uint16_t* __restrict__ X, int rows,
                                                              int cols, int ldx, int64_t x_stride, float shape_scale, float neg_lr) {
    const int mat = blockIdx.y;
    float* p = reinterpret_cast<float*>(p_ptrs[mat]);
    const uint16_t* x = X + (int64_t)mat * x_stride;
    const int64_t numel = (int64_t)rows * cols;
    for (int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x; e < numel; e += (int64_t)gridDim.x * kThreads) {
        const int64_t r = e / cols, cc = e - r * cols;
        const float o = bf16_bits_to_f32(f32_to_bf16_bits(bf16_bits_to_f32(x[r * ldx + cc]) * shape_scale));
        p[e] = fmaf(neg_lr, o, p[e]);  // p.add_(g, alpha=-lr)
    }
}

// torch.lerp with a scalar weight (ATen/native/Lerp.h)
__device__ __forceinline__ float lerp_like_torch(float a, float b, float w) {
    const float d = b - a;
    return fabsf(w) < 0.5f ? fmaf(w, d, a) : b - d * (1.f - w);
}

// The reference's "AdamW" branch, one launch for all tensors (ref:utils/muon_utils.py:178-203); blockIdx.y = tensor.
__global__ __launch_bounds__(kThreads) void adamw_multi_kernel(const int64_t* __restrict__ p_ptrs, const int64_t* __restrict__ g_ptrs,
                                                               const int64_t* __restrict__ m1_ptrs, const int64_t* __restrict__ m2_ptrs,
                                                               const int64_t* __restrict__ numels, float w1, float w2, float eps,
                                                               float decay, float step_alpha) {
    const int t = blockIdx.y;
    const int64_t n = numels[t];
    float* p = reinterpret_cast<float*>(p_ptrs[t]);
    const float* g = reinterpret_cast<const float*>(g_ptrs[t]);
    float* m1 = reinterpret_cast<float*>(m1_ptrs[t]);
    float* m2 = reinterpret_cast<float*>(m2_ptrs[t]);
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) {
        const float gv = g[i];
        const float a = lerp_like_torch(m1[i], gv, w1);
        const float b = lerp_like_torch(m2[i], gv * gv, w2);
        m1[i] = a;
        m2[i] = b;
        const float u = a / (eps + sqrtf(b));
        p[i] = fmaf(step_alpha, u, __fmul_rn(p[i], decay));  // p.mul_(decay); p.add_(u, alpha=step_alpha)
    }
}

}  // namespace

extern "C" {

int cm3p_muon_partials(int rows, int cols) {
    const int64_t numel = (int64_t)rows * cols;
    int64_t b = (numel + 16383) / 16384;  // >= 16 Ki elements per block
    if (b > 64) b = 64;
    return (int)(b < 1 ? 1 : b);
}

int cm3p_muon_momentum(const int64_t* g_ptrs, const int64_t* buf_ptrs, void* X, float* partials, int n_mat, int rows, int cols,
                       int ldx, int64_t x_stride, float momentum, int nesterov, int aligned16, void* stream) {
    CM3P_REQUIRE(g_ptrs && buf_ptrs && X && partials && n_mat > 0 && n_mat <= 65535 && rows > 0 && cols > 0);
    CM3P_REQUIRE(ldx >= cols && ldx % 8 == 0 && x_stride % 8 == 0 && x_stride >= (int64_t)rows * ldx && cm3p_aligned16(X));
    const dim3 grid(cm3p_muon_partials(rows, cols), n_mat);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (cols % 4 == 0 && aligned16)  // aligned16: every g / buf pointer in the tables is 16-byte aligned (host knows)
        muon_momentum_kernel<true><<<grid, kThreads, 0, s>>>(g_ptrs, buf_ptrs, static_cast<uint16_t*>(X), partials, rows, cols, ldx, x_stride, momentum, nesterov);
    else
        muon_momentum_kernel<false><<<grid, kThreads, 0, s>>>(g_ptrs, buf_ptrs, static_cast<uint16_t*>(X), partials, rows, cols, ldx, x_stride, momentum, nesterov);
    CM3P_LAUNCH_CHECK();
    return CM3P_OK;
}

int cm3p_muon_normalize(void* X, const float* partials, int n_mat, int rows, int cols, int64_t x_stride, float eps, void* stream) {
    CM3P_REQUIRE(X && partials && n_mat > 0 && n_mat <= 65535 && rows > 0 && cols > 0 && x_stride % 8 == 0 && cm3p_aligned16(X));
    int64_t blocks = (x_stride / 8 + kThreads - 1) / kThreads;
    if (blocks > 128) blocks = 128;
    const dim3 grid((int)blocks, n_mat);
    muon_normalize_kernel<<<grid, kThreads, 0, static_cast<hipStream_t>(stream)>>>(static_cast<uint16_t*>(X), partials,
                                                                                  cm3p_muon_partials(rows, cols), x_stride, eps);
    CM3P_LAUNCH_CHECK();
    return CM3P_OK;
}

int cm3p_muon_apply(const int64_t* p_ptrs, const void* X, int n_mat, int rows, int cols, int ldx, int64_t x_stride, float shape_scale,
                    float neg_lr, void* stream) {
    CM3P_REQUIRE(p_ptrs && X && n_mat > 0 && n_mat <= 65535 && rows > 0 && cols > 0 && ldx >= cols);
    int64_t blocks = ((int64_t)rows * cols + 4 * kThreads - 1) / (4 * kThreads);
    if (blocks > 128) blocks = 128;
    const dim3 grid((int)blocks, n_mat);
    muon_apply_kernel<<<grid, kThreads, 0, static_cast<hipStream_t>(stream)>>>(p_ptrs, static_cast<const uint16_t*>(X), rows, cols, ldx,
                                                                              x_stride, shape_scale, neg_lr);
    CM3P_LAUNCH_CHECK();
    return CM3P_OK;
}

int cm3p_adamw_multi(const int64_t* p_ptrs, const int64_t* g_ptrs, const int64_t* m1_ptrs, const int64_t* m2_ptrs, const int64_t* numels,
                     int n_tensors, int64_t max_numel, float w1, float w2, float eps, float decay, float step_alpha, void* stream) {
    CM3P_REQUIRE(p_ptrs && g_ptrs && m1_ptrs && m2_ptrs && numels && n_tensors > 0 && n_tensors <= 65535 && max_numel > 0);
    int64_t blocks = (max_numel + 4 * kThreads - 1) / (4 * kThreads);
    if (blocks > 64) blocks = 64;
    const dim3 grid((int)blocks, n_tensors);
    adamw_multi_kernel<<<grid, kThreads, 0, static_cast<hipStream_t>(stream)>>>(p_ptrs, g_ptrs, m1_ptrs, m2_ptrs, numels, w1, w2, eps, decay,
                                                                               step_alpha);
    CM3P_LAUNCH_CHECK();
    return CM3P_OK;
}

}  // extern "C"
