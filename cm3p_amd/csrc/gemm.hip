// bf16 MFMA GEMM for the encoder's bias-free linears and their gradients (gfx950).
//
//   C[m, n] = sum_k A(m, k) * B(n, k)  (+ R[m, n]),   fp32 accumulation in v_mfma_f32_16x16x32_bf16.
//
// Replaces nn.Linear forward/backward at TF:models/modernbert/modeling_modernbert.py:84,87,246,259 (Wi, Wo, Wqkv, Wo):
//   forward  y  = x  W^T : A = x  [M,K]  k-contiguous, B = W [N,K] k-contiguous
//   dgrad    dx = dy W   : A = dy [M,K'] k-contiguous, B = W stored [K'][N'] (contraction-strided)
//   wgrad    dW = dy^T x : A = dy stored [K'][M'],      B = x stored [K'][N']  (both contraction-strided), split-K
//
// Tiling: 128 x 128 x 64 per 256-thread workgroup (4 waves, 64 x 64 per wave, 4x4 MFMA tiles of 16x16).
// LDS: two stages of (A 16 KiB + B 16 KiB); global loads of stage t+1 are issued before the MFMAs of stage t and
// written to the other LDS stage after them (register staging, one barrier per k-step).
//   k-contiguous operand  -> LDS image [128 rows][64 k], 16-byte chunk index XOR (row & 7): ds_read_b128 fragments
//                            are bank-conflict free.
//   k-strided operand     -> LDS image [64 k][128 idx], 32-byte segment index XOR f(k): fragments come from two
//                            ds_read_b64_tr_b16 (hardware transpose), conflict free per 32-lane half.
// The MFMA takes the B fragment as its first operand, so an accumulator lane holds 4 consecutive n of one m:
// C is written with 8-byte (bf16) / 16-byte (fp32) stores and the residual R is read the same way.
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int kStageBytes = BM * BK * 2;  // 16 KiB per operand per stage

// byte offset of 16-byte chunk c (8 k-values) of row r in a k-contiguous tile image
__device__ __forceinline__ int kc_off(int r, int c) { return r * 128 + ((c ^ (r & 7)) << 4); }
// byte offset of column idx (multiple of 4) of k-row k in a k-strided tile image
__device__ __forceinline__ int ks_off(int k, int idx) {
    const int f = (k & 3) | (((k >> 3) & 1) << 2);
    return k * 256 + ((((idx >> 4) ^ f) & 7) << 5) + ((idx & 15) << 1);
}

template <bool KC>
struct TileLoader {
    const uint16_t* ptr[4];  // global address of this thread's 4 chunks at k-offset 0
    bool row_ok[4];          // KC: row in range;  KS: column chunk in range
    int koff[4];             // k index of the chunk inside the tile
    int lds[4];              // byte offset inside the stage image
    int64_t kstep;           // pointer advance per k-tile (elements)

    __device__ __forceinline__ void init(const uint16_t* base, int64_t ld, int64_t idx0, int64_t extent, int64_t kbeg, int tid) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = tid + 256 * i;
            if constexpr (KC) {
                const int r = q >> 3, c = q & 7;
                row_ok[i] = idx0 + r < extent;
                koff[i] = c * 8;
                ptr[i] = base + (idx0 + r) * ld + kbeg + c * 8;
                lds[i] = kc_off(r, c);
            } else {
                const int k = q >> 4, c = q & 15;
                row_ok[i] = idx0 + c * 8 < extent;
                koff[i] = k;
                ptr[i] = base + (kbeg + k) * ld + idx0 + c * 8;
                lds[i] = ks_off(k, c * 8);
            }
        }
        kstep = KC ? BK : BK * ld;
    }
    // k0 = first k of this tile, kend = exclusive end of the contraction range
    __device__ __forceinline__ void load(uint4 (&r)[4], int64_t k0, int64_t kend) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool ok = row_ok[i] && (k0 + koff[i] < kend);
            r[i] = ok ? *reinterpret_cast<const uint4*>(ptr[i]) : uint4{0u, 0u, 0u, 0u};
            ptr[i] += kstep;
        }
    }
    __device__ __forceinline__ void store(char* stage, const uint4 (&r)[4]) const {
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4*>(stage + lds[i]) = r[i];
    }
};

// fragment of 16 rows (idx0 .. idx0+15) x 32 k (kk*32 ..) for v_mfma_f32_16x16x32_bf16:
// lane l holds element (idx0 + (l & 15), kk*32 + 8*(l >> 4) + j), j = 0..7
template <bool KC>
__device__ __forceinline__ bf16x8 load_frag(const char* stage, int idx0, int kk, int lane) {
    if constexpr (KC) {
        const int r = idx0 + (lane & 15);
        return *reinterpret_cast<const bf16x8*>(stage + kc_off(r, kk * 4 + (lane >> 4)));
    } else {
        const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
        const int k = kk * 32 + 8 * g + q;
        const bf16x4 lo = lds_read_tr16(stage + ks_off(k, idx0 + 4 * p));
        const bf16x4 hi = lds_read_tr16(stage + ks_off(k + 4, idx0 + 4 * p));
        return cat_bf16x4(lo, hi);
    }
}

template <bool A_KC, bool B_KC, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(const uint16_t* A, const uint16_t* B,
                                                           void* __restrict__ Cv, const float* R, int64_t M, int64_t N,
                                                           int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int tiles_n,
                                                           int64_t kchunk, int64_t c_split_stride, RopeArgs rope, BatchArgs bt) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    // blockIdx.y = matrix of a strided batch (1 for the encoder's linears; the Muon step batches same-shaped weights)
    A += (int64_t)blockIdx.y * bt.a_stride;
    B += (int64_t)blockIdx.y * bt.b_stride;

    // XCD-aware tile order: blocks b and b+8 share an XCD (round-robin dispatch), so give each XCD a contiguous run of
    // tiles; consecutive tiles share the activation row-panel (tn fastest) and hit that XCD's L2.  Speed only.
    const int nwg = gridDim.x;
    const int bid = blockIdx.x;
    const int q8 = nwg / 8, r8 = nwg % 8, xcd = bid % 8;
    const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + bid / 8;
    const int tm = swz / tiles_n, tn = swz % tiles_n;
    const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;

    const int64_t kbeg = (int64_t)blockIdx.z * kchunk;
    const int64_t kend = min(K, kbeg + kchunk);
    const int nk = (int)((kend - kbeg + BK - 1) / BK);

    TileLoader<A_KC> la;
    TileLoader<B_KC> lb;
    la.init(A, lda, m0, M, kbeg, tid);
    lb.init(B, ldb, n0, N, kbeg, tid);

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    uint4 ra[4], rb[4];
    if (nk > 0) {
        la.load(ra, kbeg, kend);
        lb.load(rb, kbeg, kend);
        la.store(smem, ra);
        lb.store(smem + kStageBytes, rb);
    }
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const char* sa = smem + (kt & 1) * 2 * kStageBytes;
        const char* sb = sa + kStageBytes;
        const bool more = kt + 1 < nk;
        if (more) {
            la.load(ra, kbeg + (int64_t)(kt + 1) * BK, kend);
            lb.load(rb, kbeg + (int64_t)(kt + 1) * BK, kend);
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = load_frag<A_KC>(sa, wm * 64 + i * 16, kk, lane);
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[j] = load_frag<B_KC>(sb, wn * 64 + j * 16, kk, lane);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        }
        if (more) {
            char* na = smem + ((kt + 1) & 1) * 2 * kStageBytes;
            la.store(na, ra);
            lb.store(na + kStageBytes, rb);
        }
        __syncthreads();
    }

    // epilogue: lane holds C[m][n .. n+3] with m = m0 + wm*64 + i*16 + (lane & 15), n = n0 + wn*64 + j*16 + 4*(lane >> 4)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t m = m0 + wm * 64 + i * 16 + (lane & 15);
        if (m >= M) continue;
        if constexpr (EPI == CM3P_EPI_BF16_ROPE) {
            // the wave's 64 columns are one head: dims d and d+32 sit in accumulator tiles j and j+2 of the same lane
            if (n0 + wn * 64 < rope.ncols) {
                const int64_t prow = rope.per_batch ? m : m % rope.S;
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    rope_rotate4<false>(acc[i][j], acc[i][j + 2], rope.cos + prow * 32, rope.sin + prow * 32, j * 16 + 4 * (lane >> 4));
                if (n0 + wn * 64 < rope.q_cols) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] *= rope.q_scale;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t n = n0 + wn * 64 + j * 16 + 4 * (lane >> 4);
            if (n >= N) continue;
            f32x4 v = acc[i][j];
            if constexpr (EPI == CM3P_EPI_BF16 || EPI == CM3P_EPI_BF16_ROPE) {
                uint16_t* C = static_cast<uint16_t*>(Cv);
                *reinterpret_cast<uint2*>(C + m * ldc + n) = uint2{pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w)};
            } else if constexpr (EPI == CM3P_EPI_BF16_AXPBY) {
                // C = bf16(alpha * acc + beta * Rb), Rb bf16 with C's layout (the Newton-Schulz polynomial steps)
                uint16_t* C = static_cast<uint16_t*>(Cv) + (int64_t)blockIdx.y * bt.c_stride;
                v *= bt.alpha;
                if (bt.Rb) {
                    const uint2 r = *reinterpret_cast<const uint2*>(bt.Rb + (int64_t)blockIdx.y * bt.r_stride + m * ldc + n);
                    v += bt.beta * f32x4{bf16lo(r.x), bf16hi(r.x), bf16lo(r.y), bf16hi(r.y)};
                }
                *reinterpret_cast<uint2*>(C + m * ldc + n) = uint2{pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w)};
            } else {
                float* C = static_cast<float*>(Cv) + (int64_t)blockIdx.z * c_split_stride;
                if constexpr (EPI == CM3P_EPI_F32_RESID) v += *reinterpret_cast<const f32x4*>(R + m * ldc + n);
                if constexpr (EPI == CM3P_EPI_F32_BIAS) v += *reinterpret_cast<const f32x4*>(R + n);
                *reinterpret_cast<f32x4*>(C + m * ldc + n) = v;
            }
        }
    }
}

// out[i] = sum_z slab[z][i], fixed order (deterministic split-K combine)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, float* __restrict__ out, int64_t n4,
                                                            int splits, int64_t stride4) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        f32x4 s = reinterpret_cast<const f32x4*>(ws)[i];
        for (int z = 1; z < splits; ++z) s += reinterpret_cast<const f32x4*>(ws)[i + z * stride4];
        reinterpret_cast<f32x4*>(out)[i] = s;
    }
}

template <bool A_KC, bool B_KC>
int launch(const void* A, const void* B, void* C, const float* R, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb,
           int64_t ldc, int epi, int splits, int64_t kchunk, int64_t c_split_stride, hipStream_t s, RopeArgs rope = RopeArgs{},
           BatchArgs bt = BatchArgs{}, int batch = 1) {
    const int tiles_m = (int)((M + BM - 1) / BM), tiles_n = (int)((N + BN - 1) / BN);
    const dim3 grid(tiles_m * tiles_n, batch, splits);
    const size_t lds = 4 * kStageBytes;
    const uint16_t* a = static_cast<const uint16_t*>(A);
    const uint16_t* b = static_cast<const uint16_t*>(B);
    switch (epi) {
        case CM3P_EPI_BF16:
            gemm_bf16_kernel<A_KC, B_KC, CM3P_EPI_BF16><<<grid, 256, lds, s>>>(a, b, C, R, M, N, K, lda, ldb, ldc, tiles_n, kchunk, c_split_stride, rope, bt);
            break;
        case CM3P_EPI_F32:
            gemm_bf16_kernel<A_KC, B_KC, CM3P_EPI_F32><<<grid, 256, lds, s>>>(a, b, C, R, M, N, K, lda, ldb, ldc, tiles_n, kchunk, c_split_stride, rope, bt);
            break;
        case CM3P_EPI_F32_RESID:
            gemm_bf16_kernel<A_KC, B_KC, CM3P_EPI_F32_RESID><<<grid, 256, lds, s>>>(a, b, C, R, M, N, K, lda, ldb, ldc, tiles_n, kchunk, c_split_stride, rope, bt);
            break;
        case CM3P_EPI_BF16_ROPE:
            if constexpr (A_KC && B_KC) {
                gemm_bf16_kernel<true, true, CM3P_EPI_BF16_ROPE><<<grid, 256, lds, s>>>(a, b, C, R, M, N, K, lda, ldb, ldc, tiles_n, kchunk, c_split_stride, rope, bt);
                break;
            }
            return CM3P_ERR_INVALID;
        case CM3P_EPI_F32_BIAS:
            if constexpr (A_KC && B_KC) {
                gemm_bf16_kernel<true, true, CM3P_EPI_F32_BIAS><<<grid, 256, lds, s>>>(a, b, C, R, M, N, K, lda, ldb, ldc, tiles_n, kchunk, c_split_stride, rope, bt);
                break;
            }
            return CM3P_ERR_INVALID;
        case CM3P_EPI_BF16_AXPBY:
            gemm_bf16_kernel<A_KC, B_KC, CM3P_EPI_BF16_AXPBY><<<grid, 256, lds, s>>>(a, b, C, R, M, N, K, lda, ldb, ldc, tiles_n, kchunk, c_split_stride, rope, bt);
            break;
        default:
            return CM3P_ERR_INVALID;
    }
    return CM3P_OK;
}

}  // namespace

// gemm256.hip: the 256 x 256 LDS-DMA kernel for the big shapes
int cm3p_gemm256_dispatch(const void* A, const void* B, void* C, const float* R, int64_t M, int64_t N, int64_t K, int64_t lda,
                          int64_t ldb, int64_t ldc, int a_kc, int b_kc, int epi, int splits, int64_t kchunk,
                          int64_t c_split_stride, hipStream_t s, RopeArgs rope);

// gemm8p.hip: the same tile on the half-tile ring ("8-phase" schedule); CM3P_ERR_INVALID = shape / layout not covered
int cm3p_gemm8p_dispatch(const void* A, const void* B, void* C, const float* R, int64_t M, int64_t N, int64_t K, int64_t lda,
                         int64_t ldb, int64_t ldc, int a_kc, int b_kc, int epi, int splits, int64_t kchunk,
                         int64_t c_split_stride, hipStream_t s, RopeArgs rope, BatchArgs bt = BatchArgs{}, int batch = 0);

// Development switch (read per call so that one process can A/B): CM3P_GEMM_IMPL=256 keeps the r01 kernel for every big shape.
static inline bool use_8p() {
    const char* e = getenv("CM3P_GEMM_IMPL");
    return !(e && e[0] == '2');
}

// CM3P_GEMM_IMPL=128: the 128 x 128 kernel (two to three workgroups per CU) for every shape - an A/B partner for shapes whose
// epilogue dominates (tools/gemm_ab.py)
static inline bool use_small_only() {
    const char* e = getenv("CM3P_GEMM_IMPL");
    return e && e[0] == '1';
}

static inline int big_gemm(const void* A, const void* B, void* C, const float* R, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb,
                           int64_t ldc, int a_kc, int b_kc, int epi, int splits, int64_t kchunk, int64_t c_split_stride, hipStream_t s,
                           RopeArgs rope) {
    if (use_8p()) {
        const int rc = cm3p_gemm8p_dispatch(A, B, C, R, M, N, K, lda, ldb, ldc, a_kc, b_kc, epi, splits, kchunk, c_split_stride, s, rope);
        if (rc != CM3P_ERR_INVALID) return rc;
    }
    return cm3p_gemm256_dispatch(A, B, C, R, M, N, K, lda, ldb, ldc, a_kc, b_kc, epi, splits, kchunk, c_split_stride, s, rope);
}

static inline int64_t tiles_of(int64_t M, int64_t N, int t) { return ((M + t - 1) / t) * ((N + t - 1) / t); }

int cm3p_ablation_flags_attention();
int cm3p_ablation_flags_attention_bwd();
int cm3p_ablation_flags_attention_bwd_fused();
int cm3p_ablation_flags_gemm256();
int cm3p_ablation_flags_gemm8p();
int cm3p_ablation_flags_attention_fwd();
#if CM3P_DMA_AUDIT
int cm3p_audit_set_attention_fwd(void*);
int cm3p_audit_set_gemm8p(void*);
int cm3p_audit_set_gemm256(void*);
int cm3p_audit_set_attention(void*);
int cm3p_audit_set_attention_bwd_fused(void*);
#endif

extern "C" {

int cm3p_build_ablation_flags(void) {
    return (cm3p_ablation_flags_attention() != 0) | (cm3p_ablation_flags_attention_bwd() != 0) << 1 | (cm3p_ablation_flags_attention_bwd_fused() != 0) << 2 |
           (cm3p_ablation_flags_gemm256() != 0) << 3 | (cm3p_ablation_flags_gemm8p() != 0) << 4 | (CM3P_DMA_AUDIT != 0) << 5 |
           (cm3p_ablation_flags_attention_fwd() != 0) << 6;
}

int cm3p_debug_set_dma_audit(void* buf) {
#if CM3P_DMA_AUDIT
    int rc = cm3p_audit_set_gemm8p(buf);
    if (rc == CM3P_OK) rc = cm3p_audit_set_gemm256(buf);
    if (rc == CM3P_OK) rc = cm3p_audit_set_attention(buf);
    if (rc == CM3P_OK) rc = cm3p_audit_set_attention_bwd_fused(buf);
    if (rc == CM3P_OK) rc = cm3p_audit_set_attention_fwd(buf);
    return rc;
#else
    (void)buf;
    return CM3P_ERR_INVALID;  // this library was built without the audit hooks
#endif
}

int cm3p_gemm_wgrad_splits(int64_t M, int64_t N, int64_t K) {
    // dW = dy^T x: few output tiles, contraction over all tokens.  Aim at ~2 workgroups per CU.
    const int64_t t256 = tiles_of(M, N, 256);
    if (K % 64 == 0 && K >= 8192) {
        int64_t s = 256 / t256;  // one 512-thread workgroup per CU, all resident in a single wave of the grid
        if (s > K / 2048) s = K / 2048;
        if (s < 1) s = 1;
        if (t256 * s >= 200) return (int)s;
    }
    const int64_t t128 = tiles_of(M, N, 128);
    if (t128 >= 512 || K <= 1024) return 1;
    int64_t s = (1024 + t128 - 1) / t128;
    if (s > K / 512) s = K / 512;
    return (int)(s < 1 ? 1 : s);
}

int cm3p_qkv_gemm_rope(const void* x, const void* Wqkv, void* qkv, int64_t M, int64_t N, int64_t K, const float* cos_tab,
                       const float* sin_tab, int S, int per_batch, int rope_cols, float q_scale, void* stream) {
    CM3P_REQUIRE(x && Wqkv && qkv && cos_tab && sin_tab && M > 0 && N > 0 && K > 0 && S > 0);
    CM3P_REQUIRE(cm3p_aligned16(x) && cm3p_aligned16(Wqkv) && cm3p_aligned16(qkv) && cm3p_aligned16(cos_tab) && cm3p_aligned16(sin_tab));
    CM3P_REQUIRE(K % 8 == 0 && N % 64 == 0 && rope_cols % 64 == 0 && rope_cols >= 0 && rope_cols <= N);
    CM3P_REQUIRE(per_batch || M % S == 0);
    hipStream_t s = static_cast<hipStream_t>(stream);
    CM3P_REQUIRE(q_scale > 0.f && rope_cols % 2 == 0);
    const RopeArgs rope{cos_tab, sin_tab, S, per_batch, rope_cols, rope_cols / 2, q_scale};
    // (the 256 x 256 kernel decides per tile whether its columns are rotated: the rotated range must end on a tile boundary)
    const bool big = (K % 64 == 0) && (rope_cols % 256 == 0) && M < (int64_t(1) << 31) && tiles_of(M, N, 256) >= 200 && !use_small_only();
    int rc;
    if (big) rc = big_gemm(x, Wqkv, qkv, nullptr, M, N, K, K, K, N, 1, 1, CM3P_EPI_BF16_ROPE, 1, K, 0, s, rope);
    else rc = launch<true, true>(x, Wqkv, qkv, nullptr, M, N, K, K, K, N, CM3P_EPI_BF16_ROPE, 1, K, 0, s, rope);
    if (rc != CM3P_OK) return rc;
    CM3P_LAUNCH_CHECK();
    return CM3P_OK;
}

int cm3p_gemm_bf16(const void* A, const void* B, void* C, const float* R, int64_t M, int64_t N, int64_t K, int64_t lda,
                   int64_t ldb, int64_t ldc, int a_kc, int b_kc, int epilogue, int split_k, float* workspace, void* stream) {
    CM3P_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0);
    CM3P_REQUIRE((epilogue >= CM3P_EPI_BF16 && epilogue <= CM3P_EPI_F32_RESID) || (epilogue == CM3P_EPI_F32_BIAS && a_kc && b_kc));
    CM3P_REQUIRE(cm3p_aligned16(A) && cm3p_aligned16(B) && cm3p_aligned16(C));
    CM3P_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && ldc % 4 == 0 && N % 4 == 0);
    CM3P_REQUIRE(a_kc ? (K % 8 == 0 && lda >= K) : (M % 8 == 0 && lda >= M));
    CM3P_REQUIRE(b_kc ? (K % 8 == 0 && ldb >= K) : (N % 8 == 0 && ldb >= N));
    CM3P_REQUIRE(ldc >= N && (epilogue != CM3P_EPI_BF16 || ldc % 8 == 0 || ldc % 4 == 0));
    CM3P_REQUIRE((epilogue != CM3P_EPI_F32_RESID && epilogue != CM3P_EPI_F32_BIAS) || (R && cm3p_aligned16(R)));
    CM3P_REQUIRE(split_k >= 1 && (split_k == 1 || (epilogue == CM3P_EPI_F32 && workspace && cm3p_aligned16(workspace) && ldc == N)));
    hipStream_t s = static_cast<hipStream_t>(stream);
    int64_t kchunk = K;
    void* out = C;
    int64_t split_stride = 0;
    if (split_k > 1) {
        // k-split length: a multiple of 128 where K allows it, so that every work item has an even number of 64-deep k-tiles
        // (gemm8p.hip's REBAL instances need that) - the last split absorbs the remainder, itself a multiple of 128 then
        const int64_t q = (K % 128 == 0) ? 128 : BK;
        kchunk = ((K + split_k - 1) / split_k + q - 1) / q * q;
        split_k = (int)((K + kchunk - 1) / kchunk);
    }
    if (split_k > 1) {
        out = workspace;
        split_stride = M * N;
    }
    int rc;
    const bool big = (K % 64 == 0) && (kchunk % 64 == 0) && tiles_of(M, N, 256) * split_k >= 200 && !use_small_only();
    if (big) rc = big_gemm(A, B, out, R, M, N, K, lda, ldb, ldc, a_kc, b_kc, epilogue, split_k, kchunk, split_stride, s, RopeArgs{});
    else if (a_kc && b_kc) rc = launch<true, true>(A, B, out, R, M, N, K, lda, ldb, ldc, epilogue, split_k, kchunk, split_stride, s);
    else if (a_kc) rc = launch<true, false>(A, B, out, R, M, N, K, lda, ldb, ldc, epilogue, split_k, kchunk, split_stride, s);
    else if (b_kc) rc = launch<false, true>(A, B, out, R, M, N, K, lda, ldb, ldc, epilogue, split_k, kchunk, split_stride, s);
    else rc = launch<false, false>(A, B, out, R, M, N, K, lda, ldb, ldc, epilogue, split_k, kchunk, split_stride, s);
    if (rc != CM3P_OK) return rc;
    CM3P_LAUNCH_CHECK();
    if (split_k > 1) {
        const int64_t n4 = M * N / 4;
        int64_t blocks = (n4 + 255) / 256;
        if (blocks > 2048) blocks = 2048;
        splitk_reduce_kernel<<<(int)blocks, 256, 0, s>>>(workspace, static_cast<float*>(C), n4, split_k, n4);
        CM3P_LAUNCH_CHECK();
    }
    return CM3P_OK;
}

int cm3p_gemm_geglu(const void* x, const void* w_interleaved, void* a, int64_t T, int64_t I, int64_t K, void* stream) {
    CM3P_REQUIRE(x && w_interleaved && a && T > 0 && I > 0 && K > 0);
    CM3P_REQUIRE(cm3p_aligned16(x) && cm3p_aligned16(w_interleaved) && cm3p_aligned16(a));
    // the half-tile-ring kernel's shapes only (the caller keeps the two-kernel path for everything else)
    CM3P_REQUIRE(K % 64 == 0 && I % 32 == 0 && T % 8 == 0 && tiles_of(T, 2 * I, 256) >= 200);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int rc = cm3p_gemm8p_dispatch(x, w_interleaved, a, nullptr, T, 2 * I, K, K, K, I, 1, 1, CM3P_EPI_BF16_GEGLU, 1, K, 0, s, RopeArgs{});
    if (rc != CM3P_OK) return rc;
    CM3P_LAUNCH_CHECK();
    return CM3P_OK;
}

int cm3p_gemm_bf16_batched(const void* A, const void* B, void* C, const void* R, int batch, int64_t M, int64_t N, int64_t K,
                           int64_t lda, int64_t ldb, int64_t ldc, int64_t stride_a, int64_t stride_b, int64_t stride_c,
                           int64_t stride_r, int a_kc, int b_kc, float alpha, float beta, void* stream) {
    CM3P_REQUIRE(A && B && C && batch > 0 && batch <= 65535 && M > 0 && N > 0 && K > 0);
    CM3P_REQUIRE(cm3p_aligned16(A) && cm3p_aligned16(B) && cm3p_aligned16(C) && (!R || cm3p_aligned16(R)));
    CM3P_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && ldc % 8 == 0 && N % 8 == 0 && ldc >= N);
    CM3P_REQUIRE(stride_a % 8 == 0 && stride_b % 8 == 0 && stride_c % 8 == 0 && stride_r % 8 == 0);
    CM3P_REQUIRE(a_kc ? (K % 8 == 0 && lda >= K) : (M % 8 == 0 && lda >= M));
    CM3P_REQUIRE(b_kc ? (K % 8 == 0 && ldb >= K) : (N % 8 == 0 && ldb >= N));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const BatchArgs bt{stride_a, stride_b, stride_c, stride_r, static_cast<const uint16_t*>(R), alpha, beta};
    int rc;
    // enough 256 x 256 tiles to occupy the chip (the 768- to 2304-wide weight groups of the Muon step: 198 to 1188): the half-tile
    // ring kernel, one work item per (matrix, tile); everything else (and what it does not cover) on the 128 x 128 kernel
    const int64_t tiles256 = ((M + 255) / 256) * ((N + 255) / 256) * batch;
    const bool filled = M * N * batch * 10 >= tiles256 * 65536 * 7;  // (edge tiles compute their padding: [520 x 264] x 30 ran at 0.59 of the small kernel)
    if (tiles256 >= 128 && filled && K % 64 == 0 && M % 8 == 0 && (a_kc || !b_kc) && use_8p()) {
        rc = cm3p_gemm8p_dispatch(A, B, C, nullptr, M, N, K, lda, ldb, ldc, a_kc, b_kc, CM3P_EPI_BF16_AXPBY, 1, K, 0, s, RopeArgs{}, bt, batch);
        if (rc == CM3P_OK) {
            CM3P_LAUNCH_CHECK();
            return CM3P_OK;
        }
        if (rc != CM3P_ERR_INVALID) return rc;
    }
    if (a_kc && b_kc) rc = launch<true, true>(A, B, C, nullptr, M, N, K, lda, ldb, ldc, CM3P_EPI_BF16_AXPBY, 1, K, 0, s, RopeArgs{}, bt, batch);
    else if (a_kc) rc = launch<true, false>(A, B, C, nullptr, M, N, K, lda, ldb, ldc, CM3P_EPI_BF16_AXPBY, 1, K, 0, s, RopeArgs{}, bt, batch);
    else if (b_kc) rc = launch<false, true>(A, B, C, nullptr, M, N, K, lda, ldb, ldc, CM3P_EPI_BF16_AXPBY, 1, K, 0, s, RopeArgs{}, bt, batch);
    else rc = launch<false, false>(A, B, C, nullptr, M, N, K, lda, ldb, ldc, CM3P_EPI_BF16_AXPBY, 1, K, 0, s, RopeArgs{}, bt, batch);
    if (rc != CM3P_OK) return rc;
    CM3P_LAUNCH_CHECK();
    return CM3P_OK;
}

}  // extern "C"
