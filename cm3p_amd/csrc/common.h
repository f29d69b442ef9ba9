// Shared device helpers for the CM3P gfx950 kernels.  CDNA4 only: 64-lane waves, MFMA, 160 KiB LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <initializer_list>
#include <mutex>

#include "../../include/cm3p_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) float f32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define CM3P_WAVE 64
#define CM3P_EPI_BF16_ROPE 3   // internal: bf16 output with rotary embedding applied to the leading columns
#define CM3P_EPI_BF16_AXPBY 4  // internal: bf16 output = alpha * acc + beta * Rb (cm3p_gemm_bf16_batched)
// 5 = CM3P_EPI_F32_BIAS (public, include/cm3p_hip.h)
#define CM3P_EPI_BF16_GEGLU 6  // internal (cm3p_gemm_geglu): every 64 output columns are [32 h | 32 g]; stores gelu_erf(h) * g, 32 columns

// Strided-batch offsets (elements) and the AXPBY epilogue operands of the 128 x 128 GEMM kernel.
struct BatchArgs {
    int64_t a_stride = 0, b_stride = 0, c_stride = 0, r_stride = 0;
    const uint16_t* Rb = nullptr;
    float alpha = 1.f, beta = 0.f;
};

// Cache policy of write-once / read-once global traffic.  A plain store leaves its line in the XCD's 4 MiB L2, where it pushes out
// what the workgroups of that XCD re-read (GEMM operand panels, the K / V or Q / dO tiles of an attention head); a non-temporal
// store ("nt") asks for the line to be the first to go.  Same bytes, same results; which sites pay was measured (DESIGN section 4,
// r04 "output stores").  Bit mask so that variants can be built for an A/B (tools/ubench/nt_variants.sh):
//   1 gemm8p bf16 epilogues   2 fused attention backward: dQ partial slabs   4 attention outputs written through store_rows32
//   8 dQ slab reduce (slab loads and dq stores)   16 LayerNorm / GeGLU outputs   32 gemm8p fp32 epilogues
//   64 gemm8p residual loads (read once)   128 LayerNorm / GeGLU input rows (read once)
#ifndef CM3P_NT
#define CM3P_NT 7
#endif
typedef __attribute__((ext_vector_type(4))) uint32_t cm3p_u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t cm3p_u32x2;
template <bool NT>
__device__ __forceinline__ void gstore16(void* p, cm3p_u32x4 v) {
    if constexpr (NT) __builtin_nontemporal_store(v, static_cast<cm3p_u32x4*>(p));
    else *static_cast<cm3p_u32x4*>(p) = v;
}
template <bool NT>
__device__ __forceinline__ void gstore16(void* p, uint4 v) { gstore16<NT>(p, cm3p_u32x4{v.x, v.y, v.z, v.w}); }
template <bool NT>
__device__ __forceinline__ void gstore16f(void* p, __attribute__((ext_vector_type(4))) float v) {
    gstore16<NT>(p, __builtin_bit_cast(cm3p_u32x4, v));
}
template <bool NT>
__device__ __forceinline__ void gstore8(void* p, uint2 v) {
    if constexpr (NT) __builtin_nontemporal_store(cm3p_u32x2{v.x, v.y}, static_cast<cm3p_u32x2*>(p));
    else *static_cast<uint2*>(p) = v;
}
template <bool NT>
__device__ __forceinline__ uint4 gload16(const void* p) {
    if constexpr (NT) {
        const cm3p_u32x4 v = __builtin_nontemporal_load(static_cast<const cm3p_u32x4*>(p));
        return uint4{v[0], v[1], v[2], v[3]};
    } else {
        return *static_cast<const uint4*>(p);
    }
}

template <bool NT>
__device__ __forceinline__ uint2 gload8(const void* p) {
    if constexpr (NT) {
        const cm3p_u32x2 v = __builtin_nontemporal_load(static_cast<const cm3p_u32x2*>(p));
        return uint2{v[0], v[1]};
    } else {
        return *static_cast<const uint2*>(p);
    }
}
template <bool NT>
__device__ __forceinline__ __attribute__((ext_vector_type(4))) float gload16f(const void* p) {
    typedef __attribute__((ext_vector_type(4))) float f4;
    if constexpr (NT) return __builtin_nontemporal_load(static_cast<const f4*>(p));
    else return *static_cast<const f4*>(p);
}

// ---- bounds audit of the LDS-DMA streams (debug builds only: -DCM3P_DMA_AUDIT=1, libcm3p_hip_audit.so) -------------------------
// An LDS-DMA load has no destination register and no bounds check, and a stray READ changes nothing a parity test can see unless it
// crosses into an unmapped page (the r03 fault of gemm8p's staging stream was exactly that).  In an audit build every staging
// helper reports the lowest and highest global byte address its wave is about to read, per operand id, to a caller-owned buffer
// (cm3p_debug_set_dma_audit); tests/test_dma_audit_gpu.py runs the edge shapes and asserts that every address lies inside the
// tensor the caller passed.  The shipped library is built without it (cm3p_build_ablation_flags() bit 5 says which one this is).
//   ids: 0 GEMM A operand, 1 GEMM B operand, 2 / 3 first / second tile matrix of an attention ring (K, V or Q, dO),
//        4 / 5 per-row statistics or mask rows (lse or mask; delta)
#ifndef CM3P_DMA_AUDIT
#define CM3P_DMA_AUDIT 0
#endif
enum { CM3P_AUD_A = 0, CM3P_AUD_B = 1, CM3P_AUD_T0 = 2, CM3P_AUD_T1 = 3, CM3P_AUD_S0 = 4, CM3P_AUD_S1 = 5, CM3P_AUD_SLOTS = 8 };
#if CM3P_DMA_AUDIT
static __device__ unsigned long long* cm3p_audit_buf = nullptr;  // [CM3P_AUD_SLOTS][2]: lowest first byte, highest last byte (one copy per object file)
static inline int cm3p_audit_set_local(void* buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(cm3p_audit_buf), &buf, sizeof(buf)) == hipSuccess ? CM3P_OK : CM3P_ERR_LAUNCH;
}
// every lane passes the address of the first byte it reads and how many bytes (all lanes of the wave are active at the call sites)
__device__ __forceinline__ void cm3p_audit(int id, const void* first_byte, int bytes) {
    unsigned long long* buf = cm3p_audit_buf;
    if (!buf) return;
    unsigned long long lo = (unsigned long long)(uintptr_t)first_byte, hi = lo + (unsigned long long)bytes - 1ull;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long l2 = __shfl_xor(lo, o, 64), h2 = __shfl_xor(hi, o, 64);
        lo = l2 < lo ? l2 : lo;
        hi = h2 > hi ? h2 : hi;
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(buf + 2 * id, lo);
        atomicMax(buf + 2 * id + 1, hi);
    }
}
#define CM3P_AUDIT(id, addr, bytes) cm3p_audit((id), (addr), (bytes))
#else
#define CM3P_AUDIT(id, addr, bytes) ((void)0)
#endif

// Every extern "C" entry point ends with this: kernels never throw, launch errors become a return code.
#define CM3P_LAUNCH_CHECK()                                         \
    do {                                                            \
        hipError_t e__ = hipGetLastError();                         \
        if (e__ != hipSuccess) return CM3P_ERR_LAUNCH;              \
    } while (0)

#define CM3P_REQUIRE(cond)                   \
    do {                                     \
        if (!(cond)) return CM3P_ERR_INVALID; \
    } while (0)

// A kernel's MaxDynamicSharedMemorySize attribute and the CU count belong to a DEVICE, and one process may drive several (a model moved
// to cuda:1, DataParallel): once-per-process statics would launch on the second device without the attribute and with the first
// device's grid (r03 advisor).  (Not thread-safe beyond what a benign double initialisation is.)
constexpr int kCm3pMaxDevices = 64;
static inline int cm3p_current_device() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= kCm3pMaxDevices) d = 0;
    return d;
}
static inline int cm3p_num_cu() {
    static int cu[kCm3pMaxDevices];
    const int d = cm3p_current_device();
    if (cu[d] == 0) {
        hipDeviceProp_t prop;
        cu[d] = (hipGetDeviceProperties(&prop, d) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    return cu[d];
}
// `static Cm3pDevOnce once; rc = once.run([&] { ...; return CM3P_OK; });`: the body runs until it has SUCCEEDED once per device.  A device
// is marked only after its body returned CM3P_OK (r04 advisor: the first form marked it before the body ran, so a failed
// hipFuncSetAttribute was never retried and every later launch went out without the attribute), under a lock (two host threads on one
// device cannot both skip while the attribute is still being applied); devices beyond the table run the body every time (the bodies are
// idempotent attribute calls) instead of sharing slot 0.
struct Cm3pDevOnce {
    std::atomic<unsigned char> done[kCm3pMaxDevices] = {};
    std::mutex mu;
    template <class F>
    int run(F&& body) {
        int d = 0;
        const bool tracked = hipGetDevice(&d) == hipSuccess && d >= 0 && d < kCm3pMaxDevices;
        if (tracked && done[d].load(std::memory_order_acquire)) return CM3P_OK;
        std::lock_guard<std::mutex> g(mu);
        if (tracked && done[d].load(std::memory_order_relaxed)) return CM3P_OK;
        const int rc = body();
        if (rc == CM3P_OK && tracked) done[d].store(1, std::memory_order_release);
        return rc;
    }
};

// dynamic LDS above the 64-KiB default for every kernel in the list
static inline int cm3p_set_max_lds(std::initializer_list<const void*> kernels, int bytes) {
    for (const void* k : kernels)
        if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return CM3P_ERR_LAUNCH;
    return CM3P_OK;
}

static inline bool cm3p_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

__device__ __forceinline__ float bf16_bits_to_f32(uint16_t b) { return __builtin_bit_cast(float, (uint32_t)b << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16_bits(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }

// ---- exact-erf GELU (TF ACT2FN["gelu"], approximate="none"), two values per call so that the polynomial runs on v_pk_fma_f32 ----------
// Phi(x) = 1/2 erfc(-x / sqrt 2) with erfc(z) = t exp(-z^2 + P(t)), t = 1 / (1 + z / 2), z >= 0 (the Chebyshev fit of Numerical Recipes'
// erfcc: fractional error < 1.2e-7 in exact arithmetic, 3.3e-6 as evaluated here in fp32 over EVERY bf16 input, relative also in the negative
// tail where 1 + erf(x / sqrt 2) cancels to zero).  16 VALU instructions per value (three of them v_rcp / v_exp) against 45 with the device
// library's erff + expf; of the 65280 finite bf16 inputs, gelu rounds to a different bf16 than the exact value for 64 (197 with
// 0.5 x (1 + erff)).  One definition for every GELU of the library (GeGLU forward / backward, the fused Wi + GeGLU epilogue, the audio
// encoder's bias + GELU): same source, same bits.  Q(t) = log2(e) P(t).
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ f32x2 gelu_q_poly(f32x2 t) {
    f32x2 q = t * 0.24651730060577393f + (-1.1861149072647095f);
    q = q * t + 2.147474527359009f;
    q = q * t + (-1.6377531290054321f);
    q = q * t + 0.40232157707214355f;
    q = q * t + (-0.2687568664550781f);
    q = q * t + 0.1396300494670868f;
    q = q * t + 0.5397006273269653f;
    q = q * t + 1.4427292346954346f;
    q = q * t + (-1.8257482051849365f);
    return q;
}
__device__ __forceinline__ f32x2 gelu_t(f32x2 x) {
    const f32x2 ax = {__builtin_fabsf(x.x), __builtin_fabsf(x.y)};
    const f32x2 w = ax * 0.35355339059327373f + 1.0f;  // 1 + z / 2, z = |x| / sqrt 2
    return f32x2{__builtin_amdgcn_rcpf(w.x), __builtin_amdgcn_rcpf(w.y)};
}
// Phi(x) and the standard normal density (the backward needs both: gelu'(x) = Phi(x) + x phi(x))
__device__ __forceinline__ void gelu_cdf_pdf2(f32x2 x, f32x2& cdf, f32x2& pdf) {
    const f32x2 t = gelu_t(x);
    const f32x2 q = gelu_q_poly(t);
    const f32x2 u = x * x * (-0.72134752044448170f);  // -x^2 / 2 in log2 units
    const f32x2 eu = {__builtin_amdgcn_exp2f(u.x), __builtin_amdgcn_exp2f(u.y)};
    const f32x2 eq = {__builtin_amdgcn_exp2f(q.x), __builtin_amdgcn_exp2f(q.y)};
    const f32x2 h = t * eu * eq * 0.5f;  // erfc(|x| / sqrt 2) / 2
    cdf.x = x.x < 0.f ? h.x : 1.0f - h.x;
    cdf.y = x.y < 0.f ? h.y : 1.0f - h.y;
    pdf = eu * 0.39894228040143267794f;
}
// gelu(x) = x Phi(x); the forward needs one exponential per value
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 x) {
    const f32x2 t = gelu_t(x);
    const f32x2 e = x * x * (-0.72134752044448170f) + gelu_q_poly(t);
    const f32x2 h = t * f32x2{__builtin_amdgcn_exp2f(e.x), __builtin_amdgcn_exp2f(e.y)} * 0.5f;
    const f32x2 cdf = {x.x < 0.f ? h.x : 1.0f - h.x, x.y < 0.f ? h.y : 1.0f - h.y};
    return x * cdf;
}
__device__ __forceinline__ f32x2 gelu_erf_grad2(f32x2 x) {
    f32x2 cdf, pdf;
    gelu_cdf_pdf2(x, cdf, pdf);
    return x * pdf + cdf;
}

// packs two floats into one dword of two bf16 (v_cvt_pk_bf16_f32, round-to-nearest-even, NaN preserving)
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    bf16x2 v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float bf16lo(uint32_t w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float bf16hi(uint32_t w) { return __builtin_bit_cast(float, w & 0xffff0000u); }

// Rotary tables for epilogue fusion: cos/sin are [n_pos, 32] fp32; a token row m uses table row (per_batch ? m : m % S);
// only the first `ncols` output columns (the q and k thirds of a packed qkv row) are rotated.
struct RopeArgs {
    const float* cos;
    const float* sin;
    int S;
    int per_batch;
    int ncols;
    // the first q_cols columns (the q third) are additionally multiplied by q_scale, in fp32, before the single bf16 rounding of
    // the rotated value: the attention kernels then get softmax-ready scores (scale * log2 e folded in) without a second rounding
    int q_cols = 0;
    float q_scale = 1.f;
};

// a = head dims [d, d+3], b = head dims [d+32, d+35] of one token (d < 32): rotate_half convention
// (TF:models/modernbert/modeling_modernbert.py:188-219).  INVERSE applies the transpose (backward pass).
template <bool INVERSE>
__device__ __forceinline__ void rope_rotate4(f32x4& a, f32x4& b, const float* cos_row, const float* sin_row, int d) {
    const f32x4 cs = *reinterpret_cast<const f32x4*>(cos_row + d);
    const f32x4 sn = *reinterpret_cast<const f32x4*>(sin_row + d);
    const f32x4 a0 = a, b0 = b;
    if constexpr (!INVERSE) {
        a = a0 * cs - b0 * sn;
        b = b0 * cs + a0 * sn;
    } else {
        a = a0 * cs + b0 * sn;
        b = b0 * cs - a0 * sn;
    }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ds_read_b64_tr_b16: per 16-lane group, reads a 4-row x 16-column block of 16-bit elements and returns it
// column-major (lane i of the group gets column i, rows 0..3).  `p` is this lane's own address: lane 4q+r of the
// group supplies row q, columns 4r..4r+3.  EXEC must be all ones.
__device__ __forceinline__ bf16x4 lds_read_tr16(const void* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(p));
}

__device__ __forceinline__ bf16x8 cat_bf16x4(bf16x4 a, bf16x4 b) {
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}
