// Stand-alone harness for the big-shape GEMM kernels (no torch): dlopens libcm3p_hip.so (CM3P_HIP_LIB or the in-tree build), runs cm3p_gemm_bf16 with CM3P_GEMM_IMPL=256
// and the default (gemm8p.hip) on the same operands, compares every element, checks guard zones around the output, then times both
// in interleaved rounds.  Every buffer sits in the middle of one large allocation with 64 MiB of owned memory on both sides, so a
// near out-of-bounds access corrupts a guard (reported) instead of faulting the GPU.
//   hipcc --offload-arch=gfx950 -O2 -o tools/ubench/gemm_harness tools/ubench/gemm_harness.cpp -ldl
//   tools/ubench/gemm_harness [check] [time]        (run from the repository root)
//   tools/ubench/gemm_harness batched               (the strided-batch a x + b y GEMMs of the Muon step: 128-tile kernel vs gemm8p.hip)
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#define CM3P_EPI_BF16 0
#define CM3P_EPI_F32 1
#define CM3P_EPI_F32_RESID 2
typedef int (*gemm_fn)(const void*, const void*, void*, const float*, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int, int, int, int,
                       float*, void*);
static gemm_fn cm3p_gemm_bf16;
typedef int (*bgemm_fn)(const void*, const void*, void*, const void*, int, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t,
                        int64_t, int64_t, int, int, float, float, void*);
static bgemm_fn cm3p_gemm_bf16_batched;

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e_ = (x);                                                   \
        if (e_ != hipSuccess) {                                                \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(2);                                                           \
        }                                                                      \
    } while (0)

static const size_t GUARD = 64u << 20;

__global__ void fill_bf16(uint16_t* p, size_t n, uint32_t seed, float scale) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)i * 2654435761u ^ seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        const float v = ((h & 0xffffff) / 8388608.0f - 1.0f) * scale;
        p[i] = __builtin_bit_cast(uint16_t, (__bf16)v);
    }
}
__global__ void fill_f32(float* p, size_t n, uint32_t seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)i * 2654435761u ^ seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = (h & 0xffffff) / 8388608.0f - 1.0f;
    }
}
__global__ void fill_u32(uint32_t* p, size_t n, uint32_t v) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void count_neq_u32(const uint32_t* p, size_t n, uint32_t v, unsigned long long* out) {
    unsigned long long c = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) c += p[i] != v;
    if (c) atomicAdd(out, c);
}
__global__ void count_diff_u32(const uint32_t* a, const uint32_t* b, size_t n, unsigned long long* out) {
    unsigned long long c = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) c += a[i] != b[i];
    if (c) atomicAdd(out, c);
}

struct Guarded {
    char* base = nullptr;
    size_t bytes = 0;
    char* p() const { return base + GUARD; }
    void alloc(size_t n) {
        bytes = (n + 255) & ~size_t(255);
        CK(hipMalloc(&base, bytes + 2 * GUARD));
        fill_u32<<<1024, 256>>>((uint32_t*)base, (bytes + 2 * GUARD) / 4, 0xDEADBEEFu);
    }
    unsigned long long guards_touched(unsigned long long* d_cnt) const {
        CK(hipMemset(d_cnt, 0, 8));
        count_neq_u32<<<1024, 256>>>((const uint32_t*)base, GUARD / 4, 0xDEADBEEFu, d_cnt);
        count_neq_u32<<<1024, 256>>>((const uint32_t*)(base + GUARD + bytes), GUARD / 4, 0xDEADBEEFu, d_cnt);
        unsigned long long h;
        CK(hipMemcpy(&h, d_cnt, 8, hipMemcpyDeviceToHost));
        return h;
    }
    void release() { CK(hipFree(base)); }
};

static void impl(const char* s) { setenv("CM3P_GEMM_IMPL", s, 1); }

int main(int argc, char** argv) {
    bool do_check = argc < 2, do_time = argc < 2;
    if (argc >= 9 && !strcmp(argv[1], "one")) {  // one M N K epi a_kc b_kc splits [impl]: 20 launches of one shape (for rocprofv3 --pmc)
        const char* libpath1 = getenv("CM3P_HIP_LIB") ? getenv("CM3P_HIP_LIB") : "cm3p_amd/csrc/libcm3p_hip.so";
        void* lib1 = dlopen(libpath1, RTLD_NOW);
        if (!lib1) return 2;
        cm3p_gemm_bf16 = (gemm_fn)dlsym(lib1, "cm3p_gemm_bf16");
        const int64_t M = atoll(argv[2]), N = atoll(argv[3]), K = atoll(argv[4]);
        const int epi = atoi(argv[5]), a_kc = atoi(argv[6]), b_kc = atoi(argv[7]), splits = atoi(argv[8]);
        impl(argc > 9 ? argv[9] : "8p");
        Guarded A, B, R, C;
        A.alloc(M * K * 2); B.alloc(N * K * 2); R.alloc(M * N * 4); C.alloc(M * N * 4);
        float* ws1;
        CK(hipMalloc(&ws1, (size_t)std::max(1, splits) * M * N * 4));
        fill_bf16<<<1024, 256>>>((uint16_t*)A.p(), M * K, 1u, 1.f);
        fill_bf16<<<1024, 256>>>((uint16_t*)B.p(), N * K, 2u, 1.f);
        fill_f32<<<1024, 256>>>((float*)R.p(), M * N, 3u);
        for (int i = 0; i < 20; ++i)
            cm3p_gemm_bf16(A.p(), B.p(), C.p(), epi == CM3P_EPI_F32_RESID ? (const float*)R.p() : nullptr, M, N, K, a_kc ? K : M, b_kc ? K : N, N, a_kc, b_kc, epi,
                           splits, ws1, nullptr);
        CK(hipDeviceSynchronize());
        printf("ran 20 launches of [%ld x %ld x %ld]\n", (long)M, (long)N, (long)K);
        return 0;
    }
    if (argc >= 2 && !strcmp(argv[1], "batched")) {
        const char* libpathb = getenv("CM3P_HIP_LIB") ? getenv("CM3P_HIP_LIB") : "cm3p_amd/csrc/libcm3p_hip.so";
        void* libb = dlopen(libpathb, RTLD_NOW);
        if (!libb) return 2;
        cm3p_gemm_bf16_batched = (bgemm_fn)dlsym(libb, "cm3p_gemm_bf16_batched");
        unsigned long long* d_cnt;
        CK(hipMalloc(&d_cnt, 8));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        // (n, M, N, K, lda, ldb, a_kc, b_kc, with R): the three products of a Newton-Schulz step for the [2304, 768] x 44, [768, 768] x 22,
        // [1152, 768] x 22 ([768, 1152] stored) groups; operands in one buffer each, matrix strides = their sizes
        struct T { int n; int64_t M, N, K; int a_kc, b_kc, r; const char* name; };
        const T ts[] = {{44, 768, 768, 2304, 0, 0, 0, "A = X^T X   [2304 x 768] x 44 (ks,ks)"},
                        {44, 768, 768, 768, 1, 1, 1, "B = bA + cAA [768 x 768] x 44 (kc,kc)"},
                        {44, 2304, 768, 768, 1, 1, 1, "X' = aX + XB [2304 x 768] x 44 (kc,kc)"},
                        {22, 768, 768, 768, 1, 1, 0, "A = X X^T   [768 x 768] x 22 (kc,kc)"},
                        {22, 768, 768, 768, 1, 0, 1, "X' = aX + BX [768 x 768] x 22 (kc,ks)"},
                        {22, 768, 768, 1152, 1, 1, 0, "A = X X^T   [768 x 1152] x 22 (kc,kc)"},
                        {22, 768, 1152, 768, 1, 0, 1, "X' = aX + BX [768 x 1152] x 22 (kc,ks)"},
                        {30, 520, 264, 192, 1, 1, 1, "edge tiles  [520 x 264 x 192] x 30 (kc,kc)"},
                        {130, 256, 256, 128, 0, 0, 1, "many small  [256 x 256 x 128] x 130 (ks,ks)"}};
        int badb = 0;
        for (const T& t : ts) {
            const int64_t lda = t.a_kc ? t.K : t.M, ldb = t.b_kc ? t.K : t.N;
            const int64_t sa = t.M * t.K, sb = t.N * t.K, sc = t.M * t.N;
            Guarded A, B, R, C0, C1;
            A.alloc(t.n * sa * 2); B.alloc(t.n * sb * 2); R.alloc(t.n * sc * 2); C0.alloc(t.n * sc * 2); C1.alloc(t.n * sc * 2);
            fill_bf16<<<1024, 256>>>((uint16_t*)A.p(), t.n * sa, 1u, 1.f);
            fill_bf16<<<1024, 256>>>((uint16_t*)B.p(), t.n * sb, 2u, 1.f);
            fill_bf16<<<1024, 256>>>((uint16_t*)R.p(), t.n * sc, 3u, 1.f);
            CK(hipDeviceSynchronize());
            std::vector<float> ms[2];
            int rc[2] = {0, 0};
            for (int round = 0; round < 5; ++round)
                for (int which = 0; which < 2; ++which) {
                    impl(which ? "8p" : "256");
                    void* C = which ? C1.p() : C0.p();
                    const void* r = t.r ? R.p() : nullptr;
                    for (int i = 0; i < 2; ++i)
                        rc[which] |= cm3p_gemm_bf16_batched(A.p(), B.p(), C, r, t.n, t.M, t.N, t.K, lda, ldb, t.N, sa, sb, sc, sc, t.a_kc, t.b_kc, 2.0315f, -4.775f, nullptr);
                    CK(hipEventRecord(e0));
                    for (int i = 0; i < 10; ++i)
                        cm3p_gemm_bf16_batched(A.p(), B.p(), C, r, t.n, t.M, t.N, t.K, lda, ldb, t.N, sa, sb, sc, sc, t.a_kc, t.b_kc, 2.0315f, -4.775f, nullptr);
                    CK(hipEventRecord(e1));
                    CK(hipEventSynchronize(e1));
                    float m;
                    CK(hipEventElapsedTime(&m, e0, e1));
                    ms[which].push_back(m / 10);
                }
            for (auto& v : ms) std::sort(v.begin(), v.end());
            CK(hipMemset(d_cnt, 0, 8));
            count_diff_u32<<<1024, 256>>>((const uint32_t*)C0.p(), (const uint32_t*)C1.p(), (size_t)t.n * sc / 2, d_cnt);
            unsigned long long nd;
            CK(hipMemcpy(&nd, d_cnt, 8, hipMemcpyDeviceToHost));
            const unsigned long long g = C1.guards_touched(d_cnt);
            const double fl = 2.0 * t.n * t.M * t.N * t.K;
            const bool fail = nd || g || rc[0] || rc[1];
            printf("%-46s 128-tile %7.3f ms %7.1f TF/s | 8p %7.3f ms %7.1f TF/s | x%.2f | differing dwords %llu, guards %llu, rc %d %d %s\n", t.name,
                   ms[0][2], fl / ms[0][2] / 1e9, ms[1][2], fl / ms[1][2] / 1e9, ms[0][2] / ms[1][2], nd, g, rc[0], rc[1], fail ? "FAIL" : "OK");
            fflush(stdout);
            badb += fail;
            A.release(); B.release(); R.release(); C0.release(); C1.release();
        }
        return badb ? 1 : 0;
    }
    for (int i = 1; i < argc; ++i) {
        do_check |= !strcmp(argv[i], "check");
        do_time |= !strcmp(argv[i], "time");
    }
    const char* libpath = getenv("CM3P_HIP_LIB") ? getenv("CM3P_HIP_LIB") : "cm3p_amd/csrc/libcm3p_hip.so";
    void* lib = dlopen(libpath, RTLD_NOW);
    if (!lib) {
        printf("cannot load %s: %s\n", libpath, dlerror());
        return 2;
    }
    cm3p_gemm_bf16 = (gemm_fn)dlsym(lib, "cm3p_gemm_bf16");
    unsigned long long* d_cnt;
    CK(hipMalloc(&d_cnt, 8));
    struct Shape { int64_t M, N, K; };
    const Shape shapes[] = {{4096, 4096, 4096}, {131072, 2304, 768}, {65536, 768, 1152}, {65536 + 64, 1152, 768}, {8192 + 8, 2304 + 64, 64}, {256 * 30, 256 * 7, 128}};
    const int epis[] = {CM3P_EPI_BF16, CM3P_EPI_F32_RESID, CM3P_EPI_F32};
    const char* epi_names[] = {"bf16", "f32+resid", "f32"};
    int bad = 0;
    if (do_check) {
        for (const Shape& s : shapes) {
            Guarded A, B, R, C0, C1;
            A.alloc(s.M * s.K * 2);
            B.alloc(s.N * s.K * 2);
            R.alloc(s.M * s.N * 4);
            C0.alloc(s.M * s.N * 4);
            C1.alloc(s.M * s.N * 4);
            fill_bf16<<<1024, 256>>>((uint16_t*)A.p(), s.M * s.K, 1u, 1.f);
            fill_bf16<<<1024, 256>>>((uint16_t*)B.p(), s.N * s.K, 2u, 1.f);
            fill_f32<<<1024, 256>>>((float*)R.p(), s.M * s.N, 3u);
            CK(hipDeviceSynchronize());
            for (int lay = 0; lay < 4; ++lay)
            for (int e = 0; e < 3; ++e) {
                const int a_kc = !(lay & 1), b_kc = !(lay & 2);
                if (lay && e == 1) continue;
                if ((!a_kc && s.M % 8) || (!b_kc && s.N % 8)) continue;
                const int64_t lda = a_kc ? s.K : s.M, ldb = b_kc ? s.K : s.N;
                const size_t out_bytes = s.M * s.N * (epis[e] == CM3P_EPI_BF16 ? 2 : 4);
                printf("check [%ld x %ld x %ld] a_kc=%d b_kc=%d %-9s ... ", (long)s.M, (long)s.N, (long)s.K, a_kc, b_kc, epi_names[e]);
                fflush(stdout);
                impl("256");
                int rc0 = cm3p_gemm_bf16(A.p(), B.p(), C0.p(), epis[e] == CM3P_EPI_F32_RESID ? (const float*)R.p() : nullptr, s.M, s.N, s.K, lda, ldb, s.N, a_kc, b_kc, epis[e], 1, nullptr, nullptr);
                CK(hipDeviceSynchronize());
                printf("256 rc=%d ", rc0);
                fflush(stdout);
                impl("8p");
                int rc1 = cm3p_gemm_bf16(A.p(), B.p(), C1.p(), epis[e] == CM3P_EPI_F32_RESID ? (const float*)R.p() : nullptr, s.M, s.N, s.K, lda, ldb, s.N, a_kc, b_kc, epis[e], 1, nullptr, nullptr);
                CK(hipDeviceSynchronize());
                printf("8p rc=%d ", rc1);
                fflush(stdout);
                CK(hipMemset(d_cnt, 0, 8));
                count_diff_u32<<<1024, 256>>>((const uint32_t*)C0.p(), (const uint32_t*)C1.p(), out_bytes / 4, d_cnt);
                unsigned long long nd;
                CK(hipMemcpy(&nd, d_cnt, 8, hipMemcpyDeviceToHost));
                const unsigned long long g = C1.guards_touched(d_cnt);
                printf("differing dwords %llu / %zu, guard dwords touched %llu  %s\n", nd, out_bytes / 4, g, (nd || g || rc0 || rc1) ? "FAIL" : "OK");
                fflush(stdout);
                bad += (nd || g || rc0 || rc1) ? 1 : 0;
                // restore the part of C1 beyond out_bytes? (none: outputs are fully overwritten each time)
            }
            A.release(); B.release(); R.release(); C0.release(); C1.release();
        }
        printf("%d failing cases\n", bad);
    }
    if (do_time) {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        struct T { int64_t M, N, K; int epi; const char* name; int a_kc = 1, b_kc = 1, splits = 1; };
        const T ts[] = {{4096, 4096, 4096, CM3P_EPI_BF16, "cube 4096 bf16"}, {8192, 8192, 8192, CM3P_EPI_BF16, "cube 8192 bf16"},
                        {131072, 2304, 768, CM3P_EPI_BF16, "Wqkv/Wi fwd bf16"}, {131072, 768, 768, CM3P_EPI_F32_RESID, "Wo fwd f32+resid"},
                        {131072, 768, 768, CM3P_EPI_BF16, "Wo-shape bf16"}, {131072, 768, 1152, CM3P_EPI_F32_RESID, "Wo2 fwd f32+resid"},
                        {131072, 768, 2304, CM3P_EPI_BF16, "dgrad-shape K=2304 bf16 (kc,kc)"},
                        {131072, 768, 2304, CM3P_EPI_BF16, "dgrad Wqkv/Wi K=2304 bf16 (kc,ks)", 1, 0},
                        {131072, 1152, 768, CM3P_EPI_BF16, "dgrad Wo2 K=768 N=1152 bf16 (kc,ks)", 1, 0},
                        {131072, 768, 768, CM3P_EPI_BF16, "dgrad Wo K=768 bf16 (kc,ks)", 1, 0},
                        {8192, 8192, 8192, CM3P_EPI_BF16, "cube 8192 (kc,ks)", 1, 0},
                        {8192, 8192, 8192, CM3P_EPI_F32, "cube 8192 f32 (ks,ks)", 0, 0},
                        {2304, 768, 131072, CM3P_EPI_F32, "wgrad Wqkv/Wi split 9 (ks,ks)", 0, 0, 9},
                        {768, 768, 131072, CM3P_EPI_F32, "wgrad Wo split 28 (ks,ks)", 0, 0, 28},
                        {768, 1152, 131072, CM3P_EPI_F32, "wgrad Wo2 split 17 (ks,ks)", 0, 0, 17}};
        float* ws;
        CK(hipMalloc(&ws, (size_t)28 * 2304 * 1152 * 4));
        for (const T& t : ts) {
            Guarded A, B, R, C;
            A.alloc(t.M * t.K * 2); B.alloc(t.N * t.K * 2); R.alloc(t.M * t.N * 4); C.alloc(t.M * t.N * 4);
            fill_bf16<<<1024, 256>>>((uint16_t*)A.p(), t.M * t.K, 1u, 1.f);
            fill_bf16<<<1024, 256>>>((uint16_t*)B.p(), t.N * t.K, 2u, 1.f);
            fill_f32<<<1024, 256>>>((float*)R.p(), t.M * t.N, 3u);
            CK(hipDeviceSynchronize());
            std::vector<float> ms[2];
            for (int round = 0; round < 5; ++round)
                for (int which = 0; which < 2; ++which) {
                    impl(which ? "8p" : "256");
                    const float* r = t.epi == CM3P_EPI_F32_RESID ? (const float*)R.p() : nullptr;
                    for (int i = 0; i < 3; ++i) cm3p_gemm_bf16(A.p(), B.p(), C.p(), r, t.M, t.N, t.K, t.a_kc ? t.K : t.M, t.b_kc ? t.K : t.N, t.N, t.a_kc, t.b_kc, t.epi, t.splits, ws, nullptr);
                    CK(hipEventRecord(e0));
                    const int iters = 20;
                    for (int i = 0; i < iters; ++i) cm3p_gemm_bf16(A.p(), B.p(), C.p(), r, t.M, t.N, t.K, t.a_kc ? t.K : t.M, t.b_kc ? t.K : t.N, t.N, t.a_kc, t.b_kc, t.epi, t.splits, ws, nullptr);
                    CK(hipEventRecord(e1));
                    CK(hipEventSynchronize(e1));
                    float m;
                    CK(hipEventElapsedTime(&m, e0, e1));
                    ms[which].push_back(m / iters);
                }
            for (auto& v : ms) std::sort(v.begin(), v.end());
            const double fl = 2.0 * t.M * t.N * t.K;
            printf("%-34s 256: %7.3f ms (min %7.3f) %7.1f TF/s | 8p: %7.3f ms (min %7.3f) %7.1f TF/s | x%.3f\n", t.name, ms[0][2], ms[0][0],
                   fl / ms[0][2] / 1e9, ms[1][2], ms[1][0], fl / ms[1][2] / 1e9, ms[0][2] / ms[1][2]);
            fflush(stdout);
            A.release(); B.release(); R.release(); C.release();
        }
    }
    return bad ? 1 : 0;
}
