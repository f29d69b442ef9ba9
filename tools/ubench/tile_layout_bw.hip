// Does the TOKEN-MAJOR qkv layout ([T][3][nh][64] bf16: a head's row is 128 bytes out of every 4608) cost the sliding-window kernels memory rate
// against a HEAD-MAJOR layout ([3][nh][T][64]: a head's 64-row tile is 8 contiguous KiB)?  A stand-in with the band forward's traffic: one workgroup per
// (64-row tile, head) reads its q, k and v tiles (16-byte loads, all requested before the first is used) and writes one o tile.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/tile_layout_bw tools/ubench/tile_layout_bw.hip && tools/ubench/tile_layout_bw
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

template <bool HEAD_MAJOR, bool HEAD_FASTEST>
__global__ __launch_bounds__(256) void tile_kernel(const uint16_t* __restrict__ qkv, uint16_t* __restrict__ o, int T, int nh) {
    const int ntile = T / 64;
    int tile, head;
    if (HEAD_FASTEST) {
        head = blockIdx.x % nh;
        tile = blockIdx.x / nh;
    } else {
        tile = blockIdx.x % ntile;
        head = blockIdx.x / ntile;
    }
    const int tid = threadIdx.x;
    u32x4 v[3][2];
#pragma unroll
    for (int part = 0; part < 3; ++part)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int ch = tid + 256 * u, row = ch >> 3, c = ch & 7;  // 512 16-byte chunks per tile
            const int64_t off = HEAD_MAJOR ? (((int64_t)part * nh + head) * T + (int64_t)tile * 64 + row) * 64 + c * 8
                                           : ((int64_t)tile * 64 + row) * 3 * nh * 64 + ((int64_t)part * nh + head) * 64 + c * 8;
            v[part][u] = *reinterpret_cast<const u32x4*>(qkv + off);
        }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int ch = tid + 256 * u, row = ch >> 3, c = ch & 7;
        const u32x4 s = v[0][u] ^ v[1][u] ^ v[2][u];
        const int64_t off = HEAD_MAJOR ? ((int64_t)head * T + (int64_t)tile * 64 + row) * 64 + c * 8 : ((int64_t)tile * 64 + row) * nh * 64 + head * 64 + c * 8;
        *reinterpret_cast<u32x4*>(o + off) = s;
    }
}

int main() {
    const int T = 32 * 4096, nh = 12;
    uint16_t *qkv, *o;
    hipMalloc(&qkv, (size_t)T * 3 * nh * 64 * 2);
    hipMalloc(&o, (size_t)T * nh * 64 * 2);
    hipMemset(qkv, 1, (size_t)T * 3 * nh * 64 * 2);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const dim3 grid((T / 64) * nh);
    const double gb = (double)T * nh * 64 * 2 * 4 / 1e9;
    for (int round = 0; round < 2; ++round)
        for (int mode = 0; mode < 4; ++mode) {
            auto launch = [&]() {
                switch (mode) {
                    case 0: tile_kernel<false, false><<<grid, 256>>>(qkv, o, T, nh); break;
                    case 1: tile_kernel<false, true><<<grid, 256>>>(qkv, o, T, nh); break;
                    case 2: tile_kernel<true, false><<<grid, 256>>>(qkv, o, T, nh); break;
                    default: tile_kernel<true, true><<<grid, 256>>>(qkv, o, T, nh); break;
                }
            };
            for (int i = 0; i < 3; ++i) launch();
            hipEventRecord(e0);
            for (int i = 0; i < 20; ++i) launch();
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            ms /= 20;
            printf("round %d  %-12s %-22s %7.1f us  %.2f TB/s\n", round, mode < 2 ? "token-major" : "head-major", (mode & 1) ? "heads side by side" : "tiles of a head first", ms * 1e3,
                   gb / ms);
        }
    return 0;
}
