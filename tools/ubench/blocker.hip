// A stand-in for a communication kernel: `n` workgroups that each hold 64 KiB of LDS on a CU for `us` microseconds and do nothing else
// (tools/blocker_probe.py).  A ring-kernel GEMM workgroup needs a whole CU, so it cannot start on a CU where one of these sits.
//   hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ubench/libblocker.so tools/ubench/blocker.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ __launch_bounds__(256) void blocker_kernel(int64_t ticks, int* sink) {
    __shared__ int hold[16384];
    hold[threadIdx.x] = (int)threadIdx.x;
    __syncthreads();
    const int64_t t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);  // (exit condition every wave reaches: a fixed time)
    if (hold[(threadIdx.x * 7) & 16383] == -1) *sink = 1;
}

extern "C" int blocker_launch(int n, int us, int* sink, void* stream) {
    // wall_clock64 ticks at 100 MHz on gfx9
    blocker_kernel<<<n, 256, 0, static_cast<hipStream_t>(stream)>>>((int64_t)us * 100, sink);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
