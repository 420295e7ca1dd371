// Standalone HBM bandwidth probe for this box: what a plain streaming kernel reaches (the practical ceiling the
// HBM-bound kernels of this build are compared with, next to the 8 TB/s spec figure).
// hipcc --offload-arch=gfx950 -O3 tools/hbm_probe.hip -o tools/hbm_probe.bin && tools/hbm_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;
__global__ __launch_bounds__(256) void copy_k(const f32x4* __restrict__ in, f32x4* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = in[i];
}
__global__ __launch_bounds__(256) void read_k(const f32x4* __restrict__ in, float* __restrict__ out, size_t n) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += in[i];
    if (s[0] + s[1] + s[2] + s[3] == 12345.678f) out[0] = 1.f;  // keeps the loads alive
}
__global__ __launch_bounds__(256) void write_k(f32x4* __restrict__ out, size_t n, float v) {
    const f32x4 x = {v, v, v, v};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = x;
}
int main() {
    const size_t bytes = (size_t)2 << 30, n = bytes / 16;  // 2 GiB per buffer: far beyond the 256 MB Infinity Cache
    f32x4 *a, *b;
    float* flag;
    hipMalloc(&a, bytes), hipMalloc(&b, bytes), hipMalloc(&flag, 4);
    hipMemset(a, 0, bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    for (int blocks : {2048, 8192, 32768}) {
        for (int mode = 0; mode < 3; ++mode) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) copy_k<<<blocks, 256>>>(a, b, n);
                else if (mode == 1) read_k<<<blocks, 256>>>(a, flag, n);
                else write_k<<<blocks, 256>>>(b, n, 1.f);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            const double moved = mode == 0 ? 2.0 * bytes : (double)bytes;
            printf("%-5s blocks %6d: %7.3f ms  %6.2f TB/s\n", mode == 0 ? "copy" : mode == 1 ? "read" : "write", blocks, best,
                   moved / (best * 1e-3) / 1e12);
        }
    }
    return 0;
}
