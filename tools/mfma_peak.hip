// Standalone MFMA fp32 peak probe for this box: back-to-back v_mfma_f32_32x32x2_f32 on 4 independent accumulators.
// hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/mfma_peak.bin && tools/mfma_peak.bin
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
__global__ __launch_bounds__(256) void peak(float* out, int iters, float a, float b) {
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c3, 0, 0, 0);
    }
    float s = 0;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* d;
    const int blocks = 256 * 8, iters = 4000;
    hipMalloc(&d, blocks * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        peak<<<blocks, 256>>>(d, iters, 1.0f + rep * 1e-3f, 0.5f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        double flops = (double)blocks * 4 /*waves*/ * iters * 4 * 2.0 * 32 * 32 * 2;
        printf("mfma_f32_32x32x2 peak probe: %.3f ms  %.1f TFLOP/s\n", ms, flops / ms / 1e9);
    }
    return 0;
}
