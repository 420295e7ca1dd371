// What fp32-MFMA rate does this chip SUSTAIN? Back-to-back v_mfma_f32_32x32x2_f32 / 16x16x4 on every SIMD with RANDOM operands
// (the r01 probe multiplied constants for 3.5 ms: 155 TFLOP/s; the chip clocks to its power budget and random mantissas toggle
// far more of the multiplier array), for launches of ~0.1 ms to ~50 ms, 1..4 waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 tools/mfma_sustained.hip -o tools/mfma_sustained.bin && tools/mfma_sustained.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int SHAPE>  // 0: 32x32x2, 1: 16x16x4
__global__ __launch_bounds__(256) void burn(const float* __restrict__ src, float* out, int iters, int zero) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    float a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = zero ? 0.f : src[(t * 16 + i) & 0xfffff], b[i] = zero ? 0.f : src[(t * 16 + 8 + i) & 0xfffff];
    float s = 0;
    if (SHAPE == 0) {
        f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 8; k += 4) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k], b[k], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k + 1], b[k + 1], c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k + 2], b[k + 2], c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k + 3], b[k + 3], c3, 0, 0, 0);
            }
        }
        for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    } else {
        f32x4 c[8];
        for (int i = 0; i < 8; ++i) c[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 8; ++k) c[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[k], b[k], c[k], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < 8; ++k) c[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(k + 3) & 7], b[(k + 5) & 7], c[k], 0, 0, 0);
        }
        for (int k = 0; k < 8; ++k) s += c[k][0] + c[k][1] + c[k][2] + c[k][3];
    }
    out[t] = s;
}

int main() {
    const int N = 1 << 20;
    std::vector<float> h(N);
    srand(1);
    for (auto& v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    float *src, *out;
    hipMalloc(&src, N * 4), hipMalloc(&out, 256 * 16 * 256 * 4);
    hipMemcpy(src, h.data(), N * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    for (int shape = 0; shape < 2; ++shape)
        for (int zero = 1; zero >= 0; --zero)
            for (int wps = 1; wps <= 4; wps *= 2)          // waves per SIMD (blocks per CU)
                for (int iters : {200, 2000, 20000, 100000}) {
                    const int blocks = 256 * wps;
                    const int it = iters / wps;
                    float best = 1e30f;
                    for (int rep = 0; rep < 3; ++rep) {
                        hipEventRecord(e0);
                        if (shape == 0) burn<0><<<blocks, 256>>>(src, out, it, zero);
                        else burn<1><<<blocks, 256>>>(src, out, it, zero);
                        hipEventRecord(e1);
                        hipEventSynchronize(e1);
                        float ms;
                        hipEventElapsedTime(&ms, e0, e1);
                        if (ms < best) best = ms;
                    }
                    const double per_iter = shape == 0 ? 8 * 2.0 * 32 * 32 * 2 : 16 * 2.0 * 16 * 16 * 4;
                    const double flops = (double)blocks * 4 * it * per_iter;
                    printf("%s %s operands  %d wave(s)/SIMD  %8.3f ms  %6.1f TFLOP/s\n", shape ? "16x16x4" : "32x32x2",
                           zero ? "zero  " : "random", wps, best, flops / best / 1e9);
                }
    return 0;
}
