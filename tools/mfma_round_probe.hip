// How do the matrix cores round? D = C + sum_k A[i][k] B[k][j] with B = 1, A[i][k] = a_k for every i, C = c for every (i, j):
// every output element is c + sum_k a_k. Cases place the sum at fractions of c's ulp. Prints what v_mfma_f32_32x32x16_bf16 and
// v_mfma_f32_32x32x2_f32 return beside the exact value and its round-to-nearest-even / truncated fp32 forms.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/mfma_round_probe.bin tools/mfma_round_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void probe_bf16(const float* a16, float c, float* out) {
    const int lane = threadIdx.x, lh = lane >> 5;
    bf16x8 a, b;
    for (int q = 0; q < 8; ++q) a[q] = (__bf16)a16[8 * lh + q], b[q] = (__bf16)1.0f;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = c;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    if (lane == 0) out[0] = acc[0];
}
__global__ void probe_f32(const float* a2, float c, float* out) {
    const int lane = threadIdx.x, lh = lane >> 5;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = c;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[lh], 1.0f, acc, 0, 0, 0);
    if (lane == 0) out[0] = acc[0];
}

static float trunc_f32(double v) {  // toward zero
    float f = (float)v;
    if (std::fabs((double)f) > std::fabs(v)) f = std::nextafterf(f, 0.0f);
    return f;
}

int main() {
    float *da, *dout;
    hipMalloc(&da, 16 * sizeof(float));
    hipMalloc(&dout, sizeof(float));
    struct Case { const char* name; float c; float a[16]; } cases[] = {
        {"c=2^26 (ulp 8), 16 x 0.5  (sum = 1 ulp, every addend 1/16 ulp)", 67108864.f, {.5f,.5f,.5f,.5f,.5f,.5f,.5f,.5f,.5f,.5f,.5f,.5f,.5f,.5f,.5f,.5f}},
        {"c=2^26, 12 x 0.5 (sum = 0.75 ulp)", 67108864.f, {.5f,.5f,.5f,.5f,.5f,.5f,.5f,.5f,.5f,.5f,.5f,.5f,0,0,0,0}},
        {"c=2^26, 8 x 0.5 (sum = 0.5 ulp: tie)", 67108864.f, {.5f,.5f,.5f,.5f,.5f,.5f,.5f,.5f,0,0,0,0,0,0,0,0}},
        {"c=2^26+8 (odd mantissa), 8 x 0.5 (tie)", 67108872.f, {.5f,.5f,.5f,.5f,.5f,.5f,.5f,.5f,0,0,0,0,0,0,0,0}},
        {"c=2^26, one addend 6 (0.75 ulp)", 67108864.f, {6.f,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0}},
        {"c=2^26, one addend 3 (0.375 ulp)", 67108864.f, {3.f,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0}},
        {"c=-2^26, 12 x 0.5", -67108864.f, {.5f,.5f,.5f,.5f,.5f,.5f,.5f,.5f,.5f,.5f,.5f,.5f,0,0,0,0}},
        {"c=-2^26, 12 x -0.5", -67108864.f, {-.5f,-.5f,-.5f,-.5f,-.5f,-.5f,-.5f,-.5f,-.5f,-.5f,-.5f,-.5f,0,0,0,0}},
        {"c=1, addends 2^-24 x 16 (sum = 2^-20: 8 ulp of 1)", 1.0f, {5.9604645e-8f,5.9604645e-8f,5.9604645e-8f,5.9604645e-8f,5.9604645e-8f,5.9604645e-8f,5.9604645e-8f,5.9604645e-8f,5.9604645e-8f,5.9604645e-8f,5.9604645e-8f,5.9604645e-8f,5.9604645e-8f,5.9604645e-8f,5.9604645e-8f,5.9604645e-8f}},
        {"c=1, addends 2^-26 x 16 (sum = 2^-22: 2 ulp, each 1/8 ulp)", 1.0f, {1.4901161e-8f,1.4901161e-8f,1.4901161e-8f,1.4901161e-8f,1.4901161e-8f,1.4901161e-8f,1.4901161e-8f,1.4901161e-8f,1.4901161e-8f,1.4901161e-8f,1.4901161e-8f,1.4901161e-8f,1.4901161e-8f,1.4901161e-8f,1.4901161e-8f,1.4901161e-8f}},
        {"c=0, 256 + 15 x 2^-18 (sum of the small ones = 15 x 2^-18)", 0.0f, {256.f,3.8146973e-6f,3.8146973e-6f,3.8146973e-6f,3.8146973e-6f,3.8146973e-6f,3.8146973e-6f,3.8146973e-6f,3.8146973e-6f,3.8146973e-6f,3.8146973e-6f,3.8146973e-6f,3.8146973e-6f,3.8146973e-6f,3.8146973e-6f,3.8146973e-6f}},
    };
    for (auto& cs : cases) {
        double exact = cs.c;
        for (int k = 0; k < 16; ++k) exact += cs.a[k];
        hipMemcpy(da, cs.a, sizeof(cs.a), hipMemcpyHostToDevice);
        float got = 0;
        probe_bf16<<<1, 64>>>(da, cs.c, dout);
        hipMemcpy(&got, dout, 4, hipMemcpyDeviceToHost);
        // the fp32 MFMA takes two addends per instruction: feed the first two of the case (sum them pairwise into two values)
        float a2[2] = {0, 0};
        for (int k = 0; k < 16; ++k) a2[k >= 8] += cs.a[k];  // exact for these cases
        hipMemcpy(da, a2, sizeof(a2), hipMemcpyHostToDevice);
        float got32 = 0;
        probe_f32<<<1, 64>>>(da, cs.c, dout);
        hipMemcpy(&got32, dout, 4, hipMemcpyDeviceToHost);
        printf("%-62s exact %.10g  rne %.10g  trunc %.10g | bf16 MFMA %.10g | f32 MFMA (2 addends) %.10g\n", cs.name, exact,
               (double)(float)exact, (double)trunc_f32(exact), (double)got, (double)got32);
    }
    return 0;
}
