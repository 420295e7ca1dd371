// Does a wave's VALU stream run under another wave's MFMA chain on the same SIMD? Blocks of 8 waves (one block per CU):
// waves 0-3 (one per SIMD) run a dependent chain of v_mfma_f32_32x32x2_f32, waves 4-7 (one per SIMD) a VALU stream
// (plain FMA / packed FMA / SiLU with its two transcendentals). Timed alone and together with HIP events.
//   hipcc --offload-arch=gfx950 -O3 tools/coexec_probe.hip -o tools/coexec_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using v2f = __attribute__((ext_vector_type(2))) float;
constexpr int N = 8;

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
template <int OP, int MK = 0>
__global__ __launch_bounds__(512) void mix(float* out, int it_mfma, int it_valu, float a, float b) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float s = 0.f;
    if (wave < 4) {
        f32x16 acc, acc1, acc2, acc3;
        for (int r = 0; r < 16; ++r) acc[r] = a * r, acc1[r] = a + r, acc2[r] = b * r, acc3[r] = b + r;
        for (int it = 0; it < it_mfma; ++it) {
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                if (MK == 0) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
                if (MK == 2) {  // four independent accumulators, round robin
                    if ((k & 3) == 0) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
                    if ((k & 3) == 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc1, 0, 0, 0);
                    if ((k & 3) == 2) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc2, 0, 0, 0);
                    if ((k & 3) == 3) acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc3, 0, 0, 0);
                }
                if (MK == 3) {  // dependent chain, the wave idles on SALU nops until the previous MFMA has drained
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
                    asm volatile("s_nop 14");
                }
                if (MK == 4) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
                    asm volatile("s_nop 15\n s_nop 1");
                }
                if (MK == 1) {
                    bf16x8 av, bv;
                    for (int q = 0; q < 8; ++q) av[q] = (__bf16)a, bv[q] = (__bf16)b;
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc, 0, 0, 0);
                }
            }
        }
        for (int r = 0; r < 16; ++r) s += acc[r] + acc1[r] + acc2[r] + acc3[r];
    } else {
        float x[N];
        v2f p[N];
#pragma unroll
        for (int i = 0; i < N; ++i) x[i] = a + 0.001f * i + threadIdx.x * 1e-6f, p[i] = (v2f){x[i], x[i] + 0.5f};
        for (int it = 0; it < it_valu; ++it) {
#pragma unroll
            for (int i = 0; i < N; ++i) {
                if (OP == 0) x[i] = __builtin_fmaf(x[i], b, a);
                if (OP == 1) p[i] = __builtin_elementwise_fma(p[i], (v2f){b, b}, (v2f){a, a});
                if (OP == 2) x[i] = x[i] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x[i] * -1.44269502f));
            }
        }
#pragma unroll
        for (int i = 0; i < N; ++i) s += x[i] + p[i].x + p[i].y;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// one wave per SIMD: a dependent MFMA chain with NV independent v_fma_f32 after every MFMA, in the same instruction stream
template <int NV>
__global__ __launch_bounds__(256) void inwave(float* out, int iters, float a, float b) {
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = a * r;
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = a + 0.001f * i + threadIdx.x * 1e-6f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NV; ++i) x[i % 16] = __builtin_fmaf(x[i % 16], b, a);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc[r] + x[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NV>
void test_inwave(float* d) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    float best = 1e30f;
    const int iters = 4096;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        inwave<NV><<<256, 256>>>(d, iters, 0.7f, 0.99f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("in one wave: MFMA + %2d independent v_fma_f32 each: %6.1f ns per MFMA group\n", NV, 1e6 * best / (iters * 12));
}

template <int OP, int MK>
float run(float* d, int im, int iv) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        mix<OP, MK><<<256, 512>>>(d, im, iv, 0.7f, 0.99f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best * 1e3f;
}
template <int OP, int MK>
void test(const char* name, float* d, int iv) {
    const int im = MK == 1 ? 8192 : 4096;
    const float tm = run<OP, MK>(d, im, 0), tv = run<OP, MK>(d, 0, iv), tb = run<OP, MK>(d, im, iv);
    printf("%-28s MFMA chain alone %7.1f us (%.1f ns per MFMA)   VALU stream alone %7.1f us   together %7.1f us   (sum %7.1f, max %7.1f)\n",
           name, tm, 1e3 * tm / (im * 12), tv, tb, tm + tv, tm > tv ? tm : tv);
}
int main() {
    float* d;
    hipMalloc(&d, 256 * 512 * 4);
    test<0, 0>("f32 32x32x2  + v_fma_f32", d, 100000);
    test<1, 0>("f32 32x32x2  + v_pk_fma", d, 64000);
    test<2, 0>("f32 32x32x2  + SiLU", d, 12000);
    test<0, 2>("f32 4 indep accs + v_fma_f32", d, 100000);
    test<2, 2>("f32 4 indep accs + SiLU", d, 12000);
    test<0, 3>("f32 chain+s_nop + v_fma_f32", d, 100000);
    test<2, 3>("f32 chain+s_nop + SiLU", d, 12000);
    test<0, 4>("f32 chain+s_nop15,1 + v_fma", d, 100000);
    test<2, 4>("f32 chain+s_nop15,1 + SiLU", d, 12000);
    test<0, 1>("bf16 32x32x16 + v_fma_f32", d, 100000);
    test<1, 1>("bf16 32x32x16 + v_pk_fma", d, 64000);
    test<2, 1>("bf16 32x32x16 + SiLU", d, 12000);
    test_inwave<0>(d), test_inwave<4>(d), test_inwave<8>(d), test_inwave<12>(d), test_inwave<16>(d), test_inwave<24>(d), test_inwave<32>(d);
    return 0;
}
