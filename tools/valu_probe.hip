// VALU issue-rate probe: SIMD time per wave-instruction of v_fma_f32, v_pk_fma_f32, v_exp_f32, v_rcp_f32 and of the library's
// SiLU (exp + rcp + 2 mul/add) on this box, with 1 / 2 / 4 / 8 waves per SIMD (HIP events around a chip-filling launch of
// unrolled independent chains). Output: profiles/r03_valu_probe.txt. hipcc --offload-arch=gfx950 -O3 tools/valu_probe.hip -o tools/valu_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
using v2f = __attribute__((ext_vector_type(2))) float;
constexpr int N = 8;      // independent chains
constexpr int IT = 4096;   // iterations of N instructions

template <int OP>
__global__ __launch_bounds__(1024) void probe(float* out, long long* cyc, float a, float b) {
    float x[N];
    v2f p[N];
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = a + 0.001f * i + threadIdx.x * 1e-6f, p[i] = (v2f){x[i], x[i] + 0.5f};
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < IT; ++it) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            if (OP == 0) x[i] = __builtin_fmaf(x[i], b, a);
            if (OP == 1) p[i] = __builtin_elementwise_fma(p[i], (v2f){b, b}, (v2f){a, a});
            if (OP == 2) x[i] = __builtin_amdgcn_exp2f(x[i] * b);
            if (OP == 3) x[i] = __builtin_amdgcn_rcpf(x[i] + a);
            if (OP == 4) x[i] = x[i] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x[i] * -1.44269502f));  // SiLU
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) s += x[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

// Event-timed: every SIMD of the chip holds `waves` resident waves running the same chain mix, so the time per
// wave-instruction is the SIMD's, at that occupancy.
template <int OP>
void run(const char* name, int elems_per_instr) {
    float* d;
    long long* c;
    hipMalloc(&d, (size_t)256 * 16 * 256 * 4);
    hipMalloc(&c, 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    printf("%-14s (%3d elements per wave-instruction): ns of one SIMD per wave-instruction at", name, elems_per_instr);
    for (int waves = 1; waves <= 8; waves *= 2) {
        const int blocks = 256 * waves;  // blocks of 4 waves, `waves` of them per CU
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0);
            probe<OP><<<blocks, 256>>>(d, c, 0.7f, 0.99f);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        const double instr_per_simd = (double)waves * IT * N;
        printf("  %d waves/SIMD %6.2f", waves, 1e6 * best / instr_per_simd);
    }
    printf("\n");
    hipFree(d), hipFree(c);
}
int main() {
    run<0>("v_fma_f32", 64);
    run<1>("v_pk_fma_f32", 128);
    run<2>("v_exp_f32+mul", 64);
    run<3>("v_rcp_f32+add", 64);
    run<4>("SiLU", 64);
    return 0;
}
