// Cross-entropy loss of the learners: reference utils/optim.py:8-9 (`F.cross_entropy(test_logits, test_labels,
// reduction)`), called once per query batch from single-step-learner.py:225-232 and multi-step-learner.py /
// few_shot_recognisers.py:231-236 (FineTuner). logits are [N][C] with N <= a few thousand rows and C = the task's way
// (5-10; any C works): the op is latency-, not bandwidth-bound - what matters is that the step stays on the stream with
// no host read-back and that the sum over rows has ONE order (deterministic loss, run to run and across ranks).
//
//   forward : one wave per row: m = max_c z, s = sum_c exp(z - m), row_loss = log(s) - (z[label] - m),
//             softmax = exp(z - m) / s kept for the backward; a single block then adds the row losses in a fixed order
//             (thread t takes rows t, t + 256, ... in order, then a tree over the 256 partial sums).
//   backward: dz[i][c] = g_i * (softmax[i][c] - [c == label_i]),  g_i = grad / N (mean), grad (sum), grad[i] (none);
//             `grad` is read on the device (it is the upstream autograd value - no synchronisation).
// Accurate expf / logf (not the fast hardware approximations): the loss is compared with torch at 1e-6.
#include "common.h"

namespace orbit {

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_add(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__global__ __launch_bounds__(256) void ce_rows_kernel(const float* __restrict__ logits, const long* __restrict__ labels,
                                                      int N, int C, float* __restrict__ row_loss,
                                                      float* __restrict__ softmax) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N) return;
    const float* z = logits + (size_t)row * C;
    float m = -INFINITY;
    for (int c = lane; c < C; c += 64) m = fmaxf(m, z[c]);
    m = wave_max(m);
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += expf(z[c] - m);
    s = wave_add(s);
    const long lab = labels[row];
    // a label outside [0, C) has no defined loss: NaN, so that it cannot pass unnoticed (torch asserts on the device)
    const float picked = (lab >= 0 && lab < C) ? z[lab] - m : NAN;
    if (lane == 0) row_loss[row] = logf(s) - picked;
    if (softmax) {
        const float inv = 1.0f / s;
        for (int c = lane; c < C; c += 64) softmax[(size_t)row * C + c] = expf(z[c] - m) * inv;
    }
}

__global__ __launch_bounds__(256) void ce_reduce_kernel(const float* __restrict__ row_loss, int N, int mean,
                                                        float* __restrict__ loss) {
    __shared__ float part[256];
    float s = 0.f;
    for (int i = threadIdx.x; i < N; i += 256) s += row_loss[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = mean ? part[0] / (float)N : part[0];  // N = 0: 0/0 = NaN like torch's empty mean
}

__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* __restrict__ softmax, const long* __restrict__ labels,
                                                     const float* __restrict__ grad, int N, int C, int reduction,
                                                     float* __restrict__ dlogits) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N * C) return;
    const int row = i / C, c = i - row * C;
    const float g = reduction == ORBIT_REDUCE_NONE ? grad[row] : (reduction == ORBIT_REDUCE_MEAN ? grad[0] / (float)N : grad[0]);
    dlogits[i] = g * (softmax[i] - (labels[row] == c ? 1.0f : 0.0f));
}

}  // namespace orbit

using namespace orbit;

extern "C" {

int orbit_cross_entropy_forward(const float* logits, const int64_t* labels, int N, int C, int reduction, float* row_loss,
                                float* softmax, float* loss, orbit_stream_t stream) {
    ORBIT_REQUIRE(N >= 0 && C > 0, "cross_entropy: bad sizes (N=%d, C=%d)", N, C);
    ORBIT_REQUIRE((size_t)N * C < (1u << 31), "cross_entropy: N*C must stay below 2^31");
    ORBIT_REQUIRE(reduction == ORBIT_REDUCE_NONE || reduction == ORBIT_REDUCE_MEAN || reduction == ORBIT_REDUCE_SUM,
                  "cross_entropy: unknown reduction %d", reduction);
    ORBIT_REQUIRE(N == 0 || (logits && labels && row_loss), "cross_entropy: null pointer");
    ORBIT_REQUIRE(reduction == ORBIT_REDUCE_NONE || loss, "cross_entropy: reduced form needs the loss output");
    hipStream_t s = (hipStream_t)stream;
    if (N > 0) ce_rows_kernel<<<cdiv(N, 4), 256, 0, s>>>(logits, (const long*)labels, N, C, row_loss, softmax);
    if (reduction != ORBIT_REDUCE_NONE) ce_reduce_kernel<<<1, 256, 0, s>>>(row_loss, N, reduction == ORBIT_REDUCE_MEAN, loss);
    ORBIT_LAUNCH_CHECK();
    return ORBIT_OK;
}

int orbit_cross_entropy_backward(const float* softmax, const int64_t* labels, const float* grad, int N, int C,
                                 int reduction, float* dlogits, orbit_stream_t stream) {
    ORBIT_REQUIRE(N >= 0 && C > 0 && (size_t)N * C < (1u << 31), "cross_entropy_backward: bad sizes (N=%d, C=%d)", N, C);
    ORBIT_REQUIRE(reduction == ORBIT_REDUCE_NONE || reduction == ORBIT_REDUCE_MEAN || reduction == ORBIT_REDUCE_SUM,
                  "cross_entropy_backward: unknown reduction %d", reduction);
    if (N == 0) return ORBIT_OK;
    ORBIT_REQUIRE(softmax && labels && grad && dlogits, "cross_entropy_backward: null pointer");
    ce_bwd_kernel<<<cdiv(N * C, 256), 256, 0, (hipStream_t)stream>>>(softmax, (const long*)labels, grad, N, C, reduction,
                                                                    dlogits);
    ORBIT_LAUNCH_CHECK();
    return ORBIT_OK;
}

}  // extern "C"
