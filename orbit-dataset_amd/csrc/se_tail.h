// Squeeze-excite gate of ONE frame from pooling partials - shared by the stand-alone gate kernel (csrc/ops.hip
// se_gate2_kernel) and by the producers of the partials, whose LAST block per frame runs it itself (se_tail_finish below).
// Reference: timm SqueezeExcite inside tf_efficientnet_b0's blocks (x.mean((2, 3)) -> conv_reduce -> SiLU -> conv_expand ->
// sigmoid), reached through model/feature_extractors.py:39-43.
#pragma once
#include "common.h"

namespace orbit {

using v4f = __attribute__((ext_vector_type(4))) float;

// squeeze-excite gate from pooling partials: pooled[c] = (sum_chunks partial[b][chunk][c]) / HW, then
// g = sigmoid(W2 silu(W1 pooled + b1) + b2). w2t is W2 transposed to [R][C] so the second layer reads coalesced.
// One block per frame; the block pulls both weight matrices (up to 2 x 221 KB at C = 1152) through one CU's L1, so the
// kernel is a chain of L2 latencies: everything is float4 and every phase keeps 16-20 independent loads per lane in
// flight (layer 1: one wave per hidden unit, four units at a time; layer 2: eight hidden units per step).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float se_dpp_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float se_wave_sum(float v) {  // lane 63 holds the sum; returned wave-uniform
    v = se_dpp_add<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
    v = se_dpp_add<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
    v = se_dpp_add<0x141, 0xf>(v);  // row_half_mirror
    v = se_dpp_add<0x140, 0xf>(v);  // row_mirror
    v = se_dpp_add<0x142, 0xa>(v);  // row_bcast:15
    v = se_dpp_add<0x143, 0xc>(v);  // row_bcast:31
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// NT threads cooperate; U = hidden units a wave works on at a time (4 in the stand-alone kernel; 2 where the caller's register
// budget is tight - the arithmetic per unit, and with it every bit of the result, does not depend on U). `partial`, `gate`,
// `pooled_out` point at THIS frame's rows; sm2 needs ((C + R + 3) & ~3) + 4 * NT floats.
template <int NT, int U>
__device__ __forceinline__ void se_gate_frame(const float* __restrict__ partial, int chunks, float inv_hw,
                                              const float* __restrict__ w1, const float* __restrict__ b1,
                                              const float* __restrict__ w2t, const float* __restrict__ b2,
                                              float* __restrict__ gate, int C, int R, float* __restrict__ pooled_out, float* sm2) {
    v4f* sp4 = reinterpret_cast<v4f*>(sm2);
    float* hid = sm2 + C;
    const int tid = threadIdx.x;
    const int C4 = C >> 2;
    const v4f* part4 = reinterpret_cast<const v4f*>(partial);  // this frame's [chunks][C]
    const int parts = C4 <= NT / 2 ? NT / C4 : 1;  // thread groups sharing the chunk list of a channel quad
    if (parts > 1 && chunks > 8) {
        // many partials (the fused MBConv front writes one per 8x8 / 4x8 tile: up to 98) and few channels: 256 / C4 threads
        // per quad take every parts-th chunk, the groups' sums are added in group order (fixed order, deterministic)
        v4f* tmp = reinterpret_cast<v4f*>(sm2 + ((C + R + 3) & ~3));  // [parts][C4]
        const int q = tid % C4, part = tid / C4;
        if (part < parts) {
            v4f s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
            for (int k = part; k < chunks; k += parts) s += part4[(size_t)k * C4 + q];
            tmp[part * C4 + q] = s;
        }
        __syncthreads();
        if (tid < C4) {
            v4f s = tmp[tid];
            for (int g = 1; g < parts; ++g) s += tmp[g * C4 + tid];
            sp4[tid] = s * inv_hw;
        }
    } else {
        for (int c4 = tid; c4 < C4; c4 += NT) {
            v4f s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
            for (int k = 0; k < chunks; ++k) s += part4[(size_t)k * C4 + c4];  // loads batched, adds in chunk order
            sp4[c4] = s * inv_hw;
        }
    }
    __syncthreads();
    if (pooled_out != nullptr)  // training: the pooled means go on the tape (input of the gate MLP's backward)
        for (int c4 = tid; c4 < C4; c4 += NT) reinterpret_cast<v4f*>(pooled_out)[c4] = sp4[c4];
    // layer 1: wave w takes hidden units w, w + 4, ...; four units at a time, lanes stride the channel quads
    const int lane = tid & 63, wave = tid >> 6;
    const v4f* w14 = reinterpret_cast<const v4f*>(w1);
    for (int r0 = wave; r0 < R; r0 += U * (NT / 64)) {  // each wave takes units r0, r0 + NW, .. (U at a time)
        float acc[U];
#pragma unroll
        for (int u = 0; u < U; ++u) acc[u] = 0.f;
        for (int cb = 0; cb < C4; cb += 320) {  // 5 quads per lane per pass: C <= 1280 is a single pass
            v4f wv[U][5], pv[5];
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int c4 = cb + lane + 64 * j;
                const bool ok = c4 < C4;
                pv[j] = ok ? sp4[c4] : (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int r = r0 + (NT / 64) * u;
                    wv[u][j] = (ok && r < R) ? w14[(size_t)r * C4 + c4] : (v4f){0.f, 0.f, 0.f, 0.f};
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    const v4f t = wv[u][j] * pv[j];
                    acc[u] += (t[0] + t[1]) + (t[2] + t[3]);
                }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int r = r0 + (NT / 64) * u;
            const float sum = se_wave_sum(acc[u]);
            if (r < R && lane == 0) {
                const float t = sum + b1[r];
                hid[r] = t / (1.0f + expf(-t));
            }
        }
    }
    __syncthreads();
    // layer 2: thread = channel quad, eight hidden units (eight independent 16-byte loads) per step
    const v4f* w24 = reinterpret_cast<const v4f*>(w2t);
    for (int c4 = tid; c4 < C4; c4 += NT) {
        v4f a = *reinterpret_cast<const v4f*>(b2 + 4 * c4);
        int r = 0;
        for (; r + 8 <= R; r += 8) {
            v4f wv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) wv[u] = w24[(size_t)(r + u) * C4 + c4];
#pragma unroll
            for (int u = 0; u < 8; ++u) a += wv[u] * hid[r + u];
        }
        for (; r < R; ++r) a += w24[(size_t)r * C4 + c4] * hid[r];
        v4f g;
#pragma unroll
        for (int q = 0; q < 4; ++q) g[q] = 1.0f / (1.0f + expf(-a[q]));
        reinterpret_cast<v4f*>(gate)[c4] = g;
    }
}

// ---- the gate computed by the kernel that produces the pooling partials ------------------------------------------------------
// A depthwise / fused-MBConv launch used to be followed by a 200-block gate kernel that is nothing but a chain of L2 latencies
// (6-16 us, 16 of them per EfficientNet-B0 forward: 5 % of the kernel time, profiles/r03). Now every block of the producer
// counts itself in on a per-frame counter once its partial is written, and the block that completes a frame runs the gate for
// it - while the other frames' blocks are still computing. Hand-off between blocks of one launch (cdna_hip_programming.md
// Guideline 16, form R1): the partials are stored WRITE-THROUGH (agent-scope atomic stores = sc1: they leave the XCD's L2),
// every storing wave drains its stores (s_waitcnt vmcnt(0)), the block synchronises, ONE lane takes a ticket with a relaxed
// agent-scope fetch-add; the block that draws the last ticket issues ONE agent-scope acquire (drops its CU's stale L1 lines)
// and then reads the frame's partials with plain loads. A frame's partial rows are whole 128-byte lines (the launcher checks),
// so no line is shared with a frame another block of the same XCD read earlier. The counter is reset by the last block and
// zeroed by the plan before every forward.
struct SeTail {
    unsigned* counter = nullptr;  // [B] tickets; nullptr: no fused gate (the stand-alone kernel follows)
    int expected = 0;             // blocks per frame
    const float* w1 = nullptr;
    const float* b1 = nullptr;
    const float* w2t = nullptr;
    const float* b2 = nullptr;
    float* gate = nullptr;        // [B][C]
    const float* partial = nullptr;  // [B][chunks][C]: what this launch writes
    int chunks = 0, C = 0, R = 0;
    float inv_hw = 0.f;
};

// write-through store of a pooling partial quad (the plain store when no tail is attached)
__device__ __forceinline__ void se_store_partial(float* p, v4f t, bool write_through) {
    if (write_through) {
        __hip_atomic_store(p + 0, t[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(p + 1, t[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(p + 2, t[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(p + 3, t[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        *reinterpret_cast<v4f*>(p) = t;
    }
}

// Called by EVERY thread of a block after the block's partial stores (block-uniform control flow). sm: the block's dynamic LDS,
// free at this point, >= se_tail_lds_floats(C, R) floats.
__host__ __device__ inline size_t se_tail_lds_floats(int C, int R) { return (size_t)(((C + R + 3) & ~3) + 4 * 256 + 4); }
template <int U>
__device__ __forceinline__ void se_tail_finish(const SeTail& t, int frame, float* sm) {
    if (t.counter == nullptr) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's write-through partial stores have left the CU
    __syncthreads();                                  // (also: nobody still uses the LDS the gate is about to take)
    int* flag = reinterpret_cast<int*>(sm + (((t.C + t.R + 3) & ~3) + 4 * 256));
    if (threadIdx.x == 0) {
        const unsigned ticket = __hip_atomic_fetch_add(t.counter + frame, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool last = ticket == (unsigned)(t.expected - 1);
        if (last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __hip_atomic_store(t.counter + frame, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        *flag = last ? 1 : 0;
    }
    __syncthreads();
    if (*flag == 0) return;
    se_gate_frame<256, U>(t.partial + (size_t)frame * t.chunks * t.C, t.chunks, t.inv_hw, t.w1, t.b1, t.w2t, t.b2,
                          t.gate + (size_t)frame * t.C, t.C, t.R, nullptr, sm);
}

}  // namespace orbit
