// Training kernels of the MBConv-specific pieces of efficientnet_b0 (timm tf_efficientnet_b0 blocks; reference
// model/feature_extractors.py:39-43 builds the network, single-step-learner.py:234 backpropagates through it):
//   depthwise convolution: data gradient (gather form) and filter gradient (per-tap reduction over B*Ho*Wo),
//   squeeze-excite: gate multiply, its backward (d gate = sum_hw dxg * x; dx = dxg * gate + d pooled / HW), the SE MLP
//   backward and its parameter gradients.
// NHWC fp32, float4 over channels; every reduction has a fixed order (deterministic, no atomics). All HBM-bound.
#include <algorithm>
#include <cstdlib>
#include "common.h"

namespace orbit {

using f32x4 = __attribute__((ext_vector_type(4))) float;

static void dw_layout(int C, int& G, int& R, int& ygroups) {
    const int Q = C / 4;
    G = Q < 64 ? Q : 64;
    R = 256 / G;
    ygroups = cdiv(Q, G);
}

static int grid_for(size_t total) {
    size_t b = (total + 255) / 256;
    return (int)(b > 8192 ? 8192 : (b ? b : 1));
}

// ---- squeeze-excite gate -------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gate_mul_kernel(const float* __restrict__ x, const float* __restrict__ gate,
                                                       float* __restrict__ xg, int HW, int C4, size_t total4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (size_t)gridDim.x * 256) {
        const unsigned iu = (unsigned)i;  // 32-bit index arithmetic (launcher: total < 2^32)
        const int q = (int)(iu % (unsigned)C4);
        const size_t b = iu / ((unsigned)HW * (unsigned)C4);
        reinterpret_cast<f32x4*>(xg)[i] = reinterpret_cast<const f32x4*>(x)[i] *
                                          reinterpret_cast<const f32x4*>(gate)[b * C4 + q];
    }
}

// pooled[b][c] = mean_hw x[b][hw][c] (the squeeze of squeeze-excite over the LARGE depthwise outputs: HW up to 112*112);
// block = (frame b, group of G channel quads), R row lanes, float4 loads, fixed-order LDS combine
__global__ __launch_bounds__(256) void colmean_kernel(const float* __restrict__ x, float* __restrict__ pooled, int HW,
                                                      int C4, int G, int R) {
    __shared__ f32x4 red[256];
    const int tid = threadIdx.x, b = blockIdx.x;
    const int rl = tid / G, qi = tid - rl * G, q = blockIdx.y * G + qi;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (rl < R && q < C4) {
        const size_t base = (size_t)b * HW * C4 + q;
#pragma unroll 4
        for (int r = rl; r < HW; r += R) s += reinterpret_cast<const f32x4*>(x)[base + (size_t)r * C4];
    }
    red[tid] = s;
    __syncthreads();
    if (rl == 0 && q < C4) {
        for (int j = 1; j < R; ++j) s += red[j * G + qi];
        reinterpret_cast<f32x4*>(pooled)[(size_t)b * C4 + q] = s * (1.0f / (float)HW);
    }
}

// dgate[b][c] = sum_hw dxg[b][hw][c] * x[b][hw][c]; block = (frame b, group of G channel quads), R row lanes
__global__ __launch_bounds__(256) void gate_bwd_reduce_kernel(const float* __restrict__ dxg,
                                                              const float* __restrict__ x, float* __restrict__ dgate,
                                                              int HW, int C4, int G, int R) {
    __shared__ f32x4 red[256];
    const int tid = threadIdx.x, b = blockIdx.x;
    const int rl = tid / G, qi = tid - rl * G, q = blockIdx.y * G + qi;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (rl < R && q < C4) {
        const size_t base = (size_t)b * HW * C4 + q;
#pragma unroll 4
        for (int r = rl; r < HW; r += R)
            s += reinterpret_cast<const f32x4*>(dxg)[base + (size_t)r * C4] *
                 reinterpret_cast<const f32x4*>(x)[base + (size_t)r * C4];
    }
    red[tid] = s;
    __syncthreads();
    if (rl == 0 && q < C4) {
        for (int j = 1; j < R; ++j) s += red[j * G + qi];
        reinterpret_cast<f32x4*>(dgate)[(size_t)b * C4 + q] = s;
    }
}

// dx = dxg * gate + dpooled / HW   (the second term is the gradient through the SE average pool)
__global__ __launch_bounds__(256) void gate_bwd_apply_kernel(const float* __restrict__ dxg,
                                                             const float* __restrict__ gate,
                                                             const float* __restrict__ dpooled,
                                                             float* __restrict__ dx, int HW, int C4, size_t total4) {
    const float inv = 1.0f / (float)HW;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (size_t)gridDim.x * 256) {
        const unsigned iu = (unsigned)i;  // 32-bit index arithmetic (launcher: total < 2^32)
        const int q = (int)(iu % (unsigned)C4);
        const size_t b = iu / ((unsigned)HW * (unsigned)C4);
        reinterpret_cast<f32x4*>(dx)[i] =
            reinterpret_cast<const f32x4*>(dxg)[i] * reinterpret_cast<const f32x4*>(gate)[b * C4 + q] +
            reinterpret_cast<const f32x4*>(dpooled)[b * C4 + q] * inv;
    }
}

// The same, fused with the first pass of the depthwise BatchNorm's backward (the tensor dx is the gradient of that
// BatchNorm's activation output): g = (dxg * gate + dpooled / HW) * act'(y * scale + shift) is written instead of dx, and
// the per-channel sums of g and g * xhat (xhat = (y - mean) * invstd) of every (frame, row chunk) go to
// partial[(b * chunks + chunk)][2][C] - what bn_bwd_finalize reads. The BatchNorm backward then needs no reduction pass
// over (dx, y) of its own and its apply pass takes g as is. Block = (row chunk, column group, frame), thread = channel quad x
// row lane (the layout of scale_shift_act_pool_kernel).
__global__ __launch_bounds__(256) void gate_bwd_apply_bn_kernel(const float* __restrict__ dxg,
                                                                const float* __restrict__ gate,
                                                                const float* __restrict__ dpooled,
                                                                const float* __restrict__ y,
                                                                const float* __restrict__ mean,
                                                                const float* __restrict__ invstd,
                                                                const float* __restrict__ scale,
                                                                const float* __restrict__ shift, int act, int HW, int C,
                                                                int rows_per_chunk, int G, int R, float* __restrict__ gout,
                                                                float* __restrict__ partial) {
    __shared__ f32x4 red[2][256];
    const int tid = threadIdx.x;
    const int rl = tid / G, qi = tid - rl * G, q = blockIdx.y * G + qi;
    const bool active = rl < R && q < (C >> 2);
    const int chunk = blockIdx.x, b = blockIdx.z;
    const int r0 = chunk * rows_per_chunk, r1 = min(HW, r0 + rows_per_chunk);
    f32x4 s = {0.f, 0.f, 0.f, 0.f}, sx = {0.f, 0.f, 0.f, 0.f};
    if (active) {
        const f32x4 gt = *reinterpret_cast<const f32x4*>(gate + (size_t)b * C + q * 4);
        const f32x4 dp = *reinterpret_cast<const f32x4*>(dpooled + (size_t)b * C + q * 4) * (1.0f / (float)HW);
        const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + q * 4), is = *reinterpret_cast<const f32x4*>(invstd + q * 4);
        const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + q * 4), sh = *reinterpret_cast<const f32x4*>(shift + q * 4);
        const size_t base = (size_t)b * HW * C + q * 4;
#pragma unroll 2
        for (int r = r0 + rl; r < r1; r += R) {
            const size_t o = base + (size_t)r * C;
            f32x4 g = *reinterpret_cast<const f32x4*>(dxg + o) * gt + dp;
            const f32x4 yv = *reinterpret_cast<const f32x4*>(y + o);
            if (act == ORBIT_ACT_SILU) {
                const f32x4 z = yv * sc + sh;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-z[k]));
                    g[k] *= sg * (1.0f + z[k] * (1.0f - sg));
                }
            } else if (act == ORBIT_ACT_RELU) {
                const f32x4 z = yv * sc + sh;
#pragma unroll
                for (int k = 0; k < 4; ++k) g[k] = z[k] > 0.f ? g[k] : 0.f;
            }
            if (gout) *reinterpret_cast<f32x4*>(gout + o) = g;  // (nullptr: sums only, launch_bn_backward_reduced_gated rebuilds g)
            s += g;
            sx += g * ((yv - mu) * is);
        }
    }
    red[0][tid] = s, red[1][tid] = sx;
    __syncthreads();
    if (rl == 0 && q < (C >> 2)) {
        for (int j = 1; j < R; ++j) s += red[0][j * G + qi], sx += red[1][j * G + qi];
        const size_t blk = (size_t)b * gridDim.x + chunk;
        *reinterpret_cast<f32x4*>(partial + (blk * 2 + 0) * C + q * 4) = s;
        *reinterpret_cast<f32x4*>(partial + (blk * 2 + 1) * C + q * 4) = sx;
    }
}

// SE MLP backward for one frame per block: pooled p[C] -> v = W1 p + b1 -> h = silu(v) -> u = W2 h + b2 -> g = sigmoid(u)
// given dgate: du = dgate g (1-g); dh = W2^T du; dv = dh silu'(v); dp = W1^T dv.  W1 [R][C], W2 [C][R].
// Writes du[b][C], dv[b][R], h[b][R] (for the parameter gradients) and dp[b][C].
// w2t (optional): W2 transposed to [R][C] - what the plan packs for the inference gate kernel. With it every pass over W2 is
// contiguous along c (the [C][R] layout is read with a stride of R floats by both the u and the dh pass: 12 of the 16 lanes
// of a request wasted at R = 48, and the kernel is a chain of L2 round trips through one CU).
__global__ __launch_bounds__(256) void se_bwd_kernel(const float* __restrict__ pooled, const float* __restrict__ w1,
                                                     const float* __restrict__ b1, const float* __restrict__ w2,
                                                     const float* __restrict__ w2t,
                                                     const float* __restrict__ b2, const float* __restrict__ dgate,
                                                     float* __restrict__ du_out, float* __restrict__ dv_out,
                                                     float* __restrict__ h_out, float* __restrict__ dp_out, int C,
                                                     int R) {
    extern __shared__ float sm[];  // p[C] du[C] v[R] h[R] dv[R]
    float* p = sm;
    float* du = sm + C;
    float* v = sm + 2 * C;
    float* h = v + R;
    float* dv = h + R;
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int c = tid; c < C; c += 256) p[c] = pooled[(size_t)b * C + c];
    __syncthreads();
    // v[r]: 16 lanes per hidden unit (R <= 64 in efficientnet_b0), fixed shuffle-free order through LDS
    for (int r = tid >> 4; r < R; r += 16) {
        const int l = tid & 15;
        float acc = 0.f;
        {   // eight loads in flight per lane (a rolled loop is one L2 round trip per iteration: C / 16 = 72 of them at C = 1152)
            int c = l;
            for (; c + 7 * 16 < C; c += 8 * 16) {
                float wv[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) wv[k] = w1[(size_t)r * C + c + 16 * k];
#pragma unroll
                for (int k = 0; k < 8; ++k) acc = fmaf(wv[k], p[c + 16 * k], acc);
            }
            for (; c < C; c += 16) acc = fmaf(w1[(size_t)r * C + c], p[c], acc);
        }
        // reduce the 16 lanes
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 16);
        if (l == 0) {
            const float vv = acc + b1[r];
            v[r] = vv;
            h[r] = vv * __builtin_amdgcn_rcpf(1.0f + __expf(-vv));
        }
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        float u = b2[c];
        if (w2t != nullptr) {
            int r = 0;
            for (; r + 8 <= R; r += 8) {  // eight independent coalesced loads per step
                float wv[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) wv[k] = w2t[(size_t)(r + k) * C + c];
#pragma unroll
                for (int k = 0; k < 8; ++k) u = fmaf(wv[k], h[r + k], u);
            }
            for (; r < R; ++r) u = fmaf(w2t[(size_t)r * C + c], h[r], u);
        } else {
            for (int r = 0; r < R; ++r) u = fmaf(w2[(size_t)c * R + r], h[r], u);
        }
        const float g = __builtin_amdgcn_rcpf(1.0f + __expf(-u));
        const float d = dgate[(size_t)b * C + c] * g * (1.0f - g);
        du[c] = d;
        du_out[(size_t)b * C + c] = d;
    }
    __syncthreads();
    for (int r = tid >> 4; r < R; r += 16) {
        const int l = tid & 15;
        float acc = 0.f;
        if (w2t != nullptr) {
            int c = l;
            for (; c + 7 * 16 < C; c += 8 * 16) {
                float wv[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) wv[k] = w2t[(size_t)r * C + c + 16 * k];
#pragma unroll
                for (int k = 0; k < 8; ++k) acc = fmaf(du[c + 16 * k], wv[k], acc);
            }
            for (; c < C; c += 16) acc = fmaf(du[c], w2t[(size_t)r * C + c], acc);
        } else {
            for (int c = l; c < C; c += 16) acc = fmaf(du[c], w2[(size_t)c * R + r], acc);
        }
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 16);
        if (l == 0) {
            const float vv = v[r];
            const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-vv));
            const float d = acc * sg * (1.0f + vv * (1.0f - sg));
            dv[r] = d;
            dv_out[(size_t)b * R + r] = d;
            h_out[(size_t)b * R + r] = h[r];
        }
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        float d = 0.f;
        int r = 0;
        for (; r + 8 <= R; r += 8) {
            float wv[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) wv[k] = w1[(size_t)(r + k) * C + c];
#pragma unroll
            for (int k = 0; k < 8; ++k) d = fmaf(dv[r + k], wv[k], d);
        }
        for (; r < R; ++r) d = fmaf(dv[r], w1[(size_t)r * C + c], d);
        dp_out[(size_t)b * C + c] = d;
    }
}

// dW2[c][r] = sum_b du[b][c] h[b][r]; db2[c] = sum_b du[b][c]; dW1[r][c] = sum_b dv[b][r] p[b][c]; db1[r] = sum_b dv[b][r]
// 16 outputs per block, 16 lanes per output: lane l sums frames l, l + 16, ... (13 frames at B = 200, all loads of a lane in
// flight at once), the lanes are combined by a fixed shuffle tree - deterministic. (One thread per output walking all B
// frames was a chain of B / 20 dependent L2 round trips: 43-49 us per launch whatever the layer size, 16 launches per step.)
__global__ __launch_bounds__(256) void se_param_grad_kernel(const float* __restrict__ du, const float* __restrict__ dv,
                                                            const float* __restrict__ h,
                                                            const float* __restrict__ pooled, int B, int C, int R,
                                                            float* __restrict__ dw1, float* __restrict__ db1,
                                                            float* __restrict__ dw2, float* __restrict__ db2) {
    const int total = 2 * C * R + C + R;
    const int l = threadIdx.x & 15;
    const int i = blockIdx.x * 16 + (threadIdx.x >> 4);
    const float* pa = nullptr;  // s = sum_b pa[b * sa] * pb[b * sb]   (pb == nullptr: sum_b pa[b * sa])
    const float* pb = nullptr;
    int sa = 0, sb = 0;
    float* out = nullptr;
    if (i < C * R) {  // dW2[c][r]
        const int c = i / R, r = i - c * R;
        pa = du + c, sa = C, pb = h + r, sb = R, out = dw2 + i;
    } else if (i < 2 * C * R) {  // dW1[r][c]
        const int j = i - C * R, r = j / C, c = j - r * C;
        pa = dv + r, sa = R, pb = pooled + c, sb = C, out = dw1 + j;
    } else if (i < 2 * C * R + C) {
        pa = du + (i - 2 * C * R), sa = C, out = db2 + (i - 2 * C * R);
    } else if (i < total) {
        pa = dv + (i - 2 * C * R - C), sa = R, out = db1 + (i - 2 * C * R - C);
    }
    float s = 0.f;
    if (pa != nullptr) {
        if (pb != nullptr) {
#pragma unroll 8
            for (int b = l; b < B; b += 16) s = fmaf(pa[(size_t)b * sa], pb[(size_t)b * sb], s);
        } else {
#pragma unroll 8
            for (int b = l; b < B; b += 16) s += pa[(size_t)b * sa];
        }
    }
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) s += __shfl_xor(s, off, 16);
    if (l == 0 && out != nullptr) *out = s;
}

// The four parameter gradients of up to SE_PARAM_JOBS squeeze-excite blocks in ONE launch (blockIdx.y = block): the reverse
// pass of a network collects its blocks' (du, dv, h, pooled) - each in its own scratch - and sums the frames once at the end.
// They are two skinny GEMMs over the frames, dW2 = du^T h [C][R] and dW1 = dv^T pooled [R][C]: a block owns 64 channels,
// stages 50 frames of (du, pooled) columns and of (h, dv) rows in LDS and every thread keeps a [<= 12] strip of both products
// (channel = lane, r = strip) in registers - each du / pooled element is read from L2 once instead of R times (the one-output-
// per-16-lanes form read 1.4 GB through the L2 per step: 16 launches, 446 us). Frames are added in order: deterministic.
constexpr int SEP_CT = 64, SEP_BT = 50, SEP_RG = 12;  // channels per block, frames per stage, r per thread (R <= 48)
__global__ __launch_bounds__(256) void se_param_grad_batched_kernel(SeParamJobs jobs) {
    const SeParamJob& j = jobs.j[blockIdx.y];
    const int B = j.B, C = j.C, R = j.R;
    const int c0 = blockIdx.x * SEP_CT;
    if (c0 >= C) return;
    __shared__ float du_s[SEP_BT][SEP_CT], p_s[SEP_BT][SEP_CT], h_s[SEP_BT][48], dv_s[SEP_BT][48];
    const int tid = threadIdx.x, cl = tid & (SEP_CT - 1), g = tid / SEP_CT;  // g = 0..3: r = g, g + 4, ...
    const int c = c0 + cl;
    float a2[SEP_RG], a1[SEP_RG], sdu = 0.f, sdv = 0.f;
#pragma unroll
    for (int k = 0; k < SEP_RG; ++k) a2[k] = a1[k] = 0.f;
    for (int b0 = 0; b0 < B; b0 += SEP_BT) {
        const int nb = min(SEP_BT, B - b0);
        __syncthreads();
        for (int i = tid; i < nb * SEP_CT; i += 256) {
            const int b = i / SEP_CT, cc = i - b * SEP_CT;
            const bool ok = c0 + cc < C;
            du_s[b][cc] = ok ? j.du[(size_t)(b0 + b) * C + c0 + cc] : 0.f;
            p_s[b][cc] = ok ? j.pooled[(size_t)(b0 + b) * C + c0 + cc] : 0.f;
        }
        for (int i = tid; i < nb * R; i += 256) {
            const int b = i / R, r = i - b * R;
            h_s[b][r] = j.h[(size_t)(b0 + b) * R + r];
            dv_s[b][r] = j.dv[(size_t)(b0 + b) * R + r];
        }
        __syncthreads();
        for (int b = 0; b < nb; ++b) {
            const float a = du_s[b][cl], p = p_s[b][cl];
            if (g == 0) sdu += a;
#pragma unroll
            for (int k = 0; k < SEP_RG; ++k) {
                const int r = g + 4 * k;
                if (r < R) a2[k] = fmaf(a, h_s[b][r], a2[k]), a1[k] = fmaf(dv_s[b][r], p, a1[k]);
            }
        }
        if (blockIdx.x == 0 && tid < R)
            for (int b = 0; b < nb; ++b) sdv += dv_s[b][tid];
    }
    if (c < C) {
#pragma unroll
        for (int k = 0; k < SEP_RG; ++k) {
            const int r = g + 4 * k;
            if (r < R) j.dw2[(size_t)c * R + r] = a2[k], j.dw1[(size_t)r * C + c] = a1[k];
        }
        if (g == 0) j.db2[c] = sdu;
    }
    if (blockIdx.x == 0 && tid < R) j.db1[tid] = sdv;
}

// ---- depthwise convolution ------------------------------------------------------------------------------------------
// dx[b][h][w][c] = sum_{kh,kw} dy[b][(h+pt-kh)/s][(w+pl-kw)/s][c] * w[kh][kw][c]   (terms with non-integral / out-of-range
// source positions vanish); w_khwc is the packed forward filter [K][K][C]
template <int K>
__global__ __launch_bounds__(256) void dwconv_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                           float* __restrict__ dx, int B, int H, int W, int C4,
                                                           int stride, int pad_t, int pad_l, int Ho, int Wo) {
    const size_t total = (size_t)B * H * W * C4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        // 32-bit index arithmetic (the launcher checks total < 2^32): 64-bit div / mod are ~100 instructions each
        const unsigned iu = (unsigned)i, r1 = iu / (unsigned)C4, r2 = r1 / (unsigned)W, bu = r2 / (unsigned)H;
        const int q = (int)(iu - r1 * C4), wx = (int)(r1 - r2 * W), h = (int)(r2 - bu * H), b = (int)bu;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kh = 0; kh < K; ++kh) {
            const int t = h + pad_t - kh;
            if (t < 0 || t % stride) continue;
            const int ho = t / stride;
            if (ho >= Ho) continue;
#pragma unroll
            for (int kw = 0; kw < K; ++kw) {
                const int u = wx + pad_l - kw;
                if (u < 0 || u % stride) continue;
                const int wo = u / stride;
                if (wo >= Wo) continue;
                acc += reinterpret_cast<const f32x4*>(dy)[(((size_t)b * Ho + ho) * Wo + wo) * C4 + q] *
                       reinterpret_cast<const f32x4*>(w)[(size_t)(kh * K + kw) * C4 + q];
            }
        }
        reinterpret_cast<f32x4*>(dx)[i] = acc;
    }
}

// Stride-2 form: a thread owns a channel quad and a 2x2 block of dx (rows h0, h0+1 / columns w0, w0+1, h0 and w0 even).
// With stride 2 an input pixel only receives the taps whose parity matches (kh = (h + pad_t) mod 2, +2, ...): the four pixels
// of the block together use every tap exactly once, and all of them read the same (K+1)/2 rows x columns of dy. The
// thread loads that dy patch once (4 quads for 3x3, 9 for 5x5; the gather kernel above issues up to K*K dy loads and K*K
// filter loads behind per-tap branches for EVERY pixel: 561 / 409 us on the 112x112x96 / 56x56x144 layers of
// efficientnet_b0 against ~240 / ~90 us of HBM time) and the filter taps from LDS. PT / PL = parity of pad_t / pad_l
// (compile time, so that every register index is static).
// BNB: the BatchNorm-backward epilogue of DwBnBwd (common.h): g = dx * act'(y * scale + shift) is written instead of dx and every
// block emits the channel sums of g and g * xhat over ITS 2x2 blocks (a fixed set: deterministic) to partial[blockIdx.x][2][C].
// WG (with BNB): the filter gradient of the layer rides along (DwBnBwd::wgrad_partial): the layer's input at the 2x2 block is
// act(y * scale + shift), already formed for the activation derivative, and tap (kh, kw) of output pixel (dh, dw) multiplies
// exactly the dy patch element the data gradient pairs it with - K*K more quad FMAs per 2x2 block, K*K accumulators per thread,
// one [K*K][C] partial row per block (the separate filter-gradient pass read y and dy again: 0.74 ms of the LITE step).
template <int K, int PT, int PL, bool BNB = false, bool WG = false>
__global__ __launch_bounds__(256) void dwconv_dgrad_s2_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                              float* __restrict__ dx, int B, int H, int W, int C4,
                                                              int pad_t, int pad_l, int Ho, int Wo, int G, int R,
                                                              DwBnBwd bnb = DwBnBwd{}) {
    constexpr int NR = (K + 1) / 2;  // dy rows / columns one 2x2 block touches (2 for 3x3, 3 for 5x5)
    extern __shared__ __attribute__((aligned(16))) float smd[];
    f32x4* wl = reinterpret_cast<f32x4*>(smd);  // [K*K][G] (+ [2][256] for the BNB sums)
    const int tid = threadIdx.x;
    const int rl = tid / G, qi = tid - rl * G, q = blockIdx.y * G + qi;
    for (int i = tid; i < K * K * G; i += 256) {
        const int tap = i / G, qq = blockIdx.y * G + (i - tap * G);
        wl[i] = qq < C4 ? reinterpret_cast<const f32x4*>(w)[(size_t)tap * C4 + qq] : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
    const bool live = rl < R && q < C4;
    if (!BNB && !live) return;
    const int H2 = (H + 1) >> 1, W2 = (W + 1) >> 1;
    const unsigned total = (unsigned)B * H2 * W2;  // 2x2 blocks (launcher: < 2^31)
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 s0 = zero, s1 = zero, bsc = zero, bsh = zero, bmu = zero, bis = zero;
    f32x4 wacc[WG ? K * K : 1];
#pragma unroll
    for (int t = 0; t < (WG ? K * K : 1); ++t) wacc[t] = zero;
    if (BNB && live) {
        bsc = reinterpret_cast<const f32x4*>(bnb.scale)[q], bsh = reinterpret_cast<const f32x4*>(bnb.shift)[q];
        bmu = reinterpret_cast<const f32x4*>(bnb.mean)[q], bis = reinterpret_cast<const f32x4*>(bnb.invstd)[q];
    }
    for (unsigned i = blockIdx.x * R + rl; live && i < total; i += gridDim.x * R) {
        const unsigned bw = i / (unsigned)W2, b = bw / (unsigned)H2;
        const int w0 = (int)(i - bw * W2) * 2, h0 = (int)(bw - b * H2) * 2;
        // dy patch: rows hb .. hb + NR - 1, hb = the smallest output row any tap of row h0 reaches
        const int hb = (h0 + pad_t - (K - 1) + (((K - 1) ^ PT) & 1)) / 2;  // (h0 + pad_t - kh_max) / 2, kh_max of parity PT
        const int wb = (w0 + pad_l - (K - 1) + (((K - 1) ^ PL) & 1)) / 2;
        // (h0 + pad_t - K + 1 can be negative by at most K - 1: the division above is exact on an even numerator >= -(K-1))
        const f32x4* dyb = reinterpret_cast<const f32x4*>(dy) + (size_t)b * Ho * Wo * C4 + q;
        f32x4 patch[NR][NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int ho = hb + r;
            const bool rok = (unsigned)ho < (unsigned)Ho;
#pragma unroll
            for (int c = 0; c < NR; ++c) {
                const int wo = wb + c;
                const bool ok = rok && (unsigned)wo < (unsigned)Wo;
                const f32x4 v = dyb[(size_t)((ok ? ho : 0) * Wo + (ok ? wo : 0)) * C4];
                patch[r][c] = ok ? v : zero;
            }
        }
        f32x4* dxb = reinterpret_cast<f32x4*>(dx) + (size_t)b * H * W * C4 + q;
        f32x4 yq[2][2];  // BNB: the producer's raw outputs under the 2x2 block, in flight with the dy patch
        if (BNB) {
            const f32x4* yb = reinterpret_cast<const f32x4*>(bnb.y) + (size_t)b * H * W * C4 + q;
#pragma unroll
            for (int dh = 0; dh < 2; ++dh)
#pragma unroll
                for (int dw = 0; dw < 2; ++dw)
                    yq[dh][dw] = yb[(size_t)(min(h0 + dh, H - 1) * W + min(w0 + dw, W - 1)) * C4];
        }
#pragma unroll
        for (int dh = 0; dh < 2; ++dh) {
#pragma unroll
            for (int dw = 0; dw < 2; ++dw) {
                const bool inside = h0 + dh < H && w0 + dw < W;
                // the activation's derivative and (WG) the layer's input at this pixel, before the taps
                f32x4 dact = {1.f, 1.f, 1.f, 1.f}, ain = zero;
                if (BNB) {
                    const f32x4 z = yq[dh][dw] * bsc + bsh;
                    ain = z;
                    if (bnb.act == ORBIT_ACT_SILU) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-z[k]));
                            dact[k] = sg * (1.0f + z[k] * (1.0f - sg));
                            ain[k] = z[k] * sg;
                        }
                    } else if (bnb.act == ORBIT_ACT_RELU) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) dact[k] = z[k] > 0.f ? 1.f : 0.f, ain[k] = z[k] > 0.f ? z[k] : 0.f;
                    }
                    if (!inside) ain = zero;
                }
                f32x4 acc = zero;
#pragma unroll
                for (int kh = (dh + PT) & 1; kh < K; kh += 2) {
                    // ho = (h0 + dh + pad_t - kh) / 2; relative to hb: static index because parities are compile time
                    const int r = (dh + ((K - 1) - (((K - 1) ^ PT) & 1)) - kh) / 2 + 0;
#pragma unroll
                    for (int kw = (dw + PL) & 1; kw < K; kw += 2) {
                        const int c = (dw + ((K - 1) - (((K - 1) ^ PL) & 1)) - kw) / 2;
                        acc += patch[r][c] * wl[(kh * K + kw) * G + qi];
                        if (WG) wacc[kh * K + kw] += patch[r][c] * ain;
                    }
                }
                if (inside) {
                    const size_t o = (size_t)((h0 + dh) * W + (w0 + dw)) * C4;
                    if (BNB) {
                        acc *= dact;
                        s0 += acc;
                        s1 += acc * ((yq[dh][dw] - bmu) * bis);
                    }
                    dxb[o] = acc;
                }
            }
        }
    }
    if (BNB) {
        f32x4* red = wl + K * K * G;  // [2][256]
        red[tid] = s0, red[256 + tid] = s1;
        __syncthreads();
        if (rl == 0 && q < C4) {
            f32x4 a = red[qi], c2 = red[256 + qi];
            for (int j = 1; j < R; ++j) a += red[j * G + qi], c2 += red[256 + j * G + qi];
            float* out = bnb.partial + (size_t)blockIdx.x * 2 * (C4 * 4) + q * 4;
            *reinterpret_cast<f32x4*>(out) = a;
            *reinterpret_cast<f32x4*>(out + C4 * 4) = c2;
        }
    }
    if (WG) {
        // filter-gradient partial row of this block: taps in rounds of 8 through the [8][256] buffer, row lanes added in order
        f32x4* red = wl + K * K * G;
        constexpr int TB = 8;
#pragma unroll
        for (int t0 = 0; t0 < K * K; t0 += TB) {
            __syncthreads();
#pragma unroll
            for (int u = 0; u < TB; ++u)
                if (t0 + u < K * K) red[u * 256 + tid] = wacc[t0 + u];
            __syncthreads();
            for (int i = tid; i < TB * G; i += 256) {
                const int u = i / G, qq = i - u * G;
                if (t0 + u < K * K && blockIdx.y * G + qq < C4) {
                    f32x4 a = red[u * 256 + qq];
                    for (int j = 1; j < R; ++j) a += red[u * 256 + j * G + qq];
                    *reinterpret_cast<f32x4*>(bnb.wgrad_partial + ((size_t)blockIdx.x * K * K + t0 + u) * (C4 * 4) +
                                              (blockIdx.y * G + qq) * 4) = a;
                }
            }
        }
    }
}

// partial[chunk][tap][c] = sum over the chunk's output pixels of dy * x(tap); block = (chunk of OUTPUT ROWS, group of G quads).
// A thread (channel quad, row lane) walks whole output rows, NB output columns at a time: per tap row it loads the
// (NB-1)*S + K input columns those NB outputs touch ONCE and reuses them across the K taps and the NB outputs -
// (NB + K*((NB-1)*S + K)) / NB loads per output (11 for 5x5 / stride 1) instead of 1 + K*K (26): the first form of this
// kernel (one output pixel per step, every tap a load behind its own bounds branch, 64-bit div / mod per pixel) was bound by
// the vector L1 and by address arithmetic (287 us per launch on the 5x5 layers of efficientnet_b0 @224). All index math is
// 32-bit; out-of-image taps read a clamped address and are zeroed by a select. Fixed summation order (deterministic).
template <int K, int S, int NB>
__global__ __launch_bounds__(256) void dwconv_wgrad_partial_kernel(const float* __restrict__ x,
                                                                   const float* __restrict__ dy,
                                                                   float* __restrict__ partial, int B, int H, int W,
                                                                   int C4, int pad_t, int pad_l, int Ho, int Wo,
                                                                   int rows_per_block, int G, int R) {
    constexpr int NW = (NB - 1) * S + K;
    __shared__ f32x4 red[256];
    const int tid = threadIdx.x;
    const int rl = tid / G, qi = tid - rl * G, q = blockIdx.y * G + qi;
    const bool active = rl < R && q < C4;
    const int total_rows = B * Ho;
    const int row0 = blockIdx.x * rows_per_block;
    const int row1 = row0 + rows_per_block < total_rows ? row0 + rows_per_block : total_rows;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 acc[K * K];
#pragma unroll
    for (int t = 0; t < K * K; ++t) acc[t] = zero;
    if (active) {
        const f32x4* x4 = reinterpret_cast<const f32x4*>(x) + q;
        const f32x4* dy4 = reinterpret_cast<const f32x4*>(dy) + q;
        for (int row = row0 + rl; row < row1; row += R) {
            const int b = row / Ho, ho = row - b * Ho;
            const f32x4* dyr = dy4 + (size_t)row * Wo * C4;
            const f32x4* xb = x4 + (size_t)b * H * W * C4;
            const int hi0 = ho * S - pad_t;
            for (int wo0 = 0; wo0 < Wo; wo0 += NB) {
                f32x4 d[NB];
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const bool ok = wo0 + j < Wo;
                    const f32x4 v = dyr[(size_t)(ok ? wo0 + j : 0) * C4];
                    d[j] = ok ? v : zero;
                }
                const int wi0 = wo0 * S - pad_l;
#pragma unroll
                for (int kh = 0; kh < K; ++kh) {
                    const int hi = hi0 + kh;
                    if ((unsigned)hi >= (unsigned)H) continue;   // block-divergent only at the image's top / bottom rows
                    const f32x4* xr = xb + (size_t)hi * W * C4;
                    f32x4 win[NW];
#pragma unroll
                    for (int u = 0; u < NW; ++u) {
                        const int wi = wi0 + u;
                        const bool ok = (unsigned)wi < (unsigned)W;
                        const f32x4 v = xr[(size_t)(ok ? wi : 0) * C4];
                        win[u] = ok ? v : zero;
                    }
#pragma unroll
                    for (int kw = 0; kw < K; ++kw)
#pragma unroll
                        for (int j = 0; j < NB; ++j) acc[kh * K + kw] += d[j] * win[j * S + kw];
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < K * K; ++t) {
        red[tid] = acc[t];
        __syncthreads();
        if (rl == 0 && q < C4) {
            f32x4 s = acc[t];
            for (int j = 1; j < R; ++j) s += red[j * G + qi];
            reinterpret_cast<f32x4*>(partial)[((size_t)blockIdx.x * K * K + t) * C4 + q] = s;
        }
        __syncthreads();
    }
}

// ---- depthwise filter gradient from an LDS-staged input patch (round 5) ------------------------------------------------------
// The kernel above loads every input element K times (once per tap row, through the vector L1). Here a block stages the input
// rows of one (frame, output-row chunk) tile for a slice of cs4 channel quads in LDS ONCE - the staging of dwconv_lds_kernel,
// csrc/ops.hip - and the taps read LDS; a block walks `tpb` tiles with its K*K accumulators in registers and reduces once at the
// end. Because an element is touched once on its way in, the preceding BatchNorm + SiLU can be applied THERE (XF: `x` is the RAW
// output of the expansion conv, act(x * in_scale[c] + in_shift[c]) is staged; the zero padding stays zero): the taped forward
// of a batch-statistics step then never writes the activated 6x-expanded tensor (conv_feeds_dw_raw, csrc/extractor_train.hip).
// partial[group][tap][c]; thread = (channel quad lc of the slice, 4-column group g, row lane rl); fixed summation order.
struct DwWgLdsParams {
    const float* x;
    const float* dy;
    float* partial;
    const float* in_scale;
    const float* in_shift;
    int in_act;
    int H, W, C, pad_t, pad_l, Ho, Wo, cs4, rpc, G, IWA, chunks, tiles_total, tpb;
};

template <int K, int S, bool XF>
__global__ __launch_bounds__(256) void dwconv_wgrad_lds_kernel(const DwWgLdsParams p) {
    constexpr int NCOL = 3 * S + K, U = 8;
    extern __shared__ __attribute__((aligned(16))) float smw[];
    const int cs4 = p.cs4, CSP = cs4 * 4 + 4, IWA = p.IWA;
    const int tid = threadIdx.x;
    const int lc = tid % cs4, pp = tid / cs4, P = 256 / cs4;
    const int c = (blockIdx.x * cs4 + lc) * 4;
    const int G = p.G, RL = P / G;
    const int g = pp % G, rl = pp / G;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 acc[K * K];
#pragma unroll
    for (int t = 0; t < K * K; ++t) acc[t] = zero;
    f32x4 xsc = {1.f, 1.f, 1.f, 1.f}, xsh = zero;
    if (XF) xsc = *reinterpret_cast<const f32x4*>(p.in_scale + c), xsh = *reinterpret_cast<const f32x4*>(p.in_shift + c);
    const int t_begin = blockIdx.y * p.tpb, t_end = min(p.tiles_total, t_begin + p.tpb);
    for (int tile_i = t_begin; tile_i < t_end; ++tile_i) {
        const int b = tile_i / p.chunks, chunk = tile_i - b * p.chunks;
        const int ho0 = chunk * p.rpc;
        const int TH = min(p.Ho, ho0 + p.rpc) - ho0;
        const int IH = (TH - 1) * S + K;
        __syncthreads();  // the previous tile's readers are done with the patch
        {
            const int hi0 = ho0 * S - p.pad_t;
            const float* xb = p.x + (size_t)b * p.H * p.W * p.C + c;
            const int n_items = IH * IWA;
            const int dr = P / IWA, dc = P % IWA;
            int r = pp / IWA, col = pp % IWA;
            for (int i0 = pp; i0 < n_items; i0 += U * P) {
                f32x4 v[U];
                int dst[U];
                unsigned loaded = 0;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const bool in = i0 + u * P < n_items;
                    const int hi = hi0 + r, wi = col - p.pad_l;
                    v[u] = zero;
                    dst[u] = in ? (r * IWA + col) * CSP + lc * 4 : -1;
                    if (in && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W) {
                        v[u] = *reinterpret_cast<const f32x4*>(xb + ((size_t)hi * p.W + wi) * p.C);
                        loaded |= 1u << u;
                    }
                    r += dr, col += dc;
                    if (col >= IWA) col -= IWA, ++r;
                }
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (dst[u] >= 0) {
                        f32x4 w = v[u];
                        if (XF && ((loaded >> u) & 1u)) {  // the arithmetic of dw_xf (csrc/ops.hip): what the forward kernel read
                            w = w * xsc + xsh;
                            if (p.in_act == ORBIT_ACT_SILU) {
#pragma unroll
                                for (int k = 0; k < 4; ++k) w[k] = w[k] * __builtin_amdgcn_rcpf(1.0f + __expf(-w[k]));
                            } else if (p.in_act == ORBIT_ACT_RELU) {
#pragma unroll
                                for (int k = 0; k < 4; ++k) w[k] = fmaxf(w[k], 0.f);
                            }
                        }
                        *reinterpret_cast<f32x4*>(smw + dst[u]) = w;
                    }
            }
        }
        // the output gradients of this thread's first row are requested BEFORE the barrier (they do not depend on the patch), the
        // next row's while the current one is consumed: a load per row iteration right before its use was a full HBM round trip
        // per iteration at 2-3 waves per SIMD (28x28x240, 5x5: 182 us against 63 us for the forward kernel on the same patch)
        const float* dyb = p.dy + ((size_t)b * p.Ho * p.Wo) * p.C + c;
        f32x4 dn[4];
        auto load_dy = [&](int ro) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int wo = g * 4 + j;
                const bool ok = wo < p.Wo && ro < TH && rl < RL;
                const f32x4 v = *reinterpret_cast<const f32x4*>(dyb + ((size_t)(ho0 + (ok ? ro : 0)) * p.Wo + (ok ? wo : 0)) * p.C);
                dn[j] = ok ? v : zero;
            }
        };
        load_dy(rl);
        __syncthreads();
        if (rl < RL) {
            for (int ro = rl; ro < TH; ro += RL) {
                f32x4 d[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) d[j] = dn[j];
                load_dy(ro + RL);
                const float* t0 = smw + ((size_t)(ro * S) * IWA + g * 4 * S) * CSP + lc * 4;
#pragma unroll
                for (int kh = 0; kh < K; ++kh) {
                    f32x4 col[NCOL];
#pragma unroll
                    for (int q = 0; q < NCOL; ++q) col[q] = *reinterpret_cast<const f32x4*>(t0 + (kh * IWA + q) * CSP);
#pragma unroll
                    for (int kw = 0; kw < K; ++kw)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[kh * K + kw] += d[j] * col[j * S + kw];
                }
            }
        }
    }
    // reduce over the P pixel lanes of every channel quad through LDS, TB taps per round (TB * 256 float4 = 36 KiB for TB = 9):
    // write [tap][lane][quad], then thread (tap, quad) sums its P lanes in lane order - two barriers per round instead of four
    // per tap (at 7x7 a block walks two tiles: 100 barriers of epilogue cost more than its work)
    constexpr int TB = 9;
    f32x4* red = reinterpret_cast<f32x4*>(smw);
#pragma unroll
    for (int t0 = 0; t0 < K * K; t0 += TB) {
        __syncthreads();
#pragma unroll
        for (int t = 0; t < TB; ++t)
            if (t0 + t < K * K) red[(t * P + pp) * cs4 + lc] = acc[t0 + t];
        __syncthreads();
        for (int o = tid; o < TB * cs4; o += 256) {
            const int t = o / cs4, q = o - t * cs4;
            if (t0 + t < K * K) {
                f32x4 tot = red[(t * P) * cs4 + q];
                for (int l = 1; l < P; ++l) tot += red[(t * P + l) * cs4 + q];
                *reinterpret_cast<f32x4*>(p.partial + ((size_t)blockIdx.y * K * K + t0 + t) * p.C + (blockIdx.x * cs4 + q) * 4) = tot;
            }
        }
    }
}

struct DwWgLdsGeom {
    bool ok = false;
    int cs4 = 0, rpc = 0, G = 0, IWA = 0, chunks = 0, tpb = 0, groups = 0;
    size_t lds = 0;
};
// the (channel slice, row chunk) whose patch fits 60 KiB with the most work per staged patch; blocks walk `tpb` tiles so that
// the launch has ~2048 blocks and at most 512 partial rows
static DwWgLdsGeom dw_wgrad_lds_geom(int B, int H, int W, int C, int K, int S, int Ho, int Wo) {
    DwWgLdsGeom g;
    if (C % 4 != 0 || (K != 3 && K != 5) || (S != 1 && S != 2)) return g;
    const int c4 = C / 4, G = cdiv(Wo, 4);
    const int IWA = (4 * G - 1) * S + K;
    long best = 0;
    for (int cand : {16, 8, 4}) {
        if (c4 % cand != 0 || G > 256 / cand) continue;
        const int RL = (256 / cand) / G;
        for (int rpc = std::min(Ho, 16); rpc >= 1; --rpc) {
            const int IH = (rpc - 1) * S + K;
            const size_t lds = std::max((size_t)IH * IWA * (cand * 4 + 4) * sizeof(float), (size_t)9 * 256 * 16);  // (epilogue buffer)
            if (lds > 60 * 1024) continue;
            // every row lane busy at least once, and prefer long contiguous runs per pixel
            const long score = (long)std::min(rpc, 4 * RL) * cand * (rpc >= RL ? 2 : 1);
            if (score > best) best = score, g.cs4 = cand, g.rpc = rpc, g.lds = lds;
            break;  // the largest rpc that fits for this slice width
        }
    }
    if (best == 0) return g;
    g.ok = true, g.G = G, g.IWA = IWA;
    g.chunks = cdiv(Ho, g.rpc);
    const int tiles = B * g.chunks, ncg = c4 / g.cs4;
    // (~4096 blocks / 2048 partial rows measured no faster: 260 against 237 us on the 112x112x32 layer)
    int groups = std::max(1, std::min(std::min(512, tiles), 2048 / std::max(ncg, 1)));
    g.tpb = cdiv(tiles, groups);
    g.groups = cdiv(tiles, g.tpb);
    return g;
}
bool dwconv_wgrad_xf_supported(int B, int H, int W, int C, int K, int stride, int Ho, int Wo) {
    return dw_wgrad_lds_geom(B, H, W, C, K, stride, Ho, Wo).ok;
}

// dw[c][0][kh][kw] = sum over chunks of partial[chunk][tap][c]: 16 outputs per block, 16 lanes per output take every
// 16th chunk, fixed-order LDS combine (a single thread walking ~1000 chunks is a 200 us latency chain)
__global__ __launch_bounds__(256) void dwconv_wgrad_reduce_kernel(const float* __restrict__ partial, int chunks, int KK,
                                                                  int C, float* __restrict__ dw) {
    __shared__ float sh[16][16];
    const int total = KK * C;
    const int ol = threadIdx.x & 15, ln = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + ol;
    float s = 0.f;
    if (i < total) {
#pragma unroll 4
        for (int k = ln; k < chunks; k += 16) s += partial[(size_t)k * total + i];
    }
    sh[ln][ol] = s;
    __syncthreads();
    if (ln == 0 && i < total) {
        for (int j = 1; j < 16; ++j) s += sh[j][ol];
        const int tap = i / C, c = i - tap * C;
        dw[(size_t)c * KK + tap] = s;
    }
}

int dwconv_wgrad_chunks(int B, int Ho, int Wo, int C) {
    int G, R, yg;
    dw_layout(C, G, R, yg);
    const size_t M = (size_t)B * Ho * Wo;
    size_t rows = (M + 1023) / 1024;
    if (rows < (size_t)4 * R) rows = (size_t)4 * R;
    return (int)((M + rows - 1) / rows);
}

// ---- launchers ---------------------------------------------------------------------------------------------------
int launch_colmean(const float* x, float* pooled, int B, int HW, int C, hipStream_t s) {
    ORBIT_REQUIRE(C % 4 == 0, "colmean: C %% 4 != 0");
    int G, R, yg;
    dw_layout(C, G, R, yg);
    colmean_kernel<<<dim3(B, yg), 256, 0, s>>>(x, pooled, HW, C / 4, G, R);
    ORBIT_LAUNCH_CHECK();
    return ORBIT_OK;
}

int launch_gate_mul(const float* x, const float* gate, float* xg, int B, int HW, int C, hipStream_t s) {
    ORBIT_REQUIRE(C % 4 == 0, "gate_mul: C %% 4 != 0");
    const size_t total4 = (size_t)B * HW * (C / 4);
    ORBIT_REQUIRE((unsigned long long)(total4) < (1ull << 32), "tensor too large for the 32-bit index arithmetic of this kernel");
    gate_mul_kernel<<<grid_for(total4), 256, 0, s>>>(x, gate, xg, HW, C / 4, total4);
    ORBIT_LAUNCH_CHECK();
    return ORBIT_OK;
}

// scratch: du [B*C] | dv [B*R] | h [B*R] | dgate [B*C] | dpooled [B*C]
size_t se_bwd_scratch_floats(int B, int C, int R) { return (size_t)B * (3 * (size_t)C + 2 * (size_t)R) + 16; }

SeParamJob se_bwd_param_job(const float* scratch, const float* pooled, int B, int C, int R, float* dw1, float* db1, float* dw2,
                            float* db2) {
    SeParamJob j;
    j.du = scratch, j.dv = scratch + (size_t)B * C, j.h = j.dv + (size_t)B * R, j.pooled = pooled;
    j.B = B, j.C = C, j.R = R, j.dw1 = dw1, j.db1 = db1, j.dw2 = dw2, j.db2 = db2;
    return j;
}
int launch_se_param_grad_batched(const SeParamJobs& jobs, int n, hipStream_t s) {
    ORBIT_REQUIRE(n >= 0 && n <= SE_PARAM_JOBS, "se_param_grad_batched: %d jobs", n);
    if (n == 0) return ORBIT_OK;
    int gx = 1;
    for (int k = 0; k < n; ++k) {
        ORBIT_REQUIRE(jobs.j[k].R <= 48, "se_param_grad_batched: R = %d > 48", jobs.j[k].R);
        gx = std::max(gx, cdiv(jobs.j[k].C, SEP_CT));
    }
    se_param_grad_batched_kernel<<<dim3(gx, n), 256, 0, s>>>(jobs);
    ORBIT_LAUNCH_CHECK();
    return ORBIT_OK;
}

const float* se_bwd_dpooled(const float* scratch, int B, int C, int R) {
    return scratch + (size_t)B * C + 2 * (size_t)B * R + (size_t)B * C;  // (the layout launch_se_gate_backward lays down)
}

int launch_se_gate_backward(const float* dxg, const float* x, const float* pooled, const float* gate, const float* w1,
                            const float* b1, const float* w2, const float* b2, float* dx, float* dw1, float* db1,
                            float* dw2, float* db2, float* scratch, int B, int HW, int C, int R, hipStream_t s,
                            const SeBnFuse* bn, const float* w2t) {
    ORBIT_REQUIRE(C % 4 == 0 && R > 0 && R <= 256, "se_gate_backward: bad sizes (C=%d R=%d)", C, R);
    ORBIT_REQUIRE(dx || bn, "se_gate_backward: dx may only be omitted with the BatchNorm fusion");
    float* du = scratch;
    float* dv = du + (size_t)B * C;
    float* h = dv + (size_t)B * R;
    float* dgate = h + (size_t)B * R;
    float* dp = dgate + (size_t)B * C;
    int G, Rl, yg;
    dw_layout(C, G, Rl, yg);
    gate_bwd_reduce_kernel<<<dim3(B, yg), 256, 0, s>>>(dxg, x, dgate, HW, C / 4, G, Rl);
    ORBIT_LAUNCH_CHECK();
    const size_t lds = (size_t)(2 * C + 3 * R) * sizeof(float);
    se_bwd_kernel<<<B, 256, lds, s>>>(pooled, w1, b1, w2, w2t, b2, dgate, du, dv, h, dp, C, R);
    ORBIT_LAUNCH_CHECK();
    if (dw1) {
        se_param_grad_kernel<<<cdiv(2 * C * R + C + R, 16), 256, 0, s>>>(du, dv, h, pooled, B, C, R, dw1, db1, dw2, db2);
        ORBIT_LAUNCH_CHECK();
    }
    if (bn != nullptr) {
        // the producer of x is a BatchNorm(+activation): its backward's reduction pass rides on this one
        const int chunks = se_pool_chunks(B, HW, C);
        int Gc, Rc, ygc;
        {
            const int Q = C / 4;
            Gc = Q < 256 ? Q : 256, Rc = 256 / Gc, ygc = cdiv(Q, Gc);  // the column layout of the BatchNorm kernels
        }
        gate_bwd_apply_bn_kernel<<<dim3(chunks, ygc, B), 256, 0, s>>>(dxg, gate, dp, bn->y, bn->mean, bn->invstd, bn->scale,
                                                                      bn->shift, bn->act, HW, C, cdiv(HW, chunks), Gc, Rc, dx,
                                                                      bn->partial);
        ORBIT_LAUNCH_CHECK();
        return ORBIT_OK;
    }
    const size_t total4 = (size_t)B * HW * (C / 4);
    ORBIT_REQUIRE((unsigned long long)(total4) < (1ull << 32), "tensor too large for the 32-bit index arithmetic of this kernel");
    gate_bwd_apply_kernel<<<grid_for(total4), 256, 0, s>>>(dxg, gate, dp, dx, HW, C / 4, total4);
    ORBIT_LAUNCH_CHECK();
    return ORBIT_OK;
}

// [K][K][C] taps -> taps rotated by 180 degrees
__global__ __launch_bounds__(256) void dwconv_flip_kernel(const float* __restrict__ w, float* __restrict__ wf, int KK, int C) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < KK * C) {
        const int tap = i / C, c = i - tap * C;
        wf[(KK - 1 - tap) * C + c] = w[i];
    }
}

static int dgrad_s2_blocks(int B, int H, int W, int C) {
    int G, R, yg;
    dw_layout(C, G, R, yg);
    const size_t blocks2 = (size_t)B * ((H + 1) / 2) * ((W + 1) / 2);
    // (2048 row blocks when the BatchNorm sums ride along: each is a partial row of 2 * C floats; 16384 without)
    return (int)std::min<size_t>((blocks2 + R - 1) / R, 16384);
}
int dwconv_dgrad_bn_blocks(int B, int H, int W, int C, int stride) {
    if (stride == 1) return B * dwconv_se_chunks(H);
    return std::min(dgrad_s2_blocks(B, H, W, C), 2048);
}
// row blocks of the stride-2 kernel when the filter gradient rides along: a block's [K*K + 2][C] partial rows stay a few
// percent of what it streams (about 9 quads per row lane and 2x2 block), at most 2048 rows
static int dgrad_s2_fused_blocks(int B, int H, int W, int C, int K) {
    int G, R, yg;
    dw_layout(C, G, R, yg);
    const size_t blocks2 = (size_t)B * ((H + 1) / 2) * ((W + 1) / 2);
    const size_t per_block = (size_t)R * (size_t)cdiv(K * K + 2, R);  // 2x2 blocks per row block: >= ~2 x the epilogue's quads / 9
    return (int)std::max<size_t>(1, std::min<size_t>(2048, blocks2 / std::max<size_t>(per_block * 2, 1)));
}
size_t dwconv_bwd_fused_partial_offset(int C, int K) { return (size_t)(C * K * K + 3) / 4 * 4; }
size_t dwconv_bwd_fused_scratch_floats(int B, int H, int W, int C, int K, int stride) {
    if (C % 4 != 0 || (K != 3 && K != 5)) return 0;
    if (stride == 2) return dwconv_bwd_fused_partial_offset(C, K) + (size_t)dgrad_s2_fused_blocks(B, H, W, C, K) * K * K * C;
    // stride 1: the register-window kernel (the data gradient is a forward depthwise pass over dy with H output rows)
    if (stride != 1 || !dwconv_se_window_form(K, 1, H)) return 0;
    return dwconv_bwd_fused_partial_offset(C, K) + (size_t)B * dwconv_se_chunks(H) * K * K * C;
}
int launch_dwconv_wgrad_reduce(const float* partial, int rows, int K, int C, float* dw, hipStream_t s) {
    ORBIT_REQUIRE(partial && dw && rows > 0, "dwconv_wgrad_reduce: bad arguments");
    dwconv_wgrad_reduce_kernel<<<cdiv(K * K * C, 16), 256, 0, s>>>(partial, rows, K * K, C, dw);
    ORBIT_LAUNCH_CHECK();
    return ORBIT_OK;
}

int launch_dwconv_dgrad(const float* dy, const float* w_khwc, float* dx, int B, int H, int W, int C, int K, int stride,
                        int pad_t, int pad_l, int Ho, int Wo, hipStream_t s, float* flip_scratch, const DwBnBwd* bnb) {
    if (bnb) {
        *bnb->nblk = 0;
        ORBIT_REQUIRE((bnb->wgrad_partial == nullptr) == (bnb->wgrad_rows == nullptr), "dwconv_dgrad: wgrad_partial without wgrad_rows");
        if (bnb->wgrad_rows) *bnb->wgrad_rows = 0;
    }
    if (stride == 1 && flip_scratch != nullptr) {
        // dx[h][w] = sum dy[h + pad_t - kh][w + pad_l - kw] w[kh][kw] = a forward depthwise conv of dy with the rotated
        // taps and padding K-1-pad: the LDS-patch forward kernels (35-60 us on these layers) replace the gather below
        // (180-200 us: K*K loads of dy and K*K loads of w per pixel, each behind its bounds branch)
        dwconv_flip_kernel<<<cdiv(K * K * C, 256), 256, 0, s>>>(w_khwc, flip_scratch, K * K, C);
        ORBIT_LAUNCH_CHECK();
        return launch_dwconv_se(dy, flip_scratch, dx, nullptr, nullptr, nullptr, B, Ho, Wo, C, K, 1, K - 1 - pad_t,
                                K - 1 - pad_l, H, W, ORBIT_ACT_NONE, s, 0, nullptr, nullptr, 0, bnb);
    }

    ORBIT_REQUIRE(C % 4 == 0 && (K == 3 || K == 5), "dwconv_dgrad: C %% 4 != 0 or K not in {3,5}");
    if (stride == 2 && (long long)B * ((H + 1) / 2) * ((W + 1) / 2) < (1ll << 31)) {
        // 2x2-block form: one dy patch + LDS taps per four outputs (see dwconv_dgrad_s2_kernel)
        int G, R, yg;
        dw_layout(C, G, R, yg);
        int gx = dgrad_s2_blocks(B, H, W, C);
        const bool fuse_wg = bnb && bnb->wgrad_partial;
        if (bnb) gx = std::min(gx, 2048);
        if (fuse_wg) gx = std::min(gx, dgrad_s2_fused_blocks(B, H, W, C, K)), *bnb->wgrad_rows = gx;
        if (bnb) *bnb->nblk = gx;
        const size_t lds = (size_t)(K * K * G + (fuse_wg ? 8 * 256 : bnb ? 512 : 0)) * sizeof(f32x4);
#define ORBIT_DG2(KK, PT_, PL_)                                                                                       \
    do {                                                                                                              \
        if (fuse_wg)                                                                                                  \
            dwconv_dgrad_s2_kernel<KK, PT_, PL_, true, true><<<dim3(gx, yg), 256, lds, s>>>(dy, w_khwc, dx, B, H, W, C / 4,   \
                                                                                            pad_t, pad_l, Ho, Wo, G, R, *bnb); \
        else if (bnb)                                                                                                 \
            dwconv_dgrad_s2_kernel<KK, PT_, PL_, true><<<dim3(gx, yg), 256, lds, s>>>(dy, w_khwc, dx, B, H, W, C / 4, pad_t,  \
                                                                                      pad_l, Ho, Wo, G, R, *bnb);     \
        else                                                                                                          \
            dwconv_dgrad_s2_kernel<KK, PT_, PL_><<<dim3(gx, yg), 256, lds, s>>>(dy, w_khwc, dx, B, H, W, C / 4, pad_t, pad_l, \
                                                                                Ho, Wo, G, R);                        \
    } while (0)
        const int pt = pad_t & 1, pl = pad_l & 1;
        if (K == 3) {
            if (!pt && !pl) ORBIT_DG2(3, 0, 0); else if (!pt) ORBIT_DG2(3, 0, 1); else if (!pl) ORBIT_DG2(3, 1, 0); else ORBIT_DG2(3, 1, 1);
        } else {
            if (!pt && !pl) ORBIT_DG2(5, 0, 0); else if (!pt) ORBIT_DG2(5, 0, 1); else if (!pl) ORBIT_DG2(5, 1, 0); else ORBIT_DG2(5, 1, 1);
        }
#undef ORBIT_DG2
        ORBIT_LAUNCH_CHECK();
        return ORBIT_OK;
    }
    const int grid = grid_for((size_t)B * H * W * (C / 4));
    ORBIT_REQUIRE((unsigned long long)((size_t)B * H * W * (C / 4)) < (1ull << 32), "tensor too large for the 32-bit index arithmetic of this kernel");
    if (K == 3) dwconv_dgrad_kernel<3><<<grid, 256, 0, s>>>(dy, w_khwc, dx, B, H, W, C / 4, stride, pad_t, pad_l, Ho, Wo);
    else dwconv_dgrad_kernel<5><<<grid, 256, 0, s>>>(dy, w_khwc, dx, B, H, W, C / 4, stride, pad_t, pad_l, Ho, Wo);
    ORBIT_LAUNCH_CHECK();
    return ORBIT_OK;
}

size_t dwconv_wgrad_scratch_floats(int B, int Ho, int Wo, int C, int K) {
    // (the LDS form writes at most 512 partial rows, whatever stride the caller's layer has)
    return (size_t)std::max(dwconv_wgrad_chunks(B, Ho, Wo, C), 512) * K * K * C;
}

int launch_dwconv_wgrad(const float* x, const float* dy, float* dw, float* scratch, int B, int H, int W, int C, int K,
                        int stride, int pad_t, int pad_l, int Ho, int Wo, hipStream_t s, const float* in_scale,
                        const float* in_shift, int in_act) {
    ORBIT_REQUIRE(C % 4 == 0 && (K == 3 || K == 5), "dwconv_wgrad: C %% 4 != 0 or K not in {3,5}");
    ORBIT_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "dwconv_wgrad: the input transform needs scale and shift");
    ORBIT_REQUIRE(stride == 1 || stride == 2, "dwconv_wgrad: stride %d", stride);
    const DwWgLdsGeom lg = dw_wgrad_lds_geom(B, H, W, C, K, stride, Ho, Wo);
    // Without the input transform the global-load form below stays the default: measured on the whole LITE step (one box,
    // profiles/r05_lite_ab_taped_xf.txt) 31.40 ms against 31.55 ms with the LDS form everywhere - its gain is the transform
    // (30.31 ms). ORBIT_DW_WGRAD_LDS=1 forces the LDS form (parity tests, A/B runs).
    static const char* lds_env = getenv("ORBIT_DW_WGRAD_LDS");
    ORBIT_REQUIRE(!in_scale || lg.ok, "dwconv_wgrad: the input transform needs the LDS form, which does not fit this layer");
    if (lg.ok && (in_scale || (lds_env && atoi(lds_env) == 1))) {
        DwWgLdsParams q;
        q.x = x, q.dy = dy, q.partial = scratch, q.in_scale = in_scale, q.in_shift = in_shift, q.in_act = in_act;
        q.H = H, q.W = W, q.C = C, q.pad_t = pad_t, q.pad_l = pad_l, q.Ho = Ho, q.Wo = Wo;
        q.cs4 = lg.cs4, q.rpc = lg.rpc, q.G = lg.G, q.IWA = lg.IWA, q.chunks = lg.chunks, q.tiles_total = B * lg.chunks, q.tpb = lg.tpb;
        const dim3 grid(C / 4 / lg.cs4, lg.groups);
#define ORBIT_DWWL(KK, SS)                                                                          \
    do {                                                                                            \
        if (in_scale) dwconv_wgrad_lds_kernel<KK, SS, true><<<grid, 256, lg.lds, s>>>(q);           \
        else dwconv_wgrad_lds_kernel<KK, SS, false><<<grid, 256, lg.lds, s>>>(q);                   \
    } while (0)
        if (K == 3 && stride == 1) ORBIT_DWWL(3, 1);
        else if (K == 3) ORBIT_DWWL(3, 2);
        else if (stride == 1) ORBIT_DWWL(5, 1);
        else ORBIT_DWWL(5, 2);
#undef ORBIT_DWWL
        ORBIT_LAUNCH_CHECK();
        dwconv_wgrad_reduce_kernel<<<cdiv(K * K * C, 16), 256, 0, s>>>(scratch, lg.groups, K * K, C, dw);
        ORBIT_LAUNCH_CHECK();
        return ORBIT_OK;
    }
    int G, R, yg;
    dw_layout(C, G, R, yg);
    ORBIT_REQUIRE((long long)B * H * W * (C / 4) < (1ll << 31), "dwconv_wgrad: tensor too large for 32-bit pixel indices");
    // chunks of whole output rows; never more blocks than the scratch was sized for (dwconv_wgrad_chunks), and at least one
    // row per row lane
    const int total_rows = B * Ho;
    int chunks = dwconv_wgrad_chunks(B, Ho, Wo, C);
    if (chunks > cdiv(total_rows, R)) chunks = cdiv(total_rows, R);
    const int rows = cdiv(total_rows, chunks);
    chunks = cdiv(total_rows, rows);
    dim3 grid(chunks, yg);
#define ORBIT_DWW(KK, SS, NBB)                                                                                          \
    dwconv_wgrad_partial_kernel<KK, SS, NBB><<<grid, 256, 0, s>>>(x, dy, scratch, B, H, W, C / 4, pad_t, pad_l, Ho, Wo, \
                                                                  rows, G, R)
    if (K == 3 && stride == 1) ORBIT_DWW(3, 1, 4);
    else if (K == 3) ORBIT_DWW(3, 2, 4);
    else if (stride == 1) ORBIT_DWW(5, 1, 4);
    else ORBIT_DWW(5, 2, 2);
#undef ORBIT_DWW
    ORBIT_LAUNCH_CHECK();
    dwconv_wgrad_reduce_kernel<<<cdiv(K * K * C, 16), 256, 0, s>>>(scratch, chunks, K * K, C, dw);
    ORBIT_LAUNCH_CHECK();
    return ORBIT_OK;
}

}  // namespace orbit

using namespace orbit;

extern "C" {

int orbit_op_dwconv2d_backward(const float* x, const float* w, const float* dy, float* dx, float* dw, int B, int H, int W,
                               int C, int K, int stride, int pad_top, int pad_left, int Ho, int Wo,
                               orbit_stream_t stream) {
    ORBIT_REQUIRE(x && w && dy, "op_dwconv2d_backward: null pointer");
    hipStream_t s = (hipStream_t)stream;
    float* tmp = nullptr;
    const size_t npack = (size_t)(C * K * K + 3) / 4 * 4;
    const size_t nscr = dwconv_wgrad_scratch_floats(B, Ho, Wo, C, K);
    ORBIT_HIP_CHECK(hipMallocAsync(reinterpret_cast<void**>(&tmp), (npack + nscr) * sizeof(float), s));
    int rc = dwconv_pack_weights(w, tmp, C, K, s);
    if (rc == ORBIT_OK && dx)
        rc = launch_dwconv_dgrad(dy, tmp, dx, B, H, W, C, K, stride, pad_top, pad_left, Ho, Wo, s, tmp + npack);
    if (rc == ORBIT_OK && dw)
        rc = launch_dwconv_wgrad(x, dy, dw, tmp + npack, B, H, W, C, K, stride, pad_top, pad_left, Ho, Wo, s);
    (void)hipFreeAsync(tmp, s);
    return rc;
}

int orbit_op_dwconv2d_dgrad_bn(const float* dy, const float* w, const float* y_raw, const float* mean, const float* invstd,
                               const float* scale, const float* shift, int act, float* g, float* sums, float* dw, int B, int H,
                               int W, int C, int K, int stride, int pad_top, int pad_left, int Ho, int Wo, orbit_stream_t stream) {
    ORBIT_REQUIRE(dy && w && y_raw && mean && invstd && scale && shift && g && sums, "op_dwconv2d_dgrad_bn: null pointer");
    hipStream_t s = (hipStream_t)stream;
    float* tmp = nullptr;
    const size_t npack = (size_t)(C * K * K + 3) / 4 * 4;
    const size_t nfused = dw ? dwconv_bwd_fused_scratch_floats(B, H, W, C, K, stride) : 0;
    if (dw && nfused == 0) return set_err(ORBIT_ERR_ARG, "op_dwconv2d_dgrad_bn: no kernel form carries the filter gradient for this layer");
    const size_t nscr = std::max(dwconv_wgrad_scratch_floats(B, Ho, Wo, C, K), nfused);
    const size_t npart = bn_partial_floats((size_t)dwconv_dgrad_bn_blocks(B, H, W, C, stride), C);
    ORBIT_HIP_CHECK(hipMallocAsync(reinterpret_cast<void**>(&tmp), (npack + nscr + npart) * sizeof(float), s));
    int nblk = 0, wrows = 0;
    DwBnBwd bnb{y_raw, mean, invstd, scale, shift, act, tmp + npack + nscr, &nblk};
    if (dw) bnb.wgrad_partial = tmp + npack + dwconv_bwd_fused_partial_offset(C, K), bnb.wgrad_rows = &wrows;
    int rc = dwconv_pack_weights(w, tmp, C, K, s);
    if (rc == ORBIT_OK) rc = launch_dwconv_dgrad(dy, tmp, g, B, H, W, C, K, stride, pad_top, pad_left, Ho, Wo, s, tmp + npack, &bnb);
    if (rc == ORBIT_OK && nblk <= 0) rc = set_err(ORBIT_ERR_ARG, "op_dwconv2d_dgrad_bn: no kernel form with the epilogue for this layer");
    if (rc == ORBIT_OK && dw && wrows <= 0) rc = set_err(ORBIT_ERR_ARG, "op_dwconv2d_dgrad_bn: no kernel form carries the filter gradient for this layer");
    if (rc == ORBIT_OK) rc = launch_sum_partials(bnb.partial, nblk, C, sums, s);
    if (rc == ORBIT_OK && dw) rc = launch_dwconv_wgrad_reduce(bnb.wgrad_partial, wrows, K, C, dw, s);
    (void)hipFreeAsync(tmp, s);
    return rc;
}

int orbit_op_dwconv2d_wgrad_xf(const float* x_raw, const float* in_scale, const float* in_shift, int in_act, const float* dy,
                               float* dw, int B, int H, int W, int C, int K, int stride, int pad_top, int pad_left, int Ho, int Wo,
                               orbit_stream_t stream) {
    ORBIT_REQUIRE(x_raw && in_scale && in_shift && dy && dw, "op_dwconv2d_wgrad_xf: null pointer");
    ORBIT_REQUIRE(dwconv_wgrad_xf_supported(B, H, W, C, K, stride, Ho, Wo), "op_dwconv2d_wgrad_xf: the LDS form does not fit this layer");
    hipStream_t s = (hipStream_t)stream;
    float* tmp = nullptr;
    ORBIT_HIP_CHECK(hipMallocAsync(reinterpret_cast<void**>(&tmp), dwconv_wgrad_scratch_floats(B, Ho, Wo, C, K) * sizeof(float), s));
    const int rc = launch_dwconv_wgrad(x_raw, dy, dw, tmp, B, H, W, C, K, stride, pad_top, pad_left, Ho, Wo, s, in_scale, in_shift,
                                       in_act);
    (void)hipFreeAsync(tmp, s);
    return rc;
}

int orbit_op_se_gate_backward(const float* dxg, const float* x, const float* pooled, const float* w1, const float* b1,
                              const float* w2, const float* b2, float* dx, float* dw1, float* db1, float* dw2,
                              float* db2, int B, int HW, int C, int R, orbit_stream_t stream) {
    ORBIT_REQUIRE(dxg && x && pooled && w1 && b1 && w2 && b2 && dx, "op_se_gate_backward: null pointer");
    ORBIT_REQUIRE((dw1 == nullptr) == (db1 == nullptr) && (dw1 == nullptr) == (dw2 == nullptr) &&
                      (dw1 == nullptr) == (db2 == nullptr),
                  "op_se_gate_backward: parameter gradients come all or none");
    hipStream_t s = (hipStream_t)stream;
    float* tmp = nullptr;
    const size_t nscr = se_bwd_scratch_floats(B, C, R);
    ORBIT_HIP_CHECK(hipMallocAsync(reinterpret_cast<void**>(&tmp), (nscr + (size_t)B * C + (size_t)C * R) * sizeof(float), s));
    float* gate = tmp + nscr;
    float* w2t = gate + (size_t)B * C;  // [R][C], as the network plans pack it
    int rc = launch_se_gate(pooled, w1, b1, w2, b2, gate, B, C, R, s);
    if (rc == ORBIT_OK) rc = launch_transpose(w2, w2t, C, R, s);
    if (rc == ORBIT_OK)
        rc = launch_se_gate_backward(dxg, x, pooled, gate, w1, b1, w2, b2, dx, dw1, db1, dw2, db2, tmp, B, HW, C, R, s, nullptr,
                                     w2t);
    (void)hipFreeAsync(tmp, s);
    return rc;
}

}  // extern "C"
