// EfficientNet stem: 3x3 stride-2 convolution of the NCHW frames (3 -> 32 channels, TF-SAME padding) + folded BatchNorm
// + SiLU, written NHWC. Reference: timm `tf_efficientnet_b0.conv_stem` / `bn1` / act, reached from the reference's
// model/feature_extractors.py:39-43; FiLM (film.py:45-46 tags the root bn1) enters through scale / shift.
//
// The layer is HBM-bound (120 MB of frames in, 321 MB of activations out per 200 frames; 4.3 GFLOP): what matters is that
// the frames are read as whole rows and the output written as whole pixels. The implicit-GEMM kernel gathers its A
// operand element-wise from the NCHW frames (16 scalar loads per thread and K-tile: 26 TFLOP/s, 0.33 of HBM, 164 us).
// Here a block stages the 9 input rows x 3 channels that 4 output rows need into LDS with coalesced row loads and
// computes on the VALU: thread = (output pixel, group of 8 output channels), 27 taps x 8 channels = 216 FMAs from one LDS
// scalar (the pixel) and two LDS quads (the filter taps, broadcast across pixels) per tap; 4 adjacent lanes write the
// 128 contiguous bytes of a pixel. 864 FLOP per 140 bytes of traffic leaves the VALU far from binding (~35 us of issue).
#include "common.h"

namespace orbit {

using v4f = __attribute__((ext_vector_type(4))) float;

constexpr int STEM_TR = 4;                       // output rows per block
constexpr int STEM_IR = (STEM_TR - 1) * 2 + 3;   // input rows per block

__global__ __launch_bounds__(256) void stem_direct_kernel(const float* __restrict__ frames, const float* __restrict__ w,
                                                          const float* __restrict__ scale, const float* __restrict__ shift,
                                                          float* __restrict__ y, int H, int W, int pad_t, int pad_l, int Ho,
                                                          int Wo, int WP) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* patch = sm;                                   // [3][STEM_IR][WP]: column j holds input column j - pad_l
    float* wl = patch + 3 * STEM_IR * WP;                // [27][32] taps (kh, kw, ci) x output channel
    float* sc = wl + 27 * 32;                            // [32] scale, [32] shift
    const int tid = threadIdx.x;
    const int b = blockIdx.y, ho0 = blockIdx.x * STEM_TR;
    const int hi0 = ho0 * 2 - pad_t;

    // filter OIHW [32][3][3][3] -> wl[(kh * 3 + kw) * 3 + ci][co]; folded BatchNorm vectors
    for (int i = tid; i < 27 * 32; i += 256) {
        const int co = i & 31, k = i >> 5;
        const int ci = k % 3, kw = (k / 3) % 3, kh = k / 9;
        wl[i] = w[((co * 3 + ci) * 3 + kh) * 3 + kw];
    }
    if (tid < 32) sc[tid] = scale[tid], sc[32 + tid] = shift[tid];
    // input rows: zero outside the image; whole rows, float4 where the row start is 16-byte aligned
    const float* fb = frames + (size_t)b * 3 * H * W;
    for (int i = tid; i < 3 * STEM_IR * WP; i += 256) patch[i] = 0.f;
    __syncthreads();
    if ((W & 3) == 0) {
        const int w4 = W >> 2;
        for (int i = tid; i < 3 * STEM_IR * w4; i += 256) {
            const int x4 = i % w4, cr = i / w4, r = cr % STEM_IR, c = cr / STEM_IR;
            const int hi = hi0 + r;
            if ((unsigned)hi < (unsigned)H) {
                const v4f v = *reinterpret_cast<const v4f*>(fb + ((size_t)c * H + hi) * W + x4 * 4);
                float* dst = patch + (c * STEM_IR + r) * WP + pad_l + x4 * 4;
                dst[0] = v[0], dst[1] = v[1], dst[2] = v[2], dst[3] = v[3];
            }
        }
    } else {
        for (int i = tid; i < 3 * STEM_IR * W; i += 256) {
            const int x = i % W, cr = i / W, r = cr % STEM_IR, c = cr / STEM_IR;
            const int hi = hi0 + r;
            if ((unsigned)hi < (unsigned)H) patch[(c * STEM_IR + r) * WP + pad_l + x] = fb[((size_t)c * H + hi) * W + x];
        }
    }
    __syncthreads();

    // item = (output pixel, group of 8 output channels): 4 adjacent lanes write the 128 contiguous bytes of a pixel.
    // (A (pixel pair, channel quad) mapping - 8 lanes per pixel, fully contiguous 1 KB stores, half the tap-quad reads -
    // was measured SLOWER: 178 vs 122 us per 200 frames.)
    const int items = STEM_TR * Wo * 4;
    for (int it = tid; it < items; it += 256) {
        const int cg = it & 3, px = it >> 2;
        const int r = px / Wo, wo = px - r * Wo;
        const int ho = ho0 + r;
        if (ho >= Ho) break;  // rows are enumerated in order: the rest of this thread's items are past the image too
        v4f a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
        const float* prow = patch + (r * 2) * WP + wo * 2;
        const float* wq = wl + cg * 8;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                for (int ci = 0; ci < 3; ++ci) {
                    const float x = prow[(ci * STEM_IR + kh) * WP + kw];
                    const int k = (kh * 3 + kw) * 3 + ci;
                    a0 += x * *reinterpret_cast<const v4f*>(wq + k * 32);
                    a1 += x * *reinterpret_cast<const v4f*>(wq + k * 32 + 4);
                }
        const v4f s0 = *reinterpret_cast<const v4f*>(sc + cg * 8), s1 = *reinterpret_cast<const v4f*>(sc + cg * 8 + 4);
        const v4f h0 = *reinterpret_cast<const v4f*>(sc + 32 + cg * 8), h1 = *reinterpret_cast<const v4f*>(sc + 36 + cg * 8);
        v4f o0 = a0 * s0 + h0, o1 = a1 * s1 + h1;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            o0[q] = o0[q] * __builtin_amdgcn_rcpf(1.0f + __expf(-o0[q]));
            o1[q] = o1[q] * __builtin_amdgcn_rcpf(1.0f + __expf(-o1[q]));
        }
        float* dst = y + (((size_t)b * Ho + ho) * Wo + wo) * 32 + cg * 8;
        *reinterpret_cast<v4f*>(dst) = o0;
        *reinterpret_cast<v4f*>(dst + 4) = o1;
    }
}

// the direct kernel serves the EfficientNet stem shape with SiLU and frames whose 9-row patch fits 64 KB of LDS
bool stem_direct_supported(int Cin, int Cout, int K, int stride, int W, int act) {
    return Cin == 3 && Cout == 32 && K == 3 && stride == 2 && act == ORBIT_ACT_SILU && W >= 8 && W <= 1280;
}

int launch_stem_direct(const float* frames, const float* w_oihw, const float* scale, const float* shift, float* y, int B,
                       int H, int W, int pad_t, int pad_l, int Ho, int Wo, hipStream_t s) {
    ORBIT_REQUIRE(frames && w_oihw && scale && shift && y, "stem_direct: null pointer");
    ORBIT_REQUIRE(B > 0 && pad_t >= 0 && pad_t <= 1 && pad_l >= 0 && pad_l <= 1, "stem_direct: bad geometry");
    const int WP = ((Wo - 1) * 2 + 3 + pad_l + 3) & ~3;  // covers the right-most tap of the last output column
    const int WPn = WP > W + pad_l + 1 ? WP : ((W + pad_l + 1 + 3) & ~3);
    const size_t lds = ((size_t)3 * STEM_IR * WPn + 27 * 32 + 64) * sizeof(float);
    ORBIT_REQUIRE(lds <= 64 * 1024, "stem_direct: frame too wide (%d)", W);
    const double pix = (double)B * Ho * Wo;
    const int rec = prof_start("stem_direct<3x3/2,3->32>", 2.0 * pix * 32 * 27, 4.0 * ((double)B * 3 * H * W + pix * 32), s);
    stem_direct_kernel<<<dim3(cdiv(Ho, STEM_TR), B), 256, lds, s>>>(frames, w_oihw, scale, shift, y, H, W, pad_t, pad_l, Ho,
                                                                   Wo, WPn);
    prof_stop(rec, s);
    ORBIT_LAUNCH_CHECK();
    return ORBIT_OK;
}

}  // namespace orbit
