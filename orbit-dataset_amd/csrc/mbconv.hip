// Fused MBConv front half for gfx950: expand 1x1 conv (fp32 MFMA) -> BN1 -> SiLU -> depthwise KxK (TF-SAME) -> BN2
// -> SiLU -> squeeze-excite pooling partials, in ONE kernel. The 6x-expanded tensor lives only in LDS.
//
// Replaces, for EfficientNet-B0's early InvertedResidual blocks (Cin <= 40: blocks.1.0 .. blocks.3.0, which carry 75 %
// of the expanded-tensor bytes of the network), the reference's conv_pw -> bn1 -> act -> conv_dw -> bn2 -> act ->
// se.mean sequence (timm InvertedResidual.forward, reached from model/feature_extractors.py:39-43). Unfused, the
// expanded tensor costs one HBM write + one HBM read of up to 963 MB per 200 frames per block; fused, HBM sees only the
// block input (Cin floats per pixel, + tile halo) and the depthwise output.
//
// Block = one TH x TW output tile of one frame, ALL expanded channels in chunks of 32:
//   LDS  Xs [PR][Cin+4]   input patch ((TH-1)S+K) x ((TW-1)S+K) pixels, loaded once (PR = pixels rounded up to 32)
//        Ws [32][Cin+4]   expand weights of the current channel chunk (torch [mid][Cin] layout, K contiguous)
//        Es [P][36]       expanded patch of the chunk (BN1 + SiLU applied; ZERO outside the image: the depthwise
//                         convolution zero-pads the EXPANDED tensor, not the input)
//        Ds [K*K][8] quads depthwise weights of the chunk
//   per chunk: MFMA phase (each wave owns patch row-tiles w, w+4: Cin/2 v_mfma_f32_32x32x2_f32 each) -> barrier ->
//   depthwise phase (thread = channel quad x NOUT output columns, taps via conflict-free ds_read_b128) -> barrier.
// Pooling partials: every thread keeps a fixed (chunk-relative) channel quad, sums its activated outputs, the block
// reduces across its 32 column segments in fixed order and writes pool_partial[b][tile][c] (deterministic).
#include "common.h"

namespace orbit {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using v4f = __attribute__((ext_vector_type(4))) float;

__device__ __forceinline__ float silu_f(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

struct MbParams {
    const float* x;    // [B][H][W][Cin] NHWC
    const float* w1;   // [mid][Cin]
    const float* sc1;  // [mid] folded BN1
    const float* sh1;
    const float* wdw;  // [K][K][mid]
    const float* sc2;  // [mid] folded BN2
    const float* sh2;
    float* y;          // [B][Ho][Wo][mid]
    float* pool;       // [B][tiles][mid] or nullptr
    int H, W, Cin, mid, pad_t, pad_l, Ho, Wo, tiles_x;
    // STEM form: the "expand" stage is the network stem - a 3x3 stride-2 convolution of the NCHW frames - instead of a
    // 1x1 conv of an NHWC tensor. x = frames [B][3][FH][FW]; (H, W) above is then the stem's output grid; Cin = 32 = the
    // 27 stem taps (ci, kh, kw) padded; w1 = [mid][32] with zero columns 27..31
    int FH, FW, spad_t, spad_l;
};

template <int K, int S, int TH, int TW, bool STEM = false>
__global__ __launch_bounds__(256) void mbconv_front_kernel(const MbParams p) {
    constexpr int IH = (TH - 1) * S + K, IW = (TW - 1) * S + K;
    constexpr int P = IH * IW;                      // patch pixels
    constexpr int PT = (P + 31) / 32;               // 32-row MFMA tiles over the patch
    constexpr int PR = PT * 32;
    constexpr int NOUT = TH * TW / 32;              // outputs per thread along W (32 segments x 8 channel quads)
    constexpr int SEG_PER_ROW = TW / NOUT;
    constexpr int NCOL = (NOUT - 1) * S + K;
    constexpr int ES = 36;                          // Es row stride (floats)
    static_assert(NOUT >= 1 && TW % NOUT == 0 && 32 / SEG_PER_ROW == TH, "tile/thread mapping");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int XS = p.Cin + 4;                       // Xs / Ws row stride (floats)
    float* Xs = smem;                               // [PR][XS]
    float* Ws = Xs + PR * XS;                       // [2][32][XS]   (chunk weights are double-buffered: the barrier that
    float* Es = Ws + 2 * 32 * XS;                   // [PR][ES]       used to close a chunk is gone, see the chunk loop)
    v4f* Ds = reinterpret_cast<v4f*>(Es + PR * ES); // [2][K*K][8]
    v4f* redw = Ds + 2 * K * K * 8;                 // [4 waves][64]  wave-private scratch of the pooling reduction
    v4f* pool_all = redw + 256;                     // [8 chunks][4 waves][8 quads]
    unsigned char* valid = reinterpret_cast<unsigned char*>(pool_all + 256);  // [PR] pixel-inside-image flags

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int b = blockIdx.y, t_idx = blockIdx.x;
    const int ty = t_idx / p.tiles_x, tx = t_idx - ty * p.tiles_x;
    const int hi0 = ty * TH * S - p.pad_t, wi0 = tx * TW * S - p.pad_l;
    const int cin4 = p.Cin >> 2;

    // ---- stage the input patch (zero outside the image / beyond P) and the pixel validity flags
    // (loads of a batch of 4 are all issued before the first LDS store: a load -> store round trip per element would
    // serialise PR * Cin / 1024 HBM latencies per block)
    if (STEM) {
        // im2col of the stem directly into Xs: thread = one tap column k of 8 patch pixels per pass; all 16 scalar loads of
        // a thread are issued (from clamped addresses) before the first LDS store
        const int k = tid & 31, ci = k / 9, kh = (k - ci * 9) / 3, kw = k - ci * 9 - kh * 3;
        const float* xf = p.x + ((size_t)b * 3 + (k < 27 ? ci : 0)) * p.FH * p.FW;
        constexpr int NU = PR / 8;
        float v[NU];
        unsigned ok = 0;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int px = (tid >> 5) + 8 * u;
            const int iy = px / IW, ix = px - iy * IW;
            const int hi = hi0 + iy, wi = wi0 + ix;           // stem output pixel
            const int r = hi * 2 - p.spad_t + kh, c = wi * 2 - p.spad_l + kw;
            const bool in = k < 27 && px < P && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W &&
                            (unsigned)r < (unsigned)p.FH && (unsigned)c < (unsigned)p.FW;
            v[u] = xf[in ? r * p.FW + c : 0];
            ok |= (in ? 1u : 0u) << u;
        }
#pragma unroll
        for (int u = 0; u < NU; ++u) Xs[((tid >> 5) + 8 * u) * XS + k] = ((ok >> u) & 1u) ? v[u] : 0.f;
    } else {
    const float* xb = p.x + (size_t)b * p.H * p.W * p.Cin;
    for (int i0 = tid; i0 < PR * cin4; i0 += 4 * 256) {
        v4f v[4];
        int dst[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * 256;
            v[u] = (v4f){0.f, 0.f, 0.f, 0.f};
            dst[u] = -1;
            if (i < PR * cin4) {
                const int px = i / cin4, c4 = i - px * cin4;
                dst[u] = px * XS + c4 * 4;
                if (px < P) {
                    const int iy = px / IW, ix = px - iy * IW;
                    const int hi = hi0 + iy, wi = wi0 + ix;
                    if ((unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W)
                        v[u] = *reinterpret_cast<const v4f*>(xb + ((size_t)hi * p.W + wi) * p.Cin + c4 * 4);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (dst[u] >= 0) *reinterpret_cast<v4f*>(Xs + dst[u]) = v[u];
    }
    }
    for (int px = tid; px < PR; px += 256) {
        bool ok = false;
        if (px < P) {
            const int iy = px / IW, ix = px - iy * IW;
            ok = (unsigned)(hi0 + iy) < (unsigned)p.H && (unsigned)(wi0 + ix) < (unsigned)p.W;
        }
        valid[px] = ok ? 1 : 0;
    }
    // block-uniform: the whole patch lies inside the image (51-61 % of the tiles at 56x56). The expand epilogue then needs
    // no validity flags at all - rows beyond P of the last row tile are never read by the depthwise phase
    const bool interior = hi0 >= 0 && wi0 >= 0 && hi0 + IH <= p.H && wi0 + IW <= p.W;

    // depthwise thread mapping: channel quad lc (8 per chunk), column segment seg (32 per tile)
    const int lc = tid & 7, seg = tid >> 3;
    const int oy = seg / SEG_PER_ROW, ox0 = (seg % SEG_PER_ROW) * NOUT;
    const int ho = ty * TH + oy;
    const int ngrp = p.Cin >> 3;
    const int tiles = gridDim.x;

    // chunk weights travel through registers one chunk ahead: requested at the top of a chunk, written to LDS at the top
    // of the next one, so their latency hides under the MFMA and depthwise phases (loaded in place at the top of every
    // chunk it was 1-2 us of exposed latency per chunk)
    constexpr int WR = 2;                       // expand-weight quads per thread: 32 rows x Cin/4 <= 512 (Cin <= 64)
    v4f wreg[WR], dreg = {0.f, 0.f, 0.f, 0.f};
    // ... and so do the folded BatchNorm vectors of the chunk. NO global load may be issued inside a chunk's phases: hipcc
    // waits for one with vmcnt(0), which also drains the prefetch issued just before it (that is what made every chunk pay
    // the full weight-load latency: rocprofv3 showed the waves parked in s_waitcnt / barriers 59 % of their cycles)
    float s1n = 0.f, h1n = 0.f;
    v4f s2n = {0.f, 0.f, 0.f, 0.f}, h2n = {0.f, 0.f, 0.f, 0.f};
    auto load_weights = [&](int c0) {
        {
            const int ch = c0 + (tid & 31);
            s1n = ch < p.mid ? p.sc1[ch] : 0.f, h1n = ch < p.mid ? p.sh1[ch] : 0.f;
            const int cq = c0 + (tid & 7) * 4;
            s2n = (v4f){0.f, 0.f, 0.f, 0.f}, h2n = (v4f){0.f, 0.f, 0.f, 0.f};
            if (cq < p.mid) s2n = *reinterpret_cast<const v4f*>(p.sc2 + cq), h2n = *reinterpret_cast<const v4f*>(p.sh2 + cq);
        }
#pragma unroll
        for (int u = 0; u < WR; ++u) {
            const int i = tid + u * 256;
            wreg[u] = (v4f){0.f, 0.f, 0.f, 0.f};
            if (i < 32 * cin4) {
                const int r = i / cin4, c4 = i - r * cin4;
                if (c0 + r < p.mid) wreg[u] = *reinterpret_cast<const v4f*>(p.w1 + (size_t)(c0 + r) * p.Cin + c4 * 4);
            }
        }
        dreg = (v4f){0.f, 0.f, 0.f, 0.f};
        if (tid < K * K * 8) {
            const int tap = tid >> 3, q = tid & 7;
            if (c0 + q * 4 < p.mid) dreg = *reinterpret_cast<const v4f*>(p.wdw + (size_t)tap * p.mid + c0 + q * 4);
        }
    };
    load_weights(0);

    // Two barriers per chunk: (B1) weights + previous chunk's Es readers, (B2) Es complete. Ws / Ds alternate between two
    // buffers, so a wave that runs ahead into chunk c+1 writes the other pair while slower waves still read chunk c's
    // (it cannot reach chunk c+2 before everyone passed B1 of c+1); the pooling partials are reduced inside each wave and
    // parked per chunk, the cross-wave sum happens once after the loop.
    for (int c0 = 0; c0 < p.mid; c0 += 32) {
        const int ci = c0 >> 5;
        float* Wc = Ws + (ci & 1) * 32 * XS;
        v4f* Dc = Ds + (ci & 1) * K * K * 8;
        // ---- chunk weights: expand rows c0..c0+31 (zero rows past mid), depthwise taps
#pragma unroll
        for (int u = 0; u < WR; ++u) {
            const int i = tid + u * 256;
            if (i < 32 * cin4) {
                const int r = i / cin4, c4 = i - r * cin4;
                *reinterpret_cast<v4f*>(Wc + r * XS + c4 * 4) = wreg[u];
            }
        }
        if (tid < K * K * 8) Dc[tid] = dreg;
        const float s1 = s1n, h1 = h1n;
        const v4f s2 = s2n, h2 = h2n;
        __syncthreads();  // Xs (first chunk), Ws, Ds visible; previous chunk's readers of Es are done
        if (c0 + 32 < p.mid) load_weights(c0 + 32);

        // ---- expand: E[patch rows][32 ch] = Xs . Ws^T on the fp32 matrix cores; BN1 + SiLU; zero outside the image
        for (int t = wave; t < PT; t += 4) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const float* A = Xs + (t * 32 + l31) * XS + lh * 4;
            const float* Bq = Wc + l31 * XS + lh * 4;
            for (int g = 0; g < ngrp; ++g) {
                const v4f af = *reinterpret_cast<const v4f*>(A + g * 8);
                const v4f bf = *reinterpret_cast<const v4f*>(Bq + g * 8);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk], bf[kk], acc, 0, 0, 0);
            }
            // C/D layout: col = lane & 31 (channel), row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) (patch pixel)
            // branch-free: SiLU on every element, zeroed through a select (a per-element branch on valid[] costs an exec-mask
            // branch and a wait each); the four flags of four consecutive patch pixels are one 32-bit LDS read
            if (interior) {
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const int px0 = t * 32 + 8 * rq + 4 * lh;
#pragma unroll
                    for (int j = 0; j < 4; ++j) Es[(px0 + j) * ES + l31] = silu_f(acc[rq * 4 + j] * s1 + h1);
                }
            } else {
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const int px0 = t * 32 + 8 * rq + 4 * lh;
                    const unsigned vb = *reinterpret_cast<const unsigned*>(valid + px0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float e = silu_f(acc[rq * 4 + j] * s1 + h1);
                        Es[(px0 + j) * ES + l31] = ((vb >> (8 * j)) & 0xffu) ? e : 0.f;
                    }
                }
            }
        }
        __syncthreads();

        // ---- depthwise from LDS + BN2 + SiLU -> HBM; pooling partial
        const int cq = c0 + lc * 4;
        const bool q_ok = cq < p.mid;
        v4f acc2[NOUT];
#pragma unroll
        for (int j = 0; j < NOUT; ++j) acc2[j] = (v4f){0.f, 0.f, 0.f, 0.f};
        {
            const float* erow = Es + ((oy * S) * IW + ox0 * S) * ES + lc * 4;
            const v4f* dk = Dc + lc;
            // a rolled loop over the tap rows (unrolled, hipcc hoists all K*(NCOL+K) LDS reads: 256 VGPRs at K = 5)
#pragma unroll 1
            for (int kh = 0; kh < K; ++kh) {
                v4f col[NCOL];
#pragma unroll
                for (int q = 0; q < NCOL; ++q) col[q] = *reinterpret_cast<const v4f*>(erow + q * ES);
#pragma unroll
                for (int kw = 0; kw < K; ++kw) {
                    const v4f f = dk[kw * 8];
#pragma unroll
                    for (int j = 0; j < NOUT; ++j) acc2[j] += col[j * S + kw] * f;
                }
                erow += IW * ES;
                dk += K * 8;
            }
        }
        v4f psum = {0.f, 0.f, 0.f, 0.f};
        if (q_ok) {
#pragma unroll
            for (int j = 0; j < NOUT; ++j) {
                const int wo = tx * TW + ox0 + j;
                if (ho < p.Ho && wo < p.Wo) {
                    v4f o = acc2[j] * s2 + h2;
                    o[0] = silu_f(o[0]), o[1] = silu_f(o[1]), o[2] = silu_f(o[2]), o[3] = silu_f(o[3]);
                    *reinterpret_cast<v4f*>(p.y + (((size_t)b * p.Ho + ho) * p.Wo + wo) * p.mid + cq) = o;
                    psum += o;
                }
            }
        }
        if (p.pool) {  // this wave's 8 column segments per channel quad, summed in segment order by lanes 0-7
            redw[wave * 64 + lane] = psum;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // LDS is in-order per wave: the stores above are visible
            if (lane < 8) {
                v4f t = redw[wave * 64 + lane];
#pragma unroll
                for (int sg = 1; sg < 8; ++sg) t += redw[wave * 64 + sg * 8 + lane];
                pool_all[(ci * 4 + wave) * 8 + lane] = t;
            }
        }
    }
    if (p.pool) {
        __syncthreads();
        const int nchunk = (p.mid + 31) >> 5;
        for (int i = tid; i < nchunk * 8; i += 256) {
            const int ci = i >> 3, q = i & 7;
            if (ci * 32 + q * 4 < p.mid) {
                v4f t = pool_all[(ci * 4 + 0) * 8 + q];
#pragma unroll
                for (int w2 = 1; w2 < 4; ++w2) t += pool_all[(ci * 4 + w2) * 8 + q];
                *reinterpret_cast<v4f*>(p.pool + ((size_t)b * tiles + t_idx) * p.mid + ci * 32 + q * 4) = t;
            }
        }
    }
}

// (4 x 16 output tiles for stride 2 - better balanced MFMA row tiles, less halo - were measured: 15-40 % SLOWER; the
// kernel lives on the number of co-resident blocks, not on per-block efficiency. 3 x 8 tiles for 3x3 stride 2 - four
// exactly filled MFMA row tiles, one per wave, instead of five - were 4 % slower at 112x112 and 20 % faster at 28x28.)
static void mb_tile_geom(int stride, int& th, int& tw) { th = stride == 1 ? 8 : 4, tw = 8; }

int mbconv_front_tiles(int Ho, int Wo, int stride) {
    int th, tw;
    mb_tile_geom(stride, th, tw);
    return cdiv(Ho, th) * cdiv(Wo, tw);
}

// can this (Cin, K, stride) be served by the fused kernel? (LDS budget: two blocks per CU)
bool mbconv_front_supported(int Cin, int mid, int K, int stride) {
    return Cin % 8 == 0 && Cin <= 40 && mid % 4 == 0 && mid <= 256 && (K == 3 || K == 5) && (stride == 1 || stride == 2);
}

int launch_mbconv_front(const float* x, const float* w1, const float* sc1, const float* sh1, const float* wdw,
                        const float* sc2, const float* sh2, float* y, float* pool, int B, int H, int W, int Cin, int mid,
                        int K, int stride, int pad_t, int pad_l, int Ho, int Wo, hipStream_t s) {
    ORBIT_REQUIRE(x && w1 && sc1 && sh1 && wdw && sc2 && sh2 && y, "mbconv_front: null pointer");
    ORBIT_REQUIRE(mbconv_front_supported(Cin, mid, K, stride), "mbconv_front: unsupported shape (Cin=%d mid=%d K=%d s=%d)",
                  Cin, mid, K, stride);
    int th, tw;
    mb_tile_geom(stride, th, tw);
    MbParams p;
    p.x = x, p.w1 = w1, p.sc1 = sc1, p.sh1 = sh1, p.wdw = wdw, p.sc2 = sc2, p.sh2 = sh2, p.y = y, p.pool = pool;
    p.H = H, p.W = W, p.Cin = Cin, p.mid = mid, p.pad_t = pad_t, p.pad_l = pad_l, p.Ho = Ho, p.Wo = Wo;
    p.tiles_x = cdiv(Wo, tw);
    const int tiles = p.tiles_x * cdiv(Ho, th);
    const int ih = (th - 1) * stride + K, iw = (tw - 1) * stride + K;
    const int pr = (ih * iw + 31) / 32 * 32;
    ORBIT_REQUIRE(mid <= 256, "mbconv_front: at most 8 chunks of 32 expanded channels (mid=%d)", mid);
    const size_t lds = ((size_t)pr * (Cin + 4) + 2 * 32 * (Cin + 4) + (size_t)pr * 36) * sizeof(float) +
                       (size_t)(2 * K * K * 8 + 512) * 16 + pr;
    dim3 grid(tiles, B);
#define ORBIT_MB(KK, SS, TH_, TW_)                                                                  \
    do {                                                                                            \
        auto kern = mbconv_front_kernel<KK, SS, TH_, TW_>;                                          \
        static bool attr_set = false;                                                               \
        if (!attr_set && lds > 64 * 1024) {                                                         \
            ORBIT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            attr_set = true;                                                                        \
        }                                                                                           \
        kern<<<grid, 256, lds, s>>>(p);                                                             \
    } while (0)
    if (K == 3 && stride == 1) ORBIT_MB(3, 1, 8, 8);
    else if (K == 3) ORBIT_MB(3, 2, 4, 8);
    else if (stride == 1) ORBIT_MB(5, 1, 8, 8);
    else ORBIT_MB(5, 2, 4, 8);
#undef ORBIT_MB
    ORBIT_LAUNCH_CHECK();
    return ORBIT_OK;
}

// ---- stem form: conv_stem (3 -> mid channels, 3x3 stride 2, TF-SAME) + BN + SiLU + depthwise 3x3/1 + BN + SiLU -----------
// OIHW [mid][3][3][3] = [mid][27] -> [mid][32] (zero columns 27..31): the "expand weights" of the stem form
__global__ __launch_bounds__(256) void stem_pack_kernel(const float* __restrict__ w, float* __restrict__ wp, int mid) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < mid * 32; i += gridDim.x * 256) {
        const int r = i >> 5, k = i & 31;
        wp[i] = k < 27 ? w[r * 27 + k] : 0.f;
    }
}
int stem_pack_weights(const float* w_oihw, float* w_packed, int mid, hipStream_t s) {
    stem_pack_kernel<<<cdiv(mid * 32, 256), 256, 0, s>>>(w_oihw, w_packed, mid);
    ORBIT_LAUNCH_CHECK();
    return ORBIT_OK;
}

bool stem_dw_front_supported(int mid, int K, int stride) { return mid % 4 == 0 && mid <= 256 && K == 3 && stride == 1; }

int launch_stem_dw_front(const float* frames, const float* w1_packed, const float* sc1, const float* sh1,
                         const float* wdw, const float* sc2, const float* sh2, float* y, float* pool, int B, int FH, int FW,
                         int spad_t, int spad_l, int H, int W, int mid, int K, int pad_t, int pad_l, int Ho, int Wo,
                         hipStream_t s) {
    ORBIT_REQUIRE(frames && w1_packed && sc1 && sh1 && wdw && sc2 && sh2 && y, "stem_dw_front: null pointer");
    ORBIT_REQUIRE(stem_dw_front_supported(mid, K, 1), "stem_dw_front: unsupported shape (mid=%d K=%d)", mid, K);
    int th, tw;
    mb_tile_geom(1, th, tw);
    MbParams p;
    p.x = frames, p.w1 = w1_packed, p.sc1 = sc1, p.sh1 = sh1, p.wdw = wdw, p.sc2 = sc2, p.sh2 = sh2, p.y = y, p.pool = pool;
    p.H = H, p.W = W, p.Cin = 32, p.mid = mid, p.pad_t = pad_t, p.pad_l = pad_l, p.Ho = Ho, p.Wo = Wo;
    p.FH = FH, p.FW = FW, p.spad_t = spad_t, p.spad_l = spad_l;
    p.tiles_x = cdiv(Wo, tw);
    const int tiles = p.tiles_x * cdiv(Ho, th);
    const int ih = th - 1 + K, iw = tw - 1 + K;
    const int pr = (ih * iw + 31) / 32 * 32;
    const size_t lds = ((size_t)pr * 36 + 2 * 32 * 36 + (size_t)pr * 36) * sizeof(float) +
                       (size_t)(2 * K * K * 8 + 512) * 16 + pr;
    mbconv_front_kernel<3, 1, 8, 8, true><<<dim3(tiles, B), 256, lds, s>>>(p);
    ORBIT_LAUNCH_CHECK();
    return ORBIT_OK;
}

}  // namespace orbit

using namespace orbit;

// pooling partials per frame of orbit_op_stem_dw_front under the current options
extern "C" int orbit_op_stem_dw_front_partials(int H, int W, int mid) {
    if (get_option("mbconv_rows") && get_option("stem_rows") && stem_rows_supported(H, W, mid, 3, 1)) return stem_rows_tiles(H, W);
    return mbconv_front_tiles(H, W, 1);
}

// single-operator entry for the parity tests: frames NCHW, w_stem torch [mid][3][3][3], wdw torch [mid][1][3][3]
extern "C" int orbit_op_stem_dw_front(const float* frames, const float* w_stem, const float* scale1, const float* shift1,
                                      const float* wdw, const float* scale2, const float* shift2, float* y,
                                      float* pool_partial, int B, int FH, int FW, int spad_top, int spad_left, int H, int W,
                                      int mid, int pad_top, int pad_left, int Ho, int Wo, orbit_stream_t stream) {
    ORBIT_REQUIRE(frames && w_stem && wdw && y, "op_stem_dw_front: null pointer");
    hipStream_t s = (hipStream_t)stream;
    float* wp = nullptr;
    ORBIT_HIP_CHECK(hipMallocAsync(reinterpret_cast<void**>(&wp), (size_t)mid * (32 + 9) * sizeof(float), s));
    int rc = stem_pack_weights(w_stem, wp, mid, s);
    if (rc == ORBIT_OK) rc = dwconv_pack_weights(wdw, wp + (size_t)mid * 32, mid, 3, s);
    if (rc == ORBIT_OK && get_option("mbconv_rows") && get_option("stem_rows") && stem_rows_supported(H, W, mid, 3, 1))
        rc = launch_stem_rows(frames, wp, scale1, shift1, wp + (size_t)mid * 32, scale2, shift2, y, pool_partial, B, FH, FW,
                              spad_top, spad_left, H, W, s);
    else if (rc == ORBIT_OK)
        rc = launch_stem_dw_front(frames, wp, scale1, shift1, wp + (size_t)mid * 32, scale2, shift2, y, pool_partial, B, FH,
                                  FW, spad_top, spad_left, H, W, mid, 3, pad_top, pad_left, Ho, Wo, s);
    (void)hipFreeAsync(wp, s);
    return rc;
}

// pooling partials per frame of orbit_op_mbconv_front under the current options (the kernel selection below)
extern "C" int orbit_op_mbconv_front_partials(int H, int W, int Cin, int mid, int K, int stride) {
    if (get_option("mbconv_map") && mbconv_map_supported(H, W, Cin, mid, K, stride)) return 1;
    if (get_option("mbconv_rows") && mbconv_rows_supported(H, W, Cin, mid, K, stride))
        return mbconv_rows_tiles(H, W, Cin, mid, K, stride);
    return mbconv_front_tiles(cdiv(H, stride), cdiv(W, stride), stride);
}

// single-operator entry for the parity tests: w1 torch [mid][Cin][1][1], wdw torch [mid][1][K][K]
extern "C" int orbit_op_mbconv_front(const float* x, const float* w1, const float* scale1, const float* shift1,
                                     const float* wdw, const float* scale2, const float* shift2, float* y,
                                     float* pool_partial, int B, int H, int W, int Cin, int mid, int K, int stride,
                                     int pad_top, int pad_left, int Ho, int Wo, orbit_stream_t stream) {
    ORBIT_REQUIRE(x && w1 && wdw && y, "op_mbconv_front: null pointer");
    hipStream_t s = (hipStream_t)stream;
    float* wp = nullptr;
    ORBIT_HIP_CHECK(hipMallocAsync(reinterpret_cast<void**>(&wp), (size_t)mid * K * K * sizeof(float), s));
    int rc = dwconv_pack_weights(wdw, wp, mid, K, s);
    if (rc == ORBIT_OK) {
        // the whole-map kernel (csrc/mbconv_map.hip) where it applies - as the extractor plans choose; its pool_partial is
        // [B][1][mid] instead of [B][tiles][mid] (mbconv_map option 0 = tiled kernel everywhere it is supported)
        if (get_option("mbconv_map") && mbconv_map_supported(H, W, Cin, mid, K, stride))
            rc = launch_mbconv_map(x, w1, scale1, shift1, wp, scale2, shift2, y, pool_partial, B, H, W, Cin, mid, K, stride, s);
        else if (get_option("mbconv_rows") && mbconv_rows_supported(H, W, Cin, mid, K, stride))  // pool: [B][rows tiles][mid]
            rc = launch_mbconv_rows(x, w1, scale1, shift1, wp, scale2, shift2, y, pool_partial, B, H, W, Cin, mid, K, stride,
                                    pad_top, pad_left, Ho, Wo, s);
        else
            rc = launch_mbconv_front(x, w1, scale1, shift1, wp, scale2, shift2, y, pool_partial, B, H, W, Cin, mid, K,
                                     stride, pad_top, pad_left, Ho, Wo, s);
    }
    (void)hipFreeAsync(wp, s);
    return rc;
}
