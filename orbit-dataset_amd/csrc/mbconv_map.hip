// Whole-map form of the fused MBConv front half for EfficientNet-B0's 14x14 and 7x7 stages (gfx950):
// expand 1x1 conv (fp32 MFMA) -> BN1 -> SiLU -> depthwise KxK (TF-SAME) -> BN2 -> SiLU -> squeeze-excite pool sums in ONE
// kernel, the 6x-expanded tensor living only in LDS.
//
// Replaces, for timm `InvertedResidual` blocks 3.1-6.0 of tf_efficientnet_b0 (reached from the reference's
// model/feature_extractors.py:39-43), the conv_pw -> bn1 -> act -> conv_dw -> bn2 -> act -> se.mean sequence. Unfused
// these are three launches (MFMA expand conv, LDS depthwise kernel) that write and re-read a 45-105 MB expanded tensor per
// 200 frames; the tiled fused kernel of csrc/mbconv.hip loses there (8x8 tiles pay a 60-125 % halo on 5x5 taps).
//
// Here a block owns the WHOLE spatial map of FB frames (FB = 1 at 14x14, 2 at 7x7) and a run of 32-channel chunks of the
// expanded tensor, so there is no halo at all: the depthwise zero padding is a border of the LDS tile that is zeroed once.
//   * the block input never touches LDS: wave w keeps the A-operand fragments of its 32-pixel row tiles (w, w+4) in
//     REGISTERS for the whole block (CIN/2 floats per lane and tile), loaded once from HBM/L2 in MFMA operand order
//     (lane (i, h) holds channels 8g+4h..+3 of pixel i) - the expand GEMM re-uses them for every chunk;
//   * per chunk: W1 rows of the chunk -> LDS (prefetched through registers one chunk ahead), MFMA expand straight from
//     the register fragments, BN1 + SiLU, scatter into the padded map tile Es; barrier; depthwise from Es (ds_read_b128,
//     NOUT outputs along a row per thread) + BN2 + SiLU -> HBM, pool sums; barrier.
//   * two blocks per CU (<= 78 KB of LDS, <= 256 VGPRs): one block's MFMA phase can run beside the other's VALU phase.
// The expand uses the same k order as conv_igemm (groups of 8, kk ascending) and the depthwise the same tap order as the
// unfused kernels; pool sums are complete per (frame, channel) (pool_partial [B][1][mid]).
#include "common.h"

namespace orbit {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using v4f = __attribute__((ext_vector_type(4))) float;
using v4i = __attribute__((ext_vector_type(4))) int;

__device__ __forceinline__ float silu_m(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

struct MbMapParams {
    const float* x;    // [B][HW][HW][CIN] NHWC
    const float* w1;   // [mid][CIN]
    const float* sc1;  // [mid] folded BN1
    const float* sh1;
    const float* wdw;  // [K][K][mid]
    const float* sc2;  // [mid] folded BN2
    const float* sh2;
    float* y;          // [B][HO][HO][mid]
    float* pool;       // [B][mid] or nullptr
    int B, mid, G;     // G = most chunks (of 32 expanded channels) a block owns: chunks are dealt evenly to gridDim.x groups
};

template <int CIN, int K, int S, int HW, int FB>
struct MbMapGeom {
    static constexpr int HO = (HW + S - 1) / S;
    static constexpr int PADT = ((HO - 1) * S + K - HW) > 0 ? ((HO - 1) * S + K - HW) : 0;  // TF-SAME total padding
    static constexpr int PT_ = PADT / 2;                 // top / left (the extra row goes to the bottom / right)
    static constexpr int EH = HW + PADT;                 // padded map edge
    static constexpr int NPIX = HW * HW;
    static constexpr int P = FB * NPIX;
    static constexpr int TILES = (P + 31) / 32;
    static constexpr int TPW = (TILES + 3) / 4;          // row tiles per wave
    static constexpr int NG = CIN / 8;
    static constexpr int ES = 36;
    static constexpr int XS = CIN + 4;
    static constexpr int RO = HO > 8 ? 16 : 8;           // output rows rounded up to a power of two
    static constexpr int SPR = 32 / (FB * RO);           // thread segments per output row
    static constexpr int NOUT = (HO + SPR - 1) / SPR;    // outputs per thread along a row
    static constexpr int NCOL = (NOUT - 1) * S + K;
    static constexpr int WR = (8 * CIN + 255) / 256;     // expand-weight quads per thread and chunk
    static constexpr int ES_ROWS = FB * EH * EH + 1;     // + one dump row for the pixels that pad the last row tile
    static size_t lds_bytes(int G) {
        return ((size_t)ES_ROWS * ES + 32 * XS) * sizeof(float) + (size_t)(2 * K * K * 8 + 256 + G * 32) * sizeof(v4f) +
               (size_t)TILES * 32 * sizeof(int);
    }
};

// 256 threads, two blocks per CU (<= 78 KB of LDS, <= 256 VGPRs). Per chunk: every wave expands its row tiles on the
// matrix cores from its register-resident input fragments (B operand = the chunk's W1 rows in LDS), BN1 + SiLU, scatter
// into the padded map tile Es; barrier; depthwise from Es (ds_read_b128 gather over the tap rows, NOUT outputs along a
// row per thread) + BN2 + SiLU -> HBM, pool sums; barrier. This is the fastest of the three structures that were built
// (profiles/r02_mbconv_map.txt): 8-wave producer / consumer blocks - with the same float4 gather, or with a
// channel-major map walked row by row (3.6x fewer LDS bytes) - measured 5-15 % slower, because one 8-wave block per CU
// pays its prologue and a pipeline-fill iteration per 3-7 chunks and the period is set by the slower role anyway.
template <int CIN, int K, int S, int HW, int FB>
__global__ __launch_bounds__(256, 2) void mbconv_map_kernel(const MbMapParams p) {
    using Gm = MbMapGeom<CIN, K, S, HW, FB>;
    constexpr int HO = Gm::HO, EH = Gm::EH, NPIX = Gm::NPIX, P = Gm::P, TILES = Gm::TILES, TPW = Gm::TPW, NG = Gm::NG;
    constexpr int ES = Gm::ES, XS = Gm::XS, SPR = Gm::SPR, NOUT = Gm::NOUT, NCOL = Gm::NCOL, WR = Gm::WR;
    constexpr int PAD = Gm::PT_;
    static_assert(CIN % 8 == 0 && TPW <= 2 && 32 % (FB * Gm::RO) == 0 && SPR >= 1, "geometry");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Es = smem;                                              // [ES_ROWS][ES] padded expanded map(s) of one chunk
    float* Ws = Es + Gm::ES_ROWS * ES;                             // [32][XS] expand weights of the chunk
    v4f* Ds = reinterpret_cast<v4f*>(Ws + 32 * XS);                // [2][K*K][8] depthwise taps (double-buffered)
    v4f* redw = Ds + 2 * K * K * 8;                                // [4 waves][64] pooling scratch
    v4f* pool_all = redw + 256;                                    // [G][4 waves][8 quads]
    int* estab = reinterpret_cast<int*>(pool_all + p.G * 32);      // [TILES*32] pixel -> Es float offset

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int b0 = blockIdx.y * FB;
    const int nchunks = p.mid >> 5;
    const int cfirst = (int)((long)blockIdx.x * nchunks / gridDim.x);        // balanced partition: group sizes differ by
    const int cend = (int)((long)(blockIdx.x + 1) * nchunks / gridDim.x);    // at most one chunk
    if (cfirst >= cend) return;

    // ---- A-operand fragments of this wave's row tiles: registers for the whole block --------------------------------
    v4f xr[TPW][NG];
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int px = (wave + 4 * i) * 32 + l31;
        const int f = px / NPIX, pp = px - f * NPIX;
        const bool ok = (wave + 4 * i) < TILES && px < P && b0 + f < p.B;
        const float* src = p.x + ((size_t)(b0 + (ok ? f : 0)) * NPIX + (ok ? pp : 0)) * CIN + lh * 4;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const v4f v = *reinterpret_cast<const v4f*>(src + g * 8);  // unconditional load from a safe address
            xr[i][g] = ok ? v : (v4f){0.f, 0.f, 0.f, 0.f};
        }
    }

    // ---- zero the padded map (the border stays zero for every chunk), build the pixel -> Es offset table --------------
    for (int i = tid; i < Gm::ES_ROWS * ES / 4; i += 256) reinterpret_cast<v4f*>(Es)[i] = (v4f){0.f, 0.f, 0.f, 0.f};
    for (int px = tid; px < TILES * 32; px += 256) {
        int row = FB * EH * EH;  // dump row
        if (px < P) {
            const int f = px / NPIX, pp = px - f * NPIX;
            const int yy = pp / HW, xx = pp - yy * HW;
            row = (f * EH + yy + PAD) * EH + xx + PAD;
        }
        estab[px] = row * ES;
    }

    // ---- depthwise thread mapping: channel quad lc, segment seg = (frame, output row, row part) -----------------------
    const int lc = tid & 7, seg = tid >> 3;
    const int fseg = seg / (32 / FB), sl = seg % (32 / FB);
    const int oy = sl / SPR, ox0 = (sl % SPR) * NOUT;
    const bool seg_ok = oy < HO && b0 + fseg < p.B;
    const int oyc = oy < HO ? oy : HO - 1;  // rows beyond the map compute on a valid address, nothing is stored

    // chunk parameters travel through registers one chunk ahead (no global load inside a phase that consumes it)
    v4f wreg[WR], dreg = {0.f, 0.f, 0.f, 0.f};
    float s1n = 0.f, h1n = 0.f;
    v4f s2n = {0.f, 0.f, 0.f, 0.f}, h2n = {0.f, 0.f, 0.f, 0.f};
    auto load_weights = [&](int c0) {
        s1n = p.sc1[c0 + l31], h1n = p.sh1[c0 + l31];
        s2n = *reinterpret_cast<const v4f*>(p.sc2 + c0 + lc * 4), h2n = *reinterpret_cast<const v4f*>(p.sh2 + c0 + lc * 4);
#pragma unroll
        for (int u = 0; u < WR; ++u) {
            const int i = tid + u * 256;
            const int r = i / (CIN / 4), c4 = i - r * (CIN / 4);
            wreg[u] = (v4f){0.f, 0.f, 0.f, 0.f};
            if (i < 8 * CIN) wreg[u] = *reinterpret_cast<const v4f*>(p.w1 + (size_t)(c0 + r) * CIN + c4 * 4);
        }
        if (tid < K * K * 8) dreg = *reinterpret_cast<const v4f*>(p.wdw + (size_t)(tid >> 3) * p.mid + c0 + (tid & 7) * 4);
    };
    auto store_w = [&]() {
#pragma unroll
        for (int u = 0; u < WR; ++u) {
            const int i = tid + u * 256;
            const int r = i / (CIN / 4), c4 = i - r * (CIN / 4);
            if (i < 8 * CIN) *reinterpret_cast<v4f*>(Ws + r * XS + c4 * 4) = wreg[u];
        }
    };
    load_weights(cfirst * 32);
    store_w();

    for (int c = cfirst; c < cend; ++c) {
        const int ci = c - cfirst, c0 = c * 32;
        v4f* Dc = Ds + (ci & 1) * K * K * 8;
        if (tid < K * K * 8) Dc[tid] = dreg;
        const float s1 = s1n, h1 = h1n;
        const v4f s2 = s2n, h2 = h2n;
        __syncthreads();  // B1: Ws / Ds of this chunk visible, Es zeroed (first chunk), previous chunk's readers of Es done
        if (c + 1 < cend) load_weights(c0 + 32);

        // ---- expand: E[pixels][32 channels] = X . W1^T, A from registers, B from LDS; BN1 + SiLU -> Es ----------------
        {
            f32x16 acc[TPW];
#pragma unroll
            for (int i = 0; i < TPW; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
            const float* Bq = Ws + l31 * XS + lh * 4;
            const bool two = TPW == 2 && wave + 4 < TILES;  // wave-uniform
            if (two) {
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const v4f bf = *reinterpret_cast<const v4f*>(Bq + g * 8);
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(xr[0][g][kk], bf[kk], acc[0], 0, 0, 0);
                        acc[TPW - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(xr[TPW - 1][g][kk], bf[kk], acc[TPW - 1], 0, 0, 0);
                    }
                }
            } else {
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const v4f bf = *reinterpret_cast<const v4f*>(Bq + g * 8);
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk)
                        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(xr[0][g][kk], bf[kk], acc[0], 0, 0, 0);
                }
            }
            // C/D layout: col = lane & 31 (channel), row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) (pixel of the tile)
#pragma unroll
            for (int i = 0; i < TPW; ++i) {
                if (i == 1 && !two) break;
                const int t = wave + 4 * i;
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const v4i er = *reinterpret_cast<const v4i*>(estab + t * 32 + 8 * rq + 4 * lh);
#pragma unroll
                    for (int j = 0; j < 4; ++j) Es[er[j] + l31] = silu_m(acc[i][rq * 4 + j] * s1 + h1);
                }
            }
        }
        __syncthreads();  // B2: Es complete; nobody reads Ws any more
        if (c + 1 < cend) store_w();  // next chunk's expand weights (made visible by B1 of the next iteration)

        // ---- depthwise from LDS + BN2 + SiLU -> HBM; pool sums --------------------------------------------------------
        v4f acc2[NOUT];
#pragma unroll
        for (int j = 0; j < NOUT; ++j) acc2[j] = (v4f){0.f, 0.f, 0.f, 0.f};
        {
            const float* erow = Es + ((fseg * EH + oyc * S) * EH + ox0 * S) * ES + lc * 4;
            const v4f* dk = Dc + lc;
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
                erow += EH * ES;
                dk += K * 8;
            }
        }
        v4f psum = {0.f, 0.f, 0.f, 0.f};
        if (seg_ok) {
            float* yrow = p.y + (((size_t)(b0 + fseg) * HO + oy) * HO + ox0) * p.mid + c0 + lc * 4;
#pragma unroll
            for (int j = 0; j < NOUT; ++j) {
                if (ox0 + j < HO) {
                    v4f o = acc2[j] * s2 + h2;
                    o[0] = silu_m(o[0]), o[1] = silu_m(o[1]), o[2] = silu_m(o[2]), o[3] = silu_m(o[3]);
                    *reinterpret_cast<v4f*>(yrow + (size_t)j * p.mid) = o;
                    psum += o;
                }
            }
        }
        if (p.pool) {  // this wave's 8 segments per channel quad, summed in segment order by lanes 0-7
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
        constexpr int WPF = 4 / FB;  // waves per frame
        const int nloc = cend - cfirst;
        for (int i = tid; i < nloc * 8 * FB; i += 256) {
            const int q = i & 7, ci = (i >> 3) % nloc, f = (i >> 3) / nloc;
            if (b0 + f < p.B) {
                v4f t = pool_all[(ci * 4 + f * WPF) * 8 + q];
#pragma unroll
                for (int w2 = 1; w2 < WPF; ++w2) t += pool_all[(ci * 4 + f * WPF + w2) * 8 + q];
                *reinterpret_cast<v4f*>(p.pool + (size_t)(b0 + f) * p.mid + (cfirst + ci) * 32 + q * 4) = t;
            }
        }
    }
}

// the shapes where the whole-map kernel measured FASTER than the conv + depthwise pair (tools/mb_bench.py, 200 frames:
// -20 % 80->480 3x3, -16 % 80->480 5x5, -17 % 112->672 5x5/2, -6..-7 % at 7x7; 112->672 5x5/1 ties): mbconv_map = 1
bool mbconv_map_preferred(int H, int W, int Cin, int mid, int K, int stride) {
    return mbconv_map_supported(H, W, Cin, mid, K, stride) && !(Cin == 112 && stride == 1);
}

// shapes served by the whole-map kernel: square 14x14 / 7x7 inputs of EfficientNet-B0's stages 3-6
bool mbconv_map_supported(int H, int W, int Cin, int mid, int K, int stride) {
    if (H != W || mid % 32 != 0 || mid != 6 * Cin) return false;
    if (H == 14 && stride == 1) return (Cin == 80 && (K == 3 || K == 5)) || (Cin == 112 && K == 5);
    if (H == 14 && stride == 2) return Cin == 112 && K == 5;
    if (H == 7 && stride == 1) return Cin == 192 && (K == 3 || K == 5);
    return false;
}

template <int CIN, int K, int S, int HW, int FB>
static int launch_map(MbMapParams& p, hipStream_t s) {
    using Gm = MbMapGeom<CIN, K, S, HW, FB>;
    const int nchunks = p.mid / 32;
    const int frames = cdiv(p.B, FB);
    // Chunk groups per frame (pair): blocks = frames x groups. A block pays a prologue (register-resident input, zeroed map
    // tile: about half a chunk's time) and the chip runs 512 blocks at a time (2 per CU), so the cost of a grouping is
    // rounds(blocks / 512) x (0.5 + chunks per block); the cheapest wins (ties: fewer groups). mbmap_groups overrides.
    int groups = get_option("mbmap_groups");
    if (groups <= 0) {
        double best = 1e30;
        for (int g = 1; g <= nchunks; ++g) {
            if (cdiv(nchunks, g) > 16) continue;  // pool scratch holds 16 chunks per block
            const double rounds = (double)cdiv(frames * g, 512);
            const double cost = rounds * (0.5 + cdiv(nchunks, g));
            if (cost < best - 1e-9) best = cost, groups = g;
        }
    }
    if (groups > nchunks) groups = nchunks;
    if (cdiv(nchunks, groups) > 16) groups = cdiv(nchunks, 16);
    p.G = cdiv(nchunks, groups);
    const size_t lds = Gm::lds_bytes(p.G);
    auto kern = mbconv_map_kernel<CIN, K, S, HW, FB>;
    static bool attr_set = false;
    if (!attr_set) {
        ORBIT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int)Gm::lds_bytes(16)));
        attr_set = true;
    }
    ORBIT_REQUIRE(p.G <= 16, "mbconv_map: at most 16 chunks per block (G=%d)", p.G);
    const double pix = (double)p.B * HW * HW, opix = (double)p.B * Gm::HO * Gm::HO;
    const int rec = prof_start("mbconv_map", 2.0 * pix * p.mid * CIN, 4.0 * (pix * CIN + opix * p.mid), s);
    kern<<<dim3(groups, frames), 256, lds, s>>>(p);
    prof_stop(rec, s);
    ORBIT_LAUNCH_CHECK();
    return ORBIT_OK;
}

int launch_mbconv_map(const float* x, const float* w1, const float* sc1, const float* sh1, const float* wdw,
                      const float* sc2, const float* sh2, float* y, float* pool, int B, int H, int W, int Cin, int mid,
                      int K, int stride, hipStream_t s) {
    ORBIT_REQUIRE(x && w1 && sc1 && sh1 && wdw && sc2 && sh2 && y, "mbconv_map: null pointer");
    ORBIT_REQUIRE(mbconv_map_supported(H, W, Cin, mid, K, stride), "mbconv_map: unsupported shape (%dx%d Cin=%d K=%d s=%d)",
                  H, W, Cin, K, stride);
    MbMapParams p;
    p.x = x, p.w1 = w1, p.sc1 = sc1, p.sh1 = sh1, p.wdw = wdw, p.sc2 = sc2, p.sh2 = sh2, p.y = y, p.pool = pool;
    p.B = B, p.mid = mid, p.G = 1;
    if (H == 14 && stride == 1 && Cin == 80 && K == 3) return launch_map<80, 3, 1, 14, 1>(p, s);
    if (H == 14 && stride == 1 && Cin == 80 && K == 5) return launch_map<80, 5, 1, 14, 1>(p, s);
    if (H == 14 && stride == 1 && Cin == 112) return launch_map<112, 5, 1, 14, 1>(p, s);
    if (H == 14 && stride == 2) return launch_map<112, 5, 2, 14, 1>(p, s);
    if (K == 5) return launch_map<192, 5, 1, 7, 2>(p, s);
    return launch_map<192, 3, 1, 7, 2>(p, s);
}

}  // namespace orbit
