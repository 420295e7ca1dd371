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
//   * 8 waves, role-split: waves 0-3 expand chunk i (MFMA straight from the register fragments against the chunk's W1
//     rows in LDS, BN1 + SiLU, scatter into the padded map tile Es[i & 1]) while waves 4-7 run the depthwise taps of chunk
//     i-1 from Es[(i-1) & 1] (ds_read_b128, NOUT outputs along a row per thread) + BN2 + SiLU -> HBM + pool sums, and
//     stage chunk i+1's weights; ONE barrier per chunk. Every SIMD hosts one wave of each role, so the matrix pipe and
//     the VALU / LDS pipes overlap by construction (a first version with two co-resident 4-wave blocks, each alternating
//     the two phases, left the MFMA pipe 26 % busy: measured, rocprofv3 SQ counters).
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
    static constexpr int TPW = (TILES + 3) / 4;          // row tiles per producer wave
    static constexpr int NG = CIN / 8;
    static constexpr int XS = CIN + 4;
    // consumer side: 8 units of 32 threads (one per channel of the chunk) = FB frames x NH column parts x NB row bands
    static constexpr int NH = 2;
    static constexpr int NB = 4 / FB;
    static constexpr int NOUT = (HO + NH - 1) / NH;      // output columns per thread
    static constexpr int R = (HO + NB - 1) / NB;         // output rows per thread
    static constexpr int NCOL = (NOUT - 1) * S + K;      // input columns a thread reads per input row
    static constexpr int NROW = (R - 1) * S + K;         // input rows a thread walks
    // channel-major padded map: E[channel][PLANE]; PLANE odd -> the 32 channels of a wave instruction hit 32 banks.
    // The slack after the last frame keeps the (discarded) reads of rows / columns beyond the map inside the plane.
    static constexpr int MAXIDX = (((FB - 1) * EH + ((NB - 1) * HO / NB) * S) * EH + (NH - 1) * NOUT * S) + (NROW - 1) * EH +
                                  NCOL - 1;                               // last element any consumer thread reads
    static constexpr int PLANE_MIN = (FB * EH * EH + 1) > (MAXIDX + 1) ? (FB * EH * EH + 1) : (MAXIDX + 1);
    static constexpr int PLANE = PLANE_MIN | 1;
    static constexpr int DUMP = FB * EH * EH;            // where the pixels that pad the last row tile are written
    static constexpr int WR = (8 * CIN + 255) / 256;     // expand-weight quads per consumer thread and chunk
    static constexpr int ES_FLOATS = 32 * PLANE;
    static size_t lds_bytes(int G) {
        return ((size_t)2 * ES_FLOATS + 2 * 32 * XS + (size_t)G * 256) * sizeof(float) + (size_t)TILES * 32 * sizeof(int);
    }
};

// 512 threads: waves 0-3 PRODUCE (expand chunk i on the matrix cores from their register-resident input fragments, BN1 +
// SiLU, scatter into the channel-major map Es[i & 1]); waves 4-7 CONSUME (depthwise + BN2 + SiLU + pool sums of chunk
// i-1 from Es[(i-1) & 1], and fetch chunk i+1's weights). One barrier per chunk. Hardware places wave w on SIMD w % 4, so
// every SIMD hosts one producer (MFMA pipe) and one consumer (VALU + LDS pipes).
//
// Consumer thread = (channel of the chunk, column part, row band): it walks the NROW input rows of its band ONCE, reads
// NCOL floats of each (ds_read_b32: adjacent lanes = adjacent channels = distinct banks) and scatters them into the R
// output rows they feed (taps in 25 registers). Every map element is read ~2.5x instead of the ~8x of a gather over tap
// rows with float4 channel quads - that gather form was LDS-bound (12 k cycles per chunk against the producers' 10 k;
// measured with s_memtime stamps) - and the tap weights never pass through LDS.
template <int CIN, int K, int S, int HW, int FB>
__global__ __launch_bounds__(512, 2) void mbconv_map_kernel(const MbMapParams p) {
    using Gm = MbMapGeom<CIN, K, S, HW, FB>;
    constexpr int HO = Gm::HO, EH = Gm::EH, NPIX = Gm::NPIX, P = Gm::P, TILES = Gm::TILES, TPW = Gm::TPW, NG = Gm::NG;
    constexpr int XS = Gm::XS, NH = Gm::NH, NB = Gm::NB, NOUT = Gm::NOUT, R = Gm::R, NCOL = Gm::NCOL, NROW = Gm::NROW;
    constexpr int WR = Gm::WR, PAD = Gm::PT_, ESF = Gm::ES_FLOATS, PLANE = Gm::PLANE;
    static_assert(CIN % 8 == 0 && TPW <= 2 && FB * NH * NB == 8, "geometry");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Es = smem;                                              // [2][32][PLANE] padded expanded map(s), per chunk parity
    float* Ws = Es + 2 * ESF;                                      // [2][32][XS] expand weights
    float* pool_all = Ws + 2 * 32 * XS;                            // [G][8 units][32 channels]
    int* estab = reinterpret_cast<int*>(pool_all + p.G * 256);     // [TILES*32] pixel -> offset inside a channel plane

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const bool producer = wave < 4;                                // wave-uniform role
    const int ctid = tid & 255, cw = wave & 3;                     // index inside the role's 256 threads / 4 waves
    const int b0 = blockIdx.y * FB;
    const int nchunks = p.mid >> 5;
    const int cfirst = (int)((long)blockIdx.x * nchunks / gridDim.x);        // balanced partition: group sizes differ by
    const int cend = (int)((long)(blockIdx.x + 1) * nchunks / gridDim.x);    // at most one chunk
    const int n = cend - cfirst;
    if (n <= 0) return;

    // ---- zero both padded maps (their borders stay zero), build the pixel -> plane offset table -----------------------
    for (int i = tid; i < 2 * ESF / 4; i += 512) reinterpret_cast<v4f*>(Es)[i] = (v4f){0.f, 0.f, 0.f, 0.f};
    for (int px = tid; px < TILES * 32; px += 512) {
        int off = Gm::DUMP;
        if (px < P) {
            const int f = px / NPIX, pp = px - f * NPIX;
            const int yy = pp / HW, xx = pp - yy * HW;
            off = (f * EH + yy + PAD) * EH + xx + PAD;
        }
        estab[px] = off;
    }

    if (producer) {
        // ---- A-operand fragments of this wave's row tiles: registers for the whole block -----------------------------
        v4f xr[TPW][NG];
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int px = (cw + 4 * i) * 32 + l31;
            const int f = px / NPIX, pp = px - f * NPIX;
            const bool ok = (cw + 4 * i) < TILES && px < P && b0 + f < p.B;
            const float* src = p.x + ((size_t)(b0 + (ok ? f : 0)) * NPIX + (ok ? pp : 0)) * CIN + lh * 4;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const v4f v = *reinterpret_cast<const v4f*>(src + g * 8);  // unconditional load from a safe address
                xr[i][g] = ok ? v : (v4f){0.f, 0.f, 0.f, 0.f};
            }
        }
        const bool two = TPW == 2 && cw + 4 < TILES;  // wave-uniform: this wave owns a second row tile
        float s1n = p.sc1[cfirst * 32 + l31], h1n = p.sh1[cfirst * 32 + l31];
        __syncthreads();  // P0: maps zeroed, table built, chunk 0's weights in Ws[0] (consumers)
        for (int i = 0; i <= n; ++i) {
            if (i < n) {
                const float s1 = s1n, h1 = h1n;
                if (i + 1 < n) s1n = p.sc1[(cfirst + i + 1) * 32 + l31], h1n = p.sh1[(cfirst + i + 1) * 32 + l31];
                float* E = Es + (i & 1) * ESF + l31 * PLANE;
                const float* Bq = Ws + (i & 1) * 32 * XS + l31 * XS + lh * 4;
                f32x16 acc[TPW];
#pragma unroll
                for (int t = 0; t < TPW; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
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
                for (int t = 0; t < TPW; ++t) {
                    if (t == 1 && !two) break;
                    const int tile = cw + 4 * t;
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq) {
                        const v4i er = *reinterpret_cast<const v4i*>(estab + tile * 32 + 8 * rq + 4 * lh);
#pragma unroll
                        for (int j = 0; j < 4; ++j) E[er[j]] = silu_m(acc[t][rq * 4 + j] * s1 + h1);
                    }
                }
            }
            __syncthreads();  // chunk i expanded; consumers finished chunk i-1 and staged chunk i+1's weights
        }
    } else {
        // ---- consumer thread = (channel ch, frame f, column part, row band) --------------------------------------------
        const int ch = ctid & 31, unit = ctid >> 5;
        const int f = unit / (NH * NB), hb = unit % (NH * NB);
        const int oy0 = (hb / NH) * HO / NB, oy1 = (hb / NH + 1) * HO / NB;   // output rows [oy0, oy1) of this band
        const int ox0 = (hb % NH) * NOUT;                                     // output columns [ox0, min(ox0 + NOUT, HO))
        const bool frame_ok = b0 + f < p.B;
        const int ebase = (f * EH + oy0 * S) * EH + ox0 * S;                  // first input element of the band

        v4f wreg[WR];
        float tapn[K * K], s2n = 0.f, h2n = 0.f;
        auto load_weights = [&](int c0) {  // expand rows (for the producers) + this channel's taps -> registers
#pragma unroll
            for (int u = 0; u < WR; ++u) {
                const int i = ctid + u * 256;
                const int r = i / (CIN / 4), c4 = i - r * (CIN / 4);
                wreg[u] = (v4f){0.f, 0.f, 0.f, 0.f};
                if (i < 8 * CIN) wreg[u] = *reinterpret_cast<const v4f*>(p.w1 + (size_t)(c0 + r) * CIN + c4 * 4);
            }
        };
        auto load_taps = [&](int c0) {
#pragma unroll
            for (int t = 0; t < K * K; ++t) tapn[t] = p.wdw[(size_t)t * p.mid + c0 + ch];
            s2n = p.sc2[c0 + ch], h2n = p.sh2[c0 + ch];
        };
        auto store_weights = [&](int ci) {  // registers -> Ws[ci & 1]
            float* W = Ws + (ci & 1) * 32 * XS;
#pragma unroll
            for (int u = 0; u < WR; ++u) {
                const int i = ctid + u * 256;
                const int r = i / (CIN / 4), c4 = i - r * (CIN / 4);
                if (i < 8 * CIN) *reinterpret_cast<v4f*>(W + r * XS + c4 * 4) = wreg[u];
            }
        };
        load_weights(cfirst * 32);
        store_weights(0);
        load_taps(cfirst * 32);
        __syncthreads();  // P0
        for (int i = 0; i <= n; ++i) {
            // Ws[(i+1) & 1] was last read by the producers in iteration i-1: chunk i+1's expand rows are requested now
            // and written at the end of the iteration, so their latency hides under the depthwise work below
            if (i + 1 < n) load_weights((cfirst + i + 1) * 32);
            if (i >= 1) {
                const int ci = i - 1, c0 = (cfirst + ci) * 32;
                float tap[K * K];
#pragma unroll
                for (int t = 0; t < K * K; ++t) tap[t] = tapn[t];
                const float s2 = s2n, h2 = h2n;
                if (i < n) load_taps((cfirst + i) * 32);  // next chunk's taps / BN2: in flight during this chunk's walk
                const float* E = Es + (ci & 1) * ESF + ch * PLANE + ebase;
                float acc2[R][NOUT];
#pragma unroll
                for (int ry = 0; ry < R; ++ry)
#pragma unroll
                    for (int j = 0; j < NOUT; ++j) acc2[ry][j] = 0.f;
#pragma unroll
                for (int yy = 0; yy < NROW; ++yy) {
                    float col[NCOL];
#pragma unroll
                    for (int q = 0; q < NCOL; ++q) col[q] = E[yy * EH + q];
#pragma unroll
                    for (int ry = 0; ry < R; ++ry) {
                        constexpr int dummy = 0;
                        (void)dummy;
                        const int kh = yy - ry * S;  // compile-time after unrolling
                        if (kh >= 0 && kh < K) {
#pragma unroll
                            for (int kw = 0; kw < K; ++kw)
#pragma unroll
                                for (int j = 0; j < NOUT; ++j) acc2[ry][j] = fmaf(col[j * S + kw], tap[kh * K + kw], acc2[ry][j]);
                        }
                    }
                }
                float psum = 0.f;
#pragma unroll
                for (int ry = 0; ry < R; ++ry) {
                    const int oy = oy0 + ry;
#pragma unroll
                    for (int j = 0; j < NOUT; ++j) {
                        if (frame_ok && oy < oy1 && ox0 + j < HO) {
                            const float o = silu_m(acc2[ry][j] * s2 + h2);
                            p.y[(((size_t)(b0 + f) * HO + oy) * HO + ox0 + j) * p.mid + c0 + ch] = o;
                            psum += o;
                        }
                    }
                }
                if (p.pool) pool_all[(ci * 8 + unit) * 32 + ch] = psum;
            }
            if (i + 1 < n) store_weights(i + 1);
            __syncthreads();
        }
    }
    if (p.pool) {  // (the loop's last barrier made pool_all complete) sum the units of a frame in fixed order
        constexpr int UPF = NH * NB;
        for (int i = tid; i < n * 32 * FB; i += 512) {
            const int c = i & 31, ci = (i >> 5) % n, f = (i >> 5) / n;
            if (b0 + f < p.B) {
                float t = pool_all[(ci * 8 + f * UPF) * 32 + c];
#pragma unroll
                for (int u = 1; u < UPF; ++u) t += pool_all[(ci * 8 + f * UPF + u) * 32 + c];
                p.pool[(size_t)(b0 + f) * p.mid + (cfirst + ci) * 32 + c] = t;
            }
        }
    }
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
    // tiles) and one pipeline-fill iteration, and the chip runs 256 blocks at a time (one 8-wave block per CU), so the
    // cost of a grouping is rounds(blocks / 256) x (1.5 + chunks per block); the cheapest wins (ties: fewer groups).
    // mbmap_groups overrides.
    int groups = get_option("mbmap_groups");
    if (groups <= 0) {
        double best = 1e30;
        for (int g = 1; g <= nchunks; ++g) {
            if (cdiv(nchunks, g) > 8) continue;  // pool scratch holds 8 chunks per block (8 KB)
            const double rounds = (double)cdiv(frames * g, 256);
            const double cost = rounds * (1.5 + cdiv(nchunks, g));
            if (cost < best - 1e-9) best = cost, groups = g;
        }
    }
    if (groups > nchunks) groups = nchunks;
    if (cdiv(nchunks, groups) > 8) groups = cdiv(nchunks, 8);
    p.G = cdiv(nchunks, groups);
    const size_t lds = Gm::lds_bytes(p.G);
    auto kern = mbconv_map_kernel<CIN, K, S, HW, FB>;
    static bool attr_set = false;
    if (!attr_set) {
        ORBIT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int)Gm::lds_bytes(8)));
        attr_set = true;
    }
    ORBIT_REQUIRE(p.G <= 8, "mbconv_map: at most 8 chunks per block (G=%d)", p.G);
    const double pix = (double)p.B * HW * HW, opix = (double)p.B * Gm::HO * Gm::HO;
    const int rec = prof_start("mbconv_map", 2.0 * pix * p.mid * CIN, 4.0 * (pix * CIN + opix * p.mid), s);
    kern<<<dim3(groups, frames), 512, lds, s>>>(p);
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
