// Narrow pointwise projection: y[m][co] = act(bn(sum_k gate[b][k] * x[m][k] * w[co][k]) + residual[m][co]) for Cout <= 32 -
// EfficientNet-B0's high-resolution linear bottlenecks (blocks.0.0.conv_pw 32 -> 16 at 112x112, blocks.1.0.conv_pwl
// 96 -> 24 at 56x56; timm DepthwiseSeparableConv / InvertedResidual, reached from the reference's
// model/feature_extractors.py:39-43). These layers carry 2.6-6 FLOP per byte: pure HBM streams (301-481 MB per 200 frames).
//
// conv_igemm serves them with a 128x32 tile whose whole reduction is ONE K-tile: a block loads, waits, multiplies, stores -
// nothing of its own overlaps, and what is in flight per CU is whatever the 4 co-resident blocks happen to have outstanding
// (3.4-4.2 TB/s). Here a wave owns 32 pixels at a time and the roles of the MFMA operands are swapped:
//   A = filter rows (channel l31, k = 8g + 4*lh ..+3; from LDS, staged once per block),
//   B = pixels      (pixel   l31, same k: a lane's 16-byte loads straight from the NHWC tensor, no LDS stage),
// so the accumulator holds, per lane, 4 x 4 CONSECUTIVE channels of ONE pixel: the output leaves as 16-byte stores and
// the residual arrives as 16-byte loads without an LDS transpose. The B-fragment of group g is re-requested for the
// wave's NEXT tile right after the MFMAs of group g consumed it (rolling prefetch in place: K/8 loads of 1 KB per wave in
// flight at all times, no second register set). A block serves a run of tiles of ONE frame, so the squeeze-excite gate is
// staged once in LDS. Same k-order and the same epilogue arithmetic as conv_igemm (products commute): bit-identical.
#include "common.h"

namespace orbit {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using v4f = __attribute__((ext_vector_type(4))) float;

struct PwNarrowParams {
    const float* x;         // [B][HW][Cin]
    const float* w;         // packed [>= 32 rows][KT], rows >= Cout are zero
    const float* scale;     // [Cout] or nullptr
    const float* shift;
    const float* residual;  // [B][HW][Cout] or nullptr
    const float* gate;      // [B][Cin] or nullptr
    float* y;               // [B][HW][Cout]
    int HW, Cin, Cout, KT, act, parts, tiles_per_part;
};

__device__ __forceinline__ float pw_act(float v, int act) {
    if (act == ORBIT_ACT_RELU) return fmaxf(v, 0.f);
    if (act == ORBIT_ACT_SILU) return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
    return v;
}

template <int NG>
__global__ __launch_bounds__(256) void pw_narrow_kernel(const PwNarrowParams p) {
    constexpr int K = NG * 8, WS = K + 4;        // LDS filter row stride (floats)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ws = smem;                            // [32][WS]
    float* Gs = Ws + 32 * WS;                    // [K] gate of this block's frame (ones without a gate)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int b = blockIdx.x / p.parts, part = blockIdx.x - b * p.parts;
    const int tiles = p.HW >> 5;                 // 32-pixel tiles per frame (HW % 32 == 0)
    const int t0 = part * p.tiles_per_part;
    const int t1 = t0 + p.tiles_per_part < tiles ? t0 + p.tiles_per_part : tiles;

    for (int i = tid; i < 32 * (K / 4); i += 256) {
        const int r = i / (K / 4), c4 = i - r * (K / 4);
        *reinterpret_cast<v4f*>(Ws + r * WS + c4 * 4) = *reinterpret_cast<const v4f*>(p.w + (size_t)r * p.KT + c4 * 4);
    }
    for (int i = tid; i < K; i += 256) Gs[i] = p.gate ? p.gate[(size_t)b * p.Cin + i] : 1.0f;
    // folded BatchNorm, staged too (the epilogue reads its channel quads from LDS: 32 registers less in the tile loop)
    float* Ss = Gs + K;                          // [32] scale, [32] shift
    if (tid < 32) {
        Ss[tid] = (tid < p.Cout && p.scale) ? p.scale[tid] : 1.0f;
        Ss[32 + tid] = (tid < p.Cout && p.shift) ? p.shift[tid] : 0.0f;
    }
    const float* xb = p.x + (size_t)b * p.HW * p.Cin + (size_t)l31 * p.Cin + 4 * lh;
    v4f xa[NG];
    int t = t0 + wave;
    if (t < t1) {
#pragma unroll
        for (int g = 0; g < NG; ++g) xa[g] = *reinterpret_cast<const v4f*>(xb + (size_t)t * 32 * p.Cin + 8 * g);
    }
    __syncthreads();
    const float* wrow = Ws + l31 * WS + 4 * lh;
    const float* grow = Gs + 4 * lh;
    for (; t < t1; t += 4) {
        const int tn = t + 4 < t1 ? t + 4 : t;    // the wave's next tile (the last one re-reads itself: harmless)
        const float* xn = xb + (size_t)tn * 32 * p.Cin;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        // filter / gate quads one group ahead of their MFMAs; the scheduling barrier keeps hipcc from hoisting all K/8 LDS
        // reads of the unrolled loop to its top (K = 144: 72 + 72 more live registers, one wave per SIMD)
        v4f wv = *reinterpret_cast<const v4f*>(wrow), gv = *reinterpret_cast<const v4f*>(grow);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            v4f wn = wv, gn = gv;
            if (g + 1 < NG) {
                wn = *reinterpret_cast<const v4f*>(wrow + 8 * (g + 1));
                gn = *reinterpret_cast<const v4f*>(grow + 8 * (g + 1));
            }
            const v4f xg = xa[g] * gv;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[kk], xg[kk], acc, 0, 0, 0);
            // rolling prefetch of the next tile in bursts of four groups = the 128 bytes of a pixel's cache line together
            // (one group at a time, a line was touched four times ~250 cycles apart and fell out of the vector L1 in between)
            constexpr int BURST = 4;
            if (g % BURST == BURST - 1 || g == NG - 1) {
#pragma unroll
                for (int q = g / BURST * BURST; q <= g; ++q) xa[q] = *reinterpret_cast<const v4f*>(xn + 8 * q);
            }
            __builtin_amdgcn_sched_barrier(0);
            wv = wn, gv = gn;
        }
        // C/D layout: column = l31 (pixel), row = (r & 3) + 8 (r >> 2) + 4 lh (channel): 4 consecutive channels per rq
        const size_t m = (size_t)b * p.HW + (size_t)t * 32 + l31;
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            const int c0 = 8 * rq + 4 * lh;
            if (c0 < p.Cout) {
                v4f res = {0.f, 0.f, 0.f, 0.f};
                if (p.residual) res = *reinterpret_cast<const v4f*>(p.residual + m * p.Cout + c0);
                v4f o = {acc[4 * rq], acc[4 * rq + 1], acc[4 * rq + 2], acc[4 * rq + 3]};
                o = o * *reinterpret_cast<const v4f*>(Ss + c0) + *reinterpret_cast<const v4f*>(Ss + 32 + c0);
                o += res;
                o[0] = pw_act(o[0], p.act), o[1] = pw_act(o[1], p.act), o[2] = pw_act(o[2], p.act), o[3] = pw_act(o[3], p.act);
                *reinterpret_cast<v4f*>(p.y + m * p.Cout + c0) = o;
            }
        }
    }
}

bool pw_narrow_supported(const ConvDesc& d) {
    if (get_option("pw_narrow") == 0) return false;
    if (d.x_nchw || d.pool2 || d.KH != 1 || d.KW != 1 || d.stride != 1 || d.pad_t != 0 || d.pad_l != 0) return false;
    if (d.Cout > 32 || d.Cout % 4 != 0 || d.Cin % 8 != 0) return false;
    // the instantiated reduction lengths; Cin = 144 (blocks.1.1: 576-byte rows, 4.5 cache lines) measured 135 us against
    // conv_igemm's 123 and stays there
    if (d.Cin != 32 && d.Cin != 96) return false;
    const int HW = d.Ho * d.Wo;
    return HW % 32 == 0 && HW >= 1024;  // high-resolution maps: the HBM-bound regime (and a frame is a whole number of tiles)
}

int launch_pw_narrow(const ConvDesc& d, hipStream_t s) {
    ORBIT_REQUIRE(pw_narrow_supported(d), "pw_narrow: unsupported shape");
    const ConvPackGeom g = conv_pack_geom(d.Cin, d.Cout, 1, 1, 0);
    PwNarrowParams p;
    p.x = d.x, p.w = d.w_packed, p.scale = d.scale, p.shift = d.shift, p.residual = d.residual, p.gate = d.gate, p.y = d.y;
    p.HW = d.Ho * d.Wo, p.Cin = d.Cin, p.Cout = d.Cout, p.KT = g.kt, p.act = d.act;
    const int tiles = p.HW / 32;
    // a block = 4 waves x ~6 tiles of one frame (enough blocks to fill the chip, few enough to amortise the filter staging)
    p.parts = cdiv(tiles, 24);
    p.tiles_per_part = cdiv(tiles, p.parts);
    const int grid = d.B * p.parts;
    const size_t lds = ((size_t)32 * (d.Cin + 4) + d.Cin + 64) * sizeof(float);
    char name[48];
    snprintf(name, sizeof(name), "conv_pw_narrow<%d,%s>", d.Cin, d.gate ? "gate" : "nogate");
    const double pix = (double)d.B * p.HW;
    const int rec = prof_start(name, 2.0 * pix * d.Cout * d.Cin,
                               4.0 * (pix * d.Cin + pix * d.Cout * (d.residual ? 2.0 : 1.0) + (double)d.Cout * d.Cin), s);
    if (d.Cin == 32) pw_narrow_kernel<4><<<grid, 256, lds, s>>>(p);
    else pw_narrow_kernel<12><<<grid, 256, lds, s>>>(p);
    prof_stop(rec, s);
    ORBIT_LAUNCH_CHECK();
    return ORBIT_OK;
}

}  // namespace orbit
