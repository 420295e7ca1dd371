// Pointwise (1x1, stride 1) convolution as a REGISTER GEMM on v_mfma_f32_16x16x4_f32: no LDS stage, no barrier in the K loop.
//
// Every EfficientNet conv but the stem is a pointwise conv (reference: timm tf_efficientnet_b0 through
// model/feature_extractors.py:39-43, run by model/few_shot_recognisers.py:99-153), i.e. a plain GEMM
//     y[m][n] = act(scale[n] * sum_k x[m][k] * gate[b(m)][k] * w[n][k] + shift[n] + residual[m][n])
// with M = B*H*W pixels (NHWC rows ARE the GEMM rows), K = Cin, N = Cout. conv_igemm.hip serves it with a block tile staged
// through LDS; on these shapes (K = 80 .. 1152, a few hundred to a few thousand 64x64 tiles, 30-100 us launches) its time
// goes to what surrounds the MFMAs - one prologue + LDS-staged epilogue per 64x64 tile, two barriers per K-tile, whole-block
// granularity on 256 CUs (profiles/r03: 0.42 of the fp32 MFMA peak, the K loop itself at ~0.9). This kernel removes the
// block-level machinery instead of tuning it:
//
//   * The unit of work is ONE WAVE: 32 pixels x 16*T output channels (2 x T accumulator tiles of 16x16, T = 3..8, chosen per
//     layer so that 16*T divides the padded Cout: 80 = 5 x 16 and 112 = 7 x 16 are exact, where the 32-wide tiling pads to
//     96 / 128), over all of K or - for the long-K projections on small maps - a quarter / half of K (the K-split partials of
//     a block's waves are summed through LDS in a fixed order: deterministic). Waves never wait for each other in the K loop.
//   * Operands go global -> registers -> MFMA. The weights are packed at load time in FRAGMENT ORDER
//     wf[n_tile16][k_chunk16][lane][4]: one wave load instruction = 1 KiB contiguous, every lane gets the float4 whose four
//     elements feed four consecutive MFMAs (lane (i, q) holds w[16 t + i][16 c + 4 q + e], e = MFMA number). The activations
//     are read the same way straight from the NHWC tensor (lane (j, q) reads x[m0 + j][16 c + 4 q .. +3]: 64 contiguous bytes
//     per pixel row and instruction, half a cache line; the other half is the next chunk). The weight matrix (<= 1.5 MB)
//     lives in every XCD's L2; the XCD-aware block order keeps the column groups that share a pixel tile on one XCD, so the
//     activation tensor crosses the fabric once.
//   * The MFMA runs TRANSPOSED (A operand = weights, B operand = pixels): D[i][j] puts four consecutive CHANNELS of one pixel
//     into a lane's four accumulator registers, so the epilogue stores float4s (64 contiguous bytes per pixel and
//     instruction) without staging the tile through LDS, and scale / shift / residual are float4 loads of the same shape.
//   * Operand requests run one chunk (16 k) ahead of the MFMAs that consume them: a tile's weight float4 is refilled in place
//     right after the tile's 8 MFMAs of this chunk, the pixel float4s alternate between two register sets.
//
// fp32 in, fp32 accumulate (each output is a k-ordered fmaf chain per K slice; slices are added in slice order).
#include <algorithm>
#include <cmath>
#include "common.h"

namespace orbit {

using f32x4 = __attribute__((ext_vector_type(4))) float;

struct PwrParams {
    const float* x;
    const float* wf;
    float* y;
    const float* scale;
    const float* shift;
    const float* residual;
    const float* gate;
    int M, K, Cout, act;
    int nchunk;    // K / 16
    int n_groups;  // column groups of T 16-channel tiles
    int m_tiles;   // 32-pixel tiles
    int wk;        // waves of a block along K (1, 2, 4); the other 4 / wk wave slots are consecutive pixel tiles
    int cps;       // chunks per K slice
    FastDiv fd_hw; // pixel -> frame (gate rows)
    FastDiv fd_ng; // block -> (pixel group, column group)
};

__device__ __forceinline__ int pwr_xcd_remap(int bid, int nblk) {  // as conv_igemm.hip: contiguous logical runs per XCD
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, slot = bid >> 3;
    const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + slot;
}

__device__ __forceinline__ float pwr_act(float v, int act) {
    if (act == ORBIT_ACT_RELU) return fmaxf(v, 0.f);
    if (act == ORBIT_ACT_SILU) return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));  // the library's SiLU (conv_igemm.hip)
    return v;
}

// ODD: every K slice of the launch holds an odd number of chunks (the launcher cuts the slices so that all have one parity)
template <int T, bool GATE, bool ODD>
__global__ __launch_bounds__(256) void pw_rgemm_kernel(const PwrParams p) {
    extern __shared__ __attribute__((aligned(16))) float pwr_red[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, q = lane >> 4;
    const int bid = pwr_xcd_remap(blockIdx.x, gridDim.x);
    const int mg = (int)fdiv((unsigned)bid, p.fd_ng), ng = bid - mg * p.n_groups;  // column groups of a pixel group are neighbours
    const int wm = p.wk == 1 ? wave : p.wk == 2 ? (wave >> 1) : 0;
    const int ks = p.wk == 1 ? 0 : p.wk == 2 ? (wave & 1) : wave;
    const int rt = mg * (4 / p.wk) + wm;
    const bool live = rt < p.m_tiles;  // wave-uniform
    if (p.wk == 1 && !live) return;    // (no barrier on this path)
    const int m0 = rt * 32;
    const int c0 = ks * p.cps, c1 = ks == p.wk - 1 ? p.nchunk : c0 + p.cps;  // (the last slice takes the remainder)

    // this lane's two pixels (rows beyond M are clamped: what they compute is never stored)
    const float* px[2];
    const float* pg[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int m = min(m0 + 16 * r + j, p.M - 1);
        px[r] = p.x + (size_t)m * p.K + 4 * q;
        if (GATE) pg[r] = p.gate + (size_t)fdiv((unsigned)m, p.fd_hw) * p.K + 4 * q;
    }
    const float* pw = p.wf + (size_t)ng * T * p.nchunk * 256 + lane * 4;
    const int wstride = p.nchunk * 256;  // floats between the 16-channel tiles of one chunk (< 2^31: Cout * K bounded)

    f32x4 acc[2][T];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int t = 0; t < T; ++t) acc[r][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // Operand registers. The pixels' float4s (and gates) of a chunk feed all T tiles, so they are double-buffered; a weight
    // float4 wb[t] is dead after its tile's 8 MFMAs and is refilled IN PLACE with the next chunk's right there - the request
    // then has the other T - 1 tiles' MFMAs (256 (T - 1) cycles) to complete, with T instead of 2 T weight registers.
    f32x4 xa[2][2], ga[2][2], wb[T];
    auto load_x = [&](int c, int set) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            xa[set][r] = *reinterpret_cast<const f32x4*>(px[r] + 16 * c);
            if (GATE) ga[set][r] = *reinterpret_cast<const f32x4*>(pg[r] + 16 * c);
        }
    };
    auto load_w = [&](int c, int t) { wb[t] = *reinterpret_cast<const f32x4*>(pw + (size_t)t * wstride + c * 256); };
    // one chunk: MFMAs on x set `set` and wb, operands of chunk `cn` requested on the way (x into the other set)
    auto chunk = [&](int set, int cn) {
        load_x(cn, set ^ 1);
        f32x4 a0 = xa[set][0], a1 = xa[set][1];
        if (GATE) a0 *= ga[set][0], a1 *= ga[set][1];
#pragma unroll
        for (int t = 0; t < T; ++t) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {  // a tile's two accumulators alternate: 64 cycles between dependent MFMAs (40 needed)
                acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[t][e], a0[e], acc[0][t], 0, 0, 0);
                acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[t][e], a1[e], acc[1][t], 0, 0, 0);
            }
            load_w(cn, t);
        }
        // pin that order (0x020 = VMEM read, 0x008 = MFMA): left alone, the scheduler sinks the requests behind the last MFMA
        // of the chunk and every chunk starts by waiting for them
        __builtin_amdgcn_sched_group_barrier(0x020, GATE ? 4 : 2, 0);
#pragma unroll
        for (int t = 0; t < T; ++t) {
            __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
    };
    if (live) {
        // The steady state is a pair of chunks (x sets 0, 1) with no branch between MFMAs, so the accumulators stay where they
        // are. An odd chunk count is peeled off the front at COMPILE time: behind any runtime branch around a chunk hipcc moved
        // all 8 T accumulators to fresh registers, twice.
        int c = c0;
#pragma unroll
        for (int t = 0; t < T; ++t) load_w(c0, t);
        if (ODD) {
            load_x(c0, 1);
            chunk(1, c0 + 1 < c1 ? c0 + 1 : c0);
            ++c;
        } else {
            load_x(c0, 0);
        }
        for (; c < c1; c += 2) {
            chunk(0, c + 1);
            chunk(1, c + 2 < c1 ? c + 2 : c1 - 1);
        }
    }

    if (p.wk > 1) {  // K slices of a pixel tile: slices 1.. go through LDS, slice 0 adds them in slice order
        f32x4* R = reinterpret_cast<f32x4*>(pwr_red);
        if (ks > 0) {
            const int slot = wm * (p.wk - 1) + ks - 1;
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int t = 0; t < T; ++t) R[((slot * 2 + r) * T + t) * 64 + lane] = acc[r][t];
        }
        __syncthreads();
        if (ks > 0 || !live) return;
        for (int s2 = 1; s2 < p.wk; ++s2) {
            const int slot = wm * (p.wk - 1) + s2 - 1;
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int t = 0; t < T; ++t) acc[r][t] += R[((slot * 2 + r) * T + t) * 64 + lane];
        }
    }

    // epilogue: lane (j, q) holds channels 16 t + 4 q .. + 3 of pixels m0 + j and m0 + 16 + j
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int n = (ng * T + t) * 16 + 4 * q;
        if (n >= p.Cout) continue;  // (Cout % 4 == 0: a float4 is all inside or all outside)
        f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
        if (p.scale) sc = *reinterpret_cast<const f32x4*>(p.scale + n);
        if (p.shift) sh = *reinterpret_cast<const f32x4*>(p.shift + n);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int m = m0 + 16 * r + j;
            if (m >= p.M) continue;
            f32x4 v = acc[r][t] * sc + sh;
            if (p.residual) v += *reinterpret_cast<const f32x4*>(p.residual + (size_t)m * p.Cout + n);
            v[0] = pwr_act(v[0], p.act), v[1] = pwr_act(v[1], p.act), v[2] = pwr_act(v[2], p.act), v[3] = pwr_act(v[3], p.act);
            *reinterpret_cast<f32x4*>(p.y + (size_t)m * p.Cout + n) = v;
        }
    }
}

// ---- weight packing: [Cout][Cin] (1x1 OIHW) -> fragment order [tile16][chunk16][lane][4], zero rows beyond Cout -------
size_t conv_frag_floats(int Cin, int Cout, int KH, int KW, int x_nchw) {
    if (x_nchw || KH != 1 || KW != 1 || Cin % 16 != 0 || Cout % 4 != 0) return 0;
    return (size_t)(cdiv(Cout, 16) + 7) * (Cin / 16) * 256;  // + 7 tiles: any T <= 8 may read past the last column group
}

__global__ __launch_bounds__(256) void conv_frag_pack_kernel(const float* __restrict__ w, float* __restrict__ wf, int Cin,
                                                             int Cout, size_t total) {
    const int nchunk = Cin / 16;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int e = (int)(i & 3), lane = (int)((i >> 2) & 63);
        const size_t tc = i >> 8;
        const int c = (int)(tc % nchunk), tile = (int)(tc / nchunk);
        const int n = tile * 16 + (lane & 15), k = c * 16 + 4 * (lane >> 4) + e;
        wf[i] = n < Cout ? w[(size_t)n * Cin + k] : 0.f;
    }
}

int conv_frag_pack_weights(const float* w_oihw, float* w_frag, int Cin, int Cout, hipStream_t s) {
    const size_t total = conv_frag_floats(Cin, Cout, 1, 1, 0);
    ORBIT_REQUIRE(total > 0, "conv_frag_pack: not a fragment-packable filter (Cin=%d Cout=%d)", Cin, Cout);
    const int blocks = (int)std::min<size_t>((total + 255) / 256, 4096);
    conv_frag_pack_kernel<<<blocks, 256, 0, s>>>(w_oihw, w_frag, Cin, Cout, total);
    ORBIT_LAUNCH_CHECK();
    return ORBIT_OK;
}

bool pw_rgemm_supported(const ConvDesc& d) {
    return d.w_frag != nullptr && !d.x_nchw && d.KH == 1 && d.KW == 1 && d.stride == 1 && d.pad_t == 0 && d.pad_l == 0 &&
           !d.pool2 && d.Cin % 16 == 0 && d.Cout % 4 == 0 && d.stats == nullptr && d.y_raw == nullptr && d.Ho == d.H &&
           d.Wo == d.W && (long long)d.Cout * d.Cin < (1ll << 30);
}

// chunks per K slice for `wk` slices: slices 0 .. wk-2 take this many, the last one the remainder. With an even chunk count
// all slices then have the parity of the slice size (the kernel is compiled per parity); an odd count is never split.
static int pwr_slice(int nchunk, int wk) {
    if (wk <= 1) return nchunk;
    const int up = cdiv(nchunk, wk);
    return nchunk - (wk - 1) * up >= 1 ? up : nchunk / wk;
}

// The K slicing decides the order in which a pixel's products are summed, so it is a function of the LAYER (K) only, never of
// the pixel count: the same frame gives the same bits whatever batch it arrives in (tests/test_gpu_extractors.py holds 2 048
// frames in one forward against 8 x 256). T does not touch the summation order and is free to follow the batch.
static int pwr_default_wk(int nchunk) { return (nchunk & 1) ? 1 : nchunk >= 32 ? 4 : nchunk >= 16 ? 2 : 1; }

// (T, wk) for a layer. Measured on MI355X (tools/rgemm_bench.py sweep, the load / MFMA ablation of DESIGN.md section 4.0r4):
// a launch lasts about max(matrix time, operand time) + a fixed part, where
//   matrix time  = ceil(blocks / 256) x the MFMA cycles of one wave (a block is one wave on each SIMD of a CU and the
//                  dispatcher deals whole blocks to the 256 CUs: 770 blocks cost four rounds, 768 three);
//   operand time = the bytes the waves request (2 KiB of pixels + T KiB of weights (+ gates) per wave and chunk) / ~15 TB/s,
//                  the rate the vector L1s deliver this access pattern at - why small T loses on long-K layers;
//   fixed        = ~8 us (launch, first operand round trip, epilogue) + the output at ~5 TB/s.
// The cheapest T = 3..8 at the layer's K slicing wins. Returns the estimate (us).
static double pwr_plan(int M, int K, int Cout, bool gate, int& T, int& wk) {
    const int tiles16 = cdiv(Cout, 16), nchunk = K / 16, m_tiles = cdiv(M, 32);
    double best = 1e300;
    T = 4, wk = 1;
    for (int t = 3; t <= 8; ++t) {
        for (int w = 1; w <= 4; w *= 2) {
            if (w != pwr_default_wk(nchunk)) continue;
            if (w > 1 && (nchunk / w < 2 || (nchunk & 1))) continue;  // (K slices of one parity: pwr_slice)
            const int groups = cdiv(tiles16, t);
            const double blocks = (double)cdiv(m_tiles, 4 / w) * groups;
            const int sl = pwr_slice(nchunk, w);
            const double wave_cycles = (double)std::max(sl, nchunk - (w - 1) * sl) * 8 * t * 32;
            const double t_mfma = std::ceil(blocks / 256.0) * wave_cycles / 2150.0;  // us at the ~2.15 GHz the chip sustains
            const double t_mem = blocks * 4 * ((double)nchunk / w) * (2.0 + t + (gate ? 0.5 : 0.0)) * 1024 / 15e6;
            const double fixed = 8.0 + 4.0 * M * (double)Cout / 5e6 + (w > 1 ? 2.0 : 0.0);
            const double cost = std::max(t_mfma, t_mem) * 1.12 + fixed;
            if (cost < best) best = cost, T = t, wk = w;
        }
    }
    return best;
}

// Which pointwise convs the register GEMM takes from the LDS-tiled kernel when `conv_rgemm` = 1. Standalone (in-process A/B
// over the EfficientNet-B0 @224 shapes, 200 frames, tools/rgemm_bench.py) it wins on 11 of 16 shapes: Cout = 40 / 80 / 112
// (exact 16-channel tiling: +5..22 %), the expansions (+4..13 %), 1152 -> 320 at 7x7 (+20..27 %: 1 535 blocks = six full
// rounds where the 64x64 tiling has 770 tiles = three rounds and two blocks); it loses on the narrow HBM-bound projections of
// the large maps (Cout <= 32: -20..26 %) and the 192-channel projections of the 7x7 stage (-7..10 %). INSIDE the network the
// picture is different (bench.py, whole task, tools/ab_opts.sh): the register GEMM asks the vector L1s for ~2x the operand
// bytes of the LDS-tiled kernel, and next to the other stream's kernels that costs what the shorter launch saves - with the
// projection and expansion classes switched on the task is 1 % SLOWER under the support / query overlap and 1 % faster
// without it; only the class "projections of <= 8x8 maps from >= 512 to > 256 channels" (1152 -> 320 at 7x7) is a gain in
// both modes. That class is what `conv_rgemm` = 1 (default) routes here; 2 = every conv the kernel supports (parity tests).
bool pw_rgemm_preferred(const ConvDesc& d) {
    if (d.Cout < 40) return false;
    // (a rule on the LAYER, not on the pixel count: which kernel serves a conv decides its summation order, and a frame must
    // give the same bits in any batch; maps this small only occur behind >= 512 input channels in the supported networks)
    return d.H * d.W <= 64 && d.Cin >= 512 && d.Cout > 256;
}

template <int T, bool GATE>
static int pwr_launch(const PwrParams& p, int grid, size_t lds, hipStream_t s) {
    if (p.cps & 1) pw_rgemm_kernel<T, GATE, true><<<grid, 256, lds, s>>>(p);
    else pw_rgemm_kernel<T, GATE, false><<<grid, 256, lds, s>>>(p);
    ORBIT_LAUNCH_CHECK();
    return ORBIT_OK;
}

template <bool GATE>
static int pwr_dispatch(int T, const PwrParams& p, int grid, size_t lds, hipStream_t s) {
    switch (T) {
        case 3: return pwr_launch<3, GATE>(p, grid, lds, s);
        case 4: return pwr_launch<4, GATE>(p, grid, lds, s);
        case 5: return pwr_launch<5, GATE>(p, grid, lds, s);
        case 6: return pwr_launch<6, GATE>(p, grid, lds, s);
        case 7: return pwr_launch<7, GATE>(p, grid, lds, s);
        default: return pwr_launch<8, GATE>(p, grid, lds, s);
    }
}

int launch_pw_rgemm(const ConvDesc& d, hipStream_t s) {
    ORBIT_REQUIRE(pw_rgemm_supported(d), "pw_rgemm: unsupported convolution");
    PwrParams p;
    p.x = d.x, p.wf = d.w_frag, p.y = d.y, p.scale = d.scale, p.shift = d.shift, p.residual = d.residual, p.gate = d.gate;
    p.M = d.B * d.H * d.W, p.K = d.Cin, p.Cout = d.Cout, p.act = d.act;
    p.nchunk = d.Cin / 16;
    int T, wk;
    (void)pwr_plan(p.M, p.K, p.Cout, d.gate != nullptr, T, wk);
    p.wk = wk, p.cps = pwr_slice(p.nchunk, wk);
    p.n_groups = cdiv(cdiv(d.Cout, 16), T);
    p.m_tiles = cdiv(p.M, 32);
    p.fd_hw = make_fastdiv((unsigned)(d.H * d.W));
    p.fd_ng = make_fastdiv((unsigned)p.n_groups);
    const int grid = cdiv(p.m_tiles, 4 / wk) * p.n_groups;
    const size_t lds = wk > 1 ? (size_t)(4 / wk) * (wk - 1) * 2 * T * 64 * sizeof(f32x4) : 0;
    char name[48];
    snprintf(name, sizeof(name), "conv_pw_rgemm<%d,k%d%s>", T, wk, d.gate ? ",gate" : "");
    const double pix = (double)p.M;
    const int rec = prof_start(name, 2.0 * pix * d.Cout * d.Cin * d.prof_flop_scale,
                               4.0 * (pix * d.Cin + pix * d.Cout * (d.residual ? 2.0 : 1.0) + (double)d.Cout * d.Cin), s,
                               d.act == ORBIT_ACT_SILU ? pix * d.Cout : 0.0);
    const int rc = d.gate ? pwr_dispatch<true>(T, p, grid, lds, s) : pwr_dispatch<false>(T, p, grid, lds, s);
    prof_stop(rec, s);
    return rc;
}


}  // namespace orbit
