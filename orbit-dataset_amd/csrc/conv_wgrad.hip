// Convolution weight gradient and the weight repack of the data gradient, on the fp32 matrix cores.
//
// Reference: the autograd backward of every nn.Conv2d of the extractor / set encoder when the LITE step calls
// loss.backward() (single-step-learner.py:234; model/few_shot_recognisers.py:99-122 builds the graph).
//
//   wgrad   dW[co][(kh,kw,ci)] = sum_m dY[m][co] * Xg[m][(kh,kw,ci)]          m = (b, ho, wo)
//           GEMM with the LONG dimension (M = B*Ho*Wo, up to ~3e6) as the reduction: the output is tiny (Cout x K),
//           so the work is split along m over `splits` blocks per output tile; each block accumulates its rows with
//           v_mfma_f32_32x32x2_f32 and writes a partial tile, and a second kernel adds the partials in a fixed order
//           (deterministic, no atomics) while transposing to the OIHW layout of the parameter.
//           Both operands are "k-major" in memory (row m holds the contiguous channels), which is exactly the MFMA
//           operand order: LDS tiles are [m][64] and a lane fetches A[k = 2j + lane/32][i = lane%32] with a
//           conflict-free ds_read_b32 (32 consecutive floats per lane group).
//   dgrad   is the forward implicit-GEMM kernel (conv_igemm.hip) run on dY with the filter rotated by 180 degrees and
//           its in/out channels swapped; strided layers first spread dY over the input grid (zero insertion,
//           train_ops.hip). conv_pack_dgrad_weights builds that filter directly in the packed layout.
#include <algorithm>

#include "common.h"

namespace orbit {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

struct WgradParams {
    const float* x;
    const float* dy;
    const float* gate;  // [B][Cin] squeeze-excite gate multiplied into x on the fly (NHWC mode), or nullptr
    float* partial;  // [splits][Cout][NC]
    int B, H, W, Cin, Cout, KH, KW, stride, pad_t, pad_l, Ho, Wo;
    int M, NC;
    int co_tiles, n_tiles, splits, steps_per_split;
    FastDiv fd_howo, fd_wo;  // row decode m -> (b, ho, wo) without integer division (two per staged row and K-step)
};

constexpr int WG_BKM = 32;  // rows of m per K-step

// MODE 0: x NHWC, columns ordered (kh, kw, ci), Cin % 4 == 0.   MODE 1: x NCHW (stems), columns ordered (ci, kh, kw).
template <int MODE>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradParams p) {
    __shared__ __attribute__((aligned(16))) float As[2][WG_BKM][64];  // dY tile  [m][co]
    __shared__ __attribute__((aligned(16))) float Bs[2][WG_BKM][64];  // X  tile  [m][col]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;

    int bid = blockIdx.x;
    const int n_tile = bid % p.n_tiles;
    bid /= p.n_tiles;
    const int co_tile = bid % p.co_tiles;
    const int split = bid / p.co_tiles;
    const int co0 = co_tile * 64, n0 = n_tile * 64;
    const int m_begin = split * p.steps_per_split * WG_BKM;
    const int m_end = min(p.M, m_begin + p.steps_per_split * WG_BKM);
    const int nsteps = m_end > m_begin ? (m_end - m_begin + WG_BKM - 1) / WG_BKM : 0;
    const int HoWo = p.Ho * p.Wo;

    // ---- staging bookkeeping -------------------------------------------------------------------------------
    // dY (and x in MODE 0): thread owns float4 column c4 of rows (tid >> 4) and (tid >> 4) + 16
    const int c4 = tid & 15, lrow = tid >> 4;
    const bool a_col_ok = co0 + c4 * 4 < p.Cout;
    // MODE 0 column decode (fixed for the whole loop)
    int kh0 = 0, kw0 = 0, ci0 = 0;
    bool b_col_ok = false;
    // MODE 1: thread owns scalar column (tid & 63) of rows (tid >> 6) + 4 i
    int kh1 = 0, kw1 = 0, ci1 = 0;
    bool b1_col_ok = false;
    if (MODE == 0) {
        const int n = n0 + c4 * 4;
        b_col_ok = n < p.NC;
        const int tap = b_col_ok ? n / p.Cin : 0;
        ci0 = n - tap * p.Cin;
        kh0 = tap / p.KW, kw0 = tap - kh0 * p.KW;
    } else {
        const int n = n0 + (tid & 63);
        b1_col_ok = n < p.NC;
        const int kk = p.KH * p.KW;
        ci1 = b1_col_ok ? n / kk : 0;
        const int r = n - ci1 * kk;
        kh1 = r / p.KW, kw1 = r - kh1 * p.KW;
    }

    f32x4 a_stage[2];
    f32x4 b_stage[2];  // MODE 1 reuses these as 8 scalars
    f32x4 g_stage[2];  // gate quads of the two staged rows (multiplied in at the LDS store, so the loads share one wait)

    auto load_step = [&](int step) {
        const int mb = m_begin + step * WG_BKM;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = mb + lrow + 16 * i;
            f32x4 va = {0.f, 0.f, 0.f, 0.f}, vb = {0.f, 0.f, 0.f, 0.f};
            g_stage[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (m < m_end) {
                if (a_col_ok) va = *reinterpret_cast<const f32x4*>(p.dy + (size_t)m * p.Cout + co0 + c4 * 4);
                if (MODE == 0 && b_col_ok) {
                    const int b = (int)fdiv((unsigned)m, p.fd_howo);
                    const int r = m - b * HoWo;
                    const int ho = (int)fdiv((unsigned)r, p.fd_wo), wo = r - ho * p.Wo;
                    const int hi = ho * p.stride - p.pad_t + kh0, wi = wo * p.stride - p.pad_l + kw0;
                    if ((unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W)
                        vb = *reinterpret_cast<const f32x4*>(p.x + (((size_t)b * p.H + hi) * p.W + wi) * p.Cin + ci0);
                    if (p.gate) g_stage[i] = *reinterpret_cast<const f32x4*>(p.gate + (size_t)b * p.Cin + ci0);
                }
            }
            a_stage[i] = va;
            if (MODE == 0) b_stage[i] = vb;
        }
        if (MODE == 1) {
            const size_t plane = (size_t)p.H * p.W;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int m = mb + (tid >> 6) + 4 * i;
                float v = 0.f;
                if (m < m_end && b1_col_ok) {
                    const int b = (int)fdiv((unsigned)m, p.fd_howo);
                    const int r = m - b * HoWo;
                    const int ho = (int)fdiv((unsigned)r, p.fd_wo), wo = r - ho * p.Wo;
                    const int hi = ho * p.stride - p.pad_t + kh1, wi = wo * p.stride - p.pad_l + kw1;
                    if ((unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W)
                        v = p.x[((size_t)b * p.Cin + ci1) * plane + (size_t)hi * p.W + wi];
                }
                b_stage[i >> 2][i & 3] = v;
            }
        }
    };
    auto store_step = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *reinterpret_cast<f32x4*>(&As[buf][lrow + 16 * i][c4 * 4]) = a_stage[i];
            if (MODE == 0)
                *reinterpret_cast<f32x4*>(&Bs[buf][lrow + 16 * i][c4 * 4]) = p.gate ? b_stage[i] * g_stage[i] : b_stage[i];
        }
        if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) Bs[buf][(tid >> 6) + 4 * i][tid & 63] = b_stage[i >> 2][i & 3];
        }
    };

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    if (nsteps > 0) {
        load_step(0);
        store_step(0);
    }
    __syncthreads();
    for (int st = 0; st < nsteps; ++st) {
        const int cur = st & 1;
        if (st + 1 < nsteps) load_step(st + 1);
#pragma unroll
        for (int j = 0; j < WG_BKM / 2; ++j) {
            const float a = As[cur][2 * j + lh][wm + l31];
            const float b = Bs[cur][2 * j + lh][wn + l31];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        if (st + 1 < nsteps) store_step(cur ^ 1);
        __syncthreads();
    }

    // C/D layout: col (n) = lane & 31, row (co) = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    const int n = n0 + wn + l31;
    if (n < p.NC) {
        float* out = p.partial + (size_t)split * p.Cout * p.NC;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + wm + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (co < p.Cout) out[(size_t)co * p.NC + n] = acc[r];
        }
    }
}

// dw (OIHW) = sum over splits; MODE 0 columns (kh,kw,ci) -> (ci,kh,kw). 16 outputs per block, 16 lanes per output take
// every 16th split, combined through LDS in a fixed order (deterministic; a single thread walking up to ~2000 splits is
// a latency chain of hundreds of microseconds)
__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(const float* __restrict__ partial, int splits, int Cout,
                                                                int NC, int Cin, int KH, int KW, int mode,
                                                                float* __restrict__ dw) {
    __shared__ float sh[16][16];
    const size_t total = (size_t)Cout * NC;
    const int ol = threadIdx.x & 15, ln = threadIdx.x >> 4;
    const size_t i = (size_t)blockIdx.x * 16 + ol;
    float s = 0.f;
    if (i < total) {
#pragma unroll 4
        for (int k = ln; k < splits; k += 16) s += partial[(size_t)k * total + i];
    }
    sh[ln][ol] = s;
    __syncthreads();
    if (ln != 0 || i >= total) return;
    for (int j = 1; j < 16; ++j) s += sh[j][ol];
    if (mode == 1) {
        dw[i] = s;
    } else {
        const int co = (int)(i / NC), n = (int)(i % NC);
        const int tap = n / Cin, ci = n - tap * Cin;
        dw[((size_t)co * Cin + ci) * KH * KW + tap] = s;
    }
}

// the same for up to WGRAD_REDUCE_JOBS layers in one launch (blockIdx.y = layer): a network's reverse pass keeps every layer's
// partial tiles and reduces them all at its end - 33 launches of ~10 us, each a latency chain on the step's critical path,
// become one
__global__ __launch_bounds__(256) void conv_wgrad_reduce_batched_kernel(WgradReduceJobs jobs) {
    __shared__ float sh[16][16];
    int k = 0;
    while (k + 1 < jobs.n && blockIdx.x >= jobs.first_block[k + 1]) ++k;  // (<= 48 layers: a uniform scalar walk)
    const WgradReduceJob& j = jobs.j[k];
    const size_t total = (size_t)j.Cout * j.NC;
    const int ol = threadIdx.x & 15, ln = threadIdx.x >> 4;
    const size_t i = (size_t)(blockIdx.x - jobs.first_block[k]) * 16 + ol;
    float s = 0.f;
    if (i < total) {
#pragma unroll 4
        for (int sp = ln; sp < j.splits; sp += 16) s += j.partial[(size_t)sp * total + i];
    }
    sh[ln][ol] = s;
    __syncthreads();
    if (ln != 0 || i >= total) return;
    for (int q = 1; q < 16; ++q) s += sh[q][ol];
    if (j.mode == 1) {
        j.dw[i] = s;
    } else {
        const int co = (int)(i / j.NC), n = (int)(i % j.NC);
        const int tap = n / j.Cin, ci = n - tap * j.Cin;
        j.dw[((size_t)co * j.Cin + ci) * j.KH * j.KW + tap] = s;
    }
}

// ---- thin pointwise layers: one of (Cin, Cout) <= 48 channels, millions of rows --------------------------------------------
// EfficientNet's high-resolution expansions and projections (16 -> 96 at 112x112, 96 -> 24 / 24 <-> 144 at 56x56, 40 <-> 240 at
// 28x28; timm InvertedResidual conv_pw / conv_pwl behind model/feature_extractors.py:39-43) have a filter gradient of a few
// thousand numbers summed over 0.16 - 2.5 million rows. The 64x64-tile kernel above pads the thin side to 64 columns: for
// 16 -> 96 only 19 % of its MFMA work and LDS traffic is real and the launch takes 455 us for 1.12 GB of operands (HBM time
// ~250 us); 96 -> 24 at 56x56 takes 233 us for 300 MB. Here a WAVE owns a run of rows and keeps the whole [fat tiles of its
// group] x [thin tiles] gradient block in accumulators (v_mfma_f32_16x16x4_f32: K = 4 rows per instruction, lane = (channel of
// a 16-channel tile, row of the step) for both operands, so operands go global -> register -> MFMA with one dword per lane and
// tile - no LDS stage, no padding beyond 16 channels). The fat side is cut into groups of 8 tiles (grid.y) when it has more.
// The four waves of a block are added in wave order through LDS, blocks through the reduce kernel below: deterministic.
struct WthinParams {
    const float* fat;    // [M][Cf]
    const float* thin;   // [M][Ct]
    const float* gate;   // [B][Cin] on the x side or nullptr
    float* partial;      // [blocks][Cout][Cin]
    int M, Cf, Ct, Cout, Cin, rows_per_wave, ftiles, tpg;  // tpg: fat tiles per group (<= 8), groups balanced
    FastDiv fd_hw;
};

// TF: fat tiles a wave holds (>= the group's tile count; a missing tile costs MFMAs on zeros but no loads). S = K-steps of
// 4 rows per loop iteration, chosen so that ~24-32 operand dwords per lane are in flight: with 2 + 1 tiles (32 -> 16 at
// 112x112) two steps per iteration left 6 loads in flight and the walk was a chain of 77 HBM round trips per wave.
template <int TF, int TT, bool FAT_IS_CO, bool GATE>
__global__ __launch_bounds__(256) void conv_wgrad_thin_kernel(const WthinParams p) {
    constexpr int S = TF + TT <= 4 ? 8 : TF + TT <= 8 ? 4 : 2;
    __shared__ f32x4 red[TF * TT * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 15, kq = lane >> 4;
    const int t0 = blockIdx.y * p.tpg;               // first fat tile of this group
    const int nv = min(p.tpg, p.ftiles - t0);        // its tiles (<= TF)
    const int wg = blockIdx.x * 4 + wave;
    const int m_begin = wg * p.rows_per_wave;
    const int m_end = min(p.M, m_begin + p.rows_per_wave);
    f32x4 acc[TF][TT];
#pragma unroll
    for (int t = 0; t < TF; ++t)
#pragma unroll
        for (int u = 0; u < TT; ++u) acc[t][u] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // per-lane channel validity (fixed for the whole walk)
    bool fok[TF], tok[TT];
#pragma unroll
    for (int t = 0; t < TF; ++t) fok[t] = t < nv && (t0 + t) * 16 + col < p.Cf;
#pragma unroll
    for (int u = 0; u < TT; ++u) tok[u] = u * 16 + col < p.Ct;
    const float* fbase = p.fat + (size_t)(t0 * 16 + col);
    const float* tbase = p.thin + col;
    // squeeze-excite gate of the x side: a lane's channels are fixed, so its gate values only change with the FRAME of its
    // row - kept in registers and reloaded when the frame changes (once per H*W rows)
    float gfat[TF], gthin[TT];
    int bcur = -1;
#pragma unroll
    for (int t = 0; t < TF; ++t) gfat[t] = 1.f;
#pragma unroll
    for (int u = 0; u < TT; ++u) gthin[u] = 1.f;
    for (int m0 = m_begin; m0 < m_end; m0 += 4 * S) {
        float fv[S][TF], tv[S][TT];
#pragma unroll
        for (int h = 0; h < S; ++h) {
            const int row = m0 + 4 * h + kq;
            const bool ok = row < m_end;
            const size_t r = (size_t)(ok ? row : m_begin);
            if (GATE) {
                const int b = (int)fdiv((unsigned)r, p.fd_hw);
                if (b != bcur) {
                    bcur = b;
                    const float* grow = p.gate + (size_t)b * p.Cin;
                    if (FAT_IS_CO) {
#pragma unroll
                        for (int u = 0; u < TT; ++u) gthin[u] = tok[u] ? grow[u * 16 + col] : 0.f;
                    } else {
#pragma unroll
                        for (int t = 0; t < TF; ++t) gfat[t] = fok[t] ? grow[(t0 + t) * 16 + col] : 0.f;
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < TF; ++t) {
                float v = fok[t] ? fbase[r * p.Cf + t * 16] : 0.f;
                if (GATE && !FAT_IS_CO) v *= gfat[t];   // the fat side is x
                fv[h][t] = ok ? v : 0.f;
            }
#pragma unroll
            for (int u = 0; u < TT; ++u) {
                float v = tok[u] ? tbase[r * p.Ct + u * 16] : 0.f;
                if (GATE && FAT_IS_CO) v *= gthin[u];   // the thin side is x
                tv[h][u] = ok ? v : 0.f;
            }
        }
#pragma unroll
        for (int h = 0; h < S; ++h)
#pragma unroll
            for (int t = 0; t < TF; ++t)
#pragma unroll
                for (int u = 0; u < TT; ++u)
                    acc[t][u] = FAT_IS_CO ? __builtin_amdgcn_mfma_f32_16x16x4f32(fv[h][t], tv[h][u], acc[t][u], 0, 0, 0)
                                          : __builtin_amdgcn_mfma_f32_16x16x4f32(tv[h][u], fv[h][t], acc[t][u], 0, 0, 0);
    }
    // the block's four waves, added in wave order
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int t = 0; t < TF; ++t)
#pragma unroll
                for (int u = 0; u < TT; ++u) {
                    f32x4* slot = &red[(t * TT + u) * 64 + lane];
                    *slot = w == 0 ? acc[t][u] : *slot + acc[t][u];
                }
        }
        __syncthreads();
    }
    // D[i = 4 * (lane / 16) + r][j = lane % 16]: i indexes the A operand's channels (co), j the B operand's (ci)
    float* out = p.partial + (size_t)blockIdx.x * p.Cout * p.Cin;
    for (int e = threadIdx.x; e < nv * TT * 64; e += 256) {
        const int l = e & 63, tu = e >> 6, t = tu / TT, u = tu - t * TT;
        const f32x4 v = red[e];
        const int fat_c0 = (t0 + t) * 16, thin_c0 = u * 16;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 4 * (l >> 4) + r, j = l & 15;
            const int co = FAT_IS_CO ? fat_c0 + i : thin_c0 + i;
            const int ci = FAT_IS_CO ? thin_c0 + j : fat_c0 + j;
            if (co < p.Cout && ci < p.Cin) out[(size_t)co * p.Cin + ci] = v[r];
        }
    }
}

// the thin kernel serves: NHWC pointwise layers (1x1, stride 1, no padding) with >= 2^18 rows (the 112- and 56-pixel maps of a
// 200-frame batch) whose thinner side has <= 2 tiles of 16 channels; blocks = row splits (<= 1024, >= 64 rows per wave).
// (Measured, 200 frames: 16 -> 96 @112 455 -> 224 us; with 3 thin tiles - 40 <-> 240 @28 - the 24 MFMAs per 4-row step bound
// the walk: 159-222 us against 70 us for the tiled kernel, which keeps those layers.)
static bool wgrad_thin_geometry(int M, int Cin, int Cout, int KH, int KW, int stride, int pad_t, int pad_l, int x_nchw, int& blocks,
                                int& rows_per_wave) {
    if (x_nchw || KH != 1 || KW != 1 || stride != 1 || pad_t != 0 || pad_l != 0 || M < (1 << 18)) return false;
    if (std::min(cdiv(Cin, 16), cdiv(Cout, 16)) > 2) return false;
    blocks = std::min(1024, cdiv(M, 4 * 64));
    rows_per_wave = cdiv(cdiv(M, blocks * 4), 32) * 32;  // whole loop iterations of every instantiation (4 * S rows, S <= 8)
    blocks = cdiv(M, rows_per_wave * 4);
    return true;
}

// number of m-splits for a layer: enough blocks to fill the chip, at least 8 K-steps per block
static void wgrad_geometry(int M, int Cout, int NC, int& co_tiles, int& n_tiles, int& splits, int& steps_per_split) {
    co_tiles = cdiv(Cout, 64), n_tiles = cdiv(NC, 64);
    const int tiles = co_tiles * n_tiles;
    const int total_steps = cdiv(M, WG_BKM);
    static const int target = [] {  // (ORBIT_WGRAD_BLOCKS: tuning experiments only)
        const char* e = getenv("ORBIT_WGRAD_BLOCKS");
        // 1536 (round 6; 2048 before): a quarter less partial-tile traffic (the split reductions move ~1 GB per LITE step),
        // same-box A/B 24.54 / 24.53 -> 24.41 / 24.23 ms per step (profiles/r06_lite_ab_fused_fronts.txt)
        return e && atoi(e) > 0 ? atoi(e) : 1536;
    }();
    splits = cdiv(target, tiles);
    const int max_splits = total_steps >= 8 ? total_steps / 8 : 1;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    steps_per_split = cdiv(total_steps, splits);
    splits = cdiv(total_steps, steps_per_split);
}

size_t conv_wgrad_scratch_floats(int B, int Cin, int Cout, int KH, int KW, int Ho, int Wo) {
    int ct, nt, sp, sps;
    const int NC = KH * KW * Cin;
    wgrad_geometry(B * Ho * Wo, Cout, NC, ct, nt, sp, sps);
    int tb = 0, rpw = 0;  // (the thin form of a pointwise layer: the caller's stride / padding are not known here - size for it)
    if (wgrad_thin_geometry(B * Ho * Wo, Cin, Cout, KH, KW, 1, 0, 0, 0, tb, rpw)) sp = std::max(sp, tb);
    return (size_t)sp * Cout * NC;
}

int launch_conv_wgrad_reduce_batched(WgradReduceJobs& jobs, int n, hipStream_t s) {
    ORBIT_REQUIRE(n >= 0 && n <= WGRAD_REDUCE_JOBS, "conv_wgrad_reduce_batched: %d jobs", n);
    if (n == 0) return ORBIT_OK;
    unsigned blocks = 0;
    for (int k = 0; k < n; ++k) {
        jobs.first_block[k] = blocks;
        blocks += (unsigned)(((size_t)jobs.j[k].Cout * jobs.j[k].NC + 15) / 16);
    }
    jobs.first_block[n] = blocks, jobs.n = n;
    conv_wgrad_reduce_batched_kernel<<<blocks, 256, 0, s>>>(jobs);
    ORBIT_LAUNCH_CHECK();
    return ORBIT_OK;
}

int launch_conv_wgrad(const float* x, int x_nchw, const float* dy, float* dw_oihw, int B, int H, int W, int Cin, int Cout,
                      int KH, int KW, int stride, int pad_t, int pad_l, int Ho, int Wo, float* scratch, hipStream_t s,
                      const float* gate, WgradReduceJob* defer) {
    ORBIT_REQUIRE(x && dy && dw_oihw && scratch, "conv_wgrad: null pointer");
    ORBIT_REQUIRE(!gate || !x_nchw, "conv_wgrad: the squeeze-excite gate needs the NHWC path");
    ORBIT_REQUIRE(Cout % 4 == 0, "conv_wgrad: Cout %% 4 != 0");
    ORBIT_REQUIRE(x_nchw || Cin % 4 == 0, "conv_wgrad: NHWC path needs Cin %% 4 == 0");
    WgradParams p;
    p.x = x, p.dy = dy, p.partial = scratch, p.gate = gate;
    p.B = B, p.H = H, p.W = W, p.Cin = Cin, p.Cout = Cout, p.KH = KH, p.KW = KW, p.stride = stride;
    p.pad_t = pad_t, p.pad_l = pad_l, p.Ho = Ho, p.Wo = Wo;
    p.M = B * Ho * Wo, p.NC = KH * KW * Cin;
    p.fd_howo = make_fastdiv((unsigned)(Ho * Wo)), p.fd_wo = make_fastdiv((unsigned)Wo);
    int thin_blocks = 0, thin_rpw = 0;
    if (wgrad_thin_geometry(p.M, Cin, Cout, KH, KW, stride, pad_t, pad_l, x_nchw, thin_blocks, thin_rpw)) {
        const int tci = cdiv(Cin, 16), tco = cdiv(Cout, 16);
        const bool fat_is_co = tco >= tci;
        WthinParams q;
        q.fat = fat_is_co ? dy : x, q.thin = fat_is_co ? x : dy, q.gate = gate, q.partial = scratch;
        q.M = p.M, q.Cf = fat_is_co ? Cout : Cin, q.Ct = fat_is_co ? Cin : Cout, q.Cout = Cout, q.Cin = Cin;
        q.rows_per_wave = thin_rpw, q.ftiles = fat_is_co ? tco : tci, q.fd_hw = p.fd_howo;
        const int tt = fat_is_co ? tci : tco;
        const int groups = cdiv(q.ftiles, 8);
        q.tpg = cdiv(q.ftiles, groups);
        const dim3 grid(thin_blocks, groups);
        const int rec = prof_start("conv_wgrad_thin", 2.0 * p.M * Cout * Cin,
                                   4.0 * ((double)p.M * Cin + (double)p.M * Cout + (double)Cout * Cin), s);
#define ORBIT_WTHIN(TF_, TT_)                                                                                \
    do {                                                                                                     \
        if (fat_is_co) {                                                                                     \
            if (gate) conv_wgrad_thin_kernel<TF_, TT_, true, true><<<grid, 256, 0, s>>>(q);                  \
            else conv_wgrad_thin_kernel<TF_, TT_, true, false><<<grid, 256, 0, s>>>(q);                      \
        } else {                                                                                             \
            if (gate) conv_wgrad_thin_kernel<TF_, TT_, false, true><<<grid, 256, 0, s>>>(q);                 \
            else conv_wgrad_thin_kernel<TF_, TT_, false, false><<<grid, 256, 0, s>>>(q);                     \
        }                                                                                                    \
    } while (0)
#define ORBIT_WTHIN_TF(TT_)                                                                                  \
    do {                                                                                                     \
        if (q.tpg <= 2) ORBIT_WTHIN(2, TT_);                                                                 \
        else if (q.tpg <= 4) ORBIT_WTHIN(4, TT_);                                                            \
        else if (q.tpg <= 6) ORBIT_WTHIN(6, TT_);                                                            \
        else ORBIT_WTHIN(8, TT_);                                                                            \
    } while (0)
        if (tt == 1) ORBIT_WTHIN_TF(1);
        else ORBIT_WTHIN_TF(2);
#undef ORBIT_WTHIN_TF
#undef ORBIT_WTHIN
        prof_stop(rec, s);
        ORBIT_LAUNCH_CHECK();
        if (defer) {
            *defer = WgradReduceJob{scratch, dw_oihw, thin_blocks, Cout, p.NC, Cin, 1, 1, 0};
            return ORBIT_OK;
        }
        const size_t total = (size_t)Cout * p.NC;
        conv_wgrad_reduce_kernel<<<(int)((total + 15) / 16), 256, 0, s>>>(scratch, thin_blocks, Cout, p.NC, Cin, 1, 1, 0, dw_oihw);
        ORBIT_LAUNCH_CHECK();
        return ORBIT_OK;
    }
    wgrad_geometry(p.M, Cout, p.NC, p.co_tiles, p.n_tiles, p.splits, p.steps_per_split);
    const int grid = p.co_tiles * p.n_tiles * p.splits;
    // algorithmic work: 2*M*Cout*K flops; bytes = input + output gradient once, filter gradient once
    const int rec = prof_start(x_nchw ? "conv_wgrad<nchw>" : "conv_wgrad<nhwc>", 2.0 * p.M * Cout * p.NC,
                               4.0 * ((double)B * H * W * Cin + (double)p.M * Cout + (double)Cout * p.NC), s);
    if (x_nchw) conv_wgrad_kernel<1><<<grid, 256, 0, s>>>(p);
    else conv_wgrad_kernel<0><<<grid, 256, 0, s>>>(p);
    prof_stop(rec, s);
    ORBIT_LAUNCH_CHECK();
    if (defer) {
        *defer = WgradReduceJob{scratch, dw_oihw, p.splits, Cout, p.NC, Cin, KH, KW, x_nchw ? 1 : 0};
        return ORBIT_OK;
    }
    const size_t total = (size_t)Cout * p.NC;
    const int rblocks = (int)((total + 15) / 16);
    conv_wgrad_reduce_kernel<<<rblocks, 256, 0, s>>>(scratch, p.splits, Cout, p.NC, Cin, KH, KW, x_nchw ? 1 : 0, dw_oihw);
    ORBIT_LAUNCH_CHECK();
    return ORBIT_OK;
}

// ---- dgrad filter: packed [n = ci][k = (kh', kw', co)] with value W[co][ci][KH-1-kh'][KW-1-kw'], in the geometry the
// forward kernel expects for a convolution with Cin' = Cout, Cout' = Cin ------------------------------------------
__global__ __launch_bounds__(256) void conv_pack_dgrad_kernel(const float* __restrict__ w, float* __restrict__ wp,
                                                              int Cin, int Cout, int KH, int KW, int cin_pad, int KT,
                                                              int cout_pad) {
    const size_t total = (size_t)cout_pad * KT;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int n = (int)(i / KT), k = (int)(i % KT);  // n = ci (output channel of the dgrad conv)
        const int tap = k / cin_pad, co = k % cin_pad;  // co = input channel of the dgrad conv
        float v = 0.f;
        if (n < Cin && tap < KH * KW && co < Cout) {
            const int kh = KH - 1 - tap / KW, kw = KW - 1 - tap % KW;
            v = w[(((size_t)co * Cin + n) * KH + kh) * KW + kw];
        }
        wp[i] = v;
    }
}

size_t conv_dgrad_packed_floats(int Cin, int Cout, int KH, int KW) { return conv_packed_floats(Cout, Cin, KH, KW, 0); }

int conv_pack_dgrad_weights(const float* w_oihw, float* w_packed, int Cin, int Cout, int KH, int KW, hipStream_t s) {
    const ConvPackGeom g = conv_pack_geom(Cout, Cin, KH, KW, 0);
    const size_t total = (size_t)g.cout_pad * g.kt;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    conv_pack_dgrad_kernel<<<blocks, 256, 0, s>>>(w_oihw, w_packed, Cin, Cout, KH, KW, g.cin_pad, g.kt, g.cout_pad);
    ORBIT_LAUNCH_CHECK();
    return ORBIT_OK;
}

// dX (+= accumulate) of a convolution. `up` is scratch of B*H*W*Cout floats, only touched when stride > 1.
int launch_conv_dgrad(const float* dy, const float* w_dgrad_packed, const float* accumulate, float* dx, float* up, int B,
                      int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad_t, int pad_l, int Ho, int Wo,
                      hipStream_t s) {
    ORBIT_REQUIRE(dy && w_dgrad_packed && dx, "conv_dgrad: null pointer");
    ORBIT_REQUIRE(pad_t <= KH - 1 && pad_l <= KW - 1, "conv_dgrad: padding larger than the filter");
    const float* src = dy;
    int Hs = Ho, Ws = Wo;
    if (stride > 1) {
        ORBIT_REQUIRE(up, "conv_dgrad: strided layers need the upsampling scratch");
        if (int rc = launch_upsample_zero(dy, up, B, H, W, Cout, stride, Ho, Wo, s)) return rc;
        src = up, Hs = H, Ws = W;
    }
    ConvDesc d;
    d.x = src, d.w_packed = w_dgrad_packed, d.y = dx, d.scale = nullptr, d.shift = nullptr;
    d.residual = accumulate, d.gate = nullptr;
    d.B = B, d.H = Hs, d.W = Ws, d.Cin = Cout, d.Cout = Cin, d.KH = KH, d.KW = KW, d.stride = 1;
    d.pad_t = KH - 1 - pad_t, d.pad_l = KW - 1 - pad_l, d.Ho = H, d.Wo = W;
    d.act = ORBIT_ACT_NONE, d.pool2 = 0, d.x_nchw = 0;
    d.prof_flop_scale = 1.0f / (float)(stride * stride);
    return launch_conv(d, s);
}

}  // namespace orbit

using namespace orbit;

extern "C" {

int orbit_op_conv2d_wgrad(const float* x, int x_nchw, const float* dy, float* dw, int B, int H, int W, int Cin, int Cout,
                          int KH, int KW, int stride, int pad_top, int pad_left, int Ho, int Wo, orbit_stream_t stream) {
    ORBIT_REQUIRE(x && dy && dw, "op_conv2d_wgrad: null pointer");
    ORBIT_REQUIRE(B > 0 && KH > 0 && KW > 0 && stride > 0 && Ho > 0 && Wo > 0, "op_conv2d_wgrad: bad geometry");
    hipStream_t s = (hipStream_t)stream;
    float* scratch = nullptr;
    ORBIT_HIP_CHECK(hipMallocAsync(reinterpret_cast<void**>(&scratch),
                                   conv_wgrad_scratch_floats(B, Cin, Cout, KH, KW, Ho, Wo) * sizeof(float), s));
    const int rc = launch_conv_wgrad(x, x_nchw, dy, dw, B, H, W, Cin, Cout, KH, KW, stride, pad_top, pad_left, Ho, Wo,
                                     scratch, s);
    (void)hipFreeAsync(scratch, s);
    return rc;
}

int orbit_op_conv2d_wgrad_gated(const float* x, const float* gate, const float* dy, float* dw, int B, int H, int W, int Cin,
                                int Cout, orbit_stream_t stream) {
    ORBIT_REQUIRE(x && gate && dy && dw, "op_conv2d_wgrad_gated: null pointer");
    ORBIT_REQUIRE(B > 0 && H > 0 && W > 0, "op_conv2d_wgrad_gated: bad geometry");
    hipStream_t s = (hipStream_t)stream;
    float* scratch = nullptr;
    ORBIT_HIP_CHECK(hipMallocAsync(reinterpret_cast<void**>(&scratch), conv_wgrad_scratch_floats(B, Cin, Cout, 1, 1, H, W) * sizeof(float),
                                   s));
    const int rc = launch_conv_wgrad(x, 0, dy, dw, B, H, W, Cin, Cout, 1, 1, 1, 0, 0, H, W, scratch, s, gate);
    (void)hipFreeAsync(scratch, s);
    return rc;
}

int orbit_op_conv2d_dgrad(const float* dy, const float* w, const float* accumulate, float* dx, int B, int H, int W,
                          int Cin, int Cout, int KH, int KW, int stride, int pad_top, int pad_left, int Ho, int Wo,
                          orbit_stream_t stream) {
    ORBIT_REQUIRE(dy && w && dx, "op_conv2d_dgrad: null pointer");
    ORBIT_REQUIRE(B > 0 && KH > 0 && KW > 0 && stride > 0 && Ho > 0 && Wo > 0, "op_conv2d_dgrad: bad geometry");
    ORBIT_REQUIRE(Cin % 4 == 0 && Cout % 4 == 0, "op_conv2d_dgrad: Cin and Cout must be multiples of 4");
    hipStream_t s = (hipStream_t)stream;
    float* tmp = nullptr;
    const size_t npack = conv_dgrad_packed_floats(Cin, Cout, KH, KW);
    const size_t nup = stride > 1 ? (size_t)B * H * W * Cout : 0;
    ORBIT_HIP_CHECK(hipMallocAsync(reinterpret_cast<void**>(&tmp), (npack + nup) * sizeof(float), s));
    int rc = conv_pack_dgrad_weights(w, tmp, Cin, Cout, KH, KW, s);
    if (rc == ORBIT_OK)
        rc = launch_conv_dgrad(dy, tmp, accumulate, dx, stride > 1 ? tmp + npack : nullptr, B, H, W, Cin, Cout, KH, KW,
                               stride, pad_top, pad_left, Ho, Wo, s);
    (void)hipFreeAsync(tmp, s);
    return rc;
}

}  // extern "C"
