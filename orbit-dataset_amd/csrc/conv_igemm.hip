// Implicit-GEMM convolution on the gfx950 fp32 matrix cores (v_mfma_f32_32x32x2_f32).
//
// This kernel carries >99% of the hot path's arithmetic: every dense convolution of the feature
// extractors (reference: model/feature_extractors.py:37-79 -> timm / torchvision-layout networks run by
// model/few_shot_recognisers.py:99-153) and of the set encoder (model/set_encoders.py:81-120).
//
//   GEMM view   C[m][n] = sum_k A[m][k] * Wp[n][k]
//               m = output pixel (b, ho, wo)   n = output channel   k = (kh, kw, ci), ci fastest
//   A is never materialised: each K-tile of 32 is gathered from the NHWC activation tensor (32
//   consecutive input channels of one filter tap = one 128-byte segment per row), or, for the 3-channel
//   network stems, element-wise from the NCHW frames.
//   Epilogue (fused, so activations make exactly one HBM round trip per layer):
//               y = act(acc * scale[n] + shift[n] + residual[m][n])      folded BatchNorm (+FiLM), skip add
//               optional 2x2/2 max-pool: rows are enumerated window-major so that the four members of
//               a pooling window are the four consecutive accumulator rows a lane already holds.
//   Prologue:   optional per-(frame, input-channel) squeeze-excite gate multiplied into A.
//
// Tiling: 256 threads = 4 waves; block tile BM x BN x BK, wave grid WGM x WGN x WGK (WGK > 1: the waves of a K-group
// split the K-tile's k-groups and their partial tiles are summed through LDS), each wave owns
// (BM/WGM/32) x (BN/WGN/32) accumulator tiles of 32x32 (16 VGPRs each); layers with very few output tiles are also
// split over K across blocks (conv_splitk + conv_splitk_reduce_kernel). LDS rows are padded to BK + 4 floats:
// the fragment reads are conflict-free ds_read_b128 (lane (i, h) reads k = 8g+4h .. +3 of row i, and
// register kk of the read feeds MFMA kk, so lanes 0-31 / 32-63 supply k = 8g+kk / 8g+4+kk).
// Global->LDS staging goes through registers and is software-pipelined one K-tile ahead (the f32 MFMA
// issues once per 64 cycles per SIMD, so one tile of prefetch hides L2/HBM latency). Addressing is division-free:
// one 64-bit pointer per staged row + a wave-uniform tap/channel walk in SGPRs; pointwise convs (PW) are a
// predicate-free specialisation (see the kernel's template comment).
// fp32 in, fp32 accumulate: bit-equivalent to an fmaf chain, which is what the 1e-3 logit parity
// target needs (no TF32-class path exists on gfx950).
#include <algorithm>
#include <cstdlib>
#include <vector>
#include "common.h"

namespace orbit {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

// K-tile width BK is a template parameter (8, 16 or 32): the tile of a 1x1/3x3 conv must lie inside one filter
// tap, so EfficientNet's narrow layers (Cin = 16, 24, 40, 80, 112, 144, 240) get an exact-fit BK instead of a
// zero-padded 32. LDS rows are padded to BK + 4 floats: for 12/20/36-float strides the 16-lane groups of a
// ds_read_b128 land on 16 distinct 16-byte bank slots (conflict-free).

struct ConvParams {
    const float* x;
    const float* w;
    float* y;
    const float* scale;
    const float* shift;
    const float* residual;
    const float* gate;
    int B, H, W, Cin, Cout, KH, KW, stride, pad_t, pad_l, Ho, Wo;
    int HoP, WoP;   // pooled output dims (pool2)
    int M;          // GEMM rows (pool2: B*HoP*WoP*4)
    int KT;         // padded K
    int cin_pad;    // per-tap padded Cin (vector mode)
    int act;
    int m_tiles, n_tiles;
    FastDiv fd_per, fd_wo;   // / (Ho*Wo), / Wo  (pool2: / (HoP*WoP), / WoP)
    FastDiv fd_cin, fd_kw;   // stem mode: / Cin, / KW
    float prof_flop_scale;
    int ksplit, kt_per_split;  // split-K: blockIdx = split * tiles + tile; raw partial tiles go to part[split][M][Cout]
    float* part;
    float* stats;              // [m_tiles][2][Cout] column sums / sums of squares of the raw outputs (train-mode BatchNorm) or null
    float* y_raw;              // dual write (training tape under running-statistics BatchNorm): the RAW conv output goes
                               // here, y receives act(raw * scale + shift + residual); null = single output
};

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == ORBIT_ACT_RELU) return fmaxf(v, 0.f);
    if (act == ORBIT_ACT_SILU) return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));  // v_exp_f32 + v_rcp_f32, ~1 ulp each
    return v;
}

// XCD-aware remap: hardware places block i on XCD i % 8; give each XCD a contiguous run of logical
// tiles so that the n-tiles sharing an A row-panel hit the same L2 (bijective for any grid size).
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, slot = bid >> 3;
    const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + slot;
}

template <bool POOL2>
__device__ __forceinline__ void decode_row(const ConvParams& p, int m, int& b, int& ho, int& wo) {
    if (POOL2) {
        const int win = m >> 2, q = m & 3;
        b = (int)fdiv((unsigned)win, p.fd_per);
        const int r = win - b * (p.HoP * p.WoP);
        const int hp = (int)fdiv((unsigned)r, p.fd_wo), wp = r - hp * p.WoP;
        ho = hp * 2 + (q >> 1);
        wo = wp * 2 + (q & 1);
    } else {
        b = (int)fdiv((unsigned)m, p.fd_per);
        const int r = m - b * (p.Ho * p.Wo);
        ho = (int)fdiv((unsigned)r, p.fd_wo);
        wo = r - ho * p.Wo;
    }
}

// MODE 0: NHWC activations, Cin % 4 == 0 (float4 gathers).  MODE 1: NCHW frames, tiny Cin (stems).
// PW: pointwise (1x1, no padding) convolution - every EfficientNet conv but the stem: no tap walk, no bounds tests.
//
// Addressing is split into a per-thread part fixed for the whole K loop (one 64-bit pointer per staged row, computed in
// the prologue) and a wave-uniform part that walks K (filter tap + channel offset, kept in SGPRs and advanced
// incrementally - no division, no per-tile 64-bit multiply): a staged load costs one 64-bit add.
// WGK: waves along K. The 4 waves of a block form a WGM x WGN x WGK grid; with WGK > 1 the waves of a K-group share the
// staged K-tile, each takes every WGK-th 8-deep k-group of it, and the partial accumulators are summed through LDS in
// the epilogue. That buys 2-4x more (smaller) output tiles for layers whose 64x64 tiling leaves the chip short of blocks
// (M = 9 800 rows in EfficientNet's 7x7 stages, 1 800-7 200 in resnet18 @84's layer3/4).
template <int BM, int BN, int WGM, int WGN, int WGK, int BK, int MODE, bool POOL2, bool GATE, bool PW, bool UL>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvParams p) {
    static_assert(WGM * WGN * WGK == 4, "4 waves per block");
    static_assert(WGK == 1 || !POOL2, "the fused max-pool is not linear in the K-split partial sums");
    static_assert(BK == 8 || BK == 16 || BK == 32, "BK");
    static_assert(MODE == 0 || BK == 32, "the stem gather is written for BK = 32");
    static_assert(!(PW && (MODE == 1 || POOL2)), "pointwise specialisation: NHWC, no fused pooling");
    static_assert(!(GATE && (MODE == 1 || POOL2)), "the squeeze-excite gate prologue: NHWC, no fused pooling");
    constexpr int LDS_STRIDE = BK + 4;
    constexpr int WM = BM / WGM, WN = BN / WGN;  // wave tile
    constexpr int TM = WM / 32, TN = WN / 32;    // 32x32 accumulator tiles per wave
    static_assert(TM >= 1 && TN >= 1, "wave tile must hold at least one 32x32 MFMA tile");
    constexpr int TPR = BK / 4;                  // threads (float4s) per tile row
    constexpr int RPP = 256 / TPR;               // tile rows filled per pass of the 256 threads
    constexpr int AR = (BM + RPP - 1) / RPP;     // float4 A loads per thread per K-tile (vector mode)
    constexpr int BR = (BN + RPP - 1) / RPP;     // float4 B loads per thread per K-tile
    constexpr int KPT = BM / 8;                  // scalar A loads per thread per K-tile (stem mode)
    constexpr int NG = BK / 8;                   // k-groups of 8 per K-tile
    static_assert(NG % WGK == 0, "K-split needs BK / 8 divisible by the number of K-waves");
    constexpr int NGW = NG / WGK;                // k-groups per wave per K-tile

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                            // [2][BM][LDS_STRIDE]
    float* Bs = smem + 2 * BM * LDS_STRIDE;      // [2][BN][LDS_STRIDE]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wk = wave / (WGM * WGN), wmn = wave % (WGM * WGN);
    const int wm = (wmn / WGN) * WM, wn = (wmn % WGN) * WN;

    // One output tile per block. (A persistent form - grid sized to the co-resident blocks, each block walking several
    // tiles and requesting the next tile's first K-tile before its epilogue - was measured on MI355X: 0..-16 % on every
    // layer of resnet18 and efficientnet_b0; the hardware dispatcher's dynamic placement beats the static walk.)
    const int ntiles = p.m_tiles * p.n_tiles;
    const int split = p.ksplit > 1 ? blockIdx.x / ntiles : 0;  // consecutive blocks share a K range (and its filter rows)
    const int tile = xcd_remap(p.ksplit > 1 ? blockIdx.x - split * ntiles : blockIdx.x, ntiles);
    const int m0 = (tile / p.n_tiles) * BM;
    const int n0 = (tile % p.n_tiles) * BN;
    const int kt0 = split * p.kt_per_split;  // first K-tile of this block

    // ---- per-thread gather bookkeeping (rows are fixed for the whole K loop) ----
    // vector mode: thread owns float4 column c4 of rows lrow + RPP*i; a_ptr[i] points at (tap (0,0), channel c4*4) of
    //              the row's receptive field (it may lie outside the tensor for padded rows - those are never loaded)
    // stem mode:   thread owns KPT consecutive k of row tid % BM
    // this thread's float4 column / first row of the tile. ds_write_b128 is serviced in groups of 8 CONTIGUOUS lanes over a
    // 128-byte bank row (MI355X_MICROARCH.md, LDS): with BK = 32 the 8 lanes of a group fill one tile row (conflict-free);
    // with BK = 16 / 8 they cover 2 / 4 rows whose 16-byte slots must not collide modulo 8 - adjacent rows of the
    // 20- / 12-float row stride do (the 0.30-0.35 conflicts per LDS cycle of profiles/r02_conv_sq_counters.txt), rows 4 / 2
    // apart do not, so the row index is a bit permutation of tid / TPR inside every group of 8 rows.
    const int c4 = tid % TPR;
    const int jrow = tid / TPR;
    const int lrow = BK == 16 ? ((jrow & ~7) | ((jrow & 1) << 2) | ((jrow >> 1) & 3))
                   : BK == 8  ? ((jrow & ~7) | ((jrow & 3) << 1) | ((jrow >> 2) & 1))
                              : jrow;
    const float* a_ptr[MODE == 0 ? AR : 1];
    int a_hi0[MODE == 0 ? AR : 1], a_wi0[MODE == 0 ? AR : 1];
    bool stem_fast = false;                                              // stem mode: see the interior fast path below
    const float* stem_row = nullptr;
    int* ktab = reinterpret_cast<int*>(smem + 2 * (BM + BN) * LDS_STRIDE);  // [KT] element offset of k inside a patch
    const float* g_ptr[GATE ? AR : 1];
    // wave-uniform K walk of the staging loads (tiles are loaded strictly in order): channel offset inside the tap,
    // tap coordinates, element offset of the tap in the input, element offset in the packed filter row
    int ld_ci = 0, ld_kh = 0, ld_kw = 0, ld_tap = 0, ld_k = kt0 * BK;
    if (MODE == 0 && kt0 != 0) {  // split-K: start the walk at K-tile kt0 (one wave-uniform division per block)
        const int cpt = p.cin_pad / BK, tap = kt0 / cpt;
        ld_ci = (kt0 - tap * cpt) * BK;
        ld_kh = tap / p.KW, ld_kw = tap - ld_kh * p.KW;
        ld_tap = (ld_kh * p.W + ld_kw) * p.Cin;
    }

    if (MODE == 0) {
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            // pointwise + UL: rows beyond M are clamped to the last row instead of predicated - what they compute is never
            // stored, and the staging path then has no predicate, mask or select at all
            const int m = (PW && UL) ? min(m0 + lrow + RPP * i, p.M - 1) : m0 + lrow + RPP * i;
            const bool ok = m < p.M && (BM % RPP == 0 || lrow + RPP * i < BM);
            int b = 0, ho = 0, wo = 0;
            long off;
            if (PW && p.stride == 1) {  // NHWC rows of a stride-1 pointwise conv ARE the GEMM rows
                off = (long)m * p.Cin;
                if (GATE) b = (int)fdiv((unsigned)(ok ? m : 0), p.fd_per);
                a_hi0[i] = ok ? 0 : -(1 << 28);
                a_wi0[i] = 0;
            } else {
                if (ok) decode_row<POOL2>(p, m, b, ho, wo);
                const int hi0 = ho * p.stride - p.pad_t, wi0 = wo * p.stride - p.pad_l;
                off = (((long)b * p.H + hi0) * p.W + wi0) * p.Cin;
                a_hi0[i] = ok ? hi0 : -(1 << 28);  // invalid rows fail the bounds test
                a_wi0[i] = wi0;
            }
            a_ptr[i] = p.x + (ok ? off : 0) + c4 * 4;
            if (GATE) g_ptr[i] = p.gate + (size_t)b * p.Cin + c4 * 4;
        }
    } else {
        const int m = m0 + (tid % BM);
        int b = 0, ho = 0, wo = 0;
        const bool ok = m < p.M;
        if (ok) decode_row<POOL2>(p, m, b, ho, wo);
        a_ptr[0] = p.x + (size_t)b * p.Cin * p.H * p.W;
        a_hi0[0] = ok ? ho * p.stride - p.pad_t : -(1 << 28);
        a_wi0[0] = wo * p.stride - p.pad_l;
        // Interior fast path. The per-element (ci, kh, kw) decode + bounds tests + address arithmetic of the gather made the
        // stems instruction-bound (~12 VALU + a scalar decode per 4-byte load). A wave whose 64 rows all have their whole
        // receptive field inside the image instead adds a per-k offset - looked up in a table the block builds once in LDS -
        // to one per-row pointer; no decode, no bounds test (entries of the K padding point at the row's first pixel: the
        // packed filter is zero there).
        const bool inside = ok && a_hi0[0] >= 0 && a_hi0[0] + p.KH <= p.H && a_wi0[0] >= 0 && a_wi0[0] + p.KW <= p.W;
        stem_fast = BM == 64 && p.KT <= 256 && __all(inside);  // measured: +2..7 % at BM 64, -3 % at 128
        stem_row = a_ptr[0] + (inside ? a_hi0[0] * p.W + a_wi0[0] : 0);
        if (p.KT <= 256) {
            if (tid < p.KT) {
                const int k = tid;
                const int tap = (int)fdiv((unsigned)k, p.fd_cin), ci = k - tap * p.Cin;
                const int kh = (int)fdiv((unsigned)tap, p.fd_kw), kw = tap - kh * p.KW;
                ktab[k] = k < p.KH * p.KW * p.Cin ? ci * p.H * p.W + kh * p.W + kw : 0;
            }
            __syncthreads();
        }
    }
    const bool b_row_ok = BN % RPP == 0 || lrow < BN;  // BK = 8 with BN = 64: half of the threads stage B (BN % 8 == 0: the
                                                       // row permutation stays inside groups of 8 rows)
    const float* b_ptr = p.w + (size_t)(n0 + (b_row_ok ? lrow : 0)) * p.KT + c4 * 4;

    constexpr int NAS = MODE == 0 ? AR : KPT / 4;
    f32x4 a_stage[NAS], b_stage[BR];
    f32x4 g_stage[GATE ? AR : 1];  // the gate is multiplied in when the tile is written to LDS: doing it at load time
                                   // would wait for the loads right there and expose their latency every K-tile

    const int nk = p.ksplit > 1 ? min(p.KT / BK - kt0, p.kt_per_split) : p.KT / BK;  // K-tiles of this block
    const int ktot = p.KH * p.KW * p.Cin;            // true K (stem mode)
    (void)ktot;

    // Staged loads are UNCONDITIONAL: an element outside the image / beyond M / in the K padding is fetched from a safe
    // address instead and zeroed when the tile is written to LDS (a_mask). Predicated loads compile to an exec-mask
    // branch per load, and hipcc then waits for earlier loads inside the sequence (vmcnt(0) between the stem's loads,
    // and at the gate multiply), which exposes the load latency every K-tile.
    unsigned a_mask = 0;
    auto load_tile = [&]() {
        unsigned mask = 0;
        if (MODE == 0) {
            const bool ci_ok = ld_ci + c4 * 4 < p.Cin;
            const long koff = (long)ld_tap + ld_ci;
#pragma unroll
            for (int i = 0; i < AR; ++i) {
                bool ok;
                if (PW) ok = ci_ok && a_hi0[i] >= 0;
                else ok = ci_ok && (unsigned)(a_hi0[i] + ld_kh) < (unsigned)p.H && (unsigned)(a_wi0[i] + ld_kw) < (unsigned)p.W;
                if (UL && PW) {  // launch guarantees Cin % BK == 0 here: every element of the tile exists
                    a_stage[i] = *reinterpret_cast<const f32x4*>(a_ptr[i] + koff);
                    if (GATE) g_stage[i] = *reinterpret_cast<const f32x4*>(g_ptr[i] + ld_ci);
                } else if (UL) {
                    a_stage[i] = *reinterpret_cast<const f32x4*>(ok ? a_ptr[i] + koff : p.x);
                    if (GATE) g_stage[i] = *reinterpret_cast<const f32x4*>(ok ? g_ptr[i] + ld_ci : p.gate);
                } else {
                    f32x4 v = {0.f, 0.f, 0.f, 0.f}, g = {0.f, 0.f, 0.f, 0.f};
                    if (ok) v = *reinterpret_cast<const f32x4*>(a_ptr[i] + koff);
                    if (GATE && ok) g = *reinterpret_cast<const f32x4*>(g_ptr[i] + ld_ci);
                    a_stage[i] = v;
                    if (GATE) g_stage[i] = g;
                }
                mask |= (ok ? 1u : 0u) << i;
            }
        } else {
            // tid / BM is wave-uniform (BM is a multiple of 64): the k decode stays on the scalar unit
            const int kb = ld_k + __builtin_amdgcn_readfirstlane(tid / BM) * KPT;
            const int plane = p.H * p.W;
            if (stem_fast) {  // wave-uniform
#pragma unroll
                for (int j = 0; j < KPT; j += 4) {
                    const int4 off = *reinterpret_cast<const int4*>(ktab + kb + j);  // same address in every lane
                    a_stage[j >> 2][0] = stem_row[off.x], a_stage[j >> 2][1] = stem_row[off.y];
                    a_stage[j >> 2][2] = stem_row[off.z], a_stage[j >> 2][3] = stem_row[off.w];
                }
                mask = ~0u;
            } else
#pragma unroll
            for (int j = 0; j < KPT; ++j) {
                const int k = kb + j;
                const int tap = (int)fdiv((unsigned)k, p.fd_cin), ci = k - tap * p.Cin;
                const int kh = (int)fdiv((unsigned)tap, p.fd_kw), kw = tap - kh * p.KW;
                const int hi = a_hi0[0] + kh, wi = a_wi0[0] + kw;
                const bool ok = k < ktot && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
                if (UL) {
                    a_stage[j >> 2][j & 3] = a_ptr[0][ok ? ci * plane + hi * p.W + wi : 0];
                } else {
                    float v = 0.f;
                    if (ok) v = a_ptr[0][ci * plane + hi * p.W + wi];
                    a_stage[j >> 2][j & 3] = v;
                }
                mask |= (ok ? 1u : 0u) << j;
            }
        }
        a_mask = mask;
#pragma unroll
        for (int j = 0; j < BR; ++j)
            b_stage[j] = *reinterpret_cast<const f32x4*>(b_ptr + ((size_t)(RPP * j) * p.KT + ld_k));
        // advance to the next K-tile
        ld_k += BK;
        if (MODE == 0 && !PW) {
            ld_ci += BK;
            if (ld_ci >= p.cin_pad) {
                ld_ci = 0, ld_tap += p.Cin;
                if (++ld_kw == p.KW) ld_kw = 0, ++ld_kh, ld_tap += (p.W - p.KW) * p.Cin;
            }
        } else if (MODE == 0) {
            ld_ci += BK;
        }
    };

    auto store_tile = [&](int buf) {
        float* A = As + buf * BM * LDS_STRIDE;
        float* Bq = Bs + buf * BN * LDS_STRIDE;
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < AR; ++i)
                if (BM % RPP == 0 || lrow + RPP * i < BM) {
                    const f32x4 v = GATE ? a_stage[i] * g_stage[i] : a_stage[i];
                    *reinterpret_cast<f32x4*>(A + (lrow + RPP * i) * LDS_STRIDE + c4 * 4) =
                        (!UL || PW || ((a_mask >> i) & 1u)) ? v : zero;
                }
        } else {
            float* dst = A + (tid % BM) * LDS_STRIDE + (tid / BM) * KPT;
#pragma unroll
            for (int j = 0; j < KPT / 4; ++j) {
                f32x4 v = a_stage[j];
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = (!UL || ((a_mask >> (4 * j + q)) & 1u)) ? v[q] : 0.f;
                *reinterpret_cast<f32x4*>(dst + 4 * j) = v;
            }
        }
#pragma unroll
        for (int j = 0; j < BR; ++j)
            if (BN % RPP == 0 || lrow + RPP * j < BN)
                *reinterpret_cast<f32x4*>(Bq + (lrow + RPP * j) * LDS_STRIDE + c4 * 4) = b_stage[j];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto compute_tile = [&](int cur) {
        const float* A = As + cur * BM * LDS_STRIDE + (wm + l31) * LDS_STRIDE + lh * 4 + wk * 8;
        const float* Bq = Bs + cur * BN * LDS_STRIDE + (wn + l31) * LDS_STRIDE + lh * 4 + wk * 8;
        // fragment reads are register double-buffered: group g+1 is fetched from LDS while the MFMAs of group g
        // issue, so only the first read of a K-tile is exposed
        f32x4 af[2][TM], bf[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[0][i] = *reinterpret_cast<const f32x4*>(A + i * 32 * LDS_STRIDE);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[0][j] = *reinterpret_cast<const f32x4*>(Bq + j * 32 * LDS_STRIDE);
#pragma unroll
        for (int g = 0; g < NGW; ++g) {
            if (g + 1 < NGW) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    af[(g + 1) & 1][i] = *reinterpret_cast<const f32x4*>(A + i * 32 * LDS_STRIDE + (g + 1) * 8 * WGK);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    bf[(g + 1) & 1][j] = *reinterpret_cast<const f32x4*>(Bq + j * 32 * LDS_STRIDE + (g + 1) * 8 * WGK);
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[g & 1][i][kk], bf[g & 1][j][kk],
                                                                         acc[i][j], 0, 0, 0);
        }
        // pin the issue order the code above spells out: reads(g0) | reads(g+1), MFMAs(g) | ... (0x100 = DS read,
        // 0x008 = MFMA); without it the scheduler batches the reads behind all issued MFMAs and exposes LDS latency
        __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
#pragma unroll
        for (int g = 0; g < NGW; ++g) {
            if (g + 1 < NGW) __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4 * TM * TN, 0);
        }
        // (s_setprio(1) around this cluster was measured: +-2 % on most layers, -10..-17 % on three of the gated
        // projections - the co-resident blocks are not role-split enough for the arbiter to help)
    };

    // folded-BatchNorm scale / shift of this wave's output columns: requested before the K loop so the
    // epilogue does not open with an L2 round trip - on the 1-3 K-tile EfficientNet layers that is a visible share
    // (split-K partial: plain sums, the reduce kernel applies the epilogue; dual write: the tile is staged raw, scale / shift
    // enter in the output pass; the shift enters the sum once)
    float sc_pre[TN], sh_pre[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn + j * 32 + l31;
        const bool use = n < p.Cout && p.ksplit <= 1 && p.y_raw == nullptr;
        sc_pre[j] = (use && p.scale) ? p.scale[n] : 1.0f;
        sh_pre[j] = (use && p.shift && wk == 0) ? p.shift[n] : 0.0f;
    }

    load_tile();
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) load_tile();  // global loads in flight under the MFMAs below
        compute_tile(cur);
        if (kt + 1 < nk) store_tile(cur ^ 1);  // the other buffer was last read before the previous barrier
        __syncthreads();
    }

    // ---- epilogue. C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
    // The tile goes through LDS (the pipeline buffers are free after the last barrier) so that HBM sees whole
    // rows: every thread then moves 16 bytes, a wave instruction covers 256-1024 contiguous bytes per output row
    // (the direct form issues 4-byte stores, 128 B per row fragment, and is store-issue-bound on the wide, short-K
    // EfficientNet layers).
    // LDS row stride (floats) of the staged C tile: NO padding. The 16-lane groups of the ds_read_b128 below
    // ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, +32; MI355X_MICROARCH.md, LDS) fall on 16 distinct 16-byte slots of a
    // linear tile for BN = 32 / 64 / 96 / 128; the BN + 4 padding of rounds 1-2 put rows 0 / 1 and 2 / 3 of a group on
    // shared slots (most of the 0.1-0.16 conflicts per LDS cycle that were left on the one-K-tile layers)
    constexpr int CS = BN;
    constexpr int OROWS = POOL2 ? BM / 4 : BM;    // output rows of this tile
    float* Cs = smem + wk * (OROWS * CS);        // [WGK][OROWS][CS]: one partial tile per K-wave group
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const float sc = sc_pre[j], sh = sh_pre[j];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int rowl = wm + i * 32 + 8 * rq + 4 * lh;  // first of 4 consecutive tile rows
                if (POOL2) {  // the 4 rows are one pooling window (window-major row order)
                    float v = -INFINITY;
#pragma unroll
                    for (int r = 0; r < 4; ++r) v = fmaxf(v, apply_act(acc[i][j][rq * 4 + r] * sc + sh, p.act));
                    Cs[(rowl >> 2) * CS + wn + j * 32 + l31] = v;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        Cs[(rowl + r) * CS + wn + j * 32 + l31] = acc[i][j][rq * 4 + r] * sc + sh;
                }
            }
        }
    }
    __syncthreads();
    {
        constexpr int TPO = BN / 4;      // threads per output row
        constexpr int RPO = 256 / TPO;   // rows per pass
        const bool epi_thread = tid < RPO * TPO;  // BN = 96: 10 rows x 24 threads, the last 16 threads sit the epilogue out
        const int oc = (tid % TPO) * 4;
        const int n = epi_thread ? n0 + oc : p.Cout;
        const int mo0 = POOL2 ? (m0 >> 2) : m0;
        const int mout = POOL2 ? (p.M >> 2) : p.M;
        float* yout = p.ksplit > 1 ? p.part + (size_t)split * p.M * p.Cout : p.y;
        constexpr int ITER = (OROWS + RPO - 1) / RPO;  // output rows per thread (4 for the 64x64 and 128x32 tiles)
        f32x4 st_s = {0.f, 0.f, 0.f, 0.f}, st_q = {0.f, 0.f, 0.f, 0.f};  // column sums of this thread's rows (p.stats)
        if (n < p.Cout && ITER <= 8) {
            // batched form: the LDS reads and the residual loads of all of a thread's rows are issued before the first use
            // (row by row, every row paid an LDS - and with a skip connection an L2 - round trip of its own)
            f32x4 v[ITER], res[ITER];
            const bool has_res = !POOL2 && p.ksplit <= 1 && p.residual != nullptr;
            const bool dual = p.y_raw != nullptr;  // (launcher: never with pooling or split-K)
            f32x4 dsc = {1.f, 1.f, 1.f, 1.f}, dsh = {0.f, 0.f, 0.f, 0.f};
            if (dual && p.scale) dsc = *reinterpret_cast<const f32x4*>(p.scale + n), dsh = *reinterpret_cast<const f32x4*>(p.shift + n);
#pragma unroll
            for (int it = 0; it < ITER; ++it) {
                const int r = tid / TPO + it * RPO;
                const bool ok = r < OROWS && mo0 + r < mout;
                const int rr = ok ? r : 0;
                v[it] = *reinterpret_cast<const f32x4*>(smem + rr * CS + oc);
#pragma unroll
                for (int q = 1; q < WGK; ++q) v[it] += *reinterpret_cast<const f32x4*>(smem + (q * OROWS + rr) * CS + oc);
                res[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (has_res) res[it] = *reinterpret_cast<const f32x4*>(p.residual + (size_t)(ok ? mo0 + r : 0) * p.Cout + n);
            }
#pragma unroll
            for (int it = 0; it < ITER; ++it) {
                const int r = tid / TPO + it * RPO;
                if (r < OROWS && mo0 + r < mout) {
                    f32x4 o = v[it];
                    st_s += o, st_q += o * o;
                    if (dual) {
                        *reinterpret_cast<f32x4*>(p.y_raw + (size_t)(mo0 + r) * p.Cout + n) = o;
                        o = o * dsc + dsh;
                    }
                    if (!POOL2 && p.ksplit <= 1) {
                        o += res[it];
                        o[0] = apply_act(o[0], p.act), o[1] = apply_act(o[1], p.act);
                        o[2] = apply_act(o[2], p.act), o[3] = apply_act(o[3], p.act);
                    }
                    if (yout) *reinterpret_cast<f32x4*>(yout + (size_t)(mo0 + r) * p.Cout + n) = o;  // (null: statistics sweep)
                }
            }
        } else if (n < p.Cout) {
#pragma unroll 4
            for (int r = tid / TPO; r < OROWS; r += RPO) {
                const int m = mo0 + r;
                if (m >= mout) break;
                f32x4 v = *reinterpret_cast<const f32x4*>(smem + r * CS + oc);
#pragma unroll
                for (int q = 1; q < WGK; ++q) v += *reinterpret_cast<const f32x4*>(smem + (q * OROWS + r) * CS + oc);
                st_s += v, st_q += v * v;
                if (p.y_raw != nullptr) {
                    *reinterpret_cast<f32x4*>(p.y_raw + (size_t)m * p.Cout + n) = v;
                    if (p.scale) v = v * *reinterpret_cast<const f32x4*>(p.scale + n) + *reinterpret_cast<const f32x4*>(p.shift + n);
                }
                if (!POOL2 && p.ksplit <= 1) {
                    if (p.residual) v += *reinterpret_cast<const f32x4*>(p.residual + (size_t)m * p.Cout + n);
                    v[0] = apply_act(v[0], p.act), v[1] = apply_act(v[1], p.act);
                    v[2] = apply_act(v[2], p.act), v[3] = apply_act(v[3], p.act);
                }
                if (yout) *reinterpret_cast<f32x4*>(yout + (size_t)m * p.Cout + n) = v;
            }
        }
        // train-mode BatchNorm statistics of this tile's rows (rows beyond M never entered the sums): the RPO row lanes of a
        // column quad are combined through LDS in a fixed order, one float4 pair per (row block, column quad) goes out
        if (p.stats != nullptr) {  // uniform
            __syncthreads();       // every thread is done reading the C tile
            f32x4* red = reinterpret_cast<f32x4*>(smem);  // [2][256]
            red[tid] = st_s, red[256 + tid] = st_q;
            __syncthreads();
            if (tid < TPO && n < p.Cout) {
                f32x4 a = red[tid], b = red[256 + tid];
#pragma unroll 4
                for (int r = 1; r < RPO; ++r) a += red[r * TPO + tid], b += red[256 + r * TPO + tid];
                const size_t mt = (size_t)(tile / p.n_tiles);
                *reinterpret_cast<f32x4*>(p.stats + (mt * 2 + 0) * p.Cout + n) = a;
                *reinterpret_cast<f32x4*>(p.stats + (mt * 2 + 1) * p.Cout + n) = b;
            }
        }
    }
}

// ---- split-K: sum of the partial tiles in split order (deterministic) + the conv epilogue -------------------------------
__global__ __launch_bounds__(256) void conv_splitk_reduce_kernel(const float* __restrict__ part, int S, size_t mn4,
                                                                 int cout4, const float* __restrict__ scale,
                                                                 const float* __restrict__ shift,
                                                                 const float* __restrict__ residual, int act,
                                                                 float* __restrict__ y) {
    const f32x4* p4 = reinterpret_cast<const f32x4*>(part);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < mn4; i += (size_t)gridDim.x * 256) {
        f32x4 v = p4[i];
        for (int s2 = 1; s2 < S; ++s2) v += p4[(size_t)s2 * mn4 + i];
        const int n4 = (int)(i % (size_t)cout4);
        if (scale) v *= reinterpret_cast<const f32x4*>(scale)[n4];
        if (shift) v += reinterpret_cast<const f32x4*>(shift)[n4];
        if (residual) v += reinterpret_cast<const f32x4*>(residual)[i];
        v[0] = apply_act(v[0], act), v[1] = apply_act(v[1], act), v[2] = apply_act(v[2], act), v[3] = apply_act(v[3], act);
        reinterpret_cast<f32x4*>(y)[i] = v;
    }
}

// ---- weight packing: OIHW -> [cout_pad][KT] with k = (kh*KW + kw)*cin_pad + ci (vector mode)
//                                              or k = (kh*KW + kw)*Cin + ci     (stem mode) ----------
__global__ __launch_bounds__(256) void conv_pack_kernel(const float* __restrict__ w, float* __restrict__ wp,
                                                        int Cin, int Cout, int KH, int KW, int cin_pad,
                                                        int KT, int cout_pad, int stem) {
    const size_t total = (size_t)cout_pad * KT;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int n = (int)(i / KT), k = (int)(i % KT);
        const int per = stem ? Cin : cin_pad;
        const int tap = k / per, ci = k % per;
        float v = 0.f;
        if (n < Cout && tap < KH * KW && ci < Cin) {
            const int kh = tap / KW, kw = tap % KW;
            v = w[(((size_t)n * Cin + ci) * KH + kh) * KW + kw];
        }
        wp[i] = v;
    }
}

// K-tile width for a layer: the widest of 32/16/8 that divides Cin (stems: 32)
static int choose_bk(int Cin, int x_nchw) {
    if (x_nchw) return 32;
    const int cap = get_option("conv_bk");  // tuning sweeps: 0 = widest that divides Cin (default), else 8 / 16 / 32
    if (Cin % 32 == 0 && (cap == 0 || cap >= 32)) return 32;
    if (Cin % 16 == 0 && (cap == 0 || cap >= 16)) return 16;
    return 8;
}

ConvPackGeom conv_pack_geom(int Cin, int Cout, int KH, int KW, int x_nchw) {
    ConvPackGeom g;
    const int bk = choose_bk(Cin, x_nchw);
    if (x_nchw) {
        g.cin_pad = Cin;
        g.kt = (int)align_up((size_t)KH * KW * Cin, bk);
    } else {
        g.cin_pad = (int)align_up((size_t)Cin, bk);
        g.kt = KH * KW * g.cin_pad;
    }
    g.cout_pad = (int)align_up((size_t)Cout, 128);
    return g;
}

size_t conv_packed_floats(int Cin, int Cout, int KH, int KW, int x_nchw) {
    const ConvPackGeom g = conv_pack_geom(Cin, Cout, KH, KW, x_nchw);
    return (size_t)g.cout_pad * g.kt;
}

int conv_pack_weights(const float* w_oihw, float* w_packed, int Cin, int Cout, int KH, int KW, int x_nchw,
                      hipStream_t s) {
    const ConvPackGeom g = conv_pack_geom(Cin, Cout, KH, KW, x_nchw);
    const size_t total = (size_t)g.cout_pad * g.kt;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    conv_pack_kernel<<<blocks, 256, 0, s>>>(w_oihw, w_packed, Cin, Cout, KH, KW, g.cin_pad, g.kt,
                                            g.cout_pad, x_nchw);
    ORBIT_LAUNCH_CHECK();
    return ORBIT_OK;
}

// ---- optional per-launch profiling (bench.py roofline): HIP events recorded on the launch stream ------
struct ProfRec {
    hipEvent_t start, stop;
    int variant;
    double flops, bytes, silu;
};
struct ProfVariant {
    char name[48];
    long launches;
    double ms, flops, bytes;
    double floor_ms;  // sum over the launches of max(bytes / HBM rate, FLOP / matrix rate): the launch-by-launch roofline floor
    double silu;      // SiLU evaluations (two transcendentals each: the VALU work the matrix roof does not see)
    double floor_simd_ms;  // ... of max(bytes / HBM rate, FLOP / matrix rate + SiLU / SiLU rate): on gfx950 a SIMD issues EITHER
                           // an MFMA OR VALU instructions (profiles/r03_coexec_probe.txt), so matrix and SiLU time add up
};
// orbit_prof_set_roofs. SiLU: 11.06 ns of one SIMD per 64 evaluations at 8 waves per SIMD (v_exp_f32 + v_rcp_f32 + 3 packed
// multiply-adds, profiles/r03_valu_probe.txt) x 1024 SIMDs
static double g_roof_bytes_per_s = 6.3e12, g_roof_flop_per_s = 157.3e12, g_roof_silu_per_s = 64.0 * 1024.0 / 11.06e-9;
static bool g_prof_on = false;
bool conv_prof_enabled() { return g_prof_on; }
static std::vector<ProfRec> g_prof_recs;
static std::vector<hipEvent_t> g_prof_pool;
static std::vector<ProfVariant> g_prof_variants;

static hipEvent_t prof_event() {
    if (!g_prof_pool.empty()) {
        hipEvent_t e = g_prof_pool.back();
        g_prof_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
static int prof_variant(const char* name) {
    for (size_t i = 0; i < g_prof_variants.size(); ++i)
        if (strcmp(g_prof_variants[i].name, name) == 0) return (int)i;
    ProfVariant v;
    memset(&v, 0, sizeof(v));
    snprintf(v.name, sizeof(v.name), "%s", name);
    g_prof_variants.push_back(v);
    return (int)g_prof_variants.size() - 1;
}

// shared with the other MFMA kernels (conv_wgrad.hip): returns a record index or -1 when profiling is off
int prof_start(const char* name, double flops, double bytes, hipStream_t s, double silu) {
    if (!g_prof_on) return -1;
    ProfRec r;
    r.start = prof_event(), r.stop = prof_event(), r.variant = prof_variant(name);
    r.flops = flops, r.bytes = bytes, r.silu = silu;
    (void)hipEventRecord(r.start, s);
    g_prof_recs.push_back(r);
    return (int)g_prof_recs.size() - 1;
}
void prof_stop(int idx, hipStream_t s) {
    if (idx >= 0 && idx < (int)g_prof_recs.size()) (void)hipEventRecord(g_prof_recs[idx].stop, s);
}

template <int BM, int BN, int WGM, int WGN, int WGK, int BK, int MODE, bool POOL2, bool GATE, bool PW, bool UL>
static int launch_cfg2(ConvParams& p, hipStream_t s) {
    p.m_tiles = cdiv(p.M, BM);
    p.n_tiles = cdiv(p.Cout, BN);
    const size_t lds_pipe = (size_t)2 * (BM + BN) * (BK + 4) * sizeof(float) + (MODE == 1 ? 1024 : 0);  // + stem k table
    const size_t lds_epi = (size_t)WGK * (POOL2 ? BM / 4 : BM) * BN * sizeof(float);
    const size_t lds = lds_pipe > lds_epi ? lds_pipe : lds_epi;
    auto kern = conv_igemm_kernel<BM, BN, WGM, WGN, WGK, BK, MODE, POOL2, GATE, PW, UL>;
    static bool attr_set = false;  // >64 KiB dynamic LDS needs the opt-in once per kernel
    if (!attr_set && lds > 64 * 1024) {
        ORBIT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    const int grid = p.m_tiles * p.n_tiles * (p.ksplit > 1 ? p.ksplit : 1);
    if (g_prof_on) {
        char name[48];
        snprintf(name, sizeof(name), "conv_igemm<%d,%d,%d,%s%s%s%s%s>", BM, BN, BK, MODE ? "nchw" : "nhwc",
                 POOL2 ? ",pool2" : "", GATE ? ",gate" : "", PW ? ",pw" : "",
                 p.ksplit > 1 ? ",splitk" : WGK == 2 ? ",k2" : WGK == 4 ? ",k4" : (!p.y ? ",stats" : ""));
        ProfRec r;
        r.start = prof_event(), r.stop = prof_event(), r.variant = prof_variant(name);
        const double pix = POOL2 ? (double)p.B * p.HoP * p.WoP * 4 : (double)p.B * p.Ho * p.Wo;
        r.flops = 2.0 * pix * p.Cout * p.KH * p.KW * p.Cin * p.prof_flop_scale;  // algorithmic (unpadded) FLOPs
        // algorithmic HBM bytes: input once, output once (pooled if fused), residual once, weights once
        r.bytes = 4.0 * ((double)p.B * p.H * p.W * p.Cin +
                         (POOL2 ? pix / 4 : pix) * p.Cout * ((p.residual ? 1.0 : 0.0) + (p.y || p.ksplit > 1 ? 1.0 : 0.0)) +
                         (double)p.Cout * p.KH * p.KW * p.Cin);
        r.silu = p.act == ORBIT_ACT_SILU && p.ksplit <= 1 ? pix * p.Cout : 0.0;
        (void)hipEventRecord(r.start, s);
        kern<<<grid, 256, lds, s>>>(p);
        (void)hipEventRecord(r.stop, s);
        g_prof_recs.push_back(r);
    } else {
        kern<<<grid, 256, lds, s>>>(p);
    }
    ORBIT_LAUNCH_CHECK();
    return ORBIT_OK;
}

template <int BM, int BN, int WGM, int WGN, int WGK, int BK, int MODE, bool POOL2, bool GATE, bool PW>
static int launch_cfg(ConvParams& p, hipStream_t s) {
    // UL = staged loads without predicates, for pointwise convs (every element of a tile exists once the rows beyond M are
    // clamped): no predicate, mask or select at all. In-process A/B on MI355X: +1..2 % on most pointwise layers, +16..20 % on
    // the 128x32 projections, gated projections +3..21 % (the gate and tile loads then share one wait; predicated, hipcc
    // waits for the tile load before it issues the gate load); 3x3 convs and the stems lose 1-4 % to the selects.
    if constexpr (PW)
        if (p.cin_pad == p.Cin)  // the clamped pointwise form has no K-padding predicate
            return launch_cfg2<BM, BN, WGM, WGN, WGK, BK, MODE, POOL2, GATE, PW, true>(p, s);
    return launch_cfg2<BM, BN, WGM, WGN, WGK, BK, MODE, POOL2, GATE, PW, false>(p, s);
}

template <int BK, int MODE, bool POOL2, bool GATE, bool PW>
static int launch_tiled(ConvParams& p, hipStream_t s) {
    if (p.ksplit > 1) return launch_cfg<64, 64, 2, 2, 1, BK, MODE, POOL2, GATE, PW>(p, s);  // split-K plans on 64x64 tiles
    switch (get_option("conv_tile")) {  // 0 = the heuristic below; else force one of its three tilings (parity tests, A/B)
        case 3: return launch_cfg<64, 64, 2, 2, 1, BK, MODE, POOL2, GATE, PW>(p, s);
        case 4: return launch_cfg<128, 32, 4, 1, 1, BK, MODE, POOL2, GATE, PW>(p, s);
        case 6: if constexpr (MODE == 0 && !POOL2 && BK == 32) return launch_cfg<32, 32, 1, 1, 4, BK, MODE, POOL2, GATE, PW>(p, s); break;
        default: break;
    }
    // Measured sweep on MI355X (tools/conv_bench.py <net> sweep, in-process A/B over every layer shape of resnet18 @84
    // and @224, efficientnet_b0 @224 and the set encoder): the 64x64 tile (36 KB LDS -> 4 blocks = 4 waves per SIMD) is
    // the fastest on EVERY layer, including the large MFMA-bound ones (+10..24 % over 128x128 / 128x64, whose 2 blocks
    // per CU cover the per-K-tile bubble worse). 128x32 wins only where the last 64-wide column tile would be mostly
    // padding (Cout <= 32, or e.g. Cout = 80, 96, 144).
    // (A second K-tile of staged loads in flight was also measured: 30 more VGPRs, -6..-17 % on the short-K layers,
    // +-2 % on the long-K ones - removed.)
    const double waste64 = (double)(cdiv(p.Cout, 64) * 64 - p.Cout) / p.Cout;
    const double waste32 = (double)(cdiv(p.Cout, 32) * 32 - p.Cout) / p.Cout;
    if (p.Cout <= 32 || waste64 - waste32 >= 0.15) return launch_cfg<128, 32, 4, 1, 1, BK, MODE, POOL2, GATE, PW>(p, s);
    // fewer 64x64 tiles than CUs (resnet18 @84 layer4: 29 x 8 tiles for 256 CUs): 32x32 tiles with the four waves
    // splitting K give 4x the blocks (-7..-11 % time on those layers; neutral or worse everywhere else)
    if constexpr (MODE == 0 && !POOL2 && BK == 32)
        if (cdiv(p.M, 64) * cdiv(p.Cout, 64) < 256) return launch_cfg<32, 32, 1, 1, 4, BK, MODE, POOL2, GATE, PW>(p, s);
    return launch_cfg<64, 64, 2, 2, 1, BK, MODE, POOL2, GATE, PW>(p, s);
}

template <int MODE, bool POOL2, bool GATE, bool PW>
static int launch_bk(ConvParams& p, int bk, hipStream_t s) {
    if constexpr (MODE == 1) {
        return launch_tiled<32, 1, POOL2, false, false>(p, s);
    } else {
        if (bk == 32) return launch_tiled<32, 0, POOL2, GATE, PW>(p, s);
        if (bk == 16) return launch_tiled<16, 0, POOL2, GATE, PW>(p, s);
        return launch_tiled<8, 0, POOL2, GATE, PW>(p, s);
    }
}

// Split-K plan. A conv whose 64x64 tiling leaves the chip short of blocks but whose reduction is long (resnet18 @84
// layer3/4: 1 800-7 200 output rows, K = 1 152-4 608; EfficientNet's 7x7 projections) is cut into S K-ranges: S x the
// blocks, each writing a raw partial tile; conv_splitk_reduce_kernel sums them in split order and applies the epilogue.
int conv_splitk(const ConvDesc& d) {
    if (get_option("conv_splitk") == 0 || d.x_nchw || d.pool2 || d.Cout <= 32 || d.Cout % 4 != 0) return 1;
    const ConvPackGeom g = conv_pack_geom(d.Cin, d.Cout, d.KH, d.KW, 0);
    const int nk = g.kt / choose_bk(d.Cin, 0);
    const long tiles = (long)cdiv(d.B * d.Ho * d.Wo, 64) * cdiv(d.Cout, 64);
    // measured with the reduce pass included (tools/conv_bench.py <net> ab conv_splitk): -11..-18 % time at 232 tiles
    // (resnet18 @84 layer4), break-even at ~450 tiles (layer3, EfficientNet's 7x7 projections)
    if (tiles >= 320 || nk < 16) return 1;
    int S = (int)(1280 / tiles);
    S = S < 2 ? 2 : (S > 4 ? 4 : S);
    while (S > 1 && cdiv(nk, S) < 6) --S;
    return S;
}
size_t conv_splitk_floats(const ConvDesc& d) {
    const int S = conv_splitk(d);
    return S > 1 ? (size_t)S * d.B * d.Ho * d.Wo * d.Cout : 0;
}

int launch_conv(const ConvDesc& d, hipStream_t s) {
    ORBIT_REQUIRE(d.x && d.w_packed && (d.y || d.stats_only), "conv: null pointer");
    ORBIT_REQUIRE(!d.stats_only || (d.y == nullptr && d.stats && d.stats_blocks && !d.pool2 && !d.splitk_ws && !d.y_raw),
                  "conv: the statistics sweep stores nothing and needs the whole-tile statistics epilogue");
    ORBIT_REQUIRE(d.B > 0 && d.H > 0 && d.W > 0 && d.Cin > 0 && d.Cout > 0 && d.Ho > 0 && d.Wo > 0,
                  "conv: bad sizes");
    ORBIT_REQUIRE(d.x_nchw ? d.Cin <= 4 : (d.Cin % 4 == 0),
                  "conv: NHWC path needs Cin %% 4 == 0, NCHW stem path needs Cin <= 4 (Cin=%d)", d.Cin);
    ORBIT_REQUIRE(!(d.pool2 && d.residual), "conv: pool2 cannot be combined with a residual input");
    ORBIT_REQUIRE(d.Cout % 4 == 0, "conv: Cout %% 4 != 0 (Cout=%d): the epilogue writes float4 rows", d.Cout);
    const bool pw = !d.x_nchw && d.KH == 1 && d.KW == 1 && d.pad_t == 0 && d.pad_l == 0;
    if (d.stats_blocks) *d.stats_blocks = 0;
    // opt-in: three-way bf16 split of both operands on the bf16 matrix cores (a function of the layer only, like every routing
    // rule here: a frame's bits do not depend on its batch)
    const int rg = d.stats_only ? 0 : get_option("conv_rgemm");
    if (!d.stats_only && (get_option("conv_bf3") & 1) && conv_bf3_supported(d) &&
        !(rg == 1 && pw_rgemm_supported(d) && pw_rgemm_preferred(d)))  // (1152 -> 320 @7x7: 770 tiles on 768 slots - the register GEMM's 1 535 blocks win)
        return launch_conv_bf3(d, s);
    if (rg == 2 && pw_rgemm_supported(d)) return launch_pw_rgemm(d, s);
    if (rg && pw_rgemm_supported(d) && (rg == 2 || pw_rgemm_preferred(d))) return launch_pw_rgemm(d, s);
    ORBIT_REQUIRE(!d.gate || (!d.x_nchw && !d.pool2), "conv: the squeeze-excite gate needs the NHWC path without fused pooling");
    const ConvPackGeom g = conv_pack_geom(d.Cin, d.Cout, d.KH, d.KW, d.x_nchw);
    ConvParams p;
    p.x = d.x, p.w = d.w_packed, p.y = d.y, p.scale = d.scale, p.shift = d.shift;
    p.residual = d.residual, p.gate = d.gate;
    p.B = d.B, p.H = d.H, p.W = d.W, p.Cin = d.Cin, p.Cout = d.Cout, p.KH = d.KH, p.KW = d.KW;
    p.stride = d.stride, p.pad_t = d.pad_t, p.pad_l = d.pad_l, p.Ho = d.Ho, p.Wo = d.Wo;
    p.HoP = d.Ho / 2, p.WoP = d.Wo / 2;
    p.KT = g.kt, p.cin_pad = g.cin_pad, p.act = d.act;
    p.prof_flop_scale = d.prof_flop_scale;
    if (d.pool2) {
        ORBIT_REQUIRE(p.HoP > 0 && p.WoP > 0, "conv: pool2 needs Ho, Wo >= 2");
        p.M = d.B * p.HoP * p.WoP * 4;
        p.fd_per = make_fastdiv((unsigned)(p.HoP * p.WoP)), p.fd_wo = make_fastdiv((unsigned)p.WoP);
    } else {
        p.M = d.B * d.Ho * d.Wo;
        p.fd_per = make_fastdiv((unsigned)(d.Ho * d.Wo)), p.fd_wo = make_fastdiv((unsigned)d.Wo);
    }
    p.fd_cin = make_fastdiv((unsigned)d.Cin), p.fd_kw = make_fastdiv((unsigned)d.KW);
    ORBIT_REQUIRE((long long)d.B * d.H * d.W * d.Cin < (1ll << 40) && p.M > 0, "conv: tensor too large");
    int bk = choose_bk(d.Cin, d.x_nchw);
    // K-tile of 16 where 32 would also divide Cin (the packed filter is the same: cin_pad and the k-order do not depend on
    // BK when Cin % 32 == 0). Half the LDS stage = 7 instead of 4 blocks per CU; it pays (tools/conv_bench.py effnet_224 with
    // ORBIT_CONV_BK=16, 200 frames) on the narrow HBM-bound projections - 32 -> 16 @112x112: 131 -> 115 us, 96 -> 24 @56x56:
    // 79.5 -> 75.9 - and where the 64x64 tiling gives between one and two rounds of 4 blocks per CU, i.e. a second round that
    // is mostly empty (480 -> 112 and 672 -> 112 @14x14, 1 226 tiles: 65.7 -> 60.2, 81.0 -> 73.7 us); elsewhere 32 is as good
    // or better (fewer barriers per K: 1152 -> 320 @7x7 91.8 vs 96.1 us)
    if (bk == 32 && pw && !d.pool2 && !d.x_nchw && get_option("conv_bk") == 0) {
        const long tiles64 = (long)cdiv(p.M, 64) * cdiv(d.Cout, 64);
        if ((d.Cout <= 32 && d.Cin <= 128) || (d.Cout > 32 && tiles64 > 1024 && tiles64 <= 1792)) bk = 16;
    }
    p.ksplit = d.splitk_ws ? conv_splitk(d) : 1;
    p.kt_per_split = p.ksplit > 1 ? cdiv(g.kt / bk, p.ksplit) : 0;
    p.part = d.splitk_ws;
    // statistics of the raw outputs for a train-mode BatchNorm: whole-tile launches only (a split-K partial is not the output,
    // the fused max-pool stores pooled rows)
    p.stats = (d.stats && d.stats_blocks && p.ksplit <= 1 && !d.pool2) ? d.stats : nullptr;
    ORBIT_REQUIRE(!d.y_raw || (!d.pool2 && p.ksplit <= 1 && (d.scale == nullptr) == (d.shift == nullptr)),
                  "conv: the dual (raw + activated) output needs the whole-tile epilogue without fused pooling");
    p.y_raw = d.y_raw;
    int rc;
    if (d.x_nchw) rc = d.pool2 ? launch_bk<1, true, false, false>(p, bk, s) : launch_bk<1, false, false, false>(p, bk, s);
    else if (pw && !d.pool2) rc = d.gate ? launch_bk<0, false, true, true>(p, bk, s) : launch_bk<0, false, false, true>(p, bk, s);
    else if (d.gate) rc = launch_bk<0, false, true, false>(p, bk, s);
    else rc = d.pool2 ? launch_bk<0, true, false, false>(p, bk, s) : launch_bk<0, false, false, false>(p, bk, s);
    if (rc == ORBIT_OK && p.stats) *d.stats_blocks = p.m_tiles;  // (launch_cfg2 set the tiling it chose)
    if (rc != ORBIT_OK || p.ksplit <= 1) return rc;
    const size_t mn4 = (size_t)p.M * p.Cout / 4;
    const int blocks = (int)std::min<size_t>((mn4 + 255) / 256, 4096);
    const int rec = prof_start("conv_splitk_reduce", 0.0, 4.0 * (p.ksplit + 1) * (double)p.M * p.Cout, s);
    conv_splitk_reduce_kernel<<<blocks, 256, 0, s>>>(p.part, p.ksplit, mn4, p.Cout / 4, p.scale, p.shift, p.residual, p.act,
                                                     p.y);
    prof_stop(rec, s);
    ORBIT_LAUNCH_CHECK();
    return ORBIT_OK;
}

}  // namespace orbit

using namespace orbit;

extern "C" {

// Profiling of the dominant kernel (all conv_igemm instantiations). enable(1) starts recording one HIP event
// pair per launch on the launch stream; collect() waits for them, folds them into per-variant totals and
// returns the grand totals; variant(i) reads one row. Not thread-safe: one profiled stream at a time.
int orbit_prof_enable(int on) {
    g_prof_on = on != 0;
    if (on) {
        for (ProfRec& r : g_prof_recs) g_prof_pool.push_back(r.start), g_prof_pool.push_back(r.stop);
        g_prof_recs.clear();
        g_prof_variants.clear();
    }
    return ORBIT_OK;
}

int orbit_prof_collect(double* total_ms, double* total_flops, long* launches) {
    double ms = 0, fl = 0;
    for (ProfRec& r : g_prof_recs) {
        ORBIT_HIP_CHECK(hipEventSynchronize(r.stop));
        float t = 0.f;
        ORBIT_HIP_CHECK(hipEventElapsedTime(&t, r.start, r.stop));
        ProfVariant& v = g_prof_variants[r.variant];
        v.launches += 1, v.ms += t, v.flops += r.flops, v.bytes += r.bytes, v.silu += r.silu;
        const double fb = r.bytes / g_roof_bytes_per_s, ff = r.flops / g_roof_flop_per_s;
        v.floor_ms += 1e3 * (fb > ff ? fb : ff);
        const double fs = ff + r.silu / g_roof_silu_per_s;
        v.floor_simd_ms += 1e3 * (fb > fs ? fb : fs);
        ms += t, fl += r.flops;
        g_prof_pool.push_back(r.start), g_prof_pool.push_back(r.stop);
    }
    if (total_ms) *total_ms = ms;
    if (total_flops) *total_flops = fl;
    if (launches) *launches = (long)g_prof_recs.size();
    g_prof_recs.clear();
    return ORBIT_OK;
}

int orbit_prof_num_variants(void) { return (int)g_prof_variants.size(); }

int orbit_prof_set_roofs(double hbm_bytes_per_s, double matrix_flop_per_s, double silu_evals_per_s) {
    ORBIT_REQUIRE(hbm_bytes_per_s > 0 && matrix_flop_per_s > 0 && silu_evals_per_s > 0, "prof_set_roofs: rates must be positive");
    g_roof_bytes_per_s = hbm_bytes_per_s, g_roof_flop_per_s = matrix_flop_per_s, g_roof_silu_per_s = silu_evals_per_s;
    return ORBIT_OK;
}

int orbit_prof_variant_floor(int i, double* floor_ms, double* floor_simd_ms, double* silu_evals) {
    ORBIT_REQUIRE(i >= 0 && i < (int)g_prof_variants.size(), "prof_variant_floor: index out of range");
    if (floor_ms) *floor_ms = g_prof_variants[i].floor_ms;
    if (floor_simd_ms) *floor_simd_ms = g_prof_variants[i].floor_simd_ms;
    if (silu_evals) *silu_evals = g_prof_variants[i].silu;
    return ORBIT_OK;
}

int orbit_prof_variant(int i, char* name48, long* launches, double* ms, double* flops, double* bytes) {
    ORBIT_REQUIRE(i >= 0 && i < (int)g_prof_variants.size(), "prof_variant: index out of range");
    const ProfVariant& v = g_prof_variants[i];
    if (name48) memcpy(name48, v.name, sizeof(v.name));
    if (launches) *launches = v.launches;
    if (ms) *ms = v.ms;
    if (flops) *flops = v.flops;
    if (bytes) *bytes = v.bytes;
    return ORBIT_OK;
}

/* Training form (single-operator entry for the parity tests): y = conv(x * gate) without epilogue, plus the per-channel sums
 * and sums of squares of y that the epilogue emits per row block, reduced to stats [2][Cout]; *stat_blocks receives the
 * number of row blocks the kernel wrote (0: this shape does not emit them - fused pooling, split-K, narrow-pointwise). */
int orbit_op_conv2d_train(const float* x, int x_nchw, const float* w, float* y, const float* gate, int B, int H, int W,
                          int Cin, int Cout, int KH, int KW, int stride, int pad_top, int pad_left, int Ho, int Wo,
                          float* stats, int* stat_blocks, orbit_stream_t stream) {
    ORBIT_REQUIRE(x && w && y && stats && stat_blocks, "op_conv2d_train: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const size_t nfl = conv_packed_floats(Cin, Cout, KH, KW, x_nchw);
    const size_t M = (size_t)B * Ho * Wo;
    const size_t pfl = bn_partial_floats((M + 31) / 32, Cout);
    float* tmp = nullptr;
    ORBIT_HIP_CHECK(hipMallocAsync(reinterpret_cast<void**>(&tmp), (nfl + pfl + 64) * sizeof(float), s));
    float* part = tmp + ((nfl + 63) & ~(size_t)63);
    int rc = conv_pack_weights(w, tmp, Cin, Cout, KH, KW, x_nchw, s);
    if (rc == ORBIT_OK) {
        ConvDesc d;
        d.x = x, d.w_packed = tmp, d.y = y, d.scale = d.shift = d.residual = nullptr, d.gate = gate;
        d.B = B, d.H = H, d.W = W, d.Cin = Cin, d.Cout = Cout, d.KH = KH, d.KW = KW, d.stride = stride;
        d.pad_t = pad_top, d.pad_l = pad_left, d.Ho = Ho, d.Wo = Wo, d.act = ORBIT_ACT_NONE, d.pool2 = 0, d.x_nchw = x_nchw;
        d.stats = part, d.stats_blocks = stat_blocks;
        rc = launch_conv(d, s);
        if (rc == ORBIT_OK && *stat_blocks > 0) rc = launch_sum_partials(part, *stat_blocks, Cout, stats, s);
    }
    (void)hipFreeAsync(tmp, s);
    return rc;
}

int orbit_op_conv2d(const float* x, int x_nchw, const float* w, float* y, const float* scale,
                               const float* shift, const float* residual, const float* gate, int B, int H,
                               int W, int Cin, int Cout, int KH, int KW, int stride, int pad_top,
                               int pad_left, int Ho, int Wo, int act, int pool2, orbit_stream_t stream) {
    ORBIT_REQUIRE(x && w && y, "op_conv2d: null pointer");
    ORBIT_REQUIRE(KH > 0 && KW > 0 && stride > 0, "op_conv2d: bad kernel geometry");
    hipStream_t s = (hipStream_t)stream;
    float* wp = nullptr;
    const size_t nfl = (conv_packed_floats(Cin, Cout, KH, KW, x_nchw) + 63) & ~(size_t)63;
    const size_t ffl = conv_frag_floats(Cin, Cout, KH, KW, x_nchw);
    ORBIT_HIP_CHECK(hipMallocAsync(reinterpret_cast<void**>(&wp), (nfl + ffl) * sizeof(float), s));
    int rc = conv_pack_weights(w, wp, Cin, Cout, KH, KW, x_nchw, s);
    if (rc == ORBIT_OK && ffl) rc = conv_frag_pack_weights(w, wp + nfl, Cin, Cout, s);
    if (rc == ORBIT_OK) {
        ConvDesc d;
        d.w_frag = ffl ? wp + nfl : nullptr;
        d.x = x, d.w_packed = wp, d.y = y, d.scale = scale, d.shift = shift, d.residual = residual;
        d.gate = gate, d.B = B, d.H = H, d.W = W, d.Cin = Cin, d.Cout = Cout, d.KH = KH, d.KW = KW;
        d.stride = stride, d.pad_t = pad_top, d.pad_l = pad_left, d.Ho = Ho, d.Wo = Wo, d.act = act;
        d.pool2 = pool2, d.x_nchw = x_nchw;
        float* sk = nullptr;
        const size_t skf = conv_splitk_floats(d);
        if (skf) ORBIT_HIP_CHECK(hipMallocAsync(reinterpret_cast<void**>(&sk), skf * sizeof(float), s));
        d.splitk_ws = sk;
        rc = launch_conv(d, s);
        if (sk) (void)hipFreeAsync(sk, s);
    }
    (void)hipFreeAsync(wp, s);
    return rc;
}

}  // extern "C"
