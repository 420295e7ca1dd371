// Row-streaming fused MBConv front half for the HIGH-RESOLUTION EfficientNet-B0 stages (112x112 .. 28x28 maps):
// expand 1x1 conv (fp32 MFMA) -> BN1 -> SiLU -> depthwise KxK (TF-SAME) -> BN2 -> SiLU -> squeeze-excite pooling partials,
// with the 6x-expanded tensor in an LDS ring of rows. Same arithmetic as csrc/mbconv.hip (timm InvertedResidual.forward's
// conv_pw -> bn1 -> act -> conv_dw -> bn2 -> act -> se.mean, reached from the reference's model/feature_extractors.py:39-43).
//
// Why another form. mbconv.hip tiles the output 8x8 / 4x8 and re-expands the (K-1)-pixel halo of every tile: 20-60 % more
// expand-GEMM rows, BN1/SiLU evaluations (the kernel's bound: two transcendentals per expanded element) and LDS
// scatter than the layer has, and five MFMA row tiles for four waves. The 5x5 blocks lose to the unfused pair for that
// reason and stayed unfused: their expanded tensor (361 / 150 MB per 200 frames) makes an HBM round trip.
// Here a block owns ONE 32-channel chunk of a full-width strip of the map and walks DOWN it: every step expands the next
// TO*S input rows (4 MFMA row tiles = one per wave, no halo: vertical neighbours are already in the ring; the only
// recomputed rows are the K-S rows at a band's top) and the depthwise stage consumes the rows in the ring:
//
//   ring  Es [3 windows][TO*S rows][SWi px][36]   window w = input rows r_first + w*TO*S ..; BN1+SiLU applied, ZERO
//                                                  outside the image (the depthwise conv zero-pads the EXPANDED tensor)
//   step i: expand(window i+2)  ||  depthwise(output rows i*TO .. from windows i, i+1)  -> ONE barrier
//
// The block input never touches LDS: a lane's MFMA A-fragments are 16-byte loads straight from the NHWC tensor (pixel
// = l31, channels 8g + 4*lh ..+3 - the k-order conv_igemm uses, so sums are bit-identical to the unfused conv), requested
// one window ahead; the chunk's B-fragments (expand weights) and BN1 vectors stay in registers for the whole walk.
// Pixels of a window are enumerated in ring-memory order (pad columns included and forced to zero), so an accumulator
// row's LDS address is linear in the tile row; the last tile is shifted back to end at the window's end (a few pixels are
// computed by two waves - same values) instead of guarding its tail.
#include "bf3.h"
#include "common.h"

namespace orbit {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using v4f = __attribute__((ext_vector_type(4))) float;

using v2f = __attribute__((ext_vector_type(2))) float;

// SiLU of two values: v * rcp(1 + exp2(-log2(e) * v)), the arithmetic of mbconv.hip's silu_f (bit-identical), with the
// multiplies and the add as packed-fp32 instructions (v_pk_fma_f32 costs 1.98 ns of a SIMD per 128 lanes-elements against
// 1.26 ns per 64 for v_fma_f32; the two transcendentals cost 3.9 ns each: profiles/r03_valu_probe.txt)
__device__ __forceinline__ v2f silu2(v2f x) {
    const v2f t = x * (v2f){-1.44269502f, -1.44269502f};
    const v2f e = {__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
    const v2f d = e + (v2f){1.0f, 1.0f};
    const v2f r = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    return x * r;
}
__device__ __forceinline__ v2f fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }

struct MbRowsParams {
    const float* x;    // [B][H][W][Cin] NHWC
    const float* w1;   // [mid][Cin]
    const float* sc1;  // [mid] folded BN1
    const float* sh1;
    const float* wdw;  // [K][K][mid]
    const float* sc2;  // [mid] folded BN2
    const float* sh2;
    float* y;          // [B][Ho][Wo][mid]
    float* pool;       // [B][tiles][mid] or nullptr
    int H, W, Cin, mid, pad_t, pad_l, Ho, Wo;
    int SWo, SWi, strips, band_rows, bands, nchunk, total;
};

constexpr int ROWS_ES = 36;  // ring pixel stride (floats): 32 channels + 4

// ---- the kernel. Its bookkeeping follows two measurements (tools/coexec_probe.hip, tools/valu_probe.hip;
// profiles/r03_coexec_probe.txt, r03_valu_probe.txt):
//  (1) on gfx950 a SIMD runs EITHER an MFMA OR VALU instructions, never both: a VALU stream under another wave's (or its
//      own) v_mfma_f32_32x32x2_f32 chain takes exactly MFMA time + VALU time, for fp32 and bf16 MFMAs alike. A fused kernel's
//      SIMD time is the SUM of its matrix and vector instruction time; specialising waves (expand waves / depthwise waves,
//      four per SIMD) was built and measured: 194 us against 195 us for block 1.1, i.e. nothing. Every VALU instruction
//      that is not arithmetic of the layer is therefore paid in full.
//  (2) round 2's form of this kernel spent 137 us of SIMD time per 200 frames of block 1.1 (MFMA 37 + VALU 100) for a 195 us
//      launch, and a third of that VALU time was not arithmetic: the masked expand epilogue (every 32-pixel tile of a 58-pixel
//      ring row holds a pad column, so a six-instruction-per-element select path always ran), per-item address arithmetic
//      with 32-bit integer multiplies (quarter rate) and a division by the group count, and ds_read_b128 bank conflicts
//      (0.54 conflict cycles per active cycle: with the 36-float pixel stride two of the four slots of a 16-lane service
//      group landed on the same banks). Its window prefetch also never overlapped anything (see `expand` below).
// This form keeps round 2's walk, MFMA k-order and every float operation (y is bit-identical to it and to the unfused pair; the
// pooling partials sum the same values in a different slot order) and changes the bookkeeping: validity is one bit-field extract +
// AND per expanded element; an item's ring addresses are `scalar window base + per-lane constant` (the item pattern of a
// step never changes); the output address is `scalar row pointer + per-lane constant`; and the lane -> (slot, channel quad)
// map puts slots u and u + 4 (288 floats = 32 banks apart) with both channel halves into each ds_read_b128 service group,
// which makes the reads conflict-free at the same 36-float stride.
// BF3 (opt-in `conv_bf3`, csrc/conv_bf3.hip): the expand GEMM on v_mfma_f32_32x32x16_bf16 with both operands split three ways
// (six products per fp32 product; a k-step of 16 = two of this kernel's 8-deep k-groups, the odd last group paired with zeros):
// 6 / 12 / 18 MFMAs of 32 cycles for Cin = 16 / 24 / 40 instead of 8 / 12 / 20 of 64 cycles; the window's x quads are split as
// they arrive (~22 VALU instructions per quad), the chunk's filter quads once per block. Not bit-identical to the unfused pair.
// RAW (round 6, train-mode BatchNorm on no-grad passes of the LITE step): the depthwise outputs are stored RAW - the second
// BatchNorm needs their batch statistics before it can be applied - and p.pool receives, per (frame, tile), the column sums and
// sums of squares of those raw outputs ([B * tiles][2][mid], the partial layout bn_stats_finalize reads) instead of the
// squeeze-excite pooling partials; sc1 / sh1 are the FIRST BatchNorm's batch-statistics scale / shift, which the caller gets
// from a statistics sweep of the expansion conv (ConvDesc::stats_only) - the expanded tensor itself exists only in the ring.
template <int K, int S, int TO, int NOUT, int NG, int SPR, bool EXACT, bool BF3, bool RAW = false>
__global__ __launch_bounds__(256, K == 3 ? 3 : 2) void mbconv_rows3_kernel(const MbRowsParams p) {
    constexpr int NEW = TO * S;
    constexpr int NCOL = (NOUT - 1) * S + K;
    constexpr int ES = ROWS_ES;
    static_assert((TO - 1) * S + K <= 2 * NEW, "an output step reads two windows");
    static_assert(NEW <= 4, "row-in-window index is packed in 2 bits");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int n_new = NEW * p.SWi;
    float* ring = smem;  // [3][n_new][ES]

    const int per = gridDim.x >> 3;
    const int v = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (v >= p.total) return;
    const int chunk = v % p.nchunk;
    const int t = v / p.nchunk;
    const int tiles = p.strips * p.bands;
    const int tile = t % tiles, b = t / tiles;
    const int strip = tile % p.strips, band = tile / p.strips;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int c0 = chunk * 32;
    const int x0 = strip * p.SWo, y0 = band * p.band_rows;
    const int y1 = y0 + p.band_rows < p.Ho ? y0 + p.band_rows : p.Ho;
    const int wi0 = x0 * S - p.pad_l;
    const int r_first = y0 * S - p.pad_t;
    const int NI = (y1 - y0 + TO - 1) / TO;

    // ---- expand stage constants
    const int tstart = wave * 32 < n_new - 32 ? wave * 32 : n_new - 32;
    int a_off, a_rl;
    bool a_colok;
    {
        const int f = tstart + l31;
        a_rl = f / p.SWi;
        const int col = f - a_rl * p.SWi, wi = wi0 + col;
        a_colok = (unsigned)wi < (unsigned)p.W;
        a_off = (a_rl * p.W + wi) * p.Cin + 4 * lh;
    }
    // validity bits of the lane's 16 accumulator elements, per row of the window: rowsel[r] = elements that lie in window
    // row r AND in a column inside the image (a window's valid elements = OR of rowsel[r] over its rows inside the image)
    unsigned rowsel[NEW];
#pragma unroll
    for (int r = 0; r < NEW; ++r) rowsel[r] = 0;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int f = tstart + 8 * (e >> 2) + 4 * lh + (e & 3);
        const int rl = f / p.SWi, col = f - rl * p.SWi;
        const bool cok = (unsigned)(wi0 + col) < (unsigned)p.W;
#pragma unroll
        for (int r = 0; r < NEW; ++r)
            if (cok && rl == r) rowsel[r] |= 1u << e;
    }
    v4f wb[NG];
    float s1 = 0.f, h1 = 0.f;
    {
        const int ch = c0 + l31;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            wb[g] = (v4f){0.f, 0.f, 0.f, 0.f};
            if (ch < p.mid) wb[g] = *reinterpret_cast<const v4f*>(p.w1 + (size_t)ch * p.Cin + 8 * g + 4 * lh);
        }
        if (ch < p.mid) s1 = p.sc1[ch], h1 = p.sh1[ch];
    }
    uint2 wq[BF3 ? 3 : 1][BF3 ? NG : 1];  // BF3: the filter quads' three bf16 planes
    if constexpr (BF3) {
#pragma unroll
        for (int g = 0; g < NG; ++g) split3(wb[g], wq[0][g], wq[1][g], wq[2][g]);
    }
    const float* xb = p.x + (size_t)b * p.H * p.W * p.Cin;
    v4f xa[NG];
    auto load_x = [&](int w) {
        const int hi0 = r_first + w * NEW;
        const bool ok = a_colok && (unsigned)(hi0 + a_rl) < (unsigned)p.H;
        const float* src = ok ? xb + (ptrdiff_t)hi0 * p.W * p.Cin + a_off : xb + 4 * lh;
#pragma unroll
        for (int g = 0; g < NG; ++g) xa[g] = *reinterpret_cast<const v4f*>(src + 8 * g);
    };
    float* const Ew0 = ring + tstart * ES + l31;
    // One load site per window, outside every branch: with the request inside the two arms of the `rowmask == 0` test the
    // compiler gave the arms different registers, joined them with moves and therefore put s_waitcnt vmcnt(0) right after the
    // loads - round 2's "prefetch" waited for its own data on every step.
    auto expand = [&](int w, int slot3) {  // slot3 = w % 3, tracked by the caller
        float* Ew = Ew0 + slot3 * n_new * ES;
        const int hi0 = r_first + w * NEW;
        unsigned rowmask = 0;
#pragma unroll
        for (int r = 0; r < NEW; ++r) rowmask |= ((unsigned)(hi0 + r) < (unsigned)p.H ? 1u : 0u) << r;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        if (rowmask != 0) {  // (a window entirely above / below the image is zeros: no arithmetic)
            if constexpr (BF3) {
                const uint2 z2 = {0u, 0u};
#pragma unroll
                for (int g = 0; g < NG; g += 2) {
                    uint2 xq[3][2];
                    split3(xa[g], xq[0][0], xq[1][0], xq[2][0]);
                    if (g + 1 < NG) split3(xa[g + 1], xq[0][1], xq[1][1], xq[2][1]);
                    else xq[0][1] = xq[1][1] = xq[2][1] = z2;
                    bf16x8 af[3], bf[3];
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        af[q] = bf3_frag(xq[q][0], xq[q][1]);
                        bf[q] = bf3_frag(wq[q][g], g + 1 < NG ? wq[q][g + 1] : z2);
                    }
                    // smallest products first; K <= 48, so the sums stay on the matrix cores (csrc/conv_bf3.hip keeps long sums
                    // out of the bf16 MFMA's accumulator)
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2], bf[0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[2], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bf[1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bf[0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[0], acc, 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int g = 0; g < NG; ++g)
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[g][kk], wb[g][kk], acc, 0, 0, 0);
            }
        }
        load_x(w + 1 <= NI ? w + 1 : NI);  // the last call re-requests window NI (unused)
        unsigned okbits = 0;
#pragma unroll
        for (int r = 0; r < NEW; ++r) okbits |= ((rowmask >> r) & 1u) ? rowsel[r] : 0u;
        const v2f s1v = {s1, s1}, h1v = {h1, h1};
        if (__all(okbits == 0xffffu)) {  // wave-uniform: no pad column and no row outside the image in this wave's tile
#pragma unroll
            for (int e = 0; e < 16; e += 2) {
                const v2f val = silu2(fma2((v2f){acc[e], acc[e + 1]}, s1v, h1v));
                Ew[(8 * (e >> 2) + 4 * lh + (e & 3)) * ES] = val.x;
                Ew[(8 * (e >> 2) + 4 * lh + (e & 3) + 1) * ES] = val.y;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 16; e += 2) {
                const v2f val = silu2(fma2((v2f){acc[e], acc[e + 1]}, s1v, h1v));
                const int m0 = __builtin_amdgcn_sbfe((int)okbits, e, 1), m1 = __builtin_amdgcn_sbfe((int)okbits, e + 1, 1);  // 0 / -1
                const float vx = val.x, vy = val.y;  // (__builtin_bit_cast of a vector ELEMENT reads element 0: copy to scalars first)
                Ew[(8 * (e >> 2) + 4 * lh + (e & 3)) * ES] = __int_as_float(__float_as_int(vx) & m0);
                Ew[(8 * (e >> 2) + 4 * lh + (e & 3) + 1) * ES] = __int_as_float(__float_as_int(vy) & m1);
            }
        }
    };

    // ---- depthwise stage constants. A wave's 16 four-lane chunks map to (slot-in-wave, channel half) so that each
    // ds_read_b128 service group ({chunks 0,3,5,6}, {1,2,4,7}, {8,11,13,14}, {9,10,12,15}) holds two slots 32 banks apart, each
    // with both halves: nibble c of the table = (slot << 1) | half
    // (2-pixel item pitch = 72 floats: slots u, u + 4 are 32 banks apart; 4-pixel pitch = 144 floats: slots u, u + 2)
    static_assert(NOUT * S == 2 || NOUT * S == 4, "lane map is built for a 2- or 4-pixel item pitch");
    const unsigned long long CHUNK_MAP = NOUT * S == 2 ? 0xFDCE5764B98A1320ull : 0xFDCE9BA875461320ull;
    const int ck = (int)((CHUNK_MAP >> (4 * (lane >> 2))) & 15u);
    const int lc = ((ck & 1) << 2) | (lane & 3);  // channel quad of the chunk
    const int u = wave * 8 + (ck >> 1);           // output slot 0 .. 31
    // EXACT (host-checked: strips tile the width, bands are whole steps, the last chunk is full or half full): every slot
    // stores unconditionally - an idle slot and a channel quad beyond `mid` recompute a real owner's output and store the same
    // bits to the same address. The point is the instruction count, not the few lanes: with a FIXED number of stores after the
    // window prefetch the compiler waits for the prefetch alone (s_waitcnt vmcnt(stores)); with stores under a branch it has
    // to use vmcnt(0), which also waits for the stores issued a moment ago
    const bool q_real = c0 + lc * 4 < p.mid;
    const int lce = EXACT && !q_real ? lc - 4 : lc;  // the quad this thread computes
    const int cq = c0 + lce * 4;
    const bool q_ok = EXACT || q_real;
    static_assert(TO % (32 / SPR) == 0, "a pass covers whole rows of the step");
    v4f s2 = {0.f, 0.f, 0.f, 0.f}, h2 = {0.f, 0.f, 0.f, 0.f};
    if (!RAW && q_ok) s2 = *reinterpret_cast<const v4f*>(p.sc2 + cq), h2 = *reinterpret_cast<const v4f*>(p.sh2 + cq);
    // 3x3: the thread's nine tap quads stay in registers; 5x5: 25 quads do not fit, the chunk's taps sit in LDS behind the ring
    v4f* Ds = reinterpret_cast<v4f*>(ring + 3 * n_new * ES);  // 5x5 only: [K*K][8]
    v4f tapr[K == 3 ? K * K : 1];
    if constexpr (K == 3) {
#pragma unroll
        for (int tap = 0; tap < K * K; ++tap)
            tapr[tap] = q_ok ? *reinterpret_cast<const v4f*>(p.wdw + (size_t)tap * p.mid + cq) : (v4f){0.f, 0.f, 0.f, 0.f};
    } else if (tid < K * K * 8) {
        const int tq = c0 + (tid & 7) * 4;
        Ds[tid] = tq < p.mid ? *reinterpret_cast<const v4f*>(p.wdw + (size_t)(tid >> 3) * p.mid + tq) : (v4f){0.f, 0.f, 0.f, 0.f};
    }
    // Items of a step: output row j, column group g (NOUT outputs). Slot u serves row u / SPR and group u % SPR of each
    // pass (SPR = 32, 16 or 8 slots per row, a power of two: no division, and with SPR = 32 the row of an item is the pass
    // number - every ring row address is then `scalar + per-lane column constant`)
    constexpr int RPP = 32 / SPR;                // rows per pass
    constexpr int NPASS = (TO + RPP - 1) / RPP;
    const int G = p.SWo / NOUT;                  // <= SPR (rows_geom)
    const int jl = u / SPR, g = u % SPR;
    const int gc = g < G ? g : G - 1;            // idle slots recompute the last group and store nothing
    const int rowfl = p.SWi * ES;                // floats per ring row
    const int colpart = gc * (NOUT * S * ES) + lce * 4;
    const int o_col = (x0 + gc * NOUT) * p.mid + cq;
    unsigned okn = 0;                            // which of the NOUT columns this thread stores
#pragma unroll
    for (int n = 0; n < NOUT; ++n)
        if (g < G && q_ok && x0 + g * NOUT + n < p.Wo) okn |= 1u << n;
    const bool own = g < G && q_real;            // EXACT: this slot owns its outputs (the others store duplicates)
    float* ystep = p.y + ((size_t)b * p.Ho + y0) * p.Wo * p.mid;
    const int ystride = TO * p.Wo * p.mid;
    const int orow = p.Wo * p.mid;
    const int band_len = y1 - y0;
    const int winfl = n_new * ES;  // floats per window
    v4f psum = {0.f, 0.f, 0.f, 0.f}, psq = {0.f, 0.f, 0.f, 0.f};

    auto depthwise = [&](int i, int slotA) {  // slotA = i % 3
        const int baseA = slotA * winfl;
        const int baseB = slotA == 2 ? 0 : baseA + winfl;
        v2f alo[NOUT], ahi[NOUT];
        v4f cA[NCOL], cB[NCOL];
        auto fetch = [&](v4f* c, int pass, int kh) {
            const float* erow;
            if constexpr (RPP == 1) {
                const int rr = pass * S + kh;  // wave-uniform
                erow = ring + ((rr >= NEW ? baseB + (rr - NEW) * rowfl : baseA + rr * rowfl) + colpart);
            } else {
                const int rr = (pass * RPP + jl) * S + kh;  // per lane
                erow = ring + ((rr >= NEW ? baseB - NEW * rowfl : baseA) + __mul24(rr, rowfl) + colpart);
            }
#pragma unroll
            for (int q = 0; q < NCOL; ++q) c[q] = *reinterpret_cast<const v4f*>(erow + q * ES);
        };
        auto mac = [&](const v4f* c, const v4f* t) {
#pragma unroll
            for (int kw = 0; kw < K; ++kw) {
                const v2f flo = {t[kw][0], t[kw][1]}, fhi = {t[kw][2], t[kw][3]};
#pragma unroll
                for (int n = 0; n < NOUT; ++n) {
                    const v4f cv = c[n * S + kw];
                    alo[n] = fma2((v2f){cv[0], cv[1]}, flo, alo[n]);
                    ahi[n] = fma2((v2f){cv[2], cv[3]}, fhi, ahi[n]);
                }
            }
        };
        fetch(cA, 0, 0);
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
#pragma unroll
            for (int n = 0; n < NOUT; ++n) alo[n] = (v2f){0.f, 0.f}, ahi[n] = (v2f){0.f, 0.f};
            if constexpr (K == 3) {
#pragma unroll
                for (int kh = 0; kh < K; ++kh) {
                    const int f = pass * K + kh;  // running row index: the two row buffers alternate across passes too
                    v4f* cur = (f & 1) ? cB : cA;
                    v4f* nxt = (f & 1) ? cA : cB;
                    if (kh + 1 < K) fetch(nxt, pass, kh + 1);
                    else if (pass + 1 < NPASS) fetch(nxt, pass + 1, 0);  // the next pass's first row arrives under this epilogue
                    mac(cur, tapr + kh * K);
                }
            } else {
                // 5x5 (one pass per step): taps come from LDS with their row, the loop over row pairs stays rolled (unrolled,
                // hipcc hoists every row's reads and spills)
                static_assert(K == 3 || NPASS == 1, "the 5x5 row loop assumes one pass");
                v4f tA[K], tB[K];
                auto taps = [&](v4f* t, int kh) {
#pragma unroll
                    for (int kw = 0; kw < K; ++kw) t[kw] = Ds[(kh * K + kw) * 8 + lce];
                };
                taps(tA, 0);
#pragma unroll 1
                for (int kh = 0; kh + 1 < K; kh += 2) {
                    fetch(cB, pass, kh + 1), taps(tB, kh + 1);
                    mac(cA, tA);
                    if (kh + 2 < K) fetch(cA, pass, kh + 2), taps(tA, kh + 2);
                    mac(cB, tB);
                }
                if (K & 1) mac(cA, tA);
            }
            const int j = pass * RPP + jl;
            const bool rowok = j < TO && i * TO + j < band_len;
            float* yrow = ystep + j * orow;
#pragma unroll
            for (int n = 0; n < NOUT; ++n) {
                if constexpr (RAW) {
                    const v4f o = {alo[n].x, alo[n].y, ahi[n].x, ahi[n].y};
                    const bool mine = EXACT ? own : (rowok && ((okn >> n) & 1u));
                    if (EXACT || mine) *reinterpret_cast<v4f*>(yrow + (o_col + n * p.mid)) = o;
                    {
#pragma clang fp contract(off)  // (one rounding per product and per sum, whatever the instantiation)
                        const v4f z = {0.f, 0.f, 0.f, 0.f};
                        psum = psum + (mine ? o : z);
                        psq = psq + (mine ? o * o : z);
                    }
                } else if constexpr (EXACT) {
                    const v2f olo = silu2(fma2(alo[n], (v2f){s2[0], s2[1]}, (v2f){h2[0], h2[1]}));
                    const v2f ohi = silu2(fma2(ahi[n], (v2f){s2[2], s2[3]}, (v2f){h2[2], h2[3]}));
                    const v4f o = {olo.x, olo.y, ohi.x, ohi.y};
                    *reinterpret_cast<v4f*>(yrow + (o_col + n * p.mid)) = o;
                    psum += own ? o : (v4f){0.f, 0.f, 0.f, 0.f};
                } else if (rowok && ((okn >> n) & 1u)) {
                    const v2f olo = silu2(fma2(alo[n], (v2f){s2[0], s2[1]}, (v2f){h2[0], h2[1]}));
                    const v2f ohi = silu2(fma2(ahi[n], (v2f){s2[2], s2[3]}, (v2f){h2[2], h2[3]}));
                    const v4f o = {olo.x, olo.y, ohi.x, ohi.y};
                    *reinterpret_cast<v4f*>(yrow + (o_col + n * p.mid)) = o;
                    {
#pragma clang fp contract(off)  // sum the ROUNDED outputs (what y holds), as the branch-free instantiation does
                        psum = psum + o;
                    }
                }
            }
        }
        ystep += ystride;
    };

    load_x(0);
    expand(0, 0);
    expand(1, 1);
    __syncthreads();
    // every request made so far (taps, BN vectors, window 2) has landed before the loop: the loop then sees the same queue on
    // entry and on its back edge - the window prefetch followed by a fixed number of stores - and waits with vmcnt(stores)
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    int sa = 0;  // i % 3
    for (int i = 0; i < NI; ++i) {
        if (i + 2 <= NI) expand(i + 2, sa == 0 ? 2 : sa - 1);  // (i + 2) % 3
        depthwise(i, sa);
        sa = sa == 2 ? 0 : sa + 1;
        __syncthreads();
    }
    if (p.pool) {
        v4f* red = reinterpret_cast<v4f*>(ring);
        red[u * 8 + lc] = psum;
        if constexpr (RAW) red[256 + u * 8 + lc] = psq;  // (the ring holds >= 3 * 33 * 36 floats: room for 512 quads)
        __syncthreads();
        if (tid < 8 && c0 + tid * 4 < p.mid) {
            v4f t4 = red[tid];
            for (int sl = 1; sl < 32; ++sl) t4 += red[sl * 8 + tid];
            if constexpr (RAW) {
                v4f q4 = red[256 + tid];
                for (int sl = 1; sl < 32; ++sl) q4 += red[256 + sl * 8 + tid];
                float* row = p.pool + ((size_t)b * tiles + tile) * 2 * p.mid + c0 + tid * 4;
                *reinterpret_cast<v4f*>(row) = t4;
                *reinterpret_cast<v4f*>(row + p.mid) = q4;
            } else {
                *reinterpret_cast<v4f*>(p.pool + ((size_t)b * tiles + tile) * p.mid + c0 + tid * 4) = t4;
            }
        }
    }
}

// ---- geometry: per (K, stride) the output strip width and the band height are fixed by the map size alone (the pooling
// partial count is part of the network plan, which does not know the batch size)
struct RowsGeom {
    int SWo, SWi, strips, band_rows, bands, TO, NOUT;
};
static bool rows_geom(int H, int W, int Cin, int mid, int K, int stride, int Ho, int Wo, RowsGeom& g) {
    if (Cin % 8 != 0 || mid % 4 != 0 || H < 1 || W < 1) return false;
    const int ng = Cin / 8;
    // (K, stride, Cin) -> rows per step, outputs per depthwise item, widest strip whose window fits 4 MFMA row tiles
    int swo_max = 0;
    if (K == 3 && stride == 2 && ng == 2) g.TO = 1, g.NOUT = 1, swo_max = 28;       // 16 -> 96 3x3/2 (112x112 at 224)
    else if (K == 3 && stride == 1 && ng == 3) g.TO = 2, g.NOUT = 2, swo_max = 56;  // 24 -> 144 3x3/1 (56x56)
    else if (K == 5 && stride == 2 && ng == 3) g.TO = 2, g.NOUT = 1, swo_max = 14;  // 24 -> 144 5x5/2 (56x56)
    else if (K == 5 && stride == 1 && ng == 5) g.TO = 4, g.NOUT = 4, swo_max = 28;  // 40 -> 240 5x5/1 (28x28)
    else if (K == 3 && stride == 2 && ng == 5) g.TO = 2, g.NOUT = 1, swo_max = 14;  // 40 -> 240 3x3/2 (28x28)
    else return false;
    // equal strips (a map that is not a multiple of the widest strip is not served by one full and one sliver)
    g.strips = cdiv(Wo, swo_max);
    g.SWo = cdiv(cdiv(Wo, g.strips), g.NOUT) * g.NOUT;
    g.SWi = (g.SWo - 1) * stride + K;
    const int n_new = g.TO * stride * g.SWi;
    if (n_new <= 32 || n_new > 128) return false;
    g.band_rows = 28;  // measured: 28-row bands beat 14 (fewer pipeline fills) and 56 (too few blocks) on the whole task
    g.band_rows = cdiv(g.band_rows, g.TO) * g.TO;
    g.bands = cdiv(Ho, g.band_rows);
    return true;
}

bool mbconv_rows_supported(int H, int W, int Cin, int mid, int K, int stride) {
    RowsGeom g;
    const int Ho = cdiv(H, stride), Wo = cdiv(W, stride);
    return rows_geom(H, W, Cin, mid, K, stride, Ho, Wo, g);
}

int mbconv_rows_tiles(int H, int W, int Cin, int mid, int K, int stride) {
    RowsGeom g;
    const int Ho = cdiv(H, stride), Wo = cdiv(W, stride);
    if (!rows_geom(H, W, Cin, mid, K, stride, Ho, Wo, g)) return 0;
    return g.strips * g.bands;
}

int launch_mbconv_rows(const float* x, const float* w1, const float* sc1, const float* sh1, const float* wdw,
                       const float* sc2, const float* sh2, float* y, float* pool, int B, int H, int W, int Cin, int mid,
                       int K, int stride, int pad_t, int pad_l, int Ho, int Wo, hipStream_t s, int plan_tiles, bool raw_stats) {
    ORBIT_REQUIRE(x && w1 && sc1 && sh1 && wdw && y && (raw_stats ? pool != nullptr : (sc2 && sh2)), "mbconv_rows: null pointer");
    RowsGeom g;
    ORBIT_REQUIRE(Ho == cdiv(H, stride) && Wo == cdiv(W, stride) && rows_geom(H, W, Cin, mid, K, stride, Ho, Wo, g),
                  "mbconv_rows: unsupported shape (H=%d W=%d Cin=%d mid=%d K=%d s=%d)", H, W, Cin, mid, K, stride);
    // a plan sized its pooling partials (and the SE gate sums them) for a tile count: refuse to write a different number
    ORBIT_REQUIRE(plan_tiles <= 0 || plan_tiles == g.strips * g.bands,
                  "mbconv_rows: the plan was built for %d pooling tiles per frame, this launch writes %d", plan_tiles, g.strips * g.bands);
    MbRowsParams p;
    p.x = x, p.w1 = w1, p.sc1 = sc1, p.sh1 = sh1, p.wdw = wdw, p.sc2 = sc2, p.sh2 = sh2, p.y = y, p.pool = pool;
    p.H = H, p.W = W, p.Cin = Cin, p.mid = mid, p.pad_t = pad_t, p.pad_l = pad_l, p.Ho = Ho, p.Wo = Wo;
    p.SWo = g.SWo, p.SWi = g.SWi, p.strips = g.strips, p.band_rows = g.band_rows, p.bands = g.bands;
    p.nchunk = cdiv(mid, 32);
    p.total = p.nchunk * g.strips * g.bands * B;
    const int grid = cdiv(p.total, 8) * 8;
    const int n_new = g.TO * stride * g.SWi;
    const size_t lds = (size_t)3 * n_new * ROWS_ES * sizeof(float) + (K == 3 ? 0 : (size_t)K * K * 8 * 16);
    const double pix = (double)B * H * W;
    const int rec = prof_start(raw_stats ? "mbconv_rows,raw" : "mbconv_rows", 2.0 * pix * Cin * mid + 2.0 * B * Ho * Wo * mid * K * K,
                               4.0 * (pix * Cin + (double)B * Ho * Wo * mid), s, pix * mid + (raw_stats ? 0.0 : (double)B * Ho * Wo * mid));
#define ORBIT_MBR3(KK, SS, TO_, NOUT_, NG_, SPR_, EX_, BF_)                                                       \
    do {                                                                                                \
        if (raw_stats) {                                                                                \
            if constexpr (!BF_) {                                                                       \
                auto kern = mbconv_rows3_kernel<KK, SS, TO_, NOUT_, NG_, SPR_, EX_, false, true>;           \
                static bool attr_set = false;                                                           \
                if (!attr_set) {                                                                        \
                    ORBIT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),            \
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024)); \
                    attr_set = true;                                                                    \
                }                                                                                       \
                kern<<<grid, 256, lds, s>>>(p);                                                         \
            }                                                                                           \
            break;                                                                                      \
        }                                                                                               \
        auto kern = mbconv_rows3_kernel<KK, SS, TO_, NOUT_, NG_, SPR_, EX_, BF_>;                                \
        static bool attr_set = false;                                                                   \
        if (!attr_set) {                                                                                \
            ORBIT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                    \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024)); \
            attr_set = true;                                                                            \
        }                                                                                               \
        kern<<<grid, 256, lds, s>>>(p);                                                                 \
    } while (0)
    const int ng = Cin / 8;
    // branch-free stores need: strips tile the width exactly, every band is a whole number of steps, and the last 32-channel
    // chunk is full or exactly half full (a missing quad duplicates the quad 16 channels below)
    const bool exact = g.strips * g.SWo == Wo && Ho % g.TO == 0 && g.band_rows % g.TO == 0 && (mid % 32 == 0 || mid % 32 == 16) &&
                       g.SWo % g.NOUT == 0;
    const bool bf3 = !raw_stats && (get_option("conv_bf3") & 2) != 0;  // (opt-in bit 2; the exact-tiling instantiations only)
#define ORBIT_MBR3X(KK, SS, TO_, NOUT_, NG_, SPR_)                    \
    do {                                                              \
        if (exact && bf3) ORBIT_MBR3(KK, SS, TO_, NOUT_, NG_, SPR_, true, true);   \
        else if (exact) ORBIT_MBR3(KK, SS, TO_, NOUT_, NG_, SPR_, true, false);    \
        else ORBIT_MBR3(KK, SS, TO_, NOUT_, NG_, SPR_, false, false);              \
    } while (0)
    if (K == 3 && stride == 2 && ng == 2) ORBIT_MBR3X(3, 2, 1, 1, 2, 32);  // the (TO, NOUT) of rows_geom; SPR >= SWo / NOUT
    else if (K == 3 && stride == 1 && ng == 3) ORBIT_MBR3X(3, 1, 2, 2, 3, 32);
    else if (K == 5 && stride == 2 && ng == 3) ORBIT_MBR3X(5, 2, 2, 1, 3, 16);
    else if (K == 5 && stride == 1 && ng == 5) ORBIT_MBR3X(5, 1, 4, 4, 5, 8);
    else if (exact) ORBIT_MBR3(3, 2, 2, 1, 5, 16, true, false);  // (its BF3 form needs 11 more registers than 3 waves per SIMD leave)
    else ORBIT_MBR3(3, 2, 2, 1, 5, 16, false, false);
#undef ORBIT_MBR3X
#undef ORBIT_MBR3
    prof_stop(rec, s);
    ORBIT_LAUNCH_CHECK();
    return ORBIT_OK;
}


// ---- stem form: conv_stem (NCHW frames, 3 -> 32 channels, 3x3 stride 2, TF-SAME) + BN + SiLU + blocks.0.0.conv_dw 3x3/1 +
// BN + SiLU + squeeze-excite pooling partials with the same walk (timm tf_efficientnet_b0: conv_stem -> bn1 -> act ->
// DepthwiseSeparableConv.conv_dw -> bn1 -> act, reached from the reference's model/feature_extractors.py:39-43). Unfused, the
// stem's 112x112x32 output makes an HBM round trip (321 MB written + read per 200 frames, the largest tensor of the net).
// The "expand" stage is an im2col GEMM on the matrix cores (one window = 2 stem rows x <= 30 columns <= 64 pixels, 28
// taps, 32 channels: v_mfma_f32_16x16x4_f32), its A operand gathered from an LDS frame patch (5 frame rows x 61 columns x
// 3 planes, double-buffered, refilled one window ahead through registers), its B operand (the [32][32]-packed filter) in LDS
// in lane order. (First form: a direct VALU convolution, lane = pixel, wave = 8 channels whose 27 taps arrived as scalar
// loads in SGPRs - 108 packed FMAs with scalar operands per wave and step were 5 k of a step's 8 k cycles: 245 us per 200
// frames against 225 us for this form. With the filter in registers the MFMA form needs 182 VGPRs = two blocks per CU and
// loses its advantage; at 168 it runs three.)
struct StemRowsParams {
    const float* frames;  // [B][3][FH][FW]
    const float* w1;      // [32][32]: channel-major, tap ci*9 + kh*3 + kw (stem_pack_weights)
    const float* sc1;
    const float* sh1;
    const float* wdw;     // [3][3][32]
    const float* sc2;
    const float* sh2;
    float* y;             // [B][H][W][32]
    float* pool;          // [B][tiles][32] or nullptr
    int FH, FW, spad_t, spad_l, H, W;  // (H, W) = stem output grid = depthwise grid
    int SWo, SWi, strips, band_rows, bands, total;
};

constexpr int STEM_PW = 64;  // patch row stride (floats): 2 * 30 + 1 columns

template <bool EXACT>
__global__ __launch_bounds__(256, 3) void stem_rows_kernel(const StemRowsParams p, const float* __restrict__ w1g,
                                                           const float* __restrict__ sc1g, const float* __restrict__ sh1g) {
    constexpr int K = 3, TO = 2, NEW = 2, NOUT = 2, NCOL = 4, ES = ROWS_ES;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int n_new = NEW * p.SWi;                  // <= 64 ring pixels per window
    float* ring = smem;                             // [3][n_new][ES]
    float* patch = ring + 3 * n_new * ES;           // [2][3 planes][5 rows][STEM_PW]
    constexpr int PATCH = 3 * 5 * STEM_PW;

    const int per = gridDim.x >> 3;
    const int v = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (v >= p.total) return;
    const int tiles = p.strips * p.bands;
    const int tile = v % tiles, b = v / tiles;
    const int strip = tile % p.strips, band = tile / p.strips;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int x0 = strip * p.SWo, y0 = band * p.band_rows;
    const int y1 = y0 + p.band_rows < p.H ? y0 + p.band_rows : p.H;
    const int c_first = x0 - 1, r_first = y0 - 1;   // stem column / row of ring column 0 / window 0 row 0 (dw pad 1)
    const int NI = (y1 - y0 + TO - 1) / TO;
    const float* fb = p.frames + (size_t)b * 3 * p.FH * p.FW;

    // ---- frame patch of window w: frame rows 2 * (r_first + 2w) - spad_t + 0..4, columns 2 * c_first - spad_l + 0 .. 2 SWi
    const int fcol0 = 2 * c_first - p.spad_l;
    const int pcols = 2 * p.SWi + 1;
    constexpr int NPL = (PATCH + 255) / 256;        // patch elements per thread
    float preg[NPL];
    unsigned pmask = 0;  // which of preg[] lie inside the frame: applied when the patch is WRITTEN - a select on the loaded
                         // value would make the wave wait for the load where it is issued (no prefetch at all)
    // per-thread constants of its NPL patch elements (the patch pattern is the same for every window): frame offset for
    // patch row 0 at frame row 0, column validity, patch row - a window then costs one add and one bit test per element
    // (decoding (plane, row, column) and the 64-bit frame offset per element and step was a third of the kernel's VALU work)
    int poff[NPL];
    unsigned pcolok = 0, prows = 0;  // bit u / 3-bit field u
#pragma unroll
    for (int u = 0; u < NPL; ++u) {
        const int i = tid + u * 256;
        const int c = i & (STEM_PW - 1), pr = i >> 6;      // pr = plane * 5 + row
        const int plane = pr / 5, row = pr - plane * 5;
        const int fc = fcol0 + c;
        const bool cok = i < PATCH && c < pcols && (unsigned)fc < (unsigned)p.FW;
        poff[u] = cok ? (plane * p.FH + row) * p.FW + fc : 0;
        pcolok |= (cok ? 1u : 0u) << u;
        prows |= (unsigned)row << (3 * u);
    }
    auto patch_load = [&](int w) {
        const int frow0 = 2 * (r_first + NEW * w) - p.spad_t;
        unsigned rmask = 0;  // patch rows inside the frame (wave-uniform)
#pragma unroll
        for (int r = 0; r < 5; ++r) rmask |= ((unsigned)(frow0 + r) < (unsigned)p.FH ? 1u : 0u) << r;
        const int rowbase = frow0 * p.FW;
        pmask = 0;
#pragma unroll
        for (int u = 0; u < NPL; ++u) {
            const bool ok = ((pcolok >> u) & 1u) && ((rmask >> ((prows >> (3 * u)) & 7u)) & 1u);
            pmask |= (ok ? 1u : 0u) << u;
            preg[u] = fb[ok ? poff[u] + rowbase : 0];
        }
    };
    auto patch_store = [&](int w) {
        float* dst = patch + (w & 1) * PATCH;
#pragma unroll
        for (int u = 0; u < NPL; ++u) {
            const int i = tid + u * 256;
            if (i < PATCH) dst[i] = ((pmask >> u) & 1u) ? preg[u] : 0.f;
        }
    };

    // ---- stem stage on the matrix cores: per window an im2col GEMM [<= 64 pixels][28 taps] x [28][32 channels] with
    // v_mfma_f32_16x16x4_f32. Wave w owns pixels 16w .. 16w+15 and both 16-channel halves: 7 k-steps x 2 MFMAs. A lane's
    // A element of step s is ONE patch read (pixel l % 16, tap 4s + l / 16; tap 27 is padding with a zero weight); its B
    // elements (the filter) sit in LDS in lane order ([14][64] floats) - registers decide the blocks per CU here.
    using f32x4 = __attribute__((ext_vector_type(4))) float;
    const int l15 = lane & 15, lq = lane >> 4;
    float* Wl = patch + 2 * PATCH;                   // [2 halves][7 steps][64 lanes]
    for (int i = tid; i < 14 * 64; i += 256) {
        const int ln = i & 63, hs = i >> 6, h = hs / 7, st = hs - h * 7;
        const int k = 4 * st + (ln >> 4);
        Wl[i] = k < 27 ? w1g[(16 * h + (ln & 15)) * 32 + k] : 0.f;
    }
    int a_base;          // patch offset of this lane's A pixel
    {
        const int fa = wave * 16 + l15 < n_new ? wave * 16 + l15 : n_new - 1;
        const int rl = fa / p.SWi, col = fa - rl * p.SWi;
        a_base = (2 * rl) * STEM_PW + 2 * col;
    }
    float s1c[2], h1c[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) s1c[h] = sc1g[16 * h + l15], h1c[h] = sh1g[16 * h + l15];
    // output element r of a lane: pixel 16w + 4 lq + r (C/D layout of the 16x16 MFMA: column = l15, row = 4 lq + r).
    // Validity as bit fields (as in mbconv_rows3_kernel): sel[r2] = elements in window row r2 and in a column of the image
    unsigned o_inwin = 0, sel0 = 0, sel1 = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int fo = wave * 16 + 4 * lq + r;
        const int rl = fo / p.SWi, col = fo - rl * p.SWi;
        if (fo < n_new) o_inwin |= 1u << r;
        if (fo < n_new && (unsigned)(c_first + col) < (unsigned)p.W) (rl ? sel1 : sel0) |= 1u << r;
    }
    auto stem = [&](int w) {
        const float* P = patch + (w & 1) * PATCH + a_base;
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int st = 0; st < 7; ++st) {
            const int k = 4 * st + lq < 27 ? 4 * st + lq : 0;          // tap of this lane in step st
            const int ci = k / 9, kh = (k - ci * 9) / 3, kw = k - ci * 9 - kh * 3;
            const float av = P[(ci * 5 + kh) * STEM_PW + kw];
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, Wl[st * 64 + lane], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, Wl[(7 + st) * 64 + lane], acc1, 0, 0, 0);
        }
        const int hi0 = r_first + NEW * w;
        const unsigned okbits = ((unsigned)hi0 < (unsigned)p.H ? sel0 : 0u) | ((unsigned)(hi0 + 1) < (unsigned)p.H ? sel1 : 0u);
        float* E = ring + ((w % 3) * n_new + wave * 16 + 4 * lq) * ES + l15;
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
            const v2f v0 = silu2(fma2((v2f){acc0[r], acc0[r + 1]}, (v2f){s1c[0], s1c[0]}, (v2f){h1c[0], h1c[0]}));
            const v2f v1 = silu2(fma2((v2f){acc1[r], acc1[r + 1]}, (v2f){s1c[1], s1c[1]}, (v2f){h1c[1], h1c[1]}));
            const float a0 = v0.x, a1 = v0.y, b0 = v1.x, b1 = v1.y;
            const int m0 = __builtin_amdgcn_sbfe((int)okbits, r, 1), m1 = __builtin_amdgcn_sbfe((int)okbits, r + 1, 1);  // 0 / -1
            if ((o_inwin >> r) & 1u) {
                E[r * ES] = __int_as_float(__float_as_int(a0) & m0);
                E[r * ES + 16] = __int_as_float(__float_as_int(b0) & m0);
            }
            if ((o_inwin >> (r + 1)) & 1u) {
                E[(r + 1) * ES] = __int_as_float(__float_as_int(a1) & m1);
                E[(r + 1) * ES + 16] = __int_as_float(__float_as_int(b1) & m1);
            }
        }
    };

    // ---- depthwise stage (the bookkeeping of mbconv_rows3_kernel: conflict-free lane map for the 2-pixel item pitch, slot u
    // = row u / 16 and column group u % 16 of the step, ring address = window base + per-lane constant, EXACT = every slot
    // stores, idle slots duplicate the last group)
    const unsigned long long CHUNK_MAP = 0xFDCE5764B98A1320ull;
    const int ck = (int)((CHUNK_MAP >> (4 * (lane >> 2))) & 15u);
    const int lc = ((ck & 1) << 2) | (lane & 3);
    const int u = wave * 8 + (ck >> 1);
    const int cq = lc * 4;
    const v4f s2 = *reinterpret_cast<const v4f*>(p.sc2 + cq), h2 = *reinterpret_cast<const v4f*>(p.sh2 + cq);
    v4f tapr[K * K];
#pragma unroll
    for (int tap = 0; tap < K * K; ++tap) tapr[tap] = *reinterpret_cast<const v4f*>(p.wdw + tap * 32 + cq);
    const int G = p.SWo / NOUT;  // <= 14
    const int jl = u >> 4, g = u & 15;
    const int gc = g < G ? g : G - 1;
    const int rowfl = p.SWi * ES, winfl = n_new * ES;
    const int colpart = gc * (NOUT * ES) + lc * 4;
    const int o_col = (jl * p.W + x0 + gc * NOUT) * 32 + cq;
    const bool own = g < G;
    unsigned okn = 0;
#pragma unroll
    for (int n = 0; n < NOUT; ++n)
        if (g < G && x0 + g * NOUT + n < p.W) okn |= 1u << n;
    float* ystep = p.y + ((size_t)b * p.H + y0) * p.W * 32;
    const int band_len = y1 - y0;
    v4f psum = {0.f, 0.f, 0.f, 0.f};
    auto depthwise = [&](int i, int slotA) {  // slotA = i % 3
        const int baseA = slotA * winfl;
        const int baseB = slotA == 2 ? 0 : baseA + winfl;
        v2f alo[NOUT], ahi[NOUT];
#pragma unroll
        for (int n = 0; n < NOUT; ++n) alo[n] = (v2f){0.f, 0.f}, ahi[n] = (v2f){0.f, 0.f};
        v4f cA[NCOL], cB[NCOL];
        auto fetch = [&](v4f* c, int kh) {
            const int rr = jl + kh;  // per lane
            const float* erow = ring + ((rr >= NEW ? baseB - NEW * rowfl : baseA) + __mul24(rr, rowfl) + colpart);
#pragma unroll
            for (int q = 0; q < NCOL; ++q) c[q] = *reinterpret_cast<const v4f*>(erow + q * ES);
        };
        fetch(cA, 0);
#pragma unroll
        for (int kh = 0; kh < K; ++kh) {
            v4f* cur = (kh & 1) ? cB : cA;
            v4f* nxt = (kh & 1) ? cA : cB;
            if (kh + 1 < K) fetch(nxt, kh + 1);
#pragma unroll
            for (int kw = 0; kw < K; ++kw) {
                const v4f t = tapr[kh * K + kw];
                const v2f flo = {t[0], t[1]}, fhi = {t[2], t[3]};
#pragma unroll
                for (int n = 0; n < NOUT; ++n) {
                    const v4f cv = cur[n + kw];
                    alo[n] = fma2((v2f){cv[0], cv[1]}, flo, alo[n]);
                    ahi[n] = fma2((v2f){cv[2], cv[3]}, fhi, ahi[n]);
                }
            }
        }
        const bool rowok = i * TO + jl < band_len;
#pragma unroll
        for (int n = 0; n < NOUT; ++n) {
            if constexpr (EXACT) {
                const v2f olo = silu2(fma2(alo[n], (v2f){s2[0], s2[1]}, (v2f){h2[0], h2[1]}));
                const v2f ohi = silu2(fma2(ahi[n], (v2f){s2[2], s2[3]}, (v2f){h2[2], h2[3]}));
                const v4f o = {olo.x, olo.y, ohi.x, ohi.y};
                *reinterpret_cast<v4f*>(ystep + (o_col + n * 32)) = o;
                psum += own ? o : (v4f){0.f, 0.f, 0.f, 0.f};
            } else if (rowok && ((okn >> n) & 1u)) {
                const v2f olo = silu2(fma2(alo[n], (v2f){s2[0], s2[1]}, (v2f){h2[0], h2[1]}));
                const v2f ohi = silu2(fma2(ahi[n], (v2f){s2[2], s2[3]}, (v2f){h2[2], h2[3]}));
                const v4f o = {olo.x, olo.y, ohi.x, ohi.y};
                *reinterpret_cast<v4f*>(ystep + (o_col + n * 32)) = o;
                {
#pragma clang fp contract(off)  // sum the ROUNDED outputs (what y holds), as the branch-free instantiation does
                    psum = psum + o;
                }
            }
        }
        ystep += TO * p.W * 32;
    };

    // ---- the walk. Step w: request patch w+1, stem window w (patch w & 1), depthwise of output step w-2, park patch w+1.
    // EXACT: the stores between the patch request and its parking are a fixed number, so the parking waits with
    // vmcnt(stores) for the patch alone (see mbconv_rows3_kernel)
    patch_load(0);
    patch_store(0);
    __syncthreads();
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): taps / BN vectors have landed, the loop's queue is the same on entry and back edge
    int sa = 0;  // (w - 2) % 3
    for (int w = 0; w <= NI + 1; ++w) {
        patch_load(w + 1 <= NI ? w + 1 : NI);  // (the last two requests repeat patch NI and are not parked)
        if (w <= NI) stem(w);
        if (w >= 2) {
            depthwise(w - 2, sa);
            sa = sa == 2 ? 0 : sa + 1;
        }
        if (w + 1 <= NI) patch_store(w + 1);
        __syncthreads();
    }
    if (p.pool) {
        v4f* red = reinterpret_cast<v4f*>(ring);
        red[u * 8 + lc] = psum;
        __syncthreads();
        if (tid < 8) {
            v4f t4 = red[tid];
            for (int sl = 1; sl < 32; ++sl) t4 += red[sl * 8 + tid];
            *reinterpret_cast<v4f*>(p.pool + ((size_t)b * tiles + tile) * 32 + tid * 4) = t4;
        }
    }
}

struct StemRowsGeom {
    int SWo, SWi, strips, band_rows, bands;
};
static bool stem_rows_geom(int H, int W, int mid, int K, int stride, StemRowsGeom& g) {
    if (mid != 32 || K != 3 || stride != 1 || H < 1 || W < 1) return false;
    g.strips = cdiv(W, 28);
    g.SWo = cdiv(cdiv(W, g.strips), 2) * 2;
    g.SWi = g.SWo + 2;
    if (2 * g.SWi > 64) return false;
    g.band_rows = 28;  // measured: 28-row bands beat 14 (fewer pipeline fills) and 56 (too few blocks) on the whole task
    g.band_rows = cdiv(g.band_rows, 2) * 2;
    g.bands = cdiv(H, g.band_rows);
    return true;
}
bool stem_rows_supported(int H, int W, int mid, int K, int stride) {
    StemRowsGeom g;
    return stem_rows_geom(H, W, mid, K, stride, g);
}
int stem_rows_tiles(int H, int W) {
    StemRowsGeom g;
    return stem_rows_geom(H, W, 32, 3, 1, g) ? g.strips * g.bands : 0;
}

int launch_stem_rows(const float* frames, const float* w1_packed, const float* sc1, const float* sh1, const float* wdw,
                     const float* sc2, const float* sh2, float* y, float* pool, int B, int FH, int FW, int spad_t,
                     int spad_l, int H, int W, hipStream_t s, int plan_tiles) {
    ORBIT_REQUIRE(frames && w1_packed && sc1 && sh1 && wdw && sc2 && sh2 && y, "stem_rows: null pointer");
    StemRowsGeom g;
    ORBIT_REQUIRE(stem_rows_geom(H, W, 32, 3, 1, g), "stem_rows: unsupported shape (H=%d W=%d)", H, W);
    ORBIT_REQUIRE(plan_tiles <= 0 || plan_tiles == g.strips * g.bands,
                  "stem_rows: the plan was built for %d pooling tiles per frame, this launch writes %d", plan_tiles, g.strips * g.bands);
    StemRowsParams p;
    p.frames = frames, p.w1 = w1_packed, p.sc1 = sc1, p.sh1 = sh1, p.wdw = wdw, p.sc2 = sc2, p.sh2 = sh2, p.y = y, p.pool = pool;
    p.FH = FH, p.FW = FW, p.spad_t = spad_t, p.spad_l = spad_l, p.H = H, p.W = W;
    p.SWo = g.SWo, p.SWi = g.SWi, p.strips = g.strips, p.band_rows = g.band_rows, p.bands = g.bands;
    p.total = g.strips * g.bands * B;
    const int grid = cdiv(p.total, 8) * 8;
    const size_t lds = ((size_t)3 * 2 * g.SWi * ROWS_ES + 2 * 3 * 5 * STEM_PW + 14 * 64) * sizeof(float);
    const double pix = (double)B * H * W;
    const int rec = prof_start("stem_rows", 2.0 * pix * 32 * 27 + 2.0 * pix * 32 * 9,
                               4.0 * ((double)B * 3 * FH * FW + pix * 32), s, 2.0 * pix * 32);
    // branch-free stores: strips tile the width exactly and every band is a whole number of 2-row steps
    if (g.strips * g.SWo == W && H % 2 == 0 && g.band_rows % 2 == 0) stem_rows_kernel<true><<<grid, 256, lds, s>>>(p, w1_packed, sc1, sh1);
    else stem_rows_kernel<false><<<grid, 256, lds, s>>>(p, w1_packed, sc1, sh1);
    prof_stop(rec, s);
    ORBIT_LAUNCH_CHECK();
    return ORBIT_OK;
}

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

}  // namespace orbit

using namespace orbit;

// ---- single-operator entries for the parity tests ----------------------------------------------------------------------
// pooling partials per frame of orbit_op_stem_dw_front (0: shape not served by the fused kernel)
extern "C" int orbit_op_stem_dw_front_partials(int H, int W, int mid) {
    return stem_rows_supported(H, W, mid, 3, 1) ? stem_rows_tiles(H, W) : 0;
}

// frames NCHW, w_stem torch [mid][3][3][3], wdw torch [mid][1][3][3]
extern "C" int orbit_op_stem_dw_front(const float* frames, const float* w_stem, const float* scale1, const float* shift1,
                                      const float* wdw, const float* scale2, const float* shift2, float* y,
                                      float* pool_partial, int B, int FH, int FW, int spad_top, int spad_left, int H, int W,
                                      int mid, int pad_top, int pad_left, int Ho, int Wo, orbit_stream_t stream) {
    ORBIT_REQUIRE(frames && w_stem && wdw && y, "op_stem_dw_front: null pointer");
    ORBIT_REQUIRE(stem_rows_supported(H, W, mid, 3, 1) && pad_top == 1 && pad_left == 1 && Ho == H && Wo == W,
                  "op_stem_dw_front: unsupported shape (H=%d W=%d mid=%d)", H, W, mid);
    hipStream_t s = (hipStream_t)stream;
    float* wp = nullptr;
    ORBIT_HIP_CHECK(hipMallocAsync(reinterpret_cast<void**>(&wp), (size_t)mid * (32 + 9) * sizeof(float), s));
    int rc = stem_pack_weights(w_stem, wp, mid, s);
    if (rc == ORBIT_OK) rc = dwconv_pack_weights(wdw, wp + (size_t)mid * 32, mid, 3, s);
    if (rc == ORBIT_OK)
        rc = launch_stem_rows(frames, wp, scale1, shift1, wp + (size_t)mid * 32, scale2, shift2, y, pool_partial, B, FH, FW,
                              spad_top, spad_left, H, W, s);
    (void)hipFreeAsync(wp, s);
    return rc;
}

// pooling partials per frame of orbit_op_mbconv_front (0: shape not served by the fused kernel)
extern "C" int orbit_op_mbconv_front_partials(int H, int W, int Cin, int mid, int K, int stride) {
    return mbconv_rows_supported(H, W, Cin, mid, K, stride) ? mbconv_rows_tiles(H, W, Cin, mid, K, stride) : 0;
}

// w1 torch [mid][Cin][1][1], wdw torch [mid][1][K][K]
extern "C" int orbit_op_mbconv_front(const float* x, const float* w1, const float* scale1, const float* shift1,
                                     const float* wdw, const float* scale2, const float* shift2, float* y,
                                     float* pool_partial, int B, int H, int W, int Cin, int mid, int K, int stride,
                                     int pad_top, int pad_left, int Ho, int Wo, orbit_stream_t stream) {
    ORBIT_REQUIRE(x && w1 && wdw && y, "op_mbconv_front: null pointer");
    ORBIT_REQUIRE(mbconv_rows_supported(H, W, Cin, mid, K, stride),
                  "op_mbconv_front: unsupported shape (H=%d W=%d Cin=%d mid=%d K=%d s=%d)", H, W, Cin, mid, K, stride);
    hipStream_t s = (hipStream_t)stream;
    float* wp = nullptr;
    ORBIT_HIP_CHECK(hipMallocAsync(reinterpret_cast<void**>(&wp), (size_t)mid * K * K * sizeof(float), s));
    int rc = dwconv_pack_weights(wdw, wp, mid, K, s);
    if (rc == ORBIT_OK)
        rc = launch_mbconv_rows(x, w1, scale1, shift1, wp, scale2, shift2, y, pool_partial, B, H, W, Cin, mid, K, stride,
                                pad_top, pad_left, Ho, Wo, s);
    (void)hipFreeAsync(wp, s);
    return rc;
}
