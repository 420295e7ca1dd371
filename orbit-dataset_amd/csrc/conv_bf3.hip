// Dense convolutions with fp32 operands split three ways into bf16 and multiplied on the bf16 matrix cores
// (`conv_bf3` option bit 1, OPT-IN: the default path multiplies fp32 operands on v_mfma_f32_32x32x2_f32). Stride-1 pointwise
// convs (with the squeeze-excite gate prologue) and the general form (KxK taps, stride, zero / TF-SAME padding).
//
//   x = x0 + x1 + x2,  x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1)      (round to nearest even; the two
//   subtractions are exact in fp32, the three pieces carry 24 significand bits; csrc/bf3.h)
//   x * w  ~=  x0 w0 + (x0 w1 + x1 w0) + (x1 w1 + x0 w2 + x2 w0)                      six of the nine products; the three
//   dropped ones are <= 2^-24 |x w| each (|x1| <= 2^-8 |x|, |x2| <= 2^-16 |x|; measured max 2^-24.4, median 2^-29 of a
//   product: tests/test_bf3_numerics.py). Every bf16 x bf16 product is exact in fp32.
//
// A v_mfma_f32_32x32x16_bf16 does 8x the multiply-adds of a v_mfma_f32_32x32x2_f32 in half its cycles: six of them per
// 16 k replace eight fp32 ones per 16 k at 3/8 of the matrix-pipe time. The split itself costs ~5.5 VALU instructions per
// staged element (operands are split when a tile is written to LDS: once per block and K-tile, not once per MFMA).
// Two properties of the hardware shape the K loop (both measured; see compute_tile): the bf16 MFMA
// does not round its accumulator, so the x0 w0 sums of a K-tile start from zero and join the running sums through fp32 VALU
// adds; and operand registers of issued bf16 MFMAs must not be rewritten until the tile's last MFMA result has been read.
//
// Structure: csrc/conv_igemm.hip's unpredicated form (same tiling, same global -> register -> LDS pipeline one K-tile ahead,
// same epilogue), with LDS tiles that hold three bf16 planes per operand. Rows are 2 BK bytes, 16-byte chunks XOR-swizzled by
// row so that the ds_read_b128 fragment reads of 16-lane groups are conflict-free without padding.
#include <hip/hip_runtime.h>

#include <cstdio>

#include "bf3.h"
#include "common.h"

namespace orbit {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct Bf3Params {
    const float* x;
    const float* w;       // conv_pack_weights layout [cout_pad][KT], KT = KH * KW * Cin (Cin % 16 == 0: no padding inside a tap)
    float* y;
    const float* scale;
    const float* shift;
    const float* residual;
    const float* gate;
    int M, Cin, Cout, act;
    int m_tiles, n_tiles;
    FastDiv fd_per;  // / (Ho*Wo): frame of a row (gate; row decode of the general form)
    // general (non-pointwise) form: NHWC input, taps walked as in conv_igemm.hip (k = (kh*KW + kw)*Cin + ci, Cin % BK == 0)
    int H, W, KH, KW, stride, pad_t, pad_l, Ho, Wo, KT;
    FastDiv fd_wo;
};

__device__ __forceinline__ float bf3_act(float v, int act) {
    if (act == ORBIT_ACT_RELU) return fmaxf(v, 0.f);
    if (act == ORBIT_ACT_SILU) return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
    return v;
}

__device__ __forceinline__ int bf3_xcd_remap(int bid, int nblk) {  // as conv_igemm.hip: contiguous tile runs per XCD
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, slot = bid >> 3;
    const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + slot;
}

template <int BM, int BN, int WGM, int WGN, int BK, bool GATE, int PF, bool ODD, bool PW>
__global__ __launch_bounds__(256) void conv_bf3_kernel(const Bf3Params p) {
    static_assert(WGM * WGN == 4, "4 waves per block");
    static_assert(BK == 16 || BK == 32, "BK");
    static_assert(PW || !GATE, "the squeeze-excite gate prologue belongs to the pointwise form");
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int TM = WM / 32, TN = WN / 32;
    static_assert(TM >= 1 && TN >= 1, "wave tile");
    constexpr int TPR = BK / 4;              // threads (float4 columns) per tile row
    constexpr int RPP = 256 / TPR;           // rows per pass
    constexpr int AR = BM / RPP, BR = (BN + RPP - 1) / RPP;
    static_assert(AR >= 1 && BM % RPP == 0 && (BN % RPP == 0 || BN < RPP), "tile rows per pass");
    constexpr int ROWB = BK * 2;             // bytes per plane row
    constexpr int APL = BM * ROWB, BPL = BN * ROWB;  // bytes per plane
    constexpr int BUF = 3 * (APL + BPL);     // bytes per pipeline buffer
    constexpr int KS = BK / 16;              // MFMA k-steps per K-tile

    extern __shared__ __attribute__((aligned(16))) char smem_c[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = (wave / WGN) * WM, wn = (wave % WGN) * WN;
    const int ntiles = p.m_tiles * p.n_tiles;
    const int tile = bf3_xcd_remap(blockIdx.x, ntiles);
    const int m0 = (tile / p.n_tiles) * BM, n0 = (tile % p.n_tiles) * BN;

    auto sw = [](int r) { return BK == 32 ? ((r >> 2) & 3) : ((r >> 3) & 1); };

    const int c4 = tid % TPR, lrow = tid / TPR;
    const float* a_ptr[AR];
    const float* g_ptr[GATE ? AR : 1];
    int a_hi0[PW ? 1 : AR], a_wi0[PW ? 1 : AR];  // general form: top-left input coordinate of the row's receptive field
#pragma unroll
    for (int i = 0; i < AR; ++i) {
        if (PW) {
            const int m = min(m0 + lrow + RPP * i, p.M - 1);  // rows beyond M are clamped: computed, never stored
            a_ptr[i] = p.x + (size_t)m * p.Cin + c4 * 4;
            if (GATE) g_ptr[i] = p.gate + (size_t)fdiv((unsigned)m, p.fd_per) * p.Cin + c4 * 4;
        } else {
            const int m = m0 + lrow + RPP * i;
            const bool ok = m < p.M;
            const int mm = ok ? m : 0;
            const int b = (int)fdiv((unsigned)mm, p.fd_per), r = mm - b * (p.Ho * p.Wo);
            const int ho = (int)fdiv((unsigned)r, p.fd_wo), wo = r - ho * p.Wo;
            const int hi0 = ho * p.stride - p.pad_t, wi0 = wo * p.stride - p.pad_l;
            // (the pointer may lie outside the tensor for padded rows: it is only used where the bounds test passes)
            a_ptr[i] = p.x + (((long)b * p.H + hi0) * p.W + wi0) * p.Cin + c4 * 4;
            a_hi0[i] = ok ? hi0 : -(1 << 28);  // rows beyond M fail every bounds test
            a_wi0[i] = wi0;
        }
    }
    const bool b_row_ok = BN % RPP == 0 || lrow < BN;  // 128x32 tiles at BK = 16: half of the threads stage B
    const float* b_ptr = p.w + (size_t)(n0 + (b_row_ok ? lrow : 0)) * p.KT + c4 * 4;
    // byte offset of this thread's 8-byte slot inside a plane row block (same for A and B: both tiles are staged by row)
    int st_off[AR > BR ? AR : BR];
#pragma unroll
    for (int i = 0; i < (AR > BR ? AR : BR); ++i) {
        const int r = lrow + RPP * i;
        st_off[i] = r * ROWB + (((c4 >> 1) ^ sw(r)) << 4) + ((c4 & 1) << 3);
    }

    // PF = staged K-tiles in flight: 1 (default) or 2 (`conv_bf3_pf` = 2; measured: no gain, 16-36 more registers)
    struct Stage {
        f32x4 a[AR], b[BR], g[GATE ? AR : 1];
        unsigned mask;  // general form: bit i = row i's element of this tap lies inside the image
    };
    int ld_k = 0;
    int ld_ci = 0, ld_kh = 0, ld_kw = 0, ld_tap = 0;  // wave-uniform tap walk of the general form (conv_igemm.hip)
    auto load_tile = [&](Stage& st) {
        if (PW) {
#pragma unroll
            for (int i = 0; i < AR; ++i) {
                st.a[i] = *reinterpret_cast<const f32x4*>(a_ptr[i] + ld_k);
                if (GATE) st.g[i] = *reinterpret_cast<const f32x4*>(g_ptr[i] + ld_k);
            }
        } else {
            unsigned mask = 0;
            const long koff = (long)ld_tap + ld_ci;
#pragma unroll
            for (int i = 0; i < AR; ++i) {
                const bool ok = (unsigned)(a_hi0[i] + ld_kh) < (unsigned)p.H && (unsigned)(a_wi0[i] + ld_kw) < (unsigned)p.W;
                st.a[i] = *reinterpret_cast<const f32x4*>(ok ? a_ptr[i] + koff : p.x);  // unconditional load, zeroed at the store
                mask |= (ok ? 1u : 0u) << i;
            }
            st.mask = mask;
            ld_ci += BK;
            if (ld_ci >= p.Cin) {
                ld_ci = 0, ld_tap += p.Cin;
                if (++ld_kw == p.KW) ld_kw = 0, ++ld_kh, ld_tap += (p.W - p.KW) * p.Cin;
            }
        }
#pragma unroll
        for (int j = 0; j < BR; ++j) st.b[j] = *reinterpret_cast<const f32x4*>(b_ptr + (size_t)(RPP * j) * p.KT + ld_k);
        ld_k += BK;
    };
    auto store_tile = [&](const Stage& st, int buf) {
        char* A = smem_c + buf * BUF;
        char* Bq = A + 3 * APL;
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            uint2 q0, q1, q2;
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            split3(GATE ? st.a[i] * st.g[i] : (PW || ((st.mask >> i) & 1u)) ? st.a[i] : zero, q0, q1, q2);
            *reinterpret_cast<uint2*>(A + st_off[i]) = q0;
            *reinterpret_cast<uint2*>(A + APL + st_off[i]) = q1;
            *reinterpret_cast<uint2*>(A + 2 * APL + st_off[i]) = q2;
        }
#pragma unroll
        for (int j = 0; j < BR; ++j) {
            if (!b_row_ok) break;
            uint2 q0, q1, q2;
            split3(st.b[j], q0, q1, q2);
            *reinterpret_cast<uint2*>(Bq + st_off[j]) = q0;
            *reinterpret_cast<uint2*>(Bq + BPL + st_off[j]) = q1;
            *reinterpret_cast<uint2*>(Bq + 2 * BPL + st_off[j]) = q2;
        }
    };

    // two accumulator tiles: the x0 w0 sums (kept by VALU adds, see compute_tile) and the five small products, which are
    // 2^-8 .. 2^-16 of the former and accumulate on the matrix cores. The two are added once, in the epilogue.
    f32x16 acc[TM][TN], c1[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f, c1[i][j][r] = 0.f;

    // fragment read offsets: lane (row l31, k-half lh) reads the 16-byte chunk 2 ks + lh of its row
    int a_off[TM][KS], b_off[TN][KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int r = wm + i * 32 + l31;
            a_off[i][ks] = r * ROWB + (((2 * ks + lh) ^ sw(r)) << 4);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int r = wn + j * 32 + l31;
            b_off[j][ks] = r * ROWB + (((2 * ks + lh) ^ sw(r)) << 4);
        }
    }

    auto compute_tile = [&](int cur) {
        const char* A = smem_c + cur * BUF;
        const char* Bq = A + 3 * APL;
        // EVERY fragment of the K-tile is read into registers before its first MFMA, and no fragment register is written again
        // before the tile's last MFMA result has been read. The form that fetched k-step 1's fragments between k-step 0's MFMAs
        // (into the registers k-step 0's operands had used, as csrc/conv_igemm.hip does for the fp32 MFMAs) was correct alone
        // and WRONG with another stream's kernels on the chip: the task's logits moved by up to 0.26 from run to run
        // (tools/bf3_race_probe3.py: two independent models on two streams; hipcc's own schedule of the same loop showed it too,
        // K-tile 16 - one k-step, nothing to overwrite - did not). An issued v_mfma_f32_32x32x16_bf16 that waits behind other
        // waves' matrix instructions evidently has not read its A / B registers yet when an LDS return overwrites them.
        bf16x8 af[KS][3][TM], bf[KS][3][TN];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int q = 0; q < 3; ++q) {
#pragma unroll
                for (int i = 0; i < TM; ++i) af[ks][q][i] = *reinterpret_cast<const bf16x8*>(A + q * APL + a_off[i][ks]);
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[ks][q][j] = *reinterpret_cast<const bf16x8*>(Bq + q * BPL + b_off[j][ks]);
            }
        // The x0 w0 products of this K-tile are summed in a FRESH register tile and added to the running sums with fp32 VALU
        // adds (round to nearest even). Measured (tools/mfma_round_probe.hip): the bf16 MFMA aligns its 16 products and C to
        // the largest exponent among them and drops what falls more than ~3 bits below that term's ulp, without rounding - fed
        // a large running sum as C it loses up to ~1 ulp of the SUM per instruction, always toward zero: over the 36-72
        // instructions of a layer that is a systematic 1e-6 relative error, 4e-5 on the task's logits (30x the fp32 path).
        // With C = 0 the loss is relative to the largest PRODUCT of a 16-group instead.
        f32x16 t[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) t[i][j][r] = 0.f;
        __builtin_amdgcn_sched_barrier(0);  // (the reads above stay above: see the note on this lambda)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    // smallest products first; the x0 w0 product LAST: its result is read by the adds below, and the matrix
                    // pipe runs in order, so by then every MFMA of the tile has consumed its operands
                    c1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][2][i], bf[ks][0][j], c1[i][j], 0, 0, 0);
                    c1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][0][i], bf[ks][2][j], c1[i][j], 0, 0, 0);
                    c1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][1][i], bf[ks][1][j], c1[i][j], 0, 0, 0);
                    c1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][1][i], bf[ks][0][j], c1[i][j], 0, 0, 0);
                    c1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][0][i], bf[ks][1][j], c1[i][j], 0, 0, 0);
                    t[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][0][i], bf[ks][0][j], t[i][j], 0, 0, 0);
                }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] += t[i][j];
        __builtin_amdgcn_sched_barrier(0);  // (nothing of the next stage moves above the adds)
    };

    const int nk = p.KT / BK;
    if constexpr (PF == 1) {
        Stage s0;
        load_tile(s0);
        store_tile(s0, 0);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const int cur = kt & 1;
            if (kt + 1 < nk) load_tile(s0);
            compute_tile(cur);
            if (kt + 1 < nk) store_tile(s0, cur ^ 1);
            __syncthreads();
        }
    } else {
        // tile t travels through stage set t & 1 into LDS buffer t & 1; its loads were issued two iterations before its store.
        // ODD (the parity of the K-tile count) is a template parameter: a run-time exit between the two halves of the loop
        // body makes hipcc keep two accumulator sets and copy between them (288 v_accvgpr moves, 220 registers)
        Stage s0, s1;
        load_tile(s0);
        if (nk > 1) load_tile(s1);
        store_tile(s0, 0);
        __syncthreads();
        for (int kt = 0; kt + 1 < nk; kt += 2) {
            if (kt + 2 < nk) load_tile(s0);
            compute_tile(0);
            store_tile(s1, 1);
            __syncthreads();
            if (kt + 3 < nk) load_tile(s1);
            compute_tile(1);
            if (kt + 2 < nk) store_tile(s0, 0);
            __syncthreads();
        }
        if constexpr (ODD) {
            compute_tile(0);
            __syncthreads();
        }
    }

    // ---- epilogue (conv_igemm.hip's batched form): C tile through LDS, float4 rows out --------------------------------
    float* Cs = reinterpret_cast<float*>(smem_c);  // [BM][BN]
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn + j * 32 + l31;
        const bool n_ok = n < p.Cout;
        const float sc = (n_ok && p.scale) ? p.scale[n] : 1.0f;
        const float sh = (n_ok && p.shift) ? p.shift[n] : 0.0f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int rowl = wm + i * 32 + 8 * rq + 4 * lh;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    Cs[(rowl + r) * BN + wn + j * 32 + l31] = (acc[i][j][rq * 4 + r] + c1[i][j][rq * 4 + r]) * sc + sh;
            }
    }
    __syncthreads();
    constexpr int TPO = BN / 4, RPO = 256 / TPO, ITER = BM / RPO;
    const int oc = (tid % TPO) * 4, n = n0 + oc;
    if (n < p.Cout) {
        f32x4 v[ITER], res[ITER];
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int r = tid / TPO + it * RPO;
            const bool ok = m0 + r < p.M;
            v[it] = *reinterpret_cast<const f32x4*>(Cs + r * BN + oc);
            res[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (p.residual) res[it] = *reinterpret_cast<const f32x4*>(p.residual + (size_t)(ok ? m0 + r : 0) * p.Cout + n);
        }
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int r = tid / TPO + it * RPO;
            if (m0 + r < p.M) {
                f32x4 o = v[it] + res[it];
                o[0] = bf3_act(o[0], p.act), o[1] = bf3_act(o[1], p.act), o[2] = bf3_act(o[2], p.act), o[3] = bf3_act(o[3], p.act);
                *reinterpret_cast<f32x4*>(p.y + (size_t)(m0 + r) * p.Cout + n) = o;
            }
        }
    }
}

bool conv_bf3_supported(const ConvDesc& d) {
    if (d.x_nchw || d.pool2 || d.y_raw || d.stats) return false;
    if (d.Cin % 16 != 0 || d.Cout % 4 != 0 || d.Cin < 64 || d.Cout < 40) return false;
    const bool pw = d.KH == 1 && d.KW == 1 && d.pad_t == 0 && d.pad_l == 0 && d.stride == 1;
    return pw || d.gate == nullptr;  // the general form (taps, stride, padding) has no squeeze-excite prologue
}

template <int BM, int BN, int WGM, int WGN, int BK, bool PW>
static int bf3_launch(Bf3Params& p, const ConvDesc& d, hipStream_t s) {
    p.m_tiles = cdiv(p.M, BM), p.n_tiles = cdiv(p.Cout, BN);
    const size_t pipe = (size_t)2 * 3 * (BM + BN) * BK * 2, epi = (size_t)BM * BN * 4;
    const size_t lds = pipe > epi ? pipe : epi;
    const int grid = p.m_tiles * p.n_tiles;
    char name[48];
    snprintf(name, sizeof(name), "conv_bf3<%d,%d,%d%s%s>", BM, BN, BK, d.gate ? ",gate" : "", PW ? ",pw" : "");
    const double pix = (double)p.M;
    const int rec = prof_start(name, 2.0 * pix * d.Cout * d.KH * d.KW * d.Cin * d.prof_flop_scale,
                               4.0 * ((double)d.B * d.H * d.W * d.Cin + pix * d.Cout * (d.residual ? 2.0 : 1.0) +
                                      (double)d.Cout * d.KH * d.KW * d.Cin), s);
    if constexpr (PW) {
        if (d.gate) conv_bf3_kernel<BM, BN, WGM, WGN, BK, true, 1, false, true><<<grid, 256, lds, s>>>(p);
        else conv_bf3_kernel<BM, BN, WGM, WGN, BK, false, 1, false, true><<<grid, 256, lds, s>>>(p);
    } else {
        conv_bf3_kernel<BM, BN, WGM, WGN, BK, false, 1, false, false><<<grid, 256, lds, s>>>(p);
    }
    prof_stop(rec, s);
    ORBIT_LAUNCH_CHECK();
    return ORBIT_OK;
}

template <bool PW>
static int bf3_dispatch(Bf3Params& p, const ConvDesc& d, hipStream_t s) {
    // tiles as conv_igemm.hip chooses them: 128x32 where a 64-wide last column tile would be mostly padding
    const double waste64 = (double)(cdiv(d.Cout, 64) * 64 - d.Cout) / d.Cout;
    const double waste32 = (double)(cdiv(d.Cout, 32) * 32 - d.Cout) / d.Cout;
    const bool narrow = waste64 - waste32 >= 0.15;
    const int bk = d.Cin % 32 == 0 ? 32 : 16;
    if (narrow) return bk == 32 ? bf3_launch<128, 32, 4, 1, 32, PW>(p, d, s) : bf3_launch<128, 32, 4, 1, 16, PW>(p, d, s);
    return bk == 32 ? bf3_launch<64, 64, 2, 2, 32, PW>(p, d, s) : bf3_launch<64, 64, 2, 2, 16, PW>(p, d, s);
}

int launch_conv_bf3(const ConvDesc& d, hipStream_t s) {
    ORBIT_REQUIRE(conv_bf3_supported(d), "conv_bf3: unsupported convolution");
    Bf3Params p;
    p.x = d.x, p.w = d.w_packed, p.y = d.y, p.scale = d.scale, p.shift = d.shift, p.residual = d.residual, p.gate = d.gate;
    p.M = d.B * d.Ho * d.Wo, p.Cin = d.Cin, p.Cout = d.Cout, p.act = d.act;
    p.H = d.H, p.W = d.W, p.KH = d.KH, p.KW = d.KW, p.stride = d.stride, p.pad_t = d.pad_t, p.pad_l = d.pad_l;
    p.Ho = d.Ho, p.Wo = d.Wo, p.KT = d.KH * d.KW * d.Cin;  // (Cin % 16 == 0: conv_pack_weights pads nothing)
    p.fd_per = make_fastdiv((unsigned)(d.Ho * d.Wo)), p.fd_wo = make_fastdiv((unsigned)d.Wo);
    ORBIT_REQUIRE((long long)d.B * d.H * d.W * d.Cin < (1ll << 40) && p.M > 0, "conv_bf3: tensor too large");
    const bool pw = d.KH == 1 && d.KW == 1 && d.pad_t == 0 && d.pad_l == 0 && d.stride == 1;
    return pw ? bf3_dispatch<true>(p, d, s) : bf3_dispatch<false>(p, d, s);
}

}  // namespace orbit
