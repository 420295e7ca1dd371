// Shared helpers for liborbit_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include "../../include/orbit_hip.h"

namespace orbit {

// thread-local error message returned by orbit_last_error()
char* err_buf();
int set_err(int code, const char* fmt, ...);
// runtime options (orbit_set_option / ORBIT_* environment; the table is in csrc/head.hip)
int get_option(const char* name);
int option_epoch();  // changes whenever orbit_set_option changed a value (key of captured launch sequences)

#define ORBIT_HIP_CHECK(expr)                                                                  \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess)                                                                  \
            return ::orbit::set_err(ORBIT_ERR_HIP, "%s failed: %s (%s:%d)", #expr,             \
                                    hipGetErrorString(_e), __FILE__, __LINE__);                \
    } while (0)

#define ORBIT_LAUNCH_CHECK()                                                                   \
    do {                                                                                       \
        hipError_t _e = hipGetLastError();                                                     \
        if (_e != hipSuccess)                                                                  \
            return ::orbit::set_err(ORBIT_ERR_HIP, "kernel launch failed: %s (%s:%d)",         \
                                    hipGetErrorString(_e), __FILE__, __LINE__);                \
    } while (0)

#define ORBIT_REQUIRE(cond, ...)                                                               \
    do {                                                                                       \
        if (!(cond)) return ::orbit::set_err(ORBIT_ERR_ARG, __VA_ARGS__);                      \
    } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---- internal launchers shared between the single-op C-ABI and the network runtime -------------

// Division by a launch-time constant: q = (umulhi(n, mul) + n) >> shift, exact for n < 2^31 (Granlund-Montgomery
// round-up magic). A 32-bit integer division is ~25 VALU / ~40 SALU instructions on gfx950; the row decode below and the
// stem's k decode are made of them, and on the short-K EfficientNet layers that index arithmetic (not the MFMAs, not
// HBM) was the busiest pipe (rocprofv3 SQ counters, tools/conv_pmc.sh).
struct FastDiv {
    unsigned mul, shift;
};
static inline FastDiv make_fastdiv(unsigned d) {
    FastDiv f;
    unsigned l = 0;
    while ((1u << l) < d) ++l;  // ceil(log2 d)
    f.mul = (unsigned)(((((unsigned long long)1 << l) - d) << 32) / d + 1);
    f.shift = l;
    return f;
}
__device__ __forceinline__ unsigned fdiv(unsigned n, FastDiv f) { return (__umulhi(n, f.mul) + n) >> f.shift; }

struct ConvDesc {
    const float* x;         // NHWC activations, or NCHW frames when x_nchw
    const float* w_packed;  // [CoutPad][KT], K = (kh, kw, ci) with ci fastest, zero padded
    const float* w_frag = nullptr;  // the same filter in MFMA-fragment order (conv_frag_floats / conv_frag_pack_weights) or
                                    // nullptr: pointwise convs then stay on the LDS-tiled kernel (csrc/pw_rgemm.hip)
    float* y;               // NHWC
    const float* scale;     // [Cout] or nullptr
    const float* shift;     // [Cout] or nullptr
    const float* residual;  // NHWC like y or nullptr
    const float* gate;      // [B][Cin] or nullptr
    int B, H, W, Cin, Cout, KH, KW, stride, pad_t, pad_l, Ho, Wo;
    int act;     // ORBIT_ACT_*
    int pool2;   // fused 2x2/2 max-pool of the activated output
    int x_nchw;  // stem gather from NCHW frames
    float prof_flop_scale = 1.f;  // algorithmic / executed flops (strided dgrad runs on the zero-inserted grid)
    float* splitk_ws = nullptr;   // scratch for split-K partial tiles (conv_splitk_floats(d) floats) or nullptr: no split
    // train-mode BatchNorm statistics from the epilogue (the LITE step, csrc/extractor_train.hip): every block of output
    // rows writes the column sums of its RAW outputs (before scale / shift / residual / activation) and of their squares to
    // stats[row_block][2][Cout] - the layout bn_stats_finalize reads - so the separate statistics pass over y disappears.
    // `stats` needs conv_stats_floats(M, Cout) floats; *stats_blocks receives the number of row blocks written, or 0 when
    // this launch could not emit them (fused pooling, split-K, the narrow-pointwise kernel): the caller then runs
    // launch_bn_stats' own pass.
    float* stats = nullptr;
    int* stats_blocks = nullptr;
    // statistics sweep (round 6, the no-grad cache pass of the LITE step on the row-streaming fronts): y == nullptr with
    // `stats` set - the conv runs for its BatchNorm statistics alone and stores nothing (the 6x-expanded tensor of an MBConv
    // block never reaches HBM: the fused front re-expands it in LDS with the scale / shift these statistics give). Same
    // kernel, same tiles, same summation order as the storing form: the statistics are bit-identical to the unfused pass's.
    bool stats_only = false;
    // dual write (training tape under running-statistics BatchNorm, where scale / shift are known before the conv runs):
    // the raw conv output goes to y_raw and y receives act(raw * scale + shift + residual) - the activation pass over the
    // tensor (read y, write a) disappears. Not with fused pooling, split-K or the narrow-pointwise kernel.
    float* y_raw = nullptr;
};
inline size_t conv_stats_floats(size_t M, int Cout) { return ((M + 31) / 32) * 2 * (size_t)Cout; }  // smallest tile: 32 rows
// split-K (small output, long reduction): number of K splits launch_conv uses for this conv when scratch is provided
// (1 = none) and the scratch it needs
int conv_splitk(const ConvDesc& d);
size_t conv_splitk_floats(const ConvDesc& d);

// geometry of the packed weight matrix for a conv (shared by pack + launch)
struct ConvPackGeom {
    int cin_pad;   // per-tap padded Cin (vector mode), unused in stem mode
    int kt;        // padded K
    int cout_pad;  // padded rows
};
ConvPackGeom conv_pack_geom(int Cin, int Cout, int KH, int KW, int x_nchw);
size_t conv_packed_floats(int Cin, int Cout, int KH, int KW, int x_nchw);
// OIHW -> packed (device to device)
int conv_pack_weights(const float* w_oihw, float* w_packed, int Cin, int Cout, int KH, int KW,
                      int x_nchw, hipStream_t s);
int launch_conv(const ConvDesc& d, hipStream_t s);
// pointwise convs as a register GEMM on fragment-packed weights (csrc/pw_rgemm.hip; option conv_rgemm); launch_conv routes to it
size_t conv_frag_floats(int Cin, int Cout, int KH, int KW, int x_nchw);  // 0: this filter has no fragment-packed form
int conv_frag_pack_weights(const float* w_oihw, float* w_frag, int Cin, int Cout, hipStream_t s);
bool pw_rgemm_supported(const ConvDesc& d);
// pointwise convs on the bf16 matrix cores with both operands split three ways (csrc/conv_bf3.hip; option conv_bf3, opt-in)
bool conv_bf3_supported(const ConvDesc& d);
int launch_conv_bf3(const ConvDesc& d, hipStream_t s);
bool pw_rgemm_preferred(const ConvDesc& d);  // where it measured faster than the LDS-tiled kernel (conv_rgemm = 1)
int launch_pw_rgemm(const ConvDesc& d, hipStream_t s);
bool conv_prof_enabled();
// per-launch HIP-event records of orbit_prof_* (no-ops returning -1 while profiling is off)
int prof_start(const char* name, double flops, double bytes, hipStream_t s, double silu = 0.0);  // silu: SiLU evaluations of the launch
void prof_stop(int idx, hipStream_t s);  // per-launch event profiling is on (graphs are bypassed while it is)

// The depthwise DATA gradient fused with the first pass of the preceding BatchNorm's backward (the LITE step): the tensor a
// depthwise dgrad produces is d(act(BatchNorm(y))) of the expansion conv before it; with this the kernel reads y at its output
// pixels, writes g = dx * act'(y * scale + shift) instead of dx and emits the per-block channel sums of g and of g * xhat
// (xhat = (y - mean) * invstd) to partial[block][2][C] - the layout bn_bwd_finalize reads - so that BatchNorm's backward needs
// no reduction pass over (dx, y) of its own (launch_bn_backward_reduced finishes it).
struct DwBnBwd {
    const float* y = nullptr;  // raw output of the producing conv, laid out like the dgrad's output
    const float* mean = nullptr;
    const float* invstd = nullptr;
    const float* scale = nullptr;
    const float* shift = nullptr;
    int act = 0;
    float* partial = nullptr;
    int* nblk = nullptr;  // out: partial rows written, 0 when the chosen kernel has no fused form (then dx was written plain)
    // optional: the layer's FILTER gradient in the same pass. The layer's input is act(y * scale + shift) - the tensor the kernel
    // rebuilds at its output pixels anyway - and every (output pixel, tap) pair meets the dy element the data gradient multiplies
    // with that tap, so sum dy * input(tap) accumulates beside it: wgrad_partial[block][K*K][C] (dwconv_bwd_fused_scratch_floats),
    // finished by launch_dwconv_wgrad_reduce. *wgrad_rows = rows written, 0 when the chosen kernel does not carry it.
    float* wgrad_partial = nullptr;
    int* wgrad_rows = nullptr;
};
// depthwise + fused SE pooling partials [B][chunks][C] (pool_partial may be nullptr); chunks = dwconv_se_chunks(Ho)
int dwconv_se_chunks(int Ho);
int launch_dwconv_se(const float* x, const float* w_khwc, float* y, const float* scale, const float* shift,
                     float* pool_partial, int B, int H, int W, int C, int K, int stride, int pad_t, int pad_l, int Ho,
                     int Wo, int act, hipStream_t s, int stats = 0, const float* in_scale = nullptr,
                     const float* in_shift = nullptr, int in_act = 0, const DwBnBwd* bnb = nullptr);
// (bnb: the data-gradient use of these kernels - rotated taps, no scale / shift / activation, no pooling - with the
// BatchNorm-backward epilogue above; *bnb->nblk tells whether the chosen kernel family had it)
// in_scale / in_shift (with stats): x is the RAW output of the producing conv; act(x * in_scale[c] + in_shift[c]) is applied
// as the kernel loads it, so the activated tensor of that layer is never written (no-backward passes of the LITE step)
// stats != 0: pool_partial receives [B * dwconv_se_chunks(Ho)][2][C] column sums / sums of squares of the outputs instead
// (train-mode BatchNorm statistics from the producing kernel; the partial layout bn_stats_finalize reads)
int launch_se_gate2(const float* partial, int chunks, int HW, const float* w1, const float* b1, const float* w2t,
                    const float* b2, float* gate, int B, int C, int R, hipStream_t s, float* pooled_out = nullptr);
// (pooled_out, optional: the pooled means [B][C] the gate was computed from - the training tape keeps them)
int launch_transpose(const float* in, float* out, int rows, int cols, hipStream_t s);
// EfficientNet stem as a direct VALU kernel with LDS-staged input rows (csrc/stem.hip); w = raw OIHW filter [32][3][3][3]
bool stem_direct_supported(int Cin, int Cout, int K, int stride, int W, int act);
int launch_stem_direct(const float* frames, const float* w_oihw, const float* scale, const float* shift, float* y, int B,
                       int H, int W, int pad_t, int pad_l, int Ho, int Wo, hipStream_t s);
// fused expand(1x1, MFMA) + BN + SiLU + depthwise + BN + SiLU (+ SE pooling partials), the
// row-streaming form for the 112x112 .. 28x28 stages (csrc/mbconv_rows.hip): a block walks down a strip of the map with the
// expanded rows in an LDS ring - no tile halo; pool partials [B][mbconv_rows_tiles][mid]
bool mbconv_rows_supported(int H, int W, int Cin, int mid, int K, int stride);
int mbconv_rows_tiles(int H, int W, int Cin, int mid, int K, int stride);
int launch_mbconv_rows(const float* x, const float* w1, const float* sc1, const float* sh1, const float* wdw,
                       const float* sc2, const float* sh2, float* y, float* pool, int B, int H, int W, int Cin, int mid,
                       int K, int stride, int pad_t, int pad_l, int Ho, int Wo, hipStream_t s, int plan_tiles = 0,
                       bool raw_stats = false);
// raw_stats (train-mode BatchNorm, no-grad passes): y receives the RAW depthwise outputs (no second BatchNorm / activation;
// sc2 / sh2 are ignored) and `pool` the [B * mbconv_rows_tiles][2][mid] column sums / sums of squares of those outputs - the
// partial layout launch_bn_stats_from_partials reads
// (plan_tiles > 0: the tile count the caller sized `pool` and its consumer for; the launch fails if it would write another)
// row-streaming stem + first depthwise (csrc/mbconv_rows.hip): w1_packed = stem_pack_weights' [32][32]; pool [B][stem_rows_tiles][32]
bool stem_rows_supported(int H, int W, int mid, int K, int stride);
int stem_rows_tiles(int H, int W);
int launch_stem_rows(const float* frames, const float* w1_packed, const float* sc1, const float* sh1, const float* wdw,
                     const float* sc2, const float* sh2, float* y, float* pool, int B, int FH, int FW, int spad_t,
                     int spad_l, int H, int W, hipStream_t s, int plan_tiles = 0);
int stem_pack_weights(const float* w_oihw, float* w_packed, int mid, hipStream_t s);  // [mid][27] -> [mid][32]
// [C][1][K][K] -> [K][K][C]
int dwconv_pack_weights(const float* w, float* w_khwc, int C, int K, hipStream_t s);
int launch_maxpool(const float* x, float* y, int B, int H, int W, int C, int K, int stride, int pad,
                   int Ho, int Wo, hipStream_t s);
int launch_avgpool(const float* x, float* y, int B, int HW, int C, hipStream_t s);
int launch_se_gate(const float* pooled, const float* w1, const float* b1, const float* w2,
                   const float* b2, float* gate, int B, int C, int R, hipStream_t s);


// ---- training-side launchers (train_ops.hip, conv_wgrad.hip) ------------------------------------------------------
int bn_reduce_blocks(int M, int C);  // rows of the [blocks][2][C] partial buffer the BatchNorm reductions need
// batch statistics of y[M][C] -> mean, invstd, folded scale/shift (gamma/beta nullable = 1/0), running-stat update
int launch_bn_stats(const float* y, int M, int C, float eps, float momentum, const float* gamma, const float* beta,
                    const float* conv_bias, float* mean, float* invstd, float* scale, float* shift, float* running_mean,
                    float* running_var, float* partial, hipStream_t s);
// the same from [nblk][2][C] partials a producing kernel wrote (ConvDesc::stats, launch_dwconv_se(..., stats)); `partial` is
// bn_partial_floats(nblk, C) floats (room for the compaction stage of very long partial lists)
size_t bn_partial_floats(size_t nblk, int C);
int launch_sum_partials(const float* partial, int nblk, int C, float* stats /* [2][C] */, hipStream_t s);
// batch statistics of a POINTWISE conv's output y = W x from the Gram matrix of its input x [P][Cin] (Cin = 16 or 24;
// W [C][Cin] as torch stores a 1x1 filter): no pass over y, no y at all. scratch: bn_gram_scratch_floats(P, Cin) floats
bool bn_gram_supported(int Cin);
size_t bn_gram_scratch_floats(int P, int Cin);
int launch_bn_stats_from_gram(const float* x, int P, int Cin, const float* w, int C, float eps, float momentum,
                              const float* gamma, const float* beta, float* mean, float* invstd, float* scale, float* shift,
                              float* running_mean, float* running_var, float* scratch, hipStream_t s);
int launch_bn_stats_from_partials(float* partial, int nblk, int M, int C, float eps, float momentum, const float* gamma,
                                  const float* beta, const float* conv_bias, float* mean, float* invstd, float* scale,
                                  float* shift, float* running_mean, float* running_var, hipStream_t s);
int launch_scale_shift_act(const float* y, const float* scale, const float* shift, const float* residual, int act,
                           size_t M, int C, float* out, hipStream_t s);
// a = act(y * scale + shift) for [B][HW][C] plus the squeeze-excite pooling partials pool[B][se_pool_chunks][C] of a
int se_pool_chunks(int B, int HW, int C);
int launch_scale_shift_act_pool(const float* y, const float* scale, const float* shift, int act, int B, int HW, int C,
                                float* out, float* pool, hipStream_t s);
// coef: 3*C floats of scratch; dy nullable (reductions only); dres nullable (gradient of the residual input)
// scale/shift: folded BatchNorm of the forward (needed to rebuild the SiLU pre-activation), else nullable
int launch_bn_backward(const float* dout, const float* out, const float* y, const float* mean, const float* invstd,
                       const float* gamma, const float* scale, const float* shift, int train, int act, int M, int C,
                       float* dy, float* dres, int dres_accumulate, float* dgamma, float* dbeta, float* dbias,
                       float* partial, float* coef, hipStream_t s);
int launch_maxpool_idx(const float* x, float* y, uint8_t* idx, int B, int H, int W, int C, int K, int stride, int pad,
                       int Ho, int Wo, hipStream_t s);
int launch_maxpool_bwd(const float* dy, const uint8_t* idx, float* dx, int B, int H, int W, int C, int K, int stride,
                       int pad, int Ho, int Wo, hipStream_t s);
int launch_avgpool_bwd(const float* dy, float* dx, int B, int HW, int C, hipStream_t s);
int launch_upsample_zero(const float* src, float* dst, int B, int H, int W, int C, int stride, int Hs, int Ws,
                         hipStream_t s);
int launch_add_inplace(float* dst, const float* src, size_t n, hipStream_t s);
// MBConv pieces (train_mbconv.hip)
int launch_colmean(const float* x, float* pooled, int B, int HW, int C, hipStream_t s);  // [B][HW][C] -> means [B][C]
int launch_gate_mul(const float* x, const float* gate, float* xg, int B, int HW, int C, hipStream_t s);
size_t se_bwd_scratch_floats(int B, int C, int R);
// dxg: gradient of x*gate; writes dx (through the product, the gate MLP and the average pool) and, when dw1 != NULL, the
// gradients of the four SE tensors (W1 [R][C], b1 [R], W2 [C][R], b2 [C])
// bn (optional): x = act(BatchNorm(y)); then `dx` receives g = d x * act'(.) instead of d x and bn->partial the per-(frame,
// chunk) sums of g and g * xhat ([B * se_pool_chunks(B, HW, C)][2][C]): launch_bn_backward_reduced finishes that BatchNorm's
// backward without a reduction pass of its own
struct SeBnFuse {
    const float *y, *mean, *invstd, *scale, *shift;
    int act;
    float* partial;
};
int launch_se_gate_backward(const float* dxg, const float* x, const float* pooled, const float* gate, const float* w1,
                            const float* b1, const float* w2, const float* b2, float* dx, float* dw1, float* db1,
                            float* dw2, float* db2, float* scratch, int B, int HW, int C, int R, hipStream_t s,
                            const SeBnFuse* bn = nullptr, const float* w2t = nullptr);
// With bn, dx may be nullptr: g is then NOT written (one pass over the expanded tensor less); the BatchNorm's apply pass
// rebuilds it from dxg, the gate and the pooled-branch gradient the MLP backward left in `scratch` (se_bwd_dpooled):
const float* se_bwd_dpooled(const float* scratch, int B, int C, int R);
// The parameter gradients of several squeeze-excite blocks in one launch: call launch_se_gate_backward with dw1 = nullptr
// and a scratch of the block's own, keep se_bwd_param_job(scratch, ...) and run the batch when all blocks are through.
constexpr int SE_PARAM_JOBS = 16;
struct SeParamJob {
    const float *du, *dv, *h, *pooled;
    float *dw1, *db1, *dw2, *db2;
    int B, C, R;
};
struct SeParamJobs {
    SeParamJob j[SE_PARAM_JOBS];
};
SeParamJob se_bwd_param_job(const float* scratch, const float* pooled, int B, int C, int R, float* dw1, float* db1, float* dw2,
                            float* db2);
int launch_se_param_grad_batched(const SeParamJobs& jobs, int n, hipStream_t s);
int launch_bn_backward_reduced_gated(const float* dxg, const float* gate, const float* dpooled, int HW, const float* y,
                                     const float* mean, const float* invstd, const float* scale, const float* shift, int act,
                                     const float* gamma, int train, int M, int C, float* dy, float* dgamma, float* dbeta,
                                     float* partial, int nblk, float* coef, hipStream_t s);
// (w2t, optional: W2 transposed to [R][C] - the plan's packed copy; the MLP backward then reads W2 contiguously)
// BatchNorm backward whose reduction pass already ran (g = dout * act'(.) in `g`, [nblk][2][C] sums of g and g * xhat in
// `partial`): finalize + apply only. coef: 3*C floats
int launch_bn_backward_reduced(const float* g, const float* y, const float* mean, const float* invstd, const float* gamma,
                               int train, int M, int C, float* dy, float* dgamma, float* dbeta, float* partial, int nblk,
                               float* coef, hipStream_t s);
// flip_scratch (K*K*C floats, optional): stride-1 layers then run as a FORWARD depthwise convolution of dy with the flipped
// taps through the LDS-patch kernels of csrc/ops.hip instead of the per-pixel gather
int launch_dwconv_dgrad(const float* dy, const float* w_khwc, float* dx, int B, int H, int W, int C, int K, int stride,
                        int pad_t, int pad_l, int Ho, int Wo, hipStream_t s, float* flip_scratch = nullptr,
                        const DwBnBwd* bnb = nullptr);
int dwconv_dgrad_bn_blocks(int B, int H, int W, int C, int stride);  // upper bound of the partial rows a fused dgrad writes
// scratch of a data gradient that carries the filter gradient (0: no fused form for the layer): the flipped taps of the stride-1
// form in the first dwconv_bwd_fused_partial_offset floats, DwBnBwd::wgrad_partial behind them
size_t dwconv_bwd_fused_scratch_floats(int B, int H, int W, int C, int K, int stride);
size_t dwconv_bwd_fused_partial_offset(int C, int K);
bool dwconv_se_window_form(int K, int stride, int Ho);
int launch_dwconv_wgrad_reduce(const float* partial, int rows, int K, int C, float* dw, hipStream_t s);
size_t dwconv_wgrad_scratch_floats(int B, int Ho, int Wo, int C, int K);
int launch_dwconv_wgrad(const float* x, const float* dy, float* dw, float* scratch, int B, int H, int W, int C, int K,
                        int stride, int pad_t, int pad_l, int Ho, int Wo, hipStream_t s, const float* in_scale = nullptr,
                        const float* in_shift = nullptr, int in_act = 0);
// (in_scale / in_shift: x is the RAW output of the producing conv; act(x * in_scale[c] + in_shift[c]) is applied as the patch is
// staged in LDS - only where dwconv_wgrad_xf_supported says the LDS form fits the layer)
bool dwconv_wgrad_xf_supported(int B, int H, int W, int C, int K, int stride, int Ho, int Wo);
size_t conv_wgrad_scratch_floats(int B, int Cin, int Cout, int KH, int KW, int Ho, int Wo);
// The split reduction of several filter gradients in ONE launch: launch_conv_wgrad with `defer` fills a job instead of
// running its reduce (the layer's partial tiles must then stay in `scratch` until launch_conv_wgrad_reduce_batched has run).
constexpr int WGRAD_REDUCE_JOBS = 48;
struct WgradReduceJob {
    const float* partial;
    float* dw;
    int splits, Cout, NC, Cin, KH, KW, mode;
};
struct WgradReduceJobs {
    WgradReduceJob j[WGRAD_REDUCE_JOBS];
    unsigned first_block[WGRAD_REDUCE_JOBS + 1];  // filled by the launcher: blocks [first_block[k], first_block[k + 1]) serve layer k
    int n;
};
int launch_conv_wgrad_reduce_batched(WgradReduceJobs& jobs, int n, hipStream_t s);
int launch_conv_wgrad(const float* x, int x_nchw, const float* dy, float* dw_oihw, int B, int H, int W, int Cin, int Cout,
                      int KH, int KW, int stride, int pad_t, int pad_l, int Ho, int Wo, float* scratch, hipStream_t s,
                      const float* gate = nullptr, WgradReduceJob* defer = nullptr);  // gate [B][Cin]: the filter gradient w.r.t. x * gate without materialising it
size_t conv_dgrad_packed_floats(int Cin, int Cout, int KH, int KW);
int conv_pack_dgrad_weights(const float* w_oihw, float* w_packed, int Cin, int Cout, int KH, int KW, hipStream_t s);
int launch_conv_dgrad(const float* dy, const float* w_dgrad_packed, const float* accumulate, float* dx, float* up, int B,
                      int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad_t, int pad_l, int Ho, int Wo,
                      hipStream_t s);

}  // namespace orbit
