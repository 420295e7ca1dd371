/*
 * orbit_hip.h — C-ABI of liborbit_hip.so: the MI355X (gfx950) implementation of ORBIT's episodic
 * few-shot recognition hot path (FewShotRecogniser.personalise()/predict()).
 *
 * The reference (microsoft/ORBIT-Dataset) is pure Python/PyTorch and has no FFI of its own; the seam
 * this library replaces is the ATen op sequence under these reference call sites (file:line are
 * relative to the reference tree):
 *
 *   orbit_proto_configure / _finalize   model/classifier_heads.py:94-119 (_build_class_reps),
 *                                        :232-263 (PrototypicalClassifier.configure)
 *   orbit_proto_predict                  model/classifier_heads.py:202-230 (PrototypicalClassifier.predict)
 *   orbit_mean_pool                      model/poolers.py:7-16 (MeanPooler.forward)
 *   orbit_extractor_*                    model/feature_extractors.py:37-79 (create_feature_extractor) and
 *                                        model/few_shot_recognisers.py:99-153 (_get_features[_in_batches]);
 *                                        "set_encoder": model/set_encoders.py:81-120 (SimplePrePoolNet)
 *   film_gamma / film_beta arguments     model/few_shot_recognisers.py:114-115,143-144
 *                                        (functional_call with the FiLM dict), model/film.py:38-94
 *   orbit_set_mean                       model/set_encoders.py:61-75 (SetEncoder.aggregate 'mean')
 *   orbit_filmgen_*                      model/feature_adapters.py:36-78 (FilmParameterGenerator),
 *                                        model/mlps.py:52-63 (DenseBlock)
 *   orbit_op_*                           single operators (conv/pool/depthwise/SE) exposed for parity tests
 *
 * Conventions
 *   - every data pointer is a DEVICE pointer to contiguous memory owned by the caller (fp32 unless stated,
 *     labels int64). The library never frees or retains caller memory past the call.
 *   - `stream` is a hipStream_t passed as void*. Calls only enqueue work and return immediately.
 *   - return 0 on success, negative on error; orbit_last_error() returns a thread-local message.
 *   - the library uses the calling thread's current HIP device; handles belong to the device they were
 *     created on. One process per GPU is the intended deployment.
 *   - no global mutable state besides the thread-local error string.
 */
#ifndef ORBIT_HIP_H
#define ORBIT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orbit_extractor orbit_extractor_t;
typedef struct orbit_filmgen orbit_filmgen_t;
typedef void* orbit_stream_t; /* hipStream_t */

#define ORBIT_OK 0
#define ORBIT_ERR_ARG (-1)
#define ORBIT_ERR_HIP (-2)
#define ORBIT_ERR_STATE (-3)
#define ORBIT_ERR_NOMEM (-4)

#define ORBIT_ACT_NONE 0
#define ORBIT_ACT_RELU 1
#define ORBIT_ACT_SILU 2
#define ORBIT_ACT_ELU 3   /* alpha = 1; only orbit_dense_rows */

#define ORBIT_REDUCE_NONE 0
#define ORBIT_REDUCE_MEAN 1
#define ORBIT_REDUCE_SUM 2

/* ---- library ---------------------------------------------------------------------------------- */
int orbit_version(void);
/* Call once per device before the first launch (idempotent): freed stream-ordered scratch stays in the device's default
 * memory pool instead of being returned - and synchronously re-allocated - around every task. */
int orbit_runtime_init(void);
const char* orbit_last_error(void);
/* number of visible HIP devices (<=0: no usable GPU); does not create a context on failure */
int orbit_device_count(void);
/* Runtime options (15). Every option has an environment default ORBIT_<NAME in upper case>; the defaults are the measured-best
 * path. They exist so that the parity tests can reach every kernel a default path uses and so that A/B comparisons run
 * inside one process on one box. Setting an option bumps an epoch that invalidates captured launch sequences.
 *  network runtime:
 *   "graph"         HIP-graph replay of extractor forwards: 0 = never, 1 = always, 2 = adaptive (default: only while an
 *                   eager kernel launch costs > ~12 us of host time on this host); read per forward
 *   "train_graph"   1 (default) = the training entry points (orbit_extractor_train_forward / _backward) replay captured HIP
 *                   graphs from the third sight of a call on (same pointers, same sizes); 0 = eager launches
 *   "mbconv_rows"   1 (default) = EfficientNet blocks 1.0 .. 3.0 (112x112 .. 28x28 maps) run the row-streaming fused front
 *                   (csrc/mbconv_rows.hip: expand 1x1 + BN + SiLU + depthwise + BN + SiLU + SE partials, expanded rows in an
 *                   LDS ring); 0 = conv + depthwise kernel pair everywhere. Read when a plan is created
 *   "stem_rows"     1 (default) = conv_stem + BN + SiLU + the first depthwise as one row-streaming kernel; 0 = direct stem
 *                   kernel + depthwise kernel. Read when a plan is created
 *   "train_dw_xf"   1 (default) = ORBIT_TRAIN_NO_BACKWARD forwards skip the activation pass between an expand / stem conv and
 *                   its depthwise conv (applied on load instead); 0 = always the separate pass
 *   "train_fused_fronts"  1 (default) = on ORBIT_TRAIN_NO_BACKWARD batch-statistics forwards (the LITE cache pass) the expansion
 *                   conv + depthwise conv of EfficientNet's 112x112 / 56x56 MBConv blocks run as a statistics sweep of the
 *                   expansion conv (nothing stored) + the row-streaming fused front of the inference plans in its RAW form
 *                   (raw depthwise outputs + their column sums): the 6x-expanded tensor never reaches HBM; 2 = every shape the
 *                   fused front serves; 0 = never (conv + depthwise pair); 3 = as 2, with the first BatchNorm's statistics from a
 *                   statistics sweep of the conv itself (bit-identical to the pair's) instead of the Gram matrix of the block
 *                   input (launch_bn_stats_from_gram: Cin = 16 / 24). Statistics agree to summation order
 *  dense convolutions (csrc/conv_igemm.hip, pw_rgemm.hip, conv_bf3.hip):
 *   "conv_tile"     0 = heuristic (default); 3 = 64x64, 4 = 128x32, 6 = 32x32 with K split over the four waves (the three
 *                   tilings the heuristic chooses from)
 *   "conv_bk"       cap the K-tile width: 0 = widest of 32/16/8 dividing Cin, 16 on the layers where that measured faster
 *                   (default); 8, 16, 32 (read when weights are packed AND at launch: set it before creating / finalizing a plan)
 *   "conv_splitk"   1 (default) = convs with few output tiles and a long reduction are split over K (partial tiles + a
 *                   deterministic reduce), 0 = never
 *   "conv_rgemm"    1 (default) = the barrier-free 16x16x4-MFMA register GEMM on fragment-packed weights serves the stride-1
 *                   pointwise convs for which it is a gain inside the network: projections of at most 8x8 maps from >= 512 to
 *                   more than 256 channels (EfficientNet-B0's 1152 -> 320 at 7x7; a rule on the layer, not on the batch, so a
 *                   frame's bits do not depend on the batch it arrives in); 0 = never; 2 = every conv it supports. Read at plan
 *                   creation (which filters get a fragment-ordered copy) and at launch
 *   "conv_bf3"      0 (default) = every product is an fp32 x fp32 MFMA. OPT-IN bits: 1 = dense convs with Cin % 16 == 0,
 *                   Cin >= 64, Cout >= 40 split both operands three ways into bf16 and sum six of the nine bf16 x bf16 products on
 *                   v_mfma_f32_32x32x16_bf16 (csrc/conv_bf3.hip); 2 = the expand GEMM of the row-streaming fused fronts does the
 *                   same; 3 = both. NOT the reference's arithmetic bit for bit - a measured alternative, frozen, never the
 *                   default, never part of bench.py's value
 *  depthwise kernel families (csrc/ops.hip): 1 = where measured faster (default), 0 = never, 2 = wherever it fits
 *   "dw_window"     register-window kernel (default: 3x3, stride 1, >= 14 rows)
 *   "dw_lds"        input patch staged in LDS (default: stride-1 5x5 and small 3x3 maps)
 *   "dw_pipe"       software-pipelined streaming kernel (default: large stride-2 layers)
 *  prototype head (csrc/head.hip):
 *   "head_stream"   1 (default) = the streaming distance kernel (rows requested before the weight staging, 8 waves x 2 rows)
 *                   for T = 1, D = 512 / 1280 launches with >= 64 query rows; 0 = the general LDS-staged kernel */
int orbit_set_option(const char* name, int value);
/* current value of an option (after its environment default was applied), -1 for an unknown name */
int orbit_get_option(const char* name);

/* ---- prototype head ---------------------------------------------------------------------------- */
/* The label set of a task without a host round trip: class_ids[0 .. cap) = the ascending unique values of labels[N] - the
 * logit column order, what the reference takes from torch.unique(context_labels) with an .item() loop
 * (model/classifier_heads.py:96-100,246-248) - slots beyond the count repeat the last value; *count = number of distinct
 * labels, or cap + 1 if there are more than cap. Both outputs are device memory; the caller copies *count to the host
 * whenever it needs it (orbit-dataset_amd/model/classifier_heads.py resolves it on a side stream while the extractor runs). */
int orbit_label_set(const int64_t* labels, int N, int64_t* class_ids, int cap, int32_t* count, orbit_stream_t stream);

/* Per-class sums of per-clip mean-pooled support features.
 *   feats   [n_tasks][N*T][D]   frame features, clip-major (clip i owns rows i*T .. i*T+T-1)
 *   labels  [n_tasks][N]        int64 clip labels
 *   class_ids [n_tasks][C]      int64 ascending unique labels (column order of the logits)
 *   sums    [n_tasks][C][D] out Σ over clips of class c of mean_T(feats)   (all-reduce payload)
 *   counts  [n_tasks][C]    out number of clips of class c (as float)      (all-reduce payload)
 * Summation order per (class, d) is ascending clip index: deterministic. */
int orbit_proto_configure(const float* feats, const int64_t* labels, const int64_t* class_ids,
                          int n_tasks, int N, int T, int D, int C,
                          float* sums, float* counts, orbit_stream_t stream);

/* W[c] = 2*mu_c, b[c] = -mu_c.mu_c with mu_c = sums[c]/counts[c]  (classifier_heads.py:253-255).
 * cosine != 0: b is not written (may be NULL). */
int orbit_proto_finalize(const float* sums, const float* counts, int n_tasks, int C, int D, int cosine,
                         float* W, float* b, orbit_stream_t stream);

/* logits[m][c] = logit_scale * (q_m . W_c + b_c)                       (euclidean, :213)
 *              = logit_scale * cos(q_m, W_c), eps 1e-8 per norm        (cosine, :215-217)
 * with q_m = mean over the T frames of clip m of Q.   Q [n_tasks][M*T][D]; logits [n_tasks][M][C];
 * argmax [n_tasks][M] int32 or NULL (first maximal column, as torch.argmax). */
int orbit_proto_predict(const float* Q, const float* W, const float* b,
                        int n_tasks, int M, int T, int D, int C, float logit_scale, int cosine,
                        float* logits, int32_t* argmax, orbit_stream_t stream);

/* out[i][d] = mean_t x[i*T+t][d]  (poolers.py:13-16) */
int orbit_mean_pool(const float* x, int N, int T, int D, float* out, orbit_stream_t stream);

/* Clip pooling of a video's sliding windows without duplicating frames: x [F][D] per-frame features, out [F][D] with
 * out[f] = mean_{j<T} x[max(f-T+1+j, 0)] — attach_frame_history (data/utils.py:8-28) + MeanPooler (poolers.py:13-16) on
 * features instead of on frames, bit-identical to pooling the T-times larger clip tensor. */
int orbit_history_mean_pool(const float* x, int F, int T, int D, float* out, orbit_stream_t stream);
/* out[d] = (1/n) Σ_i x[i][d]  (set_encoders.py:70-71; also the LITE concat-mean) */
int orbit_set_mean(const float* x, int n, int D, float* out, orbit_stream_t stream);

/* ---- feature extractors / set encoder ---------------------------------------------------------- */
/* name: "resnet18" | "efficientnet_b0" | "set_encoder".  H,W: frame size the plan is built for. */
int orbit_extractor_create(const char* name, int H, int W, orbit_extractor_t** out);
/* flags: ORBIT_PLAN_UNFUSED = a plan for forwards that record a tape or use batch statistics (the LITE training step,
 * few_shot_recognisers.py:176-183): every MBConv block stays a conv + depthwise pair, whose outputs the backward needs;
 * same parameters in the same order as the default plan. */
#define ORBIT_PLAN_UNFUSED 1
int orbit_extractor_create_ex(const char* name, int H, int W, int flags, orbit_extractor_t** out);
void orbit_extractor_destroy(orbit_extractor_t* fe);

/* state_dict enumeration: parameter/buffer keys the extractor expects (torch state_dict names). */
int orbit_extractor_num_params(const orbit_extractor_t* fe);
const char* orbit_extractor_param_name(const orbit_extractor_t* fe, int i);
size_t orbit_extractor_param_numel(const orbit_extractor_t* fe, int i);

/* copy one tensor (host or device pointer, fp32, torch layout) into the extractor. */
int orbit_extractor_load(orbit_extractor_t* fe, const char* key, const float* data, size_t numel);
/* Same for a source already resident in HBM: stream-ordered device-to-device copy (no host synchronisation); how the
 * parameters follow optimizer steps during meta-training. */
int orbit_extractor_load_async(orbit_extractor_t* fe, const char* key, const float* device_data, size_t numel,
                               orbit_stream_t stream);
/* All parameters in one launch: device_ptrs[i] = device pointer of parameter i (orbit_extractor_param_name order, fp32,
 * contiguous, orbit_extractor_param_numel(i) elements). The pointer table is cached; one gather kernel copies every
 * tensor into the plan, stream-ordered (what optimizer.step() -> next forward needs every training step: one launch
 * instead of one hipMemcpyAsync per tensor). */
int orbit_extractor_load_all_async(orbit_extractor_t* fe, const float* const* device_ptrs, int n, orbit_stream_t stream);
/* repack conv weights for the kernels, precompute folded BN for the non-FiLM case. Call after loads. */
int orbit_extractor_finalize(orbit_extractor_t* fe, orbit_stream_t stream);

int orbit_extractor_output_size(const orbit_extractor_t* fe);
/* FiLM-tagged BatchNorms in module-traversal order: count, per-slot channel count and module name,
 * total channels (= length of film_gamma / film_beta). */
int orbit_extractor_film_slots(const orbit_extractor_t* fe);
int orbit_extractor_film_slot_channels(const orbit_extractor_t* fe, int slot);
const char* orbit_extractor_film_slot_name(const orbit_extractor_t* fe, int slot);
int orbit_extractor_film_size(const orbit_extractor_t* fe);

size_t orbit_extractor_workspace_bytes(const orbit_extractor_t* fe, int B);
/* multiply-accumulates per frame of the plan (for roofline accounting) */
double orbit_extractor_macs_per_frame(const orbit_extractor_t* fe);

/* frames [B][3][H][W] NCHW fp32 -> feats [B][D].
 * film_gamma/film_beta: NULL, or the per-task BatchNorm weight/bias of every FiLM slot concatenated in
 * slot order (the tensors the reference passes through functional_call). */
int orbit_extractor_forward(orbit_extractor_t* fe, const float* frames, int B,
                            const float* film_gamma, const float* film_beta,
                            float* feats, void* workspace, size_t workspace_bytes, orbit_stream_t stream);

/* ---- FiLM parameter generator ------------------------------------------------------------------ */
/* n_gen generators (sorted FiLM-name order). Generator i: Linear(z_dim,hid) -> LayerNorm(hid) -> ReLU ->
 * Linear(hid,out_size[i]); kind[i]: 0 = 'weight' (gamma' = g0*(o*r+1)), 1 = 'bias' (beta' = b0 + o*r);
 * dst_offset[i]: offset of its output inside film_gamma (kind 0) / film_beta (kind 1). */
int orbit_filmgen_create(int n_gen, int z_dim, int hidden, const int* out_size, const int* kind,
                         const int* dst_offset, orbit_filmgen_t** out);
void orbit_filmgen_destroy(orbit_filmgen_t* g);
/* tensor: "w1"[hid][z] "b1"[hid] "ln_w"[hid] "ln_b"[hid] "w2"[out][hid] "b2"[out] "reg"[out] "init"[out] */
int orbit_filmgen_load(orbit_filmgen_t* g, int gen, const char* tensor, const float* data, size_t numel);
/* Stream-ordered form for parameters that already live on the device (they follow optimizer steps): ONE gather kernel on
 * `stream` copies all 8 tensors of every generator; device_ptrs[8*gen + t], t in the order listed above. No host sync
 * unless the pointer table changed since the previous call. */
int orbit_filmgen_load_all_async(orbit_filmgen_t* g, const float* const* device_ptrs, int n, orbit_stream_t stream);
/* z [z_dim] -> film_gamma, film_beta (each film_size floats), l2[1] = Σ_i Σ reg_i^2 */
int orbit_filmgen_forward(orbit_filmgen_t* g, const float* z, float* film_gamma, float* film_beta,
                          float* l2, orbit_stream_t stream);

/* ---- single operators (NHWC activations), exposed for parity tests and reuse -------------------- */
/* General convolution as implicit GEMM on fp32 MFMA.
 *   x: NHWC [B][H][W][Cin]  (x_nchw != 0: NCHW [B][Cin][H][W], Cin <= 4 only — the network stems)
 *   w: torch OIHW [Cout][Cin][KH][KW];  y: NHWC [B][Ho][Wo][Cout]  (pool2: [B][Ho/2][Wo/2][Cout])
 *   y = act( conv(x*gate) * scale + shift + residual ), optional fused 2x2/2 max-pool (floor).
 *   scale/shift [Cout] (NULL = 1/0), residual NHWC like y or NULL, gate [B][Cin] or NULL. */
int orbit_op_conv2d(const float* x, int x_nchw, const float* w, float* y,
                    const float* scale, const float* shift, const float* residual, const float* gate,
                    int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride,
                    int pad_top, int pad_left, int Ho, int Wo, int act, int pool2,
                    orbit_stream_t stream);
/* depthwise KxK conv, NHWC, w torch [C][1][K][K]; y = act(dw(x)*scale+shift) */
int orbit_op_dwconv2d(const float* x, const float* w, float* y, const float* scale, const float* shift,
                      int B, int H, int W, int C, int K, int stride, int pad_top, int pad_left,
                      int Ho, int Wo, int act, orbit_stream_t stream);
/* Training forms of the two ops (single-operator entries of the parity tests, like the ones above): the convolution without
 * epilogue whose kernel also emits the train-mode BatchNorm statistics of its output (per-channel sums and sums of squares,
 * stats [2][Cout]; *stat_blocks = row blocks written, 0 when this launch shape does not emit them), optionally with the
 * squeeze-excite gate multiplied into x; and the depthwise convolution that applies the PRECEDING layer's BatchNorm +
 * activation to x as it loads it (in_scale / in_shift nullable) and emits the statistics of its own output. Reference: the
 * nn.Conv2d -> nn.BatchNorm2d (train()) pairs of the extractor under model/few_shot_recognisers.py:176-183. */
int orbit_op_conv2d_train(const float* x, int x_nchw, const float* w, float* y, const float* gate, int B, int H, int W,
                          int Cin, int Cout, int KH, int KW, int stride, int pad_top, int pad_left, int Ho, int Wo,
                          float* stats, int* stat_blocks, orbit_stream_t stream);
int orbit_op_dwconv2d_train(const float* x, const float* w, float* y, const float* in_scale, const float* in_shift,
                            int in_act, int B, int H, int W, int C, int K, int stride, int pad_top, int pad_left, int Ho,
                            int Wo, float* stats, orbit_stream_t stream);
/* max-pool NHWC, -inf padding */
int orbit_op_maxpool2d(const float* x, float* y, int B, int H, int W, int C, int K, int stride, int pad,
                       int Ho, int Wo, orbit_stream_t stream);
/* global average pool NHWC [B][HW][C] -> [B][C] */
int orbit_op_avgpool(const float* x, float* y, int B, int HW, int C, orbit_stream_t stream);
/* squeeze-excite gate: g[b][c] = sigmoid(W2 . silu(W1 . pooled[b] + b1) + b2);  W1 [R][C], W2 [C][R] */
int orbit_op_se_gate(const float* pooled, const float* w1, const float* b1, const float* w2,
                     const float* b2, float* gate, int B, int C, int R, orbit_stream_t stream);

/* fused MBConv front half (EfficientNet InvertedResidual, Cin <= 40): y = silu(bn2(dw_KxK(silu(bn1(x . w1^T))))) with the
 * expanded tensor kept in LDS. x NHWC [B][H][W][Cin]; w1 torch [mid][Cin][1][1]; wdw torch [mid][1][K][K];
 * scale/shift = folded BatchNorms [mid]; y NHWC [B][Ho][Wo][mid]; pool_partial [B][tiles][mid] or NULL: partial sums
 * of y for the SE average pool, tiles = orbit_op_mbconv_front_partials(...) (the kernel form - tiled, whole-map or
 * row-streaming or not, option mbconv_rows - decides how many partial sums a frame has). */
int orbit_op_mbconv_front_partials(int H, int W, int Cin, int mid, int K, int stride);
int orbit_op_mbconv_front(const float* x, const float* w1, const float* scale1, const float* shift1,
                          const float* wdw, const float* scale2, const float* shift2, float* y, float* pool_partial,
                          int B, int H, int W, int Cin, int mid, int K, int stride, int pad_top, int pad_left,
                          int Ho, int Wo, orbit_stream_t stream);
/* The stem form of the same kernel (timm tf_efficientnet_b0: conv_stem -> bn1 -> SiLU -> blocks.0.0.conv_dw -> bn1 ->
 * SiLU, reached from model/feature_extractors.py:39-43): frames NCHW [B][3][FH][FW], w_stem torch [mid][3][3][3]
 * (3x3 stride 2, padding spad_top / spad_left before, the rest after), depthwise 3x3 stride 1 on the stem's H x W output
 * grid; y NHWC [B][Ho][Wo][mid]; pool_partial [B][ceil(Ho/8)*ceil(Wo/8)][mid] or NULL. The stem output never reaches HBM. */
int orbit_op_stem_dw_front_partials(int H, int W, int mid);  /* pool_partial rows per frame under the current options */
int orbit_op_stem_dw_front(const float* frames, const float* w_stem, const float* scale1, const float* shift1,
                           const float* wdw, const float* scale2, const float* shift2, float* y, float* pool_partial,
                           int B, int FH, int FW, int spad_top, int spad_left, int H, int W, int mid, int pad_top,
                           int pad_left, int Ho, int Wo, orbit_stream_t stream);

/* ---- Versa / Mahalanobis heads -----------------------------------------------------------
 * y[r][o] = act(x[r] . W[o] + b[o]) (+ residual[r][o]) for a few rows r < R <= 16 (one row per class): the layers of
 * the Versa hyper-networks, model/mlps.py:33-50 DenseResidualBlock (linear -> ELU -> linear -> ELU -> linear (+ x)),
 * applied to the class means by VersaClassifier.configure, model/classifier_heads.py:158-180. W is torch [out][in]. */
int orbit_dense_rows(const float* x, int R, int in, const float* W, const float* b, int out, int act,
                     const float* residual, float* y, orbit_stream_t stream);
/* batched inverse of `batch` symmetric positive definite n x n matrices (in-place blocked Gauss-Jordan, no pivoting);
 * replaces torch.inverse in MahalanobisClassifier.configure (classifier_heads.py:288,311). A == Ainv allowed. */
int orbit_spd_inverse(const float* A, float* Ainv, int n, int batch, orbit_stream_t stream);
size_t orbit_mahalanobis_workspace_bytes(int N, int M, int D, int C);
/* MahalanobisClassifier.configure (classifier_heads.py:284-327): features [N][D], labels [N], class_ids [C] ascending
 * -> means [C][D], task_mean [D], precisions [C][D][D] = inverse(l cov_c + (1-l) cov_task + I), l = n_c/(n_c+1),
 * task_precision [D][D] = inverse(cov_task + I); covariances are torch.cov(correction=1), a single-example class takes
 * the reference's scalar branch (:361-364). */
int orbit_mahalanobis_configure(const float* features, const int64_t* labels, const int64_t* class_ids, int N, int D,
                                int C, float* means, float* task_mean, float* precisions, float* task_precision,
                                void* workspace, size_t workspace_bytes, orbit_stream_t stream);
/* MahalanobisClassifier.predict (:329-347): logits[m][c] = -scale * (mu_c - q_m)^T P_c (mu_c - q_m). D % 4 == 0. */
int orbit_mahalanobis_predict(const float* features, const float* means, const float* precisions, int M, int D, int C,
                              float logit_scale, float* logits, void* workspace, size_t workspace_bytes,
                              orbit_stream_t stream);

/* d(features) of orbit_mahalanobis_predict (means / precisions are constants, as in the reference :324-327):
 * dfeatures[m] = scale * sum_c dlogits[m][c] (P_c + P_c^T)(mu_c - q_m) */
int orbit_mahalanobis_predict_backward(const float* dlogits, const float* features, const float* means,
                                       const float* precisions, int M, int D, int C, float logit_scale, float* dfeatures,
                                       void* workspace, size_t workspace_bytes, orbit_stream_t stream);

/* ---- training (LITE meta-training step) -----------------------------------------------------
 * What `loss.backward()` runs in the reference (single-step-learner.py:234) for the graph recorded by
 * model/few_shot_recognisers.py:99-122 (_get_features, grad enabled), :345-356 (_get_task_embedding on the LITE
 * subset) and model/classifier_heads.py:202-230 (head). BatchNorm mode follows few_shot_recognisers.py:176-183.
 * Available for the resnet18, efficientnet_b0 and set_encoder plans (orbit_extractor_supports_training is 0 only for a
 * plan built with the opt-in fused MBConv front op). */
int orbit_extractor_supports_training(const orbit_extractor_t* fe);
size_t orbit_extractor_tape_bytes(const orbit_extractor_t* fe, int B);
size_t orbit_extractor_backward_workspace_bytes(const orbit_extractor_t* fe, int B);
size_t orbit_extractor_grad_floats(const orbit_extractor_t* fe);          /* length of the flat gradient buffer */
size_t orbit_extractor_param_offset(const orbit_extractor_t* fe, int i);  /* offset of parameter i inside it */
size_t orbit_extractor_bn_stat_floats(const orbit_extractor_t* fe);
/* dst[0][.] running means, dst[1][.] running variances of every BatchNorm (in parameter order, each padded to a
 * multiple of 4 channels; row length = orbit_extractor_bn_stat_floats): how the caller's state_dict buffers follow the
 * running-stat updates of train-mode forwards */
int orbit_extractor_export_bn_stats(orbit_extractor_t* fe, float* dst, orbit_stream_t stream);
/* Forward that records the tape (raw convolution outputs, activations, pooling argmax; caller-owned, 256-B aligned).
 * bn_train != 0: batch statistics + running-stat update with `momentum` (nn.BatchNorm2d semantics);
 * bn_train == 0: running statistics. film_gamma/film_beta as in orbit_extractor_forward. */
int orbit_extractor_train_forward(orbit_extractor_t* fe, const float* frames, int B, const float* film_gamma,
                                  const float* film_beta, int bn_train, float momentum, float* feats, void* tape,
                                  size_t tape_bytes, orbit_stream_t stream);
/* The same with flags. ORBIT_TRAIN_NO_BACKWARD: no backward will be run on this tape - the forwards the reference issues
 * under torch.no_grad() while the extractor is in train() (LITE's cache passes, few_shot_recognisers.py:134-146,388-437):
 * batch statistics and running-stat updates as above, but activations no later kernel of THIS forward needs are not
 * materialised (the depthwise convs apply the preceding BatchNorm + activation while loading the raw conv output). Features
 * are bit-identical to the flag-less call; orbit_extractor_backward on such a tape is undefined. */
#define ORBIT_TRAIN_NO_BACKWARD 1
/* ORBIT_TRAIN_DEFER_RUNNING_STATS (bn_train != 0): the running-statistics update is NOT applied; the batch mean / unbiased
 * variance of every BatchNorm stay on the tape and orbit_extractor_apply_deferred_bn_stats applies
 * running = (1 - momentum) * running + momentum * stat afterwards (same arithmetic, so the result equals the in-place update
 * made at that point of the sequence). Such a forward touches no mutable state of the plan: it may run on ANOTHER stream
 * while a forward of the same plan that does update the statistics is in flight - LITE re-encodes its H-clip subset (16
 * frames: launch-bound kernels) beside the cache pass over the whole context set (few_shot_recognisers.py:388-437), and the
 * reference's update order (cache pass, then subset) is kept by applying the subset's update after both. */
#define ORBIT_TRAIN_DEFER_RUNNING_STATS 2
int orbit_extractor_train_forward_ex(orbit_extractor_t* fe, const float* frames, int B, const float* film_gamma,
                                     const float* film_beta, int bn_train, float momentum, float* feats, void* tape,
                                     size_t tape_bytes, int flags, orbit_stream_t stream);
int orbit_extractor_apply_deferred_bn_stats(orbit_extractor_t* fe, const void* tape, size_t tape_bytes, int B, float momentum,
                                            orbit_stream_t stream);
/* Reverse pass for the tape of one train_forward (same frames / B / film / bn_train).
 *   dfeats [B][D]: gradient w.r.t. the features.
 *   param_grads: NULL (frozen extractor) or orbit_extractor_grad_floats() floats; the gradient of parameter i is
 *     WRITTEN at orbit_extractor_param_offset(i) in the parameter's torch layout (OIHW filters). Slots of buffers
 *     (running statistics) and of BatchNorm weight/bias replaced by FiLM vectors are left untouched.
 *     filter_grads == 0: only the BatchNorm weight / bias gradients are computed (FiLM fine-tuning of a frozen
 *     extractor, few_shot_recognisers.py:196-199); != 0: every parameter.
 *   dfilm_gamma / dfilm_beta: NULL or film_size floats each, written.
 * Deterministic: fixed reduction order, no atomics. */
int orbit_extractor_backward(orbit_extractor_t* fe, const float* frames, int B, const float* film_gamma,
                             const float* film_beta, int bn_train, const float* dfeats, const void* tape,
                             size_t tape_bytes, float* param_grads, int filter_grads, float* dfilm_gamma,
                             float* dfilm_beta, void* workspace, size_t workspace_bytes, orbit_stream_t stream);
/* Training entry points replay captured HIP graphs when a call repeats an earlier call's pointers and scalars exactly
 * (option train_graph = 1, off by default: it frees the host but the step is GPU-bound; the plan's own buffers never move). Diagnostics: calls that replayed / ran eagerly. */
int orbit_extractor_train_graph_stats(const orbit_extractor_t* fe, long* replays, long* eager);
/* Backward of orbit_filmgen_forward: given d(film_gamma), d(film_beta) (film_size floats each) and d(l2) ([1], NULL = 0)
 * writes the gradients of every generator parameter into `grads` (orbit_filmgen_grad_floats floats; tensor t of
 * generator i at orbit_filmgen_param_offset(g, i, t), same tensor names as orbit_filmgen_load except "init") and
 * d(z) [z_dim] (NULL to skip). Reference: autograd through model/feature_adapters.py:66-78, model/mlps.py:52-63. */
size_t orbit_filmgen_grad_floats(const orbit_filmgen_t* g);
size_t orbit_filmgen_param_offset(const orbit_filmgen_t* g, int gen, const char* tensor);
int orbit_filmgen_backward(orbit_filmgen_t* g, const float* z, const float* dfilm_gamma, const float* dfilm_beta,
                           const float* dl2, float* grads, float* dz, orbit_stream_t stream);
/* d(features) of orbit_proto_predict for ONE task: features [M*T][D] (frame features, pooled over T inside),
 * weight [C][D] (treated as a constant: the reference wraps it in nn.Parameter, classifier_heads.py:261-263),
 * dlogits [M][C] -> dfeatures [M*T][D]. C <= 64. */
int orbit_proto_predict_backward(const float* dlogits, const float* features, const float* weight, int M, int T, int D,
                                 int C, float logit_scale, int cosine, float* dfeatures, orbit_stream_t stream);

/* parameter gradients of a linear head logits = scale * (features . W^T + b) (LinearClassifier.predict,
 * classifier_heads.py:62-76, trained by the multi-step finetuner few_shot_recognisers.py:209-247): dweight [C][D],
 * dbias [C] (nullable). The feature gradient is orbit_proto_predict_backward (euclidean form). */
int orbit_linear_head_backward(const float* dlogits, const float* features, int M, int D, int C, float logit_scale,
                               float* dweight, float* dbias, orbit_stream_t stream);

/* Cross-entropy of logits [N][C] against int64 labels [N]: reference utils/optim.py:8-9 (F.cross_entropy, the `loss` of
 * single-step-learner.py:77,225-232 and of the FineTuner, few_shot_recognisers.py:231-236).
 * row_loss [N] = logsumexp(logits[i]) - logits[i][labels[i]] (NaN for a label outside [0, C)); softmax [N][C] is kept for
 * the backward (NULL to skip); loss [1] = mean / sum of the rows in one fixed order (ORBIT_REDUCE_MEAN / _SUM; unused for
 * ORBIT_REDUCE_NONE). N = 0 is allowed (mean = NaN, sum = 0, like torch). */
int orbit_cross_entropy_forward(const float* logits, const int64_t* labels, int N, int C, int reduction, float* row_loss,
                                float* softmax, float* loss, orbit_stream_t stream);
/* dlogits [N][C] = g_i * (softmax - onehot(labels)); g_i = grad[0] / N (mean), grad[0] (sum), grad[i] (none). `grad` is a
 * DEVICE pointer (the upstream gradient): nothing is read back to the host. */
int orbit_cross_entropy_backward(const float* softmax, const int64_t* labels, const float* grad, int N, int C,
                                 int reduction, float* dlogits, orbit_stream_t stream);

/* single training operators (NHWC, [M][C] = [B*H*W][C]), exposed for parity tests against torch autograd */
/* out = act(BN_train(y) + residual): batch mean / biased variance, saves mean and 1/sqrt(var+eps), updates the running
 * statistics in place when given (unbiased variance, momentum). gamma/beta NULL = 1/0. act: NONE or RELU. */
int orbit_op_bn_train_forward(const float* y, int M, int C, const float* gamma, const float* beta, float eps,
                              float momentum, float* running_mean, float* running_var, const float* residual, int act,
                              float* out, float* save_mean, float* save_invstd, orbit_stream_t stream);
/* Batch statistics (mean, 1/sqrt(biased var + eps)) of y = W x for a POINTWISE conv (W [C][Cin], torch's 1x1 filter layout;
 * x [P][Cin] NHWC pixels; Cin = 16 or 24) from the Gram matrix of x - without forming y (csrc/train_ops.hip): the first
 * sweep of the two-sweep fused MBConv front on the no-grad cache pass of the LITE step (few_shot_recognisers.py:404-408). */
int orbit_op_bn_stats_from_gram(const float* x, int P, int Cin, const float* w, int C, float eps, float* save_mean,
                                float* save_invstd, orbit_stream_t stream);
/* backward of out = act(BN(y) + residual): dy, dres (= dout * act', NULL to skip), dgamma, dbeta.
 * train != 0: batch-statistics form; train == 0: mean/invstd are running statistics (constants). */
int orbit_op_bn_backward(const float* dout, const float* out, const float* y, int M, int C, const float* gamma,
                         const float* mean, const float* invstd, int train, int act, float* dy, float* dres,
                         float* dgamma, float* dbeta, orbit_stream_t stream);
/* dx of a convolution: dy NHWC [B][Ho][Wo][Cout], w OIHW, dx NHWC [B][H][W][Cin] (= accumulate + grad if given) */
int orbit_op_conv2d_dgrad(const float* dy, const float* w, const float* accumulate, float* dx, int B, int H, int W,
                          int Cin, int Cout, int KH, int KW, int stride, int pad_top, int pad_left, int Ho, int Wo,
                          orbit_stream_t stream);
/* dw (OIHW) of a convolution; x NHWC (or NCHW when x_nchw, Cin <= 4) */
int orbit_op_conv2d_wgrad(const float* x, int x_nchw, const float* dy, float* dw, int B, int H, int W, int Cin, int Cout,
                          int KH, int KW, int stride, int pad_top, int pad_left, int Ho, int Wo, orbit_stream_t stream);
/* The filter gradient of a squeeze-excite-gated pointwise projection, dW[co][ci] = sum_m dy[m][co] * x[m][ci] * gate[b(m)][ci]
 * (timm InvertedResidual.conv_pwl behind the SE module; x NHWC [B][H][W][Cin], gate [B][Cin], dy [B][H][W][Cout]): the product
 * x * gate is never materialised. Single-operator entry for the parity tests. */
int orbit_op_conv2d_wgrad_gated(const float* x, const float* gate, const float* dy, float* dw, int B, int H, int W, int Cin,
                                int Cout, orbit_stream_t stream);
/* max-pool that records the argmax (position inside the window, first maximum in scan order) and its backward */
int orbit_op_maxpool2d_train(const float* x, float* y, uint8_t* idx, int B, int H, int W, int C, int K, int stride,
                             int pad, int Ho, int Wo, orbit_stream_t stream);
int orbit_op_maxpool2d_backward(const float* dy, const uint8_t* idx, float* dx, int B, int H, int W, int C, int K,
                                int stride, int pad, int Ho, int Wo, orbit_stream_t stream);
int orbit_op_avgpool_backward(const float* dy, float* dx, int B, int HW, int C, orbit_stream_t stream);
/* depthwise KxK convolution backward (K in {3,5}): dx NHWC [B][H][W][C] and/or dw torch [C][1][K][K] (either may be NULL) */
int orbit_op_dwconv2d_backward(const float* x, const float* w, const float* dy, float* dx, float* dw, int B, int H, int W,
                               int C, int K, int stride, int pad_top, int pad_left, int Ho, int Wo,
                               orbit_stream_t stream);
/* The depthwise data gradient continued through the producer's activation, with the first pass of that producer's BatchNorm
 * backward in its epilogue: g = dx * act'(y_raw * scale[c] + shift[c]) [B][H][W][C] and sums [2][C] = sum over (b,h,w) of g and
 * of g * (y_raw - mean[c]) * invstd[c] (reference: autograd through timm InvertedResidual's conv_dw <- act1 <- bn1,
 * few_shot_recognisers.py:99-122). dw (optional, [C][1][K][K]): the layer's filter gradient from the same pass - its input is
 * act(y_raw * scale + shift), rebuilt at the output pixels anyway (stride-2 layers). Fails with ORBIT_ERR_ARG where no kernel
 * form carries the epilogue / the filter gradient (the caller then runs the separate passes). Single-operator entry for the
 * parity tests. */
int orbit_op_dwconv2d_dgrad_bn(const float* dy, const float* w, const float* y_raw, const float* mean, const float* invstd,
                               const float* scale, const float* shift, int act, float* g, float* sums, float* dw, int B, int H,
                               int W, int C, int K, int stride, int pad_top, int pad_left, int Ho, int Wo, orbit_stream_t stream);
/* The depthwise filter gradient with the layer's input rebuilt on load: x_raw is the RAW output of the preceding convolution and
 * the layer's input is act(x_raw * in_scale[c] + in_shift[c]) (that convolution's batch-statistics BatchNorm + SiLU / ReLU;
 * reference: autograd through timm InvertedResidual's bn1 + act1 + conv_dw, few_shot_recognisers.py:99-122). The taped forward
 * of the LITE step never writes the activated tensor. dw [C][1][K][K]. Single-operator entry for the parity tests. */
int orbit_op_dwconv2d_wgrad_xf(const float* x_raw, const float* in_scale, const float* in_shift, int in_act, const float* dy,
                               float* dw, int B, int H, int W, int C, int K, int stride, int pad_top, int pad_left, int Ho, int Wo,
                               orbit_stream_t stream);
/* backward of the squeeze-excite product xg = x * gate(mean_hw(x)) (timm SqueezeExcite: conv_reduce -> SiLU ->
 * conv_expand -> sigmoid): given dxg writes dx and (all or none) dW1 [R][C], db1 [R], dW2 [C][R], db2 [C].
 * x NHWC [B][HW][C]; pooled [B][C] = mean over HW of x. */
int orbit_op_se_gate_backward(const float* dxg, const float* x, const float* pooled, const float* w1, const float* b1,
                              const float* w2, const float* b2, float* dx, float* dw1, float* db1, float* dw2,
                              float* db2, int B, int HW, int C, int R, orbit_stream_t stream);

/* ---- input side: 8-bit frames -> normalised fp32 NCHW ([B][3][H][W], what the extractors consume) -------------------
 * frames: device pointer, [B][H][W][3] when layout_hwc != 0 (decoded images) or [B][3][H][W]; mean3 / std3: HOST arrays.
 * out = ((u8 / 255) - mean) / std per channel: to_tensor + normalize of data/datasets.py:422-431, bit-identical. */
int orbit_frames_from_uint8(const uint8_t* frames, int layout_hwc, int B, int H, int W, const float* mean3,
                            const float* std3, float* out_nchw, orbit_stream_t stream);

/* ---- measurement: per-launch HIP-event timing of the dominant kernel (conv_igemm, all variants) ---- */
int orbit_prof_enable(int on);   /* on: reset and start recording an event pair per launch on its stream */
/* waits for the recorded launches, returns summed duration, summed ALGORITHMIC flops and launch count */
int orbit_prof_collect(double* total_ms, double* total_flops, long* launches);
int orbit_prof_num_variants(void);
/* one row per kernel instantiation: launches, summed duration, summed algorithmic flops and algorithmic HBM bytes */
int orbit_prof_variant(int i, char* name48, long* launches, double* ms, double* flops, double* bytes);
/* The roofs a launch's floor is priced on (defaults: 6.3e12 B/s achievable HBM, 157.3e12 FLOP/s fp32 MFMA, 5.93e12 SiLU
 * evaluations/s = 11.06 ns of a SIMD per 64 of them x 1024 SIMDs; set before orbit_prof_collect) and, per row, the sum over
 * its launches of max(bytes / HBM rate, FLOP / matrix rate), the same with the SiLU evaluations' VALU time added to the
 * matrix time (a gfx950 SIMD issues either an MFMA or VALU instructions, never both), and the SiLU count. Round 6: every
 * launch of the inference path - depthwise, squeeze-excite gate, pooling, head kernels included - records a row (bench.py
 * roofline.families). */
int orbit_prof_set_roofs(double hbm_bytes_per_s, double matrix_flop_per_s, double silu_evals_per_s);
int orbit_prof_variant_floor(int i, double* floor_ms, double* floor_simd_ms, double* silu_evals);

/* ---- RCCL over xGMI (one process per GPU) ------------------------------------------------------- */
/* The reference has no collectives; this is the exchange step of the support-sharded variant: the
 * all-reduce(SUM) of orbit_proto_configure's sums/counts (and of set-encoder embedding sums). */
#define ORBIT_COMM_ID_BYTES 128
int orbit_comm_unique_id(void* out128);                       /* rank 0: create the rendezvous id */
int orbit_comm_init(int rank, int world, const void* unique_id);
int orbit_comm_world(void);
int orbit_comm_rank(void);
int orbit_allreduce_sum(float* buf, size_t n, orbit_stream_t stream);  /* in place */
void orbit_comm_destroy(void);

/* ---- one-shot peer-to-peer all-reduce(SUM) over xGMI for the SMALL exchange steps (csrc/comm.hip) -----------------
 * The prototype payload of a support-sharded task is 25.6 KB: a latency-bound message. Every rank pushes its payload
 * into a slot of every peer's inbox (IPC-mapped device memory, all xGMI links in parallel), raises a flag and sums the
 * world slots of its own inbox in rank order: one hop, bit-identical result on every rank. orbit_allreduce_sum (RCCL)
 * stays the baseline and the path for large buffers.
 *   create (every rank) -> export 64-byte IPC handle -> exchange handles (host side) -> connect -> allreduce ... */
#define ORBIT_P2P_HANDLE_BYTES 64
typedef struct orbit_p2p orbit_p2p_t;
int orbit_p2p_create(int rank, int world, size_t max_floats, orbit_p2p_t** out);
int orbit_p2p_export(orbit_p2p_t* c, void* handle64);
int orbit_p2p_connect(orbit_p2p_t* c, const void* handles /* [world][64], own entry ignored */);
int orbit_p2p_allreduce_sum(orbit_p2p_t* c, float* buf, size_t n, orbit_stream_t stream); /* in place, n <= max_floats */
/* Large payloads (the flat gradient bucket of the task-parallel LITE step, reference single-step-learner.py:162-166,231
 * under data parallelism): direct reduce-scatter + all-gather over the point-to-point mesh, every element
 * summed once in rank order (bit-identical on all ranks). In place; n <= world * max_floats / 2. */
int orbit_p2p_allreduce_sum_sharded(orbit_p2p_t* c, float* buf, size_t n, orbit_stream_t stream);
/* 0 ok; k > 0: a wait for rank (k-1) % 100's flag timed out (~4 s) and that all-reduce's buffer was filled with NaN instead
 * of a partial sum. Reads a host-mapped word, does not synchronise the device: it covers the all-reduces that have completed. */
int orbit_p2p_error(orbit_p2p_t* c);
/* How the inbox was allocated. Peers write it and the owner polls it while kernels run, so it must be uncached or
 * fine-grained device memory (coarse-grained memory is only coherent at kernel boundaries); orbit_p2p_create fails rather
 * than fall back to coarse-grained memory unless ORBIT_P2P_ALLOW_COARSE=1. */
#define ORBIT_P2P_MEM_UNCACHED 1
#define ORBIT_P2P_MEM_FINEGRAINED 2
#define ORBIT_P2P_MEM_COARSE 3
int orbit_p2p_memory_kind(orbit_p2p_t* c);
void orbit_p2p_destroy(orbit_p2p_t* c);

#ifdef __cplusplus
}
#endif
#endif /* ORBIT_HIP_H */
