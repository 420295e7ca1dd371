// Training runtime of the network plans: forward with a tape (raw convolution outputs + activations + pooling
// argmax), train- or eval-mode BatchNorm, and the reverse pass producing parameter gradients (flat, in the layout of
// the parameter pool), FiLM gamma/beta gradients and nothing else (frames need no gradient).
//
// Reference: the graph autograd records for model/few_shot_recognisers.py:99-122 (_get_features with grad enabled),
// :345-356 (_get_task_embedding on the LITE subset) and walks in single-step-learner.py:234 (loss.backward()).
// BatchNorm mode follows :176-183: batch statistics (and running-stat updates, also on the no-grad LITE cache
// passes) iff the extractor is being learned; the set encoder always normalises with running statistics.
//
// Generic over the plan ops: OP_CONV (+BN, ReLU / SiLU, residual, squeeze-excite gate on the input, fused 2x2 pool),
// OP_DWCONV (+BN, SiLU), OP_SE, OP_MAXPOOL, OP_AVGPOOL — i.e. resnet18, efficientnet_b0 and the set encoder. (The
// opt-in fused MBConv front op of the inference plan has no training form.)
#include "extractor.h"

namespace orbit {

__global__ __launch_bounds__(256) void bn_eval_stats_kernel(const BNDev* __restrict__ descs,
                                                            const float* __restrict__ pool, float* __restrict__ mean,
                                                            float* __restrict__ invstd) {
    const BNDev d = descs[blockIdx.x];
    for (int c = blockIdx.y * 256 + threadIdx.x; c < d.C; c += gridDim.y * 256) {
        const float cb = d.conv_bias != (size_t)-1 ? pool[d.conv_bias + c] : 0.f;
        mean[d.fold_off + c] = pool[d.mean + c] - cb;  // xhat = (y_conv - mean') * invstd
        invstd[d.fold_off + c] = 1.0f / sqrtf(pool[d.var + c] + d.eps);
    }
}

// dst[0][fold_off + c] = running_mean, dst[1][fold_off + c] = running_var
__global__ __launch_bounds__(256) void bn_export_kernel(const BNDev* __restrict__ descs, const float* __restrict__ pool,
                                                        float* __restrict__ dst, size_t fold_floats) {
    const BNDev d = descs[blockIdx.x];
    for (int c = blockIdx.y * 256 + threadIdx.x; c < d.C; c += gridDim.y * 256) {
        dst[d.fold_off + c] = pool[d.mean + c];
        dst[fold_floats + d.fold_off + c] = pool[d.var + c];
    }
}

// running = (1 - momentum) * running + momentum * stat for every BatchNorm: the update an ORBIT_TRAIN_DEFER_RUNNING_STATS forward
// left undone (stat[0][fold_off + c] = batch mean (+ conv bias), stat[1][.] = unbiased batch variance; the expression is the one
// bn_stats_finalize_kernel evaluates)
__global__ __launch_bounds__(256) void bn_apply_deferred_kernel(const BNDev* __restrict__ descs, float* __restrict__ pool,
                                                                const float* __restrict__ stat, size_t fold_floats,
                                                                float momentum) {
    const BNDev d = descs[blockIdx.x];
    for (int c = blockIdx.y * 256 + threadIdx.x; c < d.C; c += gridDim.y * 256) {
        pool[d.mean + c] = (1.f - momentum) * pool[d.mean + c] + momentum * stat[d.fold_off + c];
        pool[d.var + c] = (1.f - momentum) * pool[d.var + c] + momentum * stat[fold_floats + d.fold_off + c];
    }
}

struct TapeLayout {
    // byte offsets per op ((size_t)-1: none). conv / depthwise: y raw output, a activation, p/idx fused pool, xg gated
    // input. squeeze-excite: p = pooled means [B][C], a = gate [B][C]. max-pool: p output, idx argmax.
    std::vector<size_t> y, a, p, idx, xg;
    size_t mean = 0, invstd = 0, scale = 0, shift = 0, partial = 0, pool = 0, total = 0;
    size_t rstat = 0;  // [2][fold_floats]: batch mean / unbiased variance of an ORBIT_TRAIN_DEFER_RUNNING_STATS forward
};

static const size_t NONE = (size_t)-1;

static bool plan_trainable(const orbit_extractor* fe) {
    for (const Op& o : fe->ops) {
        if (o.kind == OP_MBFRONT) return false;
        if ((o.kind == OP_CONV || o.kind == OP_DWCONV) && o.bn < 0) return false;
        if (o.kind == OP_CONV && o.use_gate && (o.pool2 || o.x_nchw)) return false;
    }
    return true;
}

static bool has_bn(const Op& o) { return o.kind == OP_CONV || o.kind == OP_DWCONV; }

// A batch-statistics conv whose activated output is read by ONE consumer, the depthwise conv that follows (EfficientNet's
// expansion convs and stem): that consumer applies the conv's BatchNorm + SiLU as it loads the RAW output (DwInXf in
// csrc/ops.hip), so the activated 6x-expanded tensor is never written - on forwards that run no backward (round 3) and, since
// round 5, on TAPED ones: the only other reader of the activation is the depthwise filter gradient, whose LDS form applies the
// same transform as it stages its input patch (dwconv_wgrad_lds_kernel, csrc/train_mbconv.hip; each element passes once). A
// first attempt with the transform inside the global-load filter-gradient kernel - every input loaded K times, SiLU per load -
// lost more there (1.76 -> 3.98 ms) than the forward gained (profiles/r05_lite_ab_taped_xf.txt). The activation must not be
// ReLU on a taped forward (its BatchNorm backward reads the mask from the activated tensor). Forward and backward evaluate
// this on the same plan and batch; the option is read by both (do not flip train_dw_xf between a forward and its backward).
static bool conv_feeds_dw_raw(const orbit_extractor* fe, size_t i, int bn_train, bool no_backward, int B) {
    const Op& o = fe->ops[i];
    if (o.kind != OP_CONV || !bn_train || !get_option("train_dw_xf") || o.pool2 || o.res >= 0 || o.Cout % 4 != 0) return false;
    if (i + 1 >= fe->ops.size() || fe->ops[i + 1].kind != OP_DWCONV || fe->ops[i + 1].in != o.out) return false;
    if (!no_backward) {
        const Op& d = fe->ops[i + 1];
        if (o.act != ORBIT_ACT_SILU && o.act != ORBIT_ACT_NONE) return false;
        if (!dwconv_wgrad_xf_supported(B, d.H, d.W, d.Cin, d.KH, d.stride, d.Ho, d.Wo)) return false;
    }
    for (size_t j = i + 2; j < fe->ops.size(); ++j) {  // no later reader of the buffer before it is written again
        const Op& q = fe->ops[j];
        if (q.in == o.out || q.res == o.out) return false;
        if (q.out == o.out) break;
    }
    return true;
}

// Round 6: on a NO-BACKWARD batch-statistics forward (the cache pass of the LITE step: 200 frames under torch.no_grad() while the
// extractor is being learned, reference few_shot_recognisers.py:404-408) an MBConv block's expansion conv + depthwise conv run
// as the row-streaming fused front of the inference plans in two sweeps: (1) the expansion conv as a STATISTICS SWEEP
// (ConvDesc::stats_only: same kernel, same tiles, nothing stored) gives the first BatchNorm's batch statistics, (2) the fused
// front (csrc/mbconv_rows.hip, RAW form) re-expands in its LDS ring with that scale / shift, stores the RAW depthwise outputs
// and their column sums - the second BatchNorm then proceeds as on the unfused path. The 6x-expanded tensor (963 MB per 200
// frames for block 1.0) is neither written nor read. Taped forwards keep the unfused pair: their backward reads that tensor.
// `train_fused_fronts`: 1 (default) = where measured faster (the 112x112 / 56x56 blocks), 0 = never, 2 = every supported shape,
// 3 = as 2 with the statistics sweep of the conv instead of the Gram-matrix statistics (parity tests: bit-identical first
// BatchNorm).
static bool fused_front_sweeps(const orbit_extractor* fe, size_t i, int bn_train, bool no_backward) {
    const Op& o = fe->ops[i];
    const int opt = get_option("train_fused_fronts");
    if (!opt || o.kind != OP_CONV || !bn_train || !no_backward || o.x_nchw || o.pool2 || o.res >= 0 || o.use_gate) return false;
    if (o.KH != 1 || o.KW != 1 || o.stride != 1 || o.act != ORBIT_ACT_SILU) return false;
    if (i + 1 >= fe->ops.size() || fe->ops[i + 1].kind != OP_DWCONV || fe->ops[i + 1].in != o.out) return false;
    const Op& d = fe->ops[i + 1];
    if (d.act != ORBIT_ACT_SILU || d.Cin != o.Cout || !mbconv_rows_supported(o.H, o.W, o.Cin, o.Cout, d.KH, d.stride)) return false;
    if (fe->bns[o.bn].conv_bias >= 0) return false;
    if (opt == 1 && o.H < 56) return false;  // (the 28x28 blocks: statistics sweep + front measured no faster than the pair)
    for (size_t j = i + 2; j < fe->ops.size(); ++j) {  // no later reader of the expanded tensor before its buffer is rewritten
        const Op& q = fe->ops[j];
        if (q.in == o.out || q.res == o.out) return false;
        if (q.out == o.out) break;
    }
    return true;
}

static size_t max_bn_partial_floats(const orbit_extractor* fe, int B) {
    size_t m = 4;
    for (const Op& o : fe->ops)
        if (has_bn(o)) {
            const size_t M = (size_t)B * o.Ho * o.Wo;
            m = std::max(m, (size_t)bn_reduce_blocks((int)M, o.Cout) * 2 * o.Cout + 3 * (size_t)o.Cout);
            // statistics partials written by the producing kernel's epilogue (one per 32+ conv rows / per depthwise chunk)
            if (o.kind == OP_CONV) {
                m = std::max(m, bn_partial_floats((M + 31) / 32, o.Cout));
                // backward: reduction written by the data gradient of the depthwise convolution it feeds (+ coefficients)
                for (int st : {1, 2})
                    m = std::max(m, bn_partial_floats((size_t)dwconv_dgrad_bn_blocks(B, o.Ho, o.Wo, o.Cout, st), o.Cout) +
                                        3 * (size_t)o.Cout);
            }
            else {
                m = std::max(m, bn_partial_floats((size_t)B * dwconv_se_chunks(o.Ho), o.Cout));
                // (the two-sweep fused front writes one row per frame and strip-band tile instead; its first sweep's Gram
                // partials of the block input live here too)
                m = std::max(m, bn_partial_floats((size_t)B * 16, o.Cout));
                m = std::max(m, bn_gram_scratch_floats(B * o.H * o.W, 24));
            }
            // backward of a depthwise BatchNorm whose reduction rides on the squeeze-excite backward (+ 3*C coefficients)
            if (o.kind == OP_DWCONV)
                m = std::max(m, bn_partial_floats((size_t)B * se_pool_chunks(B, o.Ho * o.Wo, o.Cout), o.Cout) + 3 * (size_t)o.Cout);
        }
    return m;
}

// squeeze-excite pooling partials of the pooled apply pass (launch_scale_shift_act_pool): [B][chunks][C]
static size_t max_se_pool_floats(const orbit_extractor* fe, int B) {
    size_t m = 4;
    int last_dw = -1;
    for (size_t i = 0; i < fe->ops.size(); ++i) {
        const Op& o = fe->ops[i];
        if (o.kind == OP_DWCONV) last_dw = (int)i;
        if (o.kind == OP_SE && last_dw >= 0) {
            const Op& dw = fe->ops[last_dw];
            m = std::max(m, (size_t)B * se_pool_chunks(B, dw.Ho * dw.Wo, o.Cin) * o.Cin);
        }
    }
    return m;
}

static TapeLayout tape_layout(const orbit_extractor* fe, int B) {
    TapeLayout L;
    const size_t n = fe->ops.size();
    L.y.assign(n, NONE), L.a.assign(n, NONE), L.p.assign(n, NONE), L.idx.assign(n, NONE), L.xg.assign(n, NONE);
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t o = off;
        off += align_up(bytes, 256);
        return o;
    };
    for (size_t i = 0; i < n; ++i) {
        const Op& o = fe->ops[i];
        if (o.kind == OP_CONV) {
            const size_t e = (size_t)B * o.Ho * o.Wo * o.Cout;
            L.y[i] = take(e * 4), L.a[i] = take(e * 4);
            if (o.pool2) {
                const size_t pe = (size_t)B * (o.Ho / 2) * (o.Wo / 2) * o.Cout;
                L.p[i] = take(pe * 4), L.idx[i] = take(pe);
            }
            // (a gated projection reads x and the gate separately - forward through the GATE prologue of conv_igemm, filter
            // gradient through conv_wgrad's - so the product x * gate is never materialised; L.xg stays unused)
        } else if (o.kind == OP_DWCONV) {
            const size_t e = (size_t)B * o.Ho * o.Wo * o.Cout;
            L.y[i] = take(e * 4), L.a[i] = take(e * 4);
        } else if (o.kind == OP_SE) {
            L.p[i] = take((size_t)B * o.Cin * 4), L.a[i] = take((size_t)B * o.Cin * 4);
        } else if (o.kind == OP_MAXPOOL) {
            const size_t pe = (size_t)B * o.Ho * o.Wo * o.Cout;
            L.p[i] = take(pe * 4), L.idx[i] = take(pe);
        }
    }
    L.mean = take(fe->fold_floats * 4), L.invstd = take(fe->fold_floats * 4);
    L.scale = take(fe->fold_floats * 4), L.shift = take(fe->fold_floats * 4);
    L.partial = take(max_bn_partial_floats(fe, B) * 4);
    L.pool = take(max_se_pool_floats(fe, B) * 4);
    L.rstat = take(2 * fe->fold_floats * 4);
    L.total = off;
    return L;
}

// which op produced the tensor an op reads (index into ops, -1 = the frames)
struct Producers {
    std::vector<int> in, res, gate;  // gate: the squeeze-excite op whose output a gated conv multiplies into its input
};
static Producers producers(const orbit_extractor* fe) {
    Producers P;
    std::map<int, int> last;  // buffer id -> op index
    for (size_t i = 0; i < fe->ops.size(); ++i) {
        const Op& o = fe->ops[i];
        P.in.push_back(o.in >= 0 && last.count(o.in) ? last[o.in] : -1);
        P.res.push_back(o.res >= 0 && last.count(o.res) ? last[o.res] : -1);
        P.gate.push_back(o.kind == OP_CONV && o.use_gate && last.count(102) ? last[102] : -1);
        last[o.out] = (int)i;
    }
    return P;
}

struct BwdLayout {
    static const int NSLOTS = 8;
    size_t slot_bytes = 0, slots = 0, up = 0, wgrad = 0, partial = 0, se = 0, total = 0;
    size_t cwgrad = 0;  // the dense filter gradients' partial tiles, one region per layer (their reduce runs batched at the end)
};
static BwdLayout bwd_layout(const orbit_extractor* fe, int B) {
    BwdLayout L;
    size_t slot = 4, up = 4, wg = 4, se = 4, cwg = 4;
    for (const Op& o : fe->ops) {
        if (o.kind == OP_DWCONV) {
            slot = std::max(slot, (size_t)B * o.H * o.W * o.Cin);
            slot = std::max(slot, (size_t)B * o.Ho * o.Wo * o.Cout);
            wg = std::max(wg, dwconv_wgrad_scratch_floats(B, o.Ho, o.Wo, o.Cin, o.KH));
            wg = std::max(wg, dwconv_bwd_fused_scratch_floats(B, o.H, o.W, o.Cin, o.KH, o.stride));
        } else if (o.kind == OP_SE) {
            se += align_up(se_bwd_scratch_floats(B, o.Cin, o.R), 64);  // one scratch per block (their parameter gradients run batched)
        } else if (o.kind == OP_CONV) {
            slot = std::max(slot, (size_t)B * o.Ho * o.Wo * o.Cout);
            if (!o.x_nchw) slot = std::max(slot, (size_t)B * o.H * o.W * o.Cin);
            if (o.stride > 1 && !o.x_nchw) up = std::max(up, (size_t)B * o.H * o.W * o.Cout);
            cwg += align_up(conv_wgrad_scratch_floats(B, o.Cin, o.Cout, o.KH, o.KW, o.Ho, o.Wo), 64);
        } else if (o.kind == OP_MAXPOOL || o.kind == OP_AVGPOOL) {
            slot = std::max(slot, (size_t)B * o.H * o.W * o.Cin);
        }
    }
    size_t off = 0;
    L.slot_bytes = align_up(slot * 4, 256);
    L.slots = off, off += L.slot_bytes * BwdLayout::NSLOTS;
    L.up = off, off += align_up(up * 4, 256);
    L.wgrad = off, off += align_up(wg * 4, 256);
    L.cwgrad = off, off += align_up(cwg * 4, 256);
    L.partial = off, off += align_up(max_bn_partial_floats(fe, B) * 4, 256);
    L.se = off, off += align_up(se * 4, 256);
    L.total = off;
    return L;
}

}  // namespace orbit

using namespace orbit;

// lazily built training-side device state of a plan
struct orbit_train_state {
    float* d_dgrad = nullptr;  // dgrad-packed filters
    std::vector<PackJob> jobs;  // their re-layouts, one launch for all layers (PackJob kind 4)
    PackJob* d_jobs = nullptr;
    std::vector<size_t> dgrad_off;
    size_t dgrad_floats = 0;
    bool packed = false;
};

static std::map<const orbit_extractor*, orbit_train_state>& train_states() {
    static std::map<const orbit_extractor*, orbit_train_state> m;
    return m;
}

namespace orbit {
void extractor_train_invalidate(const orbit_extractor* fe) {  // parameters changed: repack on next use
    auto it = train_states().find(fe);
    if (it != train_states().end()) it->second.packed = false;
}
void extractor_train_release(const orbit_extractor* fe) {
    auto it = train_states().find(fe);
    if (it == train_states().end()) return;
    (void)hipFree(it->second.d_dgrad);
    (void)hipFree(it->second.d_jobs);
    train_states().erase(it);
}
}  // namespace orbit

static int ensure_dgrad_filters(orbit_extractor* fe, orbit_train_state** out, hipStream_t s) {
    orbit_train_state& st = train_states()[fe];
    if (st.dgrad_off.empty()) {
        st.dgrad_off.assign(fe->ops.size(), NONE);
        size_t off = 0;
        for (size_t i = 0; i < fe->ops.size(); ++i) {
            const Op& o = fe->ops[i];
            if (o.kind != OP_CONV || o.x_nchw) continue;
            st.dgrad_off[i] = off;
            off += conv_dgrad_packed_floats(o.Cin, o.Cout, o.KH, o.KW);
        }
        st.dgrad_floats = std::max<size_t>(off, 4);
        ORBIT_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&st.d_dgrad), st.dgrad_floats * sizeof(float)));
    }
    if (!st.packed) {
        if (st.jobs.empty()) {
            for (size_t i = 0; i < fe->ops.size(); ++i) {
                const Op& o = fe->ops[i];
                if (st.dgrad_off[i] == NONE) continue;
                const ConvPackGeom g = conv_pack_geom(o.Cout, o.Cin, o.KH, o.KW, 0);  // the dgrad conv: Cin' = Cout, Cout' = Cin
                PackJob j;
                j.src = fe->d_pool + fe->params[o.weight].off, j.dst = st.d_dgrad + st.dgrad_off[i], j.kind = 4;
                j.Cin = o.Cin, j.Cout = o.Cout, j.KH = o.KH, j.KW = o.KW, j.cin_pad = g.cin_pad, j.KT = g.kt;
                j.cout_pad = g.cout_pad, j.total = (unsigned)((size_t)g.cout_pad * g.kt);
                st.jobs.push_back(j);
            }
        }
        if (int rc = run_pack_jobs(st.jobs, &st.d_jobs, s)) return rc;
        st.packed = true;
    }
    *out = &st;
    return ORBIT_OK;
}

static int train_forward_run(orbit_extractor_t* fe, const float* frames, int B, const float* film_gamma,
                             const float* film_beta, int bn_train, float momentum, float* feats, void* tape,
                             hipStream_t s, bool no_backward = false, bool defer_stats = false);
static int backward_run(orbit_extractor_t* fe, orbit_train_state* st, const float* frames, int B, const float* film_gamma,
                        const float* film_beta, int bn_train, const float* dfeats, const void* tape, float* param_grads,
                        int filter_grads, float* dfilm_gamma, float* dfilm_beta, void* workspace, hipStream_t s);

extern "C" {

int orbit_extractor_supports_training(const orbit_extractor_t* fe) { return fe && plan_trainable(fe) ? 1 : 0; }

size_t orbit_extractor_tape_bytes(const orbit_extractor_t* fe, int B) {
    if (!fe || B <= 0 || !plan_trainable(fe)) return 0;
    return tape_layout(fe, B).total;
}

size_t orbit_extractor_backward_workspace_bytes(const orbit_extractor_t* fe, int B) {
    if (!fe || B <= 0 || !plan_trainable(fe)) return 0;
    return bwd_layout(fe, B).total;
}

size_t orbit_extractor_grad_floats(const orbit_extractor_t* fe) { return fe ? fe->pool_floats : 0; }
size_t orbit_extractor_param_offset(const orbit_extractor_t* fe, int i) {
    return (fe && i >= 0 && i < (int)fe->params.size()) ? fe->params[i].off : 0;
}
size_t orbit_extractor_bn_stat_floats(const orbit_extractor_t* fe) { return fe ? fe->fold_floats : 0; }

int orbit_extractor_export_bn_stats(orbit_extractor_t* fe, float* dst, orbit_stream_t stream) {
    ORBIT_REQUIRE(fe && dst, "extractor_export_bn_stats: null pointer");
    if (!fe->finalized) return set_err(ORBIT_ERR_STATE, "extractor_export_bn_stats: plan is not finalized");
    bn_export_kernel<<<dim3((unsigned)fe->bns.size(), 2), 256, 0, (hipStream_t)stream>>>(fe->d_bn, fe->d_pool, dst,
                                                                                        fe->fold_floats);
    ORBIT_LAUNCH_CHECK();
    return ORBIT_OK;
}

int orbit_extractor_train_forward(orbit_extractor_t* fe, const float* frames, int B, const float* film_gamma,
                                  const float* film_beta, int bn_train, float momentum, float* feats, void* tape,
                                  size_t tape_bytes, orbit_stream_t stream) {
    return orbit_extractor_train_forward_ex(fe, frames, B, film_gamma, film_beta, bn_train, momentum, feats, tape, tape_bytes,
                                            0, stream);
}

int orbit_extractor_train_forward_ex(orbit_extractor_t* fe, const float* frames, int B, const float* film_gamma,
                                     const float* film_beta, int bn_train, float momentum, float* feats, void* tape,
                                     size_t tape_bytes, int flags, orbit_stream_t stream) {
    ORBIT_REQUIRE(fe && frames && feats && tape, "extractor_train_forward: null pointer");
    ORBIT_REQUIRE((flags & ~(ORBIT_TRAIN_NO_BACKWARD | ORBIT_TRAIN_DEFER_RUNNING_STATS)) == 0,
                  "extractor_train_forward: unknown flags %d", flags);
    const bool no_backward = (flags & ORBIT_TRAIN_NO_BACKWARD) != 0;
    const bool defer_stats = (flags & ORBIT_TRAIN_DEFER_RUNNING_STATS) != 0;
    ORBIT_REQUIRE(B > 0, "extractor_train_forward: empty batch");
    if (!fe->finalized) return set_err(ORBIT_ERR_STATE, "extractor_train_forward: call orbit_extractor_finalize first");
    if (!plan_trainable(fe))
        return set_err(ORBIT_ERR_STATE, "extractor_train_forward: the %s plan has no training path yet", fe->name.c_str());
    ORBIT_REQUIRE((film_gamma == nullptr) == (film_beta == nullptr),
                  "extractor_train_forward: film_gamma and film_beta must be given together");
    const TapeLayout L = tape_layout(fe, B);
    ORBIT_REQUIRE(tape_bytes >= L.total, "extractor_train_forward: tape too small (%zu < %zu bytes)", tape_bytes, L.total);
    ORBIT_REQUIRE(((uintptr_t)tape & 255) == 0, "extractor_train_forward: tape must be 256-byte aligned");
    orbit_extractor::TrainGraphKey key;
    memset(&key, 0, sizeof(key));
    key.p[0] = frames, key.p[1] = film_gamma, key.p[2] = film_beta, key.p[3] = feats, key.p[4] = tape;
    key.v[0] = (no_backward ? 3 : 1) /* forward */ + (defer_stats ? 16 : 0), key.v[1] = B, key.v[2] = bn_train;
    memcpy(&key.v[3], &momentum, sizeof(float));
    return fe->run_train_graphed(key, (hipStream_t)stream, [&](hipStream_t s) {
        return train_forward_run(fe, frames, B, film_gamma, film_beta, bn_train, momentum, feats, tape, s, no_backward,
                                 defer_stats);
    });
}

int orbit_extractor_apply_deferred_bn_stats(orbit_extractor_t* fe, const void* tape, size_t tape_bytes, int B, float momentum,
                                            orbit_stream_t stream) {
    ORBIT_REQUIRE(fe && tape && B > 0, "extractor_apply_deferred_bn_stats: null pointer or empty batch");
    if (!fe->finalized) return set_err(ORBIT_ERR_STATE, "extractor_apply_deferred_bn_stats: plan is not finalized");
    const TapeLayout L = tape_layout(fe, B);
    ORBIT_REQUIRE(tape_bytes >= L.total, "extractor_apply_deferred_bn_stats: tape too small (%zu < %zu bytes)", tape_bytes, L.total);
    bn_apply_deferred_kernel<<<dim3((unsigned)fe->bns.size(), 2), 256, 0, (hipStream_t)stream>>>(
        fe->d_bn, fe->d_pool, reinterpret_cast<const float*>(static_cast<const char*>(tape) + L.rstat), fe->fold_floats, momentum);
    ORBIT_LAUNCH_CHECK();
    // the static fold follows the running statistics this call just changed (the deferred forward skipped its own re-fold)
    bn_fold_all_kernel<<<dim3((unsigned)fe->bns.size(), 2), 256, 0, (hipStream_t)stream>>>(
        fe->d_bn, fe->d_pool, nullptr, nullptr, fe->d_fold, fe->d_fold + fe->fold_floats);
    ORBIT_LAUNCH_CHECK();
    return ORBIT_OK;
}

}  // extern "C"

static int train_forward_run(orbit_extractor_t* fe, const float* frames, int B, const float* film_gamma,
                             const float* film_beta, int bn_train, float momentum, float* feats, void* tape,
                             hipStream_t s, bool no_backward, bool defer_stats) {
    const TapeLayout L = tape_layout(fe, B);
    char* tp = static_cast<char*>(tape);
    auto fl = [&](size_t off) { return reinterpret_cast<float*>(tp + off); };
    float *mean = fl(L.mean), *invstd = fl(L.invstd), *scale = fl(L.scale), *shift = fl(L.shift);
    // deferred running statistics: the finalize kernels run their update with momentum 1 (the statistics replace the slot's
    // content, which is never read) on a slot of the tape instead of on the plan's running statistics;
    // orbit_extractor_apply_deferred_bn_stats applies the real update later - so this forward may overlap, on another stream,
    // with a forward that updates them
    defer_stats = defer_stats && bn_train;
    if (defer_stats) momentum = 1.0f;
    auto run_mean = [&](const BNDesc& bn) {
        return defer_stats ? fl(L.rstat) + bn.fold_off : fe->d_pool + fe->params[bn.mean].off;
    };
    auto run_var = [&](const BNDesc& bn) {
        return defer_stats ? fl(L.rstat) + fe->fold_floats + bn.fold_off : fe->d_pool + fe->params[bn.var].off;
    };
    const bool film = film_gamma && fe->film_size > 0;
    if (!bn_train) {
        dim3 grid((unsigned)fe->bns.size(), 2);
        bn_fold_all_kernel<<<grid, 256, 0, s>>>(fe->d_bn, fe->d_pool, film ? film_gamma : nullptr,
                                                film ? film_beta : nullptr, scale, shift);
        ORBIT_LAUNCH_CHECK();
        bn_eval_stats_kernel<<<grid, 256, 0, s>>>(fe->d_bn, fe->d_pool, mean, invstd);
        ORBIT_LAUNCH_CHECK();
    }
    std::map<int, const float*> cur;  // buffer id -> tensor currently held
    cur[-1] = frames;
    int last_dw = -1;
    bool dw_pooled = false;  // the last depthwise op's activation pass left pooling partials in L.pool
    bool dw_in_raw = false;  // the tensor the next depthwise op reads is a RAW conv output (see ORBIT_TRAIN_NO_BACKWARD)
    int dw_in_bn = -1, dw_in_act = ORBIT_ACT_NONE;
    int front_conv = -1;  // the expansion conv whose statistics sweep just ran: the next depthwise op is the fused front
    for (size_t i = 0; i < fe->ops.size(); ++i) {
        const Op& o = fe->ops[i];
        int rc = ORBIT_OK;
        if (o.kind == OP_CONV) {
            const BNDesc& bn = fe->bns[o.bn];
            const float* xin = cur[o.in];
            const bool sweep = fused_front_sweeps(fe, i, bn_train, no_backward);
            ConvDesc d;
            d.x = xin, d.w_packed = fe->d_packed + o.packed_off, d.y = fl(L.y[i]);
            d.scale = d.shift = d.residual = d.gate = nullptr;
            // squeeze-excite: the projection multiplies the gate into its A operand on the fly (as the inference plan does)
            if (o.use_gate) d.gate = cur[102];
            int stat_blocks = 0;  // train-mode BatchNorm: column sums of the raw outputs come from the conv's epilogue
            if (bn_train) d.stats = fl(L.partial), d.stats_blocks = &stat_blocks;
            d.B = B, d.H = o.H, d.W = o.W, d.Cin = o.Cin, d.Cout = o.Cout, d.KH = o.KH, d.KW = o.KW;
            d.stride = o.stride, d.pad_t = o.pad_t, d.pad_l = o.pad_l, d.Ho = o.Ho, d.Wo = o.Wo;
            d.act = ORBIT_ACT_NONE, d.pool2 = 0, d.x_nchw = o.x_nchw;
            // running-statistics BatchNorm (frozen extractor: CNAPs meta-training, FiLM fine-tuning): scale / shift are known
            // before the conv runs, so its epilogue writes BOTH the raw output (tape: xhat and the SiLU derivative need it)
            // and the activation - no separate activation pass over the tensor
            const bool dual = !bn_train && !o.pool2;
            if (dual) {
                d.y_raw = fl(L.y[i]), d.y = fl(L.a[i]);
                d.scale = scale + bn.fold_off, d.shift = shift + bn.fold_off;
                d.residual = o.res >= 0 ? cur[o.res] : nullptr;
                d.act = o.act;
            }
            if (sweep && bn_train && bn_gram_supported(o.Cin) && get_option("train_fused_fronts") != 3) {
                // sweep 1 without the conv: the first BatchNorm's batch statistics from the Gram matrix of the block's input
                // (csrc/train_ops.hip launch_bn_stats_from_gram; option value 3 keeps the statistics sweep of the conv itself)
                const bool fm = film && bn.film_off >= 0;
                rc = launch_bn_stats_from_gram(xin, B * o.H * o.W, o.Cin, fe->d_pool + fe->params[o.weight].off, o.Cout, bn.eps,
                                               momentum, fm ? film_gamma + bn.film_off : fe->d_pool + fe->params[bn.gamma].off,
                                               fm ? film_beta + bn.film_off : fe->d_pool + fe->params[bn.beta].off,
                                               mean + bn.fold_off, invstd + bn.fold_off, scale + bn.fold_off, shift + bn.fold_off,
                                               run_mean(bn), run_var(bn), fl(L.partial), s);
                if (rc != ORBIT_OK) return rc;
                front_conv = (int)i;
                cur[o.out] = nullptr;
                continue;
            }
            if (sweep) d.y = nullptr, d.stats_only = true;  // statistics only: the fused front below re-expands in LDS
            rc = launch_conv(d, s);
            if (rc != ORBIT_OK) return rc;
            if (dual) {
                cur[o.out] = fl(L.a[i]);
                continue;
            }
            if (sweep && stat_blocks <= 0)
                return set_err(ORBIT_ERR_STATE, "extractor_train_forward: the statistics sweep of op %zu emitted no partials", i);
            const int M = B * o.Ho * o.Wo;
            if (bn_train) {
                const bool fm = film && bn.film_off >= 0;
                const float* gam = fm ? film_gamma + bn.film_off : fe->d_pool + fe->params[bn.gamma].off;
                const float* bet = fm ? film_beta + bn.film_off : fe->d_pool + fe->params[bn.beta].off;
                const float* cb = bn.conv_bias >= 0 ? fe->d_pool + fe->params[bn.conv_bias].off : nullptr;
                if (stat_blocks > 0)
                    rc = launch_bn_stats_from_partials(fl(L.partial), stat_blocks, M, o.Cout, bn.eps, momentum, gam, bet, cb,
                                                       mean + bn.fold_off, invstd + bn.fold_off, scale + bn.fold_off,
                                                       shift + bn.fold_off, run_mean(bn), run_var(bn), s);
                else  // (fused pooling / split-K launches do not emit them: the statistics pass of its own)
                    rc = launch_bn_stats(d.y, M, o.Cout, bn.eps, momentum, gam, bet, cb, mean + bn.fold_off,
                                         invstd + bn.fold_off, scale + bn.fold_off, shift + bn.fold_off,
                                         run_mean(bn), run_var(bn), fl(L.partial), s);
                if (rc != ORBIT_OK) return rc;
            }
            if (sweep) {
                front_conv = (int)i;
                cur[o.out] = nullptr;
                continue;
            }
            // the only consumer is the depthwise conv that follows: it applies this BatchNorm + activation as it loads the raw
            // output (conv_feeds_dw_raw above), so the activated 6x-expanded tensor is neither written nor read back
            dw_in_raw = conv_feeds_dw_raw(fe, i, bn_train, no_backward, B);
            if (dw_in_raw) {
                cur[o.out] = d.y;
                dw_in_bn = o.bn, dw_in_act = o.act;
                continue;
            }
            rc = launch_scale_shift_act(d.y, scale + bn.fold_off, shift + bn.fold_off,
                                        o.res >= 0 ? cur[o.res] : nullptr, o.act, (size_t)M, o.Cout, fl(L.a[i]), s);
            if (rc != ORBIT_OK) return rc;
            if (o.pool2) {
                rc = launch_maxpool_idx(fl(L.a[i]), fl(L.p[i]), reinterpret_cast<uint8_t*>(tp + L.idx[i]), B, o.Ho, o.Wo,
                                        o.Cout, 2, 2, 0, o.Ho / 2, o.Wo / 2, s);
                cur[o.out] = fl(L.p[i]);
            } else {
                cur[o.out] = fl(L.a[i]);
            }
        } else if (o.kind == OP_DWCONV) {
            const BNDesc& bn = fe->bns[o.bn];
            float* y = fl(L.y[i]);
            // train-mode BatchNorm: the depthwise kernel itself emits the column sums / sums of squares of its raw outputs
            const float* in_sc = dw_in_raw ? scale + fe->bns[dw_in_bn].fold_off : nullptr;
            const float* in_sh = dw_in_raw ? shift + fe->bns[dw_in_bn].fold_off : nullptr;
            int stat_rows = B * dwconv_se_chunks(o.Ho);
            if (front_conv >= 0) {
                // sweep 2 of the fused front: expand (first BatchNorm's batch statistics from the sweep above) + depthwise in
                // one row-streaming kernel, RAW depthwise outputs + their column sums out
                const Op& c = fe->ops[front_conv];
                const BNDesc& bn1 = fe->bns[c.bn];
                const int tiles = mbconv_rows_tiles(c.H, c.W, c.Cin, c.Cout, o.KH, o.stride);
                if (tiles <= 0 || tiles > 16) return set_err(ORBIT_ERR_STATE, "extractor_train_forward: fused front tiling");
                rc = launch_mbconv_rows(cur[c.in], fe->d_pool + fe->params[c.weight].off, scale + bn1.fold_off,
                                        shift + bn1.fold_off, fe->d_packed + o.packed_off, nullptr, nullptr, y, fl(L.partial), B,
                                        c.H, c.W, c.Cin, c.Cout, o.KH, o.stride, o.pad_t, o.pad_l, o.Ho, o.Wo, s, 0, true);
                stat_rows = B * tiles;
                front_conv = -1;
            } else {
                rc = launch_dwconv_se(cur[o.in], fe->d_packed + o.packed_off, y, nullptr, nullptr,
                                      bn_train ? fl(L.partial) : nullptr, B, o.H, o.W, o.Cin, o.KH, o.stride, o.pad_t, o.pad_l,
                                      o.Ho, o.Wo, ORBIT_ACT_NONE, s, bn_train ? 1 : 0, in_sc, in_sh, dw_in_act);
            }
            dw_in_raw = false;
            if (rc != ORBIT_OK) return rc;
            const int M = B * o.Ho * o.Wo;
            if (bn_train) {
                const bool fm = film && bn.film_off >= 0;
                rc = launch_bn_stats_from_partials(fl(L.partial), stat_rows, M, o.Cout, bn.eps, momentum,
                                                   fm ? film_gamma + bn.film_off : fe->d_pool + fe->params[bn.gamma].off,
                                                   fm ? film_beta + bn.film_off : fe->d_pool + fe->params[bn.beta].off,
                                                   nullptr, mean + bn.fold_off, invstd + bn.fold_off, scale + bn.fold_off,
                                                   shift + bn.fold_off, run_mean(bn), run_var(bn), s);
                if (rc != ORBIT_OK) return rc;
            }
            // the activation pass also produces the squeeze-excite pooling partials when an SE op consumes this tensor
            dw_pooled = i + 1 < fe->ops.size() && fe->ops[i + 1].kind == OP_SE;
            if (dw_pooled)
                rc = launch_scale_shift_act_pool(y, scale + bn.fold_off, shift + bn.fold_off, o.act, B, o.Ho * o.Wo, o.Cout,
                                                 fl(L.a[i]), fl(L.pool), s);
            else
                rc = launch_scale_shift_act(y, scale + bn.fold_off, shift + bn.fold_off, nullptr, o.act, (size_t)M, o.Cout,
                                            fl(L.a[i]), s);
            cur[o.out] = fl(L.a[i]);
            last_dw = (int)i;
        } else if (o.kind == OP_SE) {
            // squeeze (mean over the depthwise output) + excite MLP -> gate [B][C]
            if (last_dw < 0) return set_err(ORBIT_ERR_STATE, "extractor_train_forward: squeeze-excite without a producer");
            const Op& dw = fe->ops[last_dw];
            if (dw_pooled) {
                // the inference gate kernel on the partial sums the activation pass left; the pooled means it forms go on the
                // tape (the gate MLP's backward reads them)
                rc = launch_se_gate2(fl(L.pool), se_pool_chunks(B, dw.Ho * dw.Wo, o.Cin), dw.Ho * dw.Wo,
                                     fe->d_pool + fe->params[o.se_w1].off, fe->d_pool + fe->params[o.se_b1].off,
                                     fe->d_packed + o.packed_off, fe->d_pool + fe->params[o.se_b2].off, fl(L.a[i]), B, o.Cin,
                                     o.R, s, fl(L.p[i]));
            } else {
                rc = launch_colmean(fl(L.a[last_dw]), fl(L.p[i]), B, dw.Ho * dw.Wo, o.Cin, s);
                if (rc != ORBIT_OK) return rc;
                // the inference gate kernel, fed with the means as a single "partial sum" over one element
                rc = launch_se_gate2(fl(L.p[i]), 1, 1, fe->d_pool + fe->params[o.se_w1].off,
                                     fe->d_pool + fe->params[o.se_b1].off, fe->d_packed + o.packed_off,
                                     fe->d_pool + fe->params[o.se_b2].off, fl(L.a[i]), B, o.Cin, o.R, s);
            }
            cur[102] = fl(L.a[i]);
        } else if (o.kind == OP_MAXPOOL) {
            rc = launch_maxpool_idx(cur[o.in], fl(L.p[i]), reinterpret_cast<uint8_t*>(tp + L.idx[i]), B, o.H, o.W, o.Cin,
                                    o.pool_k, o.stride, o.pool_pad, o.Ho, o.Wo, s);
            cur[o.out] = fl(L.p[i]);
        } else {  // OP_AVGPOOL
            rc = launch_avgpool(cur[o.in], feats, B, o.H * o.W, o.Cin, s);
        }
        if (rc != ORBIT_OK) return rc;
    }
    if (bn_train && !defer_stats) {
        // the forward plans fold the running statistics at finalize time: refresh the static fold and drop graphs. (A forward
        // with deferred running statistics did not change them and may run beside a forward that does: it must neither read
        // d_pool's statistics nor write the plan-owned d_fold here - orbit_extractor_apply_deferred_bn_stats re-folds after
        // the join)
        dim3 grid((unsigned)fe->bns.size(), 2);
        bn_fold_all_kernel<<<grid, 256, 0, s>>>(fe->d_bn, fe->d_pool, nullptr, nullptr, fe->d_fold,
                                                fe->d_fold + fe->fold_floats);
        ORBIT_LAUNCH_CHECK();
    }
    return ORBIT_OK;
}

extern "C" {

int orbit_extractor_backward(orbit_extractor_t* fe, const float* frames, int B, const float* film_gamma,
                             const float* film_beta, int bn_train, const float* dfeats, const void* tape,
                             size_t tape_bytes, float* param_grads, int filter_grads, float* dfilm_gamma,
                             float* dfilm_beta, void* workspace, size_t workspace_bytes, orbit_stream_t stream) {
    ORBIT_REQUIRE(fe && frames && dfeats && tape && workspace, "extractor_backward: null pointer");
    ORBIT_REQUIRE(B > 0, "extractor_backward: empty batch");
    if (!fe->finalized) return set_err(ORBIT_ERR_STATE, "extractor_backward: call orbit_extractor_finalize first");
    if (!plan_trainable(fe))
        return set_err(ORBIT_ERR_STATE, "extractor_backward: the %s plan has no training path yet", fe->name.c_str());
    ORBIT_REQUIRE((dfilm_gamma == nullptr) == (dfilm_beta == nullptr),
                  "extractor_backward: dfilm_gamma and dfilm_beta must be given together");
    const bool film = film_gamma && film_beta && fe->film_size > 0;
    ORBIT_REQUIRE(!dfilm_gamma || film, "extractor_backward: FiLM gradients requested without FiLM inputs");
    if (!param_grads && !dfilm_gamma) return ORBIT_OK;  // nothing to compute
    const TapeLayout L = tape_layout(fe, B);
    const BwdLayout W = bwd_layout(fe, B);
    ORBIT_REQUIRE(tape_bytes >= L.total, "extractor_backward: tape too small");
    ORBIT_REQUIRE(workspace_bytes >= W.total, "extractor_backward: workspace too small (%zu < %zu bytes)", workspace_bytes,
                  W.total);
    ORBIT_REQUIRE(((uintptr_t)workspace & 255) == 0 && ((uintptr_t)tape & 255) == 0,
                  "extractor_backward: tape and workspace must be 256-byte aligned");
    // the dgrad-packed filters follow the parameters: repacked OUTSIDE any captured graph whenever they changed
    orbit_train_state* st = nullptr;
    if (int rc = ensure_dgrad_filters(fe, &st, (hipStream_t)stream)) return rc;
    orbit_extractor::TrainGraphKey key;
    memset(&key, 0, sizeof(key));
    key.p[0] = frames, key.p[1] = film_gamma, key.p[2] = film_beta, key.p[3] = dfeats, key.p[4] = tape;
    key.p[5] = param_grads, key.p[6] = dfilm_gamma, key.p[7] = dfilm_beta, key.p[8] = workspace;
    key.v[0] = 2 /* backward */, key.v[1] = B, key.v[2] = bn_train, key.v[3] = filter_grads;
    return fe->run_train_graphed(key, (hipStream_t)stream, [&](hipStream_t s) {
        return backward_run(fe, st, frames, B, film_gamma, film_beta, bn_train, dfeats, tape, param_grads, filter_grads,
                            dfilm_gamma, dfilm_beta, workspace, s);
    });
}

}  // extern "C"

static int backward_run(orbit_extractor_t* fe, orbit_train_state* st, const float* frames, int B, const float* film_gamma,
                        const float* film_beta, int bn_train, const float* dfeats, const void* tape, float* param_grads,
                        int filter_grads, float* dfilm_gamma, float* dfilm_beta, void* workspace, hipStream_t s) {
    const bool film = film_gamma && film_beta && fe->film_size > 0;
    // filter_grads == 0: only the BatchNorm weight / bias gradients are wanted (FiLM fine-tuning of a frozen extractor,
    // few_shot_recognisers.py:196-199): skip every filter / squeeze-excite / bias gradient
    const bool wg = param_grads != nullptr && filter_grads != 0;
    const TapeLayout L = tape_layout(fe, B);
    const BwdLayout W = bwd_layout(fe, B);
    const char* tp = static_cast<const char*>(tape);
    auto tf = [&](size_t off) { return reinterpret_cast<const float*>(tp + off); };
    char* ws = static_cast<char*>(workspace);
    float* up = reinterpret_cast<float*>(ws + W.up);
    float* wgrad_scratch = reinterpret_cast<float*>(ws + W.wgrad);
    float* partial = reinterpret_cast<float*>(ws + W.partial);
    float* cwgrad_base = reinterpret_cast<float*>(ws + W.cwgrad);
    size_t cwgrad_used = 0;
    WgradReduceJobs wg_jobs;
    int n_wg_jobs = 0;
    float* se_scratch_base = reinterpret_cast<float*>(ws + W.se);
    size_t se_scratch_used = 0;
    SeParamJobs se_jobs;
    int n_se_jobs = 0;
    const float *mean = tf(L.mean), *invstd = tf(L.invstd), *scale = tf(L.scale), *shift = tf(L.shift);

    const Producers P = producers(fe);
    const int n = (int)fe->ops.size();
    // the earliest op whose backward yields something that was asked for
    int first_needed = n;
    for (int i = 0; i < n && first_needed == n; ++i) {
        const Op& o = fe->ops[i];
        if (!has_bn(o)) continue;
        if (param_grads || (dfilm_gamma && fe->bns[o.bn].film_off >= 0)) first_needed = i;
    }

    bool used[BwdLayout::NSLOTS] = {false};
    std::vector<int> grad_slot(n, -1);  // gradient w.r.t. the OUTPUT tensor of op i
    // > 0: the slot of op i already holds g = d out * act'(.) and `partial` the [n][2][C] sums of its BatchNorm backward
    // (written by the squeeze-excite backward of the block, launch_se_gate_backward with SeBnFuse, or - for the expansion
    // convolution - by the data gradient of the depthwise convolution it feeds, DwBnBwd)
    std::vector<int> pre_reduced(n, 0);
    // ... and when the squeeze-excite backward left only the SUMS: the slot of op i holds d(x * gate) and g is rebuilt in the
    // BatchNorm's apply pass from it, the gate and the pooled-branch gradient (launch_bn_backward_reduced_gated)
    struct GatedG {
        const float *gate = nullptr, *dpooled = nullptr;
        int HW = 0;
    };
    std::vector<GatedG> gated(n);
    auto slot_ptr = [&](int k) { return reinterpret_cast<float*>(ws + W.slots + (size_t)k * W.slot_bytes); };
    auto alloc = [&]() {
        for (int k = 0; k < BwdLayout::NSLOTS; ++k)
            if (!used[k]) {
                used[k] = true;
                return k;
            }
        return -1;
    };
    auto release = [&](int k) {
        if (k >= 0) used[k] = false;
    };
    auto out_tensor = [&](int i) -> const float* {  // forward output of op i
        if (i < 0) return frames;
        const Op& o = fe->ops[i];
        if (o.kind == OP_CONV) return o.pool2 ? tf(L.p[i]) : tf(L.a[i]);
        if (o.kind == OP_DWCONV) return tf(L.a[i]);
        return tf(L.p[i]);
    };
#define SLOT_OR_FAIL(var)                                                                      \
    const int var = alloc();                                                                   \
    if (var < 0) return set_err(ORBIT_ERR_STATE, "extractor_backward: gradient slots exhausted")

    for (int i = n - 1; i >= first_needed; --i) {
        const Op& o = fe->ops[i];
        int rc = ORBIT_OK;
        if (o.kind == OP_AVGPOOL) {
            const int src = P.in[i];
            if (src < first_needed) continue;
            ORBIT_REQUIRE(grad_slot[src] < 0, "extractor_backward: unexpected fan-out into the global pool");
            SLOT_OR_FAIL(k);
            rc = launch_avgpool_bwd(dfeats, slot_ptr(k), B, o.H * o.W, o.Cin, s);
            grad_slot[src] = k;
        } else if (o.kind == OP_MAXPOOL) {
            const int src = P.in[i];
            const int g = grad_slot[i];
            if (g < 0) continue;
            if (src >= first_needed) {
                ORBIT_REQUIRE(grad_slot[src] < 0, "extractor_backward: unexpected fan-out into a max-pool");
                SLOT_OR_FAIL(k);
                rc = launch_maxpool_bwd(slot_ptr(g), reinterpret_cast<const uint8_t*>(tp + L.idx[i]), slot_ptr(k), B, o.H,
                                        o.W, o.Cin, o.pool_k, o.stride, o.pool_pad, o.Ho, o.Wo, s);
                grad_slot[src] = k;
            }
            release(g), grad_slot[i] = -1;
        } else if (o.kind == OP_SE) {
            continue;  // handled together with the gated convolution that consumes the gate
        } else if (o.kind == OP_DWCONV) {
            const int g = grad_slot[i];
            if (g < 0) continue;
            const BNDesc& bn = fe->bns[o.bn];
            const int M = B * o.Ho * o.Wo;
            const bool fm = film && bn.film_off >= 0;
            const float* gamma = fm ? film_gamma + bn.film_off : fe->d_pool + fe->params[bn.gamma].off;
            float *dgam = nullptr, *dbet = nullptr;
            if (fm) {
                if (dfilm_gamma) dgam = dfilm_gamma + bn.film_off, dbet = dfilm_beta + bn.film_off;
            } else if (param_grads) {
                dgam = param_grads + fe->params[bn.gamma].off, dbet = param_grads + fe->params[bn.beta].off;
            }
            const int src = P.in[i];
            const bool need_dx = src >= first_needed;
            const bool need_dy = need_dx || wg;
            int kdy = -1;
            if (need_dy) {
                kdy = alloc();
                if (kdy < 0) return set_err(ORBIT_ERR_STATE, "extractor_backward: gradient slots exhausted");
            }
            if (pre_reduced[i] > 0 && gated[i].gate) {
                float* coef = partial + bn_partial_floats((size_t)pre_reduced[i], o.Cout);
                rc = launch_bn_backward_reduced_gated(slot_ptr(g), gated[i].gate, gated[i].dpooled, gated[i].HW, tf(L.y[i]),
                                                      mean + bn.fold_off, invstd + bn.fold_off, scale + bn.fold_off,
                                                      shift + bn.fold_off, o.act, gamma, bn_train, M, o.Cout,
                                                      need_dy ? slot_ptr(kdy) : nullptr, dgam, dbet, partial, pre_reduced[i],
                                                      coef, s);
            } else if (pre_reduced[i] > 0) {
                float* coef = partial + bn_partial_floats((size_t)pre_reduced[i], o.Cout);
                rc = launch_bn_backward_reduced(slot_ptr(g), tf(L.y[i]), mean + bn.fold_off, invstd + bn.fold_off, gamma,
                                                bn_train, M, o.Cout, need_dy ? slot_ptr(kdy) : nullptr, dgam, dbet, partial,
                                                pre_reduced[i], coef, s);
            } else {
                float* coef = partial + (size_t)bn_reduce_blocks(M, o.Cout) * 2 * o.Cout;
                rc = launch_bn_backward(slot_ptr(g), tf(L.a[i]), tf(L.y[i]), mean + bn.fold_off, invstd + bn.fold_off, gamma,
                                        scale + bn.fold_off, shift + bn.fold_off, bn_train, o.act, M, o.Cout,
                                        need_dy ? slot_ptr(kdy) : nullptr, nullptr, 0, dgam, dbet, nullptr, partial, coef, s);
            }
            if (rc != ORBIT_OK) return rc;
            release(g), grad_slot[i] = -1;
            // the producer is a convolution + BatchNorm + SiLU whose only reader is this layer (see the data gradient below)
            const Op& po = fe->ops[src >= 0 ? src : 0];
            const bool through_act = need_dx && src == i - 1 && po.kind == OP_CONV && bn_train && !po.pool2 && po.res < 0 &&
                                     po.Cout % 4 == 0 && fe->bns[po.bn].conv_bias < 0 &&
                                     (po.act == ORBIT_ACT_SILU || po.act == ORBIT_ACT_NONE) && get_option("train_dw_xf");
            const bool raw_fed = src >= 0 && conv_feeds_dw_raw(fe, (size_t)src, bn_train, false, B);
            // ... and on the stride-2 layers and the large 3x3 maps the filter gradient rides on that data-gradient kernel
            // (DwBnBwd::wgrad_partial)
            const bool wg_fused = wg && through_act && raw_fed &&
                                  dwconv_bwd_fused_scratch_floats(B, o.H, o.W, o.Cin, o.KH, o.stride) > 0;
            auto filter_gradient = [&]() {
                if (raw_fed) {
                    // the forward never wrote this layer's input: the filter gradient rebuilds it from the producing conv's raw
                    // output as it stages its patch
                    const BNDesc& sbn = fe->bns[fe->ops[src].bn];
                    return launch_dwconv_wgrad(tf(L.y[src]), slot_ptr(kdy), param_grads + fe->params[o.weight].off,
                                               wgrad_scratch, B, o.H, o.W, o.Cin, o.KH, o.stride, o.pad_t, o.pad_l, o.Ho, o.Wo, s,
                                               scale + sbn.fold_off, shift + sbn.fold_off, fe->ops[src].act);
                }
                return launch_dwconv_wgrad(out_tensor(src), slot_ptr(kdy), param_grads + fe->params[o.weight].off,
                                           wgrad_scratch, B, o.H, o.W, o.Cin, o.KH, o.stride, o.pad_t, o.pad_l, o.Ho, o.Wo, s);
            };
            if (wg && !wg_fused) {
                rc = filter_gradient();
                if (rc != ORBIT_OK) return rc;
            }
            if (need_dx) {
                ORBIT_REQUIRE(grad_slot[src] < 0, "extractor_backward: unexpected fan-out into a depthwise convolution");
                SLOT_OR_FAIL(k);
                grad_slot[src] = k;
                // wgrad_scratch is free again here (the weight-gradient launches above are earlier on the stream)
                // through_act: the data gradient goes through the producer's activation in the same kernel and leaves the channel
                // sums of that BatchNorm's backward, so the convolution's own backward starts at the finalize step (no reduction
                // pass over dx and y)
                int nblk = 0, wrows = 0;
                DwBnBwd bnb;
                const DwBnBwd* bnb_ptr = nullptr;
                if (through_act) {
                    const BNDesc& sbn = fe->bns[po.bn];
                    if (wg_fused)
                        bnb.wgrad_partial = wgrad_scratch + dwconv_bwd_fused_partial_offset(o.Cin, o.KH), bnb.wgrad_rows = &wrows;
                    bnb.y = tf(L.y[src]), bnb.mean = mean + sbn.fold_off, bnb.invstd = invstd + sbn.fold_off;
                    bnb.scale = scale + sbn.fold_off, bnb.shift = shift + sbn.fold_off, bnb.act = po.act;
                    bnb.partial = partial, bnb.nblk = &nblk;
                    bnb_ptr = &bnb;
                }
                rc = launch_dwconv_dgrad(slot_ptr(kdy), fe->d_packed + o.packed_off, slot_ptr(k), B, o.H, o.W, o.Cin, o.KH,
                                         o.stride, o.pad_t, o.pad_l, o.Ho, o.Wo, s, wgrad_scratch, bnb_ptr);
                if (rc != ORBIT_OK) return rc;
                if (bnb_ptr) pre_reduced[src] = nblk;  // 0: that kernel form has no such epilogue, the slot holds plain dx
                if (wg_fused)
                    rc = wrows > 0 ? launch_dwconv_wgrad_reduce(bnb.wgrad_partial, wrows, o.KH, o.Cin,
                                                                param_grads + fe->params[o.weight].off, s)
                                   : filter_gradient();
            }
            release(kdy);
        } else {  // OP_CONV
            int g = grad_slot[i];
            if (g < 0) continue;  // no gradient reaches this op
            const BNDesc& bn = fe->bns[o.bn];
            const int M = B * o.Ho * o.Wo;
            if (o.pool2) {
                SLOT_OR_FAIL(k);
                rc = launch_maxpool_bwd(slot_ptr(g), reinterpret_cast<const uint8_t*>(tp + L.idx[i]), slot_ptr(k), B, o.Ho,
                                        o.Wo, o.Cout, 2, 2, 0, o.Ho / 2, o.Wo / 2, s);
                if (rc != ORBIT_OK) return rc;
                release(g), g = k;
            }
            // BatchNorm (+ReLU, + residual fan-out)
            const bool fm = film && bn.film_off >= 0;
            const float* gamma = fm ? film_gamma + bn.film_off : fe->d_pool + fe->params[bn.gamma].off;
            float *dgam = nullptr, *dbet = nullptr, *dbias = nullptr;
            if (fm) {
                if (dfilm_gamma) dgam = dfilm_gamma + bn.film_off, dbet = dfilm_beta + bn.film_off;
            } else if (param_grads) {
                dgam = param_grads + fe->params[bn.gamma].off, dbet = param_grads + fe->params[bn.beta].off;
            }
            if (wg && bn.conv_bias >= 0) dbias = param_grads + fe->params[bn.conv_bias].off;
            float* dres = nullptr;
            int dres_acc = 0;
            const int rsrc = o.res >= 0 ? P.res[i] : -1;
            if (rsrc >= first_needed) {
                if (grad_slot[rsrc] >= 0) {
                    dres = slot_ptr(grad_slot[rsrc]), dres_acc = 1;
                } else {
                    SLOT_OR_FAIL(k);
                    grad_slot[rsrc] = k, dres = slot_ptr(k);
                }
            }
            const int src = P.in[i];
            const bool need_dx = src >= first_needed;
            const bool need_dy = need_dx || wg;
            int kdy = -1;
            if (need_dy) {
                kdy = alloc();
                if (kdy < 0) return set_err(ORBIT_ERR_STATE, "extractor_backward: gradient slots exhausted");
            }
            if (pre_reduced[i] > 0) {
                ORBIT_REQUIRE(!dres && !dbias && !o.pool2, "extractor_backward: pre-reduced BatchNorm with a residual or bias");
                float* coef = partial + bn_partial_floats((size_t)pre_reduced[i], o.Cout);
                rc = launch_bn_backward_reduced(slot_ptr(g), tf(L.y[i]), mean + bn.fold_off, invstd + bn.fold_off, gamma,
                                                bn_train, M, o.Cout, need_dy ? slot_ptr(kdy) : nullptr, dgam, dbet, partial,
                                                pre_reduced[i], coef, s);
            } else {
                // partial holds [blocks][2][C] followed by the 3*C apply coefficients
                float* coef = partial + (size_t)bn_reduce_blocks(M, o.Cout) * 2 * o.Cout;
                rc = launch_bn_backward(slot_ptr(g), tf(L.a[i]), tf(L.y[i]), mean + bn.fold_off, invstd + bn.fold_off, gamma,
                                        scale + bn.fold_off, shift + bn.fold_off, bn_train, o.act, M, o.Cout,
                                        need_dy ? slot_ptr(kdy) : nullptr, dres, dres_acc, dgam, dbet, dbias, partial, coef, s);
            }
            if (rc != ORBIT_OK) return rc;
            if (!need_dy && dres) {
                // reductions only, but the residual branch still needs g: rerun the apply pass is not available
                // without dy, so materialise it through a scratch slot
                return set_err(ORBIT_ERR_STATE, "extractor_backward: residual fan-out without a data gradient");
            }
            release(g), grad_slot[i] = -1;
            if (wg) {
                // a gated projection: d/dW of conv(x * gate) - the gate is multiplied into x inside the kernel. The layer's
                // partial tiles stay in a region of their own: the split reductions of all layers run as ONE launch at the end
                float* cw = cwgrad_base + cwgrad_used;
                cwgrad_used += align_up(conv_wgrad_scratch_floats(B, o.Cin, o.Cout, o.KH, o.KW, o.Ho, o.Wo), 64);
                WgradReduceJob* defer = n_wg_jobs < WGRAD_REDUCE_JOBS ? &wg_jobs.j[n_wg_jobs] : nullptr;
                rc = launch_conv_wgrad(out_tensor(src), o.x_nchw, slot_ptr(kdy), param_grads + fe->params[o.weight].off, B,
                                       o.H, o.W, o.Cin, o.Cout, o.KH, o.KW, o.stride, o.pad_t, o.pad_l, o.Ho, o.Wo, cw, s,
                                       o.use_gate ? tf(L.a[P.gate[i]]) : nullptr, defer);
                if (rc != ORBIT_OK) return rc;
                if (defer) ++n_wg_jobs;
            }
            if (need_dx && o.use_gate) {
                // d(x * gate): data gradient of the product, then squeeze-excite backward (gate MLP + average pool)
                const int se = P.gate[i];
                ORBIT_REQUIRE(se >= 0 && grad_slot[src] < 0, "extractor_backward: malformed squeeze-excite block");
                const Op& so = fe->ops[se];
                float* se_scratch = se_scratch_base + se_scratch_used;  // this block's own (see se_jobs)
                se_scratch_used += align_up(se_bwd_scratch_floats(B, o.Cin, so.R), 64);
                SLOT_OR_FAIL(kt);
                rc = launch_conv_dgrad(slot_ptr(kdy), st->d_dgrad + st->dgrad_off[i], nullptr, slot_ptr(kt), up, B, o.H, o.W,
                                       o.Cin, o.Cout, o.KH, o.KW, o.stride, o.pad_t, o.pad_l, o.Ho, o.Wo, s);
                if (rc != ORBIT_OK) return rc;
                float* pg = wg ? param_grads : nullptr;
                // the four parameter gradients of the block's MLP: summed over the frames for ALL blocks in one launch at the
                // end of the reverse pass (nothing in it reads them)
                const bool batch_params = pg != nullptr && n_se_jobs < SE_PARAM_JOBS && so.R <= 48;
                if (batch_params) {
                    se_jobs.j[n_se_jobs++] = se_bwd_param_job(se_scratch, tf(L.p[se]), B, o.Cin, so.R, pg + fe->params[so.se_w1].off,
                                                              pg + fe->params[so.se_b1].off, pg + fe->params[so.se_w2].off,
                                                              pg + fe->params[so.se_b2].off);
                    pg = nullptr;
                }
                // x = SiLU(BatchNorm(depthwise output)): the first (reduction) pass of that BatchNorm's backward is folded
                // into the last pass of the squeeze-excite backward, which then leaves only the sums - g itself is rebuilt by
                // that BatchNorm's apply pass from d(x * gate) (slot kt stays the depthwise op's gradient slot)
                SeBnFuse fuse;
                const SeBnFuse* fuse_ptr = nullptr;
                static const char* write_g_env = getenv("ORBIT_SE_BWD_WRITE_G");  // tuning experiments only
                const bool sums_only = fe->ops[src].kind == OP_DWCONV && !(write_g_env && atoi(write_g_env) != 0) &&
                                       (unsigned long long)B * o.H * o.W * (o.Cin / 4) < (1ull << 32);
                int k = kt;
                if (!sums_only) {
                    k = alloc();
                    if (k < 0) return set_err(ORBIT_ERR_STATE, "extractor_backward: gradient slots exhausted");
                }
                grad_slot[src] = k;
                if (fe->ops[src].kind == OP_DWCONV) {
                    const BNDesc& sbn = fe->bns[fe->ops[src].bn];
                    fuse.y = tf(L.y[src]), fuse.mean = mean + sbn.fold_off, fuse.invstd = invstd + sbn.fold_off;
                    fuse.scale = scale + sbn.fold_off, fuse.shift = shift + sbn.fold_off, fuse.act = fe->ops[src].act;
                    fuse.partial = partial;
                    fuse_ptr = &fuse;
                    pre_reduced[src] = B * se_pool_chunks(B, o.H * o.W, o.Cin);
                    if (sums_only)
                        gated[src].gate = tf(L.a[se]), gated[src].dpooled = se_bwd_dpooled(se_scratch, B, o.Cin, so.R),
                        gated[src].HW = o.H * o.W;
                }
                rc = launch_se_gate_backward(slot_ptr(kt), out_tensor(src), tf(L.p[se]), tf(L.a[se]),
                                             fe->d_pool + fe->params[so.se_w1].off, fe->d_pool + fe->params[so.se_b1].off,
                                             fe->d_pool + fe->params[so.se_w2].off, fe->d_pool + fe->params[so.se_b2].off,
                                             sums_only ? nullptr : slot_ptr(k), pg ? pg + fe->params[so.se_w1].off : nullptr,
                                             pg ? pg + fe->params[so.se_b1].off : nullptr,
                                             pg ? pg + fe->params[so.se_w2].off : nullptr,
                                             pg ? pg + fe->params[so.se_b2].off : nullptr, se_scratch, B, o.H * o.W, o.Cin,
                                             so.R, s, fuse_ptr, fe->d_packed + so.packed_off);
                if (!sums_only) release(kt);
            } else if (need_dx) {
                const float* acc = nullptr;
                if (grad_slot[src] < 0) {
                    SLOT_OR_FAIL(k);
                    grad_slot[src] = k;
                } else {
                    acc = slot_ptr(grad_slot[src]);
                }
                rc = launch_conv_dgrad(slot_ptr(kdy), st->d_dgrad + st->dgrad_off[i], acc, slot_ptr(grad_slot[src]), up, B,
                                       o.H, o.W, o.Cin, o.Cout, o.KH, o.KW, o.stride, o.pad_t, o.pad_l, o.Ho, o.Wo, s);
            }
            release(kdy);
        }
        if (rc != ORBIT_OK) return rc;
    }
#undef SLOT_OR_FAIL
    if (int rc = launch_conv_wgrad_reduce_batched(wg_jobs, n_wg_jobs, s)) return rc;
    return launch_se_param_grad_batched(se_jobs, n_se_jobs, s);
}

extern "C" {

// diagnostics: how many training-entry calls of this plan replayed a captured graph / ran eagerly
int orbit_extractor_train_graph_stats(const orbit_extractor_t* fe, long* replays, long* eager) {
    ORBIT_REQUIRE(fe, "extractor_train_graph_stats: null pointer");
    if (replays) *replays = fe->train_graph_replays;
    if (eager) *eager = fe->train_graph_eager;
    return ORBIT_OK;
}

}  // extern "C"
