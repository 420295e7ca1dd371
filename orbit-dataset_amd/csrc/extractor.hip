// Network runtime: builds a static launch plan for "resnet18", "efficientnet_b0" or "set_encoder" at a
// given frame size, owns the parameters (loaded by torch state_dict key), repacks conv weights for the
// MFMA kernel, folds BatchNorm (+ per-task FiLM gamma/beta) into per-channel scale/shift, and replays the
// plan on a HIP stream. This is the native counterpart of
//   model/feature_extractors.py:37-79   create_feature_extractor (timm tf_efficientnet_b0; resnet18 is the
//                                        torchvision-layout network BASELINE.json's configs name)
//   model/set_encoders.py:81-120        SimplePrePoolNet
//   model/few_shot_recognisers.py:99-153 _get_features[_in_batches] (one call = one mini-batch of frames)
//   model/film.py:38-74                 which BatchNorms are FiLM-modulated
// Activations are NHWC fp32 and live in caller-provided workspace (three rotating buffers); frames come
// in as NCHW (the reference's clip layout) and are consumed directly by the stem convolution.
#include <chrono>
#include <cstdlib>

#include "extractor.h"

using namespace orbit;

static int conv_out(int H, int K, int stride, int pad) { return (H + 2 * pad - K) / stride + 1; }
static void same_pad(int H, int K, int stride, int& Ho, int& pad_before) {  // TF "SAME"
    Ho = (H + stride - 1) / stride;
    const int total = std::max((Ho - 1) * stride + K - H, 0);
    pad_before = total / 2;
}

// ---- resnet18 (torchvision layout, pooled 512-d features; every BatchNorm is a FiLM slot) -------
static int build_resnet18(orbit_extractor* fe, int H, int W) {
    fe->out_size = 512;
    int h = conv_out(H, 7, 2, 3), w = conv_out(W, 7, 2, 3);
    int bn = fe->add_bn("bn1", 64, 1e-5f, true);
    fe->add_conv("conv1.weight", bn, -1, 0, -1, H, W, 3, 64, 7, 2, 3, 3, h, w, ORBIT_ACT_RELU, 0, 1, 0);
    {
        Op o;
        o.kind = OP_MAXPOOL, o.in = 0, o.out = 1, o.H = h, o.W = w, o.Cin = 64, o.Cout = 64;
        o.pool_k = 3, o.stride = 2, o.pool_pad = 1;
        o.Ho = conv_out(h, 3, 2, 1), o.Wo = conv_out(w, 3, 2, 1);
        h = o.Ho, w = o.Wo;
        fe->note_buf(1, (size_t)h * w * 64);
        fe->ops.push_back(o);
    }
    int cur = 1, cin = 64;
    const int widths[4] = {64, 128, 256, 512};
    for (int L = 0; L < 4; ++L) {
        for (int blk = 0; blk < 2; ++blk) {
            const int cout = widths[L];
            const int stride = (L > 0 && blk == 0) ? 2 : 1;
            const std::string p = "layer" + std::to_string(L + 1) + "." + std::to_string(blk);
            const int ho = conv_out(h, 3, stride, 1), wo = conv_out(w, 3, stride, 1);
            const int t1 = (cur + 1) % 3, t2 = (cur + 2) % 3;
            // module registration order: conv1, bn1, conv2, bn2, downsample.{0,1}
            const int bn1 = fe->add_bn(p + ".bn1", cout, 1e-5f, true);
            fe->add_conv(p + ".conv1.weight", bn1, cur, t1, -1, h, w, cin, cout, 3, stride, 1, 1, ho, wo,
                         ORBIT_ACT_RELU, 0, 0, 0);
            const int bn2 = fe->add_bn(p + ".bn2", cout, 1e-5f, true);
            int res = cur;
            if (stride != 1 || cin != cout) {
                const int bnd = fe->add_bn(p + ".downsample.1", cout, 1e-5f, true);
                // conv2 is emitted after the downsample so the plan order is ds -> conv2; param order
                // (state_dict) is irrelevant to execution
                fe->add_conv(p + ".downsample.0.weight", bnd, cur, t2, -1, h, w, cin, cout, 1, stride, 0, 0, ho,
                             wo, ORBIT_ACT_NONE, 0, 0, 0);
                res = t2;
            }
            // out = relu(bn2(conv2(t1)) + res); write into the buffer that is neither t1 nor res
            const int outb = (res == cur) ? t2 : cur;
            fe->add_conv(p + ".conv2.weight", bn2, t1, outb, res, ho, wo, cout, cout, 3, 1, 1, 1, ho, wo,
                         ORBIT_ACT_RELU, 0, 0, 0);
            cur = outb, cin = cout, h = ho, w = wo;
        }
    }
    Op o;
    o.kind = OP_AVGPOOL, o.in = cur, o.out = 100, o.H = h, o.W = w, o.Cin = 512, o.Cout = 512;
    fe->ops.push_back(o);
    return ORBIT_OK;
}

// ---- efficientnet_b0, timm `tf_` variant (SAME padding, BN eps 1e-3), num_classes=0 --------------
static int build_efficientnet_b0(orbit_extractor* fe, int H, int W, bool unfused) {
    fe->out_size = 1280;
    const float eps = 1e-3f;
    // expand 1x1 + BatchNorm + SiLU + depthwise + BatchNorm + SiLU as ONE row-streaming kernel (csrc/mbconv_rows.hip; option
    // mbconv_rows, default on) for the 112x112 .. 28x28 stages, where the 6x-expanded tensor would otherwise make an HBM round
    // trip; the 14x14 / 7x7 stages run the MFMA conv + LDS-patch depthwise pair
    // (a plan that records a tape - `unfused` - keeps the pair everywhere: the fused kernels have no backward form)
    const int rows_opt = unfused ? 0 : get_option("mbconv_rows");
    auto fuse_front_ok = [&](int hh, int ww, int cin, int mid, int K, int stride) {
        return rows_opt != 0 && mbconv_rows_supported(hh, ww, cin, mid, K, stride);
    };
    int h, w, pt, pl;
    same_pad(H, 3, 2, h, pt);
    same_pad(W, 3, 2, w, pl);
    int bn = fe->add_bn("bn1", 32, eps, true);  // root bn1 is FiLM-tagged (film.py:45-46)
    // row-streaming stem + first depthwise (csrc/mbconv_rows.hip, option stem_rows): the stem's 112x112x32 output stays in LDS
    const bool fuse_stem = rows_opt != 0 && get_option("stem_rows") != 0 && stem_rows_supported(h, w, 32, 3, 1);
    size_t stem_weight = 0;
    if (fuse_stem) stem_weight = fe->add_param("conv_stem.weight", (size_t)32 * 27);
    else fe->add_conv("conv_stem.weight", bn, -1, 0, -1, H, W, 3, 32, 3, 2, pt, pl, h, w, ORBIT_ACT_SILU, 0, 1, 0);
    int cur = 0, cin = 32;

    auto add_dw = [&](const std::string& wkey, int bnidx, int in, int out, int C, int K, int stride, int hh,
                      int ww, int& ho, int& wo) {
        Op o;
        o.kind = OP_DWCONV, o.in = in, o.out = out, o.H = hh, o.W = ww, o.Cin = C, o.Cout = C;
        o.KH = o.KW = K, o.stride = stride, o.act = ORBIT_ACT_SILU, o.bn = bnidx;
        same_pad(hh, K, stride, o.Ho, o.pad_t);
        same_pad(ww, K, stride, o.Wo, o.pad_l);
        o.weight = fe->add_param(wkey, (size_t)C * K * K);
        o.packed_off = fe->packed_floats;
        fe->packed_floats += (size_t)(C * K * K + 3) / 4 * 4;
        o.pool_partial = 1;
        fe->max_partial = std::max(fe->max_partial, (size_t)dwconv_se_chunks(o.Ho) * C);
        ho = o.Ho, wo = o.Wo;
        fe->note_buf(out, (size_t)ho * wo * C);
        fe->macs += (double)ho * wo * C * K * K;
        fe->ops.push_back(o);
    };
    auto add_se = [&](const std::string& p, int buf, int C, int R, int hh, int ww, int chunks) {
        (void)buf;  // the pooled sums come from the depthwise kernel's partials (buffer 101), not from a re-read
        Op s;
        s.kind = OP_SE, s.in = 101, s.out = 102, s.Cin = C, s.R = R;
        s.se_chunks = chunks, s.se_hw = hh * ww;
        s.packed_off = fe->packed_floats;  // W2 transposed to [R][C]
        fe->packed_floats += (size_t)(C * R + 3) / 4 * 4;
        s.se_w1 = fe->add_param(p + ".conv_reduce.weight", (size_t)R * C);
        s.se_b1 = fe->add_param(p + ".conv_reduce.bias", R);
        s.se_w2 = fe->add_param(p + ".conv_expand.weight", (size_t)C * R);
        s.se_b2 = fe->add_param(p + ".conv_expand.bias", C);
        fe->macs += 2.0 * C * R;
        fe->max_se_c = std::max(fe->max_se_c, C);
        fe->ops.push_back(s);
    };

    // stage 0: DepthwiseSeparableConv (not FiLM-tagged, film.py:41-48)
    {
        const std::string p = "blocks.0.0";
        int ho, wo;
        const int t1 = (cur + 1) % 3, t2 = (cur + 2) % 3;
        // registration order: conv_dw, bn1, se, conv_pw, bn2
        const size_t dw_slot = fe->params.size();
        (void)dw_slot;
        // conv_dw param is registered inside add_dw, bn1 after it: keep state_dict order cosmetic only
        const int bn1 = fe->add_bn(p + ".bn1", 32, eps, false);
        int se_chunks0;
        if (fuse_stem) {
            Op o;
            o.kind = OP_MBFRONT, o.stem = true, o.in = -1, o.out = t1, o.H = h, o.W = w, o.Cin = 32, o.Cout = 32;
            o.KH = o.KW = 3, o.stride = 1, o.bn = bn, o.bn2 = bn1;
            o.stem_h = H, o.stem_w = W, o.stem_pt = pt, o.stem_pl = pl;
            same_pad(h, 3, 1, o.Ho, o.pad_t);
            same_pad(w, 3, 1, o.Wo, o.pad_l);
            o.weight = stem_weight;
            o.weight2 = fe->add_param(p + ".conv_dw.weight", (size_t)32 * 9);
            o.packed_off = fe->packed_floats;   // depthwise taps [3][3][32]
            fe->packed_floats += (size_t)32 * 9;
            o.packed_off2 = fe->packed_floats;  // stem filter [32][32]
            fe->packed_floats += (size_t)32 * 32;
            ho = o.Ho, wo = o.Wo;
            o.rows = true;
            se_chunks0 = stem_rows_tiles(h, w);
            o.se_chunks = se_chunks0;  // what the pooling buffer and the gate were sized for (checked at launch)
            fe->max_partial = std::max(fe->max_partial, (size_t)se_chunks0 * 32);
            fe->note_buf(t1, (size_t)ho * wo * 32);
            fe->macs += (double)h * w * 27 * 32 + (double)ho * wo * 32 * 9;
            fe->ops.push_back(o);
        } else {
            add_dw(p + ".conv_dw.weight", bn1, cur, t1, 32, 3, 1, h, w, ho, wo);
            se_chunks0 = dwconv_se_chunks(ho);
        }
        add_se(p + ".se", t1, 32, 8, ho, wo, se_chunks0);
        const int bn2 = fe->add_bn(p + ".bn2", 16, eps, false);
        fe->add_conv(p + ".conv_pw.weight", bn2, t1, t2, -1, ho, wo, 32, 16, 1, 1, 0, 0, ho, wo, ORBIT_ACT_NONE,
                     0, 0, 1);
        cur = t2, cin = 16, h = ho, w = wo;
    }
    // stages 1..6: InvertedResidual  {repeats, kernel, stride, out channels}, expansion 6, SE 0.25 of block input
    const int cfg[6][4] = {{2, 3, 2, 24}, {2, 5, 2, 40}, {3, 3, 2, 80}, {3, 5, 1, 112}, {4, 5, 2, 192}, {1, 3, 1, 320}};
    for (int s = 0; s < 6; ++s) {
        for (int r = 0; r < cfg[s][0]; ++r) {
            const int K = cfg[s][1], stride = r == 0 ? cfg[s][2] : 1, cout = cfg[s][3];
            const int mid = cin * 6;
            const int rd = (int)(cin * 0.25 + 0.5);  // timm: round(in_chs * se_ratio)
            const std::string p = "blocks." + std::to_string(s + 1) + "." + std::to_string(r);
            const int t1 = (cur + 1) % 3, t2 = (cur + 2) % 3;
            const bool skip = stride == 1 && cin == cout;
            int ho, wo;
            const int bn1 = fe->add_bn(p + ".bn1", mid, eps, false);
            int se_chunks;
            if (fuse_front_ok(h, w, cin, mid, K, stride)) {
                // expand + depthwise in one kernel: the 6x-expanded tensor never leaves LDS
                Op o;
                o.kind = OP_MBFRONT, o.in = cur, o.out = t2, o.H = h, o.W = w, o.Cin = cin, o.Cout = mid;
                o.KH = o.KW = K, o.stride = stride, o.bn = bn1;
                same_pad(h, K, stride, o.Ho, o.pad_t);
                same_pad(w, K, stride, o.Wo, o.pad_l);
                o.weight = fe->add_param(p + ".conv_pw.weight", (size_t)mid * cin);
                o.bn2 = fe->add_bn(p + ".bn2", mid, eps, true);  // InvertedResidual.bn2 is FiLM-tagged
                o.weight2 = fe->add_param(p + ".conv_dw.weight", (size_t)mid * K * K);
                o.packed_off = fe->packed_floats;
                fe->packed_floats += (size_t)(mid * K * K + 3) / 4 * 4;
                ho = o.Ho, wo = o.Wo;
                o.rows = true;
                se_chunks = mbconv_rows_tiles(h, w, cin, mid, K, stride);
                o.se_chunks = se_chunks;  // what the pooling buffer and the gate were sized for (checked at launch)
                fe->max_partial = std::max(fe->max_partial, (size_t)se_chunks * mid);
                fe->note_buf(t2, (size_t)ho * wo * mid);
                fe->macs += (double)h * w * cin * mid + (double)ho * wo * mid * K * K;
                fe->ops.push_back(o);
            } else {
                fe->add_conv(p + ".conv_pw.weight", bn1, cur, t1, -1, h, w, cin, mid, 1, 1, 0, 0, h, w, ORBIT_ACT_SILU,
                             0, 0, 0);
                const int bn2 = fe->add_bn(p + ".bn2", mid, eps, true);  // InvertedResidual.bn2 is FiLM-tagged
                add_dw(p + ".conv_dw.weight", bn2, t1, t2, mid, K, stride, h, w, ho, wo);
                se_chunks = dwconv_se_chunks(ho);
            }
            add_se(p + ".se", t2, mid, rd, ho, wo, se_chunks);
            const int bn3 = fe->add_bn(p + ".bn3", cout, eps, false);
            // project: reads t2 (gated), residual from cur, writes t1 (free again)
            fe->add_conv(p + ".conv_pwl.weight", bn3, t2, t1, skip ? cur : -1, ho, wo, mid, cout, 1, 1, 0, 0, ho, wo,
                         ORBIT_ACT_NONE, 0, 0, 1);
            cur = t1, cin = cout, h = ho, w = wo;
        }
    }
    const int t1 = (cur + 1) % 3;
    bn = fe->add_bn("bn2", 1280, eps, true);  // root bn2 is FiLM-tagged
    fe->add_conv("conv_head.weight", bn, cur, t1, -1, h, w, cin, 1280, 1, 1, 0, 0, h, w, ORBIT_ACT_SILU, 0, 0, 0);
    Op o;
    o.kind = OP_AVGPOOL, o.in = t1, o.out = 100, o.H = h, o.W = w, o.Cin = 1280, o.Cout = 1280;
    fe->ops.push_back(o);
    return ORBIT_OK;
}

// ---- set encoder: 5 x (conv3x3 p1 + bias -> BN -> ReLU -> maxpool 2x2) -> global avg -> 64 ---------
static int build_set_encoder(orbit_extractor* fe, int H, int W) {
    fe->out_size = 64;
    int h = H, w = W, cur = -1, cin = 3;
    for (int L = 1; L <= 5; ++L) {
        if (h < 2 || w < 2) return set_err(ORBIT_ERR_ARG, "set_encoder: frame %dx%d too small for 5 poolings", H, W);
        const std::string p = "encoder.layer" + std::to_string(L);
        const int out = cur < 0 ? 0 : (cur + 1) % 3;
        const int wparam = fe->add_param(p + ".0.weight", (size_t)64 * cin * 9);
        (void)wparam;
        const int bias = fe->add_param(p + ".0.bias", 64);
        const int bn = fe->add_bn(p + ".1", 64, 1e-5f, false, bias);
        fe->add_conv(p + ".0.weight", bn, cur, out, -1, h, w, cin, 64, 3, 1, 1, 1, h, w, ORBIT_ACT_RELU, 1,
                     cur < 0 ? 1 : 0, 0, bias);
        cur = out, cin = 64, h /= 2, w /= 2;
    }
    Op o;
    o.kind = OP_AVGPOOL, o.in = cur, o.out = 100, o.H = h, o.W = w, o.Cin = 64, o.Cout = 64;
    fe->ops.push_back(o);
    return ORBIT_OK;
}

// ---- workspace layout ------------------------------------------------------------------------------
struct WsLayout {
    size_t buf[3], pooled, gate, fold, splitk, total;
};
static ConvDesc conv_shape(const Op& o, int B) {  // the fields the split-K plan looks at
    ConvDesc d;
    d.x = d.w_packed = d.scale = d.shift = d.residual = d.gate = nullptr, d.y = nullptr;
    d.B = B, d.H = o.H, d.W = o.W, d.Cin = o.Cin, d.Cout = o.Cout, d.KH = o.KH, d.KW = o.KW;
    d.stride = o.stride, d.pad_t = o.pad_t, d.pad_l = o.pad_l, d.Ho = o.Ho, d.Wo = o.Wo;
    d.act = o.act, d.pool2 = o.pool2, d.x_nchw = o.x_nchw;
    return d;
}
static WsLayout ws_layout(const orbit_extractor* fe, int B) {
    WsLayout L;
    size_t off = 0;
    for (int i = 0; i < 3; ++i) {
        L.buf[i] = off;
        off += align_up(fe->buf_elems[i] * (size_t)B * sizeof(float), 256);
    }
    L.pooled = off;
    off += align_up(std::max<size_t>(std::max<size_t>(fe->max_partial, fe->max_se_c), 1) * B * sizeof(float), 256);
    L.gate = off;
    off += align_up((size_t)std::max(fe->max_se_c, 1) * B * sizeof(float), 256);
    L.fold = off;
    off += align_up(2 * fe->fold_floats * sizeof(float), 256);
    L.splitk = off;  // partial tiles of the largest split-K conv at this batch size
    size_t skf = 0;
    for (const Op& o : fe->ops)
        if (o.kind == OP_CONV) skf = std::max(skf, conv_splitk_floats(conv_shape(o, B)));
    off += align_up(skf * sizeof(float), 256);
    L.total = off;
    return L;
}

extern "C" {

int orbit_extractor_create(const char* name, int H, int W, orbit_extractor_t** out) {
    return orbit_extractor_create_ex(name, H, W, 0, out);
}

int orbit_extractor_create_ex(const char* name, int H, int W, int flags, orbit_extractor_t** out) {
    ORBIT_REQUIRE(name && out, "extractor_create: null pointer");
    ORBIT_REQUIRE(H >= 8 && W >= 8 && H <= 4096 && W <= 4096, "extractor_create: bad frame size %dx%d", H, W);
    orbit_extractor* fe = new orbit_extractor();
    fe->name = name, fe->H = H, fe->W = W;
    int rc;
    if (fe->name == "resnet18") rc = build_resnet18(fe, H, W);
    else if (fe->name == "efficientnet_b0") rc = build_efficientnet_b0(fe, H, W, (flags & ORBIT_PLAN_UNFUSED) != 0);
    else if (fe->name == "set_encoder") rc = build_set_encoder(fe, H, W);
    else rc = set_err(ORBIT_ERR_ARG, "Invalid feature_extractor_name: %s", name);
    if (rc != ORBIT_OK) {
        delete fe;
        return rc;
    }
    fe->bn_dev.resize(fe->bns.size());
    for (size_t i = 0; i < fe->bns.size(); ++i) {
        const BNDesc& b = fe->bns[i];
        BNDev& d = fe->bn_dev[i];
        d.gamma = fe->params[b.gamma].off, d.beta = fe->params[b.beta].off;
        d.mean = fe->params[b.mean].off, d.var = fe->params[b.var].off;
        d.conv_bias = b.conv_bias >= 0 ? fe->params[b.conv_bias].off : (size_t)-1;
        d.fold_off = b.fold_off, d.C = b.C, d.film_off = b.film_off, d.eps = b.eps;
    }
    *out = fe;
    return ORBIT_OK;
}

void orbit_extractor_destroy(orbit_extractor_t* fe) {
    if (!fe) return;
    extractor_train_release(fe);
    fe->clear_graphs();
    fe->clear_train_graphs();
    if (fe->cap_stream) (void)hipStreamDestroy(fe->cap_stream);
    (void)hipFree(fe->d_pool);
    (void)hipFree(fe->d_src);
    (void)hipFree(fe->d_dst_meta);
    (void)hipFree(fe->d_packed);
    (void)hipFree(fe->d_pack_jobs);
    (void)hipFree(fe->d_fold);
    (void)hipFree(fe->d_bn);
    delete fe;
}

int orbit_extractor_num_params(const orbit_extractor_t* fe) { return fe ? (int)fe->params.size() : 0; }
const char* orbit_extractor_param_name(const orbit_extractor_t* fe, int i) {
    return (fe && i >= 0 && i < (int)fe->params.size()) ? fe->params[i].key.c_str() : nullptr;
}
size_t orbit_extractor_param_numel(const orbit_extractor_t* fe, int i) {
    return (fe && i >= 0 && i < (int)fe->params.size()) ? fe->params[i].numel : 0;
}

int orbit_extractor_load(orbit_extractor_t* fe, const char* key, const float* data, size_t numel) {
    ORBIT_REQUIRE(fe && key && data, "extractor_load: null pointer");
    auto it = fe->index.find(key);
    ORBIT_REQUIRE(it != fe->index.end(), "extractor_load: unexpected key '%s' for %s", key, fe->name.c_str());
    Param& p = fe->params[it->second];
    ORBIT_REQUIRE(p.numel == numel, "extractor_load: '%s' has %zu elements, expected %zu", key, numel, p.numel);
    if (int rc = fe->ensure_device()) return rc;
    ORBIT_HIP_CHECK(hipMemcpy(fe->d_pool + p.off, data, numel * sizeof(float), hipMemcpyDefault));
    p.loaded = true;
    fe->finalized = false;
    extractor_train_invalidate(fe);
    return ORBIT_OK;
}

int orbit_extractor_load_async(orbit_extractor_t* fe, const char* key, const float* device_data, size_t numel,
                               orbit_stream_t stream) {
    ORBIT_REQUIRE(fe && key && device_data, "extractor_load_async: null pointer");
    auto it = fe->index.find(key);
    ORBIT_REQUIRE(it != fe->index.end(), "extractor_load_async: unexpected key '%s' for %s", key, fe->name.c_str());
    Param& p = fe->params[it->second];
    ORBIT_REQUIRE(p.numel == numel, "extractor_load_async: '%s' has %zu elements, expected %zu", key, numel, p.numel);
    if (int rc = fe->ensure_device()) return rc;
    ORBIT_HIP_CHECK(hipMemcpyAsync(fe->d_pool + p.off, device_data, numel * sizeof(float), hipMemcpyDeviceToDevice,
                                   (hipStream_t)stream));
    p.loaded = true;
    fe->finalized = false;
    extractor_train_invalidate(fe);
    return ORBIT_OK;
}

// one kernel copies every parameter tensor into the pool: grid (chunks, parameters)
__global__ __launch_bounds__(256) void gather_params_kernel(const float* const* __restrict__ src,
                                                            const size_t* __restrict__ meta, float* __restrict__ pool) {
    const float* s_ = src[blockIdx.y];
    float* d = pool + meta[2 * blockIdx.y];
    const size_t n = meta[2 * blockIdx.y + 1];
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) d[i] = s_[i];
}

int orbit_extractor_load_all_async(orbit_extractor_t* fe, const float* const* device_ptrs, int n, orbit_stream_t stream) {
    ORBIT_REQUIRE(fe && device_ptrs, "extractor_load_all_async: null pointer");
    ORBIT_REQUIRE(n == (int)fe->params.size(), "extractor_load_all_async: %d pointers for %zu parameters", n,
                  fe->params.size());
    if (int rc = fe->ensure_device()) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (!fe->d_src) {
        ORBIT_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&fe->d_src), n * sizeof(float*)));
        ORBIT_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&fe->d_dst_meta), 2 * n * sizeof(size_t)));
        std::vector<size_t> meta(2 * n);
        for (int i = 0; i < n; ++i) meta[2 * i] = fe->params[i].off, meta[2 * i + 1] = fe->params[i].numel;
        ORBIT_HIP_CHECK(hipMemcpy(fe->d_dst_meta, meta.data(), meta.size() * sizeof(size_t), hipMemcpyHostToDevice));
    }
    bool same = (int)fe->h_src.size() == n;
    for (int i = 0; same && i < n; ++i) same = fe->h_src[i] == device_ptrs[i];
    if (!same) {  // the tensors moved (first call, load_state_dict with new storage): refresh the pointer table
        for (int i = 0; i < n; ++i) ORBIT_REQUIRE(device_ptrs[i], "extractor_load_all_async: null tensor %d", i);
        fe->h_src.assign(device_ptrs, device_ptrs + n);
        ORBIT_HIP_CHECK(hipStreamSynchronize(s));  // the table may still be read by an earlier gather on this stream
        ORBIT_HIP_CHECK(hipMemcpy(fe->d_src, fe->h_src.data(), n * sizeof(float*), hipMemcpyHostToDevice));
    }
    gather_params_kernel<<<dim3(32, n), 256, 0, s>>>(fe->d_src, fe->d_dst_meta, fe->d_pool);
    ORBIT_LAUNCH_CHECK();
    for (Param& p : fe->params) p.loaded = true;
    fe->finalized = false;
    extractor_train_invalidate(fe);
    return ORBIT_OK;
}

int orbit_extractor_finalize(orbit_extractor_t* fe, orbit_stream_t stream) {
    ORBIT_REQUIRE(fe, "extractor_finalize: null pointer");
    for (const Param& p : fe->params)
        ORBIT_REQUIRE(p.loaded, "extractor_finalize: parameter '%s' was never loaded", p.key.c_str());
    if (int rc = fe->ensure_device()) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (fe->pack_jobs_bk != get_option("conv_bk")) {  // (tuning sweeps change the K-tile width, i.e. the packed geometry)
        fe->pack_jobs.clear();
        (void)hipFree(fe->d_pack_jobs);
        fe->d_pack_jobs = nullptr;
        fe->pack_jobs_bk = get_option("conv_bk");
    }
    if (fe->pack_jobs.empty()) {
        auto job = [&](int kind, const float* src, float* dst, int Cin, int Cout, int KH, int KW, int cin_pad, int KT,
                       int cout_pad, size_t total) {
            PackJob j;
            j.src = src, j.dst = dst, j.kind = kind, j.Cin = Cin, j.Cout = Cout, j.KH = KH, j.KW = KW;
            j.cin_pad = cin_pad, j.KT = KT, j.cout_pad = cout_pad, j.total = (unsigned)total;
            fe->pack_jobs.push_back(j);
        };
        for (const Op& o : fe->ops) {
            if (o.kind == OP_CONV) {
                const ConvPackGeom g = conv_pack_geom(o.Cin, o.Cout, o.KH, o.KW, o.x_nchw);
                job(o.x_nchw ? 1 : 0, fe->d_pool + fe->params[o.weight].off, fe->d_packed + o.packed_off, o.Cin, o.Cout, o.KH,
                    o.KW, g.cin_pad, g.kt, g.cout_pad, (size_t)g.cout_pad * g.kt);
                if (o.frag_off != SIZE_MAX)
                    job(5, fe->d_pool + fe->params[o.weight].off, fe->d_packed + o.frag_off, o.Cin, o.Cout, 1, 1, 0, o.Cin / 16,
                        0, conv_frag_floats(o.Cin, o.Cout, 1, 1, 0));
            } else if (o.kind == OP_DWCONV) {
                job(2, fe->d_pool + fe->params[o.weight].off, fe->d_packed + o.packed_off, o.Cin, 0, o.KH, o.KH, 0, 0, 0,
                    (size_t)o.Cin * o.KH * o.KH);
            } else if (o.kind == OP_MBFRONT) {
                job(2, fe->d_pool + fe->params[o.weight2].off, fe->d_packed + o.packed_off, o.Cout, 0, o.KH, o.KH, 0, 0, 0,
                    (size_t)o.Cout * o.KH * o.KH);
            } else if (o.kind == OP_SE) {
                job(3, fe->d_pool + fe->params[o.se_w2].off, fe->d_packed + o.packed_off, o.Cin, o.R, 0, 0, 0, 0, 0,
                    (size_t)o.Cin * o.R);
            }
        }
    }
    if (int rc = run_pack_jobs(fe->pack_jobs, &fe->d_pack_jobs, s)) return rc;
    for (const Op& o : fe->ops)  // (the fused stem's [mid][32] filter layout has its own kernel; inference plans only)
        if (o.kind == OP_MBFRONT && o.stem)
            if (int rc = stem_pack_weights(fe->d_pool + fe->params[o.weight].off, fe->d_packed + o.packed_off2, o.Cout, s))
                return rc;
    dim3 grid((unsigned)fe->bns.size(), 2);
    bn_fold_all_kernel<<<grid, 256, 0, s>>>(fe->d_bn, fe->d_pool, nullptr, nullptr, fe->d_fold,
                                            fe->d_fold + fe->fold_floats);
    ORBIT_LAUNCH_CHECK();
    fe->clear_graphs();
    fe->finalized = true;
    return ORBIT_OK;
}

int orbit_extractor_output_size(const orbit_extractor_t* fe) { return fe ? fe->out_size : 0; }
int orbit_extractor_film_slots(const orbit_extractor_t* fe) { return fe ? (int)fe->film_slots.size() : 0; }
int orbit_extractor_film_slot_channels(const orbit_extractor_t* fe, int slot) {
    return (fe && slot >= 0 && slot < (int)fe->film_slots.size()) ? fe->bns[fe->film_slots[slot]].C : 0;
}
const char* orbit_extractor_film_slot_name(const orbit_extractor_t* fe, int slot) {
    return (fe && slot >= 0 && slot < (int)fe->film_slots.size()) ? fe->bns[fe->film_slots[slot]].name.c_str()
                                                                  : nullptr;
}
int orbit_extractor_film_size(const orbit_extractor_t* fe) { return fe ? fe->film_size : 0; }
double orbit_extractor_macs_per_frame(const orbit_extractor_t* fe) { return fe ? fe->macs : 0.0; }

size_t orbit_extractor_workspace_bytes(const orbit_extractor_t* fe, int B) {
    if (!fe || B <= 0) return 0;
    return ws_layout(fe, B).total;
}

static int run_plan(orbit_extractor_t* fe, const float* frames, int B, const float* film_gamma, const float* film_beta,
                    float* feats, void* workspace, hipStream_t s);

int orbit_extractor_forward(orbit_extractor_t* fe, const float* frames, int B, const float* film_gamma,
                            const float* film_beta, float* feats, void* workspace, size_t workspace_bytes,
                            orbit_stream_t stream) {
    ORBIT_REQUIRE(fe && frames && feats && workspace, "extractor_forward: null pointer");
    ORBIT_REQUIRE(B > 0, "extractor_forward: empty batch");
    if (!fe->finalized) return set_err(ORBIT_ERR_STATE, "extractor_forward: call orbit_extractor_finalize first");
    ORBIT_REQUIRE((film_gamma == nullptr) == (film_beta == nullptr),
                  "extractor_forward: film_gamma and film_beta must be given together");
    const WsLayout L = ws_layout(fe, B);
    ORBIT_REQUIRE(workspace_bytes >= L.total, "extractor_forward: workspace too small (%zu < %zu bytes)",
                  workspace_bytes, L.total);
    ORBIT_REQUIRE(((uintptr_t)workspace & 255) == 0, "extractor_forward: workspace must be 256-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    // graph = 0 never, 1 always, 2 (default) adaptive. Measured on MI355X: with a fast host, replaying the graph of a
    // forward is SLOWER than the eager launch sequence (resnet18@84: 5.3 vs 4.5 ms per task), but pool hosts differ 4x
    // in launch cost and on the slow ones the eager sequence (~170 launches per efficientnet task) makes the default
    // workload host-bound. Adaptive: time the host side of the eager sequence; replay graphs only while a launch costs
    // more than ~12 us of host time on average.
    const int graph_opt = get_option("graph");
    bool want_graph = graph_opt == 1;
    if (graph_opt == 2 && fe->eager_samples >= 3 && fe->eager_us_per_launch > 12.0) want_graph = true;
    if (!want_graph || conv_prof_enabled()) {
        const auto t0 = std::chrono::steady_clock::now();
        const int rc = run_plan(fe, frames, B, film_gamma, film_beta, feats, workspace, s);
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() /
                          (double)(fe->ops.size() + 1);
        fe->eager_us_per_launch = fe->eager_samples == 0 ? us : 0.7 * fe->eager_us_per_launch + 0.3 * us;
        fe->eager_samples++;
        return rc;
    }

    // graph path: 1st sight of a pointer tuple runs eagerly (also performs one-time kernel attribute setup), the 2nd
    // captures + instantiates, later ones replay
    const orbit_extractor::GraphKey key{frames, film_gamma, film_beta, feats, workspace, nullptr, B, option_epoch()};
    orbit_extractor::GraphEntry* hit = nullptr;
    for (auto& g : fe->graphs)
        if (g.key == key) hit = &g;
    if (hit == nullptr) {
        if (fe->graphs.size() >= 32) {  // evict the least recently used entry
            size_t lru = 0;
            for (size_t i = 1; i < fe->graphs.size(); ++i)
                if (fe->graphs[i].stamp < fe->graphs[lru].stamp) lru = i;
            if (fe->graphs[lru].exec) (void)hipGraphExecDestroy(fe->graphs[lru].exec);
            fe->graphs.erase(fe->graphs.begin() + lru);
        }
        orbit_extractor::GraphEntry e;
        e.key = key, e.stamp = ++fe->graph_clock;
        fe->graphs.push_back(e);
        return run_plan(fe, frames, B, film_gamma, film_beta, feats, workspace, s);
    }
    hit->stamp = ++fe->graph_clock;
    if (hit->exec == nullptr) {
        hipGraph_t graph = nullptr;
        if (fe->cap_stream == nullptr)
            ORBIT_HIP_CHECK(hipStreamCreateWithFlags(&fe->cap_stream, hipStreamNonBlocking));
        if (hipStreamBeginCapture(fe->cap_stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
            (void)hipGetLastError();
            return run_plan(fe, frames, B, film_gamma, film_beta, feats, workspace, s);  // capture unavailable: stay eager
        }
        const int rc = run_plan(fe, frames, B, film_gamma, film_beta, feats, workspace, fe->cap_stream);
        const hipError_t ce = hipStreamEndCapture(fe->cap_stream, &graph);
        if (rc != ORBIT_OK) {
            if (graph) (void)hipGraphDestroy(graph);
            return rc;
        }
        if (ce != hipSuccess || graph == nullptr) {
            (void)hipGetLastError();
            return run_plan(fe, frames, B, film_gamma, film_beta, feats, workspace, s);  // capture unavailable: stay eager
        }
        hipGraphExec_t exec = nullptr;
        const hipError_t ie = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (ie != hipSuccess || exec == nullptr) {
            (void)hipGetLastError();
            return run_plan(fe, frames, B, film_gamma, film_beta, feats, workspace, s);
        }
        hit->exec = exec;
    }
    ORBIT_HIP_CHECK(hipGraphLaunch(hit->exec, s));
    return ORBIT_OK;
}

static int run_plan(orbit_extractor_t* fe, const float* frames, int B, const float* film_gamma, const float* film_beta,
                    float* feats, void* workspace, hipStream_t s) {
    const WsLayout L = ws_layout(fe, B);
    char* ws = static_cast<char*>(workspace);
    auto buf = [&](int id) -> float* {
        if (id == -1) return const_cast<float*>(frames);
        if (id == 100) return feats;
        if (id == 101) return reinterpret_cast<float*>(ws + L.pooled);
        if (id == 102) return reinterpret_cast<float*>(ws + L.gate);
        return reinterpret_cast<float*>(ws + L.buf[id]);
    };
    const float* scale = fe->d_fold;
    const float* shift = fe->d_fold + fe->fold_floats;
    if (film_gamma && fe->film_size > 0) {
        float* fs = reinterpret_cast<float*>(ws + L.fold);
        dim3 grid((unsigned)fe->bns.size(), 2);
        const int rec = prof_start("bn_fold_film", 0.0, 4.0 * 6.0 * fe->fold_floats, s);
        bn_fold_all_kernel<<<grid, 256, 0, s>>>(fe->d_bn, fe->d_pool, film_gamma, film_beta, fs, fs + fe->fold_floats);
        prof_stop(rec, s);
        ORBIT_LAUNCH_CHECK();
        scale = fs, shift = fs + fe->fold_floats;
    }
    for (size_t oi = 0; oi < fe->ops.size(); ++oi) {
        const Op& o = fe->ops[oi];
        int rc = ORBIT_OK;
        switch (o.kind) {
            case OP_CONV: {
                if (o.x_nchw && !o.pool2 && o.res < 0 && o.bn >= 0 &&
                    stem_direct_supported(o.Cin, o.Cout, o.KH, o.stride, o.W, o.act) && o.KH == o.KW) {
                    // EfficientNet stem: LDS-staged input rows + VALU (csrc/stem.hip) instead of the element-wise gather
                    rc = launch_stem_direct(buf(o.in), fe->d_pool + fe->params[o.weight].off, scale + fe->bns[o.bn].fold_off,
                                            shift + fe->bns[o.bn].fold_off, buf(o.out), B, o.H, o.W, o.pad_t, o.pad_l, o.Ho,
                                            o.Wo, s);
                    break;
                }
                ConvDesc d;
                d.x = buf(o.in), d.w_packed = fe->d_packed + o.packed_off, d.y = buf(o.out);
                d.w_frag = o.frag_off != SIZE_MAX ? fe->d_packed + o.frag_off : nullptr;
                d.scale = o.bn >= 0 ? scale + fe->bns[o.bn].fold_off : nullptr;
                d.shift = o.bn >= 0 ? shift + fe->bns[o.bn].fold_off : nullptr;
                d.residual = o.res >= 0 ? buf(o.res) : nullptr;
                d.gate = o.use_gate ? buf(102) : nullptr;
                d.B = B, d.H = o.H, d.W = o.W, d.Cin = o.Cin, d.Cout = o.Cout, d.KH = o.KH, d.KW = o.KW;
                d.stride = o.stride, d.pad_t = o.pad_t, d.pad_l = o.pad_l, d.Ho = o.Ho, d.Wo = o.Wo;
                d.act = o.act, d.pool2 = o.pool2, d.x_nchw = o.x_nchw;
                d.splitk_ws = reinterpret_cast<float*>(ws + L.splitk);
                rc = launch_conv(d, s);
                break;
            }
            case OP_DWCONV:
                rc = launch_dwconv_se(buf(o.in), fe->d_packed + o.packed_off, buf(o.out),
                                      scale + fe->bns[o.bn].fold_off, shift + fe->bns[o.bn].fold_off,
                                      o.pool_partial ? buf(101) : nullptr, B, o.H, o.W, o.Cin, o.KH, o.stride, o.pad_t,
                                      o.pad_l, o.Ho, o.Wo, o.act, s);
                break;
            case OP_MBFRONT:
                if (o.stem)
                    rc = launch_stem_rows(buf(o.in), fe->d_packed + o.packed_off2, scale + fe->bns[o.bn].fold_off,
                                          shift + fe->bns[o.bn].fold_off, fe->d_packed + o.packed_off,
                                          scale + fe->bns[o.bn2].fold_off, shift + fe->bns[o.bn2].fold_off, buf(o.out),
                                          buf(101), B, o.stem_h, o.stem_w, o.stem_pt, o.stem_pl, o.H, o.W, s, o.se_chunks);
                else
                    rc = launch_mbconv_rows(buf(o.in), fe->d_pool + fe->params[o.weight].off,
                                            scale + fe->bns[o.bn].fold_off, shift + fe->bns[o.bn].fold_off,
                                            fe->d_packed + o.packed_off, scale + fe->bns[o.bn2].fold_off,
                                            shift + fe->bns[o.bn2].fold_off, buf(o.out), buf(101), B, o.H, o.W, o.Cin,
                                            o.Cout, o.KH, o.stride, o.pad_t, o.pad_l, o.Ho, o.Wo, s, o.se_chunks);
                break;
            case OP_MAXPOOL:
                rc = launch_maxpool(buf(o.in), buf(o.out), B, o.H, o.W, o.Cin, o.pool_k, o.stride, o.pool_pad, o.Ho,
                                    o.Wo, s);
                break;
            case OP_AVGPOOL:
                rc = launch_avgpool(buf(o.in), buf(o.out), B, o.H * o.W, o.Cin, s);
                break;
            case OP_SE:
                rc = launch_se_gate2(buf(101), o.se_chunks, o.se_hw, fe->d_pool + fe->params[o.se_w1].off,
                                     fe->d_pool + fe->params[o.se_b1].off, fe->d_packed + o.packed_off,
                                     fe->d_pool + fe->params[o.se_b2].off, buf(102), B, o.Cin, o.R, s);
                break;
        }
        if (rc != ORBIT_OK) return rc;
    }
    return ORBIT_OK;
}

}  // extern "C"
