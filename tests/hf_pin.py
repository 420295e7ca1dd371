"""Independent pin for the oracle's feature extractors.

The reference takes `tf_efficientnet_b0` from timm 0.6.12 (`/root/reference/model/feature_extractors.py:32,39-43`),
which is not installed offline; resnet18 is the torchvision network. Hugging Face `transformers` IS installed and
carries independently written implementations of both architectures (`EfficientNetModel`, a port of the TF/Keras
EfficientNet, and `ResNetModel(layer_type="basic")`). This module only re-keys an oracle `state_dict` into those
models: the arithmetic that produces the comparison values is entirely transformers' code.

Known, documented differences between HF EfficientNet and timm's `tf_efficientnet_b0` (both irrelevant to eval-mode
outputs at frame sizes that are multiples of 32, which is where the unmodified HF model is used):
  * HF pads stride-2 convolutions with a FIXED ZeroPad2d (0,1,0,1) / (k//2-1, k//2) chosen for even inputs; timm's
    Conv2dSame computes TF "SAME" padding from the input size. Equal whenever every stride-2 layer sees an even
    input (224, 192, 160, 128, 96, 64 ...). For other sizes `dynamic_same_padding=True` swaps HF's pad modules for
    TF-SAME padding computed from the input; that variant is a weaker (partly self-authored) check and labelled so.
  * HF uses TF's BatchNorm momentum convention (0.99) on some layers; for train-mode comparisons every BatchNorm's
    momentum is set to PyTorch's / timm's 0.1 here.
"""
import math

import torch
import torch.nn as nn

B0_STAGE_REPEATS = (1, 2, 2, 3, 3, 4, 1)


def _bn(dst, src):
    return {dst + "." + leaf: src + "." + leaf
            for leaf in ("weight", "bias", "running_mean", "running_var", "num_batches_tracked")}


def efficientnet_key_map():
    """HF EfficientNetModel key -> timm `tf_efficientnet_b0` key (the oracle's / the reference's names)."""
    m = {"embeddings.convolution.weight": "conv_stem.weight"}
    m.update(_bn("embeddings.batchnorm", "bn1"))
    b = 0
    for stage, reps in enumerate(B0_STAGE_REPEATS):
        for r in range(reps):
            hf, tm = "encoder.blocks.%d" % b, "blocks.%d.%d" % (stage, r)
            if stage == 0:  # DepthwiseSeparableConv: dw -> bn1 -> se -> pw -> bn2
                m[hf + ".depthwise_conv.depthwise_conv.weight"] = tm + ".conv_dw.weight"
                m.update(_bn(hf + ".depthwise_conv.depthwise_norm", tm + ".bn1"))
                m[hf + ".projection.project_conv.weight"] = tm + ".conv_pw.weight"
                m.update(_bn(hf + ".projection.project_bn", tm + ".bn2"))
            else:  # InvertedResidual: pw -> bn1 -> dw -> bn2 -> se -> pwl -> bn3
                m[hf + ".expansion.expand_conv.weight"] = tm + ".conv_pw.weight"
                m.update(_bn(hf + ".expansion.expand_bn", tm + ".bn1"))
                m[hf + ".depthwise_conv.depthwise_conv.weight"] = tm + ".conv_dw.weight"
                m.update(_bn(hf + ".depthwise_conv.depthwise_norm", tm + ".bn2"))
                m[hf + ".projection.project_conv.weight"] = tm + ".conv_pwl.weight"
                m.update(_bn(hf + ".projection.project_bn", tm + ".bn3"))
            for hf_se, tm_se in (("reduce", "conv_reduce"), ("expand", "conv_expand")):
                for leaf in ("weight", "bias"):
                    m["%s.squeeze_excite.%s.%s" % (hf, hf_se, leaf)] = "%s.se.%s.%s" % (tm, tm_se, leaf)
            b += 1
    m["encoder.top_conv.weight"] = "conv_head.weight"
    m.update(_bn("encoder.top_bn", "bn2"))
    return m


def resnet18_key_map():
    """HF ResNetModel(basic, [2,2,2,2]) key -> torchvision resnet18 key."""
    m = {"embedder.embedder.convolution.weight": "conv1.weight"}
    m.update(_bn("embedder.embedder.normalization", "bn1"))
    for s in range(4):
        for l in range(2):
            hf, tv = "encoder.stages.%d.layers.%d" % (s, l), "layer%d.%d" % (s + 1, l)
            for i in (0, 1):
                m["%s.layer.%d.convolution.weight" % (hf, i)] = "%s.conv%d.weight" % (tv, i + 1)
                m.update(_bn("%s.layer.%d.normalization" % (hf, i), "%s.bn%d" % (tv, i + 1)))
            if s > 0 and l == 0:
                m[hf + ".shortcut.convolution.weight"] = tv + ".downsample.0.weight"
                m.update(_bn(hf + ".shortcut.normalization", tv + ".downsample.1"))
    return m


class _SamePad(nn.Module):
    """TF "SAME" padding for a (kernel, stride) computed from the input size (timm Conv2dSame's rule)."""

    def __init__(self, k, s):
        super().__init__()
        self.k, self.s = k, s

    def forward(self, x):
        ih, iw = x.shape[-2:]
        ph = max((math.ceil(ih / self.s) - 1) * self.s + self.k - ih, 0)
        pw = max((math.ceil(iw / self.s) - 1) * self.s + self.k - iw, 0)
        return nn.functional.pad(x, [pw // 2, pw - pw // 2, ph // 2, ph - ph // 2])


def _load(model, key_map, state_dict):
    own = model.state_dict()
    assert set(own) == set(key_map), (sorted(set(own) ^ set(key_map))[:6], len(own), len(key_map))
    assert set(key_map.values()) == set(state_dict), sorted(set(key_map.values()) ^ set(state_dict))[:6]
    model.load_state_dict({k: state_dict[v].clone() for k, v in key_map.items()}, strict=True)
    return model


def hf_efficientnet_b0(state_dict, dynamic_same_padding=False):
    from transformers import EfficientNetConfig, EfficientNetModel
    cfg = EfficientNetConfig(width_coefficient=1.0, depth_coefficient=1.0, hidden_dim=1280, image_size=224,
                             hidden_act="swish", batch_norm_eps=1e-3, drop_connect_rate=0.0, dropout_rate=0.0)
    model = _load(EfficientNetModel(cfg), efficientnet_key_map(), state_dict)
    for mod in model.modules():
        if isinstance(mod, nn.BatchNorm2d):
            mod.momentum = 0.1
    if dynamic_same_padding:
        model.embeddings.padding = _SamePad(3, 2)
        for blk in model.encoder.blocks:
            dw = blk.depthwise_conv
            if dw.stride == 2:
                dw.depthwise_conv_pad = _SamePad(dw.depthwise_conv.kernel_size[0], 2)
    return model


def hf_resnet18(state_dict):
    from transformers import ResNetConfig, ResNetModel
    cfg = ResNetConfig(num_channels=3, embedding_size=64, hidden_sizes=[64, 128, 256, 512], depths=[2, 2, 2, 2],
                       layer_type="basic", hidden_act="relu", downsample_in_first_stage=False)
    return _load(ResNetModel(cfg), resnet18_key_map(), state_dict)


def hf_model(name, state_dict, dynamic_same_padding=False):
    if name == "efficientnet_b0":
        return hf_efficientnet_b0(state_dict, dynamic_same_padding)
    if name == "resnet18":
        assert not dynamic_same_padding
        return hf_resnet18(state_dict)
    raise ValueError(name)


def hf_features(model, x):
    """Pooled features [B, D] from transformers' own pooler."""
    out = model(pixel_values=x, return_dict=True).pooler_output
    return out.reshape(out.shape[0], -1)


def hf_batchnorm_state(model, key_map):
    """{oracle-named key: tensor} of every BatchNorm running statistic of the HF model."""
    sd = model.state_dict()
    return {v: sd[k] for k, v in key_map.items() if v.rsplit(".", 1)[-1] in ("running_mean", "running_var",
                                                                             "num_batches_tracked")}


KEY_MAPS = {"efficientnet_b0": efficientnet_key_map, "resnet18": resnet18_key_map}

# (case name, extractor, frame size, frames, HF variant) — sizes: BASELINE's 224 and 84, an odd size each, and for
# efficientnet a second multiple of 32 where the unmodified HF model applies.
CASES = [
    ("efficientnet_b0_224", "efficientnet_b0", 224, 2, "hf"),
    ("efficientnet_b0_96", "efficientnet_b0", 96, 4, "hf"),
    ("efficientnet_b0_231", "efficientnet_b0", 231, 2, "hf+same_pad"),
    ("efficientnet_b0_84", "efficientnet_b0", 84, 4, "hf+same_pad"),
    ("resnet18_224", "resnet18", 224, 2, "hf"),
    ("resnet18_84", "resnet18", 84, 4, "hf"),
    ("resnet18_97", "resnet18", 97, 4, "hf"),
    ("resnet18_231", "resnet18", 231, 2, "hf"),
]
TRAIN_CASES = [("efficientnet_b0_96", "efficientnet_b0", 96, 4), ("resnet18_84", "resnet18", 84, 4)]
TRAIN_STAT_KEYS = {
    "efficientnet_b0": ["bn1", "blocks.0.0.bn1", "blocks.1.0.bn2", "blocks.3.1.bn1", "blocks.5.2.bn3", "bn2"],
    "resnet18": ["bn1", "layer1.0.bn2", "layer2.0.downsample.1", "layer3.1.bn1", "layer4.1.bn2"],
}


def fixture_inputs(g):
    """Frames of a G12 fixture: stored as float16 (exactly representable), cropped per size."""
    big, small = g["x231"].astype("float32"), g["x97"].astype("float32")
    return {231: torch.from_numpy(big), 224: torch.from_numpy(big[:, :, :224, :224].copy()),
            97: torch.from_numpy(small), 96: torch.from_numpy(small[:, :, :96, :96].copy()),
            84: torch.from_numpy(small[:, :, 5:89, 7:91].copy())}
