"""ORACLE (test infrastructure) — PyTorch-CPU restatements of the two feature extractors.

The reference obtains these networks from timm==0.6.12 (`model/feature_extractors.py:31-33,39-43`), which is not
vendored; the layer arithmetic here is pinned against Hugging Face transformers' independent implementations of
both architectures (tests/test_oracle_extractors_hf.py, fixture G12). Module/parameter names follow
torchvision's `resnet18` and timm's `tf_efficientnet_b0` so that state_dicts interchange and the reference's
FiLM mechanism (functional_call with `<bn>.weight/.bias`, `model/few_shot_recognisers.py:114-115`) applies.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


# ---------------------------------------------------------------------------------------------------
# ResNet-18 (torchvision layout), num_classes=0 -> pooled 512-d features
# ---------------------------------------------------------------------------------------------------
class BasicBlock(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.relu = nn.ReLU(inplace=False)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x):
        identity = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return self.relu(out + identity)


class ResNet18(nn.Module):
    output_size = 512

    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=False)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        cin = 64
        for i, cout in enumerate((64, 128, 256, 512)):
            stride = 1 if i == 0 else 2
            setattr(self, f"layer{i + 1}", nn.Sequential(BasicBlock(cin, cout, stride), BasicBlock(cout, cout, 1)))
            cin = cout
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))

    def film_slot_names(self):
        """Every BatchNorm, in module-traversal order (build decision for the build-added resnet18)."""
        return [n for n, m in self.named_modules() if isinstance(m, nn.BatchNorm2d)]

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return torch.flatten(self.avgpool(x), 1)


# ---------------------------------------------------------------------------------------------------
# EfficientNet-B0, timm 0.6.12 `tf_efficientnet_b0` (TF "SAME" padding, BN eps 1e-3), num_classes=0
# ---------------------------------------------------------------------------------------------------
def _same_pad(x, k, s):
    ih, iw = x.shape[-2:]
    ph = max((math.ceil(ih / s) - 1) * s + k - ih, 0)
    pw = max((math.ceil(iw / s) - 1) * s + k - iw, 0)
    if ph or pw:
        x = F.pad(x, [pw // 2, pw - pw // 2, ph // 2, ph - ph // 2])
    return x


class Conv2dSame(nn.Conv2d):
    """TF-style SAME convolution: pad asymmetrically (extra on bottom/right), then VALID conv."""

    def __init__(self, cin, cout, k, stride=1, groups=1, bias=False):
        super().__init__(cin, cout, k, stride, 0, groups=groups, bias=bias)

    def forward(self, x):
        x = _same_pad(x, self.kernel_size[0], self.stride[0])
        return F.conv2d(x, self.weight, self.bias, self.stride, 0, 1, self.groups)


BN_EPS_TF = 1e-3


class SqueezeExcite(nn.Module):
    def __init__(self, chs, rd):
        super().__init__()
        self.conv_reduce = nn.Conv2d(chs, rd, 1, bias=True)
        self.conv_expand = nn.Conv2d(rd, chs, 1, bias=True)

    def forward(self, x):
        s = x.mean((2, 3), keepdim=True)
        s = self.conv_expand(F.silu(self.conv_reduce(s)))
        return x * torch.sigmoid(s)


class DepthwiseSeparableConv(nn.Module):
    def __init__(self, cin, cout, k, stride, rd):
        super().__init__()
        self.has_skip = stride == 1 and cin == cout
        self.conv_dw = Conv2dSame(cin, cin, k, stride, groups=cin)
        self.bn1 = nn.BatchNorm2d(cin, eps=BN_EPS_TF)
        self.se = SqueezeExcite(cin, rd)
        self.conv_pw = Conv2dSame(cin, cout, 1)
        self.bn2 = nn.BatchNorm2d(cout, eps=BN_EPS_TF)

    def forward(self, x):
        y = F.silu(self.bn1(self.conv_dw(x)))
        y = self.bn2(self.conv_pw(self.se(y)))
        return y + x if self.has_skip else y


class InvertedResidual(nn.Module):
    def __init__(self, cin, cout, k, stride, exp, rd):
        super().__init__()
        mid = cin * exp
        self.has_skip = stride == 1 and cin == cout
        self.conv_pw = Conv2dSame(cin, mid, 1)
        self.bn1 = nn.BatchNorm2d(mid, eps=BN_EPS_TF)
        self.conv_dw = Conv2dSame(mid, mid, k, stride, groups=mid)
        self.bn2 = nn.BatchNorm2d(mid, eps=BN_EPS_TF)
        self.se = SqueezeExcite(mid, rd)
        self.conv_pwl = Conv2dSame(mid, cout, 1)
        self.bn3 = nn.BatchNorm2d(cout, eps=BN_EPS_TF)

    def forward(self, x):
        y = F.silu(self.bn1(self.conv_pw(x)))
        y = F.silu(self.bn2(self.conv_dw(y)))
        y = self.bn3(self.conv_pwl(self.se(y)))
        return y + x if self.has_skip else y


class EfficientNetB0(nn.Module):
    output_size = 1280
    # (repeats, kernel, stride, expansion, out channels); SE reduction = round(0.25 * block input channels)
    ARCH = ((1, 3, 1, 1, 16), (2, 3, 2, 6, 24), (2, 5, 2, 6, 40), (3, 3, 2, 6, 80), (3, 5, 1, 6, 112),
            (4, 5, 2, 6, 192), (1, 3, 1, 6, 320))

    def __init__(self):
        super().__init__()
        self.conv_stem = Conv2dSame(3, 32, 3, 2)
        self.bn1 = nn.BatchNorm2d(32, eps=BN_EPS_TF)
        stages, cin = [], 32
        for si, (reps, k, s, e, cout) in enumerate(self.ARCH):
            blocks = []
            for r in range(reps):
                stride = s if r == 0 else 1
                rd = int(round(cin * 0.25))
                blocks.append(DepthwiseSeparableConv(cin, cout, k, stride, rd) if si == 0
                              else InvertedResidual(cin, cout, k, stride, e, rd))
                cin = cout
            stages.append(nn.Sequential(*blocks))
        self.blocks = nn.Sequential(*stages)
        self.conv_head = Conv2dSame(cin, 1280, 1)
        self.bn2 = nn.BatchNorm2d(1280, eps=BN_EPS_TF)

    def film_slot_names(self):
        """reference model/film.py:41-48: root bn1/bn2 and InvertedResidual.bn2 (DepthwiseSeparableConv: none)."""
        names = []
        for n, m in self.named_modules():
            if n in ("bn1", "bn2"):
                names.append(n)
            elif isinstance(m, InvertedResidual):
                names.append(n + ".bn2")
        # module-traversal order: bn1, blocks..., bn2
        return sorted(names, key=[k for k, _ in self.named_modules()].index)

    def forward(self, x):
        x = F.silu(self.bn1(self.conv_stem(x)))
        x = self.blocks(x)
        x = F.silu(self.bn2(self.conv_head(x)))
        return x.mean((2, 3))


def create(name):
    if name == "resnet18":
        return ResNet18()
    if name == "efficientnet_b0":
        return EfficientNetB0()
    raise ValueError(f"Invalid feature_extractor_name: {name}")
