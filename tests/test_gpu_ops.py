"""GPU parity of the single operators behind the C-ABI (orbit_op_*, head, pooler) against plain PyTorch-CPU fp32.

Every call goes through liborbit_hip.so via ctypes (raw device pointers + stream handle). Tolerances are
absolute, fp32: 2e-4 on O(1..30) conv outputs (K up to 4608 products), tighter on the elementwise kernels.
"""
import ctypes

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import orbit_dataset_amd  # noqa: E402,F401
from orbit_dataset_amd import _lib  # noqa: E402


def _st():
    return _lib.stream_handle()


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def run_conv(lib, device, x_cpu, w_cpu, stride, pad_t, pad_l, Ho, Wo, scale=None, shift=None, residual=None,
             gate=None, act=0, pool2=0, x_nchw=0):
    B, Cin, H, W = x_cpu.shape
    Cout, _, KH, KW = w_cpu.shape
    xin = (x_cpu if x_nchw else nhwc(x_cpu)).to(device)
    oh, ow = (Ho // 2, Wo // 2) if pool2 else (Ho, Wo)
    y = torch.full((B, oh, ow, Cout), float("nan"), device=device)
    d = lambda t: _lib.dptr(None if t is None else t.to(device).contiguous())
    res = None if residual is None else nhwc(residual)
    keep = [t.to(device).contiguous() if t is not None else None for t in (w_cpu, scale, shift, res, gate)]
    rc = lib.orbit_op_conv2d(_lib.dptr(xin), x_nchw, _lib.dptr(keep[0]), _lib.dptr(y), _lib.dptr(keep[1]),
                             _lib.dptr(keep[2]), _lib.dptr(keep[3]), _lib.dptr(keep[4]), B, H, W, Cin, Cout, KH, KW,
                             stride, pad_t, pad_l, Ho, Wo, act, pool2, _st())
    _lib.check(rc, "orbit_op_conv2d")
    torch.cuda.synchronize()
    return nchw(y.cpu())


def ref_conv(x, w, stride, pad_t, pad_l, Ho, Wo, scale=None, shift=None, residual=None, gate=None, act=0, pool2=0):
    B, Cin, H, W = x.shape
    KH, KW = w.shape[2:]
    if gate is not None:
        x = x * gate[:, :, None, None]
    pb = max((Ho - 1) * stride + KH - H - pad_t, 0)
    pr = max((Wo - 1) * stride + KW - W - pad_l, 0)
    y = F.conv2d(F.pad(x, [pad_l, pr, pad_t, pb]), w, None, stride)[:, :, :Ho, :Wo]
    if scale is not None:
        y = y * scale[None, :, None, None]
    if shift is not None:
        y = y + shift[None, :, None, None]
    if residual is not None:
        y = y + residual
    if act == 1:
        y = F.relu(y)
    elif act == 2:
        y = F.silu(y)
    if pool2:
        y = F.max_pool2d(y, 2, 2)
    return y


CONV_CASES = [
    # name, B, Cin, H, W, Cout, K, stride, pad, extras
    ("3x3_64_64", 4, 64, 21, 21, 64, 3, 1, 1, {}),
    ("3x3_s2_64_128", 3, 64, 21, 21, 128, 3, 2, 1, {}),
    ("1x1_s2_ds", 3, 64, 21, 21, 128, 1, 2, 0, {}),
    ("3x3_256_512_smallM", 5, 256, 6, 6, 512, 3, 2, 1, {}),
    ("3x3_512_512", 2, 512, 3, 3, 512, 3, 1, 1, {}),
    ("relu_bn_res", 4, 64, 11, 11, 64, 3, 1, 1, {"bn": True, "res": True, "act": 1}),
    ("bigM_128x128_tile", 40, 128, 28, 28, 256, 3, 1, 1, {"bn": True, "act": 1}),
    ("bigM_128x64_tile", 64, 64, 42, 42, 128, 1, 1, 0, {"bn": True}),
    ("pw_16_96_silu", 3, 16, 28, 28, 96, 1, 1, 0, {"bn": True, "act": 2}),
    ("pw_gate_96_24", 3, 96, 14, 14, 24, 1, 1, 0, {"bn": True, "gate": True}),
    ("pw_gate_res_240_40", 2, 240, 14, 14, 40, 1, 1, 0, {"bn": True, "gate": True, "res": True}),
    ("pw_1152_320", 2, 1152, 7, 7, 320, 1, 1, 0, {"bn": True}),
    ("pw_40_240_tailK", 2, 40, 14, 14, 240, 1, 1, 0, {"bn": True, "act": 2}),
    ("head_320_1280", 2, 320, 7, 7, 1280, 1, 1, 0, {"bn": True, "act": 2}),
    ("pool2_64_64", 3, 64, 42, 42, 64, 3, 1, 1, {"bn": True, "act": 1, "pool2": 1}),
    ("pool2_odd_21", 3, 64, 21, 21, 64, 3, 1, 1, {"bn": True, "act": 1, "pool2": 1}),
]


@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv_nhwc(lib, device, case):
    name, B, Cin, H, W, Cout, K, stride, pad, ex = case
    import zlib
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 10000)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, K, K, generator=g) / (Cin * K * K) ** 0.5
    Ho = (H + 2 * pad - K) // stride + 1
    Wo = (W + 2 * pad - K) // stride + 1
    scale = torch.rand(Cout, generator=g) + 0.5 if ex.get("bn") else None
    shift = torch.randn(Cout, generator=g) * 0.1 if ex.get("bn") else None
    res = torch.randn(B, Cout, Ho, Wo, generator=g) if ex.get("res") else None
    gate = torch.rand(B, Cin, generator=g) if ex.get("gate") else None
    kw = dict(scale=scale, shift=shift, residual=res, gate=gate, act=ex.get("act", 0), pool2=ex.get("pool2", 0))
    got = run_conv(lib, device, x, w, stride, pad, pad, Ho, Wo, **kw)
    want = ref_conv(x, w, stride, pad, pad, Ho, Wo, **kw)
    assert got.shape == want.shape
    assert not torch.isnan(got).any(), "kernel left outputs unwritten"
    err = (got - want).abs().max().item()
    assert err < 2e-4, f"{name}: max abs err {err}"


STEM_CASES = [
    ("resnet_stem_84", 3, 84, 64, 7, 2, 3, 3, 0, 1),
    ("resnet_stem_odd_37", 2, 37, 64, 7, 2, 3, 3, 0, 1),
    ("effnet_stem_same_64", 2, 64, 32, 3, 2, 0, 0, 0, 2),     # TF SAME, even input: pad only bottom/right
    ("effnet_stem_same_odd_33", 2, 33, 32, 3, 2, 1, 1, 0, 2),  # odd input: symmetric
    ("setenc_l1_pool", 3, 84, 64, 3, 1, 1, 1, 1, 1),
    ("setenc_l1_pool_32", 2, 32, 64, 3, 1, 1, 1, 1, 1),
]


@pytest.mark.parametrize("case", STEM_CASES, ids=[c[0] for c in STEM_CASES])
def test_conv_stem_nchw(lib, device, case):
    name, B, HW, Cout, K, stride, pad_t, pad_l, pool2, act = case
    g = torch.Generator().manual_seed(7)
    x = torch.randn(B, 3, HW, HW, generator=g)
    w = torch.randn(Cout, 3, K, K, generator=g) / (3 * K * K) ** 0.5
    if name.startswith("effnet"):
        Ho = Wo = -(-HW // stride)
    else:
        Ho = Wo = (HW + 2 * pad_t - K) // stride + 1
    scale = torch.rand(Cout, generator=g) + 0.5
    shift = torch.randn(Cout, generator=g) * 0.1
    got = run_conv(lib, device, x, w, stride, pad_t, pad_l, Ho, Wo, scale=scale, shift=shift, act=act, pool2=pool2,
                   x_nchw=1)
    want = ref_conv(x, w, stride, pad_t, pad_l, Ho, Wo, scale=scale, shift=shift, act=act, pool2=pool2)
    assert got.shape == want.shape and not torch.isnan(got).any()
    err = (got - want).abs().max().item()
    assert err < 1e-4, f"{name}: max abs err {err}"


def test_conv_transpose_detecting(lib, device):
    """Asymmetric weights + one-hot input: catches row/col or (kh,kw) swaps that random data could hide."""
    B, Cin, H, W, Cout = 1, 4, 5, 7, 4
    x = torch.zeros(B, Cin, H, W)
    x[0, 2, 1, 4] = 1.0
    w = torch.arange(Cout * Cin * 9, dtype=torch.float32).reshape(Cout, Cin, 3, 3)
    got = run_conv(lib, device, x, w, 1, 1, 1, H, W)
    want = F.conv2d(x, w, padding=1)
    assert torch.equal(got, want)


DW_CASES = [  # C, K, stride, HW, (dw_window, dw_lds, dw_pipe): every EfficientNet-B0 (channels, kernel, stride) combo on
    # the family the default rules give it (1, 1, 1), plus each family forced once on a shape the rules keep from it
    (32, 3, 1, 28, (1, 1, 1)),    # register window
    (96, 3, 2, 28, (1, 1, 1)),    # streaming
    (96, 3, 2, 56, (1, 1, 1)),    # software-pipelined streaming
    (144, 5, 2, 56, (1, 1, 1)),   # software-pipelined streaming, 5x5
    (144, 5, 2, 28, (1, 1, 1)),   # streaming, 5x5
    (480, 5, 1, 14, (1, 1, 1)),   # LDS patch
    (672, 5, 2, 14, (1, 1, 1)),   # streaming
    (1152, 3, 1, 7, (1, 1, 1)),   # LDS patch, 7x7
    (240, 3, 2, 15, (1, 1, 1)),   # streaming, odd map
    (32, 3, 1, 28, (0, 0, 0)),    # streaming forced (stride 1)
    (480, 5, 1, 14, (2, 0, 0)),   # register window forced on 5x5
    (96, 3, 2, 28, (0, 2, 0)),    # LDS patch forced on stride 2
    (1152, 3, 1, 7, (0, 0, 2)),   # pipelined forced on a 7x7 map
]


@pytest.mark.parametrize("C,K,stride,HW,opts", DW_CASES)
def test_dwconv(lib, device, C, K, stride, HW, opts):
    """the four depthwise kernels (streaming / register-window / LDS input patch / software-pipelined streaming)"""
    window, patch, pipe = opts
    lib.orbit_set_option(b"dw_window", window)
    lib.orbit_set_option(b"dw_lds", patch)
    lib.orbit_set_option(b"dw_pipe", pipe)
    g = torch.Generator().manual_seed(C + K)
    B = 3
    x = torch.randn(B, C, HW, HW, generator=g)
    w = torch.randn(C, 1, K, K, generator=g) / K
    scale = torch.rand(C, generator=g) + 0.5
    shift = torch.randn(C, generator=g) * 0.1
    Ho = -(-HW // stride)
    total = max((Ho - 1) * stride + K - HW, 0)
    pt = total // 2
    y = torch.full((B, Ho, Ho, C), float("nan"), device=device)
    xd, wd, sd, hd = nhwc(x).to(device), w.to(device), scale.to(device), shift.to(device)
    _lib.check(lib.orbit_op_dwconv2d(_lib.dptr(xd), _lib.dptr(wd), _lib.dptr(y), _lib.dptr(sd), _lib.dptr(hd), B, HW,
                                     HW, C, K, stride, pt, pt, Ho, Ho, 2, _st()), "dwconv")
    torch.cuda.synchronize()
    xp = F.pad(x, [pt, total - pt, pt, total - pt])
    want = F.silu(F.conv2d(xp, w, None, stride, 0, 1, C) * scale[None, :, None, None] + shift[None, :, None, None])
    lib.orbit_set_option(b"dw_window", 1)
    lib.orbit_set_option(b"dw_lds", 1)
    lib.orbit_set_option(b"dw_pipe", 1)
    err = (nchw(y.cpu()) - want).abs().max().item()
    assert err < 2e-5, err


@pytest.mark.parametrize("K,stride,pad,HW", [(3, 2, 1, 42), (3, 2, 1, 21), (2, 2, 0, 21)])
def test_maxpool(lib, device, K, stride, pad, HW):
    g = torch.Generator().manual_seed(3)
    B, C = 3, 64
    x = torch.randn(B, C, HW, HW, generator=g)
    Ho = (HW + 2 * pad - K) // stride + 1
    y = torch.full((B, Ho, Ho, C), float("nan"), device=device)
    xd = nhwc(x).to(device)
    _lib.check(lib.orbit_op_maxpool2d(_lib.dptr(xd), _lib.dptr(y), B, HW, HW, C, K, stride, pad, Ho, Ho, _st()), "maxpool")
    torch.cuda.synchronize()
    assert torch.equal(nchw(y.cpu()), F.max_pool2d(x, K, stride, pad))


@pytest.mark.parametrize("C,HW", [(512, 9), (1280, 49), (32, 12544), (100, 5)])
def test_avgpool(lib, device, C, HW):
    g = torch.Generator().manual_seed(C)
    B = 5
    x = torch.randn(B, HW, C, generator=g)
    y = torch.full((B, C), float("nan"), device=device)
    xd = x.to(device)
    _lib.check(lib.orbit_op_avgpool(_lib.dptr(xd), _lib.dptr(y), B, HW, C, _st()), "avgpool")
    torch.cuda.synchronize()
    assert (y.cpu() - x.mean(1)).abs().max().item() < 2e-6


@pytest.mark.parametrize("C,R", [(32, 8), (96, 4), (1152, 48)])
def test_se_gate(lib, device, C, R):
    g = torch.Generator().manual_seed(C)
    B = 7
    p = torch.randn(B, C, generator=g)
    w1, b1 = torch.randn(R, C, generator=g) / C ** 0.5, torch.randn(R, generator=g) * 0.1
    w2, b2 = torch.randn(C, R, generator=g) / R ** 0.5, torch.randn(C, generator=g) * 0.1
    out = torch.full((B, C), float("nan"), device=device)
    dev = [t.to(device) for t in (p, w1, b1, w2, b2)]
    _lib.check(lib.orbit_op_se_gate(*[_lib.dptr(t) for t in dev], _lib.dptr(out), B, C, R, _st()), "se_gate")
    torch.cuda.synchronize()
    want = torch.sigmoid(F.silu(p @ w1.t() + b1) @ w2.t() + b2)
    assert (out.cpu() - want).abs().max().item() < 2e-6


def test_mean_pool_and_set_mean(lib, device):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(25 * 8, 512, generator=g)
    xd = x.to(device)
    out = torch.empty(25, 512, device=device)
    _lib.check(lib.orbit_mean_pool(_lib.dptr(xd), 25, 8, 512, _lib.dptr(out), _st()), "mean_pool")
    m = torch.empty(512, device=device)
    _lib.check(lib.orbit_set_mean(_lib.dptr(xd), 200, 512, _lib.dptr(m), _st()), "set_mean")
    torch.cuda.synchronize()
    assert (out.cpu() - x.view(25, 8, 512).mean(1)).abs().max().item() < 1e-6
    assert (m.cpu() - x.mean(0)).abs().max().item() < 1e-6


MBROWS_CASES = [  # Cin, mid, K, stride, H, W: the five high-resolution block shapes of efficientnet_b0, at the 224x224
    # map sizes (full strips, several bands) and at odd sizes (ragged strips / bands, pad columns inside a row tile)
    (16, 96, 3, 2, 112, 112), (24, 144, 3, 1, 56, 56), (24, 144, 5, 2, 56, 56), (40, 240, 5, 1, 28, 28),
    (40, 240, 3, 2, 28, 28), (16, 96, 3, 2, 37, 21), (16, 80, 3, 2, 116, 58), (24, 144, 3, 1, 29, 58),
    (24, 100, 5, 2, 42, 31), (40, 240, 5, 1, 15, 29), (40, 240, 5, 1, 5, 8), (40, 236, 3, 2, 29, 30)]
# (maps whose step window is shorter than one MFMA row tile - e.g. 15x9 at 3x3/1, 6x3 at 3x3/2 - are not served by the fused
# kernel: orbit_op_mbconv_front_partials returns 0 for them and the network plans keep the conv + depthwise pair there)


@pytest.mark.parametrize("Cin,mid,K,stride,H,W", MBROWS_CASES)
def test_mbconv_front_row_streaming(lib, device, Cin, mid, K, stride, H, W):
    """csrc/mbconv_rows.hip: a block walks down a strip of the map, expanded rows in an LDS ring, input fragments straight
    from HBM. Against the unfused PyTorch-CPU sequence AND bit for bit against the library's conv + depthwise pair (same
    k-order, same tap order); pooling partials sum to the plane sums. The 224x224 map sizes run the branch-free (exact
    tiling) instantiation, the odd ones the predicated one; maps taller than 28 output rows have several bands (top-halo
    recompute, ragged last band)."""
    g = torch.Generator().manual_seed(Cin * 1000 + mid + K + stride + H)
    B = 3
    x = torch.randn(B, Cin, H, W, generator=g)
    w1 = torch.randn(mid, Cin, 1, 1, generator=g) / Cin ** 0.5
    wd = torch.randn(mid, 1, K, K, generator=g) / K
    s1, h1 = torch.rand(mid, generator=g) + 0.5, torch.randn(mid, generator=g) * 0.3
    s2, h2 = torch.rand(mid, generator=g) + 0.5, torch.randn(mid, generator=g) * 0.1
    Ho, Wo = -(-H // stride), -(-W // stride)
    ph, pw = max((Ho - 1) * stride + K - H, 0), max((Wo - 1) * stride + K - W, 0)
    e = F.silu(F.conv2d(x, w1) * s1[None, :, None, None] + h1[None, :, None, None])
    ep = F.pad(e, [pw // 2, pw - pw // 2, ph // 2, ph - ph // 2])
    want = F.silu(F.conv2d(ep, wd, None, stride, 0, 1, mid) * s2[None, :, None, None] + h2[None, :, None, None])
    dev = [t.to(device).contiguous() for t in (nhwc(x), w1, s1, h1, wd, s2, h2)]
    prev = lib.orbit_get_option(b"mbconv_rows")
    lib.orbit_set_option(b"mbconv_rows", 1)
    try:
        tiles = lib.orbit_op_mbconv_front_partials(H, W, Cin, mid, K, stride)
        assert tiles >= 1
        y = torch.full((B, Ho, Wo, mid), float("nan"), device=device)
        pool = torch.full((B, tiles, mid), float("nan"), device=device)
        _lib.check(lib.orbit_op_mbconv_front(*[_lib.dptr(t) for t in dev], _lib.dptr(y), _lib.dptr(pool), B, H, W, Cin, mid,
                                             K, stride, ph // 2, pw // 2, Ho, Wo, _st()), "mbconv_front (rows)")
        torch.cuda.synchronize()
    finally:
        lib.orbit_set_option(b"mbconv_rows", prev)
    got = nchw(y.cpu())
    assert not torch.isnan(got).any() and not torch.isnan(pool).any()
    assert (got - want).abs().max().item() < 5e-5
    assert (pool.cpu().sum(1) - want.sum((2, 3))).abs().max().item() < 2e-3 * max(1.0, want.sum((2, 3)).abs().max().item())
    # the unfused pair of the same library: identical bits
    ex = torch.empty(B, H, W, mid, device=device)
    y2 = torch.empty(B, Ho, Wo, mid, device=device)
    _lib.check(lib.orbit_op_conv2d(_lib.dptr(dev[0]), 0, _lib.dptr(dev[1]), _lib.dptr(ex), _lib.dptr(dev[2]), _lib.dptr(dev[3]),
                                   None, None, B, H, W, Cin, mid, 1, 1, 1, 0, 0, H, W, 2, 0, _st()), "conv2d")
    _lib.check(lib.orbit_op_dwconv2d(_lib.dptr(ex), _lib.dptr(dev[4]), _lib.dptr(y2), _lib.dptr(dev[5]), _lib.dptr(dev[6]), B, H,
                                     W, mid, K, stride, ph // 2, pw // 2, Ho, Wo, 2, _st()), "dwconv2d")
    torch.cuda.synchronize()
    assert torch.equal(y.cpu(), y2.cpu())


@pytest.mark.parametrize("Cin,mid,K,stride,H,W", [MBROWS_CASES[1], MBROWS_CASES[4]])
def test_mbconv_front_row_streaming_bf16x3(lib, device, Cin, mid, K, stride, H, W):
    """Opt-in `conv_bf3` inside the row-streaming fused front: the expand GEMM on the bf16 matrix cores with three-way split operands
    (exact-tiling instantiations; the 40 -> 240 3x3 / 2 shape keeps the fp32 form - register budget). Against the fp64 evaluation
    of the block front: the default kernel's bound, and an error no larger than 1.5x the default kernel's + 1e-6."""
    g = torch.Generator().manual_seed(Cin * 1000 + mid + K + stride + H)
    B = 2
    x = torch.randn(B, Cin, H, W, generator=g)
    w1 = torch.randn(mid, Cin, 1, 1, generator=g) / Cin ** 0.5
    wd = torch.randn(mid, 1, K, K, generator=g) / K
    s1, h1 = torch.rand(mid, generator=g) + 0.5, torch.randn(mid, generator=g) * 0.3
    s2, h2 = torch.rand(mid, generator=g) + 0.5, torch.randn(mid, generator=g) * 0.1
    Ho, Wo = -(-H // stride), -(-W // stride)
    ph, pw = max((Ho - 1) * stride + K - H, 0), max((Wo - 1) * stride + K - W, 0)
    d = lambda t: t.double()
    e = F.silu(F.conv2d(d(x), d(w1)) * d(s1)[None, :, None, None] + d(h1)[None, :, None, None])
    ep = F.pad(e, [pw // 2, pw - pw // 2, ph // 2, ph - ph // 2])
    want = F.silu(F.conv2d(ep, d(wd), None, stride, 0, 1, mid) * d(s2)[None, :, None, None] + d(h2)[None, :, None, None])
    dev = [t.to(device).contiguous() for t in (nhwc(x), w1, s1, h1, wd, s2, h2)]
    prev, prev3 = lib.orbit_get_option(b"mbconv_rows"), lib.orbit_get_option(b"conv_bf3")
    lib.orbit_set_option(b"mbconv_rows", 1)
    out = {}
    try:
        tiles = lib.orbit_op_mbconv_front_partials(H, W, Cin, mid, K, stride)
        for opt in (0, 2):  # bit 2 of the option = this kernel family
            lib.orbit_set_option(b"conv_bf3", opt)
            y = torch.full((B, Ho, Wo, mid), float("nan"), device=device)
            pool = torch.full((B, tiles, mid), float("nan"), device=device)
            _lib.check(lib.orbit_op_mbconv_front(*[_lib.dptr(t) for t in dev], _lib.dptr(y), _lib.dptr(pool), B, H, W, Cin, mid,
                                                 K, stride, ph // 2, pw // 2, Ho, Wo, _st()), "mbconv_front (rows)")
            torch.cuda.synchronize()
            out[opt] = (nchw(y.cpu()), pool.cpu())
    finally:
        lib.orbit_set_option(b"conv_bf3", prev3)
        lib.orbit_set_option(b"mbconv_rows", prev)
    (y0, p0), (y1, p1) = out[0], out[2]
    assert not torch.isnan(y1).any() and not torch.isnan(p1).any()
    e0, e1 = (y0.double() - want).abs().max().item(), (y1.double() - want).abs().max().item()
    scale = max(1.0, want.abs().max().item())
    assert e0 < 5e-5 * scale and e1 < 5e-5 * scale
    assert e1 <= 1.5 * e0 + 1e-6 * scale, (e1, e0)
    if (Cin, K, stride) == (40, 3, 2):
        assert torch.equal(y0, y1)      # this shape keeps the fp32 expand
    else:
        assert not torch.equal(y0, y1)  # the split form did run
    assert (p1.double().sum(1) - want.sum((2, 3))).abs().max().item() < 2e-3 * max(1.0, want.sum((2, 3)).abs().max().item())


@pytest.mark.parametrize("FH,FW", [(64, 64), (37, 45), (30, 30), (224, 224), (97, 131), (6, 5)])
def test_stem_dw_front_fused(lib, device, FH, FW):
    """stem conv (NCHW frames, 3x3 stride 2, TF-SAME) + BN + SiLU + depthwise 3x3 (SAME) + BN + SiLU in one row-streaming
    kernel (csrc/mbconv_rows.hip stem_rows_kernel), the stem output only in LDS, + SE pooling partials - against the unfused
    PyTorch-CPU sequence (odd sizes: the predicated instantiation; even ones: the branch-free one) and against the
    library's stem + depthwise kernel pair."""
    mid, rows = 32, 1
    g = torch.Generator().manual_seed(FH * 100 + FW + mid)
    B = 3
    x = torch.randn(B, 3, FH, FW, generator=g)
    ws = torch.randn(mid, 3, 3, 3, generator=g) / 27 ** 0.5
    wd = torch.randn(mid, 1, 3, 3, generator=g) / 3
    s1, h1 = torch.rand(mid, generator=g) + 0.5, torch.randn(mid, generator=g) * 0.3
    s2, h2 = torch.rand(mid, generator=g) + 0.5, torch.randn(mid, generator=g) * 0.1
    H, W = -(-FH // 2), -(-FW // 2)
    ph, pw = max((H - 1) * 2 + 3 - FH, 0), max((W - 1) * 2 + 3 - FW, 0)
    xp = F.pad(x, [pw // 2, pw - pw // 2, ph // 2, ph - ph // 2])
    e = F.silu(F.conv2d(xp, ws, None, 2) * s1[None, :, None, None] + h1[None, :, None, None])
    want = F.silu(F.conv2d(e, wd, None, 1, 1, 1, mid) * s2[None, :, None, None] + h2[None, :, None, None])
    dev = [t.to(device).contiguous() for t in (x, ws, s1, h1, wd, s2, h2)]
    prev = lib.orbit_get_option(b"mbconv_rows")
    lib.orbit_set_option(b"mbconv_rows", rows)
    try:
        tiles = lib.orbit_op_stem_dw_front_partials(H, W, mid)
        assert tiles >= 1
        y = torch.full((B, H, W, mid), float("nan"), device=device)
        pool = torch.full((B, tiles, mid), float("nan"), device=device)
        _lib.check(lib.orbit_op_stem_dw_front(*[_lib.dptr(t) for t in dev], _lib.dptr(y), _lib.dptr(pool), B, FH, FW, ph // 2,
                                              pw // 2, H, W, mid, 1, 1, H, W, _st()), "stem_dw_front")
        torch.cuda.synchronize()
    finally:
        lib.orbit_set_option(b"mbconv_rows", prev)
    got = nchw(y.cpu())
    assert not torch.isnan(got).any() and not torch.isnan(pool).any()
    assert (got - want).abs().max().item() < 5e-5
    assert (pool.cpu().sum(1) - want.sum((2, 3))).abs().max().item() < 2e-3 * max(1.0, want.sum((2, 3)).abs().max().item())
    if rows and FW >= 8:  # the unfused pair of the same library (direct stem kernel + depthwise): identical bits
        ex = torch.empty(B, H, W, mid, device=device)
        y2 = torch.empty(B, H, W, mid, device=device)
        _lib.check(lib.orbit_op_conv2d(_lib.dptr(dev[0]), 1, _lib.dptr(dev[1]), _lib.dptr(ex), _lib.dptr(dev[2]),
                                       _lib.dptr(dev[3]), None, None, B, FH, FW, 3, mid, 3, 3, 2, ph // 2, pw // 2, H, W, 2, 0,
                                       _st()), "conv2d (stem)")
        _lib.check(lib.orbit_op_dwconv2d(_lib.dptr(ex), _lib.dptr(dev[4]), _lib.dptr(y2), _lib.dptr(dev[5]), _lib.dptr(dev[6]),
                                         B, H, W, mid, 3, 1, 1, 1, H, W, 2, _st()), "dwconv2d")
        torch.cuda.synchronize()
        assert (y.cpu() - y2.cpu()).abs().max().item() < 1e-5  # same tap order; the FMA pairing differs


@pytest.mark.parametrize("Cin,Cout,K,HW,gated", [(256, 128, 3, 15, False), (512, 512, 3, 6, False), (1152, 192, 1, 7, True),
                                                 (672, 192, 1, 7, True)])
def test_conv_split_k(lib, device, Cin, Cout, K, HW, gated):
    """few output tiles + long reduction -> the split-K path (partial tiles + deterministic reduce with the fused
    epilogue): against the reference, against the unsplit kernel, and bit-identical run to run"""
    g = torch.Generator().manual_seed(Cin + Cout)
    B = 4
    x = torch.randn(B, Cin, HW, HW, generator=g)
    w = torch.randn(Cout, Cin, K, K, generator=g) / (Cin * K * K) ** 0.5
    kw = dict(scale=torch.rand(Cout, generator=g) + 0.5, shift=torch.randn(Cout, generator=g) * 0.1,
              residual=torch.randn(B, Cout, HW, HW, generator=g), gate=torch.rand(B, Cin, generator=g) if gated else None,
              act=2 if gated else 1)
    pad = K // 2
    want = ref_conv(x, w, 1, pad, pad, HW, HW, **kw)
    got = run_conv(lib, device, x, w, 1, pad, pad, HW, HW, **kw)
    again = run_conv(lib, device, x, w, 1, pad, pad, HW, HW, **kw)
    lib.orbit_set_option(b"conv_splitk", 0)
    try:
        unsplit = run_conv(lib, device, x, w, 1, pad, pad, HW, HW, **kw)
    finally:
        lib.orbit_set_option(b"conv_splitk", 1)
    assert torch.equal(got, again)
    assert (got - want).abs().max().item() < 2e-4
    assert (got - unsplit).abs().max().item() < 2e-5 and not torch.equal(got, unsplit)  # a different summation order ran


@pytest.mark.parametrize("tile", [0, 3, 4, 6])
def test_conv_random_shapes(lib, device, tile):
    """Forced tile configurations (conv_tile option; 0 = the launch heuristic; 6 = K-split waves) x 24 seeded random
    geometries through the implicit-GEMM kernel: every BK path (Cin % 32 / % 16 / % 8 / % 4), odd spatial sizes, strides 1-2, 1x1 / 3x3 / 5x5 taps, asymmetric (TF-SAME) padding, and random
    subsets of the fused epilogue features (BN, residual, SE gate, ReLU/SiLU, 2x2 pool)."""
    import random
    rnd = random.Random(20240928)
    lib.orbit_set_option(b"conv_tile", tile)
    try:
        _conv_random_cases(lib, device, rnd)
    finally:
        lib.orbit_set_option(b"conv_tile", 0)


RGEMM_CASES = [  # (Cin, Cout, (H, W), B, gated, residual, act)
    (80, 480, (14, 14), 5, False, False, 2),     # an expansion: SiLU epilogue, odd chunk count (5)
    (480, 80, (14, 14), 5, True, True, 0),       # a gated projection with skip connection, Cout = 5 tiles of 16
    (144, 40, (28, 28), 3, True, False, 0),      # Cout = 40: the last 16-channel tile is half full
    (1152, 320, (7, 7), 7, True, False, 0),      # the default class: 343 pixels = 10 tiles of 32 + 23 (ragged M, clamped rows), K slices
    (320, 1280, (7, 7), 3, False, False, 2),     # the head conv
    (16, 44, (9, 5), 2, False, False, 1),        # one chunk; Cout % 16 = 12
    (32, 64, (3, 3), 1, True, False, 0),         # fewer pixels than one tile, two chunks
    (160, 36, (11, 13), 4, False, True, 2),      # 10 chunks
]


@pytest.mark.parametrize("case", RGEMM_CASES, ids=["%dto%d_%dx%d" % (c[0], c[1], c[2][0], c[2][1]) for c in RGEMM_CASES])
def test_conv_pointwise_register_gemm(lib, device, case):
    """csrc/pw_rgemm.hip (pointwise convs as a barrier-free register GEMM on fragment-packed weights, transposed 16x16x4 MFMAs,
    float4 epilogue, K slices summed in slice order) against the fp32 reference and the LDS-tiled kernel: every EfficientNet
    epilogue form, ragged pixel counts, partly filled channel tiles, odd / even chunk counts, K slices."""
    Cin, Cout, (H, W), B, gated, res, act = case
    g = torch.Generator().manual_seed(Cin * 7 + Cout + H)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 1, 1, generator=g) / Cin ** 0.5
    kw = dict(scale=torch.rand(Cout, generator=g) + 0.5, shift=torch.randn(Cout, generator=g) * 0.1,
              residual=torch.randn(B, Cout, H, W, generator=g) if res else None,
              gate=torch.rand(B, Cin, generator=g) if gated else None, act=act)
    want = ref_conv(x.double(), w.double(), 1, 0, 0, H, W, **{k: (v.double() if torch.is_tensor(v) else v) for k, v in kw.items()})
    prev = lib.orbit_get_option(b"conv_rgemm")
    try:
        lib.orbit_set_option(b"conv_rgemm", 2)
        lib.orbit_prof_enable(1)
        got = run_conv(lib, device, x, w, 1, 0, 0, H, W, **kw)
        lib.orbit_prof_enable(0)
        lib.orbit_prof_collect(None, None, None)
        name = ctypes.create_string_buffer(48)
        lib.orbit_prof_variant(0, name, None, None, None, None)
        assert name.value.decode().startswith("conv_pw_rgemm<"), name.value  # the launch did take the register GEMM
        again = run_conv(lib, device, x, w, 1, 0, 0, H, W, **kw)
        lib.orbit_set_option(b"conv_rgemm", 0)
        igemm = run_conv(lib, device, x, w, 1, 0, 0, H, W, **kw)
    finally:
        lib.orbit_prof_enable(0)
        lib.orbit_set_option(b"conv_rgemm", prev)
    assert not torch.isnan(got).any()
    tol = 2e-5 * max(1.0, want.abs().max().item())
    assert (got.double() - want).abs().max().item() < tol
    assert (igemm.double() - want).abs().max().item() < tol
    assert torch.equal(got, again)  # deterministic (K slices are added in slice order)


BF3_CASES = [  # Cin, Cout, (H, W), B, gated, residual, act     (the opt-in path is frozen: four representative shapes)
    (144, 40, (28, 28), 3, True, False, 0),       # 64x64 tiles, K-tile 16, 9 K-tiles (odd), ragged last column tile
    (672, 112, (14, 14), 2, True, True, 0),       # 128x32 tiles, 21 K-tiles of 32 (odd), residual
    (1152, 320, (7, 7), 7, True, False, 0),       # 343 rows = 5 tiles of 64 + 23 (clamped rows), 36 K-tiles
    (96, 44, (9, 5), 2, False, True, 0),          # Cout % 16 = 12, three K-tiles
]


@pytest.mark.parametrize("case", BF3_CASES, ids=["%dto%d_%dx%d" % (c[0], c[1], c[2][0], c[2][1]) for c in BF3_CASES])
def test_conv_pointwise_bf16x3(lib, device, case):
    """csrc/conv_bf3.hip (opt-in `conv_bf3`): both operands split three ways into bf16, six bf16 x bf16 products per fp32 product
    on v_mfma_f32_32x32x16_bf16, fp32 accumulation. Against the fp64 evaluation of the same layer it must be AT LEAST as close as
    the default fp32-MFMA kernel (same 2e-5 bound, and an error no larger than 1.5x the fp32 kernel's + 1e-7), for every
    EfficientNet epilogue form, ragged row counts, both tile shapes, both K-tile widths; deterministic."""
    Cin, Cout, (H, W), B, gated, res, act = case
    g = torch.Generator().manual_seed(Cin * 5 + Cout + H)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 1, 1, generator=g) / Cin ** 0.5
    kw = dict(scale=torch.rand(Cout, generator=g) + 0.5, shift=torch.randn(Cout, generator=g) * 0.1,
              residual=torch.randn(B, Cout, H, W, generator=g) if res else None,
              gate=torch.rand(B, Cin, generator=g) if gated else None, act=act)
    want = ref_conv(x.double(), w.double(), 1, 0, 0, H, W, **{k: (v.double() if torch.is_tensor(v) else v) for k, v in kw.items()})
    prev, prev_rg = lib.orbit_get_option(b"conv_bf3"), lib.orbit_get_option(b"conv_rgemm")
    try:
        lib.orbit_set_option(b"conv_rgemm", 0)  # (its default class, 1152 -> 320 at 7x7, would keep that layer)
        lib.orbit_set_option(b"conv_bf3", 0)
        fp32 = run_conv(lib, device, x, w, 1, 0, 0, H, W, **kw)
        lib.orbit_set_option(b"conv_bf3", 1)
        lib.orbit_prof_enable(1)
        got = run_conv(lib, device, x, w, 1, 0, 0, H, W, **kw)
        lib.orbit_prof_enable(0)
        lib.orbit_prof_collect(None, None, None)
        name = ctypes.create_string_buffer(48)
        lib.orbit_prof_variant(0, name, None, None, None, None)
        assert name.value.decode().startswith("conv_bf3<"), name.value  # the launch did take the split kernel
        again = run_conv(lib, device, x, w, 1, 0, 0, H, W, **kw)
    finally:
        lib.orbit_prof_enable(0)
        lib.orbit_set_option(b"conv_bf3", prev)
        lib.orbit_set_option(b"conv_rgemm", prev_rg)
    assert not torch.isnan(got).any()
    scale = max(1.0, want.abs().max().item())
    e_bf3, e_fp32 = (got.double() - want).abs().max().item(), (fp32.double() - want).abs().max().item()
    assert e_bf3 < 2e-5 * scale and e_fp32 < 2e-5 * scale
    assert e_bf3 <= 1.5 * e_fp32 + 1e-7 * scale, (e_bf3, e_fp32)
    assert torch.equal(got, again)


BF3_GENERAL_CASES = [  # Cin, Cout, K, stride, (H, W), B, residual, act, TF-SAME padding
    (64, 64, 3, 1, (14, 14), 3, True, 1, False),      # resnet basic block conv2: residual + ReLU
    (64, 128, 1, 2, (14, 14), 2, False, 0, False),    # the 1x1 stride-2 shortcut
    (80, 48, 5, 2, (12, 12), 2, False, 2, True),      # 5x5 stride 2 with TF-SAME (asymmetric) padding, K-tile 16
]


@pytest.mark.parametrize("case", BF3_GENERAL_CASES, ids=["%dto%d_k%ds%d_%dx%d" % (c[0], c[1], c[2], c[3], c[4][0], c[4][1]) for c in BF3_GENERAL_CASES])
def test_conv_general_bf16x3(lib, device, case):
    """csrc/conv_bf3.hip, general form (opt-in `conv_bf3`): KxK taps, stride, zero / TF-SAME padding on the bf16 matrix cores with
    three-way split operands - resnet18's 3x3 and shortcut convs. Against the fp64 evaluation: the default kernel's bound, and an
    error no larger than 1.5x the default fp32-MFMA kernel's + 1e-7."""
    Cin, Cout, K, stride, (H, W), B, res, act, same = case
    if same:
        Ho, Wo = -(-H // stride), -(-W // stride)
        pt, pl = max((Ho - 1) * stride + K - H, 0) // 2, max((Wo - 1) * stride + K - W, 0) // 2
    else:
        pt = pl = K // 2
        Ho, Wo = (H + 2 * pt - K) // stride + 1, (W + 2 * pl - K) // stride + 1
    g = torch.Generator().manual_seed(Cin + 3 * Cout + K + H)
    x = torch.randn(B, Cin, H, W, generator=g).abs()  # (post-ReLU-like, non-zero mean)
    w = torch.randn(Cout, Cin, K, K, generator=g) / (Cin * K * K) ** 0.5
    kw = dict(scale=torch.rand(Cout, generator=g) + 0.5, shift=torch.randn(Cout, generator=g) * 0.1,
              residual=torch.randn(B, Cout, Ho, Wo, generator=g) if res else None, act=act)
    want = ref_conv(x.double(), w.double(), stride, pt, pl, Ho, Wo, **{k: (v.double() if torch.is_tensor(v) else v) for k, v in kw.items()})
    prev = lib.orbit_get_option(b"conv_bf3")
    try:
        lib.orbit_set_option(b"conv_bf3", 0)
        fp32 = run_conv(lib, device, x, w, stride, pt, pl, Ho, Wo, **kw)
        lib.orbit_set_option(b"conv_bf3", 1)
        lib.orbit_prof_enable(1)
        got = run_conv(lib, device, x, w, stride, pt, pl, Ho, Wo, **kw)
        lib.orbit_prof_enable(0)
        lib.orbit_prof_collect(None, None, None)
        name = ctypes.create_string_buffer(48)
        lib.orbit_prof_variant(0, name, None, None, None, None)
        assert name.value.decode().startswith("conv_bf3<") and ",pw" not in name.value.decode(), name.value
        again = run_conv(lib, device, x, w, stride, pt, pl, Ho, Wo, **kw)
    finally:
        lib.orbit_prof_enable(0)
        lib.orbit_set_option(b"conv_bf3", prev)
    assert not torch.isnan(got).any()
    scale = max(1.0, want.abs().max().item())
    e_bf3, e_fp32 = (got.double() - want).abs().max().item(), (fp32.double() - want).abs().max().item()
    assert e_bf3 < 2e-5 * scale and e_fp32 < 2e-5 * scale
    assert e_bf3 <= 1.5 * e_fp32 + 1e-7 * scale, (e_bf3, e_fp32)
    assert torch.equal(got, again)


def _conv_random_cases(lib, device, rnd):
    for case in range(24):
        Cin = rnd.choice([4, 8, 12, 16, 24, 40, 48, 64, 80, 96, 144, 160])
        Cout = rnd.choice([4, 8, 16, 24, 40, 64, 72, 128, 192, 320])
        K = rnd.choice([1, 1, 3, 3, 5])
        stride = rnd.choice([1, 1, 2])
        H, W = rnd.randint(3, 30), rnd.randint(3, 30)
        B = rnd.randint(1, 6)
        same = rnd.random() < 0.5
        if same:
            Ho, Wo = -(-H // stride), -(-W // stride)
            pt = max((Ho - 1) * stride + K - H, 0) // 2
            pl = max((Wo - 1) * stride + K - W, 0) // 2
        else:
            pad = K // 2
            pt = pl = pad
            Ho, Wo = (H + 2 * pad - K) // stride + 1, (W + 2 * pad - K) // stride + 1
        if Ho < 1 or Wo < 1:
            continue
        g = torch.Generator().manual_seed(case)
        x = torch.randn(B, Cin, H, W, generator=g)
        w = torch.randn(Cout, Cin, K, K, generator=g) / (Cin * K * K) ** 0.5
        pool2 = int(rnd.random() < 0.2 and Ho >= 2 and Wo >= 2)
        use_res = rnd.random() < 0.3 and not pool2
        use_gate = rnd.random() < 0.3 and not pool2
        kw = dict(scale=torch.rand(Cout, generator=g) + 0.5 if rnd.random() < 0.8 else None,
                  shift=torch.randn(Cout, generator=g) * 0.1 if rnd.random() < 0.8 else None,
                  residual=torch.randn(B, Cout, Ho, Wo, generator=g) if use_res else None,
                  gate=torch.rand(B, Cin, generator=g) if use_gate else None,
                  act=rnd.choice([0, 1, 2]), pool2=pool2)
        got = run_conv(lib, device, x, w, stride, pt, pl, Ho, Wo, **kw)
        want = ref_conv(x, w, stride, pt, pl, Ho, Wo, **kw)
        assert got.shape == want.shape, (case, got.shape, want.shape)
        err = (got - want).abs().max().item()
        assert err < 2e-4 and not torch.isnan(got).any(), (case, Cin, Cout, K, stride, H, W, B, pool2, err)


@pytest.mark.parametrize("B", [1, 3, 257])
def test_extractor_batch_sizes(device, B):
    """Batch sizes that do not fill a tile / exceed the reference's default batch_size of 256 in one call."""
    from oracle import extractors
    from orbit_dataset_amd import synthetic
    from orbit_dataset_amd.model.feature_extractors import create_feature_extractor
    ref = extractors.create("resnet18").eval()
    synthetic.init_parameters_(ref)
    fe, _ = create_feature_extractor("resnet18", True, False, False)
    fe.load_state_dict(ref.state_dict())
    fe = fe.cuda().eval()
    x = torch.randn(B, 3, 32, 32, generator=torch.Generator().manual_seed(B))
    with torch.no_grad():
        want = ref(x)
    got = fe(x.to(device)).cpu()
    assert (got - want).abs().max().item() < 2e-5 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("channels_last", [True, False])
@pytest.mark.parametrize("method", ["imagenet", "openai_clip"])
def test_frames_from_uint8_is_bit_identical_to_the_reference_transform(device, channels_last, method):
    """to_tensor (HWC uint8 -> CHW float / 255) + normalize((x - mean) / std), data/datasets.py:422-431."""
    from orbit_dataset_amd.data.utils import NORMALIZE_STATS, frames_from_uint8
    g = torch.Generator().manual_seed(3)
    hwc = torch.randint(0, 256, (5, 2, 37, 41, 3), generator=g, dtype=torch.uint8)  # [clips, T, H, W, 3]
    mean, std = (torch.tensor(v)[None, None, :, None, None] for v in NORMALIZE_STATS[method])
    want = (hwc.permute(0, 1, 4, 2, 3).float().div(255) - mean) / std
    src = hwc if channels_last else hwc.permute(0, 1, 4, 2, 3).contiguous()
    got = frames_from_uint8(src.pin_memory(), device, method, channels_last=channels_last)
    assert got.shape == want.shape and torch.equal(got.cpu(), want)
