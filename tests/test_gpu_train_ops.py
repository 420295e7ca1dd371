"""GPU parity of the training operators (orbit_op_bn_*, conv2d_dgrad / wgrad, maxpool / avgpool backward,
orbit_proto_predict_backward) against torch autograd on the CPU in fp32 (fp64 where the comparison itself would
otherwise be the larger error). Every call goes through liborbit_hip.so via ctypes.

Tolerances: the kernels accumulate in fp32 with fixed summation orders; measured against fp64 they are within
~2e-6 of the largest reference magnitude (tools/train_diag2.py), the checks allow 2e-5 (the fp32 CPU reference carries
its own ~1e-6).
"""
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


def rel_err(got, ref):
    return float((got.double() - ref.double()).abs().max() / max(float(ref.double().abs().max()), 1e-12))


def _dev(t, device):
    return None if t is None else t.to(device).contiguous()


# ---- BatchNorm train forward ----------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,C,H,W,act,res", [(8, 64, 14, 14, 1, True), (3, 128, 7, 5, 0, False), (5, 512, 3, 3, 1, True),
                                             (2, 24, 9, 9, 0, False), (4, 1280, 2, 2, 1, False), (64, 64, 28, 28, 1, True)])
def test_bn_train_forward(device, B, C, H, W, act, res):
    lib = _lib.load()
    g = torch.Generator().manual_seed(B * 131 + C)
    y = torch.randn(B, C, H, W, generator=g) * 1.7 + 0.6
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.2
    rm, rv = torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5
    residual = torch.randn(B, C, H, W, generator=g) if res else None
    rm_ref, rv_ref = rm.clone(), rv.clone()
    ref = F.batch_norm(y, rm_ref, rv_ref, gamma, beta, True, 0.1, 1e-5)
    if res:
        ref = ref + residual
    if act:
        ref = F.relu(ref)
    M = B * H * W
    yd = nhwc(y).to(device)
    out = torch.empty_like(yd)
    sm, si = torch.empty(C, device=device), torch.empty(C, device=device)
    rmd, rvd = rm.to(device), rv.to(device)
    keep = [_dev(gamma, device), _dev(beta, device), _dev(None if residual is None else nhwc(residual), device)]
    _lib.check(lib.orbit_op_bn_train_forward(_lib.dptr(yd), M, C, _lib.dptr(keep[0]), _lib.dptr(keep[1]), 1e-5, 0.1,
                                             _lib.dptr(rmd), _lib.dptr(rvd), _lib.dptr(keep[2]), act, _lib.dptr(out),
                                             _lib.dptr(sm), _lib.dptr(si), _st()), "bn_train_forward")
    torch.cuda.synchronize()
    assert (nchw(out.cpu()) - ref).abs().max() < 2e-5 * max(1.0, float(ref.abs().max()))
    assert (rmd.cpu() - rm_ref).abs().max() < 1e-6
    assert (rvd.cpu() - rv_ref).abs().max() < 1e-5
    mean = y.double().mean(dim=(0, 2, 3))
    var = y.double().var(dim=(0, 2, 3), unbiased=False)
    assert (sm.cpu().double() - mean).abs().max() < 1e-6
    assert ((si.cpu().double() - 1.0 / (var + 1e-5).sqrt()).abs() * (var + 1e-5).sqrt()).max() < 1e-5


# ---- BatchNorm backward -----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("train", [1, 0])
@pytest.mark.parametrize("B,C,H,W,act,res", [(8, 64, 14, 14, 1, True), (3, 128, 7, 5, 0, False), (5, 512, 3, 3, 1, True),
                                             (32, 64, 28, 28, 1, False)])
def test_bn_backward(device, train, B, C, H, W, act, res):
    lib = _lib.load()
    g = torch.Generator().manual_seed(B * 17 + C + train)
    y = (torch.randn(B, C, H, W, generator=g) * 1.3 + 0.4).requires_grad_(True)
    gamma = (torch.rand(C, generator=g) + 0.5).requires_grad_(True)
    beta = (torch.randn(C, generator=g) * 0.2).requires_grad_(True)
    residual = torch.randn(B, C, H, W, generator=g).requires_grad_(True) if res else None
    rm, rv = torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5
    out = F.batch_norm(y, rm.clone(), rv.clone(), gamma, beta, bool(train), 0.1, 1e-5)
    if res:
        out = out + residual
    if act:
        out = F.relu(out)
    dout = torch.randn(B, C, H, W, generator=g)
    out.backward(dout)
    if train:
        mean = y.detach().mean(dim=(0, 2, 3))
        invstd = 1.0 / (y.detach().var(dim=(0, 2, 3), unbiased=False) + 1e-5).sqrt()
    else:
        mean, invstd = rm, 1.0 / (rv + 1e-5).sqrt()
    M = B * H * W
    d = lambda t: _dev(t, device)
    t_dout, t_out, t_y = d(nhwc(dout)), d(nhwc(out.detach())), d(nhwc(y.detach()))
    t_gamma, t_mean, t_invstd = d(gamma.detach()), d(mean), d(invstd)
    dy = torch.empty_like(t_y)
    dres = torch.empty_like(t_y) if res else None
    dgamma, dbeta = torch.empty(C, device=device), torch.empty(C, device=device)
    _lib.check(lib.orbit_op_bn_backward(_lib.dptr(t_dout), _lib.dptr(t_out), _lib.dptr(t_y), M, C, _lib.dptr(t_gamma),
                                        _lib.dptr(t_mean), _lib.dptr(t_invstd), train, act, _lib.dptr(dy),
                                        _lib.dptr(dres), _lib.dptr(dgamma), _lib.dptr(dbeta), _st()), "bn_backward")
    torch.cuda.synchronize()
    assert rel_err(nchw(dy.cpu()), y.grad) < 2e-5
    assert rel_err(dgamma.cpu(), gamma.grad) < 2e-5
    assert rel_err(dbeta.cpu(), beta.grad) < 2e-5
    if res:
        assert rel_err(nchw(dres.cpu()), residual.grad) < 1e-6


# ---- convolution gradients -----------------------------------------------------------------------------------------------
CONV_CASES = [
    # B, Cin, H, W, Cout, K, stride, pad
    (4, 64, 21, 21, 64, 3, 1, 1),
    (3, 64, 21, 21, 128, 3, 2, 1),
    (3, 64, 22, 20, 128, 3, 2, 1),
    (3, 64, 21, 21, 128, 1, 2, 0),
    (2, 128, 11, 11, 256, 3, 1, 1),
    (2, 256, 6, 6, 512, 3, 2, 1),
    (2, 512, 3, 3, 512, 3, 1, 1),
    (5, 64, 9, 9, 64, 3, 1, 1),
    (2, 24, 10, 10, 40, 1, 1, 0),
    (16, 64, 28, 28, 64, 3, 1, 1),
]


def _conv_ref(B, Cin, H, W, Cout, K, stride, pad, seed, x_scale=1.0):
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn(B, Cin, H, W, generator=g) * x_scale).requires_grad_(True)
    w = (torch.randn(Cout, Cin, K, K, generator=g) / (Cin * K * K) ** 0.5).requires_grad_(True)
    y = F.conv2d(x, w, None, stride, pad)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    return x, w, y, dy


@pytest.mark.parametrize("case,accumulate", [(c, False) for c in CONV_CASES] + [(c, True) for c in CONV_CASES[1:4]],
                         ids=lambda v: "x".join(map(str, v)) if isinstance(v, tuple) else str(v))
def test_conv_dgrad(device, case, accumulate):
    lib = _lib.load()
    B, Cin, H, W, Cout, K, stride, pad = case
    x, w, y, dy = _conv_ref(*case, seed=sum(case))
    Ho, Wo = y.shape[2:]
    t_dy, t_w = nhwc(dy).to(device), w.detach().to(device).contiguous()
    acc_cpu = torch.randn(B, Cin, H, W, generator=torch.Generator().manual_seed(5)) if accumulate else None
    t_acc = None if acc_cpu is None else nhwc(acc_cpu).to(device)
    dx = torch.full((B, H, W, Cin), float("nan"), device=device)
    _lib.check(lib.orbit_op_conv2d_dgrad(_lib.dptr(t_dy), _lib.dptr(t_w), _lib.dptr(t_acc), _lib.dptr(dx), B, H, W, Cin,
                                         Cout, K, K, stride, pad, pad, Ho, Wo, _st()), "conv2d_dgrad")
    torch.cuda.synchronize()
    ref = x.grad + (acc_cpu if accumulate else 0)
    assert rel_err(nchw(dx.cpu()), ref) < 2e-5


# thin pointwise layers with >= 2^18 rows: the register form of csrc/conv_wgrad.hip (conv_wgrad_thin_kernel) - 1 / 2 thin tiles
# of 16 channels, the fat side as dY or as x, one and two groups of fat tiles, ragged channel counts, ragged last rows
WGRAD_THIN_CASES = [
    (21, 16, 112, 112, 96, 1, 1, 0),   # efficientnet_b0 block 1.0 expansion: fat = dY (6 tiles), 1 thin tile; 263 424 rows
    (84, 96, 56, 56, 24, 1, 1, 0),     # block 1.0 projection: fat = x, thin = 24 channels (2 tiles, the second half full)
    (85, 24, 56, 56, 144, 1, 1, 0),    # block 1.1 expansion: 9 fat tiles = two groups of 5 + 4; a ragged last row split
    (84, 136, 56, 56, 20, 1, 1, 0),    # fat = x with a ragged last tile, 20 thin channels
    (21, 32, 112, 112, 16, 1, 1, 0),   # block 0 projection
]


@pytest.mark.parametrize("case", CONV_CASES + WGRAD_THIN_CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv_wgrad(device, case):
    lib = _lib.load()
    B, Cin, H, W, Cout, K, stride, pad = case
    x, w, y, dy = _conv_ref(*case, seed=sum(case) + 1)
    Ho, Wo = y.shape[2:]
    t_x, t_dy = nhwc(x.detach()).to(device), nhwc(dy).to(device)
    dw = torch.full((Cout, Cin, K, K), float("nan"), device=device)
    _lib.check(lib.orbit_op_conv2d_wgrad(_lib.dptr(t_x), 0, _lib.dptr(t_dy), _lib.dptr(dw), B, H, W, Cin, Cout, K, K,
                                         stride, pad, pad, Ho, Wo, _st()), "conv2d_wgrad")
    torch.cuda.synchronize()
    assert rel_err(dw.cpu(), w.grad) < 2e-5


@pytest.mark.parametrize("B,Cin,HW,Cout", [(84, 96, 56, 24), (85, 144, 56, 24), (21, 32, 112, 16), (84, 24, 56, 144),
                                           (3, 672, 14, 112)])
def test_conv_wgrad_gated_projection(device, B, Cin, HW, Cout):
    """d/dW of conv1x1(x * gate) with the squeeze-excite gate multiplied in inside the kernel: the thin register form
    (gate of the fat / of the thin side kept in registers per frame; >= 2^18 rows) and the tiled form (last case)."""
    lib = _lib.load()
    g = torch.Generator().manual_seed(B + Cin + Cout)
    x = torch.randn(B, Cin, HW, HW, generator=g)
    gate = torch.rand(B, Cin, generator=g)
    w = (torch.randn(Cout, Cin, 1, 1, generator=g) / Cin ** 0.5).requires_grad_(True)
    y = F.conv2d(x * gate[:, :, None, None], w)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    t_x, t_g, t_dy = nhwc(x).to(device), gate.to(device), nhwc(dy).to(device)
    dw = torch.full((Cout, Cin, 1, 1), float("nan"), device=device)
    _lib.check(lib.orbit_op_conv2d_wgrad_gated(_lib.dptr(t_x), _lib.dptr(t_g), _lib.dptr(t_dy), _lib.dptr(dw), B, HW, HW, Cin,
                                               Cout, _st()), "conv2d_wgrad_gated")
    again = torch.empty_like(dw)
    _lib.check(lib.orbit_op_conv2d_wgrad_gated(_lib.dptr(t_x), _lib.dptr(t_g), _lib.dptr(t_dy), _lib.dptr(again), B, HW, HW,
                                               Cin, Cout, _st()), "conv2d_wgrad_gated")
    torch.cuda.synchronize()
    assert rel_err(dw.cpu(), w.grad) < 2e-5
    assert torch.equal(dw, again)  # deterministic: waves and blocks are added in a fixed order


@pytest.mark.parametrize("B,H,W,Cout,K,stride,pad", [(3, 32, 32, 64, 7, 2, 3), (2, 33, 29, 32, 3, 2, 1),
                                                     (4, 20, 20, 64, 3, 1, 1)])
def test_conv_wgrad_stem_nchw(device, B, H, W, Cout, K, stride, pad):
    lib = _lib.load()
    x, w, y, dy = _conv_ref(B, 3, H, W, Cout, K, stride, pad, seed=B + H + K)
    Ho, Wo = y.shape[2:]
    t_x, t_dy = x.detach().to(device).contiguous(), nhwc(dy).to(device)
    dw = torch.full((Cout, 3, K, K), float("nan"), device=device)
    _lib.check(lib.orbit_op_conv2d_wgrad(_lib.dptr(t_x), 1, _lib.dptr(t_dy), _lib.dptr(dw), B, H, W, 3, Cout, K, K,
                                         stride, pad, pad, Ho, Wo, _st()), "conv2d_wgrad(nchw)")
    torch.cuda.synchronize()
    assert rel_err(dw.cpu(), w.grad) < 2e-5


def test_conv_wgrad_is_deterministic(device):
    lib = _lib.load()
    case = (16, 64, 28, 28, 64, 3, 1, 1)
    B, Cin, H, W, Cout, K, stride, pad = case
    x, w, y, dy = _conv_ref(*case, seed=3)
    t_x, t_dy = nhwc(x.detach()).to(device), nhwc(dy).to(device)
    outs = []
    for _ in range(2):
        dw = torch.empty(Cout, Cin, K, K, device=device)
        _lib.check(lib.orbit_op_conv2d_wgrad(_lib.dptr(t_x), 0, _lib.dptr(t_dy), _lib.dptr(dw), B, H, W, Cin, Cout, K, K,
                                             stride, pad, pad, H, W, _st()), "conv2d_wgrad")
        outs.append(dw.cpu())
    assert torch.equal(outs[0], outs[1])


# ---- pooling ------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,C,H,W,K,stride,pad,relu", [(3, 64, 21, 21, 3, 2, 1, True), (2, 64, 16, 16, 2, 2, 0, True),
                                                       (2, 32, 15, 17, 2, 2, 0, False), (2, 64, 42, 42, 3, 2, 1, True)])
def test_maxpool_train_and_backward(device, B, C, H, W, K, stride, pad, relu):
    lib = _lib.load()
    g = torch.Generator().manual_seed(H * W + K)
    x = torch.randn(B, C, H, W, generator=g)
    if relu:
        x = F.relu(x)  # many exact ties at 0: exercises the first-maximum rule
    x.requires_grad_(True)
    y = F.max_pool2d(x, K, stride, pad)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    Ho, Wo = y.shape[2:]
    t_x = nhwc(x.detach()).to(device)
    t_y = torch.empty(B, Ho, Wo, C, device=device)
    idx = torch.empty(B, Ho, Wo, C, dtype=torch.uint8, device=device)
    _lib.check(lib.orbit_op_maxpool2d_train(_lib.dptr(t_x), _lib.dptr(t_y), _lib.dptr(idx, torch.uint8), B, H, W, C, K,
                                            stride, pad, Ho, Wo, _st()), "maxpool2d_train")
    t_dy = nhwc(dy).to(device)
    dx = torch.full((B, H, W, C), float("nan"), device=device)
    _lib.check(lib.orbit_op_maxpool2d_backward(_lib.dptr(t_dy), _lib.dptr(idx, torch.uint8), _lib.dptr(dx), B, H, W, C, K,
                                               stride, pad, Ho, Wo, _st()), "maxpool2d_backward")
    torch.cuda.synchronize()
    assert torch.equal(nchw(t_y.cpu()), y.detach())
    assert (nchw(dx.cpu()) - x.grad).abs().max() < 1e-6


def test_avgpool_backward(device):
    lib = _lib.load()
    B, HW, C = 5, 49, 512
    dy = torch.randn(B, C)
    dx = torch.empty(B, HW, C, device=device)
    t = dy.to(device)
    _lib.check(lib.orbit_op_avgpool_backward(_lib.dptr(t), _lib.dptr(dx), B, HW, C, _st()), "avgpool_backward")
    torch.cuda.synchronize()
    assert (dx.cpu() - (dy / HW)[:, None, :]).abs().max() < 1e-7


# ---- head backward ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cosine", [0, 1])
@pytest.mark.parametrize("M,T,D,C,scale", [(200, 1, 1280, 5, 1.0), (37, 1, 512, 10, 32.0), (12, 4, 512, 5, 1.0),
                                           (9, 3, 130, 7, 2.0)])
def test_proto_predict_backward(device, cosine, M, T, D, C, scale):
    lib = _lib.load()
    g = torch.Generator().manual_seed(M + D + C)
    feats = torch.randn(M * T, D, generator=g).requires_grad_(True)
    Wt = torch.randn(C, D, generator=g)
    q = feats.view(M, T, D).mean(dim=1)
    if cosine:
        logits = scale * F.cosine_similarity(q[:, :, None], Wt.t()[None, :, :], dim=1)
    else:
        b = -(Wt / 2).pow(2).sum(dim=1)
        logits = scale * (q @ Wt.t() + b)
    dl = torch.randn(M, C, generator=g)
    logits.backward(dl)
    t_dl, t_f, t_w = dl.to(device), feats.detach().to(device), Wt.to(device)
    df = torch.full((M * T, D), float("nan"), device=device)
    _lib.check(lib.orbit_proto_predict_backward(_lib.dptr(t_dl), _lib.dptr(t_f), _lib.dptr(t_w), M, T, D, C, scale, cosine,
                                                _lib.dptr(df), _st()), "proto_predict_backward")
    torch.cuda.synchronize()
    assert rel_err(df.cpu(), feats.grad) < 2e-5


# ---- MBConv pieces: depthwise convolution and squeeze-excite backward -----------------------------------------------
DW_CASES = [
    # B, C, H, W, K, stride, pad_top/left, pad_bottom/right
    (3, 32, 14, 14, 3, 1, 1, 1),
    (2, 96, 16, 16, 3, 2, 0, 1),     # TF "SAME" at an even size: all padding on the bottom/right
    (2, 144, 15, 13, 5, 2, 2, 2),
    (2, 240, 9, 9, 5, 1, 2, 2),
    (4, 24, 20, 20, 5, 2, 1, 2),     # TF "SAME": 1 before, 2 after
    (2, 1152, 4, 4, 3, 1, 1, 1),
    (2, 40, 11, 9, 3, 2, 1, 1),      # stride 2, odd map, padding parity 1 (the 2x2-block data-gradient kernel's 4th variant)
    (3, 672, 14, 14, 5, 2, 1, 2),    # efficientnet_b0 block 5.0's depthwise layer
    (1, 96, 112, 112, 3, 2, 0, 1),   # block 1.0's (the largest tensor of the network)
    (2, 48, 30, 28, 3, 1, 1, 1),     # 3x3 stride 1 above 14 rows: the register-window kernel form
]


@pytest.mark.parametrize("case", DW_CASES, ids=lambda c: "x".join(map(str, c)))
def test_dwconv_backward(device, case):
    lib = _lib.load()
    B, C, H, W, K, stride, p0, p1 = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, C, H, W, generator=g).requires_grad_(True)
    w = (torch.randn(C, 1, K, K, generator=g) / K).requires_grad_(True)
    y = F.conv2d(F.pad(x, [p0, p1, p0, p1]), w, None, stride, 0, 1, C)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    Ho, Wo = y.shape[2:]
    t_x, t_w, t_dy = nhwc(x.detach()).to(device), w.detach().to(device).contiguous(), nhwc(dy).to(device)
    dx = torch.full((B, H, W, C), float("nan"), device=device)
    dw = torch.full((C, 1, K, K), float("nan"), device=device)
    _lib.check(lib.orbit_op_dwconv2d_backward(_lib.dptr(t_x), _lib.dptr(t_w), _lib.dptr(t_dy), _lib.dptr(dx), _lib.dptr(dw),
                                              B, H, W, C, K, stride, p0, p0, Ho, Wo, _st()), "dwconv2d_backward")
    torch.cuda.synchronize()
    assert rel_err(nchw(dx.cpu()), x.grad) < 2e-5
    assert rel_err(dw.cpu(), w.grad) < 2e-5


@pytest.mark.parametrize("case", DW_CASES, ids=lambda c: "x".join(map(str, c)))
@pytest.mark.parametrize("in_act", [2, 1])
def test_dwconv_wgrad_with_input_transform(device, case, in_act):
    """The LDS form of the depthwise filter gradient with the preceding BatchNorm + activation applied as the input patch is
    staged (the taped LITE forward never writes the activated tensor): against autograd through act(x * scale + shift) ->
    depthwise conv, incl. TF-SAME padding (the padding of the ACTIVATED tensor is zero), ragged chunks, every (K, stride)."""
    lib = _lib.load()
    B, C, H, W, K, stride, p0, p1 = case
    if (C // 4) % 4 != 0:
        pytest.skip("the LDS form stages slices of 4 / 8 / 16 channel quads (every depthwise layer of efficientnet_b0); a plan "
                    "whose layer it does not fit keeps the separate activation pass")
    g = torch.Generator().manual_seed(sum(case) + in_act)
    x = torch.randn(B, C, H, W, generator=g) * 1.5
    sc, sh = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3 + 0.2
    w = (torch.randn(C, 1, K, K, generator=g) / K).requires_grad_(True)
    a = x * sc[None, :, None, None] + sh[None, :, None, None]
    a = F.silu(a) if in_act == 2 else F.relu(a)
    y = F.conv2d(F.pad(a, [p0, p1, p0, p1]), w, None, stride, 0, 1, C)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    Ho, Wo = y.shape[2:]
    t_x, t_dy, t_sc, t_sh = nhwc(x).to(device), nhwc(dy).to(device), sc.to(device), sh.to(device)
    dw = torch.full((C, 1, K, K), float("nan"), device=device)
    _lib.check(lib.orbit_op_dwconv2d_wgrad_xf(_lib.dptr(t_x), _lib.dptr(t_sc), _lib.dptr(t_sh), in_act, _lib.dptr(t_dy),
                                              _lib.dptr(dw), B, H, W, C, K, stride, p0, p0, Ho, Wo, _st()), "dwconv2d_wgrad_xf")
    torch.cuda.synchronize()
    assert rel_err(dw.cpu(), w.grad) < 2e-5


@pytest.mark.parametrize("case", DW_CASES, ids=lambda c: "x".join(map(str, c)))
@pytest.mark.parametrize("act", [2, 0])
def test_dwconv_dgrad_with_batchnorm_epilogue(device, case, act):
    """The depthwise data gradient that goes on through the producer's SiLU and leaves the sums of that producer's BatchNorm
    backward (sum g, sum g * xhat): against autograd through act(y * scale + shift) -> depthwise conv, every (K, stride),
    TF-SAME padding, ragged chunks. A layer no kernel form with the epilogue fits reports ORBIT_ERR_ARG (the plan then
    keeps the separate reduction pass)."""
    lib = _lib.load()
    B, C, H, W, K, stride, p0, p1 = case
    g = torch.Generator().manual_seed(sum(case) + 7 * act)
    yr = (torch.randn(B, C, H, W, generator=g) * 1.5).requires_grad_(True)
    sc, sh = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3 + 0.2
    mean, invstd = torch.randn(C, generator=g) * 0.2, torch.rand(C, generator=g) + 0.5
    w = (torch.randn(C, 1, K, K, generator=g) / K).requires_grad_(True)
    z = yr * sc[None, :, None, None] + sh[None, :, None, None]
    a = F.silu(z) if act == 2 else z
    out = F.conv2d(F.pad(a, [p0, p1, p0, p1]), w, None, stride, 0, 1, C)
    dy = torch.randn(out.shape, generator=g)
    out.backward(dy)
    gz = (yr.grad / sc[None, :, None, None]).double()  # d loss / d z
    xhat = ((yr.detach() - mean[None, :, None, None]) * invstd[None, :, None, None]).double()
    want_sums = torch.stack([gz.sum(dim=(0, 2, 3)), (gz * xhat).sum(dim=(0, 2, 3))])
    Ho, Wo = out.shape[2:]
    dev = lambda t: t.detach().float().to(device).contiguous()
    t_dy, t_w, t_y = dev(nhwc(dy)), dev(w), dev(nhwc(yr.detach()))
    t_m, t_i, t_sc, t_sh = dev(mean), dev(invstd), dev(sc), dev(sh)
    gout = torch.full((B, H, W, C), float("nan"), device=device)
    sums = torch.full((2, C), float("nan"), device=device)
    # (the stride-2 kernel and the register-window kernel of the 3x3 maps above 14 rows carry the filter gradient)
    dw = torch.full((C, 1, K, K), float("nan"), device=device) if stride == 2 or (K == 3 and H > 14) else None
    rc = lib.orbit_op_dwconv2d_dgrad_bn(_lib.dptr(t_dy), _lib.dptr(t_w), _lib.dptr(t_y), _lib.dptr(t_m), _lib.dptr(t_i),
                                        _lib.dptr(t_sc), _lib.dptr(t_sh), act, _lib.dptr(gout), _lib.dptr(sums),
                                        _lib.dptr(dw) if dw is not None else None, B, H, W, C, K, stride, p0, p0, Ho, Wo, _st())
    if rc == -1:
        pytest.skip("no kernel form with the epilogue for this layer: " + _lib.last_error())
    _lib.check(rc, "dwconv2d_dgrad_bn")
    torch.cuda.synchronize()
    assert rel_err(gout.cpu(), nhwc(gz.float())) < 2e-5
    assert rel_err(sums.cpu().double(), want_sums) < 2e-5
    if dw is not None:
        assert rel_err(dw.cpu(), w.grad) < 2e-5


@pytest.mark.parametrize("B,HW,C,R", [(5, 49, 96, 4), (3, 16, 1152, 48), (2, 196, 144, 6), (7, 9, 32, 8)])
def test_se_gate_backward(device, B, HW, C, R):
    lib = _lib.load()
    g = torch.Generator().manual_seed(B * C + R)
    x = torch.randn(B, HW, C, generator=g, dtype=torch.float64).requires_grad_(True)
    w1 = (torch.randn(R, C, generator=g, dtype=torch.float64) / C ** 0.5).requires_grad_(True)
    b1 = (0.1 * torch.randn(R, generator=g, dtype=torch.float64)).requires_grad_(True)
    w2 = (torch.randn(C, R, generator=g, dtype=torch.float64) / R ** 0.5).requires_grad_(True)
    b2 = (0.1 * torch.randn(C, generator=g, dtype=torch.float64)).requires_grad_(True)
    pooled = x.mean(dim=1)
    gate = torch.sigmoid(F.silu(pooled @ w1.t() + b1) @ w2.t() + b2)
    xg = x * gate[:, None, :]
    dxg = torch.randn(B, HW, C, generator=g, dtype=torch.float64)
    xg.backward(dxg)
    f = lambda t: t.detach().float().to(device).contiguous()
    ts = [f(dxg), f(x), f(pooled), f(w1), f(b1), f(w2), f(b2)]
    dx = torch.full((B, HW, C), float("nan"), device=device)
    outs = [torch.empty(R, C, device=device), torch.empty(R, device=device), torch.empty(C, R, device=device),
            torch.empty(C, device=device)]
    _lib.check(lib.orbit_op_se_gate_backward(*[_lib.dptr(t) for t in ts], _lib.dptr(dx), *[_lib.dptr(t) for t in outs], B,
                                             HW, C, R, _st()), "se_gate_backward")
    torch.cuda.synchronize()
    assert rel_err(dx.cpu(), x.grad) < 2e-5
    for got, ref in zip(outs, (w1, b1, w2, b2)):
        assert rel_err(got.cpu(), ref.grad) < 5e-5


# ---- round 3: train-mode BatchNorm statistics from the producing kernels, activation applied on load --------------------
CONV_TRAIN_CASES = [
    # B, H, W, Cin, Cout, K, stride, pad, nchw, gate
    (3, 14, 14, 80, 480, 1, 1, 0, 0, False),     # efficientnet expand: 64x64 tiles, BK 16
    (2, 14, 14, 480, 80, 1, 1, 0, 0, True),      # gated projection: 128x32 tiles, Cout not a multiple of 32
    (2, 14, 14, 672, 112, 1, 1, 0, 0, True),
    (3, 9, 11, 24, 144, 1, 1, 0, 0, False),      # M = 297 rows: ragged last row block, BK 8
    (2, 7, 7, 1152, 320, 1, 1, 0, 0, True),
    (1, 56, 56, 16, 96, 1, 1, 0, 0, False),      # 3136 rows
    (2, 20, 20, 64, 64, 3, 1, 1, 0, False),      # resnet18 3x3
    (2, 21, 21, 64, 128, 3, 2, 1, 0, False),
    (2, 64, 64, 3, 32, 3, 2, 0, 1, False),       # NCHW stem gather
]


@pytest.mark.parametrize("case", CONV_TRAIN_CASES, ids=lambda c: "x".join(str(int(v)) for v in c))
def test_conv_emits_batchnorm_statistics(device, case):
    """conv_igemm's epilogue (ConvDesc::stats): per-channel sums / sums of squares of the RAW outputs, one partial per row
    block, equal to the sums of the tensor it wrote; the gate variant multiplies x * gate[b][ci] on the fly."""
    import ctypes
    lib = _lib.load()
    B, H, W, Cin, Cout, K, stride, pad, nchw_in, gated = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, K, K, generator=g) / (Cin * K * K) ** 0.5
    gate = torch.rand(B, Cin, generator=g) if gated else None
    if nchw_in:  # TF-SAME stem: output ceil(H / 2), padding on the bottom / right
        Ho = -(-H // stride)
        tot = max((Ho - 1) * stride + K - H, 0)
        xin = F.pad(x, [0, tot, 0, tot])
        pt = 0
    else:
        Ho = (H + 2 * pad - K) // stride + 1
        xin = F.pad(x, [pad] * 4)
        pt = pad
    Wo = Ho if H == W else (W + 2 * pad - K) // stride + 1
    xg = xin * gate[:, :, None, None] if gated else xin
    want = F.conv2d(xg.double(), w.double(), None, stride)
    assert want.shape[2:] == (Ho, Wo)
    xd = x.contiguous().to(device) if nchw_in else nhwc(x).to(device)
    y = torch.full((B, Ho, Wo, Cout), float("nan"), device=device)
    stats = torch.full((2, Cout), float("nan"), device=device)
    nblk = ctypes.c_int(-1)
    wd, gd = w.to(device), (gate.to(device) if gated else None)  # named: a temporary would be freed (and its block re-used
    #                                                              by the next upload) before the launch reads it
    _lib.check(lib.orbit_op_conv2d_train(_lib.dptr(xd), nchw_in, _lib.dptr(wd), _lib.dptr(y),
                                         _lib.dptr(gd) if gated else None, B, H, W, Cin, Cout, K, K, stride, pt,
                                         pt, Ho, Wo, _lib.dptr(stats), ctypes.byref(nblk), _st()), "conv2d_train")
    torch.cuda.synchronize()
    assert rel_err(nchw(y.cpu()), want) < 2e-5
    assert nblk.value >= -(-B * Ho * Wo // 128), nblk.value  # one partial per 32..128-row block
    s = want.sum(dim=(0, 2, 3)), (want * want).sum(dim=(0, 2, 3))
    assert rel_err(stats[0].cpu(), s[0]) < 1e-5 * max(1.0, float(s[0].abs().max() and want.abs().sum() / s[0].abs().max()))
    assert rel_err(stats[1].cpu(), s[1]) < 2e-5
    # and exactly the sums of what the kernel wrote (fp32, double accumulation of the partials)
    yc = y.cpu().double().reshape(-1, Cout)
    assert rel_err(stats[1].cpu(), (yc * yc).sum(0)) < 2e-6


_DW_SHAPES = [(32, 3, 1, 28), (96, 3, 2, 28), (96, 3, 2, 56), (144, 5, 2, 28), (480, 5, 1, 14), (672, 5, 2, 14),
              (1152, 3, 1, 7), (240, 3, 2, 15)]
# every EfficientNet-B0 depthwise shape on the family the default rules give it, raw and with the SiLU input transform; each
# family forced once on a shape the rules keep from it; the ReLU input transform twice
DW_TRAIN_CASES = ([s + (1, 1, 1, a) for s in _DW_SHAPES for a in (None, 2)] +
                  [(32, 3, 1, 28, 0, 0, 0, 2), (480, 5, 1, 14, 2, 0, 0, 2), (96, 3, 2, 28, 0, 2, 0, 2), (1152, 3, 1, 7, 0, 0, 2, 2),
                   (480, 5, 1, 14, 1, 1, 1, 1), (96, 3, 2, 56, 1, 1, 1, 1)])


@pytest.mark.parametrize("C,K,stride,HW,window,patch,pipe,in_act", DW_TRAIN_CASES)
def test_dwconv_train_form(device, C, K, stride, HW, window, patch, pipe, in_act):
    """The four depthwise kernels in their training form: statistics of the raw output (STATS) and, with in_act, the
    preceding BatchNorm + activation applied while loading the raw conv output (XF) - the zero padding of the ACTIVATED
    tensor must stay zero."""
    lib = _lib.load()
    lib.orbit_set_option(b"dw_window", window)
    lib.orbit_set_option(b"dw_lds", patch)
    lib.orbit_set_option(b"dw_pipe", pipe)
    try:
        g = torch.Generator().manual_seed(C + K + stride)
        B = 3
        x = torch.randn(B, C, HW, HW, generator=g) * 1.5
        w = torch.randn(C, 1, K, K, generator=g) / K
        Ho = -(-HW // stride)
        total = max((Ho - 1) * stride + K - HW, 0)
        pt = total // 2
        sc, sh = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3 + 0.2
        a = x.double()
        if in_act is not None:
            a = a * sc.double()[None, :, None, None] + sh.double()[None, :, None, None]
            a = F.silu(a) if in_act == 2 else F.relu(a)
        want = F.conv2d(F.pad(a, [pt, total - pt, pt, total - pt]), w.double(), None, stride, 0, 1, C)
        y = torch.full((B, Ho, Ho, C), float("nan"), device=device)
        stats = torch.full((2, C), float("nan"), device=device)
        xd, wd, scd, shd = nhwc(x).to(device), w.to(device), sc.to(device), sh.to(device)  # named: kept alive for the launch
        _lib.check(lib.orbit_op_dwconv2d_train(_lib.dptr(xd), _lib.dptr(wd), _lib.dptr(y),
                                               _lib.dptr(scd) if in_act is not None else None,
                                               _lib.dptr(shd) if in_act is not None else None,
                                               in_act or 0, B, HW, HW, C, K, stride, pt, pt, Ho, Ho, _lib.dptr(stats), _st()),
                   "dwconv2d_train")
        torch.cuda.synchronize()
    finally:
        lib.orbit_set_option(b"dw_window", 1)
        lib.orbit_set_option(b"dw_lds", 1)
        lib.orbit_set_option(b"dw_pipe", 1)
    assert rel_err(nchw(y.cpu()), want) < 2e-5
    yc = y.cpu().double().reshape(-1, C)
    assert rel_err(stats[0].cpu(), yc.sum(0)) < 2e-6 * max(1.0, float(yc.abs().sum(0).max() / yc.sum(0).abs().max()))
    assert rel_err(stats[1].cpu(), (yc * yc).sum(0)) < 2e-6
