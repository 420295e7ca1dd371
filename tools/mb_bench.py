"""Fused MBConv front (expand 1x1 + BN + SiLU + depthwise + BN + SiLU) vs the unfused kernel pair, per EfficientNet-B0
block shape, 200 frames. GPU box only. The op wrappers pack weights per call (a few us each, on both sides)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import orbit_dataset_amd  # noqa
from orbit_dataset_amd import _lib

SHAPES = [("b1.0 16->96 k3s2 112", 112, 16, 96, 3, 2), ("b1.1 24->144 k3s1 56", 56, 24, 144, 3, 1),
          ("b2.0 24->144 k5s2 56", 56, 24, 144, 5, 2), ("b2.1 40->240 k5s1 28", 28, 40, 240, 5, 1),
          ("b3.0 40->240 k3s2 28", 28, 40, 240, 3, 2),
          # whole-map kernel (csrc/mbconv_map.hip)
          ("b3.1 80->480 k3s1 14", 14, 80, 480, 3, 1), ("b4.0 80->480 k5s1 14", 14, 80, 480, 5, 1),
          ("b4.1 112->672 k5s1 14", 14, 112, 672, 5, 1), ("b5.0 112->672 k5s2 14", 14, 112, 672, 5, 2),
          ("b5.1 192->1152 k5s1 7", 7, 192, 1152, 5, 1), ("b6.0 192->1152 k3s1 7", 7, 192, 1152, 3, 1)]
if len(sys.argv) > 1:
    SHAPES = [s_ for s_ in SHAPES if any(a in s_[0] for a in sys.argv[1:])]
lib = _lib.load()
lib.orbit_set_option(b"mbconv_map", 1)  # A/B of the opt-in whole-map kernel on the 14x14 / 7x7 shapes
dev = torch.device("cuda", 0)
B = 200
st = _lib.stream_handle
for name, H, Cin, mid, K, S in SHAPES:
    Ho = -(-H // S)
    tot = max((Ho - 1) * S + K - H, 0)
    pad = tot // 2
    x = torch.randn(B, H, H, Cin, device=dev)
    w1 = torch.randn(mid, Cin, 1, 1, device=dev) / Cin ** 0.5
    wd = torch.randn(mid, 1, K, K, device=dev) / K
    s1, h1, s2, h2 = (torch.rand(mid, device=dev) + 0.5 for _ in range(4))
    e = torch.empty(B, H, H, mid, device=dev)
    y = torch.empty(B, Ho, Ho, mid, device=dev)
    y2 = torch.empty(B, Ho, Ho, mid, device=dev)
    pool = torch.empty(B, 4096, mid, device=dev)

    def unfused():
        _lib.check(lib.orbit_op_conv2d(_lib.dptr(x), 0, _lib.dptr(w1), _lib.dptr(e), _lib.dptr(s1), _lib.dptr(h1), None, None,
                                       B, H, H, Cin, mid, 1, 1, 1, 0, 0, H, H, 2, 0, st()))
        _lib.check(lib.orbit_op_dwconv2d(_lib.dptr(e), _lib.dptr(wd), _lib.dptr(y), _lib.dptr(s2), _lib.dptr(h2), B, H, H, mid,
                                         K, S, pad, pad, Ho, Ho, 2, st()))

    def fused():
        _lib.check(lib.orbit_op_mbconv_front(_lib.dptr(x), _lib.dptr(w1), _lib.dptr(s1), _lib.dptr(h1), _lib.dptr(wd),
                                             _lib.dptr(s2), _lib.dptr(h2), _lib.dptr(y2), _lib.dptr(pool), B, H, H, Cin, mid,
                                             K, S, pad, pad, Ho, Ho, st()))

    def timeit(fn):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 100

    tu, tf = timeit(unfused), timeit(fused)
    tu2, tf2 = timeit(unfused), timeit(fused)
    err = (y - y2).abs().max().item()
    gb = 4.0 * B * (H * H * Cin + Ho * Ho * mid) / 1e9
    print("%-24s unfused %7.1f us  fused %7.1f us  (again %7.1f / %7.1f)  ideal(4.2TB/s) %6.1f us  max|diff| %.1e" % (
        name, tu, tf, tu2, tf2, gb / 4.2e3 * 1e6, err))
