"""Row-streaming fused MBConv front (csrc/mbconv_rows.hip) vs the unfused conv + depthwise pair, per high-resolution EfficientNet-B0 block shape, 200 frames. GPU box only.
   python tools/mbrows_bench.py [shape substrings] [B=<frames>]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import orbit_dataset_amd  # noqa
from orbit_dataset_amd import _lib

SHAPES = [("b1.0 16->96 k3s2 112", 112, 16, 96, 3, 2), ("b1.1 24->144 k3s1 56", 56, 24, 144, 3, 1),
          ("b2.0 24->144 k5s2 56", 56, 24, 144, 5, 2), ("b2.1 40->240 k5s1 28", 28, 40, 240, 5, 1),
          ("b3.0 40->240 k3s2 28", 28, 40, 240, 3, 2)]
args = [a for a in sys.argv[1:] if "=" not in a]
kv = dict(a.split("=") for a in sys.argv[1:] if "=" in a)
if args:
    SHAPES = [s_ for s_ in SHAPES if any(a in s_[0] for a in args)]
lib = _lib.load()
dev = torch.device("cuda", 0)
B = int(kv.get("B", 200))
st = _lib.stream_handle
for name, H, Cin, mid, K, S in SHAPES:
    Ho = -(-H // S)
    tot = max((Ho - 1) * S + K - H, 0)
    pad = tot // 2
    g = torch.Generator(device=dev).manual_seed(5)
    x = torch.randn(B, H, H, Cin, device=dev, generator=g)
    w1 = torch.randn(mid, Cin, 1, 1, device=dev, generator=g) / Cin ** 0.5
    wd = torch.randn(mid, 1, K, K, device=dev, generator=g) / K
    s1, h1, s2, h2 = (torch.rand(mid, device=dev, generator=g) + 0.5 for _ in range(4))
    e = torch.empty(B, H, H, mid, device=dev)
    y = torch.empty(B, Ho, Ho, mid, device=dev)
    yf = torch.zeros(B, Ho, Ho, mid, device=dev)
    yr = torch.zeros(B, Ho, Ho, mid, device=dev)
    pool = torch.zeros(B, 4096, mid, device=dev)

    def unfused():
        _lib.check(lib.orbit_op_conv2d(_lib.dptr(x), 0, _lib.dptr(w1), _lib.dptr(e), _lib.dptr(s1), _lib.dptr(h1), None, None,
                                       B, H, H, Cin, mid, 1, 1, 1, 0, 0, H, H, 2, 0, st()))
        _lib.check(lib.orbit_op_dwconv2d(_lib.dptr(e), _lib.dptr(wd), _lib.dptr(y), _lib.dptr(s2), _lib.dptr(h2), B, H, H, mid,
                                         K, S, pad, pad, Ho, Ho, 2, st()))

    def front(out, rows):
        lib.orbit_set_option(b"mbconv_rows", rows)
        _lib.check(lib.orbit_op_mbconv_front(_lib.dptr(x), _lib.dptr(w1), _lib.dptr(s1), _lib.dptr(h1), _lib.dptr(wd),
                                             _lib.dptr(s2), _lib.dptr(h2), _lib.dptr(out), _lib.dptr(pool), B, H, H, Cin, mid,
                                             K, S, pad, pad, Ho, Ho, st()))

    def timeit(fn):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 100

    tu, tf, tr = timeit(unfused), timeit(lambda: front(yf, 0)), timeit(lambda: front(yr, 1))
    tu2, tf2, tr2 = timeit(unfused), timeit(lambda: front(yf, 0)), timeit(lambda: front(yr, 1))
    pool.zero_()
    front(yr, 1)
    torch.cuda.synchronize()
    rows_used = int((pool.view(-1, mid).abs().sum(1) > 0).sum().item())  # [B][tiles][mid] packed at the buffer's start
    psum = pool.view(-1, mid)[:rows_used].view(B, rows_used // B, mid).sum(1)
    want = yr.double().sum((1, 2))
    perr = ((psum.double() - want).abs().max() / want.abs().max()).item()
    gb = 4.0 * B * (H * H * Cin + Ho * Ho * mid) / 1e9
    print("%-22s pair %6.1f  tiled %6.1f  rows %6.1f us  (again %6.1f / %6.1f / %6.1f)  ideal(4.2TB/s) %5.1f us  "
          "max|rows-pair| %.1e  max|rows-tiled| %.1e  pool rel err %.1e" % (
              name, tu, tf, tr, tu2, tf2, tr2, gb / 4.2e3 * 1e6, (yr - y).abs().max().item(), (yr - yf).abs().max().item(), perr))
