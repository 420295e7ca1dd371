"""Row-streaming stem + first depthwise (csrc/mbconv_rows.hip stem_rows_kernel) vs the direct stem kernel + depthwise pair,
224x224 frames, 200 frames. GPU box only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import orbit_dataset_amd  # noqa
from orbit_dataset_amd import _lib
lib = _lib.load()
dev = torch.device("cuda", 0)
B, FH = 200, 224
st = _lib.stream_handle
H = FH // 2
g = torch.Generator(device=dev).manual_seed(3)
x = torch.randn(B, 3, FH, FH, device=dev, generator=g)
ws = torch.randn(32, 3, 3, 3, device=dev, generator=g) / 27 ** 0.5
wd = torch.randn(32, 1, 3, 3, device=dev, generator=g) / 3
s1, h1, s2, h2 = (torch.rand(32, device=dev, generator=g) + 0.5 for _ in range(4))
e = torch.empty(B, H, H, 32, device=dev); y = torch.empty(B, H, H, 32, device=dev); yr = torch.zeros(B, H, H, 32, device=dev)
pool = torch.zeros(B, 4096, 32, device=dev)
def pair():
    _lib.check(lib.orbit_op_conv2d(_lib.dptr(x), 1, _lib.dptr(ws), _lib.dptr(e), _lib.dptr(s1), _lib.dptr(h1), None, None, B, FH, FH, 3, 32, 3, 3, 2, 0, 0, H, H, 2, 0, st()))
    _lib.check(lib.orbit_op_dwconv2d(_lib.dptr(e), _lib.dptr(wd), _lib.dptr(y), _lib.dptr(s2), _lib.dptr(h2), B, H, H, 32, 3, 1, 1, 1, H, H, 2, st()))
def rows():
    _lib.check(lib.orbit_op_stem_dw_front(_lib.dptr(x), _lib.dptr(ws), _lib.dptr(s1), _lib.dptr(h1), _lib.dptr(wd), _lib.dptr(s2), _lib.dptr(h2), _lib.dptr(yr), _lib.dptr(pool), B, FH, FH, 0, 0, H, H, 32, 1, 1, H, H, st()))
def timeit(fn):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 100
for band in (28,):
    print("band %3d: pair %6.1f us   rows %6.1f us   (again %6.1f / %6.1f)   max|diff| %.1e" % (band, timeit(pair), timeit(rows), timeit(pair), timeit(rows), (y - yr).abs().max().item()))
