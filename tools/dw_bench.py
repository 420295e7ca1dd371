"""Micro-benchmark of the depthwise kernel on EfficientNet-B0's layer shapes (200 frames). GPU box only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import orbit_dataset_amd  # noqa
from orbit_dataset_amd import _lib

SHAPES = [("b0 k3s1 112x32", 112, 32, 3, 1), ("b1.0 k3s2 112x96", 112, 96, 3, 2), ("b1.1 k3s1 56x144", 56, 144, 3, 1),
          ("b2.0 k5s2 56x144", 56, 144, 5, 2), ("b2.1 k5s1 28x240", 28, 240, 5, 1), ("b3.0 k3s2 28x240", 28, 240, 3, 2),
          ("b3.1 k3s1 14x480", 14, 480, 3, 1), ("b4.0 k5s1 14x480", 14, 480, 5, 1), ("b4.1 k5s1 14x672", 14, 672, 5, 1),
          ("b5.0 k5s2 14x672", 14, 672, 5, 2), ("b5.1 k5s1 7x1152", 7, 1152, 5, 1), ("b6.0 k3s1 7x1152", 7, 1152, 3, 1)]
lib = _lib.load()
dev = torch.device("cuda", 0)
B = 200
tot = {0: 0.0, 2: 0.0}
for name, H, C, K, S in SHAPES:
  line = "%-20s" % name
  OPT = sys.argv[1].encode() if len(sys.argv) > 1 else b"dw_lds"
  for opt in (0, 2, 0, 2):  # in-process A/B of one option: 0 vs 2 (dw_lds) / 0 vs 1 (others)
    lib.orbit_set_option(OPT, opt)
    if OPT == b"dw_pipe": lib.orbit_set_option(b"dw_lds", 0); lib.orbit_set_option(b"dw_window", 0)
    Ho = -(-H // S)
    pad = max((Ho - 1) * S + K - H, 0) // 2
    x = torch.randn(B, H, H, C, device=dev); w = torch.randn(C, 1, K, K, device=dev)
    y = torch.empty(B, Ho, Ho, C, device=dev); sc = torch.rand(C, device=dev); sh = torch.rand(C, device=dev)
    def run():
        _lib.check(lib.orbit_op_dwconv2d(_lib.dptr(x), _lib.dptr(w), _lib.dptr(y), _lib.dptr(sc), _lib.dptr(sh), B, H, H, C, K, S,
                                         pad, pad, Ho, Ho, 2, _lib.stream_handle()))
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    gb = 4.0 * B * C * (H * H + Ho * Ho) / 1e9
    tot[opt] += us / 2
    line += "  %s %7.1f us %5.2f TB/s" % ("on " if opt else "off", us, gb / (us * 1e-6) / 1e3)
  print(line)
print("sum of one instance each: off %.1f us, on %.1f us" % (tot[0], tot[2]))
