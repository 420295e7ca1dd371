"""Micro-benchmark of the TRAIN-form depthwise convolution (input transform on load + statistics epilogue: the kernel of the
LITE forwards) on EfficientNet-B0's layer shapes at 200 frames, one line per layer with each kernel form forced in turn.
GPU box only:  python tools/dw_train_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import orbit_dataset_amd  # noqa
from orbit_dataset_amd import _lib

SHAPES = [("b0 k3s1 112x32", 112, 32, 3, 1), ("b1.0 k3s2 112x96", 112, 96, 3, 2), ("b1.1 k3s1 56x144", 56, 144, 3, 1),
          ("b2.0 k5s2 56x144", 56, 144, 5, 2), ("b2.1 k5s1 28x240", 28, 240, 5, 1), ("b3.0 k3s2 28x240", 28, 240, 3, 2),
          ("b3.1 k3s1 14x480", 14, 480, 3, 1), ("b4.0 k5s1 14x480", 14, 480, 5, 1), ("b4.1 k5s1 14x672", 14, 672, 5, 1),
          ("b5.0 k5s2 14x672", 14, 672, 5, 2), ("b5.1 k5s1 7x1152", 7, 1152, 5, 1), ("b6.0 k3s1 7x1152", 7, 1152, 3, 1)]
# (dw_lds, dw_window, dw_pipe): 1 = the plan's automatic choice
FORMS = [("auto", 1, 1, 1), ("lds", 2, 0, 0), ("window", 0, 2, 0), ("pipe", 0, 0, 2), ("plain", 0, 0, 0)]
lib = _lib.load()
dev = torch.device("cuda", 0)
B = 200
tot = {f[0]: 0.0 for f in FORMS}
best = 0.0
for name, H, C, K, S in SHAPES:
    Ho = -(-H // S)
    pad = max((Ho - 1) * S + K - H, 0) // 2
    x = torch.randn(B, H, H, C, device=dev)
    w = torch.randn(C, 1, K, K, device=dev)
    y = torch.empty(B, Ho, Ho, C, device=dev)
    sc, sh = torch.rand(C, device=dev) + 0.5, torch.rand(C, device=dev)
    stats = torch.empty(2, C, device=dev)
    gb = 4.0 * B * C * (H * H + Ho * Ho) / 1e9
    line, row = "%-20s" % name, []
    for form, lds, win, pipe in FORMS:
        lib.orbit_set_option(b"dw_lds", lds), lib.orbit_set_option(b"dw_window", win), lib.orbit_set_option(b"dw_pipe", pipe)

        def run():
            _lib.check(lib.orbit_op_dwconv2d_train(_lib.dptr(x), _lib.dptr(w), _lib.dptr(y), _lib.dptr(sc), _lib.dptr(sh), 2, B, H,
                                                   H, C, K, S, pad, pad, Ho, Ho, _lib.dptr(stats), _lib.stream_handle()), form)
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        tot[form] += us
        row.append(us)
        line += "  %s %6.1f us %4.2f TB/s" % (form, us, gb / (us * 1e-6) / 1e3)
    best += min(row)
    print(line)
print("sum over the 12 shapes: " + ", ".join("%s %.0f us" % (k, v) for k, v in tot.items()) + ", best per layer %.0f us" % best)
