"""Narrow pointwise projections: csrc/pw_narrow.hip vs conv_igemm, the three high-resolution EfficientNet-B0 shapes, 200 frames."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import orbit_dataset_amd  # noqa
from orbit_dataset_amd import _lib
lib = _lib.load(); dev = torch.device("cuda", 0); st = _lib.stream_handle; B = 200
for name, H, Cin, Cout, res in (("b0.0 32->16 @112", 112, 32, 16, False), ("b1.0 96->24 @56", 56, 96, 24, False), ("b1.1 144->24 @56", 56, 144, 24, True)):
    x = torch.randn(B, H, H, Cin, device=dev); w = torch.randn(Cout, Cin, 1, 1, device=dev) / Cin ** 0.5
    sc, sh = torch.rand(Cout, device=dev) + 0.5, torch.randn(Cout, device=dev)
    gate = torch.rand(B, Cin, device=dev); r = torch.randn(B, H, H, Cout, device=dev) if res else None
    y0 = torch.empty(B, H, H, Cout, device=dev); y1 = torch.empty_like(y0)
    def run(y):
        _lib.check(lib.orbit_op_conv2d(_lib.dptr(x), 0, _lib.dptr(w), _lib.dptr(y), _lib.dptr(sc), _lib.dptr(sh), _lib.dptr(r), _lib.dptr(gate), B, H, H, Cin, Cout, 1, 1, 1, 0, 0, H, H, 0, 0, st()))
    def timeit(opt, y):
        lib.orbit_set_option(b"pw_narrow", opt)
        for _ in range(3): run(y)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run(y)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 100
    a, b_ = timeit(0, y0), timeit(1, y1); a2, b2 = timeit(0, y0), timeit(1, y1)
    gb = 4.0 * B * H * H * (Cin + Cout * (2 if res else 1)) / 1e9
    print("%-18s igemm %6.1f us  narrow %6.1f us  (again %6.1f / %6.1f)  %.0f MB -> %.2f / %.2f TB/s  identical %s" % (name, a, b_, a2, b2, gb * 1e3, gb / a2 * 1e3, gb / b2 * 1e3, torch.equal(y0, y1)))
