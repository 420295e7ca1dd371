"""Op-level: the same conv_bf3 layer on two streams (different data), repeated: every output must equal its solo run. (debug probe)"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import orbit_dataset_amd
from orbit_dataset_amd import _lib
lib = _lib.load()
dev = torch.device("cuda", 0)
def conv(xx, ww, yy, gg, Bn, Hh, Ci, Co, stream):
    _lib.check(lib.orbit_op_conv2d(_lib.dptr(xx), 0, _lib.dptr(ww), _lib.dptr(yy), None, None, None, _lib.dptr(gg) if gg is not None else None,
                                   Bn, Hh, Hh, Ci, Co, 1, 1, 1, 0, 0, Hh, Hh, 0, 0, ctypes.c_void_p(stream.cuda_stream)))
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
for (B, H, Cin, Cout, gated) in ((200, 14, 480, 112, True), (200, 7, 1152, 320, True), (200, 14, 112, 672, False), (200, 7, 320, 1280, False), (200, 14, 480, 80, True)):
    for bk in (0, 16):
        lib.orbit_set_option(b"conv_bf3", 1)
        lib.orbit_set_option(b"conv_bf3_bk", bk)
        g = torch.Generator(device=dev).manual_seed(0)
        xs = [torch.randn(B, H, H, Cin, device=dev, generator=g) for _ in range(2)]
        w = torch.randn(Cout, Cin, 1, 1, device=dev, generator=g) / Cin ** 0.5
        gates = [torch.rand(B, Cin, device=dev, generator=g) if gated else None for _ in range(2)]
        refs = []
        for i in range(2):
            r = torch.empty(B, H, H, Cout, device=dev)
            conv(xs[i], w, r, gates[i], B, H, Cin, Cout, torch.cuda.current_stream())
            torch.cuda.synchronize()
            refs.append(r)
        bad, worst = 0, 0.0
        for rep in range(20):
            ys = [torch.empty(B, H, H, Cout, device=dev) for _ in range(2)]
            torch.cuda.synchronize()
            for k in range(3):
                for i in range(2):
                    conv(xs[i], w, ys[i], gates[i], B, H, Cin, Cout, streams[i])
            torch.cuda.synchronize()
            for i in range(2):
                if not torch.equal(ys[i], refs[i]):
                    bad += 1
                    worst = max(worst, (ys[i] - refs[i]).abs().max().item())
        print("%4d -> %4d @%2d gate %d  forced K-tile %2d: %d of 40 outputs differ from the solo run (max %.3e)" % (Cin, Cout, H, gated, bk, bad, worst), flush=True)
lib.orbit_set_option(b"conv_bf3", 0)
lib.orbit_set_option(b"conv_bf3_bk", 0)
