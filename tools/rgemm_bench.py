"""Pointwise register GEMM (csrc/pw_rgemm.hip) against the LDS-tiled implicit GEMM (csrc/conv_igemm.hip) on the pointwise layer
shapes of efficientnet_b0 @224 (200 frames): max |difference| of the outputs and us per launch, in one process.
Usage (GPU box): python tools/rgemm_bench.py [sweep]      sweep = also every (T, wk) the register GEMM can run a layer with"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import orbit_dataset_amd  # noqa
from orbit_dataset_amd import _lib

SHAPES = [  # (name, H, Cin, Cout, gate, residual, act)
    ("pwl32_16", 112, 32, 16, 1, 0, 0), ("pwl96_24", 56, 96, 24, 1, 0, 0), ("pwl144_24", 56, 144, 24, 1, 1, 0),
    ("pwl144_40", 28, 144, 40, 1, 0, 0), ("pwl240_40", 28, 240, 40, 1, 1, 0), ("pwl240_80", 14, 240, 80, 1, 0, 0),
    ("pw80_480", 14, 80, 480, 0, 0, 1), ("pwl480_80", 14, 480, 80, 1, 1, 0), ("pwl480_112", 14, 480, 112, 1, 0, 0),
    ("pw112_672", 14, 112, 672, 0, 0, 1), ("pwl672_112", 14, 672, 112, 1, 1, 0), ("pwl672_192", 7, 672, 192, 1, 0, 0),
    ("pw192_1152", 7, 192, 1152, 0, 0, 1), ("pwl1152_192", 7, 1152, 192, 1, 1, 0), ("pwl1152_320", 7, 1152, 320, 1, 0, 0),
    ("head320_1280", 7, 320, 1280, 0, 0, 1),
]


def main():
    sweep = len(sys.argv) > 1 and sys.argv[1] == "sweep"
    B = int(os.environ.get("FRAMES", "200"))
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    tot = {0: 0.0, 1: 0.0}
    only = os.environ.get("LAYERS")
    for name, H, Cin, Cout, g, r, act in SHAPES:
        if only and name not in only.split(","):
            continue
        x = torch.randn(B, H, H, Cin, device=dev)
        w = torch.randn(Cout, Cin, 1, 1, device=dev) / Cin ** 0.5
        sc, sh = torch.rand(Cout, device=dev) + 0.5, torch.randn(Cout, device=dev)
        gate = torch.rand(B, Cin, device=dev) if g else None
        res = torch.randn(B, H, H, Cout, device=dev) if r else None
        ys = {}

        def run(y):
            _lib.check(lib.orbit_op_conv2d(_lib.dptr(x), 0, _lib.dptr(w), _lib.dptr(y), _lib.dptr(sc), _lib.dptr(sh),
                                           _lib.dptr(res) if r else None, _lib.dptr(gate) if g else None, B, H, H, Cin, Cout, 1, 1, 1,
                                           0, 0, H, H, 1 if act else 0, 0, _lib.stream_handle()))

        def measure(y, reps=8):
            for _ in range(2):
                run(y)
            lib.orbit_prof_enable(1)
            for _ in range(reps):
                run(y)
            torch.cuda.synchronize()
            lib.orbit_prof_enable(0)
            ms, fl, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_long()
            lib.orbit_prof_collect(ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(n))
            nm = ctypes.create_string_buffer(48)
            lib.orbit_prof_variant(0, nm, None, None, None, None)
            return 1e3 * ms.value / reps, nm.value.decode()

        res_t = {}
        for rep in range(2):
            for opt in (0, 2):
                lib.orbit_set_option(b"conv_rgemm", opt)
                ys[opt] = torch.empty(B, H, H, Cout, device=dev)
                us, nm = measure(ys[opt])
                res_t[opt] = (min(us, res_t.get(opt, (1e9, ""))[0]), nm)
        ref = torch.nn.functional.conv2d((x * gate[:, None, None, :] if g else x).permute(0, 3, 1, 2).double(), w.double())
        ref = ref.permute(0, 2, 3, 1) * sc.double() + sh.double()
        if r:
            ref = ref + res.double()
        if act:
            ref = ref * torch.sigmoid(ref)
        e0 = (ys[0].double() - ref).abs().max().item() / ref.abs().max().item()
        e2 = (ys[2].double() - ref).abs().max().item() / ref.abs().max().item()
        M = B * H * H
        roof = 1e6 * max(2.0 * M * Cout * Cin / 157.3e12, 4.0 * (M * Cin + M * Cout * (2 if r else 1)) / 8e12)
        line = "%-13s M=%7d K=%4d N=%4d  igemm %6.1f us (%s)  rgemm %6.1f us (%s) %+5.0f%%  roof %5.1f  err %.1e / %.1e" % (
            name, M, Cin, Cout, res_t[0][0], res_t[0][1].split("<")[1][:-1], res_t[2][0], res_t[2][1].split("<")[1][:-1],
            100 * (res_t[0][0] / res_t[2][0] - 1), roof, e0, e2)
        tot[0] += res_t[0][0]
        tot[1] += res_t[2][0]
        print(line, flush=True)
    print("sum: igemm %.1f us  rgemm %.1f us" % (tot[0], tot[1]))


if __name__ == "__main__":
    main()
