"""conv_bf3 (csrc/conv_bf3.hip: both operands split three ways into bf16, six products on the bf16 matrix cores) against the default
fp32-MFMA conv kernels on the pointwise layer shapes of efficientnet_b0 @224 (200 frames): us per launch and the error of BOTH against
an fp64 evaluation of the same layer (max |difference| / max |reference|, and the rms of the difference / rms of the reference).
Usage (GPU box): python tools/bf3_bench.py"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import orbit_dataset_amd  # noqa
from orbit_dataset_amd import _lib
from tools.rgemm_bench import SHAPES


def main():
    B = int(os.environ.get("FRAMES", "200"))
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    tot = [0.0, 0.0]
    for name, H, Cin, Cout, g, r, act in SHAPES:
        x = torch.randn(B, H, H, Cin, device=dev)
        w = torch.randn(Cout, Cin, 1, 1, device=dev) / Cin ** 0.5
        sc, sh = torch.rand(Cout, device=dev) + 0.5, torch.randn(Cout, device=dev)
        gate = torch.rand(B, Cin, device=dev) if g else None
        res = torch.randn(B, H, H, Cout, device=dev) if r else None

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

        ys, t = {}, {}
        for rep in range(2):
            for opt in (0, 1):
                lib.orbit_set_option(b"conv_bf3", opt)
                ys[opt] = torch.empty(B, H, H, Cout, device=dev)
                us, nm = measure(ys[opt])
                t[opt] = (min(us, t.get(opt, (1e9, ""))[0]), nm)
        lib.orbit_set_option(b"conv_bf3", 0)
        ref = torch.nn.functional.conv2d((x * gate[:, None, None, :] if g else x).permute(0, 3, 1, 2).double(), w.double())
        ref = ref.permute(0, 2, 3, 1) * sc.double() + sh.double()
        if r:
            ref = ref + res.double()
        if act:
            ref = ref * torch.sigmoid(ref)
        err = []
        for opt in (0, 1):
            d = ys[opt].double() - ref
            err.append((d.abs().max().item() / ref.abs().max().item(), (d.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()))
        fl = 2.0 * B * H * H * Cin * Cout
        print("%-13s K=%4d N=%4d  fp32 %6.1f us %5.1f TF (%s)   bf3 %6.1f us %5.1f TF (%s) %+5.0f%%   err max %.1e / %.1e  rms %.1e / %.1e"
              % (name, Cin, Cout, t[0][0], fl / t[0][0] / 1e6, t[0][1], t[1][0], fl / t[1][0] / 1e6, t[1][1],
                 100 * (t[0][0] / t[1][0] - 1), err[0][0], err[1][0], err[0][1], err[1][1]), flush=True)
        tot[0] += t[0][0]
        tot[1] += t[1][0]
    print("sum: fp32 %.1f us  bf3 %.1f us" % (tot[0], tot[1]))
    if len(sys.argv) > 1 and sys.argv[1] == "resnet":
        resnet(lib, dev, B)


def resnet(lib, dev, B):
    """resnet18 @224 3x3 and shortcut layers (NHWC, ReLU epilogue), default fp32-MFMA kernel against conv_bf3's general form."""
    tot = [0.0, 0.0]
    for name, H, Cin, Cout, K, stride, n in (("l1 3x3", 56, 64, 64, 3, 1, 4), ("l2 3x3/2", 56, 64, 128, 3, 2, 1), ("l2 1x1/2", 56, 64, 128, 1, 2, 1),
                                             ("l2 3x3", 28, 128, 128, 3, 1, 3), ("l3 3x3/2", 28, 128, 256, 3, 2, 1), ("l3 3x3", 14, 256, 256, 3, 1, 3),
                                             ("l4 3x3/2", 14, 256, 512, 3, 2, 1), ("l4 3x3", 7, 512, 512, 3, 1, 3)):
        pad = K // 2
        Ho = (H + 2 * pad - K) // stride + 1
        x = torch.randn(B, H, H, Cin, device=dev).abs()
        w = torch.randn(Cout, Cin, K, K, device=dev) / (Cin * K * K) ** 0.5
        sc, sh = torch.rand(Cout, device=dev) + 0.5, torch.randn(Cout, device=dev)

        def run(y):
            _lib.check(lib.orbit_op_conv2d(_lib.dptr(x), 0, _lib.dptr(w), _lib.dptr(y), _lib.dptr(sc), _lib.dptr(sh), None, None,
                                           B, H, H, Cin, Cout, K, K, stride, pad, pad, Ho, Ho, 1, 0, _lib.stream_handle()))

        def measure(y, reps=6):
            for _ in range(2):
                run(y)
            lib.orbit_prof_enable(1)
            for _ in range(reps):
                run(y)
            torch.cuda.synchronize()
            lib.orbit_prof_enable(0)
            ms, fl, n_ = ctypes.c_double(), ctypes.c_double(), ctypes.c_long()
            lib.orbit_prof_collect(ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(n_))
            nm = ctypes.create_string_buffer(48)
            lib.orbit_prof_variant(0, nm, None, None, None, None)
            return 1e3 * ms.value / reps, nm.value.decode()

        ys, t = {}, {}
        for rep in range(2):
            for opt in (0, 1):
                lib.orbit_set_option(b"conv_bf3", opt)
                ys[opt] = torch.empty(B, Ho, Ho, Cout, device=dev)
                us, nm = measure(ys[opt])
                t[opt] = (min(us, t.get(opt, (1e9, ""))[0]), nm)
        lib.orbit_set_option(b"conv_bf3", 0)
        ref = torch.nn.functional.conv2d(x[:8].permute(0, 3, 1, 2).double(), w.double(), None, stride, pad).permute(0, 2, 3, 1)
        ref = torch.relu(ref * sc.double() + sh.double())
        err = [(ys[o][:8].double() - ref).abs().max().item() / ref.abs().max().item() for o in (0, 1)]
        fl = 2.0 * B * Ho * Ho * Cin * Cout * K * K
        print("%-9s x%d  %3d->%3d @%2d  fp32 %7.1f us %5.1f TF (%s)   bf3 %7.1f us %5.1f TF (%s) %+5.0f%%   err max %.1e / %.1e"
              % (name, n, Cin, Cout, H, t[0][0], fl / t[0][0] / 1e6, t[0][1], t[1][0], fl / t[1][0] / 1e6, t[1][1],
                 100 * (t[0][0] / t[1][0] - 1), err[0], err[1]), flush=True)
        tot[0] += n * t[0][0]
        tot[1] += n * t[1][0]
    print("resnet18 @224 dense convs without the stem, per %d-frame pass: fp32 %.1f us  bf3 %.1f us" % (B, tot[0], tot[1]))


if __name__ == "__main__":
    main()
