"""The vendor library's fp32 GEMM (torch.mm -> rocBLAS / hipBLASLt, TF32 off) on the pointwise layer shapes of efficientnet_b0 @224
(200 frames), beside this library's conv kernels on the same tensors - plain GEMM only on the vendor side (no BatchNorm / SiLU /
gate / residual epilogue, which orbit_op_conv2d includes). A measurement tool, not part of the product path.
Usage (GPU box): python tools/blas_compare.py            prints one line per layer: us and TFLOP/s for both"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import orbit_dataset_amd  # noqa
from orbit_dataset_amd import _lib
from tools.rgemm_bench import SHAPES


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / reps


def main():
    torch.backends.cuda.matmul.allow_tf32 = False
    torch.set_float32_matmul_precision("highest")
    B = int(os.environ.get("FRAMES", "200"))
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    tot = [0.0, 0.0, 0.0]
    print("# %d frames; vendor = torch.mm fp32 (TF32 off) on [M, K] x [K, N]; ours = orbit_op_conv2d incl. its epilogue" % B)
    for name, H, Cin, Cout, g, r, act in SHAPES:
        M = B * H * H
        x = torch.randn(B, H, H, Cin, device=dev)
        w = torch.randn(Cout, Cin, 1, 1, device=dev) / Cin ** 0.5
        sc, sh = torch.rand(Cout, device=dev) + 0.5, torch.randn(Cout, device=dev)
        gate = torch.rand(B, Cin, device=dev) if g else None
        res = torch.randn(B, H, H, Cout, device=dev) if r else None
        y = torch.empty(B, H, H, Cout, device=dev)
        x2, wt, wn = x.view(M, Cin), w.view(Cout, Cin).t().contiguous(), w.view(Cout, Cin)
        out = torch.empty(M, Cout, device=dev)

        def ours():
            _lib.check(lib.orbit_op_conv2d(_lib.dptr(x), 0, _lib.dptr(w), _lib.dptr(y), _lib.dptr(sc), _lib.dptr(sh),
                                           _lib.dptr(res) if r else None, _lib.dptr(gate) if g else None, B, H, H, Cin, Cout, 1, 1, 1,
                                           0, 0, H, H, 1 if act else 0, 0, _lib.stream_handle()))

        t_nn = timed(lambda: torch.mm(x2, wt, out=out))        # B operand [K, N] row-major
        t_nt = timed(lambda: torch.mm(x2, wn.t(), out=out))    # B operand given as the transposed view of [N, K]
        t_o = timed(ours)
        fl = 2.0 * M * Cin * Cout
        best = min(t_nn, t_nt)
        tot[0] += best
        tot[1] += t_o
        print("%-14s M %7d K %4d N %4d   vendor %7.1f us (%5.1f TF; nn %6.1f nt %6.1f)   ours %7.1f us (%5.1f TF)   ours/vendor %.2f"
              % (name, M, Cin, Cout, best, fl / best / 1e6, t_nn, t_nt, t_o, fl / t_o / 1e6, t_o / best))
    print("sum: vendor %.1f us, ours %.1f us" % (tot[0], tot[1]))
    if len(sys.argv) > 1 and sys.argv[1] == "conv3x3":
        conv3x3(lib, dev, B)


def conv3x3(lib, dev, B):
    """resnet18 @224 3x3 layers: F.conv2d (MIOpen, fp32, NCHW and channels_last) beside orbit_op_conv2d (NHWC)."""
    import torch.nn.functional as F
    print("# resnet18 @224 3x3 stride-1 layers, %d frames: vendor = F.conv2d fp32 (MIOpen), no epilogue" % B)
    for H, C in ((56, 64), (28, 128), (14, 256), (7, 512)):
        x = torch.randn(B, H, H, C, device=dev)
        w = torch.randn(C, C, 3, 3, device=dev) / (9 * C) ** 0.5
        sc, sh = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
        y = torch.empty(B, H, H, C, device=dev)
        xn = x.permute(0, 3, 1, 2).contiguous()
        xc = xn.contiguous(memory_format=torch.channels_last)
        wc = w.contiguous(memory_format=torch.channels_last)

        def ours():
            _lib.check(lib.orbit_op_conv2d(_lib.dptr(x), 0, _lib.dptr(w), _lib.dptr(y), _lib.dptr(sc), _lib.dptr(sh), None, None,
                                           B, H, H, C, C, 3, 3, 1, 1, 1, H, H, 2, 0, _lib.stream_handle()))

        fl = 2.0 * B * H * H * C * C * 9
        t_o = timed(ours)
        t_n = timed(lambda: F.conv2d(xn, w, padding=1))
        t_c = timed(lambda: F.conv2d(xc, wc, padding=1))
        print("3x3 %3d ch @%2d   vendor nchw %7.1f us (%5.1f TF)  channels_last %7.1f us (%5.1f TF)   ours %7.1f us (%5.1f TF)"
              % (C, H, t_n, fl / t_n / 1e6, t_c, fl / t_c / 1e6, t_o, fl / t_o / 1e6))


if __name__ == "__main__":
    main()
