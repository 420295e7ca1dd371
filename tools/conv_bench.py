"""Micro-benchmark of the MFMA implicit-GEMM convolution on the layer shapes of the bench workloads.
Usage (GPU box): python tools/conv_bench.py [resnet18_84|resnet18_224|effnet_224]   — prints TFLOP/s per layer."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import orbit_dataset_amd  # noqa
from orbit_dataset_amd import _lib

SHAPES = {
    "resnet18_84": [  # (name, B, H, Cin, Cout, K, stride, pad, nchw)
        ("stem7x7", 200, 84, 3, 64, 7, 2, 3, 1), ("l1", 200, 21, 64, 64, 3, 1, 1, 0), ("l2a", 200, 21, 64, 128, 3, 2, 1, 0),
        ("l2", 200, 11, 128, 128, 3, 1, 1, 0), ("l3a", 200, 11, 128, 256, 3, 2, 1, 0), ("l3", 200, 6, 256, 256, 3, 1, 1, 0),
        ("l4a", 200, 6, 256, 512, 3, 2, 1, 0), ("l4", 200, 3, 512, 512, 3, 1, 1, 0)],
    "resnet18_224": [
        ("stem7x7", 200, 224, 3, 64, 7, 2, 3, 1), ("l1", 200, 56, 64, 64, 3, 1, 1, 0), ("l2", 200, 28, 128, 128, 3, 1, 1, 0),
        ("l3", 200, 14, 256, 256, 3, 1, 1, 0), ("l4", 200, 7, 512, 512, 3, 1, 1, 0)],
    "effnet_224": [
        ("stem3x3", 200, 224, 3, 32, 3, 2, 0, 1), ("pw16_96", 200, 112, 16, 96, 1, 1, 0, 0), ("pwl96_24", 200, 56, 96, 24, 1, 1, 0, 0),
        ("pw24_144", 200, 56, 24, 144, 1, 1, 0, 0), ("pw40_240", 200, 28, 40, 240, 1, 1, 0, 0), ("pw80_480", 200, 14, 80, 480, 1, 1, 0, 0),
        ("pwl480_112", 200, 14, 480, 112, 1, 1, 0, 0), ("pw112_672", 200, 14, 112, 672, 1, 1, 0, 0), ("pw192_1152", 200, 7, 192, 1152, 1, 1, 0, 0),
        ("pwl1152_320", 200, 7, 1152, 320, 1, 1, 0, 0), ("head320_1280", 200, 7, 320, 1280, 1, 1, 0, 0),
        ("pwl32_16", 200, 112, 32, 16, 1, 1, 0, 0), ("pwl144_24", 200, 56, 144, 24, 1, 1, 0, 0),
        ("pwl144_40", 200, 28, 144, 40, 1, 1, 0, 0), ("pwl240_40", 200, 28, 240, 40, 1, 1, 0, 0),
        ("pwl240_80", 200, 14, 240, 80, 1, 1, 0, 0), ("pwl480_80", 200, 14, 480, 80, 1, 1, 0, 0),
        ("pwl672_112", 200, 14, 672, 112, 1, 1, 0, 0), ("pwl672_192", 200, 7, 672, 192, 1, 1, 0, 0),
        ("pwl1152_192", 200, 7, 1152, 192, 1, 1, 0, 0)],
    "setenc_224": [
        ("se_l1", 200, 224, 3, 64, 3, 1, 1, 1), ("se_l2", 200, 112, 64, 64, 3, 1, 1, 0), ("se_l3", 200, 56, 64, 64, 3, 1, 1, 0),
        ("se_l4", 200, 28, 64, 64, 3, 1, 1, 0), ("se_l5", 200, 14, 64, 64, 3, 1, 1, 0)],
}
TILES = {0: "auto", 1: "128x128", 2: "128x64", 3: "64x64", 4: "128x32", 5: "64x32k2", 6: "32x32k4", 7: "32x64k2",
         8: "32x96k4", 9: "64x96k2", 10: "128x96", 11: "32x128k4", 12: "64x128k2", 13: "128x128w4"}


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "resnet18_84"
    sweep = len(sys.argv) > 2 and sys.argv[2] in ("sweep", "gatesweep")
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    import ctypes
    for name, B, H, Cin, Cout, K, stride, pad, nchw in SHAPES[which]:
        Ho = -(-H // stride) if (which.startswith("effnet") and nchw) else (H + 2 * pad - K) // stride + 1
        x = torch.randn(B, Cin, H, H, device=dev) if nchw else torch.randn(B, H, H, Cin, device=dev)
        w = torch.randn(Cout, Cin, K, K, device=dev)
        y = torch.empty(B, Ho, Ho, Cout, device=dev)
        sc, sh = torch.rand(Cout, device=dev), torch.rand(Cout, device=dev)
        use_gate = len(sys.argv) > 2 and sys.argv[2] in ("gate", "abgate", "gatesweep") and not nchw and K == 1 and name.startswith("pwl")
        gate = torch.rand(B, Cin, device=dev) if use_gate else None

        def run():
            _lib.check(lib.orbit_op_conv2d(_lib.dptr(x), nchw, _lib.dptr(w), _lib.dptr(y), _lib.dptr(sc), _lib.dptr(sh), None,
                                           _lib.dptr(gate) if use_gate else None,
                                           B, H, H, Cin, Cout, K, K, stride, pad, pad, Ho, Ho, 1, 0, _lib.stream_handle()))

        def measure(tile):
            lib.orbit_set_option(b"conv_tile", tile)
            for _ in range(2):
                run()
            lib.orbit_prof_enable(1)
            for _ in range(8):
                run()
            torch.cuda.synchronize()
            lib.orbit_prof_enable(0)
            ms, fl, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_long()
            lib.orbit_prof_collect(ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(n))
            nm = ctypes.create_string_buffer(48)
            lib.orbit_prof_variant(0, nm, None, None, None, None)
            return 1e3 * ms.value / 8, fl.value / ms.value / 1e9, nm.value.decode()  # per run (8 runs), all kernels of the conv

        us, tf, nm = measure(0)
        Mr, Kr = B * Ho * Ho, Cin * K * K
        ideal = 1e6 * max(2.0 * Mr * Cout * Kr / 157.3e12, 4.0 * (B * H * H * Cin + Mr * Cout + Cout * Kr) / 8e12)
        line = "%-14s M=%7d N=%4d K=%5d  %-26s %8.1f us %6.1f TF  roof %6.1f us (%.2f)" % (
            name, Mr, Cout, Kr, nm, us, tf, ideal, ideal / us)
        if len(sys.argv) > 2 and sys.argv[2] == "bk":
            res = {}
            for bk in (32, 16, 8):
                lib.orbit_set_option(b"conv_bk", bk)
                res[bk] = measure(0)
            lib.orbit_set_option(b"conv_bk", 0)
            line += "  | " + "  ".join("BK<=%d %.1f us (%s)" % (bk, r[0], r[2].split("<")[1][:9]) for bk, r in res.items())
        if len(sys.argv) > 2 and sys.argv[2] in ("ab", "abgate"):  # in-process A/B of a runtime option (interleaved)
            opt = sys.argv[3].encode()
            res = {0: [], 1: []}
            for rep_ in range(3):
                for v in (0, 1):
                    lib.orbit_set_option(opt, v)
                    res[v].append(measure(0)[0])
            a0, a1 = min(res[0]), min(res[1])
            line += "  | %s=0 %.1f us  =1 %.1f us  (%+.1f%%)" % (sys.argv[3], a0, a1, 100 * (a0 / a1 - 1))
            lib.orbit_set_option(opt, 1)
        if sweep:
            cand = (3, 4, 6)
            res = {t: measure(t)[0] for t in cand}
            best = min(res, key=res.get)
            line += "  | " + "  ".join("%s %.1f" % (TILES[t], res[t]) for t in res) + "  -> best %s (%.0f%% vs auto)" % (
                TILES[best], 100 * (us / res[best] - 1))
        lib.orbit_set_option(b"conv_tile", 0)
        print(line)


if __name__ == "__main__":
    main()
