"""Does frame sub-batching of an (expand 1x1 conv -> depthwise) pair keep the expanded tensor in the 256 MB Infinity
Cache? Times full-batch vs sub-batched execution of EfficientNet-B0's early block shapes with the single-op C-ABI."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import orbit_dataset_amd  # noqa
from orbit_dataset_amd import _lib

lib = _lib.load()
dev = torch.device("cuda", 0)
B = 200
CASES = [("b1.0", 112, 16, 96, 3, 2), ("b1.1", 56, 24, 144, 3, 1), ("b2.0", 56, 24, 144, 5, 2), ("b2.1", 28, 40, 240, 5, 1),
         ("b3.0", 28, 40, 240, 3, 2), ("b3.1", 14, 80, 480, 3, 1)]


def conv(x, w, y, sc, sh, n, H, Cin, Cout):
    _lib.check(lib.orbit_op_conv2d(_lib.dptr(x), 0, _lib.dptr(w), _lib.dptr(y), _lib.dptr(sc), _lib.dptr(sh), None, None,
                                   n, H, H, Cin, Cout, 1, 1, 1, 0, 0, H, H, 2, 0, _lib.stream_handle()))


def dw(x, w, y, sc, sh, n, H, C, K, S, Ho, pad):
    _lib.check(lib.orbit_op_dwconv2d(_lib.dptr(x), _lib.dptr(w), _lib.dptr(y), _lib.dptr(sc), _lib.dptr(sh), n, H, H, C, K, S,
                                     pad, pad, Ho, Ho, 2, _lib.stream_handle()))


def timed(fn, reps=6):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for name, H, Cin, mid, K, S in CASES:
    Ho = -(-H // S)
    pad = max((Ho - 1) * S + K - H, 0) // 2
    x = torch.randn(B, H, H, Cin, device=dev)
    w1 = torch.randn(mid, Cin, 1, 1, device=dev) * 0.1
    wd = torch.randn(mid, 1, K, K, device=dev) * 0.1
    sc, sh = torch.rand(mid, device=dev), torch.rand(mid, device=dev)
    E = torch.empty(B, H, H, mid, device=dev)
    D = torch.empty(B, Ho, Ho, mid, device=dev)

    def full():
        conv(x, w1, E, sc, sh, B, H, Cin, mid)
        dw(E, wd, D, sc, sh, B, H, mid, K, S, Ho, pad)

    res = ["%-5s E=%6.0f MB  full %7.1f us" % (name, E.numel() * 4 / 1e6, timed(full))]
    for nb in (2, 4, 8, 16):
        cb = B // nb
        Es = E[:cb]

        def sub():
            for i in range(nb):
                conv(x[i * cb:(i + 1) * cb], w1, Es, sc, sh, cb, H, Cin, mid)
                dw(Es, wd, D[i * cb:(i + 1) * cb], sc, sh, cb, H, mid, K, S, Ho, pad)

        res.append("x%d(%3.0fMB) %7.1f" % (nb, Es.numel() * 4 / 1e6, timed(sub)))
    print("  ".join(res))
