"""Whole-map fused MBConv front: time vs chunk grouping / debug phase switches. GPU box only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import orbit_dataset_amd  # noqa
from orbit_dataset_amd import _lib
lib = _lib.load()
lib.orbit_set_option(b"mbconv_map", 1)
dev = torch.device("cuda", 0)
B = int(os.environ.get("B", 200))
st = _lib.stream_handle
SHAPES = [("b3.1", 14, 80, 3, 1), ("b4.1", 14, 112, 5, 1), ("b5.0", 14, 112, 5, 2), ("b5.1", 7, 192, 5, 1)]
for name, H, Cin, K, S in SHAPES:
    if len(sys.argv) > 1 and name not in sys.argv[1:]:
        continue
    mid = 6 * Cin
    Ho = -(-H // S)
    pad = max((Ho - 1) * S + K - H, 0) // 2
    x = torch.randn(B, H, H, Cin, device=dev)
    w1 = torch.randn(mid, Cin, device=dev) / Cin ** 0.5
    wd = torch.randn(K, K, mid, device=dev) / K
    s1, h1, s2, h2 = (torch.rand(mid, device=dev) + 0.5 for _ in range(4))
    y = torch.empty(B, Ho, Ho, mid, device=dev)
    pool = torch.empty(B, mid, device=dev)

    def run():
        _lib.check(lib.orbit_op_mbconv_front(_lib.dptr(x), _lib.dptr(w1), _lib.dptr(s1), _lib.dptr(h1), _lib.dptr(wd),
                                             _lib.dptr(s2), _lib.dptr(h2), _lib.dptr(y), _lib.dptr(pool), B, H, H, Cin, mid,
                                             K, S, pad, pad, Ho, Ho, st()))

    def timeit():
        for _ in range(3): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 100
    if os.environ.get("ORBIT_PROBE_QUICK"):
        print(name, "B", B, "quick %.0f us" % timeit())
        continue
    out = []
    for groups in (0, 1, 2, 3, 4, 6, 8, 12):
        lib.orbit_set_option(b"mbmap_groups", groups)
        out.append("g%d %.0f" % (groups, timeit()))
    lib.orbit_set_option(b"mbmap_groups", 0)
    print(name, "B", B, "|", " ".join(out))
