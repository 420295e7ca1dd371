"""HBM roofline of the distance kernel on the 64-task batched launch, per head_stream option, and a
bit-equality check between the forms. GPU box only. usage: python tools/head_roofline.py [quick]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import orbit_dataset_amd  # noqa: F401
from orbit_dataset_amd import _lib

lib = _lib.load()
dev = torch.device("cuda", 0)
quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
for (n_tasks, M, D, C) in ((64, 200, 1280, 5),) if quick else ((64, 200, 1280, 5), (64, 200, 1280, 10), (64, 200, 512, 5), (1, 200, 1280, 5)):
    g = torch.Generator(device=dev).manual_seed(7)
    qs = [torch.rand(n_tasks, M, D, device=dev, generator=g) for _ in range(8)]
    W = torch.rand(n_tasks, C, D, device=dev, generator=g)
    b = torch.rand(n_tasks, C, device=dev, generator=g)
    outs = {}
    for stream in (0, 1):
        lib.orbit_set_option(b"head_stream", stream)
        out = torch.empty(n_tasks, M, C, device=dev)

        def run(i):
            _lib.check(lib.orbit_proto_predict(_lib.dptr(qs[i % 8]), _lib.dptr(W), _lib.dptr(b), n_tasks, M, 1, D, C, 1.0, 0,
                                               _lib.dptr(out), None, _lib.stream_handle()), "orbit_proto_predict")
        for i in range(8):
            run(i)
        reps = 5 if quick else 40
        # per-launch event pairs: a ctypes launch costs the host ~20 us, more than the kernel runs - timing a loop of
        # launches as a whole measures the host, not the kernel
        evs = []
        for i in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run(i)
            e1.record()
            evs.append((e0, e1))
        torch.cuda.synchronize()
        ts = sorted(1e3 * a.elapsed_time(b) for a, b in evs)
        us = ts[len(ts) // 2]
        nbytes = 4.0 * (M * D + C * D + C + M * C) * n_tasks
        run(0)
        outs[stream] = out.clone()
        print("tasks %2d M %d D %4d C %2d  head_stream %d: %6.1f us  %.2f TB/s" % (n_tasks, M, D, C, stream, us, nbytes / us / 1e6))
    print("   max |diff| between forms: %.2e" % (outs[0] - outs[1]).abs().max().item())
lib.orbit_set_option(b"head_stream", 1)
