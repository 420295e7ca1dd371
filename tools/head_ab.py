"""A/B of the prototype distance kernel (LDS-staged vs one wave per row), GPU box."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import orbit_dataset_amd  # noqa
from orbit_dataset_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0")
for (n_tasks, M, D, C) in ((64, 200, 1280, 5), (64, 200, 512, 5), (64, 200, 1280, 10), (1, 200, 1280, 5), (8, 200, 1280, 5)):
    g = torch.Generator(device=dev).manual_seed(7)
    qs = [torch.rand(n_tasks, M, D, device=dev, generator=g) for _ in range(8)]
    W = torch.rand(n_tasks, C, D, device=dev, generator=g); b = torch.rand(n_tasks, C, device=dev, generator=g)
    outs = {}
    line = "tasks %2d M %d D %4d C %2d:" % (n_tasks, M, D, C)
    for opt in (0, 1, 2):
        lib.orbit_set_option(b"head_lds", opt)
        out = torch.empty(n_tasks, M, C, device=dev)
        run = lambda i: _lib.check(lib.orbit_proto_predict(_lib.dptr(qs[i % 8]), _lib.dptr(W), _lib.dptr(b), n_tasks, M, 1, D, C, 1.0, 0, _lib.dptr(out), None, _lib.stream_handle()))
        for i in range(8): run(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(40): run(i)
        e1.record(); torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / 40
        nbytes = 4.0 * (M * D + C * D + C + M * C) * n_tasks
        run(0); torch.cuda.synchronize(); outs[opt] = out.clone()
        line += "  %s %6.1f us %5.2f TB/s" % (("row", "lds4", "lds8")[opt], us, nbytes / (us * 1e-6) / 1e12)
    line += "  lds4==lds8: %s  max|row-lds| %.1e" % (torch.equal(outs[1], outs[2]), float((outs[0]-outs[1]).abs().max()))
    print(line)
lib.orbit_set_option(b"head_lds", 1)
