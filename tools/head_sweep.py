import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import orbit_dataset_amd
from orbit_dataset_amd import _lib
lib = _lib.load(); dev = torch.device("cuda", 0)
M, D, C = 200, 1280, 5
for n_tasks in (64, 256, 1024):
    g = torch.Generator(device=dev).manual_seed(7)
    nb = 8 if n_tasks <= 256 else 2
    qs = [torch.rand(n_tasks, M, D, device=dev, generator=g) for _ in range(nb)]
    W = torch.rand(n_tasks, C, D, device=dev, generator=g); b = torch.rand(n_tasks, C, device=dev, generator=g)
    out = torch.empty(n_tasks, M, C, device=dev)
    for stream, ldsopt in ((0, 1), (1, 1), (2, 1), (3, 1)):
        lib.orbit_set_option(b"head_stream", stream); lib.orbit_set_option(b"head_lds", ldsopt)
        def run(i):
            _lib.check(lib.orbit_proto_predict(_lib.dptr(qs[i % nb]), _lib.dptr(W), _lib.dptr(b), n_tasks, M, 1, D, C, 1.0, 0, _lib.dptr(out), None, _lib.stream_handle()))
        for i in range(4): run(i)
        evs = []
        for i in range(20):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(i); e1.record(); evs.append((e0, e1))
        torch.cuda.synchronize()
        ts = sorted(1e3 * a.elapsed_time(b_) for a, b_ in evs); us = ts[len(ts)//2]
        nbytes = 4.0 * (M * D + C * D + C + M * C) * n_tasks
        print("tasks %4d stream %d lds %d: %7.1f us %.2f TB/s" % (n_tasks, stream, ldsopt, us, nbytes / us / 1e6))
