"""Is the whole task deterministic with conv_bf3 under the support / query stream overlap? (debug probe)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from orbit_dataset_amd import _lib, synthetic
lib = _lib.load()
device = torch.device("cuda", 0)
model = bench.build_model("efficientnet_b0_224", device)
bench.load_trained_checkpoint(model, bench.trained_checkpoint("efficientnet_b0_224"))
task = synthetic.make_task_on_device(0, 5, 1, 40, 200, 224, 1, device, template="blobs")
tasks = [synthetic.make_task_on_device(i, 5, 1, 40, 200, 224, 1, device, template="blobs") for i in range(1, 4)]
def runs(n, warm=0):
    out = []
    for i in range(n):
        for t in tasks[:warm]:
            bench.run_task(model, t)
        out.append(bench.run_task(model, task).clone())
    torch.cuda.synchronize()
    return out
for graph in (0, 2, 1):
    lib.orbit_set_option(b"graph", graph)
    for ov in (False, True):
        model.overlap_query = ov
        for opt in (0, 1):
            lib.orbit_set_option(b"conv_bf3", opt)
            for warm in (0, 2):
                o = runs(6, warm)
                d = max((x - o[0]).abs().max().item() for x in o)
                print("graph %d overlap %d conv_bf3 %d other-tasks-between %d: max diff between repeats %.3e" % (graph, ov, opt, warm, d), flush=True)
lib.orbit_set_option(b"conv_bf3", 0)
lib.orbit_set_option(b"graph", 2)
