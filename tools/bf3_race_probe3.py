"""Two INDEPENDENT model instances, each on its own stream (no overlap machinery, separate plans / workspaces), issued back to
back so their kernels co-run: are conv_bf3's results still timing-dependent? (debug probe)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from orbit_dataset_amd import _lib, synthetic
lib = _lib.load()
device = torch.device("cuda", 0)
os.environ["ORBIT_BENCH_OVERLAP"] = "0"
models = []
for i in range(2):
    m = bench.build_model("efficientnet_b0_224", device)
    bench.load_trained_checkpoint(m, bench.trained_checkpoint("efficientnet_b0_224"))
    m.overlap_query = False
    models.append(m)
tasks = [synthetic.make_task_on_device(i, 5, 1, 40, 200, 224, 1, device, template="blobs") for i in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
for opt in (0, 1, 2, 3):
    lib.orbit_set_option(b"conv_bf3", opt)
    lib.orbit_set_option(b"graph", 0)
    lib.orbit_set_option(b"conv_bf3_bk", int(os.environ.get("BK", "0")))
    ref = [bench.run_task(models[i], tasks[i]).clone() for i in range(2)]  # one at a time
    torch.cuda.synchronize()
    worst = 0.0
    for rep in range(8):
        outs = []
        for i in range(2):
            with torch.cuda.stream(streams[i]):
                outs.append(bench.run_task(models[i], tasks[i]))
        torch.cuda.synchronize()
        worst = max(worst, max((outs[i] - ref[i]).abs().max().item() for i in range(2)))
    print("conv_bf3 %d: two independent models on two streams, max |logit - solo run| over 8 repeats: %.3e" % (opt, worst), flush=True)
    for only in ("mbconv_rows", "conv"):
        pass
lib.orbit_set_option(b"conv_bf3", 0)
lib.orbit_set_option(b"graph", 2)
