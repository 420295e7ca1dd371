"""Soak test of the DEFAULT (fp32-MFMA) path under two concurrent streams: two independent models, N repeats, every output
compared bit for bit with its solo run. (The opt-in bf16 path had a timing-dependent operand hazard that 8 repeats exposed;
this asks the same question of the fp32 kernels with many more repeats.)  python tools/overlap_soak.py [repeats] [conv_bf3]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from orbit_dataset_amd import _lib, synthetic
lib = _lib.load()
device = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
opt = int(sys.argv[2]) if len(sys.argv) > 2 else 0
os.environ["ORBIT_BENCH_OVERLAP"] = "0"
lib.orbit_set_option(b"conv_bf3", opt)
for workload, frames in (("efficientnet_b0_224", 200), ("resnet18_224", 100), ("resnet18_84", 200)):
    size = bench.WORKLOADS[workload][2]
    models = [bench.build_model(workload, device) for _ in range(2)]
    for m in models:
        m.overlap_query = False
    tasks = [synthetic.make_task_on_device(i, 5, 1, frames // 5, frames, size, 1, device) for i in range(2)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    ref = [bench.run_task(models[i], tasks[i]).clone() for i in range(2)]
    torch.cuda.synchronize()
    bad, worst = 0, 0.0
    for rep in range(N):
        outs = []
        for i in range(2):
            with torch.cuda.stream(streams[i]):
                outs.append(bench.run_task(models[i], tasks[i]))
        torch.cuda.synchronize()
        for i in range(2):
            if not torch.equal(outs[i], ref[i]):
                bad += 1
                worst = max(worst, (outs[i] - ref[i]).abs().max().item())
    print("%s conv_bf3 %d: %d of %d concurrent task results differ from the solo run (max |dlogit| %.3e)" % (workload, opt, bad, 2 * N, worst), flush=True)
lib.orbit_set_option(b"conv_bf3", 0)
