"""Where the HOST spends its time per inference task (personalise + predict, 200 + 200 resident frames): cProfile over N
tasks with fresh label tensors, and the wall time per task of the enqueue loop alone. GPU box only.
python tools/host_profile.py [workload] [tasks]"""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from orbit_dataset_amd import synthetic  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "efficientnet_b0_224"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
device = torch.device("cuda", 0)
torch.cuda.set_device(0)
model = bench.build_model(workload, device, 256, train=False)
size = bench.WORKLOADS[workload][2]
tasks = [synthetic.make_task_on_device(i, 5, 1, 40, 200, size, 1, device) for i in range(4)]


def fresh(k):
    return [dict(tasks[i % 4], context_labels=tasks[i % 4]["context_labels"].clone()) for i in range(k)]


import gc
gc.collect(); gc.freeze()
for t in fresh(30):
    bench.run_task(model, t)
torch.cuda.synchronize()
for rep in range(2):
    todo = fresh(n)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in todo:
        bench.run_task(model, t)
    issued = time.perf_counter() - t0
    torch.cuda.synchronize()
    total = time.perf_counter() - t0
    print("rep %d: host enqueue %.3f ms / task, wall %.3f ms / task" % (rep, 1e3 * issued / n, 1e3 * total / n))
todo = fresh(n)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for t in todo:
    bench.run_task(model, t)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
