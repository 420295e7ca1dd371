"""Where the HOST spends its time in one LITE meta-training step (bench.LiteTrainStep): cProfile over N steps. GPU box only.
python tools/host_profile_lite.py [workload] [steps]"""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from orbit_dataset_amd import synthetic  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "cnaps_resnet18_224"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
device = torch.device("cuda", 0)
torch.cuda.set_device(0)
model = bench.build_model(workload, device, 256, train=True)
step = bench.LiteTrainStep(model, 1, 256, 1)
size = bench.WORKLOADS[workload][2]
tasks = [dict(synthetic.make_task_on_device(i, 5, 1, 40, 200, size, 1, device), task_index=i) for i in range(2)]
import gc
gc.collect(); gc.freeze()
for i in range(12):
    step(model, tasks[i % 2])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(n):
    step(model, tasks[i % 2])
issued = time.perf_counter() - t0
torch.cuda.synchronize()
print("host enqueue %.2f ms / step, wall %.2f ms / step" % (1e3 * issued / n, 1e3 * (time.perf_counter() - t0) / n))
if os.environ.get("SINGLE_THREAD_AUTOGRAD"):
    torch.autograd.set_multithreading_enabled(False)
pr = cProfile.Profile()
pr.enable()
for i in range(n):
    step(model, tasks[i % 2])
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(40)
