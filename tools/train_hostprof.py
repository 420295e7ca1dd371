"""cProfile of the host side of the LITE training step (GPU box)."""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from orbit_dataset_amd import synthetic
dev = torch.device("cuda:0")
w = sys.argv[1] if len(sys.argv) > 1 else "resnet18_84"
model = bench.build_model(w, dev, 256, train=True)
step = bench.LiteTrainStep(model, 1, 256)
size = bench.WORKLOADS[w][2]
tasks = [synthetic.make_task_on_device(i, 5, 5, 8, 200, size, 1, dev) for i in range(2)]
for i in range(3): step(model, tasks[i % 2])
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for i in range(10): step(model, tasks[i % 2])
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumtime").print_stats(45); print(s.getvalue()[:9000])
