"""How long does a fresh process / fresh box take to reach steady state? Per-chunk timing of the default bench step."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from orbit_dataset_amd import synthetic
dev = torch.device("cuda:0")
model = bench.build_model("efficientnet_b0_224", dev, 256)
tasks = [synthetic.make_task_on_device(i, 5, 5, 8, 200, 224, 1, dev) for i in range(4)]
for t in tasks: model.classifier.unique_labels(t["context_labels"], dev)
import gc; gc.collect(); gc.freeze()
for i in range(3): bench.run_task(model, tasks[i % 4])
out = []
for chunk in range(12):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(10): bench.run_task(model, tasks[i % 4])
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    out.append("%.2f(%.2f)" % (1e3 * (t2 - t0) / 10, 1e3 * (t1 - t0) / 10))
print("ms/step(host) per chunk of 10:", " ".join(out))
