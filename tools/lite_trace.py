"""One LITE meta-training step under rocprofv3 --kernel-trace: per-launch timeline of the LAST step (between the last two
optimizer launches), the 40 longest launches and the per-kernel totals of that step.
usage (GPU box): cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --output-format csv -d <out> -o trace -- python tools/lite_trace.py [workload]
then: python tools/lite_trace.py --parse <out>/.../trace_kernel_trace.csv"""
import collections
import csv
import os
import sys

if len(sys.argv) > 1 and sys.argv[1] == "--parse":
    rows = list(csv.DictReader(open(sys.argv[2])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    opt = [i for i, r in enumerate(rows) if "FusedAdam" in r["Kernel_Name"]]
    # consecutive FusedAdam launches belong to one optimizer.step(): step boundaries = first launch of each run
    starts = [i for k, i in enumerate(opt) if k == 0 or opt[k - 1] != i - 1]
    lo, hi = starts[-2], starts[-1]
    step = rows[lo:hi]
    short = lambda r: r["Kernel_Name"].replace("orbit::", "").replace("void ", "").split("(")[0][:78]
    dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    total = sum(dur(r) for r in step)
    span = (int(step[-1]["End_Timestamp"]) - int(step[0]["Start_Timestamp"])) / 1e3
    print("# last LITE step: %d launches, sum of kernels %.1f us, span %.1f us" % (len(step), total, span))
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in step:
        a = agg[short(r)]
        a[0] += 1
        a[1] += dur(r)
    print("# per kernel (this step)")
    for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
        print("%9.1f us  %4d x %7.1f  %5.1f %%  %s" % (t, n, t / n, 100 * t / total, k))
    print("# 40 longest launches")
    for r in sorted(step, key=lambda r: -dur(r))[:40]:
        print("%9.1f us  grid %-16s %s" % (dur(r), "%sx%sx%s" % (int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1),
                                                                 r["Grid_Size_Y"], r["Grid_Size_Z"]), short(r)))
    if len(sys.argv) > 3:
        with open(sys.argv[3], "w") as f:
            prev = None
            for r in step:
                gap = 0.0 if prev is None else (int(r["Start_Timestamp"]) - prev) / 1e3
                prev = int(r["End_Timestamp"])
                f.write("%8.1f us  gap %6.1f  grid %-16s %s\n" % (
                    dur(r), gap, "%sx%sx%s" % (int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), r["Grid_Size_Y"],
                                               r["Grid_Size_Z"]), short(r)))
    sys.exit(0)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from orbit_dataset_amd import synthetic  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "efficientnet_b0_224"
device = torch.device("cuda", 0)
torch.cuda.set_device(0)
model = bench.build_model(workload, device, 256, train=True)
step = bench.LiteTrainStep(model, 1, 256, 1)
size = bench.WORKLOADS[workload][2]
tasks = [synthetic.make_task_on_device(i, 5, 1, 40, 200, size, 1, device) for i in range(2)]
for t in tasks:
    model.classifier.unique_labels(t["context_labels"], device)
for i in range(8):
    step(model, tasks[i % 2])
torch.cuda.synchronize()
