"""One eager extractor forward under rocprofv3 --kernel-trace: per-launch timeline (launch order = layer order).
usage (GPU box): cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace -d <out> -o trace -- python tools/layer_trace.py efficientnet_b0 224 200
then: python tools/layer_trace.py --parse <out>/.../trace_kernel_trace.csv"""
import csv
import os
import sys

if len(sys.argv) > 1 and sys.argv[1] == "--parse":
    rows = list(csv.DictReader(open(sys.argv[2])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    names = [r["Kernel_Name"] for r in rows]
    # the last forward = the rows after the last bn_fold_all_kernel launch
    last = max(i for i, n in enumerate(names) if "bn_fold_all" in n)
    total = 0.0
    prev_end = None
    for r in rows[last:]:
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        gap = 0.0 if prev_end is None else (int(r["Start_Timestamp"]) - prev_end) / 1e3
        prev_end = int(r["End_Timestamp"])
        total += dur
        n = r["Kernel_Name"].replace("orbit::", "").replace("void ", "")
        n = n.split("(")[0][:70]
        print("%8.1f us  gap %6.1f  grid %-14s %s" % (dur, gap, "%sx%sx%s" % (int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), r["Grid_Size_Y"], r["Grid_Size_Z"]), n))
    span = (int(rows[-1]["End_Timestamp"]) - int(rows[last]["Start_Timestamp"])) / 1e3
    print("sum of kernels %.1f us, span %.1f us, %d launches" % (total, span, len(rows) - last))
    sys.exit(0)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import orbit_dataset_amd  # noqa: F401
from orbit_dataset_amd import _lib, synthetic
from orbit_dataset_amd.model.feature_extractors import create_feature_extractor

name = sys.argv[1] if len(sys.argv) > 1 else "efficientnet_b0"
size = int(sys.argv[2]) if len(sys.argv) > 2 else 224
B = int(sys.argv[3]) if len(sys.argv) > 3 else 200
_lib.load().orbit_set_option(b"graph", 0)
fe, _ = create_feature_extractor(name, True, False, False)
synthetic.init_parameters_(fe)
fe = fe.cuda().eval()
x = torch.randn(B, 3, size, size, device="cuda")
with torch.no_grad():
    for _ in range(6):
        fe(x)
    torch.cuda.synchronize()
    fe.sync()
    for pl in fe._plans.values():
        pl.stamp = None  # force one more upload + bn_fold_all: marks the start of the last forward in the trace
    fe(x)
    torch.cuda.synchronize()
