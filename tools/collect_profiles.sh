#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 evidence for bench.py's numbers.
#   tools/collect_profiles.sh <outdir> <round-tag>
# Produces (CSV) kernel-trace stats of the default bench command and of the resnet18_84 workload, and the HBM
# traffic counters of the dominant kernel in their own passes (FETCH_SIZE and WRITE_SIZE cannot share a pass; PMC is
# never combined with other trace domains).
set -u
OUT=${1:-gpurun_out/profiles}; TAG=${2:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$R/$OUT"; O="$R/$OUT"
cd /tmp && export TMPDIR=/tmp
# kernels must run one at a time for their durations / counters to be attributable: the support/query stream overlap of
# the timed leg is switched off here, exactly as bench.py does for its own roofline leg
export ORBIT_BENCH_OVERLAP=0 ORBIT_LITE_OVERLAP=0  # (LITE: the H-subset pass serial, not beside the cache pass)
# the inference commands are profiled WITHOUT the LITE block the default line carries since round 6 (its kernels would land in the
# inference kernel statistics); the LITE step has its own profiled command below (--mode lite_train)
export ORBIT_BENCH_LITE_BLOCK=0
for W in efficientnet_b0_224 resnet18_84 ${EXTRA_WORKLOADS:-}; do
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$W -- \
      python $R/bench.py --workload $W --steps 10 --warmup 3 --no-cpu-baseline > $O/${TAG}_bench_$W.json 2> $O/${TAG}_bench_$W.err
  cp $(ls $O/stats_$W/*/*kernel_stats.csv | head -1) $O/${TAG}_kernel_stats_$W.csv
done
# the LITE meta-training step (forward + backward + optimizer) of the headline extractor
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_lite -- \
    python $R/bench.py --mode lite_train --steps 10 --warmup 5 --no-cpu-baseline > $O/${TAG}_bench_lite_train_profiled.json 2> $O/${TAG}_bench_lite.err
cp $(ls $O/stats_lite/*/*kernel_stats.csv | head -1) $O/${TAG}_kernel_stats_lite_train_efficientnet_b0_224.csv
traffic() {  # traffic <name> <workload tag> <kernel substrings, |-separated> <bench.py args...>
  local NAME=$1 WL=$2 KERN=$3; shift 3
  for C in FETCH_SIZE WRITE_SIZE; do
    # (counter collection serialises every dispatch: no settling chunks, eager launches, two steps - a LITE step is ~900 launches)
    ORBIT_BENCH_SETTLE=0 ORBIT_TRAIN_GRAPH=0 ORBIT_GRAPH=0 timeout 600 rocprofv3 --pmc $C --output-format csv -d $O/pmc_${NAME}_$C -- \
        python $R/bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/${TAG}_pmc_${NAME}_$C.err
    F=$(ls $O/pmc_${NAME}_$C/*/*counter_collection.csv | head -1)
    python - "$F" "$C" > $O/${TAG}_pmc_${NAME}_${C}_summary.txt <<'PY'
import csv, sys, collections
f, c = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    if r.get("Counter_Name") != c:
        continue
    k = r["Kernel_Name"].split("(")[0][:90]
    agg[k][0] += 1
    agg[k][1] += float(r["Counter_Value"])
print("# %s per kernel (sum over dispatches, and per launch); unit = KiB as reported by rocprofv3" % c)
for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print("%-92s launches %6d  total %14.0f  per_launch %12.1f" % (k, n, v, v / n))
PY
  done
  # HBM traffic of the kernel family per launch from the two PMC passes (rocprofv3 reports KiB; on gfx950 FETCH_SIZE counts
  # wide 16-B/lane streaming reads at exactly half their bytes -> x2, see MI355X_MICROARCH.md section HBM)
  python - "$O" "$NAME" "$WL" "$KERN" > $O/${TAG}_${NAME}.json <<'PY'
import csv, glob, json, os, subprocess, sys
out, name, wl, kern = sys.argv[1:5]
kern = kern.split("|")
tot = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("%s/pmc_%s_%s/*/*counter_collection.csv" % (out, name, c))[0]
    n, v = 0, 0.0
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") == c and any(k in r["Kernel_Name"] for k in kern):
            n += 1
            v += float(r["Counter_Value"])
    tot[c] = (n, v)
launches = tot["FETCH_SIZE"][0]
fetch = 2.0 * tot["FETCH_SIZE"][1] * 1024 / max(launches, 1)
write = tot["WRITE_SIZE"][1] * 1024 / max(tot["WRITE_SIZE"][0], 1)
sha = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); import bench; print(bench.kernel_sources_sha16())" % os.environ.get("GRAFT_REPO_ROOT", ".")],
                     capture_output=True, text=True).stdout.strip().splitlines()[-1]
print(json.dumps({"workload": wl, "kernel": " + ".join("orbit::" + k for k in kern), "kernel_sources_sha16": sha,
                  "launches_profiled": launches, "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write,
                  "traffic_bytes_per_launch": fetch + write,
                  "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes of bench.py; FETCH_SIZE x2 "
                            "(gfx950 wide-load correction), WRITE_SIZE as reported"}))
PY
}
traffic traffic efficientnet_b0_224 "conv_igemm_kernel|pw_rgemm_kernel" --workload efficientnet_b0_224
# the LITE training step's dense-conv family: forward + data-gradient (conv_igemm) and filter-gradient (conv_wgrad, conv_wgrad_thin) kernels
traffic lite_traffic efficientnet_b0_224:lite_train "conv_igemm_kernel|pw_rgemm_kernel|conv_wgrad_kernel|conv_wgrad_thin_kernel" --mode lite_train --workload efficientnet_b0_224
# ---- distance kernel (64-task batched launch): kernel-trace stats + FETCH_SIZE / WRITE_SIZE in their own passes ----------
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_head -- python $R/tools/head_roofline.py quick > $O/${TAG}_head_roofline.log 2>&1
cp $(ls $O/stats_head/*/*kernel_stats.csv | head -1) $O/${TAG}_kernel_stats_head.csv
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --output-format csv -d $O/pmc_head_$C -- python $R/tools/head_roofline.py quick > /dev/null 2> $O/${TAG}_pmc_head_$C.err
done
python - "$O" "$TAG" > $O/${TAG}_head_traffic.json <<'PY'
import csv, glob, json, os, subprocess, sys
out, tag = sys.argv[1], sys.argv[2]
tot = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("%s/pmc_head_%s/*/*counter_collection.csv" % (out, c))[0]
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f))
            if r.get("Counter_Name") == c and "proto_predict_stream_kernel<8" in r["Kernel_Name"] and int(r["Grid_Size"]) >= 64 * 13 * 512]
    tot[c] = vals
fetch = 2.0 * 1024 * sum(tot["FETCH_SIZE"]) / max(len(tot["FETCH_SIZE"]), 1)
write = 1024 * sum(tot["WRITE_SIZE"]) / max(len(tot["WRITE_SIZE"]), 1)
sha = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); import bench; print(bench.kernel_sources_sha16())" % os.environ.get("GRAFT_REPO_ROOT", ".")],
                     capture_output=True, text=True).stdout.strip().splitlines()[-1]
print(json.dumps({"kernel": "orbit::proto_predict_stream_kernel<8, 2, 5, lean> (64 tasks x 200 x 1280, 5-way)", "kernel_sources_sha16": sha,
                  "launches_profiled": len(tot["FETCH_SIZE"]), "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write,
                  "traffic_bytes_per_launch": fetch + write, "algorithmic_bytes_per_launch": 4.0 * (200 * 1280 + 5 * 1280 + 5 + 200 * 5) * 64,
                  "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes of tools/head_roofline.py quick; "
                            "FETCH_SIZE x2 (gfx950 wide-load correction), WRITE_SIZE as reported"}))
PY
# ---- SQ counters of the dense-conv kernels per layer shape, and of the pointwise register GEMM (own passes; FULL=1 only) -----
cd $R
if [ "${FULL:-0}" = "1" ]; then
  bash tools/conv_pmc.sh $OUT/convpmc effnet_224 > /dev/null 2>&1; cp $O/convpmc/summary.txt $O/${TAG}_conv_sq_counters.txt 2>/dev/null
  bash tools/kernel_pmc.sh $OUT/rgemmpmc pw_rgemm python tools/rgemm_bench.py > /dev/null 2>&1; cp $O/rgemmpmc/summary.txt $O/${TAG}_rgemm_sq_counters.txt 2>/dev/null
fi
# raw rocprofv3 output stays on the box: only the summaries travel back (gpurun merges at most 64 MiB)
rm -rf $O/stats_* $O/pmc_* $O/convpmc/pass* $O/rgemmpmc/pass* $O/bf3pmc/pass*
ls -la $O | head -60
