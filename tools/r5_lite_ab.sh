# GPU box: LITE step A/B on one box (interleaved) + a per-launch trace of one step. bash tools/r5_lite_ab.sh <outdir> "ENV=V ..." ...
O=${1:-gpurun_out/r5d}; shift; mkdir -p $O
for rep in 1 2; do
  for cfg in "$@"; do
    tag=$(echo "$cfg" | tr ' =' '__')
    env $cfg timeout 300 python bench.py --mode lite_train --no-cpu-baseline --steps 30 --warmup 10 > $O/lite_${tag}_$rep.json 2> $O/lite_${tag}_$rep.err
  done
done
R=${GRAFT_REPO_ROOT:-$(pwd)}
# (the traced step runs the H-subset pass serially: per-kernel durations are only attributable one kernel at a time)
cd /tmp && export TMPDIR=/tmp ORBIT_LITE_OVERLAP=0 && rm -rf /tmp/lt && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/lt -o trace -- python $R/tools/lite_trace.py > /dev/null 2>&1
cd $R && python tools/lite_trace.py --parse $(find /tmp/lt -name "trace_kernel_trace.csv" | head -1) $O/lite_timeline.txt > $O/lite_summary.txt 2>&1
python - $O <<'PY'
import json, glob, os, sys
for f in sorted(glob.glob(sys.argv[1] + '/lite_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print("%-60s ms %.2f host %.2f graph %s frac %.3f" % (os.path.basename(f), d['ms_per_step'], d['host_enqueue_ms_per_step'], d['train_graph_calls_replayed_eager'], d['roofline']['frac']))
    except Exception as e:
        print(f, 'ERR', e)
PY
