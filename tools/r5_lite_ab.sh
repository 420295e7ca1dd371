mkdir -p gpurun_out/r5c
timeout 900 python -m pytest tests -m gpu -q --durations=25 -p no:cacheprovider > gpurun_out/r5c/pytest.log 2>&1; tail -3 gpurun_out/r5c/pytest.log
for cfg in "ORBIT_TRAIN_DW_XF=1" "ORBIT_TRAIN_DW_XF=0" "ORBIT_TRAIN_DW_XF=1 ORBIT_BENCH_FRESH_LABELS=0" "ORBIT_TRAIN_DW_XF=1 ORBIT_TRAIN_GRAPH=0" "ORBIT_TRAIN_DW_XF=1" "ORBIT_TRAIN_DW_XF=0"; do
  tag=$(echo "$cfg" | tr ' =' '__')
  env $cfg timeout 300 python bench.py --mode lite_train --no-cpu-baseline --steps 30 --warmup 10 > gpurun_out/r5c/lite_$tag.$RANDOM.json 2> gpurun_out/r5c/lite_$tag.err
done
cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/lt -o trace -- python $GRAFT_REPO_ROOT/tools/lite_trace.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python tools/lite_trace.py --parse $(find /tmp/lt -name "trace_kernel_trace.csv" | head -1) gpurun_out/r5c/lite_timeline.txt > gpurun_out/r5c/lite_summary.txt 2>&1
tail -2 gpurun_out/r5c/lite_summary.txt
