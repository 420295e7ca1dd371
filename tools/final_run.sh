set -u
# Round-end evidence run (GPU box): full -m gpu suite, smoke, one bench line per workload (inference + LITE training),
# config 5 shape (10-way, 8 tasks per optimizer step on this one GPU). Outputs under gpurun_out/<dir>.
R=$GRAFT_REPO_ROOT; D=${1:-r2final}; O=$R/gpurun_out/$D; mkdir -p $O
cd $R
python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
for w in efficientnet_b0_224 resnet18_84 resnet18_224 cnaps_resnet18_224; do
  python bench.py --workload $w > $O/full_$w.json 2> $O/full_$w.err
  python bench.py --mode lite_train --workload $w > $O/lite_$w.json 2> $O/lite_$w.err
done
for w in cnaps_versa_resnet18_224 simple_cnaps_resnet18_224; do python bench.py --workload $w --no-cpu-baseline > $O/full_$w.json 2> $O/full_$w.err; done
python bench.py --workload efficientnet_b0_224 --way 10 --no-cpu-baseline > $O/full_efficientnet_b0_224_10way.json 2> $O/full_10way.err
python bench.py --mode lite_train --workload efficientnet_b0_224 --way 10 --tasks-per-rank 8 --steps 10 --warmup 3 --no-cpu-baseline > $O/lite_config5_10way_8tasks.json 2> $O/lite_config5.err
python - "$O" <<'PY'
import json, glob, os, sys
for f in sorted(glob.glob(sys.argv[1] + '/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        cb = d.get('cpu_baseline') or {}
        print(os.path.basename(f), round(d['value']), round(d['ms_per_step'], 2), 'frac', round(d['roofline']['frac'], 3), 'share',
              round(d['roofline'].get('kernel_time_share', 0), 2), 'host', round(d.get('host_enqueue_ms_per_step', 0), 2), 'acc',
              round(d.get('frame_accuracy', 0), 3), 'cpu', cb.get('value'), cb.get('cores'), cb.get('max_abs_dlogit_vs_gpu'), cb.get('argmax_identical'))
        b = d.get('opt_in_conv_bf3')
        if b:
            print('    opt-in conv_bf3:', round(b['query_frames_per_s']), round(b['ms_per_step'], 2), 'frac', round(b['conv_frac_of_fp32_mfma_peak'], 3),
                  'dlogit vs oracle', b.get('max_abs_dlogit_vs_oracle'), b.get('argmax_identical_to_oracle'))
    except Exception as e:
        print(f, 'ERR', e)
PY
