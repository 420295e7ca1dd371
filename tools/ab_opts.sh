# whole-task A/B of runtime options (GPU box): bash tools/ab_opts.sh <outdir> "ENV=V ENV2=V2" "..." ...   (two rounds, interleaved)
O=$1; shift; mkdir -p $O
i=0; for cfg in "$@"; do i=$((i+1)); echo "$cfg" > $O/cfg_$i.txt; done
for rep in 1 2; do i=0; for cfg in "$@"; do i=$((i+1)); env $cfg python bench.py --no-cpu-baseline --steps 40 --warmup 10 > $O/b_${i}_$rep.json 2>$O/b_${i}_$rep.err; done; done
python - "$O" "$#" <<'PY'
import json, sys
o, n = sys.argv[1], int(sys.argv[2])
for i in range(1, n + 1):
    v = []
    for rep in (1, 2):
        try:
            d = json.loads(open("%s/b_%d_%d.json" % (o, i, rep)).read().strip().splitlines()[-1])
            v.append((round(d["ms_per_step"], 3), round(d.get("value_overlap_off", 0)), round(d["roofline"]["frac"], 3)))
        except Exception as e:
            v.append(("ERR", str(e)[:40]))
    print("%-50s %s" % (open("%s/cfg_%d.txt" % (o, i)).read().strip(), v))
PY
