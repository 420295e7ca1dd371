O=gpurun_out/r4f; mkdir -p $O
for rep in 1 2; do for m in 0 17 18 20 21 23 19 22; do ORBIT_CONV_RGEMM=$m python bench.py --no-cpu-baseline --steps 40 --warmup 10 > $O/b_${m}_$rep.json 2>/dev/null; done; done
python - <<PY
import json,glob
for m in (0,17,18,20,21,23,19,22):
    v=[]
    for rep in (1,2):
        d=json.loads(open("gpurun_out/r4f/b_%d_%d.json"%(m,rep)).read().strip().splitlines()[-1])
        v.append((round(d["ms_per_step"],3), round(d.get("value_overlap_off",0)), round(d["roofline"]["frac"],3)))
    print("mask", m-16 if m else "off", v)
PY
