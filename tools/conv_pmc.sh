#!/bin/bash
# Run on the GPU box: SQ counters of the conv kernel per layer shape (tools/conv_bench.py <net>), one rocprofv3 --pmc pass
# per counter group (PMC passes are never combined with other trace domains).
#   tools/conv_pmc.sh <outdir> <net>
set -u
OUT=${1:-gpurun_out/convpmc}; NET=${2:-effnet_224}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$R/$OUT"; O="$R/$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z0-9_]*\|GRBM_[A-Z_]*\|TCP_[A-Z0-9_]*\|TA_[A-Z0-9_]*" | sort -u > $O/counters_avail.txt
G1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES"
G2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM"
G3="GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVES_EQ_64 SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC"
i=0
for G in "$G1" "$G2" "$G3"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $G --output-format csv -d $O/pass$i -- python $R/tools/conv_bench.py $NET > $O/pass$i.log 2> $O/pass$i.err
done
python - "$O" <<'PY' > $O/summary.txt
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
meta = {}
for f in glob.glob(out + "/pass*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "conv_igemm" not in k and "pw_conv" not in k:
            continue
        short = k.split("<", 1)[1].split(">")[0] if "<" in k else k
        key = (short, int(r["Grid_Size"]))
        a = agg[key][r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
        meta[key] = (r.get("Arch_VGPR_Count") or r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"))
for key in sorted(agg, key=lambda k: -k[1]):
    c = {n: v[1] / v[0] for n, v in agg[key].items()}
    print("== tmpl<%s> grid %d (blocks %d)  vgpr/agpr/sgpr/lds %s" % (key[0], key[1], key[1] // 256, meta[key]))
    wc = c.get("SQ_WAVE_CYCLES", 0) or 1
    print("   " + "  ".join("%s=%.4g" % (n.replace("SQ_", ""), v) for n, v in sorted(c.items())))
    print("   per wave-cycle: wait_any %.2f  wait_inst %.2f  active %.2f  wait_lds %.3f | mfma_busy/busy %.3f | lds conflict/idx %.3f" % (
        c.get("SQ_WAIT_ANY", 0) / wc, c.get("SQ_WAIT_INST_ANY", 0) / wc, c.get("SQ_ACTIVE_INST_ANY", 0) / wc,
        c.get("SQ_WAIT_INST_LDS", 0) / wc, c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(c.get("SQ_BUSY_CYCLES", 1), 1),
        c.get("SQ_LDS_BANK_CONFLICT", 0) / max(c.get("SQ_LDS_IDX_ACTIVE", 1), 1)))
PY
head -c 6000 $O/summary.txt
