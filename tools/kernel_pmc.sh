#!/bin/bash
# Run on the GPU box: SQ counters of kernels matching a name filter for an arbitrary command.
#   tools/kernel_pmc.sh <outdir> <kernel-name-substring> <command...>
set -u
OUT=$1; FILT=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$R/$OUT"; O="$R/$OUT"
cd /tmp && export TMPDIR=/tmp
G1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES"
G2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM"
G3="GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVES_EQ_64 SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC"
i=0
for G in "$G1" "$G2" "$G3"; do
  i=$((i+1))
  (cd $R && timeout 600 rocprofv3 --pmc $G --output-format csv -d $O/pass$i -- "$@" > $O/pass$i.log 2> $O/pass$i.err)
done
python - "$O" "$FILT" <<'PY' > $O/summary.txt
import csv, glob, sys, collections
out, filt = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
meta = {}
for f in glob.glob(out + "/pass*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if filt not in k:
            continue
        key = (k.split("(")[0][-60:], int(r["Grid_Size"]))
        a = agg[key][r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
        meta[key] = (r.get("Arch_VGPR_Count") or r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"))
for key in sorted(agg, key=lambda k: -k[1]):
    v = {n: x[1] / x[0] for n, x in agg[key].items()}
    if "SQ_WAVES" not in v or "GRBM_GUI_ACTIVE" not in v:
        continue
    w = v["SQ_WAVES"]; gui = v["GRBM_GUI_ACTIVE"] / 8; cap = gui * 1024
    print("%-60s grid %9d vgpr/agpr/sgpr/lds %s" % (key[0], key[1], meta[key]))
    print("   us %7.1f valu/w %5.0f salu/w %5.0f lds/w %4.0f vmem/w %3.0f mfma/w %4.0f | VALU %.2f SCA %.2f MFMA %.2f LDS %.2f (conflict %.2f) | waves/simd %.1f | wait_any %.2f wait_inst %.2f wait_lds %.3f" % (
        gui / 2150, v["SQ_INSTS_VALU"] / w, v["SQ_INSTS_SALU"] / w, v["SQ_INSTS_LDS"] / w, v["SQ_INSTS_VMEM_RD"] / w, v["SQ_VALU_MFMA_BUSY_CYCLES"] / 64 / w,
        v["SQ_ACTIVE_INST_VALU"] * 4 / cap, v["SQ_ACTIVE_INST_SCA"] * 4 / cap, v["SQ_VALU_MFMA_BUSY_CYCLES"] / cap, v["SQ_LDS_IDX_ACTIVE"] / (gui * 256),
        v["SQ_LDS_BANK_CONFLICT"] / max(v["SQ_LDS_IDX_ACTIVE"], 1), v["SQ_WAVE_CYCLES"] * 4 / cap, v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"], v["SQ_WAIT_INST_ANY"] / v["SQ_WAVE_CYCLES"], v["SQ_WAIT_INST_LDS"] / v["SQ_WAVE_CYCLES"]))
PY
cat $O/summary.txt
