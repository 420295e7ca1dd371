# per-launch timeline of one extractor forward (GPU box): bash tools/layer_trace.sh <out.txt> <net> <size> <frames>
OUT=$1; NET=${2:-efficientnet_b0}; SIZE=${3:-224}; B=${4:-200}
R=${GRAFT_REPO_ROOT:-$(pwd)}
D=/tmp/lt_$$; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $D -o trace -- python $R/tools/layer_trace.py $NET $SIZE $B > $D.log 2>&1
F=$(find $D -name "*kernel_trace.csv" | head -1)
cd $R && python tools/layer_trace.py --parse $F > $OUT 2>&1
tail -1 $OUT
