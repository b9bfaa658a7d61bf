#!/bin/bash
# PMC passes over the native bench (rtbench), counters in their own runs (rocprofv3 serialises the
# launches it counts, so these are one-launch-at-a-time figures; per-launch instruction counts do not
# depend on overlap).  usage: gpu_pmc.sh <outdir under gpurun_out> ["<grid_div list>"] ["<scene list>"]
# Output: <outdir>/pmc_summary.csv  (run = <scene>_gd<grid_div>, kernel, counter, dispatches, mean)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/${1:-pmc}
GDS=${2:-"0 4"}
SCENES=${3:-"rgbbox irreg"}
mkdir -p $OUT
PASSES=(
"SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
"SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT"
"SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU"
"GRBM_GUI_ACTIVE FETCH_SIZE"
"GRBM_COUNT WRITE_SIZE"
)
cd /tmp
for s in $SCENES; do for gd in $GDS; do
  np=${#PASSES[@]}
  [ "$gd" != "0" ] && np=2        # other launch sizes: instruction counts and LDS cycles only
  for ((i = 0; i < np; i++)); do
    d=$OUT/${s}_gd${gd}_p${i}
    timeout 300 rocprofv3 --pmc ${PASSES[$i]} --kernel-trace --output-format csv -d $d -- $OLDPWD/build/rtbench -s $s -n 1000 -m 1000 -r 8 -v 3 -o sync_policy=1 $([ "$gd" = "0" ] || echo "-o grid_div=$gd -o deep_class=0") > $d.log 2>&1
  done
done
# batch launches (rt_render_batch, what bench.py times): 20 frames per launch, batch-only mode of rtbench
for i in 0 1 3 4; do
  d=$OUT/${s}_batch20_p${i}
  timeout 300 rocprofv3 --pmc ${PASSES[$i]} --kernel-trace --output-format csv -d $d -- $OLDPWD/build/rtbench -s $s -n 1000 -m 1000 -r 0 -B 20 > $d.log 2>&1
done
done
# the 10^6-sphere frame (configs[4]): L2 hit rate and memory-side traffic -- the one workload whose scene exceeds the L2s
if [ -z "$PMC_NO_BIG" ]; then
  i=0
  for pass in "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE FETCH_SIZE SQ_INSTS_VMEM" "GRBM_COUNT WRITE_SIZE SQ_INSTS_VALU"; do
    d=$OUT/big2000_p$i
    timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $d -- $OLDPWD/build/rtbench -s big -n 2000 -m 2000 -r 3 > $d.log 2>&1
    i=$((i+1))
  done
fi
cd $OLDPWD
python - "$OUT" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
rows = []
for d in sorted(glob.glob(os.path.join(out, "*_p[0-9]"))):
    run = os.path.basename(d).rsplit("_p", 1)[0]
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            acc[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in sorted(acc.items()):
            rows.append((run, k, c, len(v), sum(v) / len(v)))
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:60]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        for k, v in sorted(acc.items()):
            rows.append((run, k, "kernel_ns_p" + d[-1], len(v), sum(v) / len(v)))
with open(os.path.join(out, "pmc_summary.csv"), "w") as f:
    f.write("run,kernel,counter,dispatches,mean_value\n")
    for r in rows:
        f.write("%s,\"%s\",%s,%d,%.1f\n" % r)
print("wrote", len(rows), "rows")
PY
rm -rf $OUT/*_p[0-9]/
echo pmc done
