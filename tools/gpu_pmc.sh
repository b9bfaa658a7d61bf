#!/bin/bash
# PMC passes over rtbench (one kernel family per run).  usage: gpu_pmc.sh <outdir> <variant list> <scene list>
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/${1:-pmc}
VARS=${2:-"1 2"}
SCENES=${3:-"rgbbox irreg"}
mkdir -p $OUT
PASSES=(
"SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
"SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU"
"SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS_F32 SQ_IFETCH SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INSTS_LDS_ATOMIC"
"GRBM_GUI_ACTIVE FETCH_SIZE"
"GRBM_COUNT WRITE_SIZE"
)
# PMC_FIRST_ONLY=1: only the instruction-count pass
if [ -n "$PMC_FIRST_ONLY" ]; then PASSES=("${PASSES[0]}"); fi
cd /tmp
for s in $SCENES; do for v in $VARS; do
  i=0
  for pass in "${PASSES[@]}"; do
    d=$OUT/${s}_v${v}_p${i}
    timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $d -- $OLDPWD/build/rtbench -s $s -n 1000 -m 1000 -r 3 -v $v $EXTRA_OPTS > $d.log 2>&1
    i=$((i+1))
  done
done; done
cd $OLDPWD
# condense: per kernel name, mean of each counter
python - "$OUT" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
rows = []
for d in sorted(glob.glob(os.path.join(out, "*_p[0-9]"))):
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            acc[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in sorted(acc.items()):
            rows.append((os.path.basename(d), k, c, len(v), sum(v) / len(v)))
with open(os.path.join(out, "pmc_summary.csv"), "w") as f:
    f.write("run,kernel,counter,dispatches,mean_value\n")
    for r in rows:
        f.write("%s,\"%s\",%s,%d,%.1f\n" % r)
print("wrote", len(rows), "rows")
PY
echo pmc done
