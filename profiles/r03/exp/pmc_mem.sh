#!/bin/bash
# the vector-memory path of the frames whose scene is not LDS resident: TA / TCP / UTCL1 / TCC counters, one pass each
# usage: pmc_mem.sh <tag> "<rtbench args>" ...   -> gpurun_out/<tag>/pmc_mem.txt
cd "$(dirname "$0")/../../.."
export TMPDIR=/tmp
TAG=$1; shift
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
# (at most two or three counters of one block per pass: a pass the hardware cannot schedule aborts -- and then hangs until its timeout)
PASSES=(
"GRBM_GUI_ACTIVE TA_BUSY_avr TA_BUFFER_READ_WAVEFRONTS_sum"
"TA_BUFFER_TOTAL_CYCLES_sum SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD"
"TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"
"TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum"
"TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum"
)
: > $OUT/pmc_mem.txt
for args in "$@"; do
  echo "== rtbench $args" >> $OUT/pmc_mem.txt
  for ((i = 0; i < ${#PASSES[@]}; i++)); do
    d=$OUT/pm_p$i
    (cd /tmp && timeout 45 rocprofv3 --pmc ${PASSES[$i]} --kernel-trace --output-format csv -d $d -- $OLDPWD/build/rtbench $args > $d.log 2>&1)
    python - $d >> $OUT/pmc_mem.txt <<'PY'
import csv, glob, os, sys, collections
d = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "pooled_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "pooled_kernel" in r["Kernel_Name"]:
            dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
if not acc:
    print("  (no counters: see", d + ".log)")
for k, v in sorted(acc.items()):
    v = v[1:] if len(v) > 1 else v       # not the first (recording) frame
    print("  %-44s %16.0f   (%d launches, kernel %.3f ms)" % (k, sum(v) / len(v), len(v), (sum(dur[1:]) / max(1, len(dur) - 1)) / 1e6 if len(dur) > 1 else dur[0] / 1e6))
PY
    rm -rf $d
  done
done
cat $OUT/pmc_mem.txt
