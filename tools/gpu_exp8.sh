#!/bin/bash
# Experiment 8: dynamic instruction counts of a 20-frame batch launch, library at 504921f (build/lib_base) vs this tree
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/${1:-exp8}
mkdir -p $OUT
R=$PWD
cd /tmp
for lib in base new; do for s in rgbbox irreg; do
  d=$OUT/${lib}_${s}
  ( [ "$lib" = base ] && export LD_LIBRARY_PATH=$R/build/lib_base:$LD_LIBRARY_PATH
    timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $d -- $R/build/rtbench -s $s -n 1000 -m 1000 -r 0 -B 20 > $d.log 2>&1 )
done; done
cd $R
python - "$OUT" <<'PY' > $OUT/counts.txt 2>&1
import csv, glob, os, sys, collections
out = sys.argv[1]
for d in sorted(glob.glob(os.path.join(out, "*_*"))):
    if not os.path.isdir(d): continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "pooled" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        print(os.path.basename(d), {c: round(max(v) / 20 / 1e6, 3) for c, v in sorted(acc.items())}, "(per frame, millions; largest dispatch = the batch)")
PY
cat $OUT/counts.txt
rm -rf $OUT/*_rgbbox $OUT/*_irreg
bash tools/gpu_exp6.sh ${1:-exp8}_ab > /dev/null 2>&1
cat gpurun_out/${1:-exp8}_ab/ab.txt | grep -v "4000\|big"
