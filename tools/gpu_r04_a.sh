#!/bin/bash
# Round 4, first GPU call: the whole GPU suite on the new code, the bench line, an A/B of the render kernels against round 3's
# library (build/lib_base), and PMC passes of the two dominated kernel families (pixel, persistent: VERDICT r3 item 7).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r04a
mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log
tail -4 $OUT/pytest_gpu.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_line.json 2> $OUT/bench.err
echo "bench exit $?"
bash tools/gpu_ab.sh r04a/ab <<'AB'
base|rgbbox|1000|-r 20|
new|rgbbox|1000|-r 20|
base|irreg|1000|-r 20|
new|irreg|1000|-r 20|
base|rgbbox|1000|-r 0 -B 20|
new|rgbbox|1000|-r 0 -B 20|
base|irreg|1000|-r 0 -B 20|
new|irreg|1000|-r 0 -B 20|
base|irreg|4000|-r 5|
new|irreg|4000|-r 5|
base|big|2000|-r 3|
new|big|2000|-r 3|
base|rgbbox|1000|-r 20|
new|rgbbox|1000|-r 20|
base|irreg|1000|-r 0 -B 20|
new|irreg|1000|-r 0 -B 20|
AB
cd /tmp
for v in 1 2; do for s in rgbbox irreg; do
  i=0
  for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE FETCH_SIZE TCC_HIT_sum TCC_MISS_sum"; do
    d=$OUT/fam_v${v}_${s}_p$i
    timeout 200 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $d -- $OLDPWD/build/rtbench -s $s -n 1000 -m 1000 -r 4 -v $v > $d.log 2>&1
    i=$((i+1))
  done
done; done
cd $OLDPWD
python - "$OUT" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
rows = []
for d in sorted(glob.glob(os.path.join(out, "fam_*_p[0-9]"))):
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
with open(os.path.join(out, "pmc_dominated_families.csv"), "w") as f:
    f.write("run,kernel,counter,dispatches,mean_value\n")
    for r in rows:
        f.write("%s,\"%s\",%s,%d,%.1f\n" % r)
print("wrote", len(rows), "rows")
PY
rm -rf $OUT/fam_*_p[0-9]/
echo r04a done
