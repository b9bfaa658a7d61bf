#!/bin/bash
# The first-process effect of a fresh box: one mode per fresh box.
#   pmc           bench.py twice, each under rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE: the batch launches' cycle counts
#                 next to their durations -> was the first process clocked lower, or did it need more cycles?
#   native_child  build/hip_touch (hipInit + an empty launch), then bench.py twice
#   torch_child   python -c "import torch; one tiny kernel", then bench.py twice
#   none          bench.py three times (the control)
# usage: first_process_probe.sh <mode> <tag>     -> gpurun_out/<tag>/first_process_<mode>.txt
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
MODE=$1
OUT=$PWD/gpurun_out/${2:-fp}
mkdir -p $OUT
LOG=$OUT/first_process_$MODE.txt
B="bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-serial-extra"
line() { python - "$1" "$2" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith('{"metric"'):
        d = json.loads(l)
        print(sys.argv[2], d["value"], d["ms_per_step"], d["roofline"].get("per_launch", {}).get("kernel_ms"))
PY
}
smi() { rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|Power|Temperature \(Sensor (edge|junction)" | tr -s ' ' | tr '\n' ';'; echo; }
{
echo "mode $MODE"; echo -n "before: "; smi
case $MODE in
native_child) ./build/hip_touch ;;
torch_child) python -c "import torch; x = torch.zeros(64, device='cuda') + 1; torch.cuda.synchronize(); print('torch touched')" ;;
esac
if [ $MODE = pmc ]; then
  for i in 1 2; do
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU --output-format csv -d $OUT/fp_p$i -- python $OLDPWD/$B > $OUT/fp_$i.log 2>&1)
    line $OUT/fp_$i.log "process $i (under rocprofv3 --pmc)"
    python - $OUT/fp_p$i <<'PY'
import csv, glob, os, sys, collections
d = sys.argv[1]
dur = {}
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "pooled_kernel" in r["Kernel_Name"]:
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Grid_Size_X"] if "Grid_Size_X" in r else "")
cnt = collections.defaultdict(dict)
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "pooled_kernel" in r["Kernel_Name"]:
            cnt[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
for k in sorted(dur, key=int):
    ns = dur[k][0]
    c = cnt.get(k, {})
    if ns < 1.5e6:      # the batch launches only
        continue
    g = c.get("GRBM_GUI_ACTIVE", 0.0)
    print("  dispatch %s: %.3f ms  GRBM_GUI_ACTIVE %.0f (%.1f MHz x XCDs)  SQ_BUSY_CYCLES %.0f  SQ_WAVE_CYCLES %.4g  SQ_INSTS_VALU %.0f"
          % (k, ns / 1e6, g, g / ns * 1e3, c.get("SQ_BUSY_CYCLES", 0), c.get("SQ_WAVE_CYCLES", 0), c.get("SQ_INSTS_VALU", 0)))
PY
    rm -rf $OUT/fp_p$i
  done
else
  for i in 1 2 3; do
    [ $i = 3 ] && [ $MODE != none ] && break
    timeout 400 python $B > $OUT/fp_$i.log 2>&1
    line $OUT/fp_$i.log "bench.py process $i"
    echo -n "  after: "; smi
  done
fi
} > $LOG 2>&1
cat $LOG
