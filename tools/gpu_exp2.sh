#!/bin/bash
# Experiment 2: the BOX append diet (masked stores, scalar list bases) and deep-aware multi-tile tickets against the previous
# build of the library (build/lib_base), parity first.  usage: gpu_exp2.sh <tag>
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
TAG=${1:-exp2}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log
ab() {  # lib scene size mode label opts...
  local lib=$1 s=$2 n=$3 mode=$4; shift 4
  local o=""; for kv in "$@"; do o="$o -o $kv"; done
  local res=$( ( [ "$lib" = base ] && export LD_LIBRARY_PATH=$PWD/build/lib_base:$LD_LIBRARY_PATH; timeout 120 ./build/rtbench -s $s -n $n -m $n $mode $o 2>&1 ) | grep -E "HIP-event|Checksum|Batch|Overlapped|failed|unknown" | tr '\n' ' ')
  echo "$lib $s $n $mode [$*] : $res"
}
{
export GPU_MAX_HW_QUEUES=20
for rep in 1 2; do
  for lib in base new; do
    for s in rgbbox irreg; do
      ab $lib $s 1000 "-r 0 -B 20"
      ab $lib $s 1000 "-r 20"
    done
  done
done
for lib in base new; do
  for s in rgbbox irreg; do ab $lib $s 1000 "-r 0 -B 200"; ab $lib $s 1000 "-r 20 -L 24" grid_div=4 deep_class=0; done
done
for cfg in "xcd_queues=1 tpt_log2=0" "xcd_queues=1 tpt_log2=1" "xcd_queues=1 tpt_log2=2" "xcd_queues=0 tpt_log2=2" "xcd_queues=0 tpt_log2=0"; do
  ab new irreg 4000 "-r 8" $cfg
  ab new big 2000 "-r 5" $cfg
done
ab base irreg 4000 "-r 8" xcd_queues=1 tpt_log2=0
ab base big 2000 "-r 5" xcd_queues=1 tpt_log2=0
ab new irreg 2000 "-r 10" xcd_queues=1
ab new irreg 2000 "-r 10" xcd_queues=0
ab new rgbbox 2000 "-r 10" xcd_queues=1
ab new rgbbox 2000 "-r 10" xcd_queues=0
} > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_line.json 2> $OUT/bench.err
python - $OUT/bench_line.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("bench value", d["value"], "ms_per_step", d["ms_per_step"], "serial", d.get("serial_ms_per_frame"), "verified", d.get("verified"))
except Exception as e:
    print("bench line unreadable:", e)
PY
tail -3 $OUT/bench.err
echo exp2 done
