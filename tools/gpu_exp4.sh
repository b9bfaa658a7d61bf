#!/bin/bash
# Experiment 4: the batch entry with eight counters taking turns, tiles per ticket 0..2; defaults re-checked.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
TAG=${1:-exp4}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
timeout 400 python -m pytest tests -m gpu -x -q -k "tile_queue_layouts or batch_of_frames or bench_configuration or bench_line_contract or golden_500" > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log
ab() {  # scene size mode opts...
  local s=$1 n=$2 mode=$3; shift 3
  local o=""; for kv in "$@"; do o="$o -o $kv"; done
  echo "$s $n $mode [$*] : $(timeout 120 ./build/rtbench -s $s -n $n -m $n $mode $o 2>&1 | grep -E "HIP-event|Checksum|Batch|failed|unknown" | tr '\n' ' ')"
}
{
for rep in 1 2; do
  for cfg in "xcd_queues=0 tpt_log2=2" "xcd_queues=2 tpt_log2=2" "xcd_queues=2 tpt_log2=1" "xcd_queues=2 tpt_log2=0"; do
    ab rgbbox 1000 "-r 0 -B 20" $cfg
    ab irreg 1000 "-r 0 -B 20" $cfg
  done
done
for cfg in "xcd_queues=0 tpt_log2=2" "xcd_queues=2 tpt_log2=2" "xcd_queues=2 tpt_log2=1" "xcd_queues=2 tpt_log2=0"; do
  ab rgbbox 1000 "-r 0 -B 200" $cfg
  ab irreg 1000 "-r 0 -B 200" $cfg
done
ab rgbbox 1000 "-r 20"
ab irreg 1000 "-r 20"
ab irreg 4000 "-r 8"
ab big 2000 "-r 5"
} > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
echo exp4 done
