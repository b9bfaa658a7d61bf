#!/bin/bash
# Experiment 3: ticket counters taking turns over one queue (xcd_queues=2) against strips (1) and one counter (0).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
TAG=${1:-exp3}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
timeout 400 python -m pytest tests -m gpu -x -q -k "tile_queue_layouts or golden_500 or adaptive_tile_order or irreg_4000 or big_2000 or batch_of_frames or repeated_launches or bench_configuration" > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log
RT_XCD_QUEUES=2 timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pixels_bit_exact or parts_assemble or stacked_parts or multi_device_context or many_views or bounce_limit or tall_trees" > $OUT/pytest_x2.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_x2.log
tail -3 $OUT/pytest_x2.log
ab() {  # scene size mode opts...
  local s=$1 n=$2 mode=$3; shift 3
  local o=""; for kv in "$@"; do o="$o -o $kv"; done
  echo "$s $n $mode [$*] : $(timeout 120 ./build/rtbench -s $s -n $n -m $n $mode $o 2>&1 | grep -E "HIP-event|Checksum|failed|unknown" | tr '\n' ' ')"
}
{
for rep in 1 2; do
  for x in 0 2 1; do
    ab rgbbox 1000 "-r 20" xcd_queues=$x
    ab irreg 1000 "-r 20" xcd_queues=$x
  done
done
for x in 0 2 1; do
  ab rgbbox 2000 "-r 10" xcd_queues=$x
  ab irreg 2000 "-r 10" xcd_queues=$x
  ab big 2000 "-r 5" xcd_queues=$x tpt_log2=0
  ab irreg 4000 "-r 8" xcd_queues=$x tpt_log2=0
done
ab irreg 4000 "-r 8" xcd_queues=2 tpt_log2=1
ab irreg 4000 "-r 8" xcd_queues=2 tpt_log2=2
ab big 2000 "-r 5" xcd_queues=2 tpt_log2=1
ab rgbbox 2000 "-r 10" xcd_queues=2 tpt_log2=1
ab rgbbox 1000 "-r 20" xcd_queues=2 deep_split=3
ab irreg 1000 "-r 20" xcd_queues=2 deep_split=3
ab rgbbox 1000 "-r 20" xcd_queues=0 deep_split=3
ab irreg 1000 "-r 20" xcd_queues=0 deep_split=3
ab rgbbox 500 "-r 20" xcd_queues=0
ab rgbbox 500 "-r 20" xcd_queues=2
ab irreg 500 "-r 20" xcd_queues=0
ab irreg 500 "-r 20" xcd_queues=2
} > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
for x in 0 2; do
  echo "== RT_XCD_QUEUES=$x" >> $OUT/rank_share.txt
  RT_XCD_QUEUES=$x timeout 200 python tools/rank_share_probe.py 20 1,8 1 2 2s >> $OUT/rank_share.txt 2>&1
done
cat $OUT/rank_share.txt
echo exp3 done
