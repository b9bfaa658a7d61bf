#!/bin/bash
# Experiment 11: one-frame knobs re-swept on the final queue (deep_class, deep_split, thr_shade, grid_div, prio_depth)
cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out/${1:-exp11}
mkdir -p $OUT
ab() {
  local s=$1 n=$2 mode=$3; shift 3
  local o=""; for kv in "$@"; do o="$o -o $kv"; done
  echo "$s $n $mode [$*] : $(timeout 120 ./build/rtbench -s $s -n $n -m $n $mode $o 2>&1 | grep -E "HIP-event|Checksum|failed|unknown" | tr '\n' ' ')"
}
{
for s in rgbbox irreg; do
  ab $s 1000 "-r 30"
  for cfg in "deep_class=2" "deep_class=4" "deep_class=5" "deep_split=1" "deep_split=3" "thr_shade=32" "thr_shade=48" "thr_shade=56" "grid_div=1" "grid_div=2" "prio_depth=0" "prio_depth=8" "deep_class=4 deep_split=3" "deep_class=4 thr_shade=48"; do
    ab $s 1000 "-r 30" $cfg
  done
  ab $s 1000 "-r 30"
done
} > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
