#!/bin/bash
# quick A/B of rtbench option sets: gpu_quick.sh <outname> "<opts A>" "<opts B>" ...
cd "$(dirname "$0")/.."
OUT=gpurun_out/$1; shift
mkdir -p $(dirname $OUT)
{
for opts in "$@"; do
  for s in rgbbox irreg; do
    r=$(timeout 120 ./build/rtbench -s $s -n 1000 -m 1000 -r 30 $opts 2>&1 | grep -E "HIP-event")
    echo "$s [$opts] : $r"
  done
done
} > $OUT 2>&1
cat $OUT
