#!/bin/bash
# Round 5: the bulk of the pixel list zipped (tickets alternately from its long and its short end) against sorted.
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05s; mkdir -p $OUT
export AB_TIMEOUT=60
{
for rep in 1 2 3; do for s in rgbbox irreg; do
echo "new|$s|1000|-r 20|px_zip=0"
echo "new|$s|1000|-r 20|px_zip=1"
done; done
for s in rgbbox irreg; do for n in 500 700 1400; do
echo "new|$s|$n|-r 20|px_zip=0"
echo "new|$s|$n|-r 20|px_zip=1"
done; done
} | bash tools/gpu_ab.sh r05s/ab > /dev/null
timeout 100 python tools/part_probe.py irreg 4000 8 "px_zip=0" "px_zip=1" 2>&1 | grep -v amdgpu > $OUT/part_probe.txt
echo done
