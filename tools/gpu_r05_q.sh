#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05q; mkdir -p $OUT
for n in 1000 500 2000; do timeout 200 python tools/cold_probe.py $n "first_order=0" "first_order=1" "first_order=2" "first_order=3" 2>&1 | grep -v amdgpu; done > $OUT/cold_probe.txt
timeout 100 python tools/part_probe.py irreg 4000 8 "" 2>&1 | grep -v amdgpu > /dev/null
echo done
