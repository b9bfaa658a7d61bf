#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05l; mkdir -p $OUT
for a in "rgbbox 1000 1000" "rgbbox 1000 1000 pixel_order=0" "rgbbox 1000 1000 px_prio=0" "irreg 4000 4000 trace_part=0 trace_nparts=8" "rgbbox 700 700"; do timeout 100 python tools/trace_groups.py $a 2>&1 | grep -v amdgpu; done > $OUT/trace_groups.txt 2>&1
echo done
