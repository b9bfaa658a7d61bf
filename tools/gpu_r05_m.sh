#!/bin/bash
# Round 5, call M: rotating issue priorities among the waves that share a SIMD (the arbiter prefers the oldest wave: e10).
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05m; mkdir -p $OUT
export AB_TIMEOUT=60
{
for rep in 1 2; do for s in rgbbox irreg; do
for v in 0 1 2 3; do echo "new|$s|1000|-r 20|prio_rot=$v"; done
done; done
for s in rgbbox irreg; do
for v in 0 1; do
echo "new|$s|700|-r 20|prio_rot=$v"
echo "new|$s|1400|-r 20|prio_rot=$v"
echo "new|$s|2000|-r 20|prio_rot=$v"
echo "new|$s|1000|-r 0 -B 20|prio_rot=$v"
echo "new|$s|1000|-r 0 -B 20|prio_rot=$v"
echo "new|$s|1000|-r 20|pixel_order=0 prio_rot=$v"
done; done
for v in 0 1; do
echo "new|irreg|4000|-r 5|prio_rot=$v"
echo "new|big|2000|-r 3|prio_rot=$v"
done
} | bash tools/gpu_ab.sh r05m/ab > /dev/null
timeout 100 python tools/part_probe.py irreg 4000 8 "" "prio_rot=1" "prio_rot=3" 2>&1 | grep -v amdgpu > $OUT/part_probe.txt
timeout 150 python tools/cold_probe.py 1000 "prio_rot=0" "prio_rot=1" 2>&1 | grep -v amdgpu > $OUT/cold_probe_1000.txt
timeout 100 python tools/trace_groups.py rgbbox 1000 1000 prio_rot=1 2>&1 | grep -v amdgpu > $OUT/trace_groups.txt
echo r05m done
