#!/bin/bash
# Round 5, call I: u / v by division in the list's refill; the issue priority of waves that hold a ticket; traces with the queue's end stamped.
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05i; mkdir -p $OUT
export AB_TIMEOUT=60
{
for rep in 1 2; do
for s in rgbbox irreg; do
echo "pxg|$s|1000|-r 20|"
echo "new|$s|1000|-r 20|"
echo "new|$s|1000|-r 20|px_prio=0"
echo "new|$s|1000|-r 20|px_prio=1"
echo "new|$s|1000|-r 20|px_prio=2"
done; done
for s in rgbbox irreg; do for n in 500 700 1400; do
echo "pxg|$s|$n|-r 20|"
echo "new|$s|$n|-r 20|"
echo "new|$s|$n|-r 20|px_prio=0"
done; done
} | bash tools/gpu_ab.sh r05i/ab > /dev/null
timeout 100 python tools/part_probe.py irreg 4000 8 "" "px_prio=0" "px_prio=1" 2>&1 | grep -v amdgpu > $OUT/part_probe.txt
for a in "rgbbox 1000 1000" "rgbbox 1000 1000 px_prio=0" "irreg 1000 1000"; do echo "=== $a"; timeout 100 python tools/trace_waves.py $a 2>&1 | grep -v amdgpu; done > $OUT/wave_traces.txt 2>&1
echo r05i done
