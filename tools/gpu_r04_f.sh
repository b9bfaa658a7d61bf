#!/bin/bash
# Round 4: first frames at 500 / 700 / 1400: every workgroup vs the COLD instantiation (-> profiles/r04/exp/e7; option names of that day)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r04f
mkdir -p $OUT
for n in 500 700 1400; do
timeout 200 python tools/cold_probe.py $n "cold_first=0" "cold_first=0,grid_div=1" "cold_first=1,cold_hold_depth=64" "cold_first=1,cold_hold_depth=24" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/cold_probe.txt
done
echo r04f done
