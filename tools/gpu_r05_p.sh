#!/bin/bash
# Round 5: the visiting order of a view's FIRST frame -- tile rows top to bottom against bit-reversed rows.
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05p; mkdir -p $OUT
for n in 1000 500 2000; do timeout 150 python tools/cold_probe.py $n "first_order=0" "first_order=1" 2>&1 | grep -v amdgpu; done > $OUT/cold_probe.txt
timeout 100 python tools/donate_probe.py 2>&1 | grep -v amdgpu | head -5 > /dev/null
FUZZ_FORCE=first_order=1 timeout 100 python tools/fuzz_parity.py 70 70101 > $OUT/fuzz_first_order.txt 2>&1; tail -n1 $OUT/fuzz_first_order.txt
echo done
