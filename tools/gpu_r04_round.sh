#!/bin/bash
# Round 4's measurement set: tools/gpu_round.sh (tests, PMC -> pmc.json, bench, native bench, traces, rank shares, rocprof of the
# bench command) + the scale prediction with the direct-store exchange + the fuzz campaign on the final kernel.
cd "$(dirname "$0")/.."
SKIP_PEAK=1 bash tools/gpu_round.sh r04
OUT=$PWD/gpurun_out/r04
timeout 120 python tools/cold_probe.py 700 "handover=0" "handover=1" 2>&1 | grep -v amdgpu > $OUT/cold_probe_700.txt
timeout 120 python tools/cold_probe.py 500 "handover=0" "handover=1" 2>&1 | grep -v amdgpu > $OUT/cold_probe_500.txt
timeout 400 python tools/scale_prediction.py 20 > $OUT/scale_prediction.json 2> $OUT/scale_prediction.err
tail -4 $OUT/scale_prediction.err
timeout 170 python tools/fuzz_parity.py 150 101 > $OUT/fuzz_small_final.txt 2>&1; tail -1 $OUT/fuzz_small_final.txt
timeout 170 python tools/fuzz_parity.py 150 201 700 300000 > $OUT/fuzz_large_final.txt 2>&1; tail -1 $OUT/fuzz_large_final.txt
bash tools/gpu_ab.sh r04/look_max_ab <<'AB'
new|rgbbox|1000|-r 20|look_max=64
new|rgbbox|1000|-r 20|look_max=32
new|irreg|1000|-r 20|look_max=64
new|irreg|1000|-r 20|look_max=32
new|rgbbox|1000|-r 0 -B 20|look_max=64
new|rgbbox|1000|-r 0 -B 20|look_max=32
new|irreg|1000|-r 0 -B 20|look_max=64
new|irreg|1000|-r 0 -B 20|look_max=32
new|irreg|4000|-r 5|look_max=64
new|irreg|4000|-r 5|look_max=32
new|big|2000|-r 4|look_max=64
new|big|2000|-r 4|look_max=32
new|irreg|2000|-r 8|look_max=64
new|irreg|2000|-r 8|look_max=32
base|rgbbox|1000|-r 0 -B 20|
base|irreg|1000|-r 0 -B 20|
AB
timeout 100 python tools/part_probe.py irreg 4000 8 "look_max=64" "look_max=32" 2>&1 | grep -v amdgpu > $OUT/part_probe_look_max.txt
echo r04 round done
