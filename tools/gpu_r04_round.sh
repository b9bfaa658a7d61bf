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
echo r04 round done
