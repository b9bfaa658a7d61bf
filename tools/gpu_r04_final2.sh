#!/bin/bash
# (-> profiles/r04: the round's measurement set again, on the kernel sources with the DONATE instantiation; exp/e13)
# tools/gpu_round.sh (tests, smoke, PMC -> pmc.json, bench, native bench, traces, rank shares, rocprof of the bench command),
# first / steady frames under the shipped policy and with the tails switched off, scale prediction, fuzz.
cd "$(dirname "$0")/.."
SKIP_PEAK=1 bash tools/gpu_round.sh r04f
OUT=$PWD/gpurun_out/r04f
export GPU_MAX_HW_QUEUES=20
timeout 200 python tools/donate_probe.py "rgbbox:500,irreg:500,rgbbox:700,irreg:700,rgbbox:1000,irreg:1000,irreg:1400,irreg:4000:8,big:2000" \
  "handover=0" "" "handover=2,donate_max=64" 2>&1 | grep -v amdgpu > $OUT/donate_probe_final.txt
timeout 300 python tools/scale_prediction.py 20 > $OUT/scale_prediction.json 2> $OUT/scale_prediction.err
tail -4 $OUT/scale_prediction.err
timeout 100 python tools/fuzz_parity.py 75 301 > $OUT/fuzz_small_final.txt 2>&1; tail -1 $OUT/fuzz_small_final.txt
timeout 100 python tools/fuzz_parity.py 60 401 700 300000 > $OUT/fuzz_large_final.txt 2>&1; tail -1 $OUT/fuzz_large_final.txt
tail -3 $OUT/pytest_gpu.log; tail -2 $OUT/smoke.log; tail -c 300 $OUT/bench_line.json; cat $OUT/donate_probe_final.txt
echo r04f done
