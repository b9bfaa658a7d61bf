#!/bin/bash
# Round 5's measurement set: tools/gpu_round.sh (tests, smoke, PMC -> pmc.json, bench = the driver's command, native bench, traces, rank
# shares, rocprof of the bench command) + first frames, the scale prediction, the fuzz campaigns on the final kernels.
cd "$(dirname "$0")/.."
SKIP_PEAK=1 bash tools/gpu_round.sh r05
OUT=$PWD/gpurun_out/r05
timeout 150 python tools/cold_probe.py 1000 "pixel_order=0" "pixel_order=1" 2>&1 | grep -v amdgpu > $OUT/cold_probe_1000.txt
timeout 150 python tools/cold_probe.py 500 "pixel_order=0" "pixel_order=1" 2>&1 | grep -v amdgpu > $OUT/cold_probe_500.txt
timeout 400 python tools/scale_prediction.py 20 > $OUT/scale_prediction.json 2> $OUT/scale_prediction.err
tail -n4 $OUT/scale_prediction.err
timeout 100 python tools/fuzz_parity.py 80 90101 > $OUT/fuzz_small_final.txt 2>&1; tail -n1 $OUT/fuzz_small_final.txt
timeout 100 python tools/fuzz_parity.py 80 90201 700 300000 > $OUT/fuzz_large_final.txt 2>&1; tail -n1 $OUT/fuzz_large_final.txt
export AB_TIMEOUT=60
{
for s in rgbbox irreg; do for n in 200 300 500 700 1000 1400 2000; do
echo "new|$s|$n|-r 20|pixel_order=0"
echo "new|$s|$n|-r 20|"
done; done
echo "new|irreg|4000|-r 5|pixel_order=0"
echo "new|irreg|4000|-r 5|"
echo "new|big|2000|-r 3|pixel_order=0"
echo "new|big|2000|-r 3|"
} | bash tools/gpu_ab.sh r05/pixel_tickets_vs_tile_tickets_ab > /dev/null
for W in 8; do timeout 100 python tools/part_probe.py irreg 4000 $W "pixel_order=0" "" 2>&1 | grep -v amdgpu; done > $OUT/part_probe.txt
echo r05 round done
