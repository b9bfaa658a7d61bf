#!/bin/bash
# Round 5: the first frame's visiting order built on the device (no host copy); the reference's harness again; first-frame tests + fuzz.
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05t; mkdir -p $OUT
{
echo "== reference harness (futhark/main.c, unmodified) on our library"
for s in rgbbox irreg; do timeout 120 ./oracle/_ref/futhark_main -s $s -n 1000 -m 1000 2>&1 | grep -E "construction|Rendering"; done
echo "== rtbench, library defaults, one frame at a time"
for s in rgbbox irreg; do timeout 120 ./build/rtbench -s $s -n 1000 -m 1000 -r 20 2>&1 | grep -E "BVH|Rendering|HIP-event|Checksum"; done
} > $OUT/harness.log 2>&1
cat $OUT/harness.log
timeout 150 python tools/cold_probe.py 1000 "first_order=0" "first_order=1" 2>&1 | grep -v amdgpu > $OUT/cold_probe_1000.txt; cat $OUT/cold_probe_1000.txt
timeout 600 python -m pytest tests -m gpu -x -q -k "first_frames or camera_path or pixel_tickets or reference_harness or futhark_abi or golden or ragged or never_wait" > $OUT/pytest.log 2>&1; tail -n2 $OUT/pytest.log
FUZZ_FORCE=first_order=1 timeout 80 python tools/fuzz_parity.py 60 95101 > $OUT/fuzz.txt 2>&1; tail -n1 $OUT/fuzz.txt
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_line.json 2> $OUT/bench.err; tail -n1 $OUT/bench.err
echo done
