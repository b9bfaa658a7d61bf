#!/bin/bash
# Round 5: longer randomised campaigns on the final code (all knobs random; then the pixel list forced on with its cuts random).
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05fuzz; mkdir -p $OUT
timeout 330 python tools/fuzz_parity.py 300 60101 > $OUT/fuzz_small_long.txt 2>&1; tail -n1 $OUT/fuzz_small_long.txt
timeout 280 python tools/fuzz_parity.py 250 60201 900 450000 > $OUT/fuzz_large_long.txt 2>&1; tail -n1 $OUT/fuzz_large_long.txt
FUZZ_FORCE=pixel_order=2,adaptive_order=1,handover=1 timeout 230 python tools/fuzz_parity.py 200 60301 400 150000 > $OUT/fuzz_pixel_list_forced.txt 2>&1; tail -n1 $OUT/fuzz_pixel_list_forced.txt
FUZZ_FORCE=pixel_order=2,adaptive_order=2 timeout 130 python tools/fuzz_parity.py 100 60401 200 30000 > $OUT/fuzz_rerecord_every_frame.txt 2>&1; tail -n1 $OUT/fuzz_rerecord_every_frame.txt
echo done
