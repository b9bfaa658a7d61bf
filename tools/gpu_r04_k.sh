#!/bin/bash
# (-> profiles/r04/exp/e13)
# Round 4: rays of waves that cannot refill, given to waiting sibling waves through LDS -- handover=2
cd "$(dirname "$0")/.."
OUT=gpurun_out/r04k; mkdir -p $OUT
export GPU_MAX_HW_QUEUES=20
timeout 240 python tools/donate_probe.py "rgbbox:500,irreg:500,rgbbox:1000,irreg:1000,irreg:4000:8,big:2000" \
  "handover=1" "handover=2,donate_max=2" "handover=2,donate_max=8" "handover=2,donate_max=64" "handover=1" > $OUT/donate_probe.txt 2>&1
echo "probe exit $?" >> $OUT/donate_probe.txt
FUZZ_FORCE="handover=2,donate_max=8,waves_per_wg=16" timeout 120 python tools/fuzz_parity.py 40 880000 > $OUT/fuzz_donate.txt 2>&1
echo "fuzz exit $?" >> $OUT/fuzz_donate.txt
tail -30 $OUT/donate_probe.txt; tail -3 $OUT/fuzz_donate.txt
