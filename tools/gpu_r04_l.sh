#!/bin/bash
# (-> profiles/r04/exp/e14)
# Round 4: the donation without the "both lists empty" condition -- every just-scattered ray of a wave that cannot refill may go
cd "$(dirname "$0")/.."
OUT=gpurun_out/r04l; mkdir -p $OUT
export GPU_MAX_HW_QUEUES=20
timeout 200 python tools/donate_probe.py "rgbbox:500,irreg:500,rgbbox:1000,irreg:1000,irreg:4000:8,big:2000" \
  "handover=0" "" "handover=2,donate_max=64" "handover=2,donate_max=16" 2>&1 | grep -v amdgpu > $OUT/donate_probe.txt
echo "probe exit $?" >> $OUT/donate_probe.txt
FUZZ_FORCE="handover=2,donate_max=64,waves_per_wg=16" timeout 100 python tools/fuzz_parity.py 30 990000 > $OUT/fuzz_donate.txt 2>&1
echo "fuzz exit $?" >> $OUT/fuzz_donate.txt
cat $OUT/donate_probe.txt; tail -2 $OUT/fuzz_donate.txt
