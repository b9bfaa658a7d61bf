#!/bin/bash
# the idle gap ahead of the timed bracket: bench value and per-launch kernel ms against the gap's length
cd "$(dirname "$0")/../../.."
mkdir -p gpurun_out/gap
for g in 0 1 5 20 100 1000 0; do
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-serial-extra --idle-before-ms $g 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][0])
print('idle $g ms:', round(d['value']), d['ms_per_step'], d['roofline']['per_launch']['kernel_ms'])"
done | tee gpurun_out/gap/idle_gap.txt
