#!/bin/bash
cd "$(dirname "$0")/.."
for cfg in "8 16 4" "8 16 8" "12 16 4" "12 16 8" "16 16 8" "16 24 8" "16 24 16" "12 16 6"; do set -- $cfg
  GPU_MAX_HW_QUEUES=$2 python bench.py --steps 96 --warmup 24 --frames-in-flight $1 --no-cpu-baseline --no-serial-extra --opt grid_div=$3 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('S=$1 Q=$2 div=$3', round(d['value']), round(d['ms_per_step'],4), round(d['roofline']['frac'],3))"
done
