#!/bin/bash
cd "$(dirname "$0")/.."
for cfg in "8 2 16" "8 1 16" "4 2 16" "4 4 16" "16 1 16" "4 2 32" "8 1 32" "4 2 8"; do set -- $cfg
  python bench.py --steps 60 --warmup 10 --no-cpu-baseline --opt waves_per_wg=$1 --opt wgs_per_cu=$2 --opt thr_shade=$3 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('wpw=$1 wpc=$2 thr=$3', round(d['value']), round(d['ms_per_step'],3), 'serial', round(d['serial']['value']), {k:round(v,3) for k,v in d['serial']['kernel_ms'].items()})"
done
