#!/bin/bash
# pipelined bench value for several option sets: gpu_bench_opts.sh "<opts>" "<opts>" ...   (opts: k=v k=v)
cd "$(dirname "$0")/.."
for opts in "$@"; do
  args=""; for kv in $opts; do args="$args --opt $kv"; done
  python bench.py --steps 80 --warmup 10 --no-cpu-baseline --no-serial-extra $args 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('[$opts]', round(d['value']), round(d['ms_per_step'],4))"
done
