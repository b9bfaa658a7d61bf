#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05last; mkdir -p $OUT
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_line.json 2> $OUT/bench.err; tail -n2 $OUT/bench.err
timeout 300 python -m pytest tests -m gpu -x -q -k "bench_line_contract or bench_refuses or bench_configuration" > $OUT/pytest.log 2>&1; tail -n2 $OUT/pytest.log
echo done
