#!/bin/bash
# Round 5: rt_context_last_launch in the suite and in the bench line.
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05o; mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -x -q -k "pixel_tickets or bench_line_contract or futhark_abi or symbols" > $OUT/pytest.log 2>&1; tail -n3 $OUT/pytest.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_line.json 2> $OUT/bench.err; tail -n2 $OUT/bench.err
echo done
