#!/bin/bash
# Round 4, last call: the whole GPU suite, smoke() and the bench line on the final tree
cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out/r04final
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log; tail -3 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_line.json 2> $OUT/bench.err; echo "bench exit $?"
timeout 600 python bench.py > $OUT/bench_line_default_args.json 2> $OUT/bench_default.err; echo "bench (default args) exit $?"
echo final done
