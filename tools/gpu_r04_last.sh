#!/bin/bash
# Round 4, closing call: rank shares + scale prediction with the final look_max policy, then suite, smoke, bench lines
cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out/r04last
mkdir -p $OUT
timeout 200 python tools/rank_share_probe.py 20 1,2,4,8 1 0 2s > $OUT/rank_share_probe.txt 2>&1
timeout 400 python tools/scale_prediction.py 20 > $OUT/scale_prediction.json 2> $OUT/scale_prediction.err; tail -4 $OUT/scale_prediction.err
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log; tail -3 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_line.json 2> $OUT/bench.err; echo "bench exit $?"
echo last done
