#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05n; mkdir -p $OUT
RT_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 6 --warmup 2 > $OUT/bench_line_two_ranks_sharing_the_gpu.json 2> $OUT/bench2.err; tail -n3 $OUT/bench2.err
timeout 300 python -m pytest tests -m gpu -x -q -k "bench_launches_its_own_ranks or bench_line_two_ranks or wave_trace or multi_rank_render" > $OUT/pytest.log 2>&1; tail -n3 $OUT/pytest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -n2 $OUT/smoke.log
echo done
