#!/bin/bash
# Round 5, last call: the GPU suite on the final code (the instrumented launch walks one-pixel tickets in the pooled loop, so its item
# counters are complete), smoke(), the bench line once more.
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05final; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
tail -n3 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_line.json 2> $OUT/bench.err; tail -n2 $OUT/bench.err
timeout 600 python bench.py > $OUT/bench_line_default_args.json 2> $OUT/bench_default.err
for a in "rgbbox 1000 1000" "irreg 1000 1000" "irreg 4000 4000 trace_part=0 trace_nparts=8"; do echo "=== $a"; timeout 100 python tools/trace_waves.py $a 2>&1 | grep -v amdgpu; done > $OUT/wave_traces.txt 2>&1
echo r05final done
