#!/bin/bash
# (-> profiles/r04: pmc.json, bench lines, rocprof summary, native bench, traces, rank shares and the GPU suite on the kernel
#  sources with the donation's final condition -- every just-scattered ray of a wave that cannot refill may go; exp/e14)
cd "$(dirname "$0")/.."
SKIP_TESTS=1 SKIP_PEAK=1 bash tools/gpu_round.sh r04m
OUT=$PWD/gpurun_out/r04m
tail -c 200 $OUT/bench_line.json; echo
timeout 160 python -m pytest tests -m gpu -x -q -k "not random_parity_campaign" > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
echo r04m done
