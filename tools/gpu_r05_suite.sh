#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05suite; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
tail -n3 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -n2 $OUT/smoke.log
echo done
