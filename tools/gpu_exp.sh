#!/bin/bash
# One experiment call on the GPU box: a pytest selection, A/B lines for tools/gpu_ab.sh, chain probes and wave traces.
#   gpu_exp.sh <tag> ["pytest -k expression" | -]   with the A/B lines on stdin; CHAIN="opts;opts" TRACE="scene h w opts;..."
cd "$(dirname "$0")/.."
TAG=${1:-exp}; KEXPR=${2:--}
OUT=gpurun_out/$TAG
mkdir -p $OUT
if [ "$KEXPR" != "-" ]; then
  timeout ${PYTEST_TIMEOUT:-600} python -m pytest tests -m gpu -x -q -k "$KEXPR" > $OUT/pytest.log 2>&1
  echo "pytest exit $?" >> $OUT/pytest.log
  tail -5 $OUT/pytest.log
fi
bash tools/gpu_ab.sh $TAG/ab
IFS=';' read -ra CH <<< "$CHAIN"
for c in "${CH[@]}"; do [ -n "$c" ] && timeout 200 python tools/chain_probe.py $c 2>&1 | grep -v amdgpu.ids; done | tee $OUT/chain.txt
IFS=';' read -ra TR <<< "$TRACE"
for c in "${TR[@]}"; do [ -n "$c" ] && timeout 100 python tools/trace_waves.py $c 2>&1 | grep -v amdgpu.ids; done | tee $OUT/trace.txt
