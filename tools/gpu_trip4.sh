#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
TRIP=${1:-trip4}
OUT=gpurun_out/$TRIP
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/02_pytest.log 2>&1
echo "pytest exit $?" >> $OUT/02_pytest.log
{
for s in rgbbox irreg; do for v in 3; do echo "== $s v$v"; timeout 120 ./build/rtbench -s $s -n 1000 -m 1000 -r 20 -v $v 2>&1 | grep -E "HIP-event|Throughput|Algorithmic"; echo "== $s v$v adaptive_order=0";  timeout 120 ./build/rtbench -s $s -n 1000 -m 1000 -r 20 -v $v -o adaptive_order=0 2>&1 | grep -E "HIP-event|Throughput|Algorithmic"; done; done
echo "== sweep v3"
for s in rgbbox irreg; do
 for cfg in "4 1" "4 2" "4 4" "8 1" "8 2" "16 1"; do set -- $cfg
  for thr in 4 16 32 48; do
   r=$(timeout 60 ./build/rtbench -s $s -n 1000 -m 1000 -r 10 -v 3 -o waves_per_wg=$1 -o wgs_per_cu=$2 -o thr_shade=$thr 2>&1 | grep "HIP-event")
   echo "$s wpw=$1 wpc=$2 thr=$thr : $r"
  done
 done
done
echo "== irreg 4000 / big v3"
timeout 120 ./build/rtbench -s irreg -n 4000 -m 4000 -r 5 -v 3 2>&1 | grep -E "HIP-event|Throughput"
timeout 300 ./build/rtbench -s big -n 2000 -m 2000 -r 3 -v 3 2>&1 | grep -E "BVH|HIP-event|Throughput"
} > $OUT/03_sweep.log 2>&1
timeout 600 python bench.py --steps 50 --warmup 10 > $OUT/04_bench.json 2> $OUT/04_bench.err
bash tools/gpu_pmc.sh $TRIP/pmc "3" "rgbbox irreg" > $OUT/06_pmc.log 2>&1
echo trip done
