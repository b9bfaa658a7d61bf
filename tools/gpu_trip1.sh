#!/bin/bash
# GPU trip 1: smoke, parity tests, knob sweep, bench line, rocprof summary.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/trip1
mkdir -p $OUT
{
echo "== rocminfo"; rocminfo | grep -E "Marketing Name|Compute Unit|gfx" | head -8
echo "== nproc $(nproc)"
echo "== smoke"
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5
echo "== rtbench quick"
for s in rgbbox irreg; do for v in 1 2; do timeout 120 ./build/rtbench -s $s -n 1000 -m 1000 -r 20 -v $v 2>&1 | grep -E "Rendering|HIP-event|Throughput|Algorithmic|BVH"; done; done
} > $OUT/01_smoke.log 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/02_pytest.log 2>&1
echo "pytest exit $?" >> $OUT/02_pytest.log
{
echo "== sweep"
for s in rgbbox irreg; do
 for wpw in 4 8 16; do for wpc in 1 2 4 8; do
  if [ $((wpw*wpc)) -le 32 ]; then
  for thr in 8 16 24 32 48; do
   r=$(timeout 60 ./build/rtbench -s $s -n 1000 -m 1000 -r 10 -v 2 -o waves_per_wg=$wpw -o wgs_per_cu=$wpc -o thr_shade=$thr -o thr_leaf=$thr 2>&1 | grep "HIP-event")
   echo "$s wpw=$wpw wpc=$wpc thr=$thr : $r"
  done; fi
 done; done
done
} > $OUT/03_sweep.log 2>&1
timeout 600 python bench.py --steps 50 --warmup 10 > $OUT/04_bench.json 2> $OUT/04_bench.err
timeout 600 python bench.py --steps 50 --warmup 10 --variant 1 --no-cpu-baseline > $OUT/04_bench_pixel.json 2>> $OUT/04_bench.err
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof_bench -- python $OLDPWD/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OLDPWD/$OUT/05_rocprof_bench.log 2>&1
cd $OLDPWD
rocprofv3 -L > $OUT/counters.txt 2>&1
find $OUT/prof_bench -name "*stats*" | head -20 > $OUT/prof_files.txt
echo trip1 done
