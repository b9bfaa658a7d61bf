#!/bin/bash
# Round-end measurement set: GPU tests, bench line, rocprofv3 kernel-trace stats of the bench
# command, PMC passes (incl. FETCH_SIZE / WRITE_SIZE) over the native bench.  usage: gpu_round.sh <tag>
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
TAG=${1:-round}
OUT=$PWD/gpurun_out/$TAG
# (bench.py sets GPU_MAX_HW_QUEUES itself)
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
{
for s in rgbbox irreg; do for v in 1 2 3; do echo "== $s 1000x1000 variant $v"; timeout 120 ./build/rtbench -s $s -n 1000 -m 1000 -r 20 -v $v 2>&1 | grep -E "BVH|HIP-event|Throughput|Algorithmic"; done; done
echo "== irreg 4000x4000 variant 3"; timeout 120 ./build/rtbench -s irreg -n 4000 -m 4000 -r 5 -v 3 2>&1 | grep -E "HIP-event|Throughput|Algorithmic"
echo "== big 2000x2000 variant 3"; timeout 300 ./build/rtbench -s big -n 2000 -m 2000 -r 3 -v 3 2>&1 | grep -E "BVH|HIP-event|Throughput|Algorithmic"
echo "== reference harness (futhark/main.c, unmodified) on our library"
for s in rgbbox irreg; do timeout 120 ./oracle/_ref/futhark_main -s $s -n 1000 -m 1000 2>&1 | grep -E "construction|Rendering"; done
} > $OUT/rtbench.log 2>&1
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -- python $OLDPWD/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-serial-extra > $OUT/rocprof_bench.log 2>&1
cd $OLDPWD
bash tools/gpu_pmc.sh $TAG/pmc "3" "rgbbox irreg" > $OUT/pmc.log 2>&1
PMC_FIRST_ONLY=1 EXTRA_OPTS="-o grid_div=8" bash tools/gpu_pmc.sh $TAG/pmc_gd8 "3" "rgbbox irreg" > $OUT/pmc_gd8.log 2>&1
python tools/make_traffic_json.py $OUT/pmc $OUT/pmc_gd8 > $OUT/traffic.json 2> $OUT/traffic.err
python tools/rocpd_summary.py --last 50 $OUT/prof_bench > $OUT/summary_bench_kernel_trace.txt 2>&1
grep '^{"metric"' $OUT/rocprof_bench.log > $OUT/bench_under_rocprof.json
python tools/rocpd_summary.py $OUT/pmc/*_p[0-9] > $OUT/summary_pmc.txt 2>&1
echo round done
