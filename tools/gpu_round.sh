#!/bin/bash
# Round-end measurement set: GPU tests, bench line (the driver's command), rocprofv3 kernel-trace stats of
# that command, PMC passes over the native bench -> pmc.json, native bench log.  usage: gpu_round.sh <tag>
# Everything lands in gpurun_out/<tag>/; copy what is to be judged into profiles/<round>/ (pmc.json and
# issue_peak.json go to profiles/ itself: bench.py reads them).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
TAG=${1:-round}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
# env: SKIP_TESTS=1 (none) | quick (a subset); SKIP_PEAK=1 keeps profiles/issue_peak.json (the microbenchmark does not depend
# on the kernel sources); PMC_GDS="0 4" adds the quarter-size launches of --protocol lanes to the PMC passes
if [ -z "$SKIP_TESTS" ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
  echo "pytest exit $?" >> $OUT/pytest_gpu.log
elif [ "$SKIP_TESTS" = quick ]; then
  timeout 400 python -m pytest tests -m gpu -x -q -k "golden_500 or tile_queue_layouts or bench_line_contract or irreg_4000 or big_2000 or reference_harness" > $OUT/pytest_gpu.log 2>&1
  echo "pytest (subset) exit $?" >> $OUT/pytest_gpu.log
fi
[ "$SKIP_TESTS" != 1 ] && timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
if [ -z "$SKIP_PEAK" ]; then
  bash tools/gpu_issue_peak.sh gpurun_out/$TAG/issue_peak.txt > /dev/null 2>&1
  python tools/make_issue_peak_json.py $OUT/issue_peak.txt > $OUT/issue_peak.json
else
  cp profiles/issue_peak.json $OUT/issue_peak.json
fi
bash tools/gpu_pmc.sh $TAG/pmc "${PMC_GDS:-0}" "rgbbox irreg" > $OUT/pmc.log 2>&1
python tools/make_pmc_json.py $OUT/pmc $OUT/issue_peak.json > $OUT/pmc.json 2> $OUT/pmc_json.err
cp $OUT/pmc.json $OUT/issue_peak.json profiles/        # on the GPU box only: the bench runs below read them
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_line.json 2> $OUT/bench.err
{
export GPU_MAX_HW_QUEUES=20
for s in rgbbox irreg; do for v in 1 2 3; do echo "== $s 1000x1000 variant $v"; timeout 120 ./build/rtbench -s $s -n 1000 -m 1000 -r 20 -v $v $([ $v = 3 ] && echo "-L 24 -o grid_div=4 -o deep_class=0") 2>&1 | grep -E "BVH|HIP-event|Throughput|Algorithmic|Overlapped|Batch"; done; done
for s in rgbbox irreg; do echo "== $s 1000x1000 variant 3, batch entry (rt_render_batch), library defaults"; timeout 120 ./build/rtbench -s $s -n 1000 -m 1000 -r 0 -B 20 2>&1 | grep -E "Batch"; timeout 120 ./build/rtbench -s $s -n 1000 -m 1000 -r 0 -B 200 2>&1 | grep -E "Batch"; done
for s in rgbbox irreg; do echo "== $s 1000x1000 variant 3, library defaults, one frame at a time"; timeout 120 ./build/rtbench -s $s -n 1000 -m 1000 -r 20 2>&1 | grep -E "BVH|Rendering|HIP-event|Throughput|Checksum"; done
echo "== irreg 4000x4000 variant 3"; timeout 120 ./build/rtbench -s irreg -n 4000 -m 4000 -r 5 -v 3 2>&1 | grep -E "HIP-event|Throughput|Algorithmic|Checksum"
echo "== big 2000x2000 variant 3"; timeout 300 ./build/rtbench -s big -n 2000 -m 2000 -r 3 -v 3 2>&1 | grep -E "BVH|HIP-event|Throughput|Algorithmic|Checksum"
echo "== reference harness (futhark/main.c, unmodified) on our library"
for s in rgbbox irreg; do timeout 120 ./oracle/_ref/futhark_main -s $s -n 1000 -m 1000 2>&1 | grep -E "construction|Rendering"; done
} > $OUT/rtbench.log 2>&1
# per-wave timelines of one frame (instrumented launch) and one rank's share of the bench at world size 8
for a in "rgbbox 1000 1000" "irreg 1000 1000" "irreg 4000 4000" "big 2000 2000"; do timeout 100 python tools/trace_waves.py $a; done > $OUT/wave_traces.txt 2>&1
timeout 200 python tools/rank_share_probe.py 20 1,2,4,8 1 0 2s > $OUT/rank_share_probe.txt 2>&1
timeout 200 python tools/rank_share_probe.py 20 1,8 1 2 2s >> $OUT/rank_share_probe.txt 2>&1
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -- python $OLDPWD/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/rocprof_bench.log 2>&1   # (WITH the serial region: the ORD / DONATE instantiations, the sorts and first_order show up with their durations)
cd $OLDPWD
python tools/rocpd_summary.py --last 1 $OUT/prof_bench > $OUT/bench_kernel_trace_summary.txt 2>&1
find $OUT/prof_bench -name "*kernel_stats.csv" -exec cp {} $OUT/bench_kernel_stats.csv \;
grep '^{"metric"' $OUT/rocprof_bench.log > $OUT/bench_line_under_rocprof.json
rm -rf $OUT/prof_bench
echo round done
