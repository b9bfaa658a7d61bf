#!/bin/bash
# One script for every call on the GPU box (through gpurun): tools/gpu.sh <tag> <step> [<step> ...]
# Everything lands under gpurun_out/<tag>/ (copy what is to be judged into profiles/).  Steps:
#   suite            the whole GPU test suite (pytest -m gpu) -> pytest_gpu.log
#   tests:<expr>     a pytest -k selection -> pytest_<n>.log
#   smoke            __graft_entry__.smoke() -> smoke.log
#   ab               A/B lines for tools/gpu_ab.sh on stdin (lib|scene|size|rtbench mode|opt=value ...) -> ab.txt
#   fuzz:<s>[:<seed>[:<side>[:<spheres>]]]   tools/fuzz_parity.py for <s> seconds -> fuzz_<seed>.txt   (FUZZ_FORCE=k=v,.. pins knobs)
#   tsan             build/tsan/ctx_threads (the library's host code under -fsanitize=thread) -> tsan.log
#   bench[:args]     python bench.py <args> (default: the driver's --steps 20 --warmup 5) -> bench_line.json, bench.err
#   prof_bench       rocprofv3 --kernel-trace --stats of the bench command -> bench_kernel_stats.csv, bench_kernel_trace_summary.txt
#   harness          the reference's unmodified main.c on the library (oracle/_ref/futhark_main) + rtbench's serial protocol -> harness.log
#   cold[:size]      tools/cold_probe.py (first frames of new views, a camera path view by view) -> cold_probe_<size>.txt
#   pmc              tools/gpu_pmc.sh passes -> pmc_summary.csv, pmc.json
#   parts[:W]        tools/part_probe.py irreg 4000 W -> part_probe.txt;   scale   tools/scale_prediction.py -> scale_prediction.json
#   round            the round's measurement set: tools/gpu_round.sh <tag>
cd "$(dirname "$0")/.."
TAG=${1:?tag}; shift
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-20}
nt=0
for step in "$@"; do
  name=${step%%:*}; arg=""; [ "$step" != "$name" ] && arg=${step#*:}
  echo "== $step"
  case $name in
    suite) timeout ${PYTEST_TIMEOUT:-1500} python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log; tail -4 $OUT/pytest_gpu.log ;;
    tests) nt=$((nt + 1)); timeout ${PYTEST_TIMEOUT:-900} python -m pytest tests -m gpu -x -q -k "$arg" > $OUT/pytest_$nt.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_$nt.log; tail -4 $OUT/pytest_$nt.log ;;
    smoke) timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?" >> $OUT/smoke.log; tail -3 $OUT/smoke.log ;;
    ab) bash tools/gpu_ab.sh $TAG/ab ;;
    fuzz) IFS=: read -r secs seed side spheres <<< "$arg"
          timeout $((${secs:-60} + 120)) python tools/fuzz_parity.py ${secs:-60} ${seed:-1} ${side:-160} ${spheres:-20000} > $OUT/fuzz_${seed:-1}.txt 2>&1; tail -n2 $OUT/fuzz_${seed:-1}.txt ;;
    tsan) { for exe in build/tsan/ctx_threads build/tsan_nolock/ctx_threads; do for m in "rt irreg" "rt rgbbox" "futhark irreg"; do
              echo "== $exe $m 256 200 2 (host code under -fsanitize=thread; suppressions: tools/tsan.supp)"
              TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0 suppressions=$PWD/tools/tsan.supp" timeout 600 ./$exe $m 256 200 2; echo "exit $?"; done; done; } > $OUT/tsan.log 2>&1
          grep -E "^== |exit|frames on one|SUMMARY" $OUT/tsan.log | sort | uniq -c | sort -rn | head -30 ;;
    bench) timeout 900 python bench.py ${arg:---steps 20 --warmup 5} > $OUT/bench_line.json 2> $OUT/bench.err; echo "bench exit $?"; head -c 600 $OUT/bench_line.json; echo ;;
    prof_bench) ( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -- python $OLDPWD/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline ${arg} > $OUT/rocprof_bench.log 2>&1 )
                python tools/rocpd_summary.py --last 1 $OUT/prof_bench > $OUT/bench_kernel_trace_summary.txt 2>&1
                find $OUT/prof_bench -name "*kernel_stats.csv" -exec cp {} $OUT/bench_kernel_stats.csv \;
                grep '^{"metric"' $OUT/rocprof_bench.log > $OUT/bench_line_under_rocprof.json; rm -rf $OUT/prof_bench; head -n 12 $OUT/bench_kernel_stats.csv ;;
    harness) { for s in rgbbox irreg; do echo "== futhark/main.c (unmodified) -s $s -n 1000 -m 1000"; ./oracle/_ref/futhark_main -s $s -n 1000 -m 1000 -f /dev/null; done
               for s in rgbbox irreg; do echo "== rtbench $s 1000 -r 20"; ./build/rtbench -s $s -n 1000 -m 1000 -r 20; done; } > $OUT/harness.log 2>&1; grep -E "==|Rendering|construction" $OUT/harness.log ;;
    cold) timeout 300 python tools/cold_probe.py ${arg:-1000} "" 2>&1 | grep -v amdgpu > $OUT/cold_probe_${arg:-1000}.txt; tail -n 12 $OUT/cold_probe_${arg:-1000}.txt ;;
    pmc) bash tools/gpu_pmc.sh $TAG ;;
    parts) timeout 200 python tools/part_probe.py irreg 4000 ${arg:-8} "" 2>&1 | grep -v amdgpu > $OUT/part_probe.txt; cat $OUT/part_probe.txt ;;
    scale) timeout 500 python tools/scale_prediction.py 20 > $OUT/scale_prediction.json 2> $OUT/scale_prediction.err; tail -n4 $OUT/scale_prediction.err ;;
    round) bash tools/gpu_round.sh $TAG ;;
    *) echo "unknown step $step"; exit 2 ;;
  esac
done
echo "$TAG done"
