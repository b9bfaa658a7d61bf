#!/bin/bash
# A/B through the native bench, one line per run (times + the frame's checksum), optionally against another build of the
# library.  Reads lines  lib|scene|size|rtbench mode|opt=value ...  from stdin, e.g.
#   new|irreg|4000|-r 8|xcd_queues=0
#   base|rgbbox|1000|-r 0 -B 20|
# a name other than `new` runs with build/lib_<name>/libray_mi355x.so in front of the library path, e.g. `base` (build it from another commit:
# git archive <commit> | tar -x -C /tmp/b && make -C /tmp/b raytracers_amd/libray_mi355x.so && cp … build/lib_base/).
# Round 2's experiment logs in profiles/r02/ (queue_ab_*, box_diet_ab_e2, peek_ab_e10, knob_sweep_e11, ab_vs_504921f_e9) are
# loops over this line format.      usage: gpu_ab.sh <outname> < configs
cd "$(dirname "$0")/.."
OUT=gpurun_out/${1:-ab}.txt
mkdir -p "$(dirname "$OUT")"
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-20}
while IFS='|' read -r lib s n mode rest; do
  [ -z "$lib" ] && continue
  o=""; for kv in $rest; do o="$o -o $kv"; done
  res=$( ( [ "$lib" != new ] && export LD_LIBRARY_PATH=$PWD/build/lib_$lib:$LD_LIBRARY_PATH; timeout ${AB_TIMEOUT:-120} ./build/rtbench -s $s -n $n -m $n $mode $o 2>&1 ) | grep -E "HIP-event|Checksum|Batch|Overlapped|failed|unknown|no HIP" | tr '\n' ' ')
  echo "$lib $s $n $mode [$rest] : $res"
done | tee "$OUT"
