#!/bin/bash
# quick A/B of rtbench option sets (one frame at a time and 24 frames overlapped on 20 hardware queues):
#   gpu_quick.sh <outname> "<opts A>" "<opts B>" ...      (LIB=<dir> picks another build of the library)
cd "$(dirname "$0")/.."
OUT=gpurun_out/$1; shift
mkdir -p $(dirname $OUT)
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-20}
[ -n "$LIB" ] && export LD_LIBRARY_PATH=$PWD/$LIB:$LD_LIBRARY_PATH
{
for opts in "$@"; do
  for s in ${SCENES:-rgbbox irreg}; do
    r=$(timeout 120 ./build/rtbench -s $s -n ${SIZE:-1000} -m ${SIZE:-1000} -r 20 -L ${LANES:-24} $opts 2>&1 | grep -E "HIP-event|Overlapped|failed" | tr '\n' ' ')
    echo "$s [$LIB $opts] : $r"
  done
done
} >> $OUT 2>&1
cat $OUT
