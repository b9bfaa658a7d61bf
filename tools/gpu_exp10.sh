#!/bin/bash
cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out/${1:-exp10}
mkdir -p $OUT
ab() {
  local lib=$1 s=$2 n=$3 mode=$4; shift 4
  local o=""; for kv in "$@"; do o="$o -o $kv"; done
  local res=$( ( [ "$lib" = base ] && export LD_LIBRARY_PATH=$PWD/build/lib_base:$LD_LIBRARY_PATH; timeout 120 ./build/rtbench -s $s -n $n -m $n $mode $o 2>&1 ) | grep -E "HIP-event|Checksum|Batch|failed|unknown" | tr '\n' ' ')
  echo "$lib $s $n $mode [$*] : $res"
}
{
for rep in 1 2 3; do for lib in base new; do for s in rgbbox irreg; do
  ab $lib $s 1000 "-r 30"
done; done; done
for lib in base new; do
  ab $lib rgbbox 200 "-r 30"; ab $lib irreg 500 "-r 30"; ab $lib irreg 4000 "-r 8"; ab $lib big 2000 "-r 5"; ab $lib rgbbox 1000 "-r 0 -B 20"; ab $lib irreg 1000 "-r 0 -B 20"
done
} > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
