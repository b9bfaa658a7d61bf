#!/bin/bash
# Round 5, call H: wave traces of frames rendered through the pixel list (the instrumented ORD launch), and the loop's knobs under it.
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05h; mkdir -p $OUT
export AB_TIMEOUT=60
for a in "rgbbox 1000 1000" "rgbbox 1000 1000 pixel_order=0" "irreg 1000 1000" "irreg 1000 1000 pixel_order=0" "rgbbox 500 500" "irreg 4000 4000 trace_part=0 trace_nparts=8" "irreg 4000 4000 trace_part=0 trace_nparts=8 pixel_order=0"; do echo "=== $a"; timeout 100 python tools/trace_waves.py $a 2>&1 | grep -v amdgpu; done > $OUT/wave_traces.txt 2>&1
{
for s in rgbbox irreg; do
echo "new|$s|1000|-r 20|"
echo "new|$s|1000|-r 20|thr_shade=24"
echo "new|$s|1000|-r 20|thr_shade=56"
echo "new|$s|1000|-r 20|look_max=32"
echo "new|$s|1000|-r 20|look_max=16"
echo "new|$s|1000|-r 20|box2=0"
echo "new|$s|1000|-r 20|px_hold=7"
echo "new|$s|1000|-r 20|px_hold=3"
echo "new|$s|1000|-r 20|prio_depth=0"
echo "new|$s|1000|-r 20|xcd_queues=0"
echo "new|$s|1000|-r 20|static_first=0"
echo "new|$s|1000|-r 20|px_g32=70"
echo "new|$s|1000|-r 20|px_g32=160"
echo "new|$s|1000|-r 20|px_g16=45"
echo "new|$s|1000|-r 20|px_g16=110"
echo "new|$s|1000|-r 20|px_g8=30"
echo "new|$s|1000|-r 20|px_solo_div=64"
done
} | bash tools/gpu_ab.sh r05h/ab > /dev/null
echo r05h done
