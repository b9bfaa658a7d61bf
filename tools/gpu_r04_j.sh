#!/bin/bash
# (-> profiles/r04/exp/e12)
# Round 4: look_max swept in the batch regime and on large frames (run-time knob)
cd "$(dirname "$0")/.."
bash tools/gpu_ab.sh r04j/ab <<'AB'
new|rgbbox|1000|-r 0 -B 20|look_max=8
new|rgbbox|1000|-r 0 -B 20|look_max=16
new|rgbbox|1000|-r 0 -B 20|look_max=24
new|rgbbox|1000|-r 0 -B 20|look_max=32
new|rgbbox|1000|-r 0 -B 20|look_max=40
new|irreg|1000|-r 0 -B 20|look_max=8
new|irreg|1000|-r 0 -B 20|look_max=16
new|irreg|1000|-r 0 -B 20|look_max=24
new|irreg|1000|-r 0 -B 20|look_max=32
new|irreg|1000|-r 0 -B 20|look_max=40
new|irreg|4000|-r 5|look_max=16
new|irreg|4000|-r 5|look_max=24
new|irreg|4000|-r 5|look_max=32
new|rgbbox|1000|-r 0 -B 20|look_max=32 thr_shade=32
new|rgbbox|1000|-r 0 -B 20|look_max=32 thr_shade=48
new|irreg|1000|-r 0 -B 20|look_max=32 thr_shade=32
new|irreg|1000|-r 0 -B 20|look_max=32 thr_shade=48
new|rgbbox|1000|-r 0 -B 20|look_max=32
new|irreg|1000|-r 0 -B 20|look_max=32
AB
echo r04j done
