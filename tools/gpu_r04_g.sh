#!/bin/bash
# (-> profiles/r04/exp/e8; `cold_warm` / `cold_rays` were experiment knobs: the product hands over 1 ray in a first frame, 3 in an ordered one)
# Round 4: the in-loop hand-over to the solo loop for ORDERED single frames (cold_warm=1), 1 .. 4 rays
cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out/r04g
mkdir -p $OUT
timeout 300 python -m pytest tests -m gpu -x -q -k "first_frames_of_new_views or golden_500 or solo_pixels" > $OUT/pytest.log 2>&1; tail -2 $OUT/pytest.log
bash tools/gpu_ab.sh r04g/ab <<'AB'
new|rgbbox|1000|-r 20|
new|rgbbox|1000|-r 20|cold_warm=1 cold_rays=1
new|rgbbox|1000|-r 20|cold_warm=1 cold_rays=2
new|rgbbox|1000|-r 20|cold_warm=1 cold_rays=3
new|rgbbox|1000|-r 20|cold_warm=1 cold_rays=4
new|rgbbox|1000|-r 20|cold_warm=1 cold_rays=8
new|irreg|1000|-r 20|
new|irreg|1000|-r 20|cold_warm=1 cold_rays=1
new|irreg|1000|-r 20|cold_warm=1 cold_rays=3
new|irreg|1000|-r 20|cold_warm=1 cold_rays=8
new|rgbbox|500|-r 20|
new|rgbbox|500|-r 20|cold_warm=1 cold_rays=3
new|irreg|500|-r 20|
new|irreg|500|-r 20|cold_warm=1 cold_rays=3
new|irreg|4000|-r 5|
new|irreg|4000|-r 5|cold_warm=1 cold_rays=3
new|big|2000|-r 4|
new|big|2000|-r 4|cold_warm=1 cold_rays=3
new|rgbbox|1000|-r 20|
AB
timeout 200 python tools/part_probe.py irreg 4000 8 "" "cold_warm=1,cold_rays=1" "cold_warm=1,cold_rays=3" "cold_warm=1,cold_rays=8" 2>&1 | grep -v amdgpu | tee $OUT/part_probe.txt
timeout 200 python tools/cold_probe.py 500 "cold_first=1" "cold_first=1,cold_rays=3" "cold_first=1,cold_rays=8" 2>&1 | grep -v amdgpu | tee $OUT/cold_probe.txt
timeout 200 python tools/cold_probe.py 1000 "cold_first=0" "cold_first=1,cold_rays=3,cold_warm=1" 2>&1 | grep -v amdgpu | tee -a $OUT/cold_probe.txt
echo r04g done
