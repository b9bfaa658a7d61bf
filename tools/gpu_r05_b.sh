#!/bin/bash
# Round 5, call B: pixel tickets with tile-major lists, per-ticket entry prefetch and the device-side class model.
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05b; mkdir -p $OUT
export AB_TIMEOUT=60
{
for s in rgbbox irreg; do for n in 1000 500 300 200 2000; do
echo "new|$s|$n|-r 20|pixel_order=0"
echo "new|$s|$n|-r 20|pixel_order=2"
done; done
# the model's constants
for s in rgbbox irreg; do
for v in 200 400 500; do echo "new|$s|1000|-r 20|pixel_order=2 px_ray_ns=$v"; done
echo "new|$s|1000|-r 20|pixel_order=2 px_hold=7"
echo "new|$s|1000|-r 20|pixel_order=2 px_solo_div=16"
echo "new|$s|1000|-r 20|pixel_order=2 px_solo_div=1"
done
echo "new|rgbbox|1000|-r 20|pixel_order=2 px_g64=120"
echo "new|rgbbox|1000|-r 20|pixel_order=2 px_g64=200"
echo "new|rgbbox|1000|-r 20|pixel_order=2 px_g64=240"
echo "new|rgbbox|1000|-r 20|pixel_order=2 px_g16=50"
echo "new|rgbbox|1000|-r 20|pixel_order=2 px_g16=90"
echo "new|rgbbox|1000|-r 20|pixel_order=2 px_g1=15"
echo "new|rgbbox|1000|-r 20|pixel_order=2 px_g1=40"
echo "new|irreg|1000|-r 20|pixel_order=2 px_g64=250"
echo "new|irreg|1000|-r 20|pixel_order=2 px_g64=420"
echo "new|irreg|1000|-r 20|pixel_order=2 px_g32=150"
echo "new|irreg|1000|-r 20|pixel_order=2 px_g32=240"
echo "new|irreg|1000|-r 20|pixel_order=2 px_g16=80"
echo "new|irreg|1000|-r 20|pixel_order=2 px_g16=130"
echo "new|irreg|1000|-r 20|pixel_order=2 px_g1=35"
echo "new|irreg|1000|-r 20|pixel_order=2 px_g1=60"
# by hand, the best of call A and the pure sort
echo "new|rgbbox|1000|-r 20|pixel_order=2 px_solo=255 px_w8=255 px_w16=22 px_w32=22"
echo "new|rgbbox|1000|-r 20|pixel_order=2 px_solo=255 px_w8=255 px_w16=255 px_w32=255"
echo "new|rgbbox|1000|-r 20|pixel_order=2 px_solo=255 px_w8=255 px_w16=28 px_w32=18"
echo "new|irreg|1000|-r 20|pixel_order=2 px_solo=12 px_w8=12 px_w16=12 px_w32=8"
echo "new|irreg|1000|-r 20|pixel_order=2 px_solo=24 px_w8=16 px_w16=10 px_w32=7"
echo "new|irreg|1000|-r 20|pixel_order=2 px_solo=255 px_w8=255 px_w16=255 px_w32=255"
echo "new|irreg|4000|-r 5|pixel_order=0"
echo "new|irreg|4000|-r 5|pixel_order=2"
echo "new|big|2000|-r 3|pixel_order=0"
echo "new|big|2000|-r 3|pixel_order=2"
} | bash tools/gpu_ab.sh r05b/ab > /dev/null
timeout 100 python tools/part_probe.py irreg 4000 8 "pixel_order=0" "pixel_order=2" "pixel_order=2,px_ray_ns=400" 2>&1 | grep -v amdgpu > $OUT/part_probe.txt
timeout 100 python tools/part_probe.py irreg 4000 4 "pixel_order=0" "pixel_order=2" 2>&1 | grep -v amdgpu >> $OUT/part_probe.txt
timeout 100 python tools/part_probe.py irreg 4000 2 "pixel_order=0" "pixel_order=2" 2>&1 | grep -v amdgpu >> $OUT/part_probe.txt
timeout 400 python -m pytest tests -m gpu -x -q -k "golden_500 or pixels_bit_exact or first_frames or camera_path or parts_rendered_in_place or solo_pixels or random_parity" > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log
FUZZ_FORCE=pixel_order=2,adaptive_order=1,handover=1,px_solo=0 timeout 100 python tools/fuzz_parity.py 70 12001 300 100000 > $OUT/fuzz_model.txt 2>&1; tail -2 $OUT/fuzz_model.txt
echo r05b done
