#!/bin/bash
# Round 5, call C: coarse bins for the bulk of the pixel list, look-ahead counter read, re-fitted model constants; against call B's library (pxb).
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05c; mkdir -p $OUT
export AB_TIMEOUT=60
{
for rep in 1 2; do
for s in rgbbox irreg; do
echo "new|$s|1000|-r 20|pixel_order=0"
echo "new|$s|1000|-r 20|"
echo "new|$s|1000|-r 20|px_coarse=0"
echo "pxb|$s|1000|-r 20|pixel_order=2"
done; done
for s in rgbbox irreg; do for n in 500 300 200 2000; do
echo "new|$s|$n|-r 20|pixel_order=0"
echo "new|$s|$n|-r 20|"
done; done
echo "new|irreg|4000|-r 5|pixel_order=0"
echo "new|irreg|4000|-r 5|pixel_order=2"
echo "new|irreg|4000|-r 5|pixel_order=2 px_coarse=0"
echo "pxb|irreg|4000|-r 5|pixel_order=2"
echo "new|big|2000|-r 3|pixel_order=0"
echo "new|big|2000|-r 3|pixel_order=2"
echo "new|rgbbox|1000|-r 20|px_g64=200"
echo "new|rgbbox|1000|-r 20|px_g64=300"
echo "new|rgbbox|1000|-r 20|px_g32=140"
echo "new|rgbbox|1000|-r 20|px_g32=80"
echo "new|rgbbox|1000|-r 20|px_ray_ns=250"
echo "new|rgbbox|1000|-r 20|px_ray_ns=350"
echo "new|irreg|1000|-r 20|px_g1=35"
echo "new|irreg|1000|-r 20|px_g8=120"
echo "new|irreg|1000|-r 20|px_g16=170"
echo "new|irreg|1000|-r 20|px_g32=260"
echo "new|irreg|1000|-r 20|px_g64=420"
echo "new|irreg|1000|-r 20|px_ray_ns=250"
echo "new|irreg|500|-r 20|px_g1=35"
echo "new|irreg|500|-r 20|px_g8=120 px_g16=170"
echo "new|rgbbox|1000|-r 0 -B 20|"
echo "new|irreg|1000|-r 0 -B 20|"
echo "pxb|rgbbox|1000|-r 0 -B 20|"
echo "pxb|irreg|1000|-r 0 -B 20|"
} | bash tools/gpu_ab.sh r05c/ab > /dev/null
timeout 100 python tools/part_probe.py irreg 4000 8 "pixel_order=0" "" "px_coarse=0" 2>&1 | grep -v amdgpu > $OUT/part_probe.txt
timeout 100 python tools/part_probe.py irreg 4000 4 "pixel_order=0" "" 2>&1 | grep -v amdgpu >> $OUT/part_probe.txt
timeout 100 python tools/part_probe.py irreg 4000 2 "pixel_order=0" "pixel_order=2" 2>&1 | grep -v amdgpu >> $OUT/part_probe.txt
timeout 100 python tools/part_probe.py rgbbox 1000 8 "pixel_order=0" "" 2>&1 | grep -v amdgpu >> $OUT/part_probe.txt
FUZZ_FORCE=pixel_order=2,adaptive_order=1,handover=1 timeout 100 python tools/fuzz_parity.py 70 15001 300 100000 > $OUT/fuzz.txt 2>&1; tail -2 $OUT/fuzz.txt
timeout 300 python -m pytest tests -m gpu -x -q -k "golden_500 or pixels_bit_exact or first_frames or camera_path or parts_rendered_in_place or bench_line_contract" > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log
echo r05c done
