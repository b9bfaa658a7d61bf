#!/bin/bash
# Round 5, call D: the hybrid pixel list (long chains by their own length, the bulk in tile order by the tile's longest bulk chain),
# look-ahead read reverted; against the exact list (px_hybrid=0), the tile tickets (pixel_order=0) and call B's library (pxb).
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05d; mkdir -p $OUT
export AB_TIMEOUT=60
{
for rep in 1 2; do
for s in rgbbox irreg; do
echo "new|$s|1000|-r 20|pixel_order=0"
echo "new|$s|1000|-r 20|pixel_order=2"
echo "new|$s|1000|-r 20|pixel_order=2 px_hybrid=0"
done; done
for s in rgbbox irreg; do for n in 500 300 200 2000; do
echo "new|$s|$n|-r 20|pixel_order=0"
echo "new|$s|$n|-r 20|pixel_order=2"
echo "new|$s|$n|-r 20|pixel_order=2 px_hybrid=0"
done; done
echo "new|irreg|4000|-r 5|pixel_order=0"
echo "new|irreg|4000|-r 5|pixel_order=2"
echo "new|irreg|4000|-r 5|pixel_order=2 px_hybrid=0"
echo "new|big|2000|-r 3|pixel_order=0"
echo "new|big|2000|-r 3|pixel_order=2"
echo "new|big|2000|-r 3|pixel_order=2 px_hybrid=0"
echo "new|rgbbox|1000|-r 20|pixel_order=2 px_g64=180"
echo "new|rgbbox|1000|-r 20|pixel_order=2 px_g64=320"
echo "new|rgbbox|1000|-r 20|pixel_order=2 px_g64=400"
echo "new|rgbbox|1000|-r 20|pixel_order=2 px_g16=90 px_g32=140"
echo "new|rgbbox|1000|-r 20|pixel_order=2 px_ray_ns=250"
echo "new|rgbbox|1000|-r 20|pixel_order=2 thr_shade=32"
echo "new|rgbbox|1000|-r 20|pixel_order=2 thr_shade=48"
echo "new|irreg|1000|-r 20|pixel_order=2 px_g64=250"
echo "new|irreg|1000|-r 20|pixel_order=2 px_g64=450"
echo "new|irreg|1000|-r 20|pixel_order=2 px_g1=35"
echo "new|irreg|1000|-r 20|pixel_order=2 px_g8=95 px_g16=135 px_g32=200"
echo "new|irreg|1000|-r 20|pixel_order=2 px_ray_ns=250"
echo "new|irreg|1000|-r 20|pixel_order=2 grid_div=2"
} | bash tools/gpu_ab.sh r05d/ab > /dev/null
for W in 8 4 2; do timeout 100 python tools/part_probe.py irreg 4000 $W "pixel_order=0" "pixel_order=2" "pixel_order=2,px_hybrid=0" 2>&1 | grep -v amdgpu; done > $OUT/part_probe.txt
timeout 100 python tools/part_probe.py rgbbox 1000 8 "pixel_order=0" "pixel_order=2" "pixel_order=2,px_hybrid=0" 2>&1 | grep -v amdgpu >> $OUT/part_probe.txt
FUZZ_FORCE=pixel_order=2,adaptive_order=1,handover=1 timeout 100 python tools/fuzz_parity.py 70 17001 300 100000 > $OUT/fuzz.txt 2>&1; tail -2 $OUT/fuzz.txt
timeout 300 python -m pytest tests -m gpu -x -q -k "golden_500 or pixels_bit_exact or first_frames or camera_path or parts_rendered_in_place" > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log
echo r05d done
