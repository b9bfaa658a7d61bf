#!/bin/bash
# Round 5, call A: pixel tickets (the ORD instantiation) -- first A/B against tile tickets + threshold sweep, then the GPU suite and a fuzz run.
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05a; mkdir -p $OUT
export AB_TIMEOUT=60
{
for s in rgbbox irreg; do for n in 1000 500; do
echo "new|$s|$n|-r 20|pixel_order=0"
echo "new|$s|$n|-r 20|pixel_order=2"
done; done
for s in rgbbox irreg; do
for v in 12 16 20 32 255; do echo "new|$s|1000|-r 20|pixel_order=2 px_solo=$v px_w8=$v"; done
for v in 10 18 22; do echo "new|$s|1000|-r 20|pixel_order=2 px_w16=$v"; done
for v in 5 7 12 14; do echo "new|$s|1000|-r 20|pixel_order=2 px_w32=$v"; done
for v in 16 20; do echo "new|$s|1000|-r 20|pixel_order=2 px_w8=$v"; done
for v in 7 3 1 0; do echo "new|$s|1000|-r 20|pixel_order=2 px_hold=$v"; done
for v in 1 2 16; do echo "new|$s|1000|-r 20|pixel_order=2 px_solo_div=$v"; done
echo "new|$s|1000|-r 20|pixel_order=2 grid_div=1"
echo "new|$s|1000|-r 20|pixel_order=2 thr_shade=32"
echo "new|$s|1000|-r 20|pixel_order=2 thr_shade=48"
done
echo "new|irreg|4000|-r 5|pixel_order=0"
echo "new|irreg|4000|-r 5|pixel_order=2"
echo "new|big|2000|-r 3|pixel_order=0"
echo "new|big|2000|-r 3|pixel_order=2"
echo "new|irreg|2000|-r 8|pixel_order=0"
echo "new|irreg|2000|-r 8|pixel_order=2"
echo "new|rgbbox|200|-r 20|pixel_order=0"
echo "new|rgbbox|200|-r 20|pixel_order=2"
echo "new|rgbbox|1000|-r 0 -B 20|"
echo "new|irreg|1000|-r 0 -B 20|"
} | bash tools/gpu_ab.sh r05a/ab > /dev/null
timeout 100 python tools/part_probe.py irreg 4000 8 "pixel_order=0" "pixel_order=2" "pixel_order=2,px_w16=10,px_w32=6" 2>&1 | grep -v amdgpu > $OUT/part_probe.txt
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log
timeout 150 python tools/fuzz_parity.py 120 501 > $OUT/fuzz_small.txt 2>&1; tail -2 $OUT/fuzz_small.txt
FUZZ_FORCE=pixel_order=2,adaptive_order=1,handover=1 timeout 120 python tools/fuzz_parity.py 90 9001 300 100000 > $OUT/fuzz_forced.txt 2>&1; tail -2 $OUT/fuzz_forced.txt
echo r05a done
