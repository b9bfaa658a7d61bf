#!/bin/bash
# Round 5, call F: the pixel list's sort without atomics and with a parallel scan (the first version's one-workgroup scan cost a view's
# first frame 0.2 ms): first frames, warm frames, fuzz on the new sort kernels.
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05f; mkdir -p $OUT
export AB_TIMEOUT=60
timeout 150 python tools/cold_probe.py 1000 "pixel_order=0" "pixel_order=1" 2>&1 | grep -v amdgpu > $OUT/cold_probe_1000.txt
timeout 150 python tools/cold_probe.py 500 "pixel_order=0" "pixel_order=1" 2>&1 | grep -v amdgpu > $OUT/cold_probe_500.txt
timeout 150 python tools/cold_probe.py 2000 "pixel_order=0" "pixel_order=1" 2>&1 | grep -v amdgpu > $OUT/cold_probe_2000.txt
{
for rep in 1 2; do for s in rgbbox irreg; do
echo "new|$s|1000|-r 20|pixel_order=0"
echo "new|$s|1000|-r 20|"
done; done
for s in rgbbox irreg; do for n in 500 700 1400; do
echo "new|$s|$n|-r 20|"
done; done
} | bash tools/gpu_ab.sh r05f/ab > /dev/null
timeout 100 python tools/part_probe.py irreg 4000 8 "pixel_order=0" "" 2>&1 | grep -v amdgpu > $OUT/part_probe.txt
cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof_first -- $OLDPWD/build/rtbench -s rgbbox -n 1000 -m 1000 -r 3 > $OLDPWD/$OUT/rocprof_first.log 2>&1
cd $OLDPWD
find $OUT/prof_first -name "*kernel_stats.csv" -exec cp {} $OUT/first_frame_kernel_stats.csv \;
rm -rf $OUT/prof_first
timeout 120 python tools/fuzz_parity.py 90 21001 > $OUT/fuzz_small.txt 2>&1; tail -1 $OUT/fuzz_small.txt
FUZZ_FORCE=pixel_order=2,adaptive_order=1,handover=1 timeout 100 python tools/fuzz_parity.py 70 23001 300 100000 > $OUT/fuzz_forced.txt 2>&1; tail -1 $OUT/fuzz_forced.txt
timeout 300 python -m pytest tests -m gpu -x -q -k "golden_500 or pixels_bit_exact or first_frames or camera_path or parts_rendered_in_place" > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
tail -2 $OUT/pytest_gpu.log
echo r05f done
