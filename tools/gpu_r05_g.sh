#!/bin/bash
# Round 5, call G: lazy sorts (the frame that records a view does not launch them), the sort kernels with their loads in flight and a
# lane-parallel header: first frames, second frames, warm frames; fuzz + suite on the new kernels.
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05g; mkdir -p $OUT
export AB_TIMEOUT=60
timeout 150 python tools/cold_probe.py 1000 "pixel_order=0" "pixel_order=1" 2>&1 | grep -v amdgpu > $OUT/cold_probe_1000.txt
timeout 150 python tools/cold_probe.py 500 "pixel_order=0" "pixel_order=1" 2>&1 | grep -v amdgpu > $OUT/cold_probe_500.txt
{
for s in rgbbox irreg; do
echo "new|$s|1000|-r 20|pixel_order=0"
echo "new|$s|1000|-r 20|"
done
} | bash tools/gpu_ab.sh r05g/ab > /dev/null
cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof_first -- $OLDPWD/build/rtbench -s rgbbox -n 1000 -m 1000 -r 3 > $OLDPWD/$OUT/rocprof_first.log 2>&1
cd $OLDPWD
find $OUT/prof_first -name "*kernel_stats.csv" -exec cp {} $OUT/first_frame_kernel_stats.csv \;
rm -rf $OUT/prof_first
timeout 120 python tools/fuzz_parity.py 90 31001 > $OUT/fuzz_small.txt 2>&1; tail -n1 $OUT/fuzz_small.txt
FUZZ_FORCE=pixel_order=2,adaptive_order=1,handover=1 timeout 100 python tools/fuzz_parity.py 70 33001 300 100000 > $OUT/fuzz_forced.txt 2>&1; tail -n1 $OUT/fuzz_forced.txt
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
tail -n3 $OUT/pytest_gpu.log
echo r05g done
