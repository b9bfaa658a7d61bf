#!/bin/bash
# Round 5, call E: the exact pixel list as the product default (gates: <= 65536 tiles; scenes read from L2 >= 1024 tiles), the ORD
# instantiation without the solo call for lists without a one-pixel class, first frames with the per-pixel record + list sort,
# the issue-peak microbenchmark with the round's new instruction classes, the GPU suite, the bench line.
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05e; mkdir -p $OUT
export AB_TIMEOUT=60
{
for rep in 1 2; do
for s in rgbbox irreg; do
echo "new|$s|1000|-r 20|pixel_order=0"
echo "new|$s|1000|-r 20|"
done
echo "new|rgbbox|1000|-r 20|px_g64=320"
echo "new|rgbbox|1000|-r 20|px_ray_ns=250"
echo "new|rgbbox|1000|-r 20|px_g64=320 px_ray_ns=250"
done
for s in rgbbox irreg; do for n in 500 300 200 2000 700 1400; do
echo "new|$s|$n|-r 20|pixel_order=0"
echo "new|$s|$n|-r 20|"
done; done
echo "new|irreg|4000|-r 5|pixel_order=0"
echo "new|irreg|4000|-r 5|pixel_order=2"
echo "new|big|2000|-r 3|pixel_order=0"
echo "new|big|2000|-r 3|"
} | bash tools/gpu_ab.sh r05e/ab > /dev/null
for W in 8 4 2; do timeout 100 python tools/part_probe.py irreg 4000 $W "pixel_order=0" "" 2>&1 | grep -v amdgpu; done > $OUT/part_probe.txt
timeout 150 python tools/cold_probe.py 1000 "pixel_order=0" "pixel_order=1" 2>&1 | grep -v amdgpu > $OUT/cold_probe_1000.txt
timeout 150 python tools/cold_probe.py 500 "pixel_order=0" "pixel_order=1" 2>&1 | grep -v amdgpu > $OUT/cold_probe_500.txt
: > $OUT/issue_peak_new.txt
for k in k_add k_cndmask_sgpr k_bfi k_xor k_or k_lshlrev k_ashrrev k_max k_max3 k_med3 k_perm k_and_or k_lshl_or k_lshl_add k_add3 k_sub_u32 k_min_u32 k_fmac k_pk_fma k_pk_mov k_pk_add k_mbcnt; do
  timeout 60 ./build/issue_peak -W 40 -k $k 2>&1 | grep -v '^#' >> $OUT/issue_peak_new.txt
done
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_line.json 2> $OUT/bench.err; tail -2 $OUT/bench.err
echo r05e done
