#!/bin/bash
# (-> profiles/r04/exp/e4.  The in-launch scout is not in the product: scout_in_launch_hot_list_as_measured.patch.)
# Round 4, third GPU call: the in-launch scout (COLD instantiation): parity, then cold-frame timings.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r04c
mkdir -p $OUT
timeout 500 python -m pytest tests -m gpu -x -q -k "scouted or golden_500 or pixels_bit_exact or camera_path or many_views or adaptive_tile_order" > $OUT/pytest.log 2>&1
echo "pytest exit $?" >> $OUT/pytest.log
tail -4 $OUT/pytest.log
timeout 300 python tools/cold_probe.py 1000 "scout=0" "scout=1" "scout=1,cold_hold_depth=6" "scout=1,cold_hold_depth=24" 2>&1 | grep -v amdgpu.ids | tee $OUT/cold_probe.txt
timeout 200 python tools/cold_probe.py 500 "scout=0" "scout=1" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/cold_probe.txt
timeout 200 python tools/cold_probe.py 1400 "scout=0" "scout=1" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/cold_probe.txt
timeout 120 python tools/fuzz_parity.py 80 21 520 30000 > $OUT/fuzz_large.txt 2>&1; tail -2 $OUT/fuzz_large.txt
echo r04c done
