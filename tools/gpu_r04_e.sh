#!/bin/bash
# (-> profiles/r04/exp/e6.  `cold_first` / `cold_hold_depth` were that day's names; the product's option is `handover`.)
# Round 4: the COLD instantiation without scouts (dynamic hold + in-loop solo hand-over): parity, then first-frame timings.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r04e
mkdir -p $OUT
timeout 400 python -m pytest tests -m gpu -x -q -k "first_frames_of_new_views or golden_500 or pixels_bit_exact or camera_path or many_views" > $OUT/pytest.log 2>&1
echo "pytest exit $?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
timeout 300 python tools/cold_probe.py 1000 "cold_first=0" "cold_first=1" "cold_first=1,cold_hold_depth=6" "cold_first=1,cold_hold_depth=24" "cold_first=1,cold_hold_depth=64" "cold_first=0,grid_div=1" 2>&1 | grep -v amdgpu.ids | tee $OUT/cold_probe.txt
timeout 200 python tools/cold_probe.py 500 "cold_first=0" "cold_first=1" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/cold_probe.txt
timeout 200 python tools/cold_probe.py 1400 "cold_first=0" "cold_first=1" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/cold_probe.txt
timeout 100 python tools/fuzz_parity.py 60 31 520 30000 > $OUT/fuzz_large.txt 2>&1; tail -1 $OUT/fuzz_large.txt
echo r04e done
