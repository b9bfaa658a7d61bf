#!/bin/bash
# (-> profiles/r04/exp/e2, e3.  The options `wide` and `scout` of that day's library no longer exist: see the *_as_measured.patch files.)
# Round 4, second GPU call: the WIDE and COLD instantiations -- parity first, then A/B.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r04b
mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -x -q -k "scouted or persistent_knobs or big_2000 or golden_500 or pixels_bit_exact or camera_path or adaptive_tile_order or solo_pixels or many_views" > $OUT/pytest.log 2>&1
echo "pytest exit $?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
bash tools/gpu_ab.sh r04b/ab <<'AB'
new|big|2000|-r 4|
new|big|2000|-r 4|wide=1
new|irreg|1000|-r 0 -B 20|
new|irreg|1000|-r 0 -B 20|wide=1
new|irreg|4000|-r 4|
new|irreg|4000|-r 4|wide=1
new|irreg|1000|-r 12|
new|irreg|1000|-r 12|wide=1
new|big|2000|-r 4|
new|big|2000|-r 4|wide=1
AB
timeout 300 python tools/cold_probe.py 1000 "scout=0" "scout=1" "scout=1,cold_hold_depth=6" "scout=1,cold_hold_depth=20" "scout=1,cold_hold_depth=64" 2>&1 | grep -v amdgpu.ids | tee $OUT/cold_probe.txt
timeout 200 python tools/cold_probe.py 500 "scout=0" "scout=1" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/cold_probe.txt
timeout 100 python tools/fuzz_parity.py 50 7 > $OUT/fuzz.txt 2>&1; tail -2 $OUT/fuzz.txt
timeout 120 python tools/fuzz_parity.py 70 11 520 30000 > $OUT/fuzz_large.txt 2>&1; tail -2 $OUT/fuzz_large.txt
echo r04b done
