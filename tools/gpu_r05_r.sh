#!/bin/bash
# Round 5: first frames with the bit-reversed visiting order as the default: the 10^6-sphere frame, a rank's eighth, 700 / 1400; the suite's first-frame tests.
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05r; mkdir -p $OUT
timeout 300 python tools/donate_probe.py "big:2000,irreg:4000:8,rgbbox:700,irreg:700,rgbbox:1400,irreg:1400,irreg:4000" "first_order=0" "first_order=1" 2>&1 | grep -v amdgpu > $OUT/first_frames_probe.txt
timeout 600 python -m pytest tests -m gpu -x -q -k "first_frames or camera_path or pixel_tickets or parts_rendered or tile_queue_layouts or knob or golden" > $OUT/pytest.log 2>&1; tail -n2 $OUT/pytest.log
echo done
