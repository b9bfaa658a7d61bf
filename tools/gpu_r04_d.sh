#!/bin/bash
# Round 4, debug call of the in-launch scout: the hot list's word and the device-side counters after a first frame
# (-> profiles/r04/exp/e5; needs the library of scout_in_launch_hot_list_as_measured.patch built with -DRT_COLD_COUNTERS)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r04d
mkdir -p $OUT
RT_COLD_DEBUG=1 timeout 60 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $OUT/cold_debug2.txt | tail -30
import torch, raytracers_amd as R
ctx = R.Context(0, torch.cuda.current_stream().cuda_stream)
for scene in ("irreg", "rgbbox"):
    sc = ctx.scene(scene)
    img = torch.empty((1000, 1000), dtype=torch.int32, device="cuda")
    for rep in range(2):
        ps = R.prepare_scene(1000, 1000, sc)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); R.render_into(img.data_ptr(), 1000, 1000, ps); b.record(); torch.cuda.synchronize()
        print(scene, "first frame", a.elapsed_time(b), "ms", flush=True)
PY
echo r04d done
