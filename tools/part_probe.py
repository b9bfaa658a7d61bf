#!/usr/bin/env python3
"""One part of W of a frame, rendered repeatedly (warm view, render + sync): kernel time by events.
usage: part_probe.py scene size W "opt=v,opt=v" ..."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracers_amd as R
scene, n, W = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
for spec in sys.argv[4:] or [""]:
    ctx = R.Context(0, torch.cuda.current_stream().cuda_stream)
    for kv in spec.split(","):
        if kv:
            k, v = kv.split("=")
            ctx.set_option(k, int(v))
    ps = R.prepare_scene(n, n, ctx.scene(scene))
    res = []
    for part in range(W):
        rows = R.part_rows(n, part, W)
        out = torch.empty((rows, n), dtype=torch.int32, device="cuda")
        for _ in range(4):
            R.render_into(out.data_ptr(), n, n, ps, part=part, nparts=W)
            torch.cuda.synchronize()
        ts = []
        for _ in range(8):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); R.render_into(out.data_ptr(), n, n, ps, part=part, nparts=W); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        res.append(float(np.median(ts)))
    print(f"[{spec}] {scene} {n}x{n} part of {W}: slowest {max(res)*1e3:.0f} us, fastest {min(res)*1e3:.0f} us", flush=True)
    ctx.close()
