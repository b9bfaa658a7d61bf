#!/usr/bin/env python3
"""Single frames under option sets, one after the other in one process: first frame of a fresh view (4 repetitions) and the
steady frame (median of 16 after 4), every frame's checksum held against the first option set's (and the oracle's table
where it has the size).  Parts: "scene:size:W" renders part 0 and part W-1 of W.
usage: donate_probe.py "scene:size[:W],..." "opt=v,opt=v" ..."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import raytracers_amd as R

cases = [c.split(":") for c in sys.argv[1].split(",")]
sets = sys.argv[2:]
dev = torch.device("cuda", 0)
cks = bench.Checksummer(dev)
ref = {}
for spec in sets:
    ctx = R.Context(0, torch.cuda.current_stream().cuda_stream)
    for kv in spec.split(","):
        if kv:
            k, v = kv.split("=")
            ctx.set_option(k, int(v))
    for case in cases:
        scene, n = case[0], int(case[1])
        W = int(case[2]) if len(case) > 2 else 1
        sc = ctx.scene(scene)
        for part in sorted({0, W - 1}):
            rows = R.part_rows(n, part, W)
            img = torch.empty((rows, n), dtype=torch.int32, device=dev)
            kw = dict(part=part, nparts=W) if W > 1 else {}
            bad = 0

            def frame(ps):
                global_bad = 0
                img.fill_(0x5a5a5a5a)
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                R.render_into(img.data_ptr(), n, n, ps, **kw)
                b.record()
                torch.cuda.synchronize()
                c = cks(img)
                key = (scene, n, W, part)
                want = bench.FRAME_CHECKSUM.get((scene, n, n)) if W == 1 else None
                want = ref.setdefault(key, c if want is None else want)
                if c != want:
                    global_bad = 1
                return a.elapsed_time(b), global_bad

            first = []
            for _ in range(4):
                ps = R.prepare_scene(n, n, sc)
                t, e = frame(ps)
                first.append(t)
                bad += e
                ps.free()
            ps = R.prepare_scene(n, n, sc)
            steady = []
            for i in range(20):
                t, e = frame(ps)
                bad += e
                if i >= 4:
                    steady.append(t)
            ps.free()
            print(f"[{spec}] {scene} {n}x{n} part {part}/{W}: first {min(first)*1e3:.0f} (median {np.median(first)*1e3:.0f}) us, "
                  f"steady median {np.median(steady)*1e3:.0f} min {min(steady)*1e3:.0f} us"
                  + (f"   WRONG PIXELS in {bad} frame(s)" if bad else ""), flush=True)
    ctx.close()
