#!/usr/bin/env python3
"""First frames of views never seen before (the stateless figure: the reference's render keeps nothing between calls).
For each scene and option set: a fresh prepared scene per repetition, the first frame's time (events around the call,
the frame + the tile order's sort), the image's checksum against the oracle's table; then a 12-view camera
path rendered view by view.   usage: cold_probe.py [size=1000] ["opt=v,opt=v" ...]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import raytracers_amd as R

size = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
sets = sys.argv[2:] or ["handover=0", "handover=1"]
dev = torch.device("cuda", 0)
cks = bench.Checksummer(dev)
for spec in sets:
    ctx = R.Context(0, torch.cuda.current_stream().cuda_stream)
    for kv in spec.split(","):
        if kv:
            k, v = kv.split("=")
            ctx.set_option(k, int(v))
    for scene in ("rgbbox", "irreg"):
        h = w = size
        sc = ctx.scene(scene)
        img = torch.empty((h, w), dtype=torch.int32, device=dev)
        warm = R.prepare_scene(h, w, sc)
        for _ in range(3):
            R.render_into(img.data_ptr(), h, w, warm)
        torch.cuda.synchronize()
        first, second = [], []
        for rep in range(6):
            ps = R.prepare_scene(h, w, sc)
            torch.cuda.synchronize()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            img.fill_(0x5a5a5a5a)
            ev[0].record()
            R.render_into(img.data_ptr(), h, w, ps)
            ev[1].record()
            ctx.sync()                   # (the ABI's completion point, as the reference's harness uses it: the context's stream)
            ev[2].record()
            R.render_into(img.data_ptr(), h, w, ps)
            ev[3].record()
            ctx.sync()
            torch.cuda.synchronize()
            ok = cks(img) == bench.FRAME_CHECKSUM.get((scene, h, w), cks(img))
            first.append(ev[0].elapsed_time(ev[1]))
            second.append(ev[2].elapsed_time(ev[3]))
            if not ok:
                print(f"[{spec}] {scene}: WRONG PIXELS in a first frame")
            ps.free()
        # camera path, view by view
        ps = R.prepare_scene(h, w, sc)
        nb = 12
        cams = np.tile(np.asarray(ps.camera(), dtype=np.float32).reshape(1, 12), (nb, 1))
        cams[:, 0] += 0.05 * np.arange(nb, dtype=np.float32)
        cams[:, 3] += 0.05 * np.arange(nb, dtype=np.float32)
        per = []
        for f in range(nb):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            R.render_into(img.data_ptr(), h, w, ps, cam=cams[f])
            b.record()
            ctx.sync()
            torch.cuda.synchronize()
            per.append(a.elapsed_time(b))
        ps.free()
        # ... and the same path enqueued back to back, one sync at the end (what bench.py's camera_path_frame_by_frame_ms measures)
        ps = R.prepare_scene(h, w, sc)
        nb2 = 20
        cams2 = np.tile(np.asarray(ps.camera(), dtype=np.float32).reshape(1, 12), (nb2, 1))
        cams2[:, 0] += 0.05 * np.arange(nb2, dtype=np.float32)
        cams2[:, 3] += 0.05 * np.arange(nb2, dtype=np.float32)
        evb = [torch.cuda.Event(enable_timing=True) for _ in range(nb2 + 1)]
        torch.cuda.synchronize()
        evb[0].record()
        for f in range(nb2):
            R.render_into(img.data_ptr(), h, w, ps, cam=cams2[f])
            evb[f + 1].record()
        torch.cuda.synchronize()
        b2b = [evb[f].elapsed_time(evb[f + 1]) for f in range(nb2)]
        ps.free()
        print(f"[{spec}] {scene} {w}x{h}: first frame min {min(first):.3f} median {np.median(first):.3f} ms; second {np.median(second):.3f}; "
              f"camera path view by view: first {per[0]:.3f}, mean of the rest {np.mean(per[1:]):.3f} ms; back to back (no sync): first {b2b[0]:.3f}, mean of the rest {np.mean(b2b[1:]):.3f} ms", flush=True)
        warm.free()
    ctx.close()
