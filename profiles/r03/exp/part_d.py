#!/usr/bin/env python3
"""a rank's eighth (and quarter) of irreg 4000x4000, one frame at a time, host-built BVH (the library in place decides the treelet cut)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from raytracers_amd.dist import HipPartRenderer, max_part_rows
dev = torch.device("cuda", 0)
for scene, n in (("irreg", 4000), ("irreg", 2000), ("irreg", 500), ("rgbbox", 500)):
    pr = HipPartRenderer(scene, n, n, dev, options={"gpu_build": 0})
    for W in ((8, 4) if n == 4000 else (1,)):
        res = []
        for p in range(min(W, 4)):
            o = torch.zeros((max_part_rows(n, W), n), dtype=torch.int32, device=dev)
            for _ in range(4): pr(p, W, o)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(8):
                pr(p, W, o)
                torch.cuda.synchronize()
            res.append(1e6 * (time.perf_counter() - t0) / 8)
        print(f"{sys.argv[1]}: {scene} {n}x{n} part of {W}: {max(res):.0f} us slowest ({' '.join('%.0f' % r for r in res)})", flush=True)
