#!/usr/bin/env python3
"""a rank's share of the bench's K = 20 steps at world size W, one scene's batch launch ALONE on the GPU: how long is it,
and how much of that is its work (the same launch's share of a 200-frame batch)?"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from raytracers_amd.dist import HipPartRenderer, max_part_rows
dev = torch.device("cuda", 0)
opts = dict(kv.split("=") for kv in sys.argv[1:])
opts = {k: int(v) for k, v in opts.items()}
for scene in ("irreg", "rgbbox"):
    pr = HipPartRenderer(scene, 1000, 1000, dev, options=opts)
    for W in (8, 4, 1):
        for K in (20, 200):
            rows = max_part_rows(1000, W)
            out = torch.zeros((K * rows, 1000), dtype=torch.int32, device=dev)
            ts = []
            for p in range(min(W, 3)):
                for _ in range(3):
                    pr.batch(p, W, K, out, rows * 1000)
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(4):
                    pr.batch(p, W, K, out, rows * 1000)
                b.record()
                torch.cuda.synchronize()
                ts.append(a.elapsed_time(b) / 4)
            print(f"{scene} W={W} K={K}: launch {max(ts):.3f} ms (parts 0..{len(ts)-1}: {' '.join('%.3f' % t for t in ts)}) = {max(ts) / K * 1e3:.1f} us per step", flush=True)
