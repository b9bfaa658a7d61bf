#!/usr/bin/env python3
"""A rank's share of the bench's K = 20 steps at world size 8 (and 4): the two scenes' batch launches on two streams --
which goes first, and how many persistent workgroups does irreg's launch take?  (A persistent workgroup keeps its CU's LDS
until its last wave is done: irreg's launch ends with lone bounce chains, one wave on a CU that rgbbox's launch could use.)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from raytracers_amd.dist import HipPartRenderer, max_part_rows
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
K = 20
streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
for W in (8, 4):
    for gd_irreg, gd_rgb in ((0, 0), (2, 0), (4, 0), (8, 0), (2, 2), (4, 2)):
        with torch.cuda.stream(streams[0]):
            a = HipPartRenderer("irreg", 1000, 1000, dev, options={"grid_div": gd_irreg} if gd_irreg else None)
        with torch.cuda.stream(streams[1]):
            b = HipPartRenderer("rgbbox", 1000, 1000, dev, options={"grid_div": gd_rgb} if gd_rgb else None)
        rows = max_part_rows(1000, W)
        for first in ("irreg", "rgbbox"):
            res = []
            for p in range(min(W, 3)):
                oa = torch.zeros((K * rows, 1000), dtype=torch.int32, device=dev)
                ob = torch.zeros((K * rows, 1000), dtype=torch.int32, device=dev)
                def run():
                    seq = [(a, oa, streams[0]), (b, ob, streams[1])]
                    for pr, o, st in (seq if first == "irreg" else seq[::-1]):
                        with torch.cuda.stream(st):
                            pr.batch(p, W, K, o, rows * 1000)
                for _ in range(3): run()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(5):
                    run()
                    torch.cuda.synchronize()
                res.append(1e6 * (time.perf_counter() - t0) / 5 / K)
            print(f"W={W} grid_div irreg {gd_irreg} rgbbox {gd_rgb}, {first} first: {max(res):.1f} us/step (parts 0..2: {' '.join('%.1f' % r for r in res)})", flush=True)
