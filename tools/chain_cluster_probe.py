#!/usr/bin/env python3
"""Do long bounce chains cluster, and would a low-resolution scout frame find them?  (CPU only: the oracle.)
Per-pixel chain-length classes come from rendering the frame with max_depth d = 1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 50: a pixel
whose chain needs more than d rays is black at d (ray_colour returns light * 0 when the budget is spent, ray.fut:136-147), so
the smallest d at which it has its final colour bounds its chain.  Then: (a) how well do 1 / 4 / 8 sample pixels of an 8 x 8
tile predict the tile's longest chain, (b) the scout: a frame of 1 / div^2 of the pixels with max_depth `cut`, a tile flagged
when a scout pixel of its footprint is black -- how many tiles are flagged, and which fraction of the tiles with chains of
>= 8 / 16 / 32 are among them.      usage: chain_cluster_probe.py [size=1000]   (profiles/r04/README.md)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import oracle_lib as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
ds = [1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 50]
for scene in ("irreg", "rgbbox"):
    sc = O.OracleScene(scene)
    h = w = n
    imgs = {d: sc.render(h, w, max_depth=d)[0] for d in ds}
    final = imgs[50]
    cls = np.full((h, w), len(ds) - 1)
    same = np.ones((h, w), bool)
    for i in range(len(ds) - 1, -1, -1):
        same &= imgs[ds[i]] == final
        cls[same] = i
    depth = np.array(ds)[cls]
    print(f"{scene} {w}x{h}: pixels by chain-length class (<= d rays):", {d: int((depth == d).sum()) for d in ds})
    T = depth[:h // 8 * 8, :w // 8 * 8].reshape(h // 8, 8, w // 8, 8).transpose(0, 2, 1, 3).reshape(h // 8, w // 8, 64)
    tmax = T.max(axis=2)
    for name, idx in (("1 sample", [3 * 8 + 3]), ("4 samples", [9, 13, 41, 45]), ("8 samples", [0, 12, 18, 30, 33, 45, 51, 63])):
        smax = T[:, :, idx].max(axis=2)
        for thr in (16, 32):
            deep = tmax >= thr
            print(f"  {name} per tile: " + "; ".join(f"sample >= {st}: {(smax >= st).sum()} tiles flagged, recall of chains >= {thr}: "
                                                     f"{(deep & (smax >= st)).sum() / max(1, deep.sum()):.2f}" for st in (4, 6, 8)))
    for div in (3, 4):
        for cut in (3, 4, 6):
            hs, ws = (h + div - 1) // div, (w + div - 1) // div
            black = sc.render(hs, ws, max_depth=cut)[0] == 0
            flag = np.zeros_like(tmax, bool)
            for ty in range(h // 8):
                r0, r1 = (8 * ty) // div, (8 * ty + 7) // div
                for tx in range(w // 8):
                    c0, c1 = (8 * tx) // div, (8 * tx + 7) // div
                    flag[ty, tx] = black[r0:r1 + 1, c0:c1 + 1].any()
            print(f"  scout 1/{div * div} of the pixels, bounce limit {cut}: {flag.sum()} of {flag.size} tiles flagged; recall of tiles with chains "
                  + ", ".join(f">= {t}: {(flag & (tmax >= t)).sum() / max(1, (tmax >= t).sum()):.2f} (of {(tmax >= t).sum()})" for t in (8, 16, 32)), flush=True)
