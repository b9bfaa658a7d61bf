#!/usr/bin/env python3
"""How long does a frame's longest bounce chain take on an otherwise idle chip?  Renders every 8-row band of a frame
as a launch of its own (125 bands at 1000 rows: <= 125 busy waves on 256 CUs, every wave alone on its CU) and prints
the slowest bands next to the whole frame's time: the frame cannot be faster than its slowest band.
usage: chain_probe.py scene h w [opt=value ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracers_amd as R

scene, h, w = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
ctx = R.Context()
for kv in sys.argv[4:]:
    k, v = kv.split("=")
    ctx.set_option(k, int(v))
ps = R.prepare_scene(h, w, ctx.scene(scene))
buf = ctx.alloc_i32(h * w)
whole = R.render_timed(buf.ptr, h, w, ps, 4, 10)
nb = (h + 7) // 8
t = np.zeros(nb)
for b in range(nb):
    t[b] = R.render_timed(buf.ptr, h, w, ps, 3, 5, part=b, nparts=nb).min()
order = np.argsort(t)[::-1]
print(f"{scene} {w}x{h} {' '.join(sys.argv[4:])}: whole frame {whole.mean():.4f} ms (min {whole.min():.4f}); bands alone: "
      f"sum {t.sum():.3f} ms, slowest " + ", ".join(f"band {i}: {t[i]:.4f}" for i in order[:4]) + f"; median {np.median(t):.4f}")
