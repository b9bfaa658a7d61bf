#!/usr/bin/env python3
"""Per-wave end times of an instrumented launch (rt_render_trace), grouped: by the wave's place in its workgroup (wave % 4 = its SIMD),
by the longest chain it carried, by its XCD -- who ends late?   usage: trace_groups.py scene h w [opt=value ...]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracers_amd as R
from raytracers_amd._lib import lib

scene, h, w = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
ctx = R.Context()
ctx.set_variant(3)
for kv in sys.argv[4:]:
    k, v = kv.split("=")
    ctx.set_option(k, int(v))
ps = R.prepare_scene(h, w, ctx.scene(scene))
opts = dict(kv.split("=") for kv in sys.argv[4:])
part, nparts = int(opts.get("trace_part", 0)), int(opts.get("trace_nparts", 1))
buf = ctx.alloc_i32(h * w)
for _ in range(3):
    R.render_into(buf.ptr, h, w, ps, part=part, nparts=nparts)
ctx.sync()
rec = np.zeros((8192, 16), dtype=np.uint64)
n = C.c_int32()
ctx._check(lib.rt_render_trace(ctx._h, ps._h, h, w, 50, rec.ctypes.data, 8192, C.byref(n)))
rec = rec[: n.value].astype(np.int64)
t0 = rec[:, 0].min()
end = (rec[:, 2] - t0) * 0.01
exh = np.where(rec[:, 1] > 0, (rec[:, 1] - t0) * 0.01, end)
deep = rec[:, 7] & 0xFFFF
ops = (rec[:, 3] & 0x1FFFFF) + ((rec[:, 3] >> 21) & 0x1FFFFF) + ((rec[:, 3] >> 42) & 0x1FFFFF)
wave = np.arange(n.value)
wpw = 16
win, blk = wave % wpw, wave // wpw
print(f"{scene} {w}x{h} {' '.join(sys.argv[4:])}: {n.value} waves, span {end.max():.0f} us, end p50 {np.median(end):.0f} p90 {np.percentile(end, 90):.0f}")


def row(name, m):
    if m.sum() == 0:
        return
    print(f"  {name:34s} n {int(m.sum()):5d}  end p10 {np.percentile(end[m], 10):6.0f} p50 {np.median(end[m]):6.0f} p90 {np.percentile(end[m], 90):6.0f} max {end[m].max():6.0f}"
          f"   exhausted p50 {np.median(exh[m]):6.0f}  ops p50 {int(np.median(ops[m])):5d}  cycles/op p50 {np.median(rec[m, 4] / np.maximum(ops[m], 1)):6.0f}")


for s in range(4):
    row(f"wave % 4 == {s} (its SIMD)", win % 4 == s)
for s in range(4):
    row(f"wave in workgroup {4 * s}..{4 * s + 3}", win // 4 == s)
for lo, hi in ((0, 4), (5, 8), (9, 12), (13, 20), (21, 32), (33, 64)):
    row(f"longest chain {lo}..{hi}", (deep >= lo) & (deep <= hi))
for x in range(8):
    row(f"XCD {x} (workgroup % 8)", blk % 8 == x)
late = end >= np.percentile(end, 90)
print(f"  the last tenth: wave%4 counts {np.bincount(win[late] % 4, minlength=4)}, wave-in-workgroup//4 counts {np.bincount(win[late] // 4, minlength=4)}, "
      f"longest chain p10/p50/p90 {np.percentile(deep[late], [10, 50, 90])}")
