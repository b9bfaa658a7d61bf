#!/usr/bin/env python3
"""Per-wave timeline of the pooled kernel (rt_render_trace): where does the frame time go?
usage: trace_waves.py scene h w [opt=value ...]"""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
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
for _ in range(3):   # warm up + let the adaptive order of the traced view settle
    R.render_into(buf.ptr, h, w, ps, part=part, nparts=nparts)
ctx.sync()
rec = np.zeros((8192, 16), dtype=np.uint64)
n = C.c_int32()
ctx._check(lib.rt_render_trace(ctx._h, ps._h, h, w, 50, rec.ctypes.data, 8192, C.byref(n)))
rec = rec[: n.value].astype(np.int64)
t0 = rec[:, 0].min()
TICK_US = 0.01   # wall clock: 100 MHz
start, exh, end = (rec[:, 0] - t0) * TICK_US, (rec[:, 1] - t0) * TICK_US, (rec[:, 2] - t0) * TICK_US
exh = np.where(rec[:, 1] > 0, exh, end)
print(f"{scene} {w}x{h}: {n.value} waves; first wave start -> last wave end {end.max():.1f} us")
for name, v in (("start", start), ("queue exhausted", exh), ("end", end)):
    print(f"  {name:16s} min {v.min():8.1f}  p50 {np.median(v):8.1f}  p90 {np.percentile(v, 90):8.1f}  max {v.max():8.1f} us")
life_us = end - start
print(f"  mean wave lifetime {life_us.mean():.1f} us = {life_us.mean() / end.max():.2f} of the span; shader clock while alive {np.mean(rec[:, 4] / np.maximum(life_us, 0.01)):.0f} MHz")
maxbox, maxleaf = (rec[:, 7] >> 16) & 0xFFFF, (rec[:, 7] >> 32) & 0xFFFF
rec[:, 7] &= 0xFFFF
print(f"  box stack high-water mark: max {maxbox.max()} p99 {int(np.percentile(maxbox, 99))} p50 {int(np.median(maxbox))} items"
      f" (capacity 64*(height+3)); leaf list: max {maxleaf.max()}")
ops = np.stack([rec[:, 3] & 0x1FFFFF, (rec[:, 3] >> 21) & 0x1FFFFF, (rec[:, 3] >> 42) & 0x1FFFFF], axis=1)
items_box, items_leaf = rec[:, 6] >> 32, rec[:, 6] & 0xFFFFFFFF
tot = ops.sum(axis=1)
print(f"  ops/wave: mean {tot.mean():.0f} max {tot.max()}  (BOX {ops[:,0].sum()} LEAF {ops[:,1].sum()} SHADE {ops[:,2].sum()})")
print(f"  lane efficiency: BOX {items_box.sum() / (64.0 * ops[:,0].sum()):.3f} LEAF {items_leaf.sum() / (64.0 * ops[:,1].sum()):.3f}")
print(f"  shader cycles per op (cycles lived / ops): mean {(rec[:, 4] / np.maximum(tot, 1)).mean():.0f}")
n_t, n_b2 = rec[:, 5] & 0xFFFFFFFF, rec[:, 5] >> 32
n_b1 = ops[:, 0] - n_t - n_b2
cyc = rec[:, 8:13]   # BOX, BOX2, BOXT, LEAF, SHADE
cnt = np.stack([n_b1, n_b2, n_t, ops[:, 1], ops[:, 2]], axis=1)
names = ("BOX", "BOX2", "BOXT", "LEAF", "SHADE")
print("  shader cycles per operation, by kind (all waves): " + ", ".join(
    f"{nm} {cyc[:, k].sum() / max(cnt[:, k].sum(), 1):.0f} x {cnt[:, k].sum()}" for k, nm in enumerate(names))
    + f"; between operations {(rec[:, 4].sum() - cyc.sum()) / max(tot.sum(), 1):.0f} per op")
late = np.argsort(end)[-6:]
for i in late[::-1]:
    per = ", ".join(f"{nm} {cyc[i, k] / max(cnt[i, k], 1):.0f}x{cnt[i, k]}" for k, nm in enumerate(names))
    print(f"    wave {i:5d}: start {start[i]:7.1f} exhausted {exh[i]:7.1f} end {end[i]:7.1f} us; ops {tot[i]:6d} (box {ops[i,0]} leaf {ops[i,1]} shade {ops[i,2]}) deepest chain {rec[i,7]}")
    print(f"                cycles/op: {per}; between ops {(rec[i, 4] - cyc[i].sum()) / max(tot[i], 1):.0f}")
print(f"  drain phase (exhausted -> end): mean {np.mean(end - exh):.1f} max {np.max(end - exh):.1f} us")
# option trace_solo=1: the solo loop's own counters (words 13 .. 15): cycles inside treelet operations / sphere-test operations / the rest of a
# bounce (ray_derive, root box, the winner's loads, shade), with their counts
M40 = (1 << 40) - 1
s_tre, n_tre, s_leaf, n_leaf, s_rest, n_ray = rec[:, 13] & M40, rec[:, 13] >> 40, rec[:, 14] & M40, rec[:, 14] >> 40, rec[:, 15] & M40, rec[:, 15] >> 40
if n_ray.sum() > 0:
    mhz = float(np.mean(rec[:, 4] / np.maximum(life_us, 0.01)))
    w = n_ray > 0
    per_bounce = (s_tre + s_leaf + s_rest)[w] / n_ray[w]
    print(f"  SOLO loop ({int(w.sum())} waves, {int(n_ray.sum())} rays): per ray {n_tre.sum() / n_ray.sum():.2f} treelet operations x {s_tre.sum() / max(n_tre.sum(), 1):.0f} cycles, "
          f"{n_leaf.sum() / n_ray.sum():.2f} sphere-test operations x {s_leaf.sum() / max(n_leaf.sum(), 1):.0f}, the rest of a bounce {s_rest.sum() / n_ray.sum():.0f} cycles")
    print(f"    cycles per bounce over the waves: mean {per_bounce.mean():.0f} p50 {np.median(per_bounce):.0f} p95 {np.percentile(per_bounce, 95):.0f} max {per_bounce.max():.0f}"
          f"  = {per_bounce.mean() / mhz:.2f} / {np.median(per_bounce) / mhz:.2f} / {np.percentile(per_bounce, 95) / mhz:.2f} / {per_bounce.max() / mhz:.2f} us at {mhz:.0f} MHz")
    deep = np.argsort(n_ray)[-5:][::-1]
    for i in deep:
        if n_ray[i]:
            print(f"    wave {i:5d}: {n_ray[i]} rays in the solo loop: treelet {s_tre[i] / max(n_tre[i], 1):.0f} x {n_tre[i] / n_ray[i]:.1f}, sphere {s_leaf[i] / max(n_leaf[i], 1):.0f} x {n_leaf[i] / n_ray[i]:.1f}, "
                  f"rest {s_rest[i] / n_ray[i]:.0f}: {(s_tre[i] + s_leaf[i] + s_rest[i]) / n_ray[i]:.0f} cycles per bounce")
