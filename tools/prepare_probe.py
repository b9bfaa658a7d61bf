import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracers_amd as R
from raytracers_amd import api
ctx = api.Context(0)
for name in ("rgbbox", "irreg"):
    sc = ctx.scene(name)
    ts = []
    ps = None
    for i in range(12):
        t0 = time.perf_counter()
        ps = api.prepare_scene(200, 200, sc)
        ctx.sync()
        ts.append(1e6 * (time.perf_counter() - t0))
    print(name, " ".join(f"{t:.0f}" for t in ts), "us")
import numpy as np
for name in ("rgbbox", "irreg"):
    ctx2 = api.Context(0)
    sc = ctx2.scene(name)
    ps = api.prepare_scene(1000, 1000, sc)
    ts = []
    buf = ctx2.alloc_i32(1000 * 1000)
    for i in range(6):
        t0 = time.perf_counter()
        api.render_into(buf.ptr, 1000, 1000, ps)
        ctx2.sync()
        ts.append(1e6 * (time.perf_counter() - t0))
    print("render", name, " ".join(f"{t:.0f}" for t in ts), "us")
