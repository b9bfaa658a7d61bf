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
# prepare_scene against the number of spheres (random spheres in a box: every size class of the GPU builder)
if len(sys.argv) > 1 and sys.argv[1] == "sizes":
    rng = np.random.default_rng(5)
    sizes = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else (3000, 6144, 6145, 8000, 10000, 15360, 15361, 20000, 40000, 90000, 131072, 131073, 300000, 1000000)
    for n in sizes:
        sph = np.empty((n, 7), np.float32)
        sph[:, 0:3] = rng.uniform(-100, 100, (n, 3))
        sph[:, 3:6] = rng.uniform(0, 1, (n, 3))
        sph[:, 6] = rng.uniform(0.2, 1.5, n)
        sc = ctx.scene_from_spheres(sph, (0, 0, 300), (0, 0, 0), 60.0)
        ts = []
        for i in range(8):
            t0 = time.perf_counter()
            ps = api.prepare_scene(200, 200, sc)
            ctx.sync()
            ts.append(1e6 * (time.perf_counter() - t0))
        print(f"n={n} height={ps.height if hasattr(ps, 'height') else '?'}", " ".join(f"{t:.0f}" for t in ts), "us")
