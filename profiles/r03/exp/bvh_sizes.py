#!/usr/bin/env python3
"""prepare_scene (GPU BVH build) against the number of spheres: steady-state wall time per call, host-timed around a sync."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import raytracers_amd as R
ctx = R.Context()
rng = np.random.default_rng(7)
for n in (1000, 1089, 4000, 10000, 16384, 16641, 30000, 50000, 90000, 200000, 1000000):
    s = np.zeros((n, 7), np.float32)
    s[:, 0:3] = rng.uniform(-100, 100, (n, 3)); s[:, 3:6] = rng.uniform(0, 1, (n, 3)); s[:, 6] = rng.uniform(0.1, 1.0, n)
    sc = ctx.scene_from_spheres(s, (0, 0, -300), (0, 0, 0), 60.0)
    ps = [R.prepare_scene(100, 100, sc) for _ in range(3)]
    ctx.sync()
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        p = R.prepare_scene(100, 100, sc)
        p.free()
    ctx.sync()
    dt = (time.perf_counter() - t0) / reps
    print(f"n = {n:8d}: {dt * 1e3:7.3f} ms per prepare_scene (height {ps[0].height})", flush=True)
    for p in ps:
        p.free()
