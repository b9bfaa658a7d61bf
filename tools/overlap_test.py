#!/usr/bin/env python3
"""Experiment: throughput when S frames are in flight on one GPU (S contexts on S streams)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracers_amd as R

h = w = 1000
for S in (1, 2, 3, 4, 6, 8):
    ctxs, items = [], []
    streams = [torch.cuda.Stream() for _ in range(S)]
    for s in range(S):
        c = R.Context(0, stream=streams[s].cuda_stream)
        c.set_variant(3)
        ctxs.append(c)
        for scene in ("rgbbox", "irreg"):
            ps = R.prepare_scene(h, w, c.scene(scene))
            out = torch.empty((h, w), dtype=torch.int32, device="cuda")
            items.append((s, ps, out))
    def step(k):
        for j in range(2):
            s, ps, out = items[(k % S) * 2 + j]
            R.render_into(out.data_ptr(), h, w, ps)
    for k in range(3 * S):
        step(k)
    torch.cuda.synchronize()
    K = 60
    t0 = time.perf_counter()
    for k in range(K):
        step(k)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    rays = (4022099 + 1728608) * K
    print(f"S={S}: {dt / K * 1e3:.3f} ms/step  {rays / dt / 1e6:.0f} Mray/s", flush=True)
    for c in ctxs:
        c.close()
