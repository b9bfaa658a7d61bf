#!/usr/bin/env python3
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracers_amd as R
from raytracers_amd.dist import HipPartRenderer, ShardedRenderer
h = w = 1000
S = 4
dev = torch.device("cuda", 0)
streams = [torch.cuda.Stream(dev) for _ in range(S)]
lanes = []
for st in streams:
    with torch.cuda.stream(st):
        lane = []
        for scene in ("rgbbox", "irreg"):
            pr = HipPartRenderer(scene, h, w, dev)
            lane.append((pr, ShardedRenderer(pr, h, w, dev)))
        lanes.append(lane)
torch.cuda.synchronize()
def run(mode, K=60):
    evs = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(2)] for _ in range(K)]
    def step(k):
        li = k % S
        with torch.cuda.stream(streams[li]):
            for i, (pr, sr) in enumerate(lanes[li]):
                if mode == "raw":
                    pr(0, 1, sr.send)
                elif mode == "raw+place":
                    pr(0, 1, sr.send); sr._assemble([sr.send])
                elif mode == "render":
                    sr.render()
                elif mode == "render+events":
                    sr.render(evs[k][i])
    for k in range(8): step(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(K): step(k)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{mode:16s}: enqueue {1e3*(t1-t0)/K:.3f} ms/step, total {1e3*(t2-t0)/K:.3f} ms/step", flush=True)
for m in ("raw", "raw+place", "render", "render+events", "raw"):
    run(m)
