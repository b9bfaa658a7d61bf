"""bench-vs-probe difference hunt: same lanes, vary (steps, via ShardedStep or direct)."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracers_amd.dist import HipPartRenderer, ShardedStep
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
frames = [("rgbbox", 1000, 1000), ("irreg", 1000, 1000)]
for S in (8, 16):
    streams = [torch.cuda.Stream(dev) for _ in range(S)]
    lanes = []
    for st in streams:
        with torch.cuda.stream(st):
            prs = [HipPartRenderer(s, h, w, dev, options={"grid_div": 4}) for s, h, w in frames]
            lanes.append((prs, ShardedStep([(pr, h, w) for pr, (_, h, w) in zip(prs, frames)], dev)))
    torch.cuda.synchronize()
    for mode in ("direct", "step"):
        def step(k):
            li = k % S
            with torch.cuda.stream(streams[li]):
                prs, ss = lanes[li]
                if mode == "step":
                    ss.render(None)
                else:
                    for pr, o in zip(prs, ss.outs):
                        pr(0, 1, o)
        for n in (50, 200, 1000):
            for rep in range(2):
                for k in range(2 * S): step(k)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for k in range(n): step(k)
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                print(f"S={S} mode={mode} n={n}: {1e6*(t2-t0)/n:.1f} us/step", flush=True)
