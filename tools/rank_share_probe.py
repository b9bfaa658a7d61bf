"""How fast can ONE rank turn over its share of a step at world size W?  One-GPU stand-in for
the strong-scaling run: renders part p of W for every frame of the workload on S lanes (HIP
streams), no exchange.  With part = all, every part is measured in turn and the slowest one
(what a real W-GPU step would wait for) is reported.

usage: rank_share_probe.py <workload> W S grid_div [part|all] [steps]
       workload: 1000 (rgbbox + irreg 1000x1000, the bench step) | irreg4000
       (GPU_MAX_HW_QUEUES from the environment, default 20)"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracers_amd.dist import HipPartRenderer, max_part_rows

wl, W, S, gd = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
part = sys.argv[5] if len(sys.argv) > 5 else str(W - 1)
n = int(sys.argv[6]) if len(sys.argv) > 6 else 400
frames = {"1000": [("rgbbox", 1000, 1000), ("irreg", 1000, 1000)], "irreg4000": [("irreg", 4000, 4000)]}[wl]
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
streams = [torch.cuda.Stream(dev) for _ in range(S)]
lanes = []
for st in streams:
    with torch.cuda.stream(st):
        prs = [HipPartRenderer(s, h, w, dev, options={"grid_div": gd}) for s, h, w in frames]
        outs = [torch.zeros((max_part_rows(h, W), w), dtype=torch.int32, device=dev) for _, h, w in frames]
        lanes.append((prs, outs))
torch.cuda.synchronize()
worst = 0.0
for p in (range(W) if part == "all" else [int(part)]):
    def step(k):
        li = k % S
        with torch.cuda.stream(streams[li]):
            prs, outs = lanes[li]
            for pr, o in zip(prs, outs):
                pr(p, W, o)
    for k in range(2 * S): step(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(n): step(k)
    torch.cuda.synchronize()
    us = 1e6 * (time.perf_counter() - t0) / n
    worst = max(worst, us)
    print(f"Q={os.environ['GPU_MAX_HW_QUEUES']} {wl} W={W} S={S} grid_div={gd} part {p}: {us:.1f} us/step", flush=True)
print(f"RESULT {wl} W={W} S={S} grid_div={gd}: slowest part {worst:.1f} us/step", flush=True)
