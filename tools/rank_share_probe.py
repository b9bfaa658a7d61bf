"""How fast can ONE rank turn over its share of a step at world size W?  (one-GPU stand-in for
the strong-scaling run: renders part `p` of W for both scenes on S lanes, no exchange)"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracers_amd.dist import HipPartRenderer, max_part_rows
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
frames = [("rgbbox", 1000, 1000), ("irreg", 1000, 1000)]
# usage: rank_share_probe.py W S grid_div [reps]   (GPU_MAX_HW_QUEUES from the environment)
W, S, gd = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
REPS = int(sys.argv[4]) if len(sys.argv) > 4 else 2
if True:
    if True:
      for rep in range(REPS):
            streams = [torch.cuda.Stream(dev) for _ in range(S)]
            lanes = []
            for st in streams:
                with torch.cuda.stream(st):
                    prs = [HipPartRenderer(s, h, w, dev, options={"grid_div": gd}) for s, h, w in frames]
                    outs = [torch.zeros((max_part_rows(h, W), w), dtype=torch.int32, device=dev) for _, h, w in frames]
                    lanes.append((prs, outs))
            torch.cuda.synchronize()
            def step(k):
                li = k % S
                with torch.cuda.stream(streams[li]):
                    prs, outs = lanes[li]
                    for pr, o in zip(prs, outs):
                        pr(W - 1, W, o)
            for k in range(2 * S): step(k)
            torch.cuda.synchronize()
            n = 1000
            t0 = time.perf_counter()
            for k in range(n): step(k)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            print(f"Q={os.environ['GPU_MAX_HW_QUEUES']} W={W} S={S} grid_div={gd}: issue {1e6*(t1-t0)/n:.1f} us/step, complete {1e6*(t2-t0)/n:.1f} us/step "
                  f"(ideal {570.0/W:.0f})", flush=True)
            del lanes, streams
