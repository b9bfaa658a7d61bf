"""How fast can ONE rank turn over its share of the bench's K steps at world size W?  One-GPU stand-in for the
strong-scaling run (batch protocol of bench.py): for every part p of W, the K steps' frames of each scene in C batch
launches (rt_render_batch) of this part's rows -- no exchange -- and, for irreg 4000x4000, one frame at a time.  The
slowest part is what a real W-GPU run would wait for.

usage: rank_share_probe.py [K=20] [worlds=1,2,4,8]"""
import sys, time
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from raytracers_amd.dist import HipPartRenderer, max_part_rows

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
worlds = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,2,4,8").split(",")]
CH = int(sys.argv[3]) if len(sys.argv) > 3 else 0       # chunks per scene (0: bench.py's default)
GD = int(sys.argv[4]) if len(sys.argv) > 4 else 0       # grid_div of the batch launches (0: library default)
TWO = len(sys.argv) > 5 and sys.argv[5] == "2s"         # the two scenes' launches on two streams
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
frames = [("rgbbox", 1000, 1000), ("irreg", 1000, 1000)]
streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)] if TWO else [torch.cuda.current_stream(dev)] * 2
prs = []
for (s_, h_, w_), st in zip(frames, streams):
    with torch.cuda.stream(st):
        prs.append(HipPartRenderer(s_, h_, w_, dev, options={"grid_div": GD} if GD else None))
big = HipPartRenderer("irreg", 4000, 4000, dev)
for W in worlds:
    C = CH or (1 if W == 1 else 2)
    sizes = [K // C + (1 if i < K % C else 0) for i in range(C)]
    res, res4 = [], []
    for p in range(W):
        outs = [[torch.zeros((nb * max_part_rows(h, W), w), dtype=torch.int32, device=dev) for _, h, w in frames] for nb in sizes]
        def run():
            for nb, o in zip(sizes, outs):
                for pr, (_, h, w), t, st in zip(prs, frames, o, streams):
                    with torch.cuda.stream(st):
                        pr.batch(p, W, nb, t, max_part_rows(h, W) * w)
        for _ in range(3): run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): run()
        torch.cuda.synchronize()
        res.append(1e6 * (time.perf_counter() - t0) / 5 / K)
        o4 = torch.zeros((max_part_rows(4000, W), 4000), dtype=torch.int32, device=dev)
        for _ in range(4):           # (render + sync: a view's deep-tile policy reaches the host asynchronously)
            big(p, W, o4)
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            big(p, W, o4)
            torch.cuda.synchronize()
        res4.append(1e6 * (time.perf_counter() - t0) / 5)
    print(f"W={W}: K={K} steps in {C} batch launch(es) per scene: {max(res):.1f} us/step slowest part, {min(res):.1f} fastest "
          f"(render-share scaling {res_1 / max(res):.2f}x); " if W > 1 else f"W=1: {max(res):.1f} us/step; ", end="")
    if W == 1: res_1, res4_1 = max(res), max(res4)
    print(f"irreg 4000x4000 one frame: {max(res4):.0f} us slowest part, {min(res4):.0f} fastest"
          + (f" ({res4_1 / max(res4):.2f}x)" if W > 1 else ""), flush=True)
