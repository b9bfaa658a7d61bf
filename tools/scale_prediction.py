#!/usr/bin/env python3
"""What the 1 / 2 / 4 / 8-GPU runs of bench.py should give, predicted on ONE MI355X -- so that a real SCALE_rNN.json can be
held against it.  Per world size W and workload:

  render   the slowest rank's share, MEASURED: part p of W of every frame through the very launches bench.py makes (one
           rt_render_batch launch per scene for the K = 20 steps, the scenes on two streams; irreg 4000x4000 one frame at a
           time, and its six frames as one batch launch), every part in turn on this GPU, no exchange;
  assemble rank 0's placement kernel for the gathered parts, MEASURED (rt_place_parts_batch on buffers of the real sizes);
  gather   MODELLED: rank 0 receives (W - 1) parts over (W - 1) xGMI links in parallel, so the wire time is one part's bytes
           / 153 GB/s (MI355X_MICROARCH.md: ~153 GB/s per link and direction) + a fixed 25 us for the collective's launch
           and rendezvous (the single-rank gather of tools/gather_probe.py costs that much without moving a byte far);
  step     render + gather + assemble (bench.py's bracket holds one gather per scene; the second scene's gather overlaps
           the first's assembly, not modelled: the figure is the conservative sum).
  direct   the direct-store exchange (round 4; what bench.py uses when it works): render_inplace = the slowest rank's share
           rendered IN PLACE into a full-size image (measured: the 4-byte pixel stores go where the assembled image has
           them), floored by the part's bytes over one link at HALF its rate (small scattered stores are not bulk
           copies), + two one-element all-reduces per launch at the same fixed 25 us each.  No gather, no assembly.

usage: scale_prediction.py [K=20] > profiles/r04/scale_prediction.json"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracers_amd as R
from raytracers_amd.dist import HipPartRenderer, max_part_rows

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
LINK_GBS, FIXED_US = 153.0, 25.0
STORE_EFF = 0.5      # direct stores: fraction of a link's bulk rate that 4-byte pixel stores (32-byte row segments at best) are assumed to reach
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
frames = [("rgbbox", 1000, 1000), ("irreg", 1000, 1000)]
RAYS = {"rgbbox": 4022099, "irreg": 1728608, "irreg4000": 27663974}
streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
prs = []
for (s_, h_, w_), st in zip(frames, streams):
    with torch.cuda.stream(st):
        prs.append(HipPartRenderer(s_, h_, w_, dev))
big = HipPartRenderer("irreg", 4000, 4000, dev)


def timed(fn, reps=5, warm=3, sync_each=False):
    for _ in range(warm + 1):        # (render + sync: a view's deep-tile policy reaches the host asynchronously)
        fn()
        torch.cuda.synchronize()
    groups = []
    for _ in range(3):               # (the median of three groups: one host hiccup in one part would otherwise be "the slowest rank")
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
            if sync_each:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        groups.append(1e6 * (time.perf_counter() - t0) / reps)
    return sorted(groups)[1]


def assemble_us(pr, h, w, W, nb):
    pad = max_part_rows(h, W)
    stacked = torch.zeros((W, nb * pad * w), dtype=torch.int32, device=dev)
    images = torch.empty((nb, h, w), dtype=torch.int32, device=dev)
    return timed(lambda: pr.place_batch(W, nb * pad * w, nb, pad * w, stacked, images), reps=10)


out = {"K": K, "model": {"link_GBs": LINK_GBS, "fixed_us_per_gather": FIXED_US, "note": __doc__.split("\n\n")[1]}, "worlds": {}}
base = {}
for W in (1, 2, 4, 8):
    rec = {}
    # ---- headline: K steps of rgbbox + irreg 1000x1000, one batch launch per scene
    shares = []
    for p in range(W):
        outs = [torch.zeros((K * max_part_rows(h, W), w), dtype=torch.int32, device=dev) for _, h, w in frames]

        def run():
            for pr, (_, h, w), t, st in zip(prs, frames, outs, streams):
                with torch.cuda.stream(st):
                    pr.batch(p, W, K, t, max_part_rows(h, W) * w)
        shares.append(timed(run))
    render = max(shares)
    if W == 1:
        gather = asm = 0.0
    else:
        part_bytes = sum(K * max_part_rows(h, W) * w * 4 for _, h, w in frames)
        gather = 2 * FIXED_US + part_bytes / (LINK_GBS * 1e3)
        asm = sum(assemble_us(pr, h, w, W, K) for pr, (_, h, w) in zip(prs, frames))
    step_us = (render + gather + asm) / K
    rec["headline_1000"] = {"render_us_slowest_rank": render, "render_us_fastest_rank": min(shares), "gather_us_model": gather,
                            "assemble_us": asm, "us_per_step": step_us, "Mray_s": (RAYS["rgbbox"] + RAYS["irreg"]) / step_us}
    # the same through the direct-store exchange: every part in place into full-size images
    dshares = []
    imgs = [torch.zeros((K, h, w), dtype=torch.int32, device=dev) for _, h, w in frames]
    for p in range(W):
        def run_d():
            for pr, (_, h, w), t, st in zip(prs, frames, imgs, streams):
                with torch.cuda.stream(st):
                    pr.inplace(p, W, K, t.data_ptr(), h * w)
        dshares.append(timed(run_d))
    d_render = max(dshares)
    d_floor = 0.0 if W == 1 else sum(K * max_part_rows(h, W) * w * 4 for _, h, w in frames) / (STORE_EFF * LINK_GBS * 1e3)
    d_sig = 0.0 if W == 1 else 2 * 2 * FIXED_US       # two launches (scenes), two signals each
    d_step = (max(d_render, d_floor) + d_sig) / K
    rec["headline_1000_direct"] = {"render_inplace_us_slowest_rank": d_render, "store_floor_us_model": d_floor, "signals_us_model": d_sig,
                                   "us_per_step": d_step, "Mray_s": (RAYS["rgbbox"] + RAYS["irreg"]) / d_step}
    del imgs
    # ---- irreg 4000x4000: one frame at a time, and six frames in one batch launch
    one, six = [], []
    for p in range(W):
        o4 = torch.zeros((max_part_rows(4000, W), 4000), dtype=torch.int32, device=dev)
        one.append(timed(lambda: big(p, W, o4), sync_each=True))
        o6 = torch.zeros((6 * max_part_rows(4000, W), 4000), dtype=torch.int32, device=dev)
        six.append(timed(lambda: big.batch(p, W, 6, o6, max_part_rows(4000, W) * 4000), reps=3, warm=2))
    pb = max_part_rows(4000, W) * 4000 * 4
    g1 = 0.0 if W == 1 else FIXED_US + pb / (LINK_GBS * 1e3)
    a1 = 0.0 if W == 1 else assemble_us(big, 4000, 4000, W, 1)
    g6 = 0.0 if W == 1 else FIXED_US + 6 * pb / (LINK_GBS * 1e3)
    a6 = 0.0 if W == 1 else assemble_us(big, 4000, 4000, W, 6)
    rec["irreg_4000_one_frame"] = {"render_us_slowest_rank": max(one), "render_us_fastest_rank": min(one), "gather_us_model": g1,
                                   "assemble_us": a1, "us_per_frame": max(one) + g1 + a1, "Mray_s": RAYS["irreg4000"] / (max(one) + g1 + a1)}
    rec["irreg_4000_batch_of_6"] = {"render_us_slowest_rank": max(six), "gather_us_model": g6, "assemble_us": a6,
                                    "us_per_frame": (max(six) + g6 + a6) / 6, "Mray_s": RAYS["irreg4000"] * 6 / (max(six) + g6 + a6)}
    # ... and through the direct-store exchange
    done, dsix = [], []
    i6 = torch.zeros((6, 4000, 4000), dtype=torch.int32, device=dev)
    for p in range(W):
        done.append(timed(lambda: big.inplace(p, W, 1, i6.data_ptr(), 16000000), sync_each=True))
        dsix.append(timed(lambda: big.inplace(p, W, 6, i6.data_ptr(), 16000000), reps=3, warm=2))
    del i6
    fl = 0.0 if W == 1 else pb / (STORE_EFF * LINK_GBS * 1e3)
    sg = 0.0 if W == 1 else 2 * FIXED_US
    rec["irreg_4000_one_frame_direct"] = {"render_inplace_us_slowest_rank": max(done), "store_floor_us_model": fl, "signals_us_model": sg,
                                          "us_per_frame": max(max(done), fl) + sg, "Mray_s": RAYS["irreg4000"] / (max(max(done), fl) + sg)}
    rec["irreg_4000_batch_of_6_direct"] = {"render_inplace_us_slowest_rank": max(dsix), "store_floor_us_model": 6 * fl, "signals_us_model": sg,
                                           "us_per_frame": (max(max(dsix), 6 * fl) + sg) / 6,
                                           "Mray_s": RAYS["irreg4000"] * 6 / (max(max(dsix), 6 * fl) + sg)}
    if W == 1:
        base = {k: (v.get("us_per_step") or v.get("us_per_frame")) for k, v in rec.items()}
    for k, v in rec.items():
        v["speedup_vs_1"] = base[k] / (v.get("us_per_step") or v.get("us_per_frame"))
    out["worlds"][str(W)] = rec
    print(f"W={W}: " + "; ".join(f"{k} {v.get('us_per_step') or v.get('us_per_frame'):.0f} us ({v['speedup_vs_1']:.2f}x)" for k, v in rec.items()),
          file=sys.stderr, flush=True)
print(json.dumps(out, indent=1))
