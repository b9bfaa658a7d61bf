#!/usr/bin/env python3
"""Randomised parity campaign: random scenes / cameras / image sizes, GPU (both BVH builders, all
kernel families) against the CPU oracle, bit-exact.
usage: fuzz_parity.py [seconds] [seed] [max image side = 160] [max spheres = 20000]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
import raytracers_amd as R

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
max_side = int(sys.argv[3]) if len(sys.argv) > 3 else 160
max_n = int(sys.argv[4]) if len(sys.argv) > 4 else 20000
ctx = R.Context()
t_end = time.time() + budget
cases = fails = culled_launches = borrowed_launches = 0
while time.time() < t_end:
    seed = seed0 + cases
    rng = np.random.default_rng(seed)
    n = int(np.exp(rng.uniform(np.log(2), np.log(3 * max_n if rng.random() < 0.1 else max_n))))
    kind = rng.choice(["uniform", "clustered", "grid", "line", "dupes", "shell"])
    s = np.zeros((n, 7), np.float32)
    ext = float(rng.choice([5.0, 40.0, 300.0, 3000.0]))
    if kind == "uniform":
        s[:, 0:3] = rng.uniform(-ext, ext, (n, 3))
    elif kind == "clustered":
        c = rng.uniform(-ext, ext, (max(1, n // 50), 3))
        s[:, 0:3] = c[rng.integers(0, len(c), n)] + rng.normal(0, ext / 40, (n, 3))
    elif kind == "grid":
        k = max(1, int(round(n ** (1 / 3))))
        g = np.stack(np.meshgrid(*[np.arange(k)] * 3, indexing="ij"), -1).reshape(-1, 3)[:n]
        s[:len(g), 0:3] = (g - k / 2) * (2 * ext / k)
        s[len(g):, 0:3] = rng.uniform(-ext, ext, (n - len(g), 3))
    elif kind == "line":
        s[:, int(rng.integers(0, 3))] = np.linspace(-ext, ext, n)
    elif kind == "dupes":
        base = rng.uniform(-ext, ext, (max(1, n // 7), 3))
        s[:, 0:3] = base[rng.integers(0, len(base), n)]
    else:
        v = rng.normal(0, 1, (n, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True) + 1e-9
        s[:, 0:3] = v * ext
    s[:, 3:6] = rng.uniform(0.1, 1.0, (n, 3))
    s[:, 6] = rng.uniform(0.02, 0.2) * ext * rng.uniform(0.2, 1.0, n) if rng.random() < 0.7 else ext * 0.05
    s = s.astype(np.float32)
    lf = tuple(float(x) for x in rng.uniform(-2 * ext, 2 * ext, 3))
    la = tuple(float(x) for x in rng.uniform(-ext / 4, ext / 4, 3))
    fov = float(rng.uniform(20, 100))
    h, w = int(rng.integers(1, max_side)), int(rng.integers(1, max_side))
    md = int(rng.choice([50, 50, 50, 1, 3, 7]))
    orc = O.OracleScene("custom", spheres7=s, look_from=lf, look_at=la, fov=fov)
    want_bvh = orc.arrays()
    ref, _ = orc.render(h, w, max_depth=md, threads=min(16, os.cpu_count() or 1))
    ok = True
    # random launch knobs (must never change pixels)
    knobs = dict(grid_div=int(rng.choice([0, 1, 2, 4, 8, 16])), thr_shade=int(rng.choice([1, 8, 24, 48, 64])),
                 deep_class=int(rng.integers(-1, 9)), adaptive_order=int(rng.choice([0, 1, 1, 2])),
                 lds_scene_bytes=int(rng.choice([-1, -1, 0, 2048, 20000])), waves_per_wg=int(rng.choice([0, 0, 4, 8, 12, 16])), wide_waves=int(rng.choice([0, 1, 2])), stack_cap=int(rng.choice([0, 0, 192])),
                 wgs_per_cu=int(rng.choice([1, 2, 4, 5])), ray_planes=int(rng.choice([0, 2, 3])), box2=int(rng.choice([0, 1, 1])),
                 deep_split=int(rng.choice([0, 1, 2, 3, 4, 5, 6, 6, 6])), deep_cap_log2=int(rng.integers(0, 6)), solo=int(rng.choice([0, 1, 1, 1])),
                 # the tile queue: one counter / a strip of tile columns per counter / counters taking turns, tiles per ticket,
                 # the waves' first tickets without an atomic
                 xcd_queues=int(rng.choice([-1, 0, 1, 2])), tpt_log2=int(rng.choice([-1, -1, 0, 1, 2, 3, 4])),
                 static_first=int(rng.choice([0, 1, 1])),
                 # the host builder's treelet cut (another cut than the shipped one switches the solo loop off)
                 treelet=int(rng.choice([4, 4, 4, 1, 2, 3])),
                 # the tails of single frames: 1 = DONATE for unordered frames (rays of waves that cannot refill go to waiting sibling
                 # waves) and COLD for small ordered ones (in-loop hand-over to the solo loop), 2 = DONATE for every single frame
                 handover=int(rng.choice([0, 1, 1, 2, 2])), donate_max=int(rng.choice([1, 4, 64])), look_max=int(rng.choice([0, 0, 1, 16, 32, 64])),
                 # pixel tickets (ordered single frames draw from the view's pixel list): off / where the library would / always;
                 # the chain lengths at which the list's classes (1, 8, 16, 32, 64 pixels per ticket) are cut, which classes hold their wave
                 first_order=int(rng.integers(0, 2)),     # a view's first frame: tile rows top to bottom / in bit-reversed order
                 pixel_order=int(rng.choice([0, 1, 2, 2, 2])), px_solo=int(rng.choice([0, 0, 0, 1, 2, 4, 24, 255])), px_w8=int(rng.choice([1, 3, 24, 255])),
                 px_w16=int(rng.choice([1, 2, 5, 14])), px_w32=int(rng.choice([1, 2, 3, 9])), px_hold=int(rng.integers(0, 32)), px_zip=int(rng.integers(0, 2)),
                 px_solo_div=int(rng.choice([1, 4, 64, 4096])),
                 # ... and the constants of the model that cuts the classes when px_solo is 0 (bounce cadences in 0.1 us, ns per ray)
                 px_g1=int(rng.choice([0, 1, 25, 1000])), px_g8=int(rng.choice([0, 3, 45])), px_g16=int(rng.choice([0, 5, 65])),
                 px_g32=int(rng.choice([0, 7, 100])), px_g64=int(rng.choice([0, 9, 160, 5000])), px_ray_ns=int(rng.choice([0, 1, 300, 20000])),
                 # culling by the best hit so far (the CULL instantiations): off / where the library would / wherever the proof's guards hold
                 cull=int(rng.choice([0, -1, 1, 1, 1])),
                 # a new view borrows the previous view's order / pixel list; the sorts of a recorded view on the second stream or in line
                 borrow=int(rng.choice([0, 1, 1, 1])), eager_sort=int(rng.choice([0, 1, 1])))
    for kv in os.environ.get("FUZZ_FORCE", "").split(","):     # e.g. FUZZ_FORCE=handover=2,donate_max=8: knobs pinned for an experiment
        if kv:
            knobs[kv.split("=")[0]] = int(kv.split("=")[1])
    for k, v in knobs.items():
        ctx.set_option(k, v)
    for gpu_build in (1, 0):
        ctx.set_option("gpu_build", gpu_build)
        ps = R.prepare_scene(h, w, ctx.scene_from_spheres(s, lf, la, fov))
        got = ps.bvh_arrays()
        for k in ("left", "right", "parent"):
            ok &= bool((got[k] == want_bvh[k]).all())
        for k in ("L", "bmin", "bmax"):
            ok &= got[k].tobytes() == want_bvh[k].tobytes()
        for variant in ((3, 1, 2) if gpu_build else (3,)):
            ctx.set_variant(variant)
            px = R.render(h, w, ps, max_depth=md)
            culled_launches += "+CULL" in ctx.last_launch
            px2 = R.render(h, w, ps, max_depth=md)      # second frame: adaptive tile order / deep tiles
            culled_launches += "+CULL" in ctx.last_launch
            ok &= int((px != ref).sum()) == 0 and int((px2 != ref).sum()) == 0
            if variant == 3:                            # third frame: the ticket counter after a frame with deep-tile pieces
                ok &= int((R.render(h, w, ps, max_depth=md) != ref).sum()) == 0
    # a second and a third camera on the same prepared scene: new views, rendered through the previous view's order / pixel list (borrow)
    ctx.set_variant(3)
    cam0 = np.asarray(ps.camera(), dtype=np.float32).reshape(12)
    for step in (1, 2):
        cam2 = cam0.copy()
        cam2[0:3] += np.float32(0.01 * ext * step); cam2[3:6] += np.float32(0.01 * ext * step)
        ref2, _ = orc.render(h, w, max_depth=md, threads=min(16, os.cpu_count() or 1), cam=cam2)
        for rep in range(2):
            got2 = R.render_image(ps, w, h, cam2, max_depth=md)
            borrowed_launches += "(borrowed)" in ctx.last_launch
            ok &= int((got2 != ref2).sum()) == 0
    # a batch of three frames in one launch (class-major tickets over the frames), after the view's order has settled
    ctx.set_variant(3)
    if h * w * 3 < (1 << 28):
        import torch
        nb = 3
        batch = torch.full((nb, h, w), -7, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        R.render_batch_into(batch.data_ptr(), h, w, ps, nb, frame_stride=h * w, max_depth=md)
        ctx.sync()
        ok &= all(int((f != ref).sum()) == 0 for f in batch.cpu().numpy())
    # the row-tile partition: every part rendered on its own, assembled in one launch
    ctx.set_option("gpu_build", 1)
    ctx.set_variant(0)
    nparts = int(rng.integers(1, 9))
    pad = max(R.part_rows(h, p, nparts) for p in range(nparts))
    if pad > 0:
        import torch
        stacked = torch.full((nparts, pad, w), -3, dtype=torch.int32, device="cuda")
        image = torch.full((h, w), -1, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        for p in range(nparts):
            if R.part_rows(h, p, nparts):
                R.render_into(stacked[p].data_ptr(), h, w, ps, max_depth=md, part=p, nparts=nparts)
        R.place_parts(ctx, h, w, nparts, pad, stacked.data_ptr(), image.data_ptr())
        ctx.sync()
        ok &= int((image.cpu().numpy() != ref).sum()) == 0
        # ... and stored in place (what a rank does into rank 0's image): no part buffers, no assembly
        image.fill_(-1)
        torch.cuda.synchronize()
        for p in range(nparts):
            R.render_inplace_into(image.data_ptr(), h, w, ps, max_depth=md, part=p, nparts=nparts)
        ctx.sync()
        ok &= int((image.cpu().numpy() != ref).sum()) == 0
    cases += 1
    if not ok:
        fails += 1
        print(f"MISMATCH seed {seed}: n={n} kind={kind} ext={ext} {w}x{h} max_depth={md} knobs={knobs} nparts={nparts}", flush=True)
ctx.set_option("gpu_build", 1)
print(f"fuzz: {cases} random cases, {fails} mismatches (seeds {seed0}..{seed0 + cases - 1}); {culled_launches} of the first / second frames ran a CULL instantiation, {borrowed_launches} frames of new views went through a borrowed order", flush=True)
sys.exit(1 if fails else 0)
