"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI,
against the CPU oracle on the same deterministic inputs.  Bar: BIT-EXACT packed pixels and
bit-exact BVH arrays -- there is no fp tolerance: the image is chaotic in the
last ulp, so any deviation in operation order shows up as differing pixels."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

VARIANTS = {"pixel": 1, "persistent": 2, "pooled": 3}


@pytest.fixture(scope="module")
def R():
    import raytracers_amd as R
    return R


@pytest.fixture(scope="module")
def ctx(R):
    c = R.Context()
    yield c
    c.close()


def _oracle(scene, **kw):
    if scene.startswith("floor:"):
        _, n, k = scene.split(":")
        return O.OracleScene("floor", n=int(n), k=float(k))
    return O.OracleScene(scene, **kw)


def _scene(ctx, scene):
    if scene.startswith("floor:"):
        _, n, k = scene.split(":")
        return ctx.floor(int(n), float(k))
    return ctx.scene(scene)


# ---------------------------------------------------------------- BVH + camera ------------
@pytest.mark.parametrize("gpu_build", [1, 0])
@pytest.mark.parametrize("scene", ["rgbbox", "irreg", "floor:37:222", "floor:2:12", "floor:300:1800",
                                   # the builder's size boundaries: the largest scene of the one-workgroup build (729 <= 768 spheres)
                                   # and the smallest of the ranked chain (784); ranked sizes of 1 / 2 / 12 / 32 keys per thread of
                                   # the ranking kernel; the largest ranked scene (24 336 <= 24 576) and the smallest with chained
                                   # sort passes (24 649); the largest with fused sweeps (131 044 <= 131 072) and the smallest without
                                   "floor:27:162", "floor:28:168", "floor:32:192", "floor:33:198", "floor:78:468", "floor:79:474",
                                   "floor:128:768", "floor:129:774", "floor:156:936", "floor:157:942", "floor:362:2172", "floor:363:2178"])
def test_bvh_arrays_bit_exact(R, ctx, scene, gpu_build):
    """prepare_scene's {L, I} (bvh.fut:28) from the GPU builder (bvh_build.hip) and from the
    host builder, both against the oracle: spheres, child pointers, parents and boxes bit-exact."""
    ctx.set_option("gpu_build", gpu_build)
    ps = R.prepare_scene(40, 56, _scene(ctx, scene))
    ctx.set_option("gpu_build", 1)
    got = ps.bvh_arrays()
    want = _oracle(scene).arrays()
    for k in ("left", "right", "parent"):
        assert (got[k] == want[k]).all(), k
    for k in ("L", "bmin", "bmax"):
        assert got[k].tobytes() == want[k].tobytes(), k


@pytest.mark.parametrize("h,w", [(200, 200), (500, 500), (1000, 1000), (37, 53), (1, 1), (4000, 4000), (480, 640)])
def test_camera_bit_exact(R, ctx, h, w):
    for scene in ("rgbbox", "irreg"):
        ps = R.prepare_scene(h, w, ctx.scene(scene))
        assert ps.camera().tobytes() == _oracle(scene).camera_floats(h, w).tobytes()


# ---------------------------------------------------------------- pixels ------------------
@pytest.mark.parametrize("variant", list(VARIANTS))
@pytest.mark.parametrize("scene", ["rgbbox", "irreg"])
def test_golden_500(R, ctx, scene, variant):
    """The reference's own known answers (rgbbox.png / irreg.png, decoded in tests/golden)."""
    ctx.set_variant(VARIANTS[variant])
    px = R.render(500, 500, R.prepare_scene(500, 500, ctx.scene(scene)))
    assert int((px != O.load_golden(scene)).sum()) == 0


@pytest.mark.parametrize("variant", list(VARIANTS))
@pytest.mark.parametrize("scene,h,w", [
    ("rgbbox", 200, 200), ("irreg", 200, 200),          # BASELINE configs[0] size
    ("rgbbox", 1000, 1000), ("irreg", 1000, 1000),      # BASELINE configs[1], [2]
    ("rgbbox", 37, 53), ("irreg", 53, 37),              # ragged: partial 8x8 tiles on both edges
    ("rgbbox", 1, 1), ("irreg", 1, 7), ("rgbbox", 9, 1),
    ("floor:37:222", 120, 90), ("floor:2:12", 64, 64),  # other tree shapes (2 spheres: one inner node)
])
def test_pixels_bit_exact(R, ctx, scene, h, w, variant):
    ctx.set_variant(VARIANTS[variant])
    got = R.render(h, w, R.prepare_scene(h, w, _scene(ctx, scene)))
    want, _ = _oracle(scene).render(h, w)
    assert got.shape == want.shape
    assert int((got != want).sum()) == 0


@pytest.mark.parametrize("variant", list(VARIANTS))
@pytest.mark.parametrize("max_depth", [0, 1, 2, 3, 50])
def test_bounce_limit(R, ctx, variant, max_depth):
    """configs[0] reads '1 bounce': max_depth 1 gives one ray per pixel (sky or black)."""
    ctx.set_variant(VARIANTS[variant])
    got = R.render(96, 120, R.prepare_scene(96, 120, ctx.rgbbox()), max_depth=max_depth)
    want, _ = _oracle("rgbbox").render(96, 120, max_depth=max_depth)
    assert int((got != want).sum()) == 0


def test_irreg_4000_strong_scaling_config(R, ctx):
    """BASELINE configs[3] size on one GPU: full-size parity against the oracle."""
    ctx.set_variant(0)
    got = R.render(4000, 4000, R.prepare_scene(4000, 4000, ctx.irreg()))
    want, _ = _oracle("irreg").render(4000, 4000)
    assert O.checksum(got) == O.checksum(want) == 0xDB269D43
    assert int((got != want).sum()) == 0


def test_big_scene_million_spheres(R, ctx):
    """BASELINE configs[4] scene (irreg generator at n=1000, k=6000 -> 10^6 spheres, tree
    height 20, BVH far larger than LDS): BVH arrays and a 512x512 render, bit-exact."""
    ctx.set_variant(0)
    ps = R.prepare_scene(512, 512, ctx.scene("big"))
    orc = _oracle("big")
    got, want = ps.bvh_arrays(), orc.arrays()
    for k in ("left", "right", "parent"):
        assert (got[k] == want[k]).all(), k
    for k in ("L", "bmin", "bmax"):
        assert got[k].tobytes() == want[k].tobytes(), k
    px = R.render(512, 512, ps)
    ref, _ = orc.render(512, 512)
    assert int((px != ref).sum()) == 0
    for v in (1, 2, 3):
        ctx.set_variant(v)
        assert int((R.render(512, 512, ps) != ref).sum()) == 0, v


def test_big_2000_at_size(R, ctx):
    """BASELINE configs[4] AT ITS STATED SIZE: 10^6 spheres, 2000x2000 (only the top of the 64 MB node array is LDS
    resident), every pixel against the oracle, checksum 3a198726 and the work counters of SURVEY.md 8d; also through
    a three-part multi-device context (configs[4] names 8 GPUs; one is present here)."""
    import bench
    ctx.set_variant(0)
    ps = R.prepare_scene(2000, 2000, ctx.scene("big"))
    px = R.render(2000, 2000, ps)
    ref, cnt = _oracle("big").render(2000, 2000)
    assert int((px != ref).sum()) == 0
    assert O.checksum(px) == bench.FRAME_CHECKSUM[("big", 2000, 2000)] == 0x3A198726
    st = ps.stats()
    assert (st["rays"], st["box_tests"], st["leaf_tests"]) == bench.FRAME_WORK[("big", 2000, 2000)] == \
        (cnt["rays"], cnt["box_tests"], cnt["leaf_tests"])
    ps.free()
    mc = R.Context(devices=[0, 0, 0])
    ps = R.prepare_scene(2000, 2000, mc.scene("big"))
    assert O.checksum(R.render(2000, 2000, ps)) == 0x3A198726
    ps.free()
    mc.close()


def _solo_cases():
    rng = np.random.default_rng(5)
    s = np.zeros((700, 7), np.float32)
    s[:, 0:3] = rng.uniform(-40, 40, (700, 3))
    s[:, 3:6] = rng.uniform(0.2, 1.0, (700, 3))
    s[:, 6] = rng.uniform(0.5, 6.0, 700)
    s[100:160, 0:3] = s[0:60, 0:3]          # coincident centres
    s[200:260] = s[300:360]                 # exact duplicates
    two = np.zeros((2, 7), np.float32)      # one inner node: a treelet of one
    two[:, 0] = (-4.0, 4.0); two[:, 3:6] = 0.8; two[:, 6] = 3.0
    pts = [(1023.0, 1023.0, 1023.0)] + [tuple(float(2 ** m) if a == k else 0.0 for k in range(3)) for a in range(3) for m in range(10)]
    tall = np.zeros((len(pts) + 64, 7), np.float32)   # a 30-level chain + 64 duplicates: height 36
    tall[:len(pts), 0:3] = np.array(pts, np.float32)
    tall[:, 3:6] = 0.7; tall[:, 6] = 0.4
    return [("rgbbox", None, 333, 250), ("irreg", None, 200, 280), ("floor:37:222", None, 90, 120), ("floor:300:1800", None, 120, 160),
            ("custom", (s, (5.0, 25.0, 70.0), (0.0, 0.0, 0.0), 60.0), 150, 200),
            ("custom", (two, (0.0, 3.0, 30.0), (0.0, 0.0, 0.0), 50.0), 72, 96),
            ("custom", (tall, (30.0, 20.0, 60.0), (0.0, 0.0, 0.0), 40.0), 90, 120)]


@pytest.mark.parametrize("opts", [dict(), dict(deep_class=8, deep_split=6, deep_cap_log2=0), dict(deep_class=8, deep_split=6, deep_cap_log2=0, grid_div=16),
                                  dict(deep_class=5, deep_split=6, deep_cap_log2=2, xcd_queues=1), dict(deep_class=8, deep_split=6, deep_cap_log2=0, static_first=0),
                                  dict(deep_class=8, deep_split=6, deep_cap_log2=0, solo=0), dict(deep_class=4, deep_split=5, deep_cap_log2=1)])
@pytest.mark.parametrize("gpu_build", [1, 0])
def test_solo_pixels_and_treelet_numbering(R, opts, gpu_build):
    """Single-pixel tickets traced by the solo loop (render_kernels.hip: solo_trace -- treelet operations on the traversal
    copy numbered treelet by treelet, by both builders): the view's chosen policy (default), every recorded tile handed out
    pixel by pixel (more tickets than waves with grid_div=16: the solo prologue also draws from the counters), strips, no
    static first tickets, the same tickets through the pooled loop (solo=0).  Frames of a view (recording, ordered),
    a part of three and a batch, buffers poisoned: reference scenes, 90 000 spheres (the multi-kernel builder), duplicates,
    a single inner node, a 36-level tree."""
    import torch
    from raytracers_amd.dist import tile_rows
    c = R.Context()
    c.set_variant(3)
    c.set_option("gpu_build", gpu_build)
    for k, v in opts.items():
        c.set_option(k, v)
    for name, custom, h, w in _solo_cases():
        if custom is None:
            orc, sc = _oracle(name), _scene(c, name)
        else:
            orc = O.OracleScene("custom", spheres7=custom[0], look_from=custom[1], look_at=custom[2], fov=custom[3])
            sc = c.scene_from_spheres(*custom)
        want, _ = orc.render(h, w)
        ps = R.prepare_scene(h, w, sc)
        out = torch.empty((h, w), dtype=torch.int32, device="cuda")
        for frame in range(4):
            out.fill_(-1)
            torch.cuda.synchronize()
            R.render_into(out.data_ptr(), h, w, ps)
            c.sync()
            assert int((out.cpu().numpy() != want).sum()) == 0, (name, frame)
        part = torch.empty((R.part_rows(h, 1, 3), w), dtype=torch.int32, device="cuda")
        for frame in range(3):
            part.fill_(-3)
            torch.cuda.synchronize()
            R.render_into(part.data_ptr(), h, w, ps, part=1, nparts=3)
            c.sync()
            assert int((part.cpu().numpy() != want[tile_rows(h, 1, 3)]).sum()) == 0, (name, "part", frame)
        buf = torch.full((3, h, w), -7, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        R.render_batch_into(buf.data_ptr(), h, w, ps, 3, frame_stride=h * w)
        c.sync()
        assert all(int((f != want).sum()) == 0 for f in buf.cpu().numpy()), (name, "batch")
        ps.free()
    c.close()


@pytest.mark.parametrize("opts", [dict(pixel_order=2), dict(pixel_order=1), dict(pixel_order=0),
                                  dict(pixel_order=2, px_solo=1, px_w8=1, px_w16=1, px_w32=1),          # every pixel a one-pixel ticket (capped: then 8s)
                                  dict(pixel_order=2, px_solo=255, px_w8=255, px_w16=255, px_w32=255),  # no classes: the plain sorted list
                                  dict(pixel_order=2, px_solo=4, px_w8=3, px_w16=2, px_w32=2, px_hold=0, px_prio=0),
                                  dict(pixel_order=2, px_solo=255, px_w8=2, px_w16=2, px_w32=1, px_hold=31, xcd_queues=0, static_first=0),
                                  dict(pixel_order=2, solo=0), dict(pixel_order=2, grid_div=16, px_solo_div=1),
                                  dict(pixel_order=2, px_g1=1, px_g8=2, px_g16=3, px_g32=4, px_g64=5, px_ray_ns=1),
                                  dict(pixel_order=2, px_g1=1000, px_g64=20000, px_ray_ns=20000, thr_shade=8),
                                  dict(pixel_order=2, gpu_build=0, treelet=2), dict(pixel_order=2, xcd_queues=1)])
def test_pixel_tickets(R, opts):
    """Ordered single frames that draw their tickets from the view's PIXEL LIST (the first frame records every pixel's
    chain length, the sorts run ahead of the second frame, the ORD instantiation renders from it): the list's classes cut by the device's
    model, by hand at both extremes, with and without holding / the solo loop / static first tickets, on one counter, with strips (no list:
    the tile tickets), by the host builder with another treelet cut (no solo loop).  Frames 1 .. 4 of a view into a poisoned buffer, a part
    of three packed and then IN PLACE (the same view: the list recorded by the packed frame serves the in-place one), a view first recorded
    by a BATCH (no per-pixel record: its first single frame records again), and the batch entry after the list exists.  Pixels: the oracle's."""
    import torch
    from raytracers_amd.dist import tile_rows
    c = R.Context()
    c.set_variant(3)
    for k, v in opts.items():
        c.set_option(k, v)
    for name, custom, h, w in _solo_cases()[:5] + [("rgbbox", None, 8, 8), ("irreg", None, 1, 77)]:
        if custom is None:
            orc, sc = _oracle(name), _scene(c, name)
        else:
            orc = O.OracleScene("custom", spheres7=custom[0], look_from=custom[1], look_at=custom[2], fov=custom[3])
            sc = c.scene_from_spheres(*custom)
        want, _ = orc.render(h, w)
        ps = R.prepare_scene(h, w, sc)
        out = torch.empty((h, w), dtype=torch.int32, device="cuda")
        for frame in range(4):
            out.fill_(-1)
            torch.cuda.synchronize()
            R.render_into(out.data_ptr(), h, w, ps)
            c.sync()
            assert int((out.cpu().numpy() != want).sum()) == 0, (name, frame)
        if name in ("rgbbox", "irreg") and h * w > 64 and "xcd_queues" not in opts:
            # (which kernel rendered the view's fourth frame: rt_context_last_launch)
            if opts.get("pixel_order") == 2:
                assert "tickets=pixel-list" in c.last_launch and "instantiation=ORD" in c.last_launch, c.last_launch
            elif opts.get("pixel_order") == 0:
                assert "tickets=tiles-ordered" in c.last_launch, c.last_launch
        rows = R.part_rows(h, 1, 3)
        if rows:
            part = torch.empty((rows, w), dtype=torch.int32, device="cuda")
            for frame in range(3):
                part.fill_(-3)
                torch.cuda.synchronize()
                R.render_into(part.data_ptr(), h, w, ps, part=1, nparts=3)
                c.sync()
                assert int((part.cpu().numpy() != want[tile_rows(h, 1, 3)]).sum()) == 0, (name, "part", frame)
            image = torch.full((h, w), -5, dtype=torch.int32, device="cuda")
            torch.cuda.synchronize()
            for frame in range(2):
                R.render_inplace_into(image.data_ptr(), h, w, ps, part=1, nparts=3)
            c.sync()
            got = image.cpu().numpy()
            mine = np.zeros(h, bool)
            mine[tile_rows(h, 1, 3)] = True
            assert int((got[mine] != want[mine]).sum()) == 0 and bool((got[~mine] == -5).all()), (name, "in place")
        ps.free()
        # a view whose tiles a batch recorded first
        ps = R.prepare_scene(h, w, sc)
        buf = torch.full((3, h, w), -7, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        R.render_batch_into(buf.data_ptr(), h, w, ps, 3, frame_stride=h * w)
        c.sync()
        assert all(int((f != want).sum()) == 0 for f in buf.cpu().numpy()), (name, "batch first")
        for frame in range(3):
            out.fill_(-1)
            torch.cuda.synchronize()
            R.render_into(out.data_ptr(), h, w, ps)
            c.sync()
            assert int((out.cpu().numpy() != want).sum()) == 0, (name, "after a batch", frame)
        buf.fill_(-7)
        torch.cuda.synchronize()
        R.render_batch_into(buf.data_ptr(), h, w, ps, 3, frame_stride=h * w)
        c.sync()
        assert all(int((f != want).sum()) == 0 for f in buf.cpu().numpy()), (name, "batch after the list")
        ps.free()
    c.close()


@pytest.mark.parametrize("opts", [dict(stack_cap=192), dict(stack_cap=192, cull=0, look_max=64, thr_shade=1), dict(stack_cap=192, box2=0, cull=1), dict(stack_cap=192, waves_per_wg=4, wgs_per_cu=5),
                                  dict(stack_cap=0)])
def test_box_stack_spills_to_memory(R, opts):
    """The shape of twenty waves per CU caps a wave's LDS box stack at 1 088 dwords; trees taller than 15 levels have a larger bound (64 H + 63), and
    the four-wave kernels then keep the oldest half of a stack that would overflow in device memory (render_kernels.hip: SPILL; KParams::spill) and
    fetch it back when the LDS part has run empty.  stack_cap forces tiny capacities, so the path runs thousands of times per frame: single
    frames of a view (recording, ordered), a part, a batch -- the reference scenes, 90 000 spheres, a 36-level tree -- all bit-exact."""
    import torch
    from raytracers_amd.dist import tile_rows
    c = R.Context()
    c.set_variant(3)
    c.set_option("wide_waves", 2)
    for k, v in opts.items():
        c.set_option(k, v)
    cases = [cs for cs in _solo_cases() if cs[0] != "floor:37:222"] + [("irreg", None, 500, 500)]
    for name, custom, h, w in cases:
        if custom is None:
            orc, sc = _oracle(name), _scene(c, name)
        else:
            orc = O.OracleScene("custom", spheres7=custom[0], look_from=custom[1], look_at=custom[2], fov=custom[3])
            sc = c.scene_from_spheres(*custom)
        want, _ = orc.render(h, w)
        ps = R.prepare_scene(h, w, sc)
        out = torch.empty((h, w), dtype=torch.int32, device="cuda")
        for frame in range(3):
            out.fill_(-1)
            torch.cuda.synchronize()
            R.render_into(out.data_ptr(), h, w, ps)
            c.sync()
            assert int((out.cpu().numpy() != want).sum()) == 0, (name, frame, c.last_launch)
            if name in ("irreg", "floor:300:1800"):
                assert "waves=4" in c.last_launch, (name, c.last_launch)     # (scenes read from L2, whatever their height: the twenty-wave shape)
                # (irreg's 15 levels fit the LDS stack unless stack_cap says otherwise; 90 000 spheres are taller)
                assert ("+SPILL" in c.last_launch) == (opts["stack_cap"] != 0 or name == "floor:300:1800"), (name, c.last_launch)
        rows = R.part_rows(h, 1, 3)
        part = torch.full((rows, w), -3, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        R.render_into(part.data_ptr(), h, w, ps, part=1, nparts=3)
        c.sync()
        assert int((part.cpu().numpy() != want[tile_rows(h, 1, 3)]).sum()) == 0, (name, "part")
        buf = torch.full((3, h, w), -7, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        R.render_batch_into(buf.data_ptr(), h, w, ps, 3, frame_stride=h * w)
        c.sync()
        assert all(int((f != want).sum()) == 0 for f in buf.cpu().numpy()), (name, "batch", c.last_launch)
        ps.free()
    c.close()


@pytest.mark.parametrize("opts", [dict(cull=1), dict(cull=-1), dict(cull=0), dict(cull=1, pixel_order=2, look_max=1, thr_shade=64), dict(cull=1, pixel_order=0, deep_class=8, deep_split=6, deep_cap_log2=0),
                                  dict(cull=1, handover=2, donate_max=8), dict(cull=1, box2=0, thr_shade=8, solo=0), dict(cull=1, lds_scene_bytes=0, gpu_build=0),
                                  dict(cull=1, wide_waves=2), dict(cull=0, wide_waves=2, look_max=64, thr_shade=1), dict(cull=-1, waves_per_wg=4, wgs_per_cu=5)])
def test_cull_by_best_hit(R, opts):
    """The CULL instantiations of the pooled kernel (lane_core.h: cull_limit): boxes are tested against the slot's best
    root so far, widened by a proven margin, instead of the reference's fixed 1e9 (ray.fut:77) -- fewer tests, the SAME fold result
    (ray.fut:76-86).  Every flavour (plain, SOLO, COLD, DONATE, ORD; BOX, BOX2 and the solo loop's treelet operation), frames 1 .. 4 of a
    view, a part packed and in place, batches with and without their own cameras; scenes: the reference's, floors, random spheres with
    radii 0.5 .. 6 and coincident centres, a single inner node.  Where the proof's guards do NOT hold the library must fall back by
    itself: a 36-level tree (unconverged boxes: height > sweeps), a camera 10^6 radii away; rt_context_last_launch says which ran."""
    import torch
    from raytracers_amd.dist import tile_rows
    c = R.Context()
    c.set_variant(3)
    for k, v in opts.items():
        c.set_option(k, v)
    cases = _solo_cases() + [("irreg", None, 500, 500), ("floor:12:72", None, 64, 64)]
    for name, custom, h, w in cases:
        if custom is None:
            orc, sc = _oracle(name), _scene(c, name)
        else:
            orc = O.OracleScene("custom", spheres7=custom[0], look_from=custom[1], look_at=custom[2], fov=custom[3])
            sc = c.scene_from_spheres(*custom)
        want, _ = orc.render(h, w)
        ps = R.prepare_scene(h, w, sc)
        tall = custom is not None and len(custom[0]) == 95       # the 36-level tree: never culled
        whole_lds = name in ("rgbbox", "floor:37:222", "floor:12:72") or (custom is not None and len(custom[0]) <= 700)
        out = torch.empty((h, w), dtype=torch.int32, device="cuda")
        for frame in range(4):
            out.fill_(-1)
            torch.cuda.synchronize()
            R.render_into(out.data_ptr(), h, w, ps)
            c.sync()
            assert int((out.cpu().numpy() != want).sum()) == 0, (name, frame, c.last_launch)
            culled = "+CULL" in c.last_launch
            if opts["cull"] == 0 or tall or name == "rgbbox":    # (rgbbox: height 14 > 10 sweeps)
                assert not culled, (name, c.last_launch)
            elif opts["cull"] == 1 and ("waves=16" in c.last_launch or "waves=4" in c.last_launch):
                assert culled, (name, c.last_launch)
            elif opts["cull"] == -1 and "waves=16" in c.last_launch and "lds_scene_bytes" not in opts:
                assert culled == (not whole_lds), (name, c.last_launch)
            # (wide_waves = 2 / the shape configured: five workgroups of four waves per CU for every launch of a scene that is read from L2 and
            # whose tree is at most 15 levels tall -- irreg's is exactly that; scenes that fit in LDS and taller trees keep 16 waves)
            if name == "irreg" and (opts.get("wide_waves") == 2 or opts.get("wgs_per_cu") == 5):
                assert "waves=4" in c.last_launch and ("grid=1280" in c.last_launch or "grid=640" in c.last_launch), c.last_launch
            if (name == "rgbbox" or tall) and opts.get("wide_waves") == 2:
                assert "grid=1280" not in c.last_launch and "grid=640" not in c.last_launch, c.last_launch
        rows = R.part_rows(h, 1, 3)
        part = torch.empty((rows, w), dtype=torch.int32, device="cuda")
        for frame in range(3):
            part.fill_(-3)
            torch.cuda.synchronize()
            R.render_into(part.data_ptr(), h, w, ps, part=1, nparts=3)
            c.sync()
            assert int((part.cpu().numpy() != want[tile_rows(h, 1, 3)]).sum()) == 0, (name, "part", frame)
        image = torch.full((h, w), -5, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        for frame in range(2):
            R.render_inplace_into(image.data_ptr(), h, w, ps, part=1, nparts=3)
        c.sync()
        got = image.cpu().numpy()
        mine = np.zeros(h, bool)
        mine[tile_rows(h, 1, 3)] = True
        assert int((got[mine] != want[mine]).sum()) == 0 and bool((got[~mine] == -5).all()), (name, "in place")
        buf = torch.full((3, h, w), -7, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        R.render_batch_into(buf.data_ptr(), h, w, ps, 3, frame_stride=h * w)
        c.sync()
        assert all(int((f != want).sum()) == 0 for f in buf.cpu().numpy()), (name, "batch")
        if opts["cull"] == 1 and not tall and name != "rgbbox" and ("waves=16" in c.last_launch or "waves=4" in c.last_launch):
            assert "+CULL" in c.last_launch, (name, c.last_launch)
        # a batch with its own cameras: the prepared one, one a little off, and one far outside the scene guard (the whole batch then
        # renders un-culled: every origin must pass)
        cam = ps.camera()
        near = cam.copy(); near[0:3] += np.float32(0.5); near[3:6] += np.float32(0.5)     # origin and lower-left corner move together
        far = cam.copy(); far[0:3] += np.float32(4.0e6); far[3:6] += np.float32(4.0e6)
        for cams, label in ((np.stack([cam, near, cam]), "cams near"), (np.stack([cam, far, near]), "cams far")):
            buf.fill_(-7)
            torch.cuda.synchronize()
            R.render_batch_into(buf.data_ptr(), h, w, ps, 3, frame_stride=h * w, cams=cams)
            c.sync()
            ll = c.last_launch
            frames = buf.cpu().numpy()
            for f in range(3):
                wf, _ = orc.render(h, w, cam=cams[f])
                assert int((frames[f] != wf).sum()) == 0, (name, label, f, ll)
            if label == "cams far":
                assert "+CULL" not in ll, (name, ll)
        ps.free()
    c.close()


@pytest.mark.parametrize("opts", [dict(), dict(handover=0), dict(thr_shade=8), dict(gpu_build=0), dict(box2=0), dict(static_first=0),
                                  dict(xcd_queues=0, thr_shade=64), dict(handover=2, donate_max=1), dict(handover=2, donate_max=64),
                                  dict(handover=2, donate_max=8, thr_shade=8), dict(handover=2, donate_max=64, grid_div=4)])
def test_first_frames_of_new_views(R, opts):
    """A view's FIRST frame (no tile order yet) runs on every workgroup, through the DONATE instantiation (a wave that cannot
    refill gives its rays to sibling waves of its workgroup that have left the loop; each walks its ray in the solo loop);
    small ORDERED frames through the COLD one (a wave's last rays go to the solo loop from INSIDE the pooled loop);
    handover=2 donates in every single frame.  The pixels are the oracle's, on both scenes and a random one, at sizes on
    either side of the range COLD is used for; every first frame is a new view (fresh prepared scenes and a camera path),
    into a poisoned buffer."""
    import bench
    import torch
    c = R.Context()
    for k, v in opts.items():
        c.set_option(k, v)
    cks = bench.Checksummer(torch.device("cuda"))
    for scene, h, w in (("irreg", 1000, 1000), ("rgbbox", 700, 700), ("irreg", 500, 500), ("rgbbox", 360, 360), ("rgbbox", 500, 500)):
        ps = R.prepare_scene(h, w, c.scene(scene))
        img = torch.full((h, w), 0x5a5a5a5a, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        R.render_into(img.data_ptr(), h, w, ps)
        c.sync()
        if (scene, h, w) in bench.FRAME_CHECKSUM:
            assert cks(img) == bench.FRAME_CHECKSUM[(scene, h, w)], (scene, h, w, opts)
        else:
            want, _ = _oracle(scene).render(h, w)
            assert int((img.cpu().numpy() != want).sum()) == 0, (scene, h, w, opts)
        first = img.clone()
        for _ in range(2):                     # the view's second and third frame (exact order, then the policy)
            img.fill_(0x5a5a5a5a)
            torch.cuda.synchronize()      # (the fill runs on torch's stream, the render on the context's own: order them)
            R.render_into(img.data_ptr(), h, w, ps)
            c.sync()
            assert bool((img == first).all())
        # a camera path, every view new
        if h == 500:
            orc = _oracle(scene)
            for f in range(4):
                cam = orc.camera_floats(h + 8 * f, w)
                img.fill_(0x5a5a5a5a)
                torch.cuda.synchronize()      # (the fill runs on torch's stream, the render on the context's own: order them)
                R.render_into(img.data_ptr(), h, w, ps, cam=cam)
                c.sync()
                one = img.clone()
                c.set_option("handover", 0)
                c.set_option("adaptive_order", 0)
                img.fill_(0x5a5a5a5a)
                torch.cuda.synchronize()      # (the fill runs on torch's stream, the render on the context's own: order them)
                R.render_into(img.data_ptr(), h, w, ps, cam=cam)
                c.sync()
                c.set_option("adaptive_order", 1)
                c.set_option("handover", opts.get("handover", 1))
                assert bool((img == one).all()), (scene, f)
        ps.free()
    rng = np.random.default_rng(11)
    s = np.zeros((3000, 7), np.float32)
    s[:, 0:3] = rng.uniform(-60, 60, (3000, 3)); s[:, 1] *= 0.1
    s[:, 3:6] = rng.uniform(0.3, 1.0, (3000, 3)); s[:, 6] = rng.uniform(0.5, 3.0, 3000)
    sc = (s, (5.0, 30.0, 90.0), (0.0, 0.0, 0.0), 55.0)
    ps = R.prepare_scene(640, 640, c.scene_from_spheres(*sc))
    want, _ = O.OracleScene("custom", spheres7=s, look_from=sc[1], look_at=sc[2], fov=sc[3]).render(640, 640)
    assert int((R.render(640, 640, ps) != want).sum()) == 0
    ps.free()
    c.close()


@pytest.mark.parametrize("treelet", [1, 2, 3, 5])
def test_host_builder_other_treelet_cuts(R, treelet):
    """The host builder's numbering under other cuts than the shipped one (treelet.h; the solo loop is off then): the
    pooled loop does not care how the traversal copy is numbered."""
    c = R.Context()
    c.set_variant(3)
    c.set_option("gpu_build", 0)
    c.set_option("treelet", treelet)
    for name in ("rgbbox", "irreg", "floor:37:222"):
        want, _ = _oracle(name).render(96, 120)
        ps = R.prepare_scene(96, 120, _scene(c, name))
        for frame in range(2):
            assert int((R.render(96, 120, ps) != want).sum()) == 0, (name, frame)
        ps.free()
    c.close()


@pytest.mark.parametrize("variant", [1, 3])
def test_host_built_scene_renders_identically(R, variant):
    """The two builders number the traversal copy differently (breadth-first vs by depth); pixels
    must not care."""
    c = R.Context()
    c.set_variant(variant)
    for scene in ("rgbbox", "irreg"):
        want, _ = _oracle(scene).render(150, 170)
        for gpu_build in (0, 1):
            c.set_option("gpu_build", gpu_build)
            got = R.render(150, 170, R.prepare_scene(150, 170, c.scene(scene)))
            assert int((got != want).sum()) == 0, (scene, gpu_build)
    c.close()


def test_random_scene_with_duplicates_and_explicit_camera(R, ctx):
    """Arbitrary spheres through rt_scene_from_spheres, including exact duplicates (equal
    Morton keys -> index tie-break in the radix tree; coincident spheres -> lowest leaf index
    wins) and render_image with an explicit camera."""
    rng = np.random.default_rng(1234)
    n = 600
    s = np.zeros((n, 7), np.float32)
    s[:, 0:3] = rng.uniform(-40, 40, (n, 3))
    s[:, 3:6] = rng.uniform(0.2, 1.0, (n, 3))
    s[:, 6] = rng.uniform(0.5, 4.0, n)
    s[100:140, 0:3] = s[0:40, 0:3]          # coincident centres, different colours/radii
    s[140:150] = s[40:50]                   # exact duplicates
    lf, la, fov = (5.0, 25.0, 70.0), (0.0, 0.0, 0.0), 60.0
    orc = O.OracleScene("custom", spheres7=s, look_from=lf, look_at=la, fov=fov)
    for variant in (1, 2, 3):
        ctx.set_variant(variant)
        ps = R.prepare_scene(150, 200, ctx.scene_from_spheres(s, lf, la, fov))
        got, want = ps.bvh_arrays(), orc.arrays()
        for k in ("left", "right", "parent"):
            assert (got[k] == want[k]).all(), k
        for k in ("L", "bmin", "bmax"):
            assert got[k].tobytes() == want[k].tobytes(), k
        px = R.render(150, 200, ps)
        ref, _ = orc.render(150, 200)
        assert int((px != ref).sum()) == 0
        # explicit camera: the prepared camera of ANOTHER size, passed by hand
        cam = orc.camera_floats(90, 160)
        px2 = R.render_image(ps, 160, 90, cam)
        ref2, _ = orc.render(90, 160)
        assert int((px2 != ref2).sum()) == 0


@pytest.mark.parametrize("n,kind", [(2, "apart"), (3, "apart"), (2, "same"), (5, "same"), (64, "same"), (65, "line"),
                                    (2000, "same"), (3000, "line"), (20000, "same")])
def test_tiny_and_degenerate_scenes(R, ctx, n, kind):
    """The smallest trees (one or two inner nodes), all spheres identical (every Morton key equal:
    the radix tree falls back to the index tie-break at every level), and a flat line of spheres
    (two degenerate Morton axes: 0/0 = NaN quantised to 0).  The sizes in the thousands go through
    the ranked chain: all keys equal is ONE bucket holding everything, the ranking kernel's slowest case."""
    s = np.zeros((n, 7), np.float32)
    s[:, 3:6] = np.linspace(0.3, 1.0, 3 * n, dtype=np.float32).reshape(n, 3)
    s[:, 6] = 2.0
    if kind == "apart":
        s[:, 0] = np.arange(n, dtype=np.float32) * 7.0 - 7.0
    elif kind == "line":
        s[:, 0] = np.arange(n, dtype=np.float32) * 1.5 - 40.0
    lf, la, fov = (0.0, 3.0, 30.0), (0.0, 0.0, 0.0), 50.0
    h, w = (72, 96) if n <= 100 else (24, 32)      # (thousands of spheres in one place: every ray that hits tests them all)
    orc = O.OracleScene("custom", spheres7=s, look_from=lf, look_at=la, fov=fov)
    for gpu_build in (1, 0):
        ctx.set_option("gpu_build", gpu_build)
        ps = R.prepare_scene(h, w, ctx.scene_from_spheres(s, lf, la, fov))
        got, want = ps.bvh_arrays(), orc.arrays()
        for k in ("left", "right", "parent"):
            assert (got[k] == want[k]).all(), (k, gpu_build)
        for k in ("L", "bmin", "bmax"):
            assert got[k].tobytes() == want[k].tobytes(), (k, gpu_build)
        ref, _ = orc.render(h, w)
        for variant in (1, 2, 3):
            ctx.set_variant(variant)
            assert int((R.render(h, w, ps) != ref).sum()) == 0, (variant, gpu_build)
    ctx.set_option("gpu_build", 1)
    ctx.set_variant(0)


@pytest.mark.parametrize("dupes,want_height", [(1, 30), (64, 36), (1100, 41), (5000, 43)])
def test_tall_trees(R, ctx, dupes, want_height):
    """Single-bit Morton codes make a 30-level chain, exact duplicates at the origin add log2(dupes)
    levels on top (index tie-break): the pooled kernel's per-wave box stack (64 * (height + 3)
    entries) no longer fits two 8-wave workgroups per CU, so the launch plan must fall back to a
    smaller workgroup shape (or, for AUTO, to the pixel kernel) -- never fail, never differ."""
    pts = [(1023.0, 1023.0, 1023.0)]
    for a in range(3):
        for m in range(10):
            p = [0.0, 0.0, 0.0]
            p[a] = float(2 ** m)
            pts.append(tuple(p))
    pts += [(0.0, 0.0, 0.0)] * dupes
    s = np.zeros((len(pts), 7), np.float32)
    s[:, 0:3] = np.array(pts, np.float32)
    s[:, 3:6] = np.linspace(0.3, 1.0, 3 * len(pts), dtype=np.float32).reshape(-1, 3)
    s[:, 6] = 0.4
    lf, la, fov = (30.0, 20.0, 60.0), (0.0, 0.0, 0.0), 40.0
    orc = O.OracleScene("custom", spheres7=s, look_from=lf, look_at=la, fov=fov)
    ps = R.prepare_scene(90, 120, ctx.scene_from_spheres(s, lf, la, fov))
    got, want = ps.bvh_arrays(), orc.arrays()
    for k in ("left", "right", "parent"):
        assert (got[k] == want[k]).all(), k
    for k in ("L", "bmin", "bmax"):
        assert got[k].tobytes() == want[k].tobytes(), k
    assert ps.height == want_height
    ref, _ = orc.render(90, 120)
    for variant in (0, 1, 2, 3):
        ctx.set_variant(variant)
        assert int((R.render(90, 120, ps) != ref).sum()) == 0, variant
    ctx.set_variant(0)


def test_many_views_of_one_prepared_scene(R, ctx):
    """More views (size x camera) of one prepared scene than its tile-order cache holds (8): the
    oldest records are evicted and re-recorded, every frame stays bit-exact."""
    ps = R.prepare_scene(64, 64, ctx.scene("rgbbox"))
    orc = _oracle("rgbbox")
    sizes = [(40 + 7 * i, 52 + 5 * i) for i in range(11)]
    refs = {}
    for rnd in range(3):
        for h, w in sizes:
            if (h, w) not in refs:
                refs[(h, w)] = orc.render(h, w)[0]
            px = R.render_image(ps, w, h, orc.camera_floats(h, w))
            assert int((px != refs[(h, w)]).sum()) == 0, (rnd, h, w)


@pytest.mark.parametrize("scene,h,w", [("irreg", 200, 280), ("rgbbox", 333, 250)])
def test_camera_path_one_frame_at_a_time(R, scene, h, w):
    """A camera path rendered frame by frame: every view is new (its first frame records its tile costs), and from the ninth
    view on a new view takes over the least recently used view's buffers as they are (no synchronisation).  Each frame
    against the same camera rendered without any order."""
    plain = R.Context()
    plain.set_variant(3)
    plain.set_option("adaptive_order", 0)
    c = R.Context()
    c.set_variant(3)
    ps0, ps = R.prepare_scene(h, w, plain.scene(scene)), R.prepare_scene(h, w, c.scene(scene))
    base = np.asarray(ps.camera(), dtype=np.float32).reshape(12)
    cams = []
    for f in range(13):
        cam = base.copy()
        cam[0] += 0.4 * f; cam[3] += 0.4 * f          # origin and lower-left corner move together: the camera slides
        cams.append(cam)
    want = [R.render_image(ps0, w, h, cam) for cam in cams]
    assert int((want[0] != _oracle(scene).render(h, w)[0]).sum()) == 0
    for rnd in range(3):                               # 13 views > 8 kept: borrowed first frames, evictions, revisits
        for f, cam in enumerate(cams):
            assert int((R.render_image(ps, w, h, cam) != want[f]).sum()) == 0, (rnd, f)
        for f in (3, 3, 7):                            # the same view again: its own order, the policy's read-back
            assert int((R.render_image(ps, w, h, cams[f]) != want[f]).sum()) == 0, (rnd, "again", f)
    ps.free(); ps0.free()
    c.close(); plain.close()


@pytest.mark.parametrize("opts", [dict(), dict(borrow=3), dict(borrow=2, cull=1), dict(borrow=4, handover=0), dict(sync_policy=0), dict(eager_sort=0), dict(borrow=0, eager_sort=0), dict(borrow=3, pixel_order=0),
                                  dict(borrow=3, pixel_order=2, handover=2, donate_max=8), dict(borrow=3, solo=0, thr_shade=8), dict(borrow=3, gpu_build=0, treelet=2, xcd_queues=0, static_first=0), dict(adaptive_order=2)])
def test_new_views_borrow_the_previous_views_order(R, opts):
    """Round 6: a NEW view of a prepared scene (another camera, same image size / partition) renders its first frame through the order /
    pixel list of the view rendered last (`borrow`), with the DONATE tail for the chains that list places wrongly (the ORD + DONATE
    instantiation), while it records its own; and the sorts of a recorded view run on the context's SECOND stream behind the recording
    frame (`eager_sort`) -- the next user of the order waits for their event.  A sliding camera, frame by frame, each frame against the
    CPU checker through that camera: whole frames, a part packed and in place, a view re-rendered later (its own list by then), 13 views
    (more than the 8 kept: the evicted views' buffers -- possibly still being sorted -- are taken over), and the knobs off."""
    import torch
    from raytracers_amd.dist import tile_rows
    c = R.Context()
    c.set_variant(3)
    c.set_option("sync_policy", 1)      # (a new view borrows from the view before it whatever the state of its sorts: the same launches in every run)
    for k, v in opts.items():
        c.set_option(k, v)
    for scene, h, w in (("irreg", 333, 250), ("rgbbox", 280, 400), ("floor:37:222", 90, 120)):
        orc = _oracle(scene)
        ps = R.prepare_scene(h, w, _scene(c, scene))
        base = np.asarray(ps.camera(), dtype=np.float32).reshape(12)
        cams = []
        for f in range(13):
            cam = base.copy()
            cam[0] += np.float32(0.25 * f); cam[3] += np.float32(0.25 * f)
            cams.append(cam)
        want = [orc.render(h, w, cam=cam)[0] for cam in cams]
        out = torch.empty((h, w), dtype=torch.int32, device="cuda")
        kept = []                                     # the library keeps the 8 most recently used views of a prepared scene
        for f in list(range(13)) + [2, 2, 12, 0, 0, 5]:
            out.fill_(-1)
            torch.cuda.synchronize()
            R.render_into(out.data_ptr(), h, w, ps, cam=cams[f])
            c.sync()
            ll = c.last_launch
            assert int((out.cpu().numpy() != want[f]).sum()) == 0, (scene, f, ll)
            in_lds = scene != "irreg"                                      # (auto borrows for scenes read from L2 only; 2 / 3 force it)
            if "waves=16" in ll and opts.get("adaptive_order", 1) == 1 and opts.get("sync_policy", 1) == 1:
                if f not in kept and kept and opts.get("borrow", 1) and not (in_lds and opts.get("borrow", 1) == 1):   # a view the prepared scene does not hold (any more): borrowed
                    assert "(borrowed)" in ll and "recording=" in ll and "recording=0" not in ll, (scene, f, ll)
                    if opts.get("handover", 1) and h * w > 4096 and "solo" not in opts and "treelet" not in opts:
                        assert "DONATE" in ll, (scene, f, ll)
                else:
                    assert "(borrowed)" not in ll, (scene, f, ll)
            kept = ([k for k in kept if k != f] + [f])[-8:]
        # a part of three, packed and in place, along the same path (views of another partition: their own records and orders)
        rows = R.part_rows(h, 1, 3)
        part = torch.empty((rows, w), dtype=torch.int32, device="cuda")
        image = torch.full((h, w), -5, dtype=torch.int32, device="cuda")
        mine = np.zeros(h, bool)
        mine[tile_rows(h, 1, 3)] = True
        for f in (0, 1, 2, 1):
            part.fill_(-3)
            torch.cuda.synchronize()
            R.render_into(part.data_ptr(), h, w, ps, part=1, nparts=3, cam=cams[f])
            c.sync()
            assert int((part.cpu().numpy() != want[f][mine]).sum()) == 0, (scene, "part", f, c.last_launch)
        for f in (3, 4, 3):
            image.fill_(-5)
            torch.cuda.synchronize()
            R.render_inplace_into(image.data_ptr(), h, w, ps, part=1, nparts=3, cams=cams[f])
            c.sync()
            got = image.cpu().numpy()
            assert int((got[mine] != want[f][mine]).sum()) == 0 and bool((got[~mine] == -5).all()), (scene, "in place", f, c.last_launch)
        ps.free()
    c.close()


@pytest.mark.parametrize("seconds,seed,side,spheres", [(15, 77000, 160, 20000), (15, 88000, 520, 60000)])
def test_random_parity_campaign(seconds, seed, side, spheres):
    """tools/fuzz_parity.py as a gate of the driver-run suite, half a minute of it: random scenes /
    cameras / sizes / bounce limits / launch knobs / partitions (packed and in place), both BVH builders and all kernel
    families against the oracle -- small images (~1 000 cases) and images large enough for the first-frame policy and the
    in-loop hand-over (~150 cases).  (The rounds' long runs: profiles/rNN/fuzz_*_final.txt.)"""
    import sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), str(seconds), str(seed), str(side), str(spheres)],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-2000:])
    assert "0 mismatches" in out.stdout


def test_axis_aligned_rays_and_nan_slabs(R, ctx):
    """A camera on the z axis over a grid whose box faces lie on x = 0 and y = 0: the centre
    column / row of an even-sized image gets a direction component of exactly 0 (inverse = inf),
    and (face - origin) * inf = 0 * inf = NaN in aabb_hit.  fmaxf / fminf must drop the NaN and
    the swap must follow the SIGN of the inverse, exactly as the reference does (ray.fut:53-70)."""
    g = np.array([-9.0, -3.0, 3.0, 9.0], np.float32)
    xs, ys, zs = np.meshgrid(g, g, np.array([-6.0, 0.0, 6.0], np.float32), indexing="ij")
    n = xs.size
    s = np.zeros((n, 7), np.float32)
    s[:, 0], s[:, 1], s[:, 2] = xs.ravel(), ys.ravel(), zs.ravel()
    s[:, 3:6] = np.linspace(0.4, 1.0, 3 * n, dtype=np.float32).reshape(n, 3)
    s[:, 6] = 3.0
    lf, la, fov = (0.0, 0.0, 40.0), (0.0, 0.0, 0.0), 40.0
    orc = O.OracleScene("custom", spheres7=s, look_from=lf, look_at=la, fov=fov)
    cam = orc.camera_floats(64, 64)
    # the premise: the centre column's primary direction has x == 0 exactly
    assert cam[3] + np.float32(0.5) * cam[6] - cam[0] == 0.0
    ref, _ = orc.render(64, 64)
    for variant in (1, 2, 3):
        ctx.set_variant(variant)
        ps = R.prepare_scene(64, 64, ctx.scene_from_spheres(s, lf, la, fov))
        assert int((R.render(64, 64, ps) != ref).sum()) == 0, variant
    ctx.set_variant(0)


# ---------------------------------------------------------------- work counters -----------
@pytest.mark.parametrize("scene,h", [("rgbbox", 200), ("irreg", 200), ("rgbbox", 1000), ("irreg", 1000)])
def test_work_counters_match_oracle(R, ctx, scene, h):
    ps = R.prepare_scene(h, h, ctx.scene(scene))
    _, cnt = _oracle(scene).render(h, h)
    st = ps.stats()
    assert st == {k: cnt[k] for k in ("rays", "box_tests", "leaf_tests")}


# ---------------------------------------------------------------- knobs keep parity -------
@pytest.mark.parametrize("opts", [
    dict(waves_per_wg=4, wgs_per_cu=4), dict(waves_per_wg=16, wgs_per_cu=1), dict(thr_shade=1, thr_leaf=1),
    dict(thr_shade=64, thr_leaf=64), dict(lmax=2), dict(lmax=16), dict(lds_scene_bytes=0),
    dict(lds_scene_bytes=4096), dict(lds_sph_first=1, lds_scene_bytes=8192), dict(look_max=1), dict(look_max=32, thr_shade=8),
])
@pytest.mark.parametrize("variant", [2, 3])
def test_persistent_knobs_do_not_change_pixels(R, opts, variant):
    c = R.Context()
    c.set_variant(variant)
    for k, v in opts.items():
        c.set_option(k, v)
    for scene in ("rgbbox", "irreg"):
        got = R.render(120, 136, R.prepare_scene(120, 136, c.scene(scene)))
        want, _ = _oracle(scene).render(120, 136)
        assert int((got != want).sum()) == 0, (scene, opts)
    c.close()


@pytest.mark.parametrize("variant", [2, 3])
def test_repeated_launches_share_the_ticket_counter(R, ctx, variant):
    """The persistent families share the context's ticket counters: the last wave of a launch to leave the queue
    zeroes them, so launches of different sizes and families follow each other with no host-side bookkeeping."""
    ctx.set_variant(variant)
    ps_a = R.prepare_scene(64, 64, ctx.rgbbox())
    ps_b = R.prepare_scene(100, 36, ctx.irreg())
    wa, _ = _oracle("rgbbox").render(64, 64)
    wb, _ = _oracle("irreg").render(100, 36)
    for _ in range(10):
        assert (R.render(64, 64, ps_a) == wa).all()
        assert (R.render(100, 36, ps_b) == wb).all()


@pytest.mark.parametrize("adaptive,deep_class,deep_split", [(0, 3, 2), (1, 3, 2), (2, 3, 2), (1, 0, 2), (1, 5, 1), (2, 8, 2), (1, 8, 3),
                                                            (1, 5, 0), (2, 6, 3)])
@pytest.mark.parametrize("scene,h,w", [("rgbbox", 333, 250), ("irreg", 200, 200)])
def test_adaptive_tile_order_renders_every_pixel(R, scene, h, w, adaptive, deep_class, deep_split):
    """The pooled family reorders tiles by the previous frame's bounce-chain record, gives the deepest tiles a wave that
    does not refill while they are in flight (deep_class) and hands the very deepest out in 2^deep_split pieces, each to
    a wave of its own (extra tickets at the head of the queue).  Frames 1..5 of the same prepared
    scene must each write every pixel (buffer poisoned before every frame) and stay bit-exact; a second size interleaved
    in between must not disturb it (it shares the context's ticket counter)."""
    import torch
    c = R.Context()
    c.set_variant(3)
    c.set_option("adaptive_order", adaptive)
    c.set_option("deep_class", deep_class)   # 8: every recorded tile is "deep" (its wave does not refill meanwhile)
    c.set_option("deep_split", deep_split)
    ps = R.prepare_scene(h, w, c.scene(scene))
    ps2 = R.prepare_scene(64, 72, c.scene(scene))
    want, _ = _oracle(scene).render(h, w)
    want2, _ = _oracle(scene).render(64, 72)
    out = torch.empty((h, w), dtype=torch.int32, device="cuda")
    out2 = torch.empty((64, 72), dtype=torch.int32, device="cuda")
    for frame in range(5):
        out.fill_(-1)
        out2.fill_(-1)
        torch.cuda.synchronize()
        R.render_into(out.data_ptr(), h, w, ps)
        R.render_into(out2.data_ptr(), 64, 72, ps2)
        c.sync()
        assert int((out.cpu().numpy() != want).sum()) == 0, frame
        assert int((out2.cpu().numpy() != want2).sum()) == 0, frame
    c.close()


@pytest.mark.parametrize("xcd,tpt,static_first", [(1, -1, 1), (1, 0, 0), (1, 2, 1), (1, 4, 0), (0, 1, 1), (0, 3, 0), (0, -1, 0),
                                                  (2, -1, 1), (2, 1, 0), (2, 3, 1)])
@pytest.mark.parametrize("scene,h,w", [("rgbbox", 333, 250), ("irreg", 1000, 1000), ("irreg", 40, 24)])
def test_tile_queue_layouts_render_every_pixel(R, scene, h, w, xcd, tpt, static_first):
    """The tile queue of the pooled family: one ticket counter per XCD -- each with its own strip of tile columns
    (xcd_queues=1) or taking turns over one queue (2) -- with stealing, several tiles per ticket (tpt_log2), the waves'
    first tickets handed out without an atomic (static_first) --
    combined with the adaptive order, deep tiles and their pieces.  Frames 0..3 of a view (recording frame, then ordered
    ones), a part of a three-way row partition and a batch of three frames must each write every pixel (buffers poisoned)
    and stay bit-exact; the counters must come back to zero after every launch (the next launch relies on it)."""
    import torch
    c = R.Context()
    c.set_variant(3)
    c.set_option("xcd_queues", xcd)
    c.set_option("tpt_log2", tpt)
    c.set_option("static_first", static_first)
    c.set_option("deep_class", 5)            # chains of >= 8 bounces count as deep: plenty of held waves and pieces
    ps = R.prepare_scene(h, w, c.scene(scene))
    want, _ = _oracle(scene).render(h, w)
    out = torch.empty((h, w), dtype=torch.int32, device="cuda")
    for frame in range(4):
        out.fill_(-1)
        torch.cuda.synchronize()
        R.render_into(out.data_ptr(), h, w, ps)
        c.sync()
        assert int((out.cpu().numpy() != want).sum()) == 0, frame
    # one part of three (its own view of the tile grid: fewer tile rows, the same strips)
    rows = R.part_rows(h, 1, 3)
    if rows:
        part = torch.full((rows, w), -1, dtype=torch.int32, device="cuda")
        for frame in range(3):
            part.fill_(-1)
            torch.cuda.synchronize()
            R.render_into(part.data_ptr(), h, w, ps, part=1, nparts=3)
            c.sync()
            from raytracers_amd.dist import tile_rows
            assert int((part.cpu().numpy() != want[tile_rows(h, 1, 3)]).sum()) == 0, frame
    # a batch (one counter whatever xcd_queues says: class-major over the frames), twice: recording launch, ordered launch
    nb = 3
    batch = torch.empty((nb, h, w), dtype=torch.int32, device="cuda")
    for rep in range(2):
        batch.fill_(-1)
        torch.cuda.synchronize()
        R.render_batch_into(batch.data_ptr(), h, w, ps, nb, frame_stride=h * w)
        c.sync()
        got = batch.cpu().numpy()
        for f in range(nb):
            assert int((got[f] != want).sum()) == 0, (rep, f)
    # and a single frame again after the batch
    out.fill_(-1)
    torch.cuda.synchronize()
    R.render_into(out.data_ptr(), h, w, ps)
    c.sync()
    assert int((out.cpu().numpy() != want).sum()) == 0
    c.close()


@pytest.mark.parametrize("scene,h,w", [("rgbbox", 200, 200), ("irreg", 333, 250)])
def test_wave_trace_of_the_instrumented_launch(R, scene, h, w):
    """rt_render_trace (what tools/trace_waves.py reads): the instrumented pooled launch in the production workgroup shape
    writes one record per wave; the leaf items all waves processed are exactly the oracle's sphere tests, every wave
    started, ended and left the tile queue, and the launch leaves the ticket counters zeroed (the frame after it is right)."""
    import ctypes as C
    from raytracers_amd._lib import lib
    c = R.Context()
    c.set_variant(3)
    ps = R.prepare_scene(h, w, c.scene(scene))
    want, cnt = _oracle(scene).render(h, w)
    for _ in range(2):   # recording frame, ordered frame
        assert int((R.render(h, w, ps) != want).sum()) == 0
    # (cull = 0: the reference's full test set; cull = 1: the same fold with subtrees behind the best hit left out -- irreg only: rgbbox's
    # tree is taller than its sweeps, its upper boxes are unconverged and it is never culled)
    for cull in (0, 1):
        c.set_option("cull", cull)
        rec = np.zeros((8192, 16), dtype=np.uint64)
        n = C.c_int32()
        c._check(lib.rt_render_trace(c._h, ps._h, h, w, 50, rec.ctypes.data, 8192, C.byref(n)))
        assert 0 < n.value <= 8192
        rec = rec[: n.value].astype(np.int64)
        assert (rec[:, 0] > 0).all() and (rec[:, 2] >= rec[:, 0]).all()            # start / end wall clock
        leaf_items = int((rec[:, 6] & 0xFFFFFFFF).sum())
        if cull == 0 or scene == "rgbbox":
            assert leaf_items == cnt["leaf_tests"]                                   # leaf items == sphere tests
        else:
            assert 0 < leaf_items < cnt["leaf_tests"] * 0.9                          # culled: fewer sphere tests (-36 % at 1000 x 1000)
        ops = (rec[:, 3] & 0x1FFFFF) + ((rec[:, 3] >> 21) & 0x1FFFFF) + ((rec[:, 3] >> 42) & 0x1FFFFF)
        assert int(ops.sum()) > 0 and int((rec[:, 7] & 0xFFFF).max()) >= 1          # operations ran, a bounce chain was seen
        assert int((R.render(h, w, ps) != want).sum()) == 0
    c.close()


# ---------------------------------------------------------------- threads sharing a context ----
@pytest.mark.parametrize("mode,scene", [("rt", "irreg"), ("rt", "rgbbox"), ("futhark", "irreg")])
def test_host_threads_share_one_context(mode, scene):
    """a Futhark context serialises concurrent calls with an internal lock.  tools/ctx_threads: two host threads, ONE
    context, ONE prepared scene (mode rt: a view each -- both views' orders and pixel lists live in the one rt_prepared; mode futhark: a
    prepared scene each on one futhark_context, images through its pool), 200 frames each with render + sync + values per frame
    and option writes in between.  Every frame must equal its view's first, and the views' checksums the CPU checker's.
    (The same program over a -fsanitize=thread build of the library's host code: tools/gpu.sh tsan, log under profiles/r06/.)"""
    exe = os.path.join(ROOT, "build", "ctx_threads")
    subprocess.run(["make", "-s", "build/ctx_threads"], cwd=ROOT, check=True)
    n, frames, threads = 256, 200, 2
    out = subprocess.run([exe, mode, scene, str(n), str(frames), str(threads)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "every frame equals its view's first" in out.stdout, out.stdout + out.stderr
    orc = _oracle(scene)
    for k in range(threads):
        if mode == "rt":
            cam = orc.camera_floats(n, n)
            cam[0] += np.float32(0.37) * np.float32(k)
            want, _ = orc.render(n, n, cam=cam)
        else:
            want, _ = orc.render(n + 8 * k, n)
        assert f"thread {k} checksum {O.checksum(want):08x}" in out.stdout, (k, out.stdout)


# ---------------------------------------------------------------- row-tile partition ------
@pytest.mark.parametrize("variant", list(VARIANTS))
@pytest.mark.parametrize("nparts", [2, 3, 8])
def test_parts_assemble_to_the_full_image(R, ctx, variant, nparts):
    import torch
    ctx.set_variant(VARIANTS[variant])
    h, w = 203, 96
    ps = R.prepare_scene(h, w, ctx.irreg())
    want, _ = _oracle("irreg").render(h, w)
    image = torch.full((h, w), -1, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()      # (the fill runs on torch's stream, the render on the context's own: order them)
    for p in range(nparts):
        rows = R.part_rows(h, p, nparts)
        part = torch.empty((max(rows, 1), w), dtype=torch.int32, device="cuda")
        R.render_into(part.data_ptr(), h, w, ps, part=p, nparts=nparts)
        R.place_part(ctx, h, w, p, nparts, part.data_ptr(), image.data_ptr())
    ctx.sync()
    torch.cuda.synchronize()
    assert int((image.cpu().numpy() != want).sum()) == 0


@pytest.mark.parametrize("nparts", [2, 5, 8])
def test_stacked_parts_assemble_in_one_kernel(R, ctx, nparts):
    """rt_place_parts: what rank 0 runs on the gathered, padded send buffers of all ranks."""
    import torch
    from raytracers_amd.dist import max_part_rows
    ctx.set_variant(0)
    h, w = 211, 88
    ps = R.prepare_scene(h, w, ctx.rgbbox())
    want, _ = _oracle("rgbbox").render(h, w)
    pad = max_part_rows(h, nparts)
    stacked = torch.full((nparts, pad, w), -7, dtype=torch.int32, device="cuda")
    image = torch.full((h, w), -1, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    for p in range(nparts):
        R.render_into(stacked[p].data_ptr(), h, w, ps, part=p, nparts=nparts)
    R.place_parts(ctx, h, w, nparts, pad, stacked.data_ptr(), image.data_ptr())
    ctx.sync()
    assert int((image.cpu().numpy() != want).sum()) == 0
    # strided form: each part's rows sit inside a longer per-rank record (several frames per gather)
    lead, stride = 13, pad * w + 29
    rec = torch.full((nparts, stride), -9, dtype=torch.int32, device="cuda")
    rec[:, lead:lead + pad * w] = stacked.view(nparts, pad * w)
    image.fill_(-1)
    torch.cuda.synchronize()
    R.place_parts(ctx, h, w, nparts, pad, rec.data_ptr() + 4 * lead, image.data_ptr(), part_stride=stride)
    ctx.sync()
    assert int((image.cpu().numpy() != want).sum()) == 0
    with pytest.raises(Exception):   # a stride shorter than the largest part is refused
        R.place_parts(ctx, h, w, nparts, pad, rec.data_ptr(), image.data_ptr(), part_stride=w)


def test_sharded_renderer_single_rank_on_torch_stream(R):
    import torch
    from raytracers_amd.dist import HipPartRenderer, ShardedRenderer
    h, w = 200, 200
    pr = HipPartRenderer("rgbbox", h, w, "cuda:0")
    sr = ShardedRenderer(pr, h, w, device="cuda:0")
    img = sr.render()
    torch.cuda.synchronize()
    want, _ = _oracle("rgbbox").render(h, w)
    assert int((img.cpu().numpy() != want).sum()) == 0


# ---------------------------------------------------------------- the drop-in boundary ----
def test_futhark_abi_call_sequence(R):
    """The exact call sequence of futhark/main.c:59-141 through ctypes."""
    from raytracers_amd._lib import lib
    vp = C.c_void_p
    lib.futhark_context_config_new.restype = vp
    lib.futhark_context_new.restype = vp
    lib.futhark_context_new.argtypes = [vp]
    lib.futhark_context_get_error.restype = vp
    lib.futhark_context_get_error.argtypes = [vp]
    cfg = vp(lib.futhark_context_config_new())
    fctx = vp(lib.futhark_context_new(cfg))
    assert lib.futhark_context_get_error(fctx) is None
    for name, entry in (("rgbbox", lib.futhark_entry_rgbbox), ("irreg", lib.futhark_entry_irreg)):
        scene, ps, img = vp(), vp(), vp()
        entry.argtypes = [vp, C.POINTER(vp)]
        assert entry(fctx, C.byref(scene)) == 0
        lib.futhark_entry_prepare_scene.argtypes = [vp, C.POINTER(vp), C.c_int64, C.c_int64, vp]
        lib.futhark_entry_render.argtypes = [vp, C.POINTER(vp), C.c_int64, C.c_int64, vp]
        lib.futhark_values_i32_2d.argtypes = [vp, vp, vp]
        lib.futhark_context_sync.argtypes = [vp]
        h, w = 120, 200   # h != w: catches an h/w swap at the boundary
        for _ in range(2):
            if ps:
                lib.futhark_free_opaque_prepared_scene.argtypes = [vp, vp]
                lib.futhark_free_opaque_prepared_scene(fctx, ps)
            assert lib.futhark_entry_prepare_scene(fctx, C.byref(ps), h, w, scene) == 0
            assert lib.futhark_context_sync(fctx) == 0
        for _ in range(2):
            if img:
                lib.futhark_free_i32_2d.argtypes = [vp, vp]
                lib.futhark_free_i32_2d(fctx, img)
            assert lib.futhark_entry_render(fctx, C.byref(img), h, w, ps) == 0
            assert lib.futhark_context_sync(fctx) == 0
        host = np.empty((h, w), np.int32)
        assert lib.futhark_values_i32_2d(fctx, img, host.ctypes.data) == 0
        want, _ = _oracle(name).render(h, w)
        assert int((host != want).sum()) == 0
        lib.futhark_free_i32_2d.argtypes = [vp, vp]
        lib.futhark_free_opaque_prepared_scene.argtypes = [vp, vp]
        lib.futhark_free_opaque_scene.argtypes = [vp, vp]
        lib.futhark_free_i32_2d(fctx, img)
        lib.futhark_free_opaque_prepared_scene(fctx, ps)
        lib.futhark_free_opaque_scene(fctx, scene)
    # the conventional rest of a Futhark library's array API: host -> device -> host round trip
    lib.futhark_new_i32_2d.restype = vp
    lib.futhark_new_i32_2d.argtypes = [vp, vp, C.c_int64, C.c_int64]
    lib.futhark_shape_i32_2d.restype = C.POINTER(C.c_int64)
    lib.futhark_shape_i32_2d.argtypes = [vp, vp]
    lib.futhark_values_raw_i32_2d.restype = vp
    lib.futhark_values_raw_i32_2d.argtypes = [vp, vp]
    src = np.arange(7 * 11, dtype=np.int32).reshape(7, 11)
    arr = lib.futhark_new_i32_2d(fctx, src.ctypes.data, 7, 11)
    assert arr and lib.futhark_values_raw_i32_2d(fctx, arr)
    shp = lib.futhark_shape_i32_2d(fctx, arr)
    assert (shp[0], shp[1]) == (7, 11)
    back = np.zeros_like(src)
    lib.futhark_values_i32_2d.argtypes = [vp, vp, vp]
    assert lib.futhark_values_i32_2d(fctx, arr, back.ctypes.data) == 0 and (back == src).all()
    lib.futhark_free_i32_2d.argtypes = [vp, vp]
    lib.futhark_free_i32_2d(fctx, arr)
    lib.futhark_context_clear_caches.argtypes = [vp]
    assert lib.futhark_context_clear_caches(fctx) == 0
    lib.futhark_context_free.argtypes = [vp]
    lib.futhark_context_config_free.argtypes = [vp]
    lib.futhark_context_free(fctx)
    lib.futhark_context_config_free(cfg)


def _read_ppm(path):
    tok = open(path).read().split()
    assert tok[0] == "P3"
    w, h = int(tok[1]), int(tok[2])
    v = np.array(tok[4:], dtype=np.int32).reshape(h, w, 3)
    return (v[..., 0] << 16) | (v[..., 1] << 8) | v[..., 2]


@pytest.mark.parametrize("scene", ["rgbbox", "irreg"])
def test_reference_harness_unmodified(scene, tmp_path):
    """oracle/_ref/futhark_main is the reference's own futhark/main.c, compiled unmodified in
    the build container against include/ray.h; here it runs on the GPU against our library and
    its PPM output must equal the oracle's image."""
    exe = os.path.join(ROOT, "oracle", "_ref", "futhark_main")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/futhark_main was not prebuilt (needs /root/reference)")
    ppm = str(tmp_path / "out.ppm")
    # main.c: -n is the height, -m the width (main.c:36-41)
    out = subprocess.run([exe, "-s", scene, "-n", "300", "-m", "400", "-r", "3", "-f", ppm], capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "Rendering in" in out.stdout and "Scene BVH construction in" in out.stdout
    want, _ = _oracle(scene).render(300, 400)
    assert int((_read_ppm(ppm) != want).sum()) == 0


def test_native_bench_runs(tmp_path):
    exe = os.path.join(ROOT, "build", "rtbench")
    if not os.path.exists(exe):
        pytest.skip("build/rtbench not built")
    ppm = str(tmp_path / "p.ppm")
    out = subprocess.run([exe, "-s", "irreg", "-n", "160", "-m", "120", "-r", "2", "-g", "3", "-f", ppm],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    want, _ = _oracle("irreg").render(160, 120)
    assert int((_read_ppm(ppm) != want).sum()) == 0


def test_render_at_another_size_uses_the_prepared_camera(R, ctx):
    """`render h w prepared` (ray.fut:246) accepts any size: the camera -- and so the aspect ratio -- is the
    one prepare_scene derived.  Same pixels as render_image with that camera passed explicitly."""
    ps = R.prepare_scene(120, 160, ctx.rgbbox())
    a = R.render(90, 200, ps)
    b = R.render_image(ps, 200, 90, ps.camera())
    assert a.shape == (90, 200) and int((a != b).sum()) == 0
    want, _ = _oracle("rgbbox").render(120, 160)
    assert int((R.render(120, 160, ps) != want).sum()) == 0


@pytest.mark.parametrize("variant", [0, 1, 2])
def test_batch_of_frames_in_one_launch(R, variant):
    """rt_render_batch: N frames per launch, each equal to the frame rendered on its own -- the same view repeated (with
    and without the adaptive tile order, padded frame stride), a camera path, and one part of a partition."""
    import torch
    ctx = R.Context()
    ctx.set_variant(variant)
    h, w, n = 120, 168, 5
    ps = R.prepare_scene(h, w, ctx.rgbbox())
    want, _ = _oracle("rgbbox").render(h, w)
    stride = h * w + 40
    for rep in range(3):
        out = torch.full((n, stride), -7, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()      # (the fill runs on torch's stream, the render on the context's own: order them)
        R.render_batch_into(out.data_ptr(), h, w, ps, n, frame_stride=stride)
        ctx.sync()
        got = out.cpu().numpy()
        for f in range(n):
            assert int((got[f, :h * w].reshape(h, w) != want).sum()) == 0, (rep, f)
        assert (got[:, h * w:] == -7).all()
    # a camera path: frame f through its own camera == render_image with that camera
    base = ps.camera()
    cams = np.stack([base + np.float32(0.37 * f) * np.array([1, 0, 0] * 1 + [0] * 9, np.float32) for f in range(n)])
    out = torch.zeros((n, h * w), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()      # (the fill runs on torch's stream, the render on the context's own: order them)
    R.render_batch_into(out.data_ptr(), h, w, ps, n, cams=cams)
    ctx.sync()
    got = out.cpu().numpy()
    for f in range(n):
        assert int((got[f].reshape(h, w) != R.render_image(ps, w, h, cams[f])).sum()) == 0, f
    # part 1 of 3 of every frame
    rows = R.part_rows(h, 1, 3)
    out = torch.zeros((n, rows * w), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()      # (the fill runs on torch's stream, the render on the context's own: order them)
    R.render_batch_into(out.data_ptr(), h, w, ps, n, part=1, nparts=3)
    one = torch.zeros((rows, w), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()      # (the fill runs on torch's stream, the render on the context's own: order them)
    R.render_into(one.data_ptr(), h, w, ps, part=1, nparts=3)
    ctx.sync()
    for f in range(n):
        assert int((out[f].cpu().numpy() != one.cpu().numpy().reshape(-1)).sum()) == 0
    ps.free()
    ctx.close()


# ---------------------------------------------------------------- one process, several devices ---
@pytest.mark.parametrize("gather", [0, 1, 3])
@pytest.mark.parametrize("scene,h,w", [("rgbbox", 333, 250), ("irreg", 1000, 1000)])
def test_multi_device_context_on_one_gpu(R, scene, h, w, gather):
    """rt_context_create_multi with the device listed three times: the whole fan-out (replicated prepare_scene,
    cyclic row tiles on three streams, then either the devices' direct stores into the caller's image -- gather 3, and
    what auto picks -- or the peer-copy gather + assembly) through the single-device entry points."""
    import bench
    mc = R.Context(devices=[0, 0, 0])
    mc.set_option("gather", gather)
    assert mc.num_devices == 3 and mc.gather_mode == ("peer-copy" if gather == 1 else "direct-store")   # (a getter: no communicators are created for it)
    ps = R.prepare_scene(h, w, mc.scene(scene))
    want, _ = _oracle(scene).render(h, w)
    for rep in range(3):                       # frames chase each other through the shared gather buffers
        assert int((R.render(h, w, ps) != want).sum()) == 0, rep
    if (scene, h, w) in bench.FRAME_CHECKSUM:
        assert O.checksum(want) == bench.FRAME_CHECKSUM[(scene, h, w)]
    # a smaller and then a larger frame: the gather buffers shrink-fit / regrow
    for hh, ww in ((64, 80), (h + 16, w)):
        ps2 = R.prepare_scene(hh, ww, mc.scene(scene))
        w2, _ = _oracle(scene).render(hh, ww)
        assert int((R.render(hh, ww, ps2) != w2).sum()) == 0
        ps2.free()
    # a caller's own partition is refused, knobs reach every device, BVH getters serve the first device
    import torch
    out = torch.zeros((h, w), dtype=torch.int32, device="cuda")
    with pytest.raises(R.RtError, match="whole frames"):
        R.render_into(out.data_ptr(), h, w, ps, part=0, nparts=2)
    mc.set_option("thr_shade", 16)
    mc.set_variant(R.VARIANT_PERSISTENT)
    assert int((R.render(h, w, ps) != want).sum()) == 0
    assert ps.bvh_arrays()["L"].shape == (ps.num_spheres, 7)
    ps.free()
    mc.close()


@pytest.mark.parametrize("gather0", [1, 3])
@pytest.mark.parametrize("scene,h,w,nb", [("irreg", 4000, 4000, 6), ("rgbbox", 333, 250, 5)])
def test_batch_on_a_multi_device_context(R, scene, h, w, nb, gather0):
    """rt_render_batch on a multi-device context (three parts on the one GPU): every device renders its row tiles of ALL the
    frames in one launch, one gather moves nframes x part per device, ONE assembly launch writes the images.  irreg
    4000x4000 is the configuration north_star states its scaling target on: six frames, each db269d43.  Then a camera
    path (a camera per frame) and, with gather=2 on a single device, the same through RCCL send/recv."""
    import bench
    import torch
    mc = R.Context(devices=[0, 0, 0])
    mc.set_option("gather", gather0)              # peer-copy gather + one assembly launch / direct stores into the images
    ps = R.prepare_scene(h, w, mc.scene(scene))
    buf = torch.empty((nb, h, w), dtype=torch.int32, device="cuda")
    want = bench.FRAME_CHECKSUM.get((scene, h, w))
    ref = None if want is not None else _oracle(scene).render(h, w)[0]
    cks = bench.Checksummer(torch.device("cuda"))
    for rep in range(2):                          # recording launch, ordered launch
        buf.fill_(-1)
        torch.cuda.synchronize()
        R.render_batch_into(buf.data_ptr(), h, w, ps, nb, frame_stride=h * w)
        mc.sync()
        for f in range(nb):
            if want is not None:
                assert cks(buf[f]) == want, (rep, f)
            else:
                assert int((buf[f].cpu().numpy() != ref).sum()) == 0, (rep, f)
    assert int((R.render(h, w, ps) != (ref if ref is not None else buf[0].cpu().numpy())).sum()) == 0   # and a single frame after it
    ps.free()
    mc.close()
    if want is not None:
        return
    # a camera per frame, on the multi-device context and through the forced RCCL path of a single device
    orc = _oracle(scene)
    cams = np.stack([orc.camera_floats(h + 8 * f, w) for f in range(nb)])
    for devices, gather in (([0, 0, 0], gather0), ([0], 2)):
        mc = R.Context(devices=devices)
        mc.set_option("gather", gather)
        ps = R.prepare_scene(h, w, mc.scene(scene))
        single = [R.render_image(ps, w, h, cams[f]) for f in range(nb)]
        buf.fill_(-1)
        torch.cuda.synchronize()
        R.render_batch_into(buf.data_ptr(), h, w, ps, nb, frame_stride=h * w, cams=cams)
        cams_copy = cams.copy()
        cams[:] = 0.0                             # the call has copied the cameras: the caller's array is its own again
        mc.sync()
        cams[:] = cams_copy
        for f in range(nb):
            assert int((buf[f].cpu().numpy() != single[f]).sum()) == 0, (devices, f)
        ps.free()
        mc.close()


def test_batch_assembly_in_one_launch(R, ctx):
    """rt_place_parts_batch (what rank 0 runs behind the gather of a batch step) against the per-frame assembly."""
    import torch
    h, w, nparts, nb = 77, 53, 3, 4
    pad = max(R.part_rows(h, p, nparts) for p in range(nparts))
    fs_in, part_stride = pad * w + 5, nb * (pad * w + 5) + 11
    rng = np.random.default_rng(3)
    stacked = torch.from_numpy(rng.integers(-2**31, 2**31 - 1, size=(nparts * part_stride,), dtype=np.int64).astype(np.int32)).cuda()
    got = torch.full((nb, h, w), -1, dtype=torch.int32, device="cuda")
    want = torch.full((nb, h, w), -2, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    R.place_parts_batch(ctx, h, w, nparts, part_stride, nb, fs_in, stacked.data_ptr(), got.data_ptr())
    for f in range(nb):
        R.place_parts(ctx, h, w, nparts, pad, stacked[f * fs_in:].data_ptr(), want[f].data_ptr(), part_stride=part_stride)
    ctx.sync()
    assert bool((got == want).all())


def test_multi_device_rccl_gather_on_one_gpu(R):
    """gather=2 forces the RCCL path even for a single device: librccl is loaded on demand, one communicator,
    the part travels by grouped ncclSend / ncclRecv (to self) into the stacked buffer and is assembled from there."""
    mc = R.Context(devices=[0])
    mc.set_option("gather", 2)
    ps = R.prepare_scene(200, 200, mc.irreg())
    want, _ = _oracle("irreg").render(200, 200)
    for rep in range(3):
        assert int((R.render(200, 200, ps) != want).sum()) == 0
    assert mc.gather_mode == "rccl"
    ps.free()
    mc.close()


def test_multi_device_real_devices(R):
    """With two or more GPUs present: the same through RCCL over xGMI (skipped on a one-GPU box)."""
    from raytracers_amd._lib import lib
    n = int(lib.rt_device_count())
    if n < 2:
        pytest.skip("one GPU")
    import bench
    import torch
    import time
    mc = R.Context(devices=list(range(n)))
    assert mc.gather_mode == "direct-store"            # auto: every device can store into the first one's memory
    report = []                                        # the first real multi-GPU run prints the two constants tools/scale_prediction.py assumes
    for gather in (3, 2, 1, 0):                        # direct stores over xGMI, RCCL send/recv, peer copies, auto
        mc.set_option("gather", gather)
        for scene, h, w in (("irreg", 4000, 4000), ("rgbbox", 1000, 1000)):
            ps = R.prepare_scene(h, w, mc.scene(scene))
            for rep in range(3):
                assert O.checksum(R.render(h, w, ps)) == bench.FRAME_CHECKSUM[(scene, h, w)], (gather, rep)
            # (timing, printed not asserted: render + exchange + sync per frame, as the reference's harness does)
            img = torch.empty((h, w), dtype=torch.int32, device="cuda:0")
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for rep in range(8):
                R.render_into(img.data_ptr(), h, w, ps)
                mc.sync()
            one = (time.perf_counter() - t0) / 8 * 1e3
            # ... and a batch: every device its rows of all six frames in one launch (+ one gather, one assembly launch)
            buf = torch.full((6, h, w), -1, dtype=torch.int32, device="cuda:0")
            torch.cuda.synchronize()
            R.render_batch_into(buf.data_ptr(), h, w, ps, 6, frame_stride=h * w)
            mc.sync()
            cks = bench.Checksummer(torch.device("cuda:0"))
            assert all(cks(buf[f]) == bench.FRAME_CHECKSUM[(scene, h, w)] for f in range(6)), gather
            t0 = time.perf_counter()
            R.render_batch_into(buf.data_ptr(), h, w, ps, 6, frame_stride=h * w)
            mc.sync()
            report.append(f"gather={gather} ({mc.gather_mode}, rccl_ranks {mc.rccl_ranks}) {scene} {w}x{h} on {n} devices: {one:.3f} ms per frame one at a time, "
                          f"{(time.perf_counter() - t0) / 6 * 1e3:.3f} ms per frame in a batch of 6")
            ps.free()
        assert mc.gather_mode in {3: ("direct-store",), 2: ("rccl", "peer-copy"), 1: ("peer-copy",), 0: ("direct-store",)}[gather]
        if gather == 2 and mc.gather_mode == "rccl":   # distinct devices: one communicator rank per device carried the frames
            assert mc.rccl_ranks == n
    mc.close()
    text = "\n".join(["multi-device timings (tools/scale_prediction.py models 25 us per ordering signal and half the link rate for 4-byte stores):"] + report)
    print("\n" + text)
    import warnings
    warnings.warn(text)                                # (pytest -q hides a passing test's output; its warnings summary it prints)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "multi_device_timings.txt"), "a") as f:
        f.write(text + "\n")


def test_bench_line_on_two_real_gpus():
    """The driver's N = 2 command as it is (one rank per GPU, backend nccl = RCCL over xGMI): sharded steps verified against
    the oracle's checksums, one gather per launch, one assembly launch per scene (skipped on a one-GPU box)."""
    import json
    import sys
    from raytracers_amd._lib import lib
    if int(lib.rt_device_count()) < 2:
        pytest.skip("one GPU")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29579", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2"],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[:500]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["verified"] is True and d["value"] > 0
    assert d["irreg_4000"]["verified"] is True


@pytest.mark.parametrize("devices", ["0,0", "0-0"])
def test_reference_harness_unmodified_multi_device(devices, tmp_path):
    """The reference's own main.c binary, untouched, on a multi-device context: RT_DEVICES picks the devices
    (main.c never calls futhark_context_config_set_device)."""
    exe = os.path.join(ROOT, "oracle", "_ref", "futhark_main")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/futhark_main was not prebuilt (needs /root/reference)")
    ppm = str(tmp_path / "out.ppm")
    out = subprocess.run([exe, "-s", "irreg", "-n", "300", "-m", "400", "-r", "3", "-f", ppm], capture_output=True,
                         text=True, timeout=300, env=dict(os.environ, RT_DEVICES=devices))
    assert out.returncode == 0, out.stdout + out.stderr
    want, _ = _oracle("irreg").render(300, 400)
    assert int((_read_ppm(ppm) != want).sum()) == 0


@pytest.mark.parametrize("variant", list(VARIANTS))
@pytest.mark.parametrize("scene,h,w", [("rgbbox", 77, 53), ("irreg", 200, 168)])
def test_parts_rendered_in_place(R, ctx, scene, h, w, variant):
    """rt_render_part_inplace: every part stores its rows at their places in ONE full image (what a rank does into rank 0's
    buffer): all parts of 1, 2, 3 and 8 together give the oracle's image, a part alone leaves the other rows untouched;
    a batch with a padded frame stride and a camera per frame; max_depth 0."""
    import torch
    ctx.set_variant(VARIANTS[variant])
    ps = R.prepare_scene(h, w, ctx.scene(scene))
    want, _ = _oracle(scene).render(h, w)
    from raytracers_amd.dist import tile_rows
    for nparts in (1, 2, 3, 8):
        img = torch.full((h, w), -9, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        R.render_inplace_into(img.data_ptr(), h, w, ps, part=nparts - 1, nparts=nparts)
        ctx.sync()
        got = img.cpu().numpy()
        mine = tile_rows(h, nparts - 1, nparts)
        other = np.setdiff1d(np.arange(h), mine)
        assert (got[other] == -9).all() and int((got[mine] != want[mine]).sum()) == 0, nparts
        for p in range(nparts - 1):
            R.render_inplace_into(img.data_ptr(), h, w, ps, part=p, nparts=nparts)
        ctx.sync()
        assert int((img.cpu().numpy() != want).sum()) == 0, nparts
    # a batch: three frames, padded stride, a camera each, parts 0..2 of 3
    nb, stride = 3, h * w + 24
    orc = _oracle(scene)
    cams = np.stack([orc.camera_floats(h + 8 * f, w) for f in range(nb)])
    buf = torch.full((nb, stride), -9, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    for p in range(3):
        R.render_inplace_into(buf.data_ptr(), h, w, ps, nframes=nb, frame_stride=stride, cams=cams, part=p, nparts=3)
    ctx.sync()
    got = buf.cpu().numpy()
    assert (got[:, h * w:] == -9).all()
    for f in range(nb):
        assert int((got[f, :h * w].reshape(h, w) != R.render_image(ps, w, h, cams[f])).sum()) == 0, f
    # max_depth 0: black rows, only the part's own
    img = torch.full((h, w), -9, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    R.render_inplace_into(img.data_ptr(), h, w, ps, max_depth=0, part=1, nparts=2)
    ctx.sync()
    got = img.cpu().numpy()
    mine = tile_rows(h, 1, 2)
    assert (got[mine] == 0).all() and (np.delete(got, mine, axis=0) == -9).all()
    ctx.set_variant(0)
    ps.free()


@pytest.mark.parametrize("devices", [[0], [0, 0, 0]])
def test_camera_batches_of_growing_size_on_one_context(R, devices):
    """The camera block of rt_render_batch grows with the batch: a batch of 2 cameras, then of 20, then of 3 on the same
    context (single-device, and a multi-device one where every child context stages its own copy)."""
    import torch
    mc = R.Context(devices=devices) if len(devices) > 1 else R.Context()
    h, w = 64, 88
    ps = R.prepare_scene(h, w, mc.irreg())
    orc = _oracle("irreg")
    for nb in (2, 20, 3):
        cams = np.stack([orc.camera_floats(h + 8 * f, w) for f in range(nb)])
        buf = torch.full((nb, h, w), -1, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        R.render_batch_into(buf.data_ptr(), h, w, ps, nb, frame_stride=h * w, cams=cams)
        mc.sync()
        for f in (0, nb - 1):
            assert int((buf[f].cpu().numpy() != R.render_image(ps, w, h, cams[f])).sum()) == 0, (nb, f)
    ps.free()
    mc.close()


def test_render_entries_never_wait_for_the_device(R):
    """Every rt_render* call after prepare_scene ENQUEUES and returns (include/rt_mi355x.h): with the stream kept busy by
    a long launch, the calls that render a view for the second and third time -- where the view's deep-tile policy used
    to be read back behind a stream synchronise -- come back while that launch is still running."""
    import time
    import torch
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        ctx = R.Context(0, st.cuda_stream)
        big = R.prepare_scene(4000, 4000, ctx.irreg())
        ps = R.prepare_scene(200, 200, ctx.irreg())
        long_buf = torch.empty((8, 4000, 4000), dtype=torch.int32, device="cuda")
        img = torch.empty((200, 200), dtype=torch.int32, device="cuda")
        R.render_batch_into(long_buf.data_ptr(), 4000, 4000, big, 8, frame_stride=16000000)   # warm (tile order etc.)
        R.render_into(img.data_ptr(), 200, 200, ps)
        torch.cuda.synchronize()
        ps2 = R.prepare_scene(200, 200, ctx.irreg())       # a prepared scene whose view has never been rendered
        torch.cuda.synchronize()
        busy = torch.cuda.Event()
        R.render_batch_into(long_buf.data_ptr(), 4000, 4000, big, 8, frame_stride=16000000)   # ~15 ms of work on the stream
        busy.record()
        t0 = time.perf_counter()
        for _ in range(4):                                  # first frame (records), second (used to sync), third, fourth
            R.render_into(img.data_ptr(), 200, 200, ps2)
        dt = time.perf_counter() - t0
        still_running = not busy.query()
        torch.cuda.synchronize()
        want, _ = _oracle("irreg").render(200, 200)
        assert int((img.cpu().numpy() != want).sum()) == 0
        assert still_running, f"the four enqueues took {dt * 1e3:.2f} ms and the 8-frame launch ahead of them had already finished"
        # ... and once the device has caught up the policy is there: the next frames run on it (same pixels)
        for _ in range(2):
            R.render_into(img.data_ptr(), 200, 200, ps2)
            torch.cuda.synchronize()
        assert int((img.cpu().numpy() != want).sum()) == 0
        for p in (ps, ps2, big):
            p.free()
        ctx.close()


# ---------------------------------------------------------------- error behaviour ---------
def test_sync_policy_makes_the_instantiation_reproducible(R):
    """Option sync_policy = 1 (measurements, PMC passes): a render entry waits for the view's class table instead of polling for it, so
    which instantiation renders frame k of a view does not depend on when a copy lands -- two contexts, the same call sequence through
    the tile-ticket path (pixel_order = 0, the only reader of the table), the same rt_context_last_launch frame by frame; pixels exact."""
    seqs = []
    for rep in range(2):
        c = R.Context()
        c.set_variant(3)
        c.set_option("sync_policy", 1)
        c.set_option("pixel_order", 0)
        ps = R.prepare_scene(500, 500, c.scene("irreg"))
        want, _ = _oracle("irreg").render(500, 500)
        seq = []
        for frame in range(5):
            assert int((R.render(500, 500, ps) != want).sum()) == 0, frame
            seq.append(c.last_launch)
        seqs.append(seq)
        ps.free()
        c.close()
    assert seqs[0] == seqs[1], seqs
    assert "deep_split=6" in seqs[0][-1] or "SOLO" in seqs[0][-1], seqs[0]   # (the policy did arrive: irreg's deep tiles go out pixel by pixel)


def test_errors_are_codes_with_messages(R, ctx):
    """Every entry returns non-zero on failure and leaves a message (the reference's harness
    convention is `assert(ret == 0)`, futhark/main.c:74,97,116,131); nothing is written then."""
    import torch
    ps = R.prepare_scene(32, 48, ctx.rgbbox())
    out = torch.full((32, 48), 123, dtype=torch.int32, device="cuda")
    with pytest.raises(R.RtError, match="partition"):
        R.render_into(out.data_ptr(), 32, 48, ps, part=3, nparts=3)
    with pytest.raises(R.RtError, match="null"):
        R.render_into(0, 32, 48, ps)
    with pytest.raises(R.RtError, match="depth"):
        R.render_into(out.data_ptr(), 32, 48, ps, max_depth=-1)
    with pytest.raises(R.RtError, match="unknown option"):
        ctx.set_option("no_such_knob", 1)
    with pytest.raises(R.RtError, match="at least 2|2 \\.\\."):
        R.prepare_scene(8, 8, ctx.scene_from_spheres(np.zeros((1, 7), np.float32) + 1, (0, 0, 5), (0, 0, 0), 60.0))
    ctx.sync()
    assert int((out.cpu().numpy() != 123).sum()) == 0
    # and the context is still usable afterwards
    want, _ = _oracle("rgbbox").render(32, 48)
    assert int((R.render(32, 48, ps) != want).sum()) == 0


@pytest.mark.parametrize("variant", [1, 2])
def test_odd_rows_per_tile_on_the_non_pooled_families(R, ctx, variant):
    """rows_per_tile need not be a power of two for the pixel / persistent families."""
    import torch
    ctx.set_variant(variant)
    h, w, rpt, nparts = 77, 40, 5, 3
    ps = R.prepare_scene(h, w, ctx.irreg())
    want, _ = _oracle("irreg").render(h, w)
    image = torch.full((h, w), -1, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()      # (the fill runs on torch's stream, the render on the context's own: order them)
    for p in range(nparts):
        rows = R.part_rows(h, p, nparts, rows_per_tile=rpt)
        part = torch.empty((max(rows, 1), w), dtype=torch.int32, device="cuda")
        R.render_into(part.data_ptr(), h, w, ps, part=p, nparts=nparts, rows_per_tile=rpt)
        R.place_part(ctx, h, w, p, nparts, part.data_ptr(), image.data_ptr(), rows_per_tile=rpt)
    ctx.sync()
    torch.cuda.synchronize()
    assert int((image.cpu().numpy() != want).sum()) == 0


# ---------------------------------------------------------------- the bench contract ------
@pytest.mark.parametrize("force_gather", [0, 1])
def test_bench_line_contract(tmp_path, force_gather):
    """bench.py prints exactly ONE line on stdout, a JSON object with the keys the driver reads
    (force_gather: the RCCL gather + assembly path on one rank; RCCL's banner must not reach stdout)."""
    import json
    import sys
    env = dict(os.environ)
    if force_gather:
        env["RT_FORCE_GATHER"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "2",
                          "--no-cpu-baseline"] + (["--repeats", "1"] if force_gather else []), capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout[:500]
    d = json.loads(lines[0])
    if force_gather:      # --repeats 1: one bracket, the line of the rounds before --repeats existed
        assert "repeats" not in d and "brackets_ms" not in d
    else:                 # the default: five identical brackets, the line describes the median one
        assert d["repeats"] == 5 and len(d["brackets_ms"]) == 5
        assert abs(d["ms_per_step"] * d["steps"] - sorted(d["brackets_ms"])[2]) < 1e-9
        assert d["value_min"] <= d["value"] <= d["value_max"]
        assert d["verified_images"] >= 5 * 2 * 6 + 2
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["unit"] == "Mray/s" and d["n_gpus"] == 1 and d["steps"] == 6 and d["higher_is_better"] is True
    assert d["value"] > 100 and "workload" in d["config"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in d["roofline"], k
    assert d["verified"] is True and d["verified_images"] >= 2 * 6 + 2   # every lane's two images + the serial lane's
    assert d["serial_value"] > 100 and set(d["serial_ms_per_frame"]) == {"rgbbox_1000x1000", "irreg_1000x1000"}
    rf = d["roofline"]
    if rf.get("stale_pmc"):
        # profiles/pmc.json was measured on other kernel sources: bench.py must refuse it, not quote it
        assert rf["achieved"] is None and rf["frac"] is None
    else:
        assert rf["bound"] == "valu_issue" and 0.0 < rf["frac"] <= 1.0
        assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
        assert 0.0 < rf["mix"]["valu_pipe_busy"] <= 1.05


def test_bench_line_two_ranks_sharing_the_gpu():
    """The driver's N > 1 command line (torch.distributed.run, one rank per GPU) with both ranks on cuda:0
    (RT_SHARE_GPU=1: gloo, host-staged gather): the sharded steps are verified against the oracle's checksums and the
    line carries the irreg 4000x4000 sub-record north_star states its scaling target on."""
    import json
    import sys
    env = dict(os.environ, RT_SHARE_GPU="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29577", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
                          "--exchange", "gather"],
                         capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[:500]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["verified"] is True and d["scaling"] == "strong" and d["gather_mode"].startswith("gather")
    r = d["irreg_4000"]
    assert r["verified"] is True and r["ms_per_frame"] > 0 and r["Mray_s"] > 0
    assert r["render_us_per_rank"]["slowest"] >= r["render_us_per_rank"]["fastest"] > 0 and r["gather_and_assemble_us_rank0"] > 0
    assert r["batch"]["ms_per_frame"] > 0 and r["batch"]["frames_per_launch"] == r["frames"]


def test_bench_launches_its_own_ranks():
    """Plain `python bench.py --gpus 2` (no launcher, no WORLD_SIZE): bench.py re-executes itself under
    torch.distributed.run with two ranks (here both on cuda:0, RT_SHARE_GPU=1) and the line says n_gpus == 2; the
    framebuffer exchange is the direct-store one (rank 1 maps rank 0's image through the IPC entry points and stores its
    rows into it), checked against the oracle's checksums before and after the timed region."""
    import json
    import sys
    env = dict(os.environ, RT_SHARE_GPU="1")
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1"],
                         capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[:500]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["verified"] is True and d["gather_mode"].startswith("direct-store"), d["gather_mode"]
    assert d["irreg_4000"]["verified"] is True
    # ... the irreg 4000 x 4000 record through BOTH exchanges in the one run (each verified), the prediction of tools/scale_prediction.py next to them
    assert set(d["irreg_4000"]["by_exchange"]) == {"direct", "gather"} and d["irreg_4000"]["exchange"] == "direct"
    assert all(r["ms_per_frame"] > 0 and r["batch"]["ms_per_frame"] > 0 for r in d["irreg_4000"]["by_exchange"].values())
    assert d["irreg_4000"]["predicted"]["source"].startswith("profiles/r") and "irreg_4000_one_frame_direct" in d["irreg_4000"]["predicted"]


def test_bench_refuses_more_gpus_than_there_are():
    """--gpus N with fewer than N devices is an error, not a smaller run under the wrong label."""
    import sys
    env = dict(os.environ)
    env.pop("RT_SHARE_GPU", None)
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--steps", "2", "--warmup", "1"],
                         capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert out.returncode != 0 and not out.stdout.strip()
    assert "--gpus 64" in out.stderr


def test_bench_refuses_wrong_pixels(tmp_path):
    """The verification of the timed launches must be able to fail: with a wrong expected checksum
    bench.py exits non-zero and prints no JSON line."""
    import sys
    code = ("import sys, bench; bench.FRAME_CHECKSUM[('irreg', 1000, 1000)] ^= 1; "
            "sys.argv = ['bench.py', '--steps', '4', '--warmup', '1', '--no-cpu-baseline', '--no-serial-extra']; bench.main()")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode != 0
    assert "VERIFICATION FAILED" in out.stderr and not out.stdout.strip()


def test_bench_configuration_is_bit_exact(R):
    """The exact configuration bench.py times -- 1000x1000, grid_div=4, deep_class=0, many contexts on their own
    streams with frames in flight -- against the oracle's checksums, every lane, every frame."""
    import torch
    import bench
    n = 12
    streams = [torch.cuda.Stream() for _ in range(n)]
    cks = bench.Checksummer(torch.device("cuda", 0))
    lanes = []
    for st in streams:
        ctx = R.Context(0, st.cuda_stream)
        ctx.set_option("grid_div", 4)
        ctx.set_option("deep_class", 0)
        ps = [(s, R.prepare_scene(1000, 1000, ctx.scene(s)), torch.full((1000, 1000), 0x5a5a5a5a, dtype=torch.int32, device="cuda"))
              for s in ("rgbbox", "irreg")]
        lanes.append((ctx, ps))
    for rep in range(3):           # first frame records the tile order, later ones use it; all overlapped
        for ctx, ps in lanes:
            for s, p, img in ps:
                R.render_into(img.data_ptr(), 1000, 1000, p)
    torch.cuda.synchronize()
    for ctx, ps in lanes:
        for s, p, img in ps:
            assert cks(img) == bench.FRAME_CHECKSUM[(s, 1000, 1000)], s
    for ctx, ps in lanes:
        for s, p, img in ps:
            p.free()
        ctx.close()


# ---------------------------------------------------------------- multi-rank on one GPU ---
def _rank_worker(rank, world, port, scene, h, w, q):
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from raytracers_amd.dist import HipPartRenderer, ShardedRenderer, ShardedStep
        torch.cuda.set_device(0)
        pr = HipPartRenderer(scene, h, w, "cuda:0")
        sr = ShardedRenderer(pr, h, w, device="cuda:0")
        for _ in range(3):
            img = sr.render()
        # a step of two frames of different sizes: one gather, strided placement
        pr2 = HipPartRenderer("rgbbox", 77, 96, "cuda:0")
        st = ShardedStep([(pr, h, w), (pr2, 77, 96)], device="cuda:0")
        for _ in range(2):
            imgs = st.render()
        # the same step with the direct-store exchange: rank 0's image buffer mapped into the other processes (IPC), every
        # rank's kernels store their rows into it; then a batch of three frames per scene the same way
        sd = ShardedStep([(pr, h, w), (pr2, 77, 96)], device="cuda:0", exchange="direct")
        assert sd.exchange_mode == "direct", sd.exchange_note
        if rank == 0:
            for im in sd.images:
                im.fill_(-5)
        for _ in range(2):
            dimgs = sd.render()
        sb = ShardedStep([(pr, h, w), (pr2, 77, 96)], device="cuda:0", exchange="direct", nbatch=3)
        bimgs = sb.render()
        torch.cuda.synchronize()
        if rank == 0:
            assert all(bool((b[f] == d).all()) for b, d in zip(bimgs, dimgs) for f in range(3))
            q.put((img.cpu().numpy().copy(), [i.cpu().numpy().copy() for i in imgs] + [i.cpu().numpy().copy() for i in dimgs]))
        sb.close()
        sd.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_multi_rank_render_sharing_one_gpu(world):
    """Several ranks (processes) share cuda:0: each renders its cyclic row tiles with the HIP
    library, rank 0 gathers (gloo, host-staged -- the one-GPU stand-in for RCCL) and assembles
    with rt_place_parts.  The image must equal the single-GPU / oracle image."""
    import torch.multiprocessing as mp
    h, w, scene = 171, 120, "irreg"
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    port = 29600 + (os.getpid() % 1000) + world
    procs = [ctxm.Process(target=_rank_worker, args=(r, world, port, scene, h, w, q)) for r in range(world)]
    for p in procs:
        p.start()
    img, step_imgs = q.get(timeout=300)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    want, _ = _oracle(scene).render(h, w)
    assert int((img != want).sum()) == 0
    assert int((step_imgs[0] != want).sum()) == 0
    want2, _ = _oracle("rgbbox").render(77, 96)
    assert int((step_imgs[1] != want2).sum()) == 0
    assert int((step_imgs[2] != want).sum()) == 0 and int((step_imgs[3] != want2).sum()) == 0   # the direct-store exchange
