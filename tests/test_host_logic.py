"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol the headers
declare, refuses to run without a HIP device, and the host logic (row-tile partition, host
BVH build + the persistent kernel's wave state machine, via tools/wavesim) agrees with the
oracle."""
import os
import re
import subprocess

import numpy as np
import pytest

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b((?:rt|futhark)_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from raytracers_amd import _lib
    rt = _declared("rt_mi355x.h")
    fut = _declared("ray.h")
    assert len(rt) >= 25 and len(fut) >= 18
    for name in rt + fut:
        assert hasattr(_lib.lib, name), name
    # and the Python loader's own lists are complete
    assert sorted(_lib.RT_SYMBOLS) == rt
    assert sorted(_lib.FUTHARK_SYMBOLS) == fut


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import raytracers_amd as R
    with pytest.raises(R.RtError):
        R.Context()


def test_product_never_touches_the_oracle():
    """The product path must not import, link or call anything under oracle/."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "raytracers_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".h", ".hip")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.lower().replace("never touches the oracle", ""), os.path.join(dirpath, f)
    out = subprocess.run(["ldd", os.path.join(ROOT, "raytracers_amd", "libray_mi355x.so")], capture_output=True,
                         text=True).stdout
    assert "liboracle" not in out


def test_part_rows_partition_is_exact():
    from raytracers_amd import api
    from raytracers_amd.dist import tile_rows
    for h in (1, 7, 8, 9, 64, 100, 1000, 1001):
        for nparts in (1, 2, 3, 8):
            rows = [tile_rows(h, p, nparts) for p in range(nparts)]
            assert sorted(np.concatenate(rows).tolist()) == list(range(h))
            for p in range(nparts):
                assert api.part_rows(h, p, nparts) == len(rows[p])


@pytest.mark.parametrize("scene,h,w", [("rgbbox", 64, 96), ("irreg", 80, 56), ("floor:7:42", 33, 47)])
def test_wave_state_machine_matches_oracle(scene, h, w):
    """tools/wavesim runs the persistent kernel's scheduling with the product's host BVH and
    lane_core.h on 64 emulated lanes; its pixels and work counts must equal the oracle's."""
    exe = os.path.join(ROOT, "build", "wavesim")
    if not os.path.exists(exe):
        subprocess.run(["make", "-s", "build/wavesim"], cwd=ROOT, check=True)
    out = subprocess.run([exe, scene, str(h), str(w), "24", "24", "8", "7"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    if scene.startswith("floor"):
        _, n, k = scene.split(":")
        sc = O.OracleScene("floor", n=int(n), k=float(k))
    else:
        sc = O.OracleScene(scene)
    px, cnt = sc.render(h, w)
    want = "%08x" % O.checksum(px)
    lines = out.stdout.splitlines()
    simple = [l for l in lines if l.startswith("simple:")][0]
    pers = [l for l in lines if l.startswith("persistent:")][0]
    for line in (simple, pers):
        assert f"checksum {want}" in line, line
        assert f"rays {cnt['rays']} box {cnt['box_tests']} sphere {cnt['leaf_tests']}" in line, line
    assert "diff_vs_simple 0" in pers


def test_bench_checksum_arithmetic_on_the_cpu():
    """bench.py verifies every timed image with c = c * 31 + pixel evaluated as a wrapping polynomial on the device;
    the same code on CPU tensors must agree with the sequential definition (oracle_lib.checksum) -- also for pixels
    with the sign bit set (poison pattern) and for a size that is not a power of two."""
    import sys
    import torch
    sys.path.insert(0, ROOT)
    import bench
    cks = bench.Checksummer(torch.device("cpu"))
    rng = np.random.default_rng(7)
    for shape in ((1, 1), (37, 53), (200, 200)):
        px = rng.integers(-2**31, 2**31 - 1, size=shape, dtype=np.int64).astype(np.int32)
        assert cks(torch.from_numpy(px)) == O.checksum(px)
    px, _ = O.OracleScene("rgbbox").render(200, 200)
    assert "%08x" % cks(torch.from_numpy(px)) == "9082f119" == "%08x" % O.checksum(px)   # SURVEY.md 8c's rgbbox 200x200


def test_tile_queue_protocol_on_the_cpu():
    """tools/queue_check drives the pooled kernel's ticket arithmetic (the __host__ __device__ functions of
    rt_device.hpp: shards = strips of tile columns, deep-tile pieces, multi-tile tickets, static first tickets,
    stealing from the other shards' counters) with emulated waves in random interleavings: every pixel slot of
    every tile of every frame must be handed out exactly once."""
    exe = os.path.join(ROOT, "build", "queue_check")
    if not os.path.exists(exe):
        subprocess.run(["make", "-s", "build/queue_check"], cwd=ROOT, check=True)
    for seed in (1, 2):
        out = subprocess.run([exe, "1500", str(seed)], capture_output=True, text=True)
        assert out.returncode == 0, out.stdout + out.stderr
        assert "random cases" in out.stdout and "passed" in out.stdout


def test_ray_donation_protocol_on_the_cpu():
    """tools/donate_check plays the mailbox protocol of the pooled kernel's DONATE instantiation (waves that have left the
    loop offer themselves in a workgroup word; waves that cannot refill give them one ray each through their idle ray
    tables) as a model, one LDS operation per step, in random interleavings: every ray is finished exactly once, no inbox
    is overwritten or read half-written, no ray goes to a wave that has ended, every wave ends.  The checker must also
    catch the two orderings the kernel's comments insist on when they are broken (count before flag; ray before flag)."""
    exe = os.path.join(ROOT, "build", "donate_check")
    subprocess.run(["make", "-s", "build/donate_check"], cwd=ROOT, check=True)
    for seed in (1, 2, 3):
        out = subprocess.run([exe, "20000", str(seed)], capture_output=True, text=True)
        assert out.returncode == 0 and "protocol holds" in out.stdout, out.stdout + out.stderr
    for mutate in (1, 2):
        out = subprocess.run([exe, "20000", "1", str(mutate)], capture_output=True, text=True)
        assert out.returncode != 0 and "FAILED" in out.stdout, (mutate, out.stdout)


@pytest.mark.parametrize("scene,size", [("rgbbox", "96"), ("irreg", "96"), ("23", "64")])
def test_treelet_numbering_and_masks_on_the_cpu(scene, size):
    """tools/treelet_probe redoes every fold of a small frame from the 64-byte records of the traversal copy cut into
    treelets of 2 (the shipped cut) .. 5 levels, decoding the masks with the solo loop's own helpers (treelet.h) on emulated
    lanes: each fold must meet exactly the leaves the depth-first walk of the same tree meets."""
    exe = os.path.join(ROOT, "build", "treelet_probe")
    subprocess.run(["make", "-s", "build/treelet_probe"], cwd=ROOT, check=True)
    out = subprocess.run([exe, scene, size, "0"], capture_output=True, text=True)
    assert out.returncode == 0 and "treelet check OK" in out.stdout, out.stdout + out.stderr
    assert out.stdout.count("same leaf sets from the records") == 4


def test_cull_limit_is_conservative_on_the_cpu():
    """tools/cull_bound_check hammers the two inequalities the CULL instantiations rest on (lane_core.h: cull_limit)
    with the product's own binary32 code against __float128: (E1) a computed root lies within 2^-18 (D^2 + r^2) of the sphere,
    (S) a sphere's own box is never culled by the limit computed from that sphere's own root -- random and adversarial pairs
    (grazing, far, nearly axis-parallel, origin inside the sphere).  With the margin scaled down 1000 x the checker must FIND
    violations: it is not vacuous."""
    exe = os.path.join(ROOT, "build", "cull_bound_check")
    subprocess.run(["make", "-s", "build/cull_bound_check"], cwd=ROOT, check=True)
    for seed in (1, 2):
        out = subprocess.run([exe, "12", str(seed)], capture_output=True, text=True)
        assert out.returncode == 0 and "E1 violations 0 " in out.stdout and "safety violations 0" in out.stdout, out.stdout + out.stderr
    out = subprocess.run([exe, "12", "1", "0.001"], capture_output=True, text=True)
    assert out.returncode != 0 and "safety violations 0" not in out.stdout, out.stdout


@pytest.mark.parametrize("scene,h,w", [("irreg", 120, 160), ("floor:40:240", 96, 96), ("rgbbox", 96, 96)])
def test_culled_pooled_loop_keeps_the_pixels_on_the_cpu(scene, h, w):
    """tools/cull_pooled plays the pooled kernel's loop (shared LIFO, deferred sphere tests, refill in place) on emulated lanes,
    without and with the product's culling rule: same checksum as the oracle, never more tests."""
    exe = os.path.join(ROOT, "build", "cull_pooled")
    subprocess.run(["make", "-s", "build/cull_pooled"], cwd=ROOT, check=True)
    out = subprocess.run([exe, scene, str(h), str(w), "64", "0", "0", "0", "40", "32", "1"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    if scene.startswith("floor"):
        _, n, k = scene.split(":")
        sc = O.OracleScene("floor", n=int(n), k=float(k))
    else:
        sc = O.OracleScene(scene)
    px, cnt = sc.render(h, w)
    modes = [l for l in out.stdout.splitlines() if l.startswith("mode ")]
    assert len(modes) == 2
    got = [dict(re.findall(r"(checksum|diff|rays|box|sphere) ([0-9a-f]+)", l)) for l in modes]
    for g in got:
        assert g["checksum"] == "%08x" % O.checksum(px) and g["diff"] == "0" and int(g["rays"]) == cnt["rays"]
    assert int(got[0]["box"]) == cnt["box_tests"] and int(got[0]["sphere"]) == cnt["leaf_tests"]
    assert int(got[1]["box"]) <= cnt["box_tests"] and int(got[1]["sphere"]) <= cnt["leaf_tests"]


def _box_stack_high_water(height, child_inner, passes, shade, look_max, ops, rng):
    """The pooled loop's box stack as a list of node depths, played by an adversary: `shade()` roots (<= 64) are pushed whenever fewer
    than look_max (<= 64) items are left, every other operation pops the <= 64 newest items (BOX; BOX2 -- two levels at once -- when at
    most 32 are left, as the kernel does) and pushes the inner children that `passes()`; a node at depth height - 1 has leaf children."""
    stack, high = [], 0
    for _ in range(ops):
        if len(stack) < look_max and (not stack or rng.random() < 0.5):
            stack += [0] * shade()
        elif stack:
            k = min(64, len(stack))
            popped, stack = stack[-k:], stack[:-k]
            levels = 2 if k <= 32 else 1
            for _ in range(levels):
                popped = [d + 1 for d in popped for _ in range(2) if d + 1 < height and child_inner() and passes()]
            stack += popped
        high = max(high, len(stack))
    return high


@pytest.mark.parametrize("height", [1, 2, 3, 7, 15, 22])
def test_box_stack_bound_of_the_pooled_loop(height):
    """the box stack holds at most 64 H + 63 items (H = levels of inner nodes) -- the capacity of the twenty-wave shape,
    64 (H + 2) dwords, rests on it (api.cpp: make_plan).  The greedy adversary (a complete tree, every box passes, 64 roots whenever the look
    allows) reaches the bound exactly for H >= 2; random adversaries (ragged trees, partial passes, any look_max) stay below it."""
    rng = np.random.default_rng(height)
    bound = 64 * height + 63
    class Always:                 # (the greedy adversary shades whenever the look allows: 63 roots first, 64 on top of them, 64 from then on)
        def random(self):
            return 0.0
    first = [63]
    greedy = _box_stack_high_water(height, lambda: True, lambda: True, lambda: first.pop() if first else 64, 64, 40 * height + 200, Always())
    assert greedy <= max(bound, 127)
    if height >= 2:
        assert greedy == bound
    for trial in range(60):
        p_inner, p_pass = rng.choice([1.0, 0.95, 0.8, 0.5]), rng.choice([1.0, 0.9, 0.6])
        look = int(rng.choice([1, 8, 16, 32, 64]))
        high = _box_stack_high_water(height, lambda: rng.random() < p_inner, lambda: rng.random() < p_pass,
                                     lambda: int(rng.integers(1, 65)), look, 400, rng)
        assert high <= max(bound, 127), (trial, p_inner, p_pass, look, high)


@pytest.mark.parametrize("height,cap", [(7, 192), (19, 192), (19, 1088), (30, 1088), (40, 448)])
def test_spilling_box_stack_discipline(height, cap):
    """The SPILL kernels' bookkeeping (render_kernels.hip) as a model: an LDS stack of `cap` dwords, the oldest half (whole
    batches of 64) moved to memory before a full BOX operation that finds more than cap - 64 items, the newest spilled chunk (<= 128) fetched
    back when the LDS part is empty, no SHADE while anything is spilled.  Against a greedy adversary (complete tree, every box passes) and
    random ones: the LDS part never exceeds its capacity, LDS + memory never exceed the LIFO's bound 64 H + 63, nothing is lost."""
    bound = max(64 * height + 63, 127)
    for trial in range(12):
        rng = np.random.default_rng(1000 * height + trial)
        p_inner, p_pass = (1.0, 1.0) if trial == 0 else (rng.choice([1.0, 0.97, 0.9]), rng.choice([1.0, 0.95, 0.8]))
        lds, mem, pushed, popped, high_all, budget = [], [], 0, 0, 0, 40 * height + 400
        while budget > 0 or lds or mem:
            draining = budget <= 0                                # (the adversary's time is up: no more rays, no box passes -- the stack drains)
            if not lds and mem:                                   # loop top: the newest chunk comes back
                n = min(len(mem), 128)
                lds, mem = mem[-n:], mem[:-n]
            if not draining and not mem and len(lds) < 64 and (not lds or rng.random() < 0.5):   # SHADE (look_max = 64)
                k = 63 if pushed == 0 else (int(rng.integers(1, 65)) if trial else 64)
                lds += [0] * k
                pushed += k
            elif lds:
                if len(lds) >= 64 and len(lds) > cap - 64:        # a full batch may push 128 behind its 64
                    S = (len(lds) >> 1) & ~63
                    assert S >= 64
                    mem += lds[:S]
                    lds = lds[S:]
                k = min(64, len(lds))
                items, lds = lds[-k:], lds[:-k]
                popped += k
                for _ in range(2 if k <= 32 else 1):               # BOX2: two levels at once (the level between never reaches the stack)
                    items = [d + 1 for d in items for _ in range(2)
                             if not draining and d + 1 < height and rng.random() < p_inner and rng.random() < p_pass]
                lds += items
                pushed += len(items)
            budget -= 1
            high_all = max(high_all, len(lds) + len(mem))
            assert len(lds) <= cap, (trial, len(lds))
        assert high_all <= bound, (trial, high_all, bound)
        assert pushed == popped, (trial, pushed, popped)          # every item left the stack exactly once
        if trial == 0 and cap < 64 * height:
            assert high_all > cap                                  # (the greedy adversary does make it spill)


def test_reference_harness_builds_against_our_header():
    """/root/reference/futhark/main.c must compile and link unmodified (build container only)."""
    if not os.path.exists("/root/reference/futhark/main.c"):
        pytest.skip("reference tree not present")
    subprocess.run(["sh", os.path.join(ROOT, "oracle", "build_ref.sh")], check=True)
    assert os.path.exists(os.path.join(ROOT, "oracle", "_ref", "futhark_main"))
