#!/usr/bin/env python3
"""bench.py -- the render hot path on N MI355X GPUs of one node.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one frame of EACH scene of the headline metric (BASELINE.json: "Mray/s
(primary+secondary) on rgbbox & irreg 1000x1000"): render rgbbox 1000x1000, then irreg
1000x1000, scene data (BVH, spheres, camera) already resident in HBM.  With N > 1 every
frame is cut into cyclic 8-row tiles across the ranks (strong scaling: total work fixed) and
the framebuffers of a step are gathered to rank 0 over RCCL inside the timed region (ONE
gather per step: a rank's rows of both frames travel in one send buffer).

Steps are independent frames, so up to --frames-in-flight of them (default 32) are enqueued
on separate HIP streams, each with its own context and framebuffers: a 1000x1000 frame ends
with a long tail in which a handful of 50-bounce pixels keep a few waves busy (the frame's
latency floor), and the next frames' bulk work fills the otherwise idle machine.  All K
steps complete inside the barrier/synchronize bracket.  The line also reports the strictly
serial figures (one frame at a time, `serial`), measured right after the timed region.

value = rays of all K steps / wall time (max over ranks), in Mray/s; a ray = one objs_hit
call (futhark/ray.fut:130).  Ray and box/sphere-test counts come from an instrumented launch
and are cross-checked against the oracle-derived constants below.

The JSON line also carries
  roofline      for the dominant kernel (pooled_kernel): ALGORITHMIC bytes (32 B per box test +
                16 B per sphere test + 4 B per pixel) of the timed region's launches /
                its wall time, against the 8 TB/s HBM3E peak; per_launch = the dominant launch's
                bytes / its mean duration measured with events on the launch stream inside the timed
                region (stretched by the other frames in flight); valu = the VALU-issue fraction;
  cpu_baseline  the CPU oracle (a port of the reference's Futhark program, OpenMP over rows)
                timed on this box's host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# The frames in flight live on separate HIP streams; ROCm maps streams onto GPU_MAX_HW_QUEUES
# hardware queues (default 4, and torch / RCCL take some), and streams that share a queue
# largely serialise.  Measured on one MI355X (tools/rank_share_probe.py): 8 queues
# 0.69 ms/step, 12-16 queues 0.54-0.60 and bimodal, 20 queues with 32 lanes 0.53 and steady;
# with MORE than ~20 queues actually busy (24+ queues and 24+ lanes) the hardware scheduler
# oversubscribes and a step takes 0.7-1.2 ms.  Must be set before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec (MI355X_MICROARCH.md)

# (rays, box tests, sphere tests) per frame, from the CPU oracle (tests/test_oracle_golden.py
# pins the oracle to the reference's golden images; SURVEY.md 8d lists the same numbers).
FRAME_WORK = {
    ("rgbbox", 1000, 1000): (4022099, 117685724, 25443619),
    ("irreg", 1000, 1000): (1728608, 50741777, 9664717),
    ("irreg", 4000, 4000): (27663974, 812246528, 154642404),
}

WORKLOADS = {
    "rgbbox+irreg-1000": [("rgbbox", 1000, 1000), ("irreg", 1000, 1000)],
    "rgbbox-1000": [("rgbbox", 1000, 1000)],
    "irreg-1000": [("irreg", 1000, 1000)],
    "irreg-4000": [("irreg", 4000, 4000)],
    "big-2000": [("big", 2000, 2000)],
}


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def bytes_alg(box, sph, h, w):
    return 32 * box + 16 * sph + 4 * h * w


def effective_cpus():
    """CPUs this process may actually use: the affinity mask, capped by the cgroup CPU quota
    (the GPU boxes show 256 hardware threads but grant 16 CPUs' worth of time; 256 OpenMP threads
    on such a quota run 4-5x slower than 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]          # cgroup v2
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())         # cgroup v1
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, -(-quota // period)))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(frames, budget_s=12.0):
    """Times the CPU oracle (tests/oracle_lib.py -> oracle/ray_oracle.c) on the same frames."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    scenes = [(O.OracleScene(s), h, w) for s, h, w in frames]
    cores = min(effective_cpus(), int(O.lib().orc_num_threads()))
    rays = 0
    reps = 0
    t0 = time.perf_counter()
    while True:
        for sc, h, w in scenes:
            _, cnt = sc.render(h, w, threads=cores)
            rays += cnt["rays"]
        reps += 1
        if time.perf_counter() - t0 > budget_s or reps >= 200:
            break
    dt = time.perf_counter() - t0
    out = {"value": rays / dt / 1e6, "unit": "Mray/s", "cores": cores, "kind": "port",
           "sample": f"{reps} full frame(s) of each of {'+'.join(f'{s} {w}x{h}' for s, h, w in frames)}, "
                     f"OpenMP dynamic over row chunks on {cores} threads (= the CPUs this process is granted: "
                     f"affinity {len(os.sched_getaffinity(0))}, cgroup quota applied), {dt:.1f} s"}
    # second comparator: the reference's RUST algorithm (different BVH / epsilon, different image;
    # timing only -- oracle/rust_algo_port.c), same frames, same threads
    try:
        rs = [(O.RustAlgoScene(s), h, w) for s, h, w in frames if s in ("rgbbox", "irreg")]
        if rs:
            rrays, rreps, t1 = 0, 0, time.perf_counter()
            while time.perf_counter() - t1 < budget_s / 2 and rreps < 200:
                for sc, h, w in rs:
                    rrays += sc.render(h, w, threads=cores)[1]
                rreps += 1
            rdt = time.perf_counter() - t1
            out["rust_algorithm"] = {"value": rrays / rdt / 1e6, "unit": "Mray/s", "cores": cores, "kind": "port",
                                     "sample": f"{rreps} frames of each scene, {rdt:.1f} s",
                                     "note": "C restatement of rust/src/lib.rs (median-split BVH, eps 0.001): timing only"}
    except Exception as e:   # the baseline must never take the bench line down
        out["rust_algorithm"] = {"error": str(e)}
    return out


def main():
    # Only the JSON line may appear on stdout: RCCL (and others) print banners to the C-level
    # stdout, so fd 1 points at stderr for the whole run and is restored for the final print.
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="rgbbox+irreg-1000", choices=sorted(WORKLOADS))
    ap.add_argument("--variant", type=int, default=0, help="0 auto (pooled), 1 pixel, 2 persistent, 3 pooled")
    ap.add_argument("--opt", action="append", default=[], help="kernel knob name=value (repeatable)")
    ap.add_argument("--frames-in-flight", type=int, default=32, help="independent steps enqueued concurrently (streams)")
    ap.add_argument("--event-every", type=int, default=1,
                    help="bracket the launches of every n-th timed step with HIP events (kernel duration samples)")
    ap.add_argument("--no-serial-extra", action="store_true", help="skip the extra serial (one frame at a time) region")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        log(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}; using WORLD_SIZE")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback")
    # RT_SHARE_GPU=1: every rank uses cuda:0 and the gloo backend (host-staged gather) -- a test
    # mode that runs the multi-rank control flow on a one-GPU box; never the measured configuration
    share_gpu = bool(os.environ.get("RT_SHARE_GPU"))
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    use_pg = world > 1 or bool(os.environ.get("RT_FORCE_GATHER"))   # the latter: exercise RCCL on one GPU
    if use_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    from raytracers_amd.dist import HipPartRenderer, ShardedStep

    frames = WORKLOADS[args.workload]
    opts = dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in args.opt)
    # Frames in flight and launch size, from tools/rank_share_probe.py (one rank's share of a step at
    # world sizes 1..8, measured on one MI355X with 20 hardware queues): 32 lanes are steady at
    # every world size (8 or 16 lanes are bimodal, e.g. 0.50 / 0.56 ms per step), and as a rank's
    # share of a frame shrinks it takes that many frames in flight to cover a frame's latency
    # floor (its longest bounce chain): 505 / 252 / 142 / 87 us per step at 1 / 2 / 4 / 8 ranks.
    S = max(1, args.frames_in_flight)
    # With many frames in flight a launch need not fill the machine by itself: an eighth of the
    # persistent workgroups per launch gives longer-lived, better-filled waves; the longer tail is
    # hidden by the other frames.  One frame at a time keeps the library default.
    opts_pipe = dict(opts)
    if S >= 4 and args.variant in (0, 3):
        opts_pipe.setdefault("grid_div", 8)
        # dedicated waves for the deepest tiles shorten ONE frame's tail (-7 %), which overlapped
        # frames hide anyway; they cost 2-3 % of throughput here
        opts_pipe.setdefault("deep_class", 0)
    # one "lane" per frame in flight: its own HIP stream, contexts, prepared scenes, framebuffers
    streams = [torch.cuda.current_stream(device)] if S == 1 else [torch.cuda.Stream(device) for _ in range(S)]

    class Lane:
        """the renderers of one frame in flight + the step (render all, one gather, assemble)"""
        def __init__(self, o):
            self.prs = [HipPartRenderer(scene, h, w, device, variant=args.variant, options=o) for scene, h, w in frames]
            self.step = ShardedStep([(pr, h, w) for pr, (_, h, w) in zip(self.prs, frames)], device)

    def make_lane(o):
        return Lane(o)

    lanes = []
    for st in streams:
        with torch.cuda.stream(st):
            lanes.append(make_lane(opts_pipe))
    serial_lane = make_lane(opts) if (S > 1 and not args.no_serial_extra) else None   # on the default stream
    torch.cuda.synchronize()
    renderers = [(scene, h, w, pr) for (scene, h, w), pr in zip(frames, lanes[0].prs)]

    # work per frame: instrumented launch (rank 0 is enough), checked against the oracle table
    work = {}
    for scene, h, w, pr in renderers:
        st = pr.prepared.stats()
        got = (st["rays"], st["box_tests"], st["leaf_tests"])
        want = FRAME_WORK.get((scene, h, w))
        if want is not None and got != want:
            raise SystemExit(f"work counters of {scene} {w}x{h} differ from the oracle's: {got} vs {want}")
        work[(scene, h, w)] = got

    def step(k, events=None, nlanes=S):
        if nlanes == 0:       # the one-frame-at-a-time lane (library defaults, default stream)
            serial_lane.step.render(events)
            return
        li = k % nlanes
        with torch.cuda.stream(streams[li]):
            lanes[li].step.render(events)

    def fence():
        torch.cuda.synchronize()
        if use_pg:
            dist.barrier()
            torch.cuda.synchronize()

    def timed(nsteps, nlanes):
        every = max(1, args.event_every)
        ev = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in frames]
              if k % every == 0 else None for k in range(nsteps)]
        fence()
        t0 = time.perf_counter()
        for k in range(nsteps):
            step(k, ev[k], nlanes)
        fence()
        dt = time.perf_counter() - t0
        if use_pg:
            t = torch.tensor([dt], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        # per-launch durations on this rank: events recorded on the stream each kernel is launched on
        kms = [float(np.mean([ev[k][i][0].elapsed_time(ev[k][i][1]) for k in range(nsteps) if ev[k] is not None]))
               for i in range(len(frames))]
        return dt, kms

    # Lane set-up: every lane renders its frames twice, which fills the per-view caches (u/v tables;
    # the adaptive tile order is computed once from the first frame's cost record).  Then the W
    # untimed warm-up steps, then exactly K timed steps.
    for k in range(2 * S):
        step(k)
    for k in range(args.warmup):
        step(k)
    elapsed, kern_ms = timed(args.steps, S)
    serial = None
    if serial_lane is not None:
        for k in range(3):
            step(k, None, 0)
        serial = timed(max(10, args.steps // 2), 0)

    if rank == 0:
        rays_step = sum(work[(s, h, w)][0] for s, h, w in frames)
        value = rays_step * args.steps / elapsed / 1e6
        per_scene = {}
        for i, (scene, h, w, _) in enumerate(renderers):
            r, b, s = work[(scene, h, w)]
            ba = bytes_alg(b, s, h, w) / world        # this rank's share of the frame (cyclic tiles)
            per_scene[f"{scene}_{w}x{h}"] = {
                "rays": r, "kernel_ms": kern_ms[i], "Mray_s_kernel": r / world / (kern_ms[i] * 1e-3) / 1e6,
                "alg_bytes_per_launch": ba, "alg_GBs": ba / (kern_ms[i] * 1e-3) / 1e9}
        # the dominant launch = the one that carries the most algorithmic work of the step
        dom = max(range(len(renderers)), key=lambda i: per_scene[f"{frames[i][0]}_{frames[i][2]}x{frames[i][1]}"]["alg_bytes_per_launch"])
        dscene, dh, dw = frames[dom]
        dkey = f"{dscene}_{dw}x{dh}"
        achieved = per_scene[dkey]["alg_GBs"]
        out = {
            "metric": "Mray/s (primary+secondary) on rgbbox & irreg 1000x1000" if args.workload == "rgbbox+irreg-1000"
                      else f"Mray/s (primary+secondary) on {args.workload}",
            "value": value, "unit": "Mray/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic (the reference's procedural scenes)",
            "config": {"workload": " + ".join(f"{s} {w}x{h}" for s, h, w in frames) + ", max_depth 50, one frame of each per step",
                       "kernel": {0: "auto (pooled)", 1: "pixel", 2: "persistent", 3: "pooled"}[args.variant],
                       "options": opts_pipe, "frames_in_flight": S,
                       "partition": f"cyclic 8-row tiles over {world} GPU(s), one RCCL gather to rank 0 per step"
                                    + (" [RT_SHARE_GPU test mode: ranks share cuda:0, gloo host-staged gather]" if share_gpu else "")},
            "roofline": {"bound": "hbm",
                         "kernel": {0: "pooled_kernel", 1: "pixel_kernel", 2: "persistent_kernel", 3: "pooled_kernel"}[args.variant]
                                   + " (the launches of " + " and ".join(f"{s} {w}x{h}" for s, h, w in frames) + ")",
                         # launches of up to frames_in_flight frames share the GPU, so the kernel's rate is
                         # what all of them together get through per second
                         "achieved": sum(per_scene[k]["alg_bytes_per_launch"] for k in per_scene) * args.steps / elapsed / 1e9,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": sum(per_scene[k]["alg_bytes_per_launch"] for k in per_scene) * args.steps / elapsed / 1e9 / HBM_PEAK_GBS,
                         "traffic": None,
                         "per_launch": {"kernel": f"{dscene} {dw}x{dh}", "alg_bytes": per_scene[dkey]["alg_bytes_per_launch"],
                                        "avg_launch_ms": kern_ms[dom], "achieved": achieved, "frac": achieved / HBM_PEAK_GBS,
                                        "avg_launches_in_flight": sum(kern_ms) * args.steps / (elapsed * 1e3),
                                        "note": "HIP events around each launch on its own stream inside the timed region; "
                                                "the launches in flight stretch one another"},
                         "note": "achieved = ALGORITHMIC bytes (32 B/box test + 16 B/sphere test + 4 B/pixel) of all launches "
                                 "in the timed region / its wall time; per_launch = the dominant launch's bytes / its mean "
                                 "event-bracketed duration while the other frames in flight share the GPU; "
                                 "frac_one_frame_at_a_time = the same launch alone on the GPU.  The scene is LDS/L2 "
                                 "resident: measured HBM traffic (traffic, bytes per step from the PMC passes) is a few MB, so "
                                 "the nominal HBM roofline can be exceeded; what binds is VALU issue (valu)"},
            "per_scene": per_scene,
            "derived_reference": {"futhark_mi100_Mray_s": {"rgbbox": 287.3, "irreg": 216.1},
                                  "note": "README.md:50 render times / oracle ray counts; different hardware"},
        }
        # measured HBM traffic of that launch: PMC counters cannot be collected inside this run; the
        # figure comes from the committed PMC passes (profiles/traffic.json, tools/gpu_pmc.sh)
        try:
            with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
                trj = json.load(f)
            trs = [trj.get(f"pooled_kernel {s} {w}x{h}", {}).get("hbm_bytes") for s, h, w in frames]
            if all(trs) and args.variant in (0, 3) and world == 1:
                out["roofline"]["traffic"] = sum(trs)   # HBM bytes of one step's launches
            # what actually bounds this kernel: VALU issue.  wave-instructions of all launches of a
            # step (PMC SQ_INSTS_VALU at the bench's launch size) / step time, against one VALU
            # wave64 instruction per 4 clocks per SIMD at the 2.4 GHz peak clock
            key = "valu_insts_bench_launch" if opts_pipe.get("grid_div", 1) == 8 else "valu_insts"
            vi = [trj.get(f"pooled_kernel {s} {w}x{h}", {}).get(key) for s, h, w in frames]
            if all(vi) and args.variant in (0, 3) and world == 1:
                peak = 256 * 4 * 2.4e9 / 4
                ach = sum(vi) * args.steps / elapsed
                out["roofline"]["valu"] = {"bound": "valu issue", "insts_per_step": sum(vi), "achieved": ach / 1e9,
                                           "peak": peak / 1e9, "unit": "G wave-instr/s", "frac": ach / peak,
                                           "note": "SQ_INSTS_VALU per launch from the committed PMC pass (profiles/"
                                                   "traffic.json) x launches per step / measured step time"}
        except (OSError, ValueError, KeyError):
            pass
        if serial is not None:
            sdt, skms = serial
            out["roofline"]["frac_one_frame_at_a_time"] = (bytes_alg(work[frames[dom]][1], work[frames[dom]][2], dh, dw) / world
                                                          / (skms[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS)
            nser = max(10, args.steps // 2)
            out["serial"] = {
                "note": "one frame at a time on one stream (no frames overlapped), measured after the timed region",
                "value": rays_step * nser / sdt / 1e6, "ms_per_step": sdt / nser * 1e3,
                "kernel_ms": {f"{sc}_{w}x{h}": skms[i] for i, (sc, h, w) in enumerate(frames)},
                "roofline_frac": {f"{sc}_{w}x{h}": bytes_alg(work[(sc, h, w)][1], work[(sc, h, w)][2], h, w) / world
                                  / (skms[i] * 1e-3) / 1e9 / HBM_PEAK_GBS for i, (sc, h, w) in enumerate(frames)}}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(frames)
        result_line = json.dumps(out)
    else:
        result_line = None
    if use_pg:
        dist.barrier()
        dist.destroy_process_group()
    # RCCL's banner sits in the C library's stdout buffer (a pipe is fully buffered) and would
    # otherwise surface AFTER the JSON line at exit: push it out while fd 1 still is stderr
    sys.stdout.flush()
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass
    os.dup2(saved_stdout, 1)
    if result_line is not None:
        print(result_line, flush=True)


if __name__ == "__main__":
    main()
