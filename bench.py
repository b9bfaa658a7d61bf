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

Steps are independent frames, so the K timed steps are handed to the library's throughput entry,
rt_render_batch: ONE launch per scene renders that scene's frame of all K steps (the scenes' launches on
separate streams; on N > 1 GPUs each has its own framebuffer gather) -- the
persistent waves run straight across frame boundaries and the launch's fill and drain are paid once.
All K steps complete inside the barrier/synchronize bracket.  (--protocol lanes is the round-1
protocol: one launch per frame, up to --frames-in-flight steps overlapped on separate HIP streams.)
`value` is that throughput; the reference's own protocol -- one render + sync at a time
(futhark/main.c:107-124) -- is reported next to it as `serial_value` / `serial_ms_per_frame`, and the
>= 10x-MI100 target check (`targets`) is quoted on the serial figures.

The K-step bracket is timed --repeats times (default 5; each behind its own warm-up passes and poison fill, each verified): steps /
ms_per_step / value describe the MEDIAN bracket, brackets_ms / value_min / value_max the rest (--repeats 1: one bracket, no such keys).

Every timed launch is VERIFIED: the framebuffers are poisoned before the timed region and,
after its closing fence, the images of every lane are checksummed on the device
(c = c * 31 + pixel) and compared with the oracle's checksums.  A mismatch fails
the run; the line carries "verified": true.

value = rays of all K steps / wall time (max over ranks), in Mray/s; a ray = one objs_hit
call (futhark/ray.fut:130).  Ray and box/sphere-test counts come from an instrumented launch
and are cross-checked against the oracle-derived constants below.

roofline: the kernel's binding resource is VALU issue, not HBM (the scenes are LDS / L2
resident: a frame moves ~10 MB to and from HBM).  `achieved` = VALU wave-instructions per
second over the timed region (SQ_INSTS_VALU per launch from the committed PMC passes,
profiles/pmc.json, which carries the hash of the kernel sources it was measured on -- a stale
file is refused), `peak` = the rate tools/issue_peak.hip measured on this GPU model for
independent v_add_f32 / v_mul_f32 / v_fma_f32 streams (profiles/issue_peak.json); `mix` prices
the kernel's own instruction mix with the per-class costs of the same microbenchmark (selects,
min/max, compares and integer address arithmetic issue at about half the v_add rate).  The
SURVEY 8d algorithmic-bytes figure is kept under `alg_bytes` (against the LDS pipe that actually
serves those bytes, and against the 8 TB/s HBM peak for reference) together with the measured HBM
`traffic`.
"""
import argparse
import hashlib
import contextlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# The frames in flight live on separate HIP streams; ROCm maps streams onto GPU_MAX_HW_QUEUES
# hardware queues (default 4, and torch / RCCL take some), and streams that share a queue
# largely serialise.  With MORE than ~20 queues actually busy the hardware scheduler
# oversubscribes.  Must be set before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")
# one process per GPU: RCCL and the IPC mapping of rank 0's images (the direct-store exchange) need the dmabuf IPC mode on this driver
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec (MI355X_MICROARCH.md)
NUM_SIMD = 256 * 4

# (rays, box tests, sphere tests) per frame, from the CPU oracle (tests/test_oracle_golden.py
# pins the oracle to the reference's golden images; SURVEY.md 8d lists the same numbers).
FRAME_WORK = {
    ("rgbbox", 1000, 1000): (4022099, 117685724, 25443619),
    ("irreg", 1000, 1000): (1728608, 50741777, 9664717),
    ("irreg", 4000, 4000): (27663974, 812246528, 154642404),
    ("big", 2000, 2000): (6982472, 324712209, 51736177),
}
# c = c * 31 + pixel over the row-major packed pixels (u32 wrap): the CPU oracle's images
# (tests/test_oracle_golden.py::test_checksums_of_the_bench_frames recomputes the 1000x1000 ones)
FRAME_CHECKSUM = {
    ("rgbbox", 1000, 1000): 0xfc0f53a6,
    ("irreg", 1000, 1000): 0xe3b3857c,
    ("irreg", 4000, 4000): 0xdb269d43,
    ("big", 2000, 2000): 0x3a198726,
}
# README.md:50 (Futhark on an MI100): render ms at 1000x1000; north_star wants >= 10x
MI100_RENDER_MS = {"rgbbox": 14.0, "irreg": 8.0}

WORKLOADS = {
    "rgbbox+irreg-1000": [("rgbbox", 1000, 1000), ("irreg", 1000, 1000)],
    "rgbbox-1000": [("rgbbox", 1000, 1000)],
    "irreg-1000": [("irreg", 1000, 1000)],
    "irreg-4000": [("irreg", 4000, 4000)],
    "big-2000": [("big", 2000, 2000)],
}
# (api.cpp: which instantiation renders a launch, the pixel list's model constants and the culling gate live there -- a PMC file measured
# under another policy describes other launches)
KERNEL_SOURCES = ["raytracers_amd/csrc/render_kernels.hip", "raytracers_amd/csrc/lane_core.h",
                  "raytracers_amd/csrc/rt_device.hpp", "raytracers_amd/csrc/treelet.h", "raytracers_amd/csrc/api.cpp"]
# the guide's nominal VALU issue peak: 256 CUs x 4 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction (MI355X_MICROARCH.md)
VALU_NOMINAL_PEAK_G = 256 * 4 * 2.4 / 2


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def bytes_alg(box, sph, h, w):
    return 32 * box + 16 * sph + 4 * h * w


def strip_comments(src):
    """C++ source without comments and with white space collapsed (string and character literals kept as they are): what the
    compiler sees, so that an edited comment does not invalidate counters measured on unchanged code"""
    out, i, n = [], 0, len(src)
    while i < n:
        c = src[i]
        if c == '"' or c == "'":
            j = i + 1
            while j < n and src[j] != c:
                j += 2 if src[j] == "\\" else 1
            out.append(src[i:j + 1])
            i = j + 1
        elif src.startswith("//", i):
            while i < n and src[i] != "\n":
                i += 1
        elif src.startswith("/*", i):
            i = src.find("*/", i + 2)
            i = n if i < 0 else i + 2
            out.append(" ")
        else:
            out.append(c)
            i += 1
    return " ".join("".join(out).split())


def kernel_source_hash(legacy=False):
    """sha256 over the kernel sources (comments and white space removed): profiles/pmc.json records the hash it was measured on.
    legacy=True: over the raw bytes (what files written before round 4's last commit carry)"""
    m = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as f:
            raw = f.read()
        m.update(raw if legacy else strip_comments(raw.decode("utf-8")).encode("utf-8"))
    return m.hexdigest()


def load_scale_prediction(world):
    """what tools/scale_prediction.py predicted for this world size from one-GPU measurements (profiles/r*/scale_prediction.json,
    the latest round's): printed next to the measured irreg 4000x4000 figures so that one multi-GPU run tests the model's two
    assumed constants (25 us per ordering signal, half link rate for 4-byte stores)"""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "scale_prediction.json")))
    if not files:
        return None
    try:
        with open(files[-1]) as f:
            d = json.load(f)
    except (OSError, ValueError):
        return None
    out = {"source": os.path.relpath(files[-1], ROOT)}
    row = (d.get("worlds") or {}).get(str(world)) if isinstance(d, dict) else None
    if not isinstance(row, dict):
        out["note"] = f"no row for world size {world}"
        return out
    for key in ("irreg_4000_one_frame", "irreg_4000_one_frame_direct", "irreg_4000_batch_of_6", "irreg_4000_batch_of_6_direct",
                "headline_1000", "headline_1000_direct"):
        if key in row:
            out[key] = {k: row[key][k] for k in ("us_per_frame", "us_per_step", "speedup_vs_1") if k in row[key]}
    return out


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
                                     "note": "C restatement of rust/src/lib.rs (median-split BVH, eps 0.001): timing only, parity unpinned"}
    except Exception as e:   # the baseline must never take the bench line down
        out["rust_algorithm"] = {"error": str(e)}
    return out


class Checksummer:
    """c = c * 31 + pixel (u32 wrap) of a device image, computed on the device: the polynomial
    sum(pixel_i * 31^(n-1-i)) in wrapping 64-bit arithmetic, low 32 bits."""

    def __init__(self, device):
        self.device = device
        self.w = {}

    def __call__(self, img):
        import torch
        n = img.numel()
        if n not in self.w:
            pw = np.cumprod(np.concatenate([[1], np.full(n - 1, 31, dtype=np.uint64)]).astype(np.uint64))   # wraps mod 2^64
            self.w[n] = torch.from_numpy(pw[::-1].copy().view(np.int64)).to(self.device)
        v = (img.reshape(-1).to(torch.int64) & 0xffffffff) * self.w[n]
        return int(v.sum().item()) & 0xffffffff


def roofline_block(frames, world, variant, key, steps, elapsed, per_scene, work):
    """VALU-issue roofline (+ LDS, HBM side figures) from profiles/pmc.json and profiles/issue_peak.json."""
    alg = sum(per_scene[k]["alg_bytes_per_frame"] for k in per_scene) * steps / elapsed / 1e9   # (per frame x steps)
    rf = {"bound": "valu_issue", "kernel": "pooled_kernel (the launches of " + " and ".join(f"{s} {w}x{h}" for s, h, w in frames) + ")",
          "achieved": None, "peak": None, "unit": "G wave-instr/s", "frac": None, "traffic": None}
    try:
        with open(os.path.join(ROOT, "profiles", "issue_peak.json")) as f:
            ip = json.load(f)
        with open(os.path.join(ROOT, "profiles", "pmc.json")) as f:
            pmc = json.load(f)
    except (OSError, ValueError) as e:
        rf["note"] = f"profiles/pmc.json or profiles/issue_peak.json unreadable ({e}): no VALU figure"
        pmc, ip = None, None
    if pmc is not None:
        if pmc.get("source_sha256") != kernel_source_hash():
            rf["stale_pmc"] = True
            rf["note"] = ("profiles/pmc.json was measured on other kernel sources (hash mismatch): refused; "
                          "rerun tools/gpu_round.sh and commit its pmc.json")
        elif variant in (0, 3):
            ents = [pmc.get("launches", {}).get(f"{s} {w}x{h}", {}).get(key) for s, h, w in frames]
            if all(ents):
                peak = float(ip["valu_peak_G"])
                insts = sum(e["SQ_INSTS_VALU"] for e in ents)
                ach = insts * steps / elapsed / 1e9
                rf.update(achieved=ach, peak=peak, frac=ach / peak, nominal_peak=VALU_NOMINAL_PEAK_G, frac_of_nominal=ach / VALU_NOMINAL_PEAK_G,
                          insts_per_step=insts,
                          pmc_source="committed profiles/pmc.json: rocprofv3 --pmc passes of tools/gpu_pmc.sh over the native bench, one "
                                     "launch at a time, on another box than this run's; tied to the kernel sources by sha256 (checked "
                                     "above); only the wall time of the timed region is this run's",
                          peak_source="tools/issue_peak.hip (profiles/issue_peak.json): independent v_add/v_mul/v_fma_f32 streams, "
                                      ">= 2 waves per SIMD, all 256 CUs")
                # the kernel's own mix priced with the measured per-class issue costs
                if all("class_ns" in e for e in ents):
                    busy_ns = sum(e["class_ns"] for e in ents)          # SIMD-nanoseconds of VALU pipe time per step
                    rf["mix"] = {"valu_pipe_busy": busy_ns * steps / (elapsed * 1e9 * NUM_SIMD),
                                 "note": "sum over VALU classes of (PMC instruction count x measured ns per wave-instruction per SIMD) "
                                         "/ (1024 SIMDs x wall time): the fraction of the chip's VALU pipe time the timed region used"}
                hb = [e.get("hbm_bytes") for e in ents]
                if all(hb) and world == 1:
                    rf["traffic"] = sum(hb)
                lds = [e.get("SQ_LDS_IDX_ACTIVE") for e in ents]
                if all(lds):
                    clk = float(ip.get("clock_GHz", 2.4))
                    rf["lds"] = {"busy": sum(lds) * steps / (elapsed * 256 * clk * 1e9),
                                 "note": "SQ_LDS_IDX_ACTIVE cycles per step / (256 CUs x wall cycles): LDS pipe utilisation"}
    rf["alg_bytes"] = {"achieved": alg, "unit": "GB/s", "frac_of_hbm_peak": alg / HBM_PEAK_GBS, "hbm_peak": HBM_PEAK_GBS,
                       "note": "SURVEY 8d algorithmic bytes (32 B/box test + 16 B/sphere test + 4 B/pixel) of all launches / wall time. "
                               "These bytes are served from LDS (node records, ray table) and L2, not from HBM -- see traffic -- so "
                               "the ratio to the HBM peak is a rate, not a utilisation, and may exceed 1"}
    return rf


def serial_roofline(frames, kernel_ms):
    """the one-frame-at-a-time launches (the reference's protocol) against the same VALU-issue peaks, per scene:
    PMC counters of a single-frame launch (profiles/pmc.json, key grid_div=0) over this run's kernel times"""
    try:
        with open(os.path.join(ROOT, "profiles", "issue_peak.json")) as f:
            ip = json.load(f)
        with open(os.path.join(ROOT, "profiles", "pmc.json")) as f:
            pmc = json.load(f)
    except (OSError, ValueError):
        return None
    if pmc.get("source_sha256") != kernel_source_hash():
        return {"stale_pmc": True}
    out = {}
    for (s, h, w), ms in zip(frames, kernel_ms):
        e = pmc.get("launches", {}).get(f"{s} {w}x{h}", {}).get("grid_div=0")
        if not e or "SQ_INSTS_VALU" not in e:
            continue
        ach = e["SQ_INSTS_VALU"] / (ms * 1e-3) / 1e9
        r = {"achieved": ach, "unit": "G wave-instr/s", "frac": ach / float(ip["valu_peak_G"]), "frac_of_nominal": ach / VALU_NOMINAL_PEAK_G,
             "insts_per_frame": e["SQ_INSTS_VALU"]}
        if "class_ns" in e:
            r["valu_pipe_busy"] = e["class_ns"] / (ms * 1e6 * NUM_SIMD)
        if "hbm_bytes" in e:
            r["hbm_GBs"] = e["hbm_bytes"] / (ms * 1e-3) / 1e9
        out[f"{s}_{w}x{h}"] = r
    return out or None


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
    ap.add_argument("--protocol", choices=["batch", "lanes"], default="batch",
                    help="batch: one rt_render_batch launch per scene and chunk; lanes: one launch per frame, frames overlapped on streams")
    ap.add_argument("--launch-order", choices=["reversed", "listed"], default="reversed",
                    help="batch protocol: which scene's launch goes first (reversed: irreg before rgbbox)")
    ap.add_argument("--chunks", type=int, default=0, help="batch protocol: launches per scene (0 = 1)")
    ap.add_argument("--frames-in-flight", type=int, default=10, help="independent steps enqueued concurrently (streams); capped by --steps")
    ap.add_argument("--event-every", type=int, default=1,
                    help="bracket the launches of every n-th timed step with HIP events (kernel duration samples)")
    ap.add_argument("--repeats", type=int, default=5,
                    help="the timed K-step bracket is run this many times (warm-up passes, poison, bracket, verification each time); "
                         "steps / ms_per_step / value describe the MEDIAN bracket, brackets_ms lists them all (1: one bracket, as before round 5)")
    ap.add_argument("--no-serial-extra", action="store_true", help="skip the extra serial (one frame at a time) region")
    ap.add_argument("--idle-before-ms", type=float, default=0.0,
                    help="experiment: idle the GPU this long between the warm-up steps and the timed bracket (DESIGN.md 6, first-process effect)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-scale-extra", action="store_true", help="N > 1: skip the irreg 4000x4000 sub-record")
    ap.add_argument("--exchange", choices=["auto", "direct", "gather"], default="auto",
                    help="N > 1: how the framebuffer reaches rank 0.  direct: every rank's kernel stores its pixels straight into rank 0's "
                         "image (IPC mapping, over xGMI, while it renders); gather: one RCCL gather + an assembly launch behind the "
                         "renders; auto: direct, checked against the oracle's checksums before anything is timed, else gather")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    # RT_SHARE_GPU=1: every rank uses cuda:0 and the gloo backend (host-staged gather) -- a test
    # mode that runs the multi-rank control flow on a one-GPU box; never the measured configuration
    share_gpu = bool(os.environ.get("RT_SHARE_GPU"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback")
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if args.gpus > torch.cuda.device_count() and not share_gpu:
        # never a mislabelled smaller run: N GPUs were asked for, N must be there
        raise SystemExit(f"bench.py: --gpus {args.gpus} but only {torch.cuda.device_count()} GPU(s) are visible")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` on its own: become the N-rank job the contract describes (one process per GPU,
        # torch.distributed over RCCL) instead of silently measuring one GPU
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        log(f"bench.py: --gpus {args.gpus} without a launcher: re-executing under torch.distributed.run ({args.gpus} ranks, 127.0.0.1:{port})")
        sys.stderr.flush()
        os.dup2(saved_stdout, 1)
        os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
                                  "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:])
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s) (WORLD_SIZE): refusing to print a "
                         "line whose n_gpus is not what was asked for")
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
    batch = args.protocol == "batch" and args.variant in (0, 3)
    if batch:
        # K steps in C launches per scene (default 1: every launch ends with the frame's longest bounce chain, ~0.5 ms that
        # a rank with an eighth of the rows cannot afford twice), the scenes' launches on separate streams (they overlap
        # where one's tail leaves CUs idle; each has its own gather).  One-GPU stand-in for a rank's share of K = 20 steps
        # (tools/rank_share_probe.py, us per step, slowest part; profiles/r03/rank_share_probe.txt): W = 1 / 2 / 4 / 8 =
        # 370 / 191 / 106 / 62 with full-size launches, 69 at W = 8 with half-size ones (round 2, when a rank's eighth was
        # bound by its bounce chains, it was the other way round -- 95 against 76 -- and W >= 8 launched half-size).
        C = max(1, min(args.chunks or 1, args.steps))
        chunk_sizes = [args.steps // C + (1 if i < args.steps % C else 0) for i in range(C) for _ in frames]
        # launch order: the LAST scene first.  Persistent workgroups of two launches do not share a CU (each fills its LDS), so
        # the launches run one after the other with an overlap at the seam; irreg's launch ends with ~0.5 ms of lone bounce
        # chains, which rgbbox's launch fills when it comes second (measured: 0.395 -> see DESIGN.md 6)
        lane_frames = [[fr] for _ in range(C) for fr in (reversed(frames) if args.launch_order == "reversed" else frames)]
        S = len(chunk_sizes)
        opts_pipe = dict(opts)
    else:
        # Frames in flight: never more lanes than steps (a lane that gets no timed step only adds set-up).  Ten lanes at
        # a quarter-size launch each were the best of {6, 10, 20} x grid_div {2, 4, 8} at the driver's K = 20
        # (DESIGN.md 6): a workgroup fills a CU's LDS, so only 4 quarter-size launches are resident at a time and more
        # lanes only lengthen the queue -- and the drain at the end of the bracket.
        S = max(1, min(args.frames_in_flight, args.steps))
        chunk_sizes = [1] * S
        lane_frames = [frames] * S
        # With many frames in flight a launch need not fill the machine by itself: a quarter of the
        # persistent workgroups per launch gives longer-lived, better-filled waves; the longer tail is
        # hidden by the other frames.  One frame at a time keeps the library default.
        opts_pipe = dict(opts)
        if S >= 4 and args.variant in (0, 3):
            opts_pipe.setdefault("grid_div", 4)
            # dedicated waves for the deepest tiles shorten ONE frame's tail, which overlapped frames
            # hide anyway; they cost 2-3 % of throughput here
            opts_pipe.setdefault("deep_class", 0)
    # one "lane" per launch in flight: its own HIP stream, contexts, prepared scenes, framebuffers
    streams = [torch.cuda.current_stream(device)] if S == 1 else [torch.cuda.Stream(device) for _ in range(S)]

    exchange = {"auto": "direct", "direct": "direct", "gather": "gather"}[args.exchange] if use_pg else "gather"

    class Lane:
        """the renderers of one launch in flight + the step (render all, exchange, assemble)"""
        def __init__(self, o, fr=frames, nbatch=1, exch=None):
            self.fr = fr
            exch = exch or exchange
            self.prs = [HipPartRenderer(scene, h, w, device, variant=args.variant, options=o) for scene, h, w in fr]
            self.step = ShardedStep([(pr, h, w) for pr, (_, h, w) in zip(self.prs, fr)], device, nbatch=nbatch, exchange=exch)
            if exch == "direct" and self.step.exchange_mode != "direct" and rank == 0:
                log(f"bench.py: {self.step.exchange_note}")

    def build_lanes():
        ls = []
        for st, nb, fr in zip(streams, chunk_sizes, lane_frames):
            with torch.cuda.stream(st):
                ls.append(Lane(opts_pipe, fr, nbatch=nb))
        return ls, (Lane(opts) if ((S > 1 or batch) and not args.no_serial_extra) else None)   # (the serial lane: on the default stream)

    lanes, serial_lane = build_lanes()
    torch.cuda.synchronize()
    renderers = [(scene, h, w, pr) for (scene, h, w), pr in zip(frames, serial_lane.prs if serial_lane else
                                                                 [next(ln.prs[0] for ln in lanes if ln.fr[0] == fr) for fr in frames] if batch else lanes[0].prs)]

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

    def poison(which):
        """overwrite the framebuffers so that only the timed launches can make them right -- on the stream the lane's launches
        go to, i.e. ordered behind its warm-up launches without a host-side wait"""
        for ln in which:
            st = next((s_ for s_, l_ in zip(streams, lanes) if l_ is ln), None)
            with torch.cuda.stream(st) if st is not None else contextlib.nullcontext():
                for img in (ln.step.images or []):
                    img.fill_(0x5a5a5a5a)
                if not ln.step.direct:
                    ln.step.send.fill_(0x5a5a5a5a)

    cks = Checksummer(device)

    def verify(which, what):
        """rank 0: every image of every lane (a batch lane holds nbatch of them) against the oracle's checksums"""
        bad = []
        n = 0
        if rank == 0:
            for li, ln in enumerate(which):
                for (scene, h, w), imgs in zip(ln.fr, ln.step.images):
                    want = FRAME_CHECKSUM.get((scene, h, w))
                    if want is None:
                        continue
                    for fi, img in enumerate(imgs if imgs.dim() == 3 else [imgs]):
                        got = cks(img)
                        n += 1
                        if got != want:
                            bad.append(f"{what} launch {li} frame {fi} {scene} {w}x{h}: checksum {got:08x}, oracle {want:08x}")
        if bad:
            raise SystemExit("VERIFICATION FAILED (pixels differ from the oracle's):\n  " + "\n  ".join(bad))
        return n

    def timed(nsteps, nlanes):
        """K steps inside one bracket.  lanes protocol: nsteps launches of every scene, round-robin over the lanes;
        batch protocol: every lane's launches once (together they cover exactly K steps); nlanes == 0: the serial lane"""
        every = max(1, args.event_every)
        nlaunch = S if (batch and nlanes != 0) else nsteps
        lane_of = (lambda k: serial_lane) if nlanes == 0 else (lambda k: lanes[k % nlanes])
        ev = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in lane_of(k).fr]
              if k % every == 0 else None for k in range(nlaunch)]
        fence()
        t0 = time.perf_counter()
        for k in range(nlaunch):
            step(k, ev[k], nlanes)
        fence()
        dt = time.perf_counter() - t0
        if use_pg:
            t = torch.tensor([dt], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        # per-launch durations on this rank: events recorded on the stream each kernel is launched on
        per = {fr: [] for fr in frames}
        for k in range(nlaunch):
            if ev[k] is not None:
                for fr, (a, b) in zip(lane_of(k).fr, ev[k]):
                    per[fr].append(a.elapsed_time(b))
        kms = [float(np.mean(per[fr])) for fr in frames]
        return dt, kms

    # Lane set-up: every lane renders its frames twice, which fills the per-view caches (u/v tables;
    # the adaptive tile order is computed once from the first frame's cost record).  Then the W
    # untimed warm-up steps, then exactly K timed steps.
    for k in range(2 * S):
        step(k)
    if use_pg and exchange == "direct":
        # The direct-store exchange has only ever run on one GPU before the driver's multi-GPU box: check the set-up frames
        # against the oracle's checksums NOW, and put every rank back on the RCCL gather if they are not right -- a wrong
        # exchange must cost a fallback, not the run.
        fence()
        okt = torch.ones(1, dtype=torch.int32, device="cpu" if share_gpu else device)
        if rank == 0:
            try:
                verify(lanes, "direct-store self-check,")
            except SystemExit as e:
                log(f"bench.py: direct stores failed their self-check, falling back to the RCCL gather:\n{e}")
                okt.zero_()
        dist.broadcast(okt, src=0)
        if int(okt.item()) == 0 or any(ln.step.exchange_mode != "direct" for ln in lanes):
            exchange = "gather"
            for ln in lanes + ([serial_lane] if serial_lane else []):
                ln.step.close()
            lanes, serial_lane = build_lanes()
            torch.cuda.synchronize()
            for k in range(2 * S):
                step(k)
    # Everything the host does for the FIRST time goes here, ahead of the warm-up steps, not between them and the bracket:
    # the first fill_ of a process loads torch's code object for it, and as the first GPU process of a fresh box (cold page
    # cache: what the driver's run is) that takes long enough for the idle GPU to drop its clock -- the timed launches then
    # ran their unchanged cycle counts at 2.2 instead of 2.38 GHz (profiles/r03/first_process_pmc.txt).
    poison(lanes)
    torch.cuda.Event(enable_timing=True).record()
    torch.cuda.synchronize()
    for k in range(2 * S if (batch and args.warmup > 0) else args.warmup):   # (batch: two more passes of all K >= W steps)
        step(k)
    poison(lanes)                       # stream-ordered behind the warm-up launches; the bracket's fence waits for it
    if args.idle_before_ms > 0:         # the experiment behind the comment above: let the GPU idle ahead of the bracket
        torch.cuda.synchronize()
        time.sleep(args.idle_before_ms * 1e-3)
    elapsed, kern_ms = timed(args.steps, S)
    n_verified = verify(lanes, "timed region,")
    # --repeats R: the identical bracket again, R - 1 times -- each behind its own untimed warm-up passes (the verification
    # before them reads the images back, i.e. lets the GPU idle: the passes bring the clock back) and its own
    # poison, each verified.  The line reports the median bracket; a single 7 ms bracket moved by +-1.3 % from box to box.
    brackets = [(elapsed, kern_ms)]
    for _ in range(max(1, args.repeats) - 1):
        for k in range(2 * S if (batch and args.warmup > 0) else args.warmup):
            step(k)
        poison(lanes)
        brackets.append(timed(args.steps, S))
        n_verified += verify(lanes, "timed region (repeat),")
    brackets_ms = [b[0] * 1e3 for b in brackets]
    elapsed, kern_ms = sorted(brackets, key=lambda b: b[0])[(len(brackets) - 1) // 2]
    serial = None
    if serial_lane is not None:
        for k in range(4):              # render + sync, as the reference's harness does: the view's order after frame 1, its deep-tile
            step(k, None, 0)            # policy (which reaches the host asynchronously) from frame 2 or 3 on -- and the kernel
            torch.cuda.synchronize()    # instantiation it selects has run once before anything is timed
        poison([serial_lane])
        nser = max(10, args.steps // 2)
        serial = timed(nser, 0)
        n_verified += verify([serial_lane], "serial region,")
        serial_launch = {f"{sc}_{w}x{h}": pr.ctx.last_launch for (sc, h, w), pr in zip(frames, serial_lane.prs)}   # which kernel rendered them
        # Culling by the best hit (the CULL instantiations): the sphere tests the PRODUCT's launch actually makes, from an
        # instrumented launch of the same view (rt_render_trace: leaf items over all waves).  The reference's count (FRAME_WORK, what
        # alg_bytes is defined on) is an upper bound; pixels are verified above.
        culled_work = {}
        if world == 1 and rank == 0:
            import ctypes as C
            from raytracers_amd._lib import lib as _rtlib
            for (sc, h, w), pr in zip(frames, serial_lane.prs):
                if "+CULL" not in serial_launch[f"{sc}_{w}x{h}"]:
                    continue
                rec = np.zeros((8192, 16), dtype=np.uint64)
                nw = C.c_int32()
                if _rtlib.rt_render_trace(pr.ctx._h, pr.prepared._h, h, w, 50, rec.ctypes.data, 8192, C.byref(nw)) == 0:
                    made = int((rec[: nw.value, 6].astype(np.int64) & 0xFFFFFFFF).sum())
                    ref_t = work[(sc, h, w)][2]
                    if made > ref_t:
                        raise SystemExit(f"culled launch of {sc} {w}x{h} made MORE sphere tests ({made}) than the reference's fold ({ref_t})")
                    culled_work[f"{sc}_{w}x{h}"] = {"sphere_tests_made": made, "sphere_tests_reference": ref_t, "saved": 1.0 - made / ref_t}

    # The FIRST frames of a view (the reference's `render` is stateless, ray.fut:246; here the tile order, the deep-tile
    # policy and the solo pixels exist from a view's second frame on) and a camera path (a batch with a camera per frame
    # has no single view to order tiles by): one GPU only, after the timed regions.
    cold = None
    if serial_lane is not None and world == 1 and rank == 0:
        import raytracers_amd as R
        first, first_wall, path, path_serial, path_b2b, path_rb = {}, {}, {}, {}, {}, {}
        for (scene, h, w), pr in zip(frames, serial_lane.prs):
            img = torch.empty((h, w), dtype=torch.int32, device=device)
            want = FRAME_CHECKSUM.get((scene, h, w))
            reps, reps_wall = [], []
            for rep in range(5):    # five fresh prepared scenes (no view has been seen): the median per frame index -- one sample moved by +-8 %
                if rep:
                    ps2.free()
                ps2 = R.prepare_scene(h, w, pr.scene)
                ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(4)]
                wall = []
                torch.cuda.synchronize()
                for k in range(4):      # render + sync, frame by frame (main.c:113-117); wall clock around the pair, like serial_value
                    t0 = time.perf_counter()
                    ev[k][0].record()
                    R.render_into(img.data_ptr(), h, w, ps2)
                    ev[k][1].record()
                    pr.ctx.sync()       # (futhark_context_sync: the context's stream -- not the whole device, whose side streams carry the record's sorts)
                    wall.append((time.perf_counter() - t0) * 1e3)
                torch.cuda.synchronize()
                if want is not None and cks(img) != want:
                    raise SystemExit(f"VERIFICATION FAILED: first frames of {scene} {w}x{h}")
                reps.append([a.elapsed_time(b) for a, b in ev])
                reps_wall.append(wall)
            first[f"{scene}_{w}x{h}"] = [float(np.median([r[k] for r in reps])) for k in range(4)]
            first_wall[f"{scene}_{w}x{h}"] = [float(np.median([r[k] for r in reps_wall])) for k in range(4)]
            # camera path: 20 frames, the prepared camera moved sideways a little more each frame, in one batch launch;
            # checked against the same cameras rendered one at a time
            nb = 20
            cams = np.tile(np.asarray(ps2.camera(), dtype=np.float32).reshape(1, 12), (nb, 1))
            cams[:, 0] += 0.05 * np.arange(nb, dtype=np.float32)
            cams[:, 3] += 0.05 * np.arange(nb, dtype=np.float32)
            buf = torch.empty((nb, h, w), dtype=torch.int32, device=device)
            for _ in range(2):
                R.render_batch_into(buf.data_ptr(), h, w, ps2, nb, frame_stride=h * w, cams=cams)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            R.render_batch_into(buf.data_ptr(), h, w, ps2, nb, frame_stride=h * w, cams=cams)
            b.record()
            torch.cuda.synchronize()
            for f in (0, nb - 1):
                R.render_into(img.data_ptr(), h, w, ps2, cam=cams[f])
                torch.cuda.synchronize()
                if cks(img) != cks(buf[f]):
                    raise SystemExit(f"VERIFICATION FAILED: camera path frame {f} of {scene} {w}x{h} differs from its single render")
            path[f"{scene}_{w}x{h}"] = a.elapsed_time(b) / nb
            # ... and the same path one view at a time on a fresh prepared scene -- the reference's protocol along a path: render, then sync
            # (main.c:113-117) -- every view is new; wall clock around each render + sync
            ps3 = R.prepare_scene(h, w, pr.scene)
            per = []
            torch.cuda.synchronize()
            for f in range(nb):
                t0 = time.perf_counter()
                R.render_into(img.data_ptr(), h, w, ps3, cam=cams[f])
                pr.ctx.sync()
                per.append((time.perf_counter() - t0) * 1e3)
            torch.cuda.synchronize()
            if cks(img) != cks(buf[nb - 1]):
                raise SystemExit(f"VERIFICATION FAILED: camera path (view by view) of {scene} {w}x{h}")
            path_serial[f"{scene}_{w}x{h}"] = {"first": per[0], "mean_of_the_rest": float(np.mean(per[1:]))}
            ps3.free()
            # ... view by view with the frame READ BACK after every render (futhark_values_i32_2d: what a consumer of the frames does; the harness
            # with -f): the device idles ~0.3 ms per view, the previous view's sorts finish in that gap, the new view borrows its order.
            # Kernel time of the render per view (events), the copy excluded.
            ps3 = R.prepare_scene(h, w, pr.scene)
            host = torch.empty((h, w), dtype=torch.int32).pin_memory()
            evr = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(nb)]
            for f in range(nb):
                evr[f][0].record()
                R.render_into(img.data_ptr(), h, w, ps3, cam=cams[f])
                evr[f][1].record()
                pr.ctx.sync()
                host.copy_(img)
                torch.cuda.synchronize()
            per = [a.elapsed_time(b) for a, b in evr]
            path_rb[f"{scene}_{w}x{h}"] = {"first": per[0], "mean_of_the_rest": float(np.mean(per[1:])), "last_launch": pr.ctx.last_launch}
            ps3.free()
            # ... and enqueued back to back, one sync at the end (the caller is far ahead of the device: nothing a new view could borrow is sorted yet)
            ps3 = R.prepare_scene(h, w, pr.scene)
            evp = [torch.cuda.Event(enable_timing=True) for _ in range(nb + 1)]
            evp[0].record()
            for f in range(nb):
                R.render_into(img.data_ptr(), h, w, ps3, cam=cams[f])
                evp[f + 1].record()
            torch.cuda.synchronize()
            if cks(img) != cks(buf[nb - 1]):
                raise SystemExit(f"VERIFICATION FAILED: camera path (back to back) of {scene} {w}x{h}")
            per = [evp[f].elapsed_time(evp[f + 1]) for f in range(nb)]
            path_b2b[f"{scene}_{w}x{h}"] = {"first": per[0], "mean_of_the_rest": float(np.mean(per[1:]))}
            ps3.free()
            ps2.free()
        cold = {"first_frames_ms": first,
                "first_frames_wall_ms": first_wall,
                "first_frames_note": "frames 1..4 of a fresh prepared scene, render + sync each (median of five fresh prepared scenes; first_frames_ms: HIP events around the "
                                     "launch, first_frames_wall_ms: host wall clock around call + sync -- the kind of number serial_value is): frame 1 has no order "
                                     "(it records every pixel's bounce-chain length; the sorts of that record run behind it on the context's second stream, while the "
                                     "caller synchronises), frames 2.. render through the view's own pixel list",
                "camera_path_ms_per_frame": path,
                "camera_path_note": "20 frames, a camera per frame, ONE rt_render_batch launch (no per-view order); first and last "
                                    "frame checked against single renders of the same cameras",
                "camera_path_frame_by_frame_ms": path_serial,
                "camera_path_frame_by_frame_note": "the same 20 cameras one view at a time on a fresh prepared scene, render + sync per view as the reference's harness "
                                                   "does (main.c:113-117), wall clock: every view is new.  A new view borrows the order of one of the two views before "
                                                   "it only if that view's sorts are THROUGH when the call comes in -- with no gap between sync and the next render "
                                                   "they are not (they need ~0.1 ms of an idle device), so these views render unordered, as in round 5",
                "camera_path_with_readback_ms": path_rb,
                "camera_path_with_readback_note": "the same views, render + sync + a 4 MB device-to-host copy of the frame per view (what a consumer of the frames does): "
                                                  "kernel time of the render per view (events; the copy excluded).  The device idles during the copy, the previous view's "
                                                  "sorts finish in that gap, and the new view renders through the borrowed order (last_launch says so)",
                "camera_path_back_to_back_ms": path_b2b,
                "camera_path_back_to_back_note": "the same views enqueued back to back, one sync at the end (events between the calls): the caller is far ahead of the "
                                                 "device, no earlier view's sorts are through when a new view is enqueued, every view renders unordered"}

    # N > 1: the configuration north_star states its scaling target on (irreg 4000x4000), one frame at a time
    scale_extra = None

    def irreg_4000_record(exch):
        """irreg 4000x4000 across the ranks through one exchange: one frame at a time, then the same frames as one batch"""
        nonlocal n_verified
        fr4 = [("irreg", 4000, 4000)]
        big_lane = Lane(opts, fr4, exch=exch)
        mode = big_lane.step.exchange_mode
        for _ in range(3):
            big_lane.step.render()
            torch.cuda.synchronize()      # (render + sync: the view's policy arrives asynchronously, see the serial lane)
        poison([big_lane])
        n4 = 6
        evr = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n4)]
        evg = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n4)]
        fence()
        t0 = time.perf_counter()
        for k in range(n4):
            big_lane.step.render([evr[k]], gather_events=evg[k])
            torch.cuda.synchronize()          # the reference's protocol: render, then sync (main.c:113-117)
        fence()
        dt4 = time.perf_counter() - t0
        t = torch.tensor([dt4, float(np.mean([a.elapsed_time(b) for a, b in evr]))], dtype=torch.float64, device=device)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tmin = t.clone()
        dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
        n_verified += verify([big_lane], f"irreg 4000x4000 ({mode}),")
        # ... and the same frames as ONE batch launch per rank (one frame at a time cannot end before its longest bounce chain)
        big_lane.step.close()             # (direct exchange: the other ranks unmap rank 0's images before it frees them; collective)
        del big_lane
        bb = Lane(opts, fr4, nbatch=n4, exch=exch)
        for _ in range(2):
            bb.step.render()
        torch.cuda.synchronize()
        poison([bb])
        fence()
        t0 = time.perf_counter()
        bb.step.render()
        fence()
        tb = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
        dist.all_reduce(tb, op=dist.ReduceOp.MAX)
        n_verified += verify([bb], f"irreg 4000x4000 batch ({mode}),")
        bb.step.close()
        del bb
        if rank != 0:
            return None
        r4 = FRAME_WORK[("irreg", 4000, 4000)][0]
        ms = float(tmax[0].item()) / n4 * 1e3
        return {"workload": "irreg 4000x4000, one frame at a time (render + exchange + sync per frame)", "exchange": mode,
                "ms_per_frame": ms, "Mray_s": r4 / ms / 1e3,
                "render_us_per_rank": {"slowest": float(tmax[1].item()) * 1e3, "fastest": float(tmin[1].item()) * 1e3},
                "gather_and_assemble_us_rank0": float(np.mean([a.elapsed_time(b) for a, b in evg])) * 1e3,
                "batch": {"frames_per_launch": n4, "ms_per_frame": float(tb[0].item()) / n4 * 1e3,
                          "Mray_s": r4 * n4 / float(tb[0].item()) / 1e6,
                          "note": "the same frames in one rt_render_batch launch per rank + one exchange"},
                "frames": n4, "verified": True}

    if world > 1 and not args.no_scale_extra and args.workload == "rgbbox+irreg-1000":
        # BOTH exchanges in one run (the driver's multi-GPU box is the only place they can be compared): the one the headline ran
        # on first -- it is the `irreg_4000` record -- then the other one; each verified against the oracle's checksum
        recs = {}
        for exch in ([exchange] + [e for e in ("direct", "gather") if e != exchange and (e == "gather" or args.exchange != "gather")]):
            rec = irreg_4000_record(exch)
            if rank == 0 and rec["exchange"] not in recs:
                recs[rec["exchange"]] = rec
        if rank == 0:
            scale_extra = dict(next(iter(recs.values())))
            scale_extra["by_exchange"] = {m: {k: v for k, v in r.items() if k not in ("workload", "frames", "verified")} for m, r in recs.items()}
            scale_extra["predicted"] = load_scale_prediction(world)

    if rank == 0:
        rays_step = sum(work[(s, h, w)][0] for s, h, w in frames)
        value = rays_step * args.steps / elapsed / 1e6
        per_scene = {}
        for i, (scene, h, w, _) in enumerate(renderers):
            r, b, s = work[(scene, h, w)]
            ba = bytes_alg(b, s, h, w) / world        # this rank's share of the frame (cyclic tiles)
            fpl = max(chunk_sizes) if batch else 1
            per_scene[f"{scene}_{w}x{h}"] = {
                "rays": r, "kernel_ms": kern_ms[i], "frames_per_launch": fpl,
                "Mray_s_kernel": r * fpl / world / (kern_ms[i] * 1e-3) / 1e6, "alg_bytes_per_frame": ba}
        # kernel-milliseconds inside the bracket / its wall time (kern_ms: mean duration of a scene's launches; a batch has
        # S / len(frames) launches per scene, the lanes protocol one per step)
        inflight = sum(kern_ms) * ((S // len(frames)) if batch else args.steps) / (elapsed * 1e3)
        out = {
            "metric": "Mray/s (primary+secondary) on rgbbox & irreg 1000x1000" if args.workload == "rgbbox+irreg-1000"
                      else f"Mray/s (primary+secondary) on {args.workload}",
            "value": value, "unit": "Mray/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic (the reference's procedural scenes)",
            "verified": True, "verified_images": n_verified,
            **({"repeats": len(brackets), "brackets_ms": brackets_ms,
                "value_min": rays_step * args.steps / (max(brackets_ms) * 1e-3) / 1e6, "value_max": rays_step * args.steps / (min(brackets_ms) * 1e-3) / 1e6,
                "repeats_note": f"{len(brackets)} identical brackets of {args.steps} steps, each behind its own warm-up passes and poison fill, each verified; "
                                "value / ms_per_step are the median bracket's"} if len(brackets) > 1 else {}),
            "rccl_ranks": (dist.get_world_size() if (use_pg and not share_gpu) else 0),
            "gather_mode": ("none (one GPU, whole frames)" if not use_pg else
                            "direct-store: every rank's kernel stores its pixels into rank 0's image (IPC mapping) while it renders; "
                            "two one-element all-reduces per launch order them" if lanes[0].step.exchange_mode == "direct" else
                            "gather: one torch.distributed gather (RCCL) per launch + one assembly launch per scene")
                           + (" [RT_SHARE_GPU test mode: gloo]" if share_gpu else ""),
            "value_protocol": (f"batched throughput: the {args.steps} steps' frames of each scene in {S // len(frames)} rt_render_batch "
                               f"launch(es) of {'/'.join(str(c) for c in chunk_sizes[::len(frames)])} frames (the scenes on separate "
                               "streams), all inside the timed bracket" if batch else
                               f"overlapped throughput: {S} independent frames in flight on {S} HIP streams, all {args.steps} steps inside "
                               "the timed bracket") + "; the reference's protocol (one render + sync at a time) is serial_value",
            "config": {"workload": " + ".join(f"{s} {w}x{h}" for s, h, w in frames) + ", max_depth 50, one frame of each per step",
                       "kernel": {0: "auto (pooled)", 1: "pixel", 2: "persistent", 3: "pooled"}[args.variant],
                       "options": opts_pipe, "protocol": args.protocol if batch or args.protocol == "lanes" else "lanes",
                       "launches_in_flight": S, "frames_per_launch": chunk_sizes,
                       "partition": ("one GPU: whole frames, no partition, no gather" if world == 1 and not use_pg else
                                     f"cyclic 8-row tiles over {world} GPU(s); exchange: see gather_mode")
                                    + (" [RT_SHARE_GPU test mode: ranks share cuda:0, gloo host-staged gather]" if share_gpu else "")},
            "per_scene": per_scene,
            "pipeline": {"avg_launches_in_flight": inflight,
                         "fill_drain_note": "the timed bracket starts and ends with an idle GPU: with K steps of t ms and a launch "
                                            "latency of L ms (per_scene kernel_ms) the bracket is about K t + L, so short runs "
                                            "under-report the steady state by L / (K t)"},
            "derived_reference": {"futhark_mi100_Mray_s": {"rgbbox": 287.3, "irreg": 216.1},
                                  "note": "README.md:50 render times / oracle ray counts; different hardware; compare with serial_*"},
        }
        out["roofline"] = roofline_block(frames, world, args.variant, "batch" if batch else f"grid_div={opts_pipe.get('grid_div', 0)}",
                                         args.steps, elapsed, per_scene, work)
        out["roofline"]["per_launch"] = {
            "kernel_ms": {k: per_scene[k]["kernel_ms"] for k in per_scene}, "avg_launches_in_flight": inflight,
            "note": "HIP events around each launch on its own stream inside the timed region; launches in flight stretch one another"}
        if serial is not None:
            sdt, skms = serial
            out["serial_value"] = rays_step * nser / sdt / 1e6
            out["serial_ms_per_frame"] = {f"{sc}_{w}x{h}": skms[i] for i, (sc, h, w) in enumerate(frames)}
            out["serial"] = {
                "note": "the reference's protocol: one frame at a time on one stream (futhark/main.c:107-124), library-default "
                        "knobs, measured right after the timed region; kernel_ms = HIP events around each launch",
                "launch": serial_launch,
                **({"culled_work": culled_work} if culled_work else {}),
                "value": out["serial_value"], "ms_per_step": sdt / nser * 1e3, "kernel_ms": out["serial_ms_per_frame"],
                "alg_bytes_GBs": {f"{sc}_{w}x{h}": bytes_alg(work[(sc, h, w)][1], work[(sc, h, w)][2], h, w) / world
                                  / (skms[i] * 1e-3) / 1e9 for i, (sc, h, w) in enumerate(frames)}}
            if world == 1:
                sr = serial_roofline(frames, skms)
                if sr:
                    out["serial"]["roofline"] = sr
            if cold is not None:
                out["serial"].update(cold)
            tg = {}
            for i, (sc, h, w) in enumerate(frames):
                if sc in MI100_RENDER_MS and (h, w) == (1000, 1000) and world == 1:
                    tg[sc] = {"mi100_ms": MI100_RENDER_MS[sc], "ms": skms[i], "speedup": MI100_RENDER_MS[sc] / skms[i],
                              "target_10x_met": MI100_RENDER_MS[sc] / skms[i] >= 10.0}
            if cold is not None:
                # the STATELESS figure (the reference's render keeps nothing between calls, ray.fut:246): a view never seen before
                ff = cold["first_frames_ms"]
                for sc in tg:
                    c_ms = cold["first_frames_wall_ms"][f"{sc}_1000x1000"][0]
                    tg[sc].update(cold_ms=c_ms, cold_speedup=MI100_RENDER_MS[sc] / c_ms, cold_target_10x_met=MI100_RENDER_MS[sc] / c_ms >= 10.0)
                if all(f"{sc}_{w}x{h}" in ff for sc, h, w in frames):
                    fw = cold["first_frames_wall_ms"]
                    out["serial_cold_value"] = rays_step / sum(fw[f"{sc}_{w}x{h}"][0] for sc, h, w in frames) / 1e3
                    out["serial"]["cold_value"] = out["serial_cold_value"]
                    out["serial"]["cold_value_note"] = ("Mray/s with every frame the first of its view (no order, no deep-tile policy, no solo pixels; the "
                                                        "frame records its chains -- the sorts run ahead of the view's next frame, if there is one): "
                                                        "first_frames_wall_ms[0] of each scene -- wall clock around render + sync, like serial_value")
            if tg:
                out["targets"] = {"note": ">= 10x the published MI100 Futhark render times (README.md:50), one frame at a time: `ms` a view "
                                          "rendered before (warm), `cold_ms` a view never seen (the reference's render is stateless)", **tg}
        if scale_extra is not None:
            out["irreg_4000"] = scale_extra
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(frames)
        result_line = json.dumps(out)
    else:
        result_line = None
    if use_pg:
        for ln in lanes + ([serial_lane] if serial_lane is not None else []):
            ln.step.close()               # (a no-op for the gather exchange)
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
