#!/usr/bin/env python3
"""profiles/pmc.json from the PMC passes of tools/gpu_pmc.sh.

usage: make_pmc_json.py <gpurun_out/.../pmc dir> <profiles/issue_peak.json> > profiles/pmc.json

Per launch of pooled_kernel at 1000x1000 and per launch size (grid_div): SQ_INSTS_VALU and its
classes, class_ns = the VALU pipe time of one launch in SIMD-nanoseconds (class counts x the ns per
wave-instruction per SIMD tools/issue_peak.hip measured for that class), SQ_LDS_IDX_ACTIVE, and
hbm_bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB (FETCH_SIZE doubled per MI355X_MICROARCH.md: gfx950
tallies 128-B read requests at 64 B).  source_sha256 ties the file to the kernel sources it was
measured on; bench.py refuses a file whose hash differs from the tree's."""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import bench
    pmc_dir, ip_path = sys.argv[1], sys.argv[2]
    ip = json.load(open(ip_path))
    cns = ip["class_ns"]
    # A run may hold two instantiations of pooled_kernel (the first frame of a view has no tile order and runs the one
    # without the solo prologue; the later, ordered frames the one with it): per run and counter, the instantiation with
    # the most dispatches -- the steady state -- counts.
    runs, most = {}, {}
    for r in csv.DictReader(open(os.path.join(pmc_dir, "pmc_summary.csv"))):
        if "pooled_kernel" not in r["kernel"]:
            continue
        key = (r["run"], r["counter"])
        if int(r["dispatches"]) >= most.get(key, 0):
            most[key] = int(r["dispatches"])
            runs.setdefault(r["run"], {})[r["counter"]] = float(r["mean_value"])
    doc = {"_comment": __doc__.split("\n\n")[2].replace("\n", " "),
           "source_sha256": bench.kernel_source_hash(), "sources": bench.KERNEL_SOURCES, "launches": {}}
    for run, c in sorted(runs.items()):
        if run.startswith("big"):      # the 10^6-sphere frame's passes stay in pmc_summary.csv (DESIGN.md quotes them)
            continue
        if "_batch" in run:
            # a batch launch of N frames: per-FRAME figures (the launch's counters / N)
            scene, nb = run.rsplit("_batch", 1)
            c = {k: (v / int(nb) if not k.startswith("SQ_WAVES") else v) for k, v in c.items()}
            gd = None
        else:
            scene, gd = run.rsplit("_gd", 1)
        e = {k: int(c[k]) for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_WAVES", "SQ_LDS_IDX_ACTIVE",
                                     "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32",
                                     "SQ_INSTS_VALU_TRANS_F32", "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_CVT", "SQ_ACTIVE_INST_VALU",
                                     "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES") if k in c}
        if "SQ_INSTS_VALU" in e and "SQ_INSTS_VALU_ADD_F32" in e:
            fast = e["SQ_INSTS_VALU_ADD_F32"] + e["SQ_INSTS_VALU_MUL_F32"] + e["SQ_INSTS_VALU_FMA_F32"]
            trans = e["SQ_INSTS_VALU_TRANS_F32"]
            other = e["SQ_INSTS_VALU"] - fast - trans
            e["class_counts"] = {"fp32_add_mul_fma": fast, "trans": trans, "other": other}
            e["class_ns"] = fast * cns["fp32_add_mul_fma"] + trans * cns["trans"] + other * cns["other"]
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            e.update(FETCH_SIZE_KiB=round(c["FETCH_SIZE"], 1), WRITE_SIZE_KiB=round(c["WRITE_SIZE"], 1),
                     hbm_bytes=int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024))
        ns = [v for k, v in c.items() if k.startswith("kernel_ns_p")]
        if ns:
            e["kernel_ms_one_at_a_time"] = round(sum(ns) / len(ns) * 1e-6, 4)
        if gd is None:
            e["frames_per_launch"] = int(nb)
            e["note"] = "per frame: counters of one rt_render_batch launch / frames per launch"
        doc["launches"].setdefault(f"{scene} 1000x1000", {})["batch" if gd is None else f"grid_div={gd}"] = e
    # HBM traffic barely depends on the launch size: reuse the default launch's where it was not measured
    for scene, by in doc["launches"].items():
        base = by.get("grid_div=0", {})
        for k, e in by.items():
            if "hbm_bytes" not in e and "hbm_bytes" in base:
                e["hbm_bytes"] = base["hbm_bytes"]
                e["hbm_bytes_from"] = "grid_div=0"
    json.dump(doc, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
