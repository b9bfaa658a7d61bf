#!/usr/bin/env python3
"""profiles/traffic.json from the PMC passes of tools/gpu_pmc.sh.

usage: make_traffic_json.py <pmc dir (default grid)> [<pmc dir (bench launch size)>] > profiles/traffic.json

HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB -- FETCH_SIZE is doubled per
MI355X_MICROARCH.md (gfx950 tallies 128-B read requests at 64 B).  VALU wave-instructions per
launch = SQ_INSTS_VALU.  bench.py copies the entries of its launches into roofline.traffic /
roofline.valu."""
import csv
import json
import os
import sys


def load(d):
    out = {}
    p = os.path.join(d, "pmc_summary.csv")
    if not os.path.exists(p):
        return out
    for r in csv.DictReader(open(p)):
        if "pooled_kernel" not in r["kernel"]:
            continue
        scene = r["run"].split("_v")[0]
        out.setdefault(scene, {})[r["counter"]] = float(r["mean_value"])
    return out


def main():
    base = load(sys.argv[1])
    small = load(sys.argv[2]) if len(sys.argv) > 2 else {}
    doc = {"_comment": "per launch of pooled_kernel at 1000x1000, from rocprofv3 --pmc passes over build/rtbench "
                       "(tools/gpu_pmc.sh, separate runs per counter group): hbm_bytes = (2 x FETCH_SIZE + WRITE_SIZE) "
                       "KiB (FETCH_SIZE doubled per MI355X_MICROARCH.md: gfx950 tallies 128-B read requests at 64 B); "
                       "valu_insts = SQ_INSTS_VALU wave-instructions, at the library's default launch size and at the "
                       "bench's (grid_div=8); bench.py copies these into roofline.traffic / roofline.valu"}
    for scene in sorted(base):
        c = base[scene]
        e = {}
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            e.update(FETCH_SIZE_KiB=round(c["FETCH_SIZE"], 1), WRITE_SIZE_KiB=round(c["WRITE_SIZE"], 1),
                     hbm_bytes=int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024))
        if "SQ_INSTS_VALU" in c:
            e["valu_insts"] = int(c["SQ_INSTS_VALU"])
        s = small.get(scene, {})
        if "SQ_INSTS_VALU" in s:
            e["valu_insts_bench_launch"] = int(s["SQ_INSTS_VALU"])
        doc[f"pooled_kernel {scene} 1000x1000"] = e
    json.dump(doc, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
