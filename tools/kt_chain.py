#!/usr/bin/env python3
"""The chain of launches of the LAST prepare_scene in a rocprofv3 --kernel-trace CSV: start, duration, gap behind the previous kernel.
usage: kt_chain.py <kernel_trace.csv> [first-kernel substring = centres_minmax] [--sum]"""
import csv, sys
rows = sorted(csv.DictReader(open([a for a in sys.argv[1:] if not a.startswith("--")][0])), key=lambda r: int(r["Start_Timestamp"]))
args = [a for a in sys.argv[1:] if not a.startswith("--")]
first = args[1] if len(args) > 1 else "centres_minmax"
idx = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
i0, prev = idx[-1], None
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("void rtk::(anonymous namespace)::", "").replace("rtk::(anonymous namespace)::", "")[:44]
    if prev is not None and s - prev > 30000: break      # the build is over: the next kernel belongs to a render
    print("%-44s start %7.1f us  dur %5.1f  gap %5.1f  grid %s" % (name, (s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3 if prev else 0.0, r.get("Grid_Size_X", "")))
    prev = e
print("chain: %.1f us" % ((prev - t0) / 1e3))
if "--sum" in sys.argv:      # ... and per kernel name: launches, total duration
    import collections
    acc = collections.OrderedDict()
    for r in rows[i0:]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if s > prev: break
        k = r["Kernel_Name"].replace("void rtk::(anonymous namespace)::", "").replace("rtk::(anonymous namespace)::", "").split("(")[0]
        c, d = acc.get(k, (0, 0))
        acc[k] = (c + 1, d + e - s)
    for k, (c, d) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        print("  %-40s %3d launches %8.1f us" % (k[:40], c, d / 1e3))
