#!/usr/bin/env python3
"""Condenses a rocprofv3 output directory (CSV format) into the small text summaries kept
under profiles/: per-kernel call count / total / mean / min / max duration from the
kernel-trace, and (when present) the per-kernel mean of every PMC counter.

  tools/rocpd_summary.py [--last N] <rocprof_out_dir> [more dirs ...] > profiles/<name>.txt

--last N adds, per kernel, the mean over its LAST N dispatches in time order: for a bench.py
trace these are the N timed steps (the launches before them -- lane set-up and warm-up -- run at
a different degree of overlap, so the all-dispatch mean is not the timed region's).
"""
import collections
import csv
import glob
import os
import sys


# small kernels this library enqueues right behind a persistent launch of another stream (the tile-order sort of one scene's set-up frame
# behind the other scene's batch launch): alone they take ~10 us
QUEUED_BEHIND_PERSISTENT = ("tile_count_kernel",)


def main():
    args = sys.argv[1:]
    last = 0
    if args and args[0] == "--last":
        last = int(args[1])
        args = args[2:]
    for d in args:
        print(f"== {os.path.relpath(d)}")
        kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
        timed = collections.defaultdict(list)
        for f in kt:
            for r in csv.DictReader(open(f)):
                timed[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
        dur = {k: [x[1] for x in sorted(v)] for k, v in timed.items()}
        if dur:
            tot = sum(sum(v) for v in dur.values())
            print("kernel-trace: name | calls | total_ns | mean_ns | min_ns | max_ns | % of GPU time"
                  + (f" | mean_ns of the last {last} dispatches" if last else ""))
            queued = False
            for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
                tail = f" | {sum(v[-last:]) / len(v[-last:]):.0f}" if last else ""
                # a small kernel whose longest dispatch is 50x its shortest did not run that long: it WAITED (see the note below)
                mark = " [*]" if ((max(v) > 50 * max(1, min(v)) and min(v) < 100000) or any(q in k for q in QUEUED_BEHIND_PERSISTENT)) else ""
                queued |= bool(mark)
                print(f"  {k[:90]}{mark} | {len(v)} | {sum(v)} | {sum(v) / len(v):.0f} | {min(v)} | {max(v)} | {100.0 * sum(v) / tot:.2f}{tail}")
            if queued:
                print("  [*] the trace clocks a dispatch from the moment it is handed to the hardware: these dispatches sat behind another stream's "
                      "persistent launch (every CU's LDS taken) for most of the time shown; min_ns is what the kernel takes alone, and its share of "
                      "the GPU time is queueing, not work")
        cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        acc = collections.defaultdict(list)
        for f in cc:
            for r in csv.DictReader(open(f)):
                acc[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
        if acc:
            print("pmc: kernel | counter | dispatches | mean value per dispatch")
            for (k, c), v in sorted(acc.items()):
                print(f"  {k[:70]} | {c} | {len(v)} | {sum(v) / len(v):.1f}")


if __name__ == "__main__":
    main()
