#!/usr/bin/env python3
"""profiles/issue_peak.json from the output of tools/issue_peak (tools/gpu_issue_peak.sh).

usage: make_issue_peak_json.py gpurun_out/.../issue_peak.txt > profiles/issue_peak.json

Keeps, per op, the best (lowest) ns per wave-instruction per SIMD over the waves-per-SIMD settings
measured with >= 2 waves, and derives what bench.py needs:
  valu_peak_G   chip-wide rate of the fastest VALU class (v_add_f32 / v_mul_f32 / v_fma_f32 / v_mov),
                G wave-instr/s, as measured (all 256 CUs x 4 SIMDs busy)
  class_ns      ns per wave-instruction per SIMD of each VALU class the PMC counters can separate
"""
import json
import sys


def main():
    best = {}
    for line in open(sys.argv[1]):
        f = line.split()
        if len(f) < 6 or line.startswith("#") or f[0] == "op":
            continue
        try:
            w, ns = int(f[1]), float(f[2])
        except ValueError:
            continue
        if w >= 2:
            best[f[0]] = min(best.get(f[0], 1e30), ns)
    simd = 256 * 4
    fast = [best[k] for k in ("k_add", "k_mul", "k_fma") if k in best]
    ns_fast = sum(fast) / len(fast)
    slow_keys = ["k_max3", "k_min", "k_cndmask_sgpr", "k_cmp_sgpr", "k_mbcnt", "k_lshl_add", "k_bfe"]
    slow = [best[k] for k in slow_keys if k in best]
    doc = {
        "_comment": "tools/issue_peak.hip on one MI355X (256 CUs): ns per wave64 instruction per SIMD with every SIMD busy, "
                    "best of 2/4/8 waves per SIMD; wall-clock based (the chip clocks itself under load)",
        "valu_peak_G": simd / min(fast),
        "clock_GHz": 2.4,
        "class_ns": {
            "fp32_add_mul_fma": ns_fast,
            "trans": (best.get("k_rcp", 0) + best.get("k_sqrt", 0)) / 2 if "k_rcp" in best else None,
            "other": sum(slow) / len(slow) if slow else None,
        },
        "per_op_ns": dict(sorted(best.items())),
    }
    json.dump(doc, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
