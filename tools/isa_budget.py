#!/usr/bin/env python3
"""ISA-level budget of the pooled kernel's loop: static instruction counts per operation kind, from the
compiler's own assembly.  render_kernels.hip is compiled with -DRT_ISA_MARKS (assembler comments at the operations'
boundaries, no instruction), one instantiation is cut out, and the instructions between consecutive marks are classified.
The hot blocks of an operation are contiguous; code the compiler moved out of line (the cold refill of SHADE) shows up in
the range behind the last operation -- the table says which range is which.  No GPU needed.

usage: isa_budget.py [mangled-name-fragment ...]   (default: the rgbbox and irreg production instantiations)"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "raytracers_amd", "csrc", "render_kernels.hip")
DEFAULT = ["pooled_kernelILi1024ELb1ELb0ELb0ELi0ELb0E", "pooled_kernelILi1024ELb0ELb0ELb0ELi0ELb0E", "pooled_kernelILi1024ELb1ELb0ELb1ELi0ELb1E",
           "pooled_kernelILi1024ELb0ELb0ELb1ELi0ELb1E"]
NAMES = {"pooled_kernelILi1024ELb1ELb0ELb0ELi0ELb0E": "pooled_kernel<1024, ALL_LDS> (rgbbox: the whole scene in LDS; batches, frames beyond the pixel list's gate)",
         "pooled_kernelILi1024ELb0ELb0ELb0ELi0ELb0E": "pooled_kernel<1024> (irreg, 10^6 spheres: node prefix in LDS, the rest through buffer_load)",
         "pooled_kernelILi1024ELb0ELb0ELb0ELi1ELb0E": "pooled_kernel<1024, COLD> (small ordered frames, pixel_order = 0)",
         "pooled_kernelILi1024ELb0ELb0ELb0ELi2ELb0E": "pooled_kernel<1024, DONATE> (unordered frames)",
         "pooled_kernelILi1024ELb1ELb0ELb1ELi0ELb1E": "pooled_kernel<1024, ALL_LDS, SOLO, ORD> (rgbbox: ordered single frames through the pixel list)",
         "pooled_kernelILi1024ELb0ELb0ELb1ELi0ELb1E": "pooled_kernel<1024, SOLO, ORD> (irreg: ordered single frames through the pixel list)",
         "pooled_kernelILi1024ELb0ELb0ELb0ELi0ELb1E": "pooled_kernel<1024, ORD> (lists without a one-pixel class)"}

FP32 = re.compile(r"^v_(add|sub|subrev|mul|fma|mac|fmac|mad)_(f32|legacy_f32)|^v_pk_(add|mul|fma)_f32")
SEL = re.compile(r"^v_(cndmask|min|max|min3|max3|med3|cmp|cmpx)")
XLANE = re.compile(r"^v_(readlane|readfirstlane|writelane|mbcnt|permlane)|_dpp|^v_mov_b32_dpp")
TRANS = re.compile(r"^v_(rcp|rsq|sqrt|exp|log|sin|cos|div_scale|div_fmas|div_fixup|frexp|ldexp)")


def classify(op):
    if op.startswith("v_"):
        if XLANE.search(op):
            return "VALU cross-lane"
        if FP32.match(op):
            return "VALU fp32 add/mul/fma"
        if TRANS.match(op):
            return "VALU div/sqrt/rcp"
        if SEL.match(op):
            return "VALU select/min/max/cmp"
        return "VALU int/mov/other"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "VMEM"
    if op.startswith(("s_load", "s_buffer_load")):
        return "SMEM"
    if op in ("s_waitcnt", "s_nop", "s_sleep"):
        return "wait/nop"
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc", "s_endpgm")):
        return "branch"
    if op.startswith("s_"):
        return "SALU"
    return "other"


def main():
    want = sys.argv[1:] or DEFAULT
    asm = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-DRT_ISA_MARKS",
                          "--cuda-device-only", "-S", "-I" + os.path.dirname(SRC), SRC, "-o", "-"], capture_output=True, text=True, check=True).stdout
    lines = asm.splitlines()
    for frag in want:
        start = next((i for i, l in enumerate(lines) if l.startswith("_ZN3rtk13" + frag) and l.rstrip().endswith(":") or
                      (l.startswith("_ZN3rtk13" + frag) and ": " in l)), None)
        if start is None:
            print(f"## {frag}: not found\n")
            continue
        end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
        meta = {}
        for l in lines[end:end + 120]:
            m = re.match(r"\s*;\s*(NumVgprs|NumSgprs|ScratchSize|Occupancy|codeLenInByte):\s*(\d+)", l)
            if m:
                meta[m.group(1)] = int(m.group(2))
        ranges, cur, acc = [], "(kernel entry)", collections.Counter()
        for l in lines[start + 1:end + 1]:
            t = l.strip()
            m = re.match(r";\s*RT_MARK (\w+)", t)
            if m:
                ranges.append((cur, acc))
                cur, acc = m.group(1), collections.Counter()
                continue
            if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
                continue
            acc[classify(t.split()[0])] += 1
            # LEAF's body ends where it jumps back to the loop's head; what follows in layout order is code of OTHER operations
            # that the compiler moved out of line (SHADE's refill, the ticket draw)
            if cur == "LEAF_BEGIN" and t.split()[0] == "s_branch":
                ranges.append((cur, acc))
                cur, acc = "(out of line: SHADE's cold blocks -- refill, ticket draw, re-intersection)", collections.Counter()
        ranges.append((cur, acc))
        cols = ["VALU fp32 add/mul/fma", "VALU select/min/max/cmp", "VALU int/mov/other", "VALU cross-lane", "VALU div/sqrt/rcp", "SALU", "LDS",
                "VMEM", "SMEM", "branch", "wait/nop"]
        print(f"## {NAMES.get(frag, frag)}\n")
        print("registers: " + ", ".join(f"{k} {v}" for k, v in meta.items()) + "\n")
        print("| range (from this mark to the next one, in layout order) | VALU total | " + " | ".join(cols) + " |")
        print("|---|---|" + "---|" * len(cols))
        for name, a in ranges:
            valu = sum(v for k, v in a.items() if k.startswith("VALU"))
            print(f"| {name} | {valu} | " + " | ".join(str(a.get(c, 0)) for c in cols) + " |")
        print()


if __name__ == "__main__":
    main()
