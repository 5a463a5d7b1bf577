#!/usr/bin/env python3
"""Compact table of the resource usage of every gfx950 kernel of libtgsf (registers, spills, scratch, occupancy).

    python tools/kernel_resources.py [--all]        # default: only kernels with spills or scratch, and the k_mid_* family

Runs `make -C tgsfilter_amd/csrc asm` (hipcc -Rpass-analysis=kernel-resource-usage) and folds its remarks."""
import re
import subprocess
import sys
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def collect():
    out = subprocess.run(["make", "-B", "-C", os.path.join(ROOT, "tgsfilter_amd", "csrc"), "asm"], capture_output=True, text=True)
    text = out.stdout + out.stderr
    rows, cur = [], None
    for line in text.splitlines():
        m = re.search(r"remark: [^ ]+ (?:Function Name|Name): (\S+)", line) or re.search(r"Function Name: (\S+)", line) or re.search(r" Name: (\S+) \[", line)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
            continue
        if cur is None:
            continue
        for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r" AGPRs: (\d+)"), ("sgpr", r" SGPRs: (\d+)"), ("vspill", r"VGPRs Spill: (\d+)"),
                         ("sspill", r"SGPRs Spill: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"),
                         ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(pat, line)
            if m:
                cur[key] = int(m.group(1))
    return rows


def demangle(names):
    try:
        p = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"], input="\n".join(names), capture_output=True, text=True)
        return p.stdout.splitlines()
    except OSError:
        return names


def main():
    rows = collect()
    pretty = demangle([r["name"] for r in rows])
    show_all = "--all" in sys.argv
    print("%-64s %5s %6s %6s %7s %4s %7s" % ("kernel", "VGPR", "vspill", "sspill", "scratch", "occ", "LDS"))
    bad = 0
    for r, n in zip(rows, pretty):
        n = re.sub(r"\(.*$", "", n).replace("tgsf::", "").replace("void ", "")
        spilled = r.get("vspill", 0) > 0
        if spilled and "k_mid_" in n:
            bad += 1
        if not show_all and not (spilled or r.get("scratch", 0) or "k_mid_" in n):
            continue
        print("%-64s %5d %6d %6d %7d %4d %7d" % (n[:64], r.get("vgpr", 0), r.get("vspill", 0), r.get("sspill", 0), r.get("scratch", 0), r.get("occ", 0), r.get("lds", 0)))
    print("k_mid_* kernels with spilled VGPRs: %d" % bad)
    return 0


if __name__ == "__main__":
    sys.exit(main())
