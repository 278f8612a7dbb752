#!/usr/bin/env python3
"""Register / scratch / LDS / occupancy table of every kernel in a .hip file
(hipcc -Rpass-analysis=kernel-resource-usage; cross-compiles, no GPU needed).

    python tools/kernel_resources.py blah2_amd/csrc/capi.hip [filter]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
        return dict(zip(names, out))
    except FileNotFoundError:
        return {n: n for n in names}


def main():
    src = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    extra = [a for a in sys.argv[3:]]
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-I", os.path.join(ROOT, "include"),
           "-I", os.path.join(ROOT, "blah2_amd", "csrc"), "-c", src, "-o", "/dev/null",
           "-Rpass-analysis=kernel-resource-usage", *extra]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    rec, name = {}, None
    keys = {"VGPRs": r"\bVGPRs: (\d+)", "AGPRs": r"AGPRs: (\d+)", "scratch": r"ScratchSize \[bytes/lane\]: (\d+)",
            "occ": r"Occupancy \[waves/SIMD\]: (\d+)", "vspill": r"VGPRs Spill: (\d+)", "lds": r"LDS Size \[bytes/block\]: (\d+)",
            "SGPRs": r"\bSGPRs: (\d+)"}
    for line in err.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            rec[name] = {}
            continue
        for k, pat in keys.items():
            m = re.search(pat, line)
            if m and name:
                rec[name][k] = int(m.group(1))
    dm = demangle(list(rec))
    for n, r in rec.items():
        d = dm[n]
        if flt and flt not in d:
            continue
        print(f"{d[:100]:100s} " + " ".join(f"{k}={v}" for k, v in r.items()))


if __name__ == "__main__":
    main()
