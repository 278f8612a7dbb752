#!/usr/bin/env python3
"""Copies the bench JSON lines of a profiling round (gpurun_out/bench_rN*.log, the last line of each) and the replay
figures into profiles/ as <round>_bench*.json / <round>_replay.json:   python tools/collect_bench.py r03"""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
n = int(tag[1:])
for path in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", f"bench_r{n}*.log"))):
    name = os.path.basename(path)[len(f"bench_r{n}"):-4]  # "", "_b1", "_cfg3", ...
    try:
        line = [l for l in open(path).read().strip().splitlines() if l.startswith("{")][-1]
        j = json.loads(line)
    except Exception as e:
        print("skipped", path, e)
        continue
    out = os.path.join(ROOT, "profiles", f"{tag}_bench{name}.json")
    json.dump(j, open(out, "w"), indent=1)
    print(out, round(j["value"], 1), j["unit"], "parity", (j.get("parity") or {}).get("pass"))
rp = os.path.join(ROOT, "gpurun_out", "replay.json")
if os.path.exists(rp):
    shutil.copy(rp, os.path.join(ROOT, "profiles", f"{tag}_replay.json"))
