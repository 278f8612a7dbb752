#!/usr/bin/env python3
"""Copies the bench JSON lines of a profiling round (gpurun_out/bench_rN*.log, the last line of each) and the replay
figures into profiles/ as <round>_bench*.json / <round>_replay.json:   python tools/collect_bench.py r03

bench.py attaches the PMC traffic of the newest matching profiles/*_traffic.json it finds; on the GPU box that is still
the PREVIOUS round's file (this round's counters are condensed afterwards, here).  The traffic fields of the copied
lines are therefore re-attached from this round's <round>_<tag>_traffic.json of the same command (same rule as bench.py)."""
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
    cfgk = (j["config"].get("workload", "").split(":")[0], j["config"]["batch_cpis_per_step"], j["config"]["fmt"], j["config"]["chain"])
    for tp in sorted(glob.glob(os.path.join(ROOT, "profiles", f"{tag}_*_traffic.json"))):
        tj = json.load(open(tp))
        bc = tj.get("bench_config", {})
        names = {"cfg2": "BASELINE configs[1]", "cfg3": "BASELINE configs[2]", "cfg5": "BASELINE configs[4]"}
        if (names.get(bc.get("config")), bc.get("batch"), bc.get("fmt"), bc.get("chain", "amb")) != cfgk:
            continue
        sys.path.insert(0, ROOT)
        prof_names = {"range": ("rangeps_kernel", "rangew_kernel", "range_kernel", "range8_kernel"), "doppler": ("doppler_",),
                      "metrics": ("metrics_kernel",), "cfar": ("cfar2d_stream_kernel", "cfar2d_tile_kernel", "cfar2d_kernel", "cfar1d_kernel"),
                      "sat_rows": ("sat_rows_kernel",), "sat_cols": ("sat_cols_kernel",), "rotate": ("rotate_kernel",),
                      "clutter_corr": ("clutter_corr_half_kernel", "clutter_corr_kernel"), "clutter_fir": ("clutter_fir_kernel",),
                      "clutter_solve": ("clutter_solve_la_kernel", "clutter_solve_kernel"), "clutter_reduce": ("clutter_reduce_kernel",)}
        rk = j["roofline"].get("kernel", "range_kernel")  # the range kernel this line ran, no other
        prof_names["range"] = tuple(k for k in tj["kernels"] if k == rk or k.startswith(rk + "<"))
        for e in j["roofline"]["kernels"]:
            e.pop("traffic", None)
            e.pop("traffic_over_algorithmic", None)
            hit = [v for k, v in tj["kernels"].items() if any(k.startswith(pre) for pre in prof_names.get(e["kernel"], ()))]
            if hit:
                e["traffic"] = sum(h["hbm_bytes"] for h in hit)
                if "algorithmic_bytes" in e:
                    e["traffic_over_algorithmic"] = e["traffic"] / e["algorithmic_bytes"]
        j["roofline"]["traffic"] = next((e.get("traffic") for e in j["roofline"]["kernels"] if e["kernel"] == "range"), None)
        j["roofline"]["traffic_source"] = os.path.relpath(tp, ROOT) + " (re-attached by tools/collect_bench.py)"
        break
    out = os.path.join(ROOT, "profiles", f"{tag}_bench{name}.json")
    json.dump(j, open(out, "w"), indent=1)
    print(out, round(j["value"], 1), j["unit"], "parity", (j.get("parity") or {}).get("pass"))
rp = os.path.join(ROOT, "gpurun_out", "replay.json")
if os.path.exists(rp):
    shutil.copy(rp, os.path.join(ROOT, "profiles", f"{tag}_replay.json"))
