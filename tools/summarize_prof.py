#!/usr/bin/env python3
"""Condenses rocprofv3 output under gpurun_out/prof into the tracked summaries
under profiles/ (per round):

    python tools/summarize_prof.py r01

  profiles/<round>_kernel_stats.csv   rocprofv3 --kernel-trace --stats (our kernels)
  profiles/<round>_pmc.csv            per-kernel averages of every PMC pass found
  profiles/<round>_traffic.json       HBM bytes per launch for bench.py's roofline.traffic

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE
and WRITE_SIZE are in KiB, collected in separate --pmc passes; on gfx950
FETCH_SIZE counts 64 B per 128-B request, so it is doubled; WRITE_SIZE is taken
as reported (uncalibrated).
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof")
DST = os.path.join(ROOT, "profiles")


def short(name):
    for k in ("range_kernel", "doppler_tile_kernel", "doppler_fft_kernel", "doppler_dft_kernel", "metrics_kernel",
              "cfar1d_kernel", "cfar2d_kernel", "sat_rows_kernel", "sat_cols_kernel", "rotate_kernel",
              "clutter_corr_kernel", "clutter_fir_kernel", "clutter_solve_kernel", "clutter_reduce_kernel"):
        if k in name:
            return k
    return None


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    os.makedirs(DST, exist_ok=True)
    stats = os.path.join(SRC, "trace", "bench_kernel_stats.csv")
    if os.path.exists(stats):
        rows = [r for r in csv.DictReader(open(stats)) if short(r["Name"])]
        with open(os.path.join(DST, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "calls", "total_ns", "avg_ns", "pct_of_all_gpu_time", "min_ns", "max_ns", "stddev_ns"])
            for r in rows:
                w.writerow([r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                            r["MinNs"], r["MaxNs"], r["StdDev"]])
    agg = collections.defaultdict(list)
    meta = {}
    for path in sorted(glob.glob(os.path.join(SRC, "*", "bench_counter_collection.csv"))):
        for r in csv.DictReader(open(path)):
            k = short(r["Kernel_Name"])
            if not k:
                continue
            agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
            meta[k] = (r["Grid_Size"], r["Workgroup_Size"], r["LDS_Block_Size"], r["Scratch_Size"],
                       r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"])
    with open(os.path.join(DST, f"{tag}_pmc.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "counter", "dispatches", "mean_per_dispatch", "grid", "wg", "lds_bytes", "scratch",
                    "vgpr", "agpr", "sgpr"])
        for (k, c), v in sorted(agg.items()):
            w.writerow([k, c, len(v), f"{sum(v)/len(v):.6g}", *meta[k]])
    traffic = {}
    for k in {k for k, _ in agg}:
        fs = agg.get((k, "FETCH_SIZE"))
        ws = agg.get((k, "WRITE_SIZE"))
        if fs and ws:
            fetch = 2.0 * 1024.0 * sum(fs) / len(fs)
            write = 1024.0 * sum(ws) / len(ws)
            traffic[k] = {"fetch_bytes": fetch, "write_bytes": write, "hbm_bytes": fetch + write,
                          "note": "FETCH_SIZE KiB x2 (gfx950 64B-per-128B-request correction) + WRITE_SIZE KiB"}
    cfgp = os.path.join(SRC, "bench_config.json")
    out = {"round": tag, "kernels": traffic}
    if os.path.exists(cfgp):
        out["bench_config"] = json.load(open(cfgp))
    json.dump(out, open(os.path.join(DST, f"{tag}_traffic.json"), "w"), indent=1)
    print(open(os.path.join(DST, f"{tag}_kernel_stats.csv")).read() if os.path.exists(stats) else "no stats")
    print(json.dumps(traffic, indent=1))


if __name__ == "__main__":
    main()
