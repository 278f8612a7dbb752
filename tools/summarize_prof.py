#!/usr/bin/env python3
"""Condenses rocprofv3 output under gpurun_out/prof/<tag>/ (tools/profile_round.sh) into the tracked
summaries under profiles/ (per round):

    python tools/summarize_prof.py r02

  profiles/<round>_<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats (our kernels)
  profiles/<round>_<tag>_pmc.csv            per-kernel averages of every PMC pass found
  profiles/<round>_<tag>_traffic.json       HBM bytes per launch for bench.py's roofline.traffic
  profiles/<round>_pmc_calibration.json     FETCH_SIZE / WRITE_SIZE against known byte counts (pmccal)

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE are
in KiB, collected in separate --pmc passes; on gfx950 FETCH_SIZE counts 64 B per 128-B request, so it
is doubled.  tools/membench/pmccal.hip calibrates both on this engine's access patterns: the x2 holds
for 8 and 16 B/lane streaming reads; WRITE_SIZE is exact for whole lines and counts 32-byte sectors
for partial-line row segments (a 64-byte segment at an arbitrary 8-byte phase touches 2.75 sectors).
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

KERNELS = ("range_fir_kernel", "taps_spectrum_kernel", "hot_columns_kernel", "leak_fix_kernel", "rangeps_kernel", "rangew1k_kernel", "rangew_kernel", "range8_kernel", "range_kernel", "doppler_tilew2_kernel", "doppler_tilew_kernel", "doppler_tilem_kernel", "doppler_tile1k_kernel", "doppler_sub1k_kernel", "doppler_tile_kernel", "doppler_fft_kernel",
           "doppler_dft_kernel", "metrics_kernel", "cfar1d_kernel", "cfar2d_stream_kernel", "cfar2d_tile_kernel", "cfar2d_kernel", "sat_rows_kernel", "sat_cols_kernel",
           "rotate_kernel", "clutter_corr_half_kernel", "clutter_corr_kernel", "clutter_fir_kernel", "clutter_solve_la_kernel", "clutter_solve_kernel", "solve_epoch_kernel", "clutter_reduce_kernel",
           "db_map_kernel", "cal_")


# Kernels whose global reads are 64-byte (or shorter) pieces of 128-byte lines, BY INSTANTIATION: FETCH_SIZE counts
# 64 B per request, so those are counted exactly (factor 1) and whole-line readers are under-counted by 2.
# doppler_tile_kernel<16> reads whole 128-byte rows of a 16-column tile (x2), <8> the 64-byte half rows (x1);
# the one-wave / two-wave tile kernels read 64-byte half rows, doppler_tilem_kernel<16> 32-byte quarter rows.
HALF_LINE_READERS = ("doppler_tile_kernel<8>", "doppler_tilem_kernel<8>", "doppler_tilem_kernel<16>", "doppler_tilew_kernel",
                     "doppler_tilew2_kernel")
# kernels that keep their template argument in the summaries (their instantiations differ in access pattern)
KEEP_TEMPLATE = ("doppler_tile_kernel", "doppler_tilem_kernel", "clutter_solve_la_kernel", "clutter_solve_kernel")


def short(name):
    for k in KERNELS:
        if k in name:
            if k == "cal_":
                return name.split("(")[0].replace("void ", "")
            if k in KEEP_TEMPLATE and k + "<" in name:
                i = name.index(k + "<")
                return name[i:name.index(">", i) + 1].replace(" ", "")
            return k
    return None


def algorithmic_bytes(bench_config):
    """Algorithmic bytes per launch of each kernel for a bench command (the same figures as bench.py's
    roofline.kernels[]), from tools' description {"config", "batch", "fmt", "chain"}."""
    sys.path.insert(0, ROOT)
    import bench
    (dmin, dmax, fmin, fmax, fs, n), _ = bench.CONFIGS[bench_config["config"]]
    nC = dmax - dmin + 1
    nD = 2 * int(fmax * (n / fs)) + 1  # symmetric Doppler limits, resolution fs/n (Ambiguity.cpp:25-37)
    s_in = 8 if bench_config.get("fmt", "c32") == "c32" else 4
    return bench.algorithmic_bytes(n, s_in, nD * nC, bench_config["batch"], bench_config.get("cfar", "2d"))


def summarize(src, tag, prefix):
    stats = os.path.join(src, "trace", "bench_kernel_stats.csv")
    if os.path.exists(stats):
        rows = [r for r in csv.DictReader(open(stats)) if short(r["Name"])]
        with open(os.path.join(DST, f"{prefix}_kernel_stats.csv"), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "calls", "total_ns", "avg_ns", "pct_of_all_gpu_time", "min_ns", "max_ns", "stddev_ns"])
            for r in rows:
                w.writerow([r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                            r["MinNs"], r["MaxNs"], r["StdDev"]])
    agg = collections.defaultdict(list)
    meta = {}
    for path in sorted(glob.glob(os.path.join(src, "*", "*_counter_collection.csv"))):
        for r in csv.DictReader(open(path)):
            k = short(r["Kernel_Name"])
            if not k:
                continue
            agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
            meta[k] = (r["Grid_Size"], r["Workgroup_Size"], r["LDS_Block_Size"], r["Scratch_Size"],
                       r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"])
    if not agg:
        return None
    with open(os.path.join(DST, f"{prefix}_pmc.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "counter", "dispatches", "mean_per_dispatch", "grid", "wg", "lds_bytes", "scratch",
                    "vgpr", "agpr", "sgpr"])
        for (k, c), v in sorted(agg.items()):
            w.writerow([k, c, len(v), f"{sum(v)/len(v):.6g}", *meta[k]])
    cfgp = os.path.join(src, "bench_config.json")
    bench_config = json.load(open(cfgp)) if os.path.exists(cfgp) else None
    algo = algorithmic_bytes(bench_config) if bench_config else {}
    traffic = {}
    for k in sorted({k for k, _ in agg}):
        fs = agg.get((k, "FETCH_SIZE"))
        ws = agg.get((k, "WRITE_SIZE"))
        if fs and ws:
            # FETCH_SIZE counts 64 B per request: a whole-line (128 B) request is under-counted by 2, a
            # half-line request (the 8-column Doppler tile reads 64 of every 128 B) is counted exactly
            # (profiles/*_pmc_calibration.json: cal_read8/16 -> 0.5, cal_read_half -> 1.0)
            raw = 1024.0 * sum(fs) / len(fs)
            factor, why = 2.0, "64 B counted per whole-line request"
            if any(k.startswith(h) for h in HALF_LINE_READERS):
                factor, why = 1.0, "row pieces of 64 B or less: requests counted exactly"
                # ... unless the pieces of a line are read by sibling workgroups of ONE XCD (the XCD-aware tile walks): the
                # L2 then asks for whole lines.  Decided by the data: a Doppler kernel reads half of its algorithmic bytes,
                # and a raw figure far below that can only be the whole-line under-count.
                if k.startswith("doppler") and algo.get("doppler") and raw < 0.75 * (algo["doppler"] / 2):
                    factor, why = 2.0, "row pieces merged in one XCD's L2: whole-line requests, 64 B counted per request"
            fetch = factor * raw
            write = 1024.0 * sum(ws) / len(ws)
            traffic[k] = {"fetch_bytes": fetch, "write_bytes": write, "hbm_bytes": fetch + write, "fetch_factor": factor,
                          "note": f"FETCH_SIZE KiB x{factor:g} ({why}) + WRITE_SIZE KiB (32-byte sectors)"}
    cfgp = os.path.join(src, "bench_config.json")
    out = {"round": tag, "kernels": traffic}
    bad = []
    if bench_config:
        out["bench_config"] = bench_config
        names = {"rangew1k_kernel": "range", "rangew2_kernel": "range", "rangew_kernel": "range", "range_kernel": "range", "range8_kernel": "range", "doppler": "doppler",
                 "clutter_corr_half_kernel": "clutter_corr", "clutter_corr_kernel": "clutter_corr",
                 "clutter_fir_kernel": "clutter_fir", "cfar2d": "cfar", "cfar1d_kernel": "cfar"}
        for k, t in traffic.items():
            key = next((v for pre, v in names.items() if k.startswith(pre)), None)
            if key and key in algo:
                t["algorithmic_bytes"] = algo[key]
                t["hbm_over_algorithmic"] = t["hbm_bytes"] / algo[key]
                # a kernel cannot move fewer bytes than its inputs and outputs: that is a wrong fetch factor
                # (3 % slack: lines the 256 MB Infinity Cache still holds are not fetched from HBM)
                if t["hbm_bytes"] < 0.97 * algo[key]:
                    bad.append(f"{prefix}: {k} reports {t['hbm_bytes']:.4g} B < {algo[key]:.4g} B algorithmic")
    json.dump(out, open(os.path.join(DST, f"{prefix}_traffic.json"), "w"), indent=1)
    if bad:
        raise SystemExit("summarize_prof: impossible traffic figure(s): " + "; ".join(bad))
    return traffic


def calibration(tag):
    src = os.path.join(ROOT, "gpurun_out", "cal")
    known = {"cal_read8": 2 ** 30, "cal_read16": 2 ** 30, "cal_read_half": 2 ** 29, "cal_write8": 2 ** 30,
             "cal_write_seg<32>": 32 * (2 ** 30 // 3288), "cal_write_seg<64>": 64 * (2 ** 30 // 3288),
             "cal_write_seg<128>": 128 * (2 ** 30 // 3288)}
    res = {}
    for kind, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        path = os.path.join(src, kind, "cal_counter_collection.csv")
        if not os.path.exists(path):
            continue
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter:
                agg[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            if k in known and ((kind == "fetch") == ("read" in k)):
                raw = 1024.0 * sum(v) / len(v)
                res[k] = {"counter": counter, "known_bytes": known[k], "reported_bytes_raw_KiB_x1024": raw,
                          "reported_over_known": raw / known[k]}
    if res:
        json.dump({"round": tag, "tool": "tools/membench/pmccal.hip", "patterns": res}, open(os.path.join(DST, f"{tag}_pmc_calibration.json"), "w"), indent=1)
    return res


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
    os.makedirs(DST, exist_ok=True)
    for sub in sorted(os.listdir(SRC)) if os.path.isdir(SRC) else []:
        src = os.path.join(SRC, sub)
        if os.path.isdir(os.path.join(src, "trace")) or glob.glob(os.path.join(src, "pmc_*")):
            t = summarize(src, tag, f"{tag}_{sub}")
            print(sub, json.dumps(t, indent=1) if t else "no counters")
    print(json.dumps(calibration(tag), indent=1))


if __name__ == "__main__":
    main()
