#!/usr/bin/env python3
"""Regenerates the tables of DESIGN.md section 4 ("Measured") from profiles/<round>_bench*.json and the rocprofv3 summaries:
    python tools/update_design_measured.py r06
(after tools/profile_round.sh on the GPU box, tools/summarize_prof.py and tools/collect_bench.py here)."""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
P = os.path.join(ROOT, "profiles")


def B(n):
    return json.load(open(os.path.join(P, f"{tag}_bench{'_' + n if n else ''}.json")))


def ks(x):
    return {k["kernel"]: k for k in x["kernels"]}


def rocprof_avg(stats_tag, kernel_prefix):
    for row in csv.DictReader(open(os.path.join(P, f"{tag}_{stats_tag}_kernel_stats.csv"))):
        if kernel_prefix in row["kernel"]:
            return float(row["avg_ns"]) / 1e3, int(row["calls"])
    return None, 0


j = B("")
c = {x["baseline_config"]: x for x in j["configs"]}
k2, ka, k4 = ks(c["configs[2]"]), ks(c["configs[2] ambiguity only"]), ks(c["configs[4]"])
k1, kf = ks(c["configs[1], one CPI per launch"]), ks(c["configs[1], one CPI per launch, full chain"])
t2 = B("cfg3_full_twostage")
kt = {k["kernel"]: k for k in t2["roofline"]["kernels"]}
f2 = B("full")
kff = {k["kernel"]: k for k in f2["roofline"]["kernels"]}
sm = B("small")
ksm = {k["kernel"]: k for k in sm["roofline"]["kernels"]}
rl = j["roofline"]
kk = {k["kernel"]: k for k in rl["kernels"]}
r3, r1 = j["replay"]
rp_us, rp_calls = rocprof_avg("amb", rl["kernel"])
hot_us, _ = rocprof_avg("amb", "hot_columns_kernel")
new = f'''**Headline, BASELINE configs[1]** (2 MS/s, 1 s CPI, 513 × 411, fp32 planes resident in HBM, 256 CPIs per step):

| | |
|---|---|
| `value` | **{j["value"]/1e3:.1f} k CPIs/s** ({j["us_per_cpi"]:.2f} µs/CPI, {j["cells_per_s"]/1e9:.1f} Gcell/s); `headline_long` (0.5 s, {j["headline_long"]["steps"]} steps): {j["headline_long"]["value"]/1e3:.1f} k; under `torch.distributed.run`: {B("torchrun")["value"]/1e3:.1f} k; int16 `.rspduo` words: {B("i16")["value"]/1e3:.1f} k |
| dominant kernel `{rl["kernel"]}` | {rl["avg_launch_us"]:.0f} µs per launch by HIP events in the bench run, {rp_us:.1f} µs by rocprofv3 in its own run of the same command (`{tag}_amb_kernel_stats.csv`, {rp_calls} launches); algorithmic {rl["algorithmic_bytes_per_launch"]/1e9:.3f} GB → **{rl["achieved"]/1e3:.2f} TB/s = {rl["frac"]:.3f} of the 8 TB/s peak**; PMC traffic {rl["traffic"]/1e9:.3f} GB = {rl["traffic"]/rl["algorithmic_bytes_per_launch"]:.3f} × algorithmic |
| against what the memory system delivers in the same process | read-only kernel (`blah2hip_stream_read_dev`) {rl["read_ceiling"]/1e3:.2f} TB/s → {rl["frac_of_read_ceiling"]:.3f}; device-to-device copy {rl["copy_ceiling"]/1e3:.2f} TB/s of read + written bytes |
| VALU | {rl["valu"]["achieved_tflops"]:.1f} TFLOP/s of butterflies = {rl["valu"]["frac"]:.2f} of the fp32 vector peak |
| whole chain (`chain_frac`) | {rl["chain_frac"]:.3f}: Doppler {kk["doppler"]["us_per_cpi"]:.2f} µs/CPI ({kk["doppler"]["frac_hbm"]:.2f} of the peak on its own bytes), of which the hot-column launch {hot_us/256:.2f} ({hot_us:.1f} µs per 256 CPIs), metrics {kk["metrics"]["us_per_cpi"]:.2f} |
| CPU beside it | {j["cpu_baseline"]["value"]:.1f} CPIs/s: the NumPy / pocketfft restatement on 4 threads (`cpu_baseline`, 3 CPIs); {j["cpu_baseline"]["reference_source"]["value"]:.1f} CPIs/s: the reference's own sources on the shim FFT, one thread |

**The other configurations** (`configs[]` of the same default run, each with its oracle gate; in brackets the dedicated runs):

| configuration | CPIs/s | µs/CPI | kernels, µs/CPI (share of the HBM peak on own bytes) |
|---|---|---|---|
| configs[2] full chain: clutter filter (2047 taps) + map + 2-D CFAR, 32 CPIs × 2 streams, **FIR fused into the range kernel** | **{c["configs[2]"]["cpis_per_s"]:.0f}** [{B("cfg3_full")["value"]:.0f}] | {c["configs[2]"]["us_per_cpi"]:.1f} [{B("cfg3_full")["us_per_cpi"]:.1f}] | `range_fir` {k2["range"]["us_per_cpi"]:.1f} ({k2["range"]["frac_hbm"]:.2f}), corr {k2["clutter_corr"]["us_per_cpi"]:.1f} ({k2["clutter_corr"]["frac_hbm"]:.2f}), solve {k2["clutter_solve"]["us_per_cpi"]:.1f}, Doppler {k2["doppler"]["us_per_cpi"]:.1f} ({k2["doppler"]["frac_hbm"]:.2f}), cfar {k2["cfar"]["us_per_cpi"]:.1f} ({k2["cfar"]["frac_hbm"]:.2f}), reduce {k2["clutter_reduce"]["us_per_cpi"]:.1f} |
| the same on the two-stage chain (`--fir two-stage`; round 5's form + the time-domain tap) | [{t2["value"]:.0f}] | {t2["us_per_cpi"]:.1f} | range {kt["range"]["us_per_cpi"]:.1f} ({kt["range"]["frac_hbm"]:.2f}), FIR {kt["clutter_fir"]["us_per_cpi"]:.1f} ({kt["clutter_fir"]["frac_hbm"]:.2f}), corr {kt["clutter_corr"]["us_per_cpi"]:.1f}, solve {kt["clutter_solve"]["us_per_cpi"]:.1f}, Doppler {kt["doppler"]["us_per_cpi"]:.1f}, cfar {kt["cfar"]["us_per_cpi"]:.1f} |
| configs[2] ambiguity only, 32 CPIs | {c["configs[2] ambiguity only"]["cpis_per_s"]:.0f} [{B("cfg3")["value"]:.0f}] | {c["configs[2] ambiguity only"]["us_per_cpi"]:.1f} | range {ka["range"]["us_per_cpi"]:.1f} ({ka["range"]["frac_hbm"]:.2f}), Doppler {ka["doppler"]["us_per_cpi"]:.1f} ({ka["doppler"]["frac_hbm"]:.2f}) |
| configs[4] (20 MS/s, 2 s, 2049 × 411, fp16 storage), 8 CPIs | {c["configs[4]"]["cpis_per_s"]:.0f} [{B("cfg5")["value"]:.0f}] | {c["configs[4]"]["us_per_cpi"]:.1f} | range {k4["range"]["us_per_cpi"]:.1f} ({k4["range"]["frac_hbm"]:.2f}), Doppler {k4["doppler"]["us_per_cpi"]:.1f} ({k4["doppler"]["frac_hbm"]:.2f}) |
| configs[1], one CPI per launch | {c["configs[1], one CPI per launch"]["cpis_per_s"]:.0f} | {c["configs[1], one CPI per launch"]["us_per_cpi"]:.1f} (27.5 + the hot-column launch) | range {k1["range"]["us_per_cpi"]:.1f}, Doppler + hot columns {k1["doppler"]["us_per_cpi"]:.1f} |
| configs[1], one CPI per launch, filter (410 taps) + 1-D CFAR | {c["configs[1], one CPI per launch, full chain"]["cpis_per_s"]:.0f} [{B("full_b1")["value"]:.0f}] | {c["configs[1], one CPI per launch, full chain"]["us_per_cpi"]:.1f} | solve {kf["clutter_solve"]["us_per_cpi"]:.1f}, Doppler + hot columns {kf["doppler"]["us_per_cpi"]:.1f}, corr {kf["clutter_corr"]["us_per_cpi"]:.1f}, FIR {kf["clutter_fir"]["us_per_cpi"]:.1f}, range {kf["range"]["us_per_cpi"]:.1f}, cfar {kf["cfar"]["us_per_cpi"]:.1f}, reduce {kf["clutter_reduce"]["us_per_cpi"]:.1f} |
| configs[1] full chain at 256 CPIs per step | [{f2["value"]:.0f}] | {f2["us_per_cpi"]:.1f} | FIR {kff["clutter_fir"]["us_per_cpi"]:.1f} ({kff["clutter_fir"]["frac_hbm"]:.2f}), corr {kff["clutter_corr"]["us_per_cpi"]:.1f} ({kff["clutter_corr"]["frac_hbm"]:.2f}), range {kff["range"]["us_per_cpi"]:.1f} ({kff["range"]["frac_hbm"]:.2f}), Doppler {kff["doppler"]["us_per_cpi"]:.1f}, cfar {kff["cfar"]["us_per_cpi"]:.1f}, solve {kff["clutter_solve"]["us_per_cpi"]:.1f} |
| `small` (1 MS/s, 0.1 s), 1024 CPIs | [{sm["value"]/1e6:.2f} M] | {sm["us_per_cpi"]:.2f} | range {ksm["range"]["us_per_cpi"]:.2f} ({ksm["range"]["frac_hbm"]:.2f}) |

**Replay, PCIe-inclusive** (`replay[]` of the default run; never `value`): configs[3] at N = 1 — int16 capture in `/dev/shm` → pinned
ring → PCIe → range + Doppler + metrics + 1-D CFAR, {r3["cpis_timed"]} CPIs in {r3["seconds"]:.1f} s: **{r3["cpis_per_s"]:.0f} CPIs/s = {r3["effective_GBps"]:.1f} GB/s = {r3["frac_of_pinned_link"]:.2f} of the pinned-copy
rate measured beside it** ({r3["pinned_h2d_GBps"]:.1f} GB/s), GPU busy {100*r3["gpu_busy_share"]:.1f} % of the time; configs[1] geometry: {r1["cpis_per_s"]:.0f} CPIs/s ({r1["frac_of_pinned_link"]:.2f}, GPU busy {100*r1["gpu_busy_share"]:.1f} %).
Host-buffer boundary of the C++ classes (`e2e_host`): {j["e2e_host"]["classes_ms"]:.2f} ms per CPI for `blah2.cpp:264-287` on `IqData` FIFOs.

'''
path = os.path.join(ROOT, "DESIGN.md")
s = open(path).read()
a = s.index("**Headline, BASELINE configs[1]**")
b = s.index("**HBM traffic against algorithmic bytes**")
open(path, "w").write(s[:a] + new + s[b:])
print(new)
