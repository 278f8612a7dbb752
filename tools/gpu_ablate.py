#!/usr/bin/env python3
"""Ablation timing of the range kernel (one process per variant; run via gpurun)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
names = {7: "full", 3: "no global loads", 5: "loads+arith (no LDS)", 6: "loads+LDS (no arith)", 1: "arith only",
         2: "LDS only", 4: "loads only", 0: "empty loop"}
for f in (int(x) for x in os.environ.get("ABL_F", "1024,2048").split(",")):
    for abl in (7, 3, 5, 6, 1, 2, 4, 0):
        env = dict(os.environ, BLAH2HIP_RANGE_ABLATE=str(abl), BLAH2HIP_FFT_LEN=str(f), DIAG_B=os.environ.get("DIAG_B", "8"))
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu_diag.py"), "timing"], env=env,
                             capture_output=True, text=True).stdout
        line = [l for l in out.splitlines() if l.startswith(f"F={f} ")]
        print(f"F={f} abl={abl} [{names[abl]:24s}] " + (line[0].split("kernels us/CPI:")[1] if line else out[-300:]), flush=True)
