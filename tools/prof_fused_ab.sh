#!/bin/bash
# rocprofv3 kernel stats of the fused-FIR A/B   (bash tools/prof_fused_ab.sh)
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/prof_ab
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_ab -o ab --output-format csv -- python $REPO/tools/gpu_fused_fir_ab.py ${1:-16} ${2:-10} > /tmp/ab.log 2>&1
f=$(find /tmp/prof_ab -name "*kernel_stats.csv" | head -1)
echo "stats file: $f"
tail -5 /tmp/ab.log
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(f'{r["Name"][:72]:72s} calls {r["Calls"]:>5s} avg_us {float(r["AverageNs"]) / 1e3:10.1f}  {r["Percentage"]}%')
PY
