#!/bin/bash
# PMC passes of the two one-wave range kernels on the headline workload (run through gpurun); summaries to gpurun_out/pmc1k/
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc1k; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for tag in w2k w1k; do
  if [ $tag = w1k ]; then A="--fft-len 1024 --range-kernel wave1k"; else A=""; fi
  B="python $REPO/bench.py $A --steps 12 --warmup 3 --no-cpu-baseline --no-parity"
  i=0
  for pass in "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA" "FETCH_SIZE" "SQ_INSTS_VALU_MFMA_I8 SQ_ACTIVE_INST_MISC SQ_INSTS_FLAT SQ_ACTIVE_INST_ANY"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $pass -d $OUT/${tag}_p$i -o bench --output-format csv -- $B > $OUT/${tag}_p$i.log 2>&1 || echo "pass $i failed ($tag)"
  done
done
python - <<PY
import csv, glob, collections
for tag in ("w2k", "w1k"):
    acc = collections.defaultdict(list)
    for f in glob.glob("$OUT/%s_p*/**/*counter_collection.csv" % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            if "range" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(tag, {k: "%.4g" % (sum(v) / len(v)) for k, v in sorted(acc.items())})
PY
