#!/bin/bash
# GPU box: PMC counters of the stream and the tile 2-D CFAR kernels on 64 synthetic cfg 3 maps (tools/gpu_cfar_diag.py),
# each counter set in its own run, only --kernel-trace beside --pmc.  Output: gpurun_out/cfar_pmc/<set>/...
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/cfar_pmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $OUT/$tag -o diag --output-format csv -- python $REPO/tools/gpu_cfar_diag.py cfg3 64 stream,tile 0 > $OUT/$tag.log 2>&1 || echo "pass failed: $set"
done
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o diag --output-format csv -- python $REPO/tools/gpu_cfar_diag.py cfg3 64 stream,tile 0 > $OUT/trace.log 2>&1
ls $OUT | head -20
