#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out
python -m pytest tests -m gpu -q > $OUT/pytest_final.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_final.log; tail -n 4 $OUT/pytest_final.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -n 2 $OUT/smoke.log
bash tools/profile_round.sh > $OUT/profile_round.log 2>&1; tail -n 9 $OUT/profile_round.log | cut -c1-160
