#!/bin/bash
# Run ON THE GPU BOX (round 6, trip 2): the calibrator's false-reject table and the new eight-rank tests (item 3).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
mkdir -p gpurun_out
( timeout 1500 python3 -m pytest tests/test_main_driver.py -x -q -m gpu -k "eight_ranks or gpus_8" 2>&1 | tail -15 ) > gpurun_out/r06_eight_ranks.txt
{
for spec in "gaussian 4917 32" "gaussian 1000 16" "heavy7b 1000 16" "sink7b 1000 16" "heavy7b 4917 32"; do
  set -- $spec
  timeout 900 python3 tools/calibrator_false_rejects.py --weights $1 --n $2 --topk $3 --limit 40000 2>&1 | grep -v Warning | tail -2
done
} > gpurun_out/r06_calibrator_false_rejects.txt
cat gpurun_out/r06_eight_ranks.txt; grep -v '^{' gpurun_out/r06_calibrator_false_rejects.txt
