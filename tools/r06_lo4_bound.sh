#!/bin/bash
# Run ON THE GPU BOX (round 6): what could a mixed-format (e2m1) second pass gain?  Timing-only ablation builds of gemm.hip phase 2 that move and read an
# operand with the byte count of an e2m1 image (make -C blim_amd/csrc ablate_p2: 4 = W, 5 = both) next to the shipped pass (0), cycles per K-step and the compensated bench step.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
mkdir -p gpurun_out
python3 tools/gemm_waits_lo6.py > gpurun_out/r06_lo4_waits.txt 2>&1
for round in 1 2; do for v in 0 4 5; do BLIM_LIB_PATH=$R/tools/bin/libblim_hip_ablate_p2_$v.so python3 bench.py --steps 6 --warmup 2 --no-strong --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['compensated_mode']; print('p2_$v plain', d['value'], d['ms_per_step'], 'compensated', c['value'], c.get('ms_per_step'))"; done; done > gpurun_out/r06_lo4_bench.txt 2>&1
cat gpurun_out/r06_lo4_waits.txt gpurun_out/r06_lo4_bench.txt
