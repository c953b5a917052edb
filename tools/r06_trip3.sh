#!/bin/bash
# Run ON THE GPU BOX (round 6, trip 3): eight-rank tests, BASELINE configs 3 / 4 at size with the new calibration share, and the RMSNorm-fold bound.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
mkdir -p gpurun_out
( timeout 1500 python3 -m pytest tests/test_main_driver.py -q -m gpu -k "eight_ranks or gpus_8" 2>&1 | tail -15 ) > gpurun_out/r06_eight_ranks.txt
( timeout 1500 python3 -m pytest tests/test_configs_full_size.py -q -m gpu -s 2>&1 | grep -v Warning | tail -25 ) > gpurun_out/r06_configs.txt
for round in 1 2; do for lib in blim_amd/libblim_hip.so tools/bin/libblim_hip_ablate_normfold.so; do BLIM_LIB_PATH=$R/$lib python3 bench.py --steps 6 --warmup 2 --no-strong --no-cpu-baseline --no-compensated 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_classes_ms']; print('$lib', d['value'], d['ms_per_step'], 'rmsnorm', k.get('rmsnorm'), 'o', k['gemm_o_resid'], 'down', k['gemm_down_resid'], 'qkv', k['gemm_qkv_rope'], 'gateup', k['gemm_gateup_swiglu'])"; done; done > gpurun_out/r06_normfold_bench.txt 2>&1
cat gpurun_out/r06_eight_ranks.txt gpurun_out/r06_configs.txt gpurun_out/r06_normfold_bench.txt
