#!/bin/bash
# Run ON THE GPU BOX: the bench step with alternative libraries (BLIM_LIB), two rounds, per-class times.   usage: tools/ab_lib.sh lib1.so lib2.so ...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
for round in 1 2; do for lib in "$@"; do BLIM_LIB_PATH=$R/$lib python3 bench.py --steps 6 --warmup 2 --no-strong --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_classes_ms']; print('$lib', d['value'], d['ms_per_step'], 'o', k['gemm_o_resid'], 'down', k['gemm_down_resid'], 'gateup', k['gemm_gateup_swiglu'])"; done; done
