#!/bin/bash
# Run ON THE GPU BOX: bench step with alternative libraries, attention class time.   usage: tools/ab_attn.sh lib1.so lib2.so ...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
for round in 1 2; do for lib in "$@"; do BLIM_LIB_PATH=$R/$lib python3 bench.py --steps 6 --warmup 2 --no-strong --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_classes_ms']; print('$lib', d['value'], d['ms_per_step'], 'attention', k['attention'])"; done; done
