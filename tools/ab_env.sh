#!/bin/bash
# Run ON THE GPU BOX: the bench step under two values of an environment switch, two rounds, per-class times.   usage: tools/ab_env.sh VAR v1 v2 ...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
VAR=$1; shift
for round in 1 2; do for v in "$@"; do env $VAR=$v python3 bench.py --steps 6 --warmup 2 --no-strong --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_classes_ms']; print('$VAR=$v', d['value'], d['ms_per_step'], {n: round(x, 2) for n, x in k.items()})"; done; done
