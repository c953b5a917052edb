#!/bin/bash
# Run ON THE GPU BOX: compensated paths (fp16 TVG passes at the reference's shapes, the bf16 parity mode on the headline step) with alternative libraries, two
# rounds each.   usage: tools/ab_split.sh lib1.so lib2.so ...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
for round in 1 2; do for lib in "$@"; do
  BLIM_LIB_PATH=$R/$lib python3 tools/pass_bench.py --n 96 --shape ref --dtype f16 --reps 3 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())['ref']; print('$lib', 'f16 ref-shaped:', ', '.join(f\"{r['pass'].split('(')[0].strip()} {r['pairs_per_s']}\" for r in d))"
  BLIM_LIB_PATH=$R/$lib python3 bench.py --dtype bf16 --vtg-precise full --steps 6 --warmup 2 --no-strong --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_classes_ms']; print('$lib', 'bf16 parity mode:', d['value'], 'pairs/s', d['ms_per_step'], 'ms; qkv', k['gemm_qkv_rope'], 'attn', k['attention'], 'gateup', k['gemm_gateup_swiglu'])"
done; done
