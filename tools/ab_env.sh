for i in 1 2; do for w in 0 1; do BLIM_RMSNORM_WIDE=$w python bench.py --steps 5 --warmup 2 --no-strong --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wide=$w', d['value'], d['ms_per_step'], d['kernel_classes_ms']['rmsnorm'], d['kernel_classes_ms']['gemm_gateup_swiglu'])"; done; done
