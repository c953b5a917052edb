mkdir -p gpurun_out/r05
echo "=== bench --gpus 2"; BLIM_DIST_BACKEND=gloo BLIM_FORCE_DEVICE=0 python bench.py --gpus 2 --steps 2 --warmup 1 --queries 8 --strong-n 96 2>/dev/null | tail -1 | cut -c1-300
echo "=== narrow GEMM experiment"
for round in 1 2; do
 for v in "blim_amd/libblim_hip.so 8" "tools/bin/libblim_hip_rope_inline.so 8" "blim_amd/libblim_hip.so 32" "blim_amd/libblim_hip.so 128"; do set -- $v
  BLIM_LIB_PATH=$PWD/$1 BLIM_GEMM_GROUP_M=$2 python bench.py --steps 6 --warmup 2 --no-strong --no-cpu-baseline --no-compensated 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_classes_ms']; print('round $round lib $1 group_m $2:', d['value'], 'pairs/s', d['ms_per_step'], 'ms | qkv', k['gemm_qkv_rope'], 'o', k['gemm_o_resid'], 'down', k['gemm_down_resid'], 'gateup', k['gemm_gateup_swiglu'])"
 done; done
echo "=== rope inline parity on the benched step"; BLIM_LIB_PATH=$PWD/tools/bin/libblim_hip_rope_inline.so python -m pytest tests/test_gpu_parity.py -q -k "benched_step" 2>&1 | grep -E "full7b_bench|passed|failed"
echo "=== planner contention"; python tools/planner_contention.py --gpu-rank 2>&1 | grep -v "^{" | tail -16
