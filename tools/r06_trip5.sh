#!/bin/bash
# Run ON THE GPU BOX (round 6, trip 5): the e2m3 second pass on bf16 engines: parity lines, the non-headline dtypes' bench lines, configs 3 / 4 with the adjusted rate.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
mkdir -p gpurun_out
( timeout 1200 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -s -k "e2m3_second_pass" 2>&1 | grep -v Warning | tail -30 ) > gpurun_out/r06_bf16_lo6_tests.txt
( timeout 900 python3 -m pytest tests/test_configs_full_size.py tests/test_isa_guards.py -q -s 2>&1 | grep -v Warning | tail -12 ) > gpurun_out/r06_configs2.txt
python3 bench.py --dtype bf16 --vtg-precise full --no-strong --no-cpu-baseline > gpurun_out/r06_bf16full_bench.json 2>/dev/null
python3 bench.py --dtype bf16 --vtg-precise full --second-pass e2m3 --no-strong --no-cpu-baseline > gpurun_out/r06_bf16full_e2m3_bench.json 2>/dev/null
python3 bench.py --dtype f8 --no-strong --no-cpu-baseline > gpurun_out/r06_f8_bench.json 2>/dev/null
python3 bench.py --no-strong --no-cpu-baseline > gpurun_out/r06_f16_samebox_bench.json 2>/dev/null
cat gpurun_out/r06_bf16_lo6_tests.txt gpurun_out/r06_configs2.txt
for f in bf16full bf16full_e2m3 f8 f16_samebox; do python3 -c "
import json,sys
d=json.loads(open('gpurun_out/r06_${f}_bench.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['dtype'], d['config']['vtg_compensated'], d['roofline']['frac'], (d.get('compensated_mode') or {}).get('value'))"; done
