#!/bin/bash
# Run ON THE GPU BOX (round 6, trip 7): the tests added after the last full run, then a fuzz soak on fresh seeds (fused + literal vs the numpy oracle) in every 16-bit configuration
# incl. the bf16 engine with the e2m3 second pass.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
mkdir -p gpurun_out
( timeout 1200 python3 -m pytest tests/test_configs_full_size.py tests/test_gpu_parity.py tests/test_main_driver.py -q -m gpu -s -k "config5 or second_pass_auto or watchdog or e2m3_second_pass" 2>&1 | grep -v Warning | tail -25 ) > gpurun_out/r06_new_tests.txt
cat gpurun_out/r06_new_tests.txt | cut -c1-600
{
timeout 600 python3 tools/fuzz_more.py 120 9000 2>&1 | tail -3
BLIM_FUZZ_DIMS=h512 timeout 600 python3 tools/fuzz_more.py 80 9500 2>&1 | tail -3
BLIM_DTYPE=bf16 timeout 600 python3 tools/fuzz_more.py 80 9800 2>&1 | tail -3
BLIM_DTYPE=bf16 BLIM_PRECISE_LO6=1 BLIM_FUZZ_DIMS=h512 timeout 600 python3 tools/fuzz_more.py 80 9900 2>&1 | tail -3
} > gpurun_out/r06_fuzz_soak.txt 2>&1
cat gpurun_out/r06_fuzz_soak.txt | cut -c1-400
