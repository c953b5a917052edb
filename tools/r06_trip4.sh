#!/bin/bash
# Run ON THE GPU BOX (round 6, trip 4): the whole GPU suite on the refactored tree, then the round's profiles (kernel trace + PMC passes) of the bench command, plain and compensated.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
mkdir -p gpurun_out
( timeout 1700 python3 -m pytest tests -q -m gpu -s --durations=25 2>&1 | grep -v Warning ) > gpurun_out/r06a_gputest.txt
tail -5 gpurun_out/r06a_gputest.txt
python3 bench.py > gpurun_out/r06a_bench.json 2> gpurun_out/r06a_bench.err; tail -c 600 gpurun_out/r06a_bench.json
bash tools/collect_profiles.sh r06a > /dev/null 2>&1
bash tools/collect_profiles.sh r06afull --vtg-precise full > /dev/null 2>&1
ls gpurun_out/prof_r06a gpurun_out/prof_r06afull
