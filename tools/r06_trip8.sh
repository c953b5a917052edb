#!/bin/bash
# Run ON THE GPU BOX (round 6, trip 8): the whole GPU suite and the default bench line on the round's final tree.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
mkdir -p gpurun_out
( timeout 1900 python3 -m pytest tests -q -m gpu -s --durations=12 2>&1 | grep -v Warning ) > gpurun_out/r06c_gputest.txt
tail -16 gpurun_out/r06c_gputest.txt
python3 bench.py > gpurun_out/r06c_bench.json 2> gpurun_out/r06c_bench.err; tail -c 300 gpurun_out/r06c_bench.json
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
