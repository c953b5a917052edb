#!/bin/bash
# Run ON THE GPU BOX (round 6, trip 6): the whole GPU suite on the round's final tree, the default bench line, and the bench command's profiles (kernel trace + PMC passes).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
TAG=${1:-r06b}
mkdir -p gpurun_out
( timeout 1900 python3 -m pytest tests -q -m gpu -s --durations=15 2>&1 | grep -v Warning ) > gpurun_out/${TAG}_gputest.txt
tail -4 gpurun_out/${TAG}_gputest.txt
python3 bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; tail -c 400 gpurun_out/${TAG}_bench.json
bash tools/collect_profiles.sh ${TAG} > /dev/null 2>&1
bash tools/collect_profiles.sh ${TAG}full --vtg-precise full > /dev/null 2>&1
ls gpurun_out/prof_${TAG} | head -3
