#!/bin/bash
# Run ON THE GPU BOX: the feature-extraction bench with alternative libraries (BLIM_LIB_PATH), two rounds.   usage: tools/ab_vision.sh lib1.so lib2.so ...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
for round in 1 2; do for lib in "$@"; do echo "== $lib"; BLIM_LIB_PATH=$R/$lib python3 tools/vision_bench.py 8 2>/dev/null | tail -1; done; done
