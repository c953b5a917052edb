#!/bin/bash
# Run ON THE GPU BOX (via gpurun): rocprofv3 kernel trace + PMC passes (one counter per pass, never combined with traces) of
# the benchmark command.   Usage: tools/collect_profiles.sh <tag> [bench args]   -> gpurun_out/prof_<tag>/{stats,fetch,write,mfma,clk}
TAG=${1:-r01}
shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-strong --no-compensated $*"
echo "bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-strong --no-compensated $*" > $OUT/command.txt      # (tools/parse_profiles.py records it in profiles/CURRENT.json)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD > $OUT/bench_stats.json 2> $OUT/stats.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $CMD > /dev/null 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $CMD > /dev/null 2> $OUT/write.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/mfma -- $CMD > /dev/null 2> $OUT/mfma.err
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/clk -- $CMD > /dev/null 2> $OUT/clk.err
# keep the merged-back directory small: the raw per-dispatch traces are not needed
find $OUT -name "*kernel_trace.csv" -delete
ls -R $OUT | head -40
