#!/bin/bash
# One PMC pass of bench.py (only --pmc: never combined with tracing on this pool). Usage: tools/pmc.sh <tag> "<counters>" [bench args]
set -e
TAG=$1; CNT=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
rm -rf /tmp/p_$TAG
timeout -k 10 500 rocprofv3 --pmc $CNT --kernel-include-regex "ah::" --output-format csv -d /tmp/p_$TAG -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-latency --no-legs --depth 1 "$@" > $OUT/run.log 2>&1
cp /tmp/p_$TAG/*/*counter_collection.csv $OUT/counters.csv
python3 tools/pmc_summarize.py $OUT/counters.csv > $OUT/summary.txt
cat $OUT/summary.txt
