#!/bin/bash
# Kernel trace of the pipelined bench (batches in flight): per kernel the mean duration UNDER overlap, and how many kernels
# run at once on average. Usage: tools/trace_overlap.sh <tag> [bench args]
set -e
TAG=${1:-x}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/overlap_$TAG
mkdir -p $OUT
rm -rf /tmp/p_ov
timeout -k 10 400 rocprofv3 --kernel-trace --kernel-include-regex "ah::" --output-format csv -d /tmp/p_ov -- python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-latency --no-legs "$@" > $OUT/run.log 2>&1
cp /tmp/p_ov/*/*kernel_trace.csv $OUT/kernel_trace.csv
python3 tools/overlap_summarize.py $OUT/kernel_trace.csv | tee $OUT/summary.txt
rm -f $OUT/kernel_trace.csv
