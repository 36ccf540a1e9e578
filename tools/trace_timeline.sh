#!/bin/bash
# Kernel timeline of the pipelined bench (optionally truncated: needs the stage-experiment variant, tools/build_variant.sh):
# tools/trace_timeline.sh <tag> <variant.so|-> <stop_after|-> [bench args]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=$1; VAR=$2; STOP=$3; shift 3
OUT=gpurun_out/timeline_$TAG
mkdir -p $OUT
if [ "$VAR" != "-" ]; then export ARUCOHIP_LIB=$(realpath "$VAR"); fi   # loaded through ARUCOHIP_LIB, the product library stays
if [ "$STOP" != "-" ]; then export ARUCOHIP_STOP_AFTER=$STOP; fi
rm -rf /tmp/p_tl
timeout -k 10 400 rocprofv3 --kernel-trace --kernel-include-regex "ah::" --output-format csv -d /tmp/p_tl -- python3 bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-latency --no-legs "$@" > $OUT/run.log 2>&1
cp /tmp/p_tl/*/*kernel_trace.csv /tmp/kt.csv
python3 tools/timeline.py /tmp/kt.csv 7 > $OUT/timeline.txt
python3 tools/overlap_summarize.py /tmp/kt.csv > $OUT/summary.txt
head -5 $OUT/summary.txt
