#!/bin/bash
# Profiles `bench.py` on the GPU box: kernel-trace stats, then FETCH_SIZE and WRITE_SIZE in separate PMC passes
# (MI355X_MICROARCH.md: TCC slots do not fit both). Only this library's kernels are kept. Usage: tools/profile.sh <tag>
set -e
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
rm -rf /tmp/p_trace /tmp/p_fetch /tmp/p_write
ARGS="bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-latency --no-legs ${BENCH_ARGS:-}"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --kernel-include-regex "ah::" --output-format csv -d /tmp/p_trace -- python3 $ARGS > $OUT/trace.log 2>&1
cp /tmp/p_trace/*/*kernel_stats.csv $OUT/kernel_stats.csv
if [ "${PASSES:-all}" = "trace" ]; then grep -h '"metric"' $OUT/trace.log | cut -c1-400; exit 0; fi
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "ah::" --output-format csv -d /tmp/p_fetch -- python3 $ARGS > $OUT/fetch.log 2>&1
cp /tmp/p_fetch/*/*counter_collection.csv $OUT/fetch_counters.csv
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "ah::" --output-format csv -d /tmp/p_write -- python3 $ARGS > $OUT/write.log 2>&1
cp /tmp/p_write/*/*counter_collection.csv $OUT/write_counters.csv
grep -h '"metric"' $OUT/trace.log | cut -c1-400
ls -la $OUT
