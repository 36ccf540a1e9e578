#!/bin/bash
# Profiles `bench.py` on the GPU box: kernel-trace stats, then FETCH_SIZE and WRITE_SIZE in separate PMC passes
# (MI355X_MICROARCH.md: TCC slots do not fit both). Only this library's kernels are kept. Usage: tools/profile.sh <tag>
set -e
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
rm -rf /tmp/p_trace /tmp/p_fetch /tmp/p_write
# --profile-run: warm-up + timed steps at the bench's DEFAULT depth and nothing else, so that every dispatch in the trace / every counter row is an
# in-stream launch of the configuration the headline is measured on (roofline.frac can be recomputed from the two summaries alone)
ARGS="bench.py --profile-run --steps ${STEPS:-20} --warmup ${WARMUP:-5} ${BENCH_ARGS:-}"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --kernel-include-regex "ah::" --output-format csv -d /tmp/p_trace -- python3 $ARGS > $OUT/trace.log 2>&1
cp /tmp/p_trace/*/*kernel_stats.csv $OUT/kernel_stats.csv
if [ "${PASSES:-all}" = "trace" ]; then grep -h '"profile_run"' $OUT/trace.log | cut -c1-400; exit 0; fi
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "ah::" --output-format csv -d /tmp/p_fetch -- python3 $ARGS > $OUT/fetch.log 2>&1
cp /tmp/p_fetch/*/*counter_collection.csv $OUT/fetch_counters.csv
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "ah::" --output-format csv -d /tmp/p_write -- python3 $ARGS > $OUT/write.log 2>&1
cp /tmp/p_write/*/*counter_collection.csv $OUT/write_counters.csv
grep -h '"profile_run"' $OUT/trace.log $OUT/fetch.log $OUT/write.log | cut -c1-400
ls -la $OUT
