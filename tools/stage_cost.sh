#!/bin/bash
# What a stage costs the STREAM (batches in flight), as opposed to its duration alone: the pipeline is cut after stage k = 1..9 and the
# step time measured; the increments are the stages' marginal costs. Needs a library built with -DARUCOHIP_STAGE_EXPERIMENT (the shipped
# one ignores ARUCOHIP_STOP_AFTER): tools/stage_cost.sh <variant.so> [bench args]
cd "$GRAFT_REPO_ROOT"
VAR=$1; shift
cp aruco_amd/libarucohip.so /tmp/sc_base.so
cp "$VAR" aruco_amd/libarucohip.so
for k in 1 2 3 4 5 6 7 8 9 9 8 7 6 5 4 3 2 1; do
  ARUCOHIP_STOP_AFTER=$k python bench.py --no-latency --no-cpu-baseline --no-legs --steps 30 --warmup 6 "$@" 2>/dev/null | tail -1
done
cp /tmp/sc_base.so aruco_amd/libarucohip.so
