#!/bin/bash
# What a stage costs the STREAM (batches in flight), as opposed to its duration alone: the pipeline is cut after stage k = 1..9 and the
# step time measured; the increments are the stages' marginal costs. Needs a variant built with -DARUCOHIP_STAGE_EXPERIMENT
# (tools/build_variant.sh stage -DARUCOHIP_STAGE_EXPERIMENT); it is loaded through ARUCOHIP_LIB, bench.py sees the flag in arucohip_build_info()
# and prints an experiment line instead of a headline.   tools/stage_cost.sh build/variants/lib_stage.so [bench args]
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
VAR=$(realpath $1); shift
for k in 1 2 3 4 5 6 7 8 9 9 8 7 6 5 4 3 2 1; do
  ARUCOHIP_LIB=$VAR ARUCOHIP_STOP_AFTER=$k python bench.py --no-latency --no-cpu-baseline --no-legs --steps 30 --warmup 6 "$@" 2>/dev/null | tail -1
done
