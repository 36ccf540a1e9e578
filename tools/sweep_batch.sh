#!/bin/bash
# (arucohip_set_pipeline_depth takes at most 8 batches in flight)
# bench.py over batch size x batches in flight (cache locality of the sparse stages against launch count)
cd "$GRAFT_REPO_ROOT"
for cfg in "1024 3 40 8" "512 3 80 8" "512 6 80 16" "256 4 160 8" "256 8 160 16" "128 6 320 16" "128 8 320 24"; do
  set -- $cfg
  GPU_MAX_HW_QUEUES=$4 python bench.py --no-latency --no-cpu-baseline --batch $1 --depth $2 --steps $3 --warmup 8 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print('batch $1 depth $2 queues $4 :', d['value'], 'fps', d['ms_per_step'], 'ms/step')" || exit 1
done
