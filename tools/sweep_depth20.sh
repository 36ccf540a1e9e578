#!/bin/bash
# the driver's run (20 timed steps, fill and drain of the pipeline inside them) over batches in flight x hardware queues, alternating twice
cd "$GRAFT_REPO_ROOT"
for round in 1 2; do
for cfg in "3 8" "4 12" "5 12" "6 16"; do
  set -- $cfg
  GPU_MAX_HW_QUEUES=$2 python bench.py --no-latency --no-cpu-baseline --no-legs --depth $1 --steps 20 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print('depth $1 queues $2 :', d['value'], 'fps', d['ms_per_step'], 'ms/step')" || exit 1
done
done
