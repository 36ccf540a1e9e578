#!/bin/bash
# bench.py over batches in flight x hardware queues at the headline batch size (round 3: the threshold pass got shorter, the balance moved)
cd "$GRAFT_REPO_ROOT"
IFS=";" read -ra CFGS <<< "${SWEEP:-1024 2 8;1024 3 8;1024 4 8;1024 4 12;1024 5 12;1024 6 16;512 6 16}"; unset IFS
for cfg in "${CFGS[@]}"; do
  set -- $cfg
  GPU_MAX_HW_QUEUES=$3 python bench.py --no-latency --no-cpu-baseline --no-legs --batch $1 --depth $2 --steps 40 --warmup 8 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print('batch $1 depth $2 queues $3 :', d['value'], 'fps', d['ms_per_step'], 'ms/step')" || exit 1
done
