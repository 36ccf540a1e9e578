#!/bin/bash
# the default bench run (20 steps, as the driver times it) under a list of environment settings, alternating twice: tools/sweep_env.sh "A=1" "A=2 B=3" ...
cd "$GRAFT_REPO_ROOT"
for r in 1 2; do
  for e in "X=0" "$@"; do
    env $e python bench.py --no-latency --no-cpu-baseline --no-legs --steps 20 --warmup 5 ${SWEEP_ARGS:-} 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print('$e :', d['value'], 'fps', d['ms_per_step'], 'ms/step')" || exit 1
  done
done
