#!/bin/bash
# generation schedules with five batches in flight (bench.py default since the end of round 3): the bench's stream and the cluttered one, alternating
cd "$GRAFT_REPO_ROOT"
run2() {
  env "$@" python bench.py --no-latency --no-cpu-baseline --no-legs --steps 30 --warmup 6 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print('config2 $*', ':', d['value'], 'fps')"
}
runc() {
  env "$@" python bench.py --clutter --batch 256 --frames 256 --no-latency --no-cpu-baseline --no-legs --steps 30 --warmup 6 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print('clutter $*', ':', d['value'], 'fps')"
}
A="ARUCOHIP_GENS=64,64,64,128,128,256,256,1024 ARUCOHIP_FORK_AFTER=7"
B="ARUCOHIP_GENS=64,64,128,128,256,512,1024 ARUCOHIP_FORK_AFTER=6"
for r in 1 2; do run2 X=0; run2 $A; run2 $B; done
runc X=0; runc $A; runc $B; runc X=0
