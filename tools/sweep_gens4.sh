#!/bin/bash
cd "$GRAFT_REPO_ROOT"
run4() {
  env "$@" python bench.py --config 4 --no-latency --no-cpu-baseline --no-legs --steps 40 --warmup 6 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print('config4 $*', ':', d['value'], 'fps')"
}
run3() {
  env "$@" python bench.py --config 3 --no-latency --no-cpu-baseline --no-legs --steps 30 --warmup 6 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print('config3 $*', ':', d['value'], 'fps')"
}
runl() {
  env "$@" python bench.py --no-cpu-baseline --no-legs --steps 30 --warmup 6 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print('config2+latency $*', ':', d['value'], 'fps', {k: v.get('walkers_ms') for k, v in d['latency'].items() if isinstance(v, dict) and 'walkers_ms' in v})"
}
N="ARUCOHIP_GENS=64,64,64,128,128,256,256,1024 ARUCOHIP_FORK_AFTER=7"
run4 X=0; run4 $N; run4 X=0; run4 $N
run3 X=0; run3 $N
runl X=0; runl $N
