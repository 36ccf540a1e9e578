#!/bin/bash
cd "$GRAFT_REPO_ROOT"
run() {
  env "$@" python bench.py --no-latency --no-cpu-baseline --no-legs --steps 40 --warmup 6 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print('$*', ':', d['value'], 'fps', {k: round(v, 3) for k, v in d['kernel_ms_isolated'].items() if k in ('walker_kernel', 'walker_long_kernel', 'contour_quad_kernel')})"
}
run X=0
run ARUCOHIP_GENS=64,64,64,128,128,256,256,1024 ARUCOHIP_FORK_AFTER=7
run ARUCOHIP_GENS=48,48,64,64,64,128,128,256,256,1024 ARUCOHIP_FORK_AFTER=9
run ARUCOHIP_GENS=32,32,32,64,64,64,128,128,256,256,1024 ARUCOHIP_FORK_AFTER=10
run ARUCOHIP_GENS=64,64,64,64,128,128,128,256,1024 ARUCOHIP_FORK_AFTER=8
run ARUCOHIP_GENS=64,64,64,128,128,256,256,1024 ARUCOHIP_FORK_AFTER=7 ARUCOHIP_LEASH=96
run ARUCOHIP_GENS=64,64,64,128,128,256,256,1024 ARUCOHIP_FORK_AFTER=6
run ARUCOHIP_GENS=64,64,64,128,128,256,256,1024 ARUCOHIP_FORK_AFTER=7
run X=0
