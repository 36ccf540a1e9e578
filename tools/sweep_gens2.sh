#!/bin/bash
# generation schedule / fork point / first-pass leash against the PIPELINED rate (round 3: the stream is bound by vector-instruction issue, so
# idle lanes inside long generations cost throughput even where they cost no latency)
cd "$GRAFT_REPO_ROOT"
run() {
  env "$@" python bench.py --no-latency --no-cpu-baseline --no-legs --steps 40 --warmup 6 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print('$*', ':', d['value'], 'fps', {k: round(v, 3) for k, v in d['kernel_ms_isolated'].items() if k in ('walker_kernel', 'walker_long_kernel', 'contour_quad_kernel')})"
}
run X=0
run ARUCOHIP_GENS=64,64,128,256,512,1024 ARUCOHIP_FORK_AFTER=5
run ARUCOHIP_GENS=64,128,256,512,1024 ARUCOHIP_FORK_AFTER=4
run ARUCOHIP_GENS=64,64,64,128,128,256,256,1024 ARUCOHIP_FORK_AFTER=7
run ARUCOHIP_GENS=128,128,256,512,1024 ARUCOHIP_FORK_AFTER=4
run ARUCOHIP_LEASH=32
run ARUCOHIP_LEASH=96
run X=0
