#!/bin/bash
# A/B of library variants on one box: tools/ab_variants.sh <steps> <variant.so> ... (each variant is copied over aruco_amd/libarucohip.so in the
# box's copy of the repo; "base" = the library as shipped). Alternates the variants twice so that drift of the box shows.
cd "$GRAFT_REPO_ROOT"
STEPS=$1; shift
cp aruco_amd/libarucohip.so /tmp/ab_base.so
for round in 1 2; do
  for v in base "$@"; do
    if [ "$v" = base ]; then cp /tmp/ab_base.so aruco_amd/libarucohip.so; else cp "$v" aruco_amd/libarucohip.so; fi
    python bench.py --no-latency --no-cpu-baseline --no-legs --steps $STEPS --warmup 6 ${AB_ARGS:-} 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print('$v round $round :', d['value'], 'fps', d['ms_per_step'], 'ms/step', {k: round(v, 3) for k, v in d['kernel_ms_isolated'].items() if v > 0.05})" || exit 1
  done
done
cp /tmp/ab_base.so aruco_amd/libarucohip.so
