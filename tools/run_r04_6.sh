set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r04_t6.log 2>&1 || { tail -40 gpurun_out/r04_t6.log; exit 1; }
tail -3 gpurun_out/r04_t6.log
for c in single 1080p; do
  echo "== $c segments"; ARUCOHIP_CONTOURS=segments bash tools/trace_latency.sh $c > gpurun_out/r04_lat2_${c}.txt 2>&1 || true; tail -24 gpurun_out/r04_lat2_${c}.txt
done
tools/sweep.sh -r 1 -s 20 -w 5 "X=0" "ARUCOHIP_QUAD_BLOCKS=12" "ARUCOHIP_QUAD_BLOCKS=16" "ARUCOHIP_QUAD_DUAL=0" > gpurun_out/r04_ab_dual2.txt 2>&1
cat gpurun_out/r04_ab_dual2.txt
tools/sweep.sh -r 1 -s 20 -w 5 -a "--clutter" "X=0" "ARUCOHIP_QUAD_BLOCKS=12" "ARUCOHIP_QUAD_BLOCKS=16" "ARUCOHIP_QUAD_DUAL=0" > gpurun_out/r04_ab_dual2_clutter.txt 2>&1
cat gpurun_out/r04_ab_dual2_clutter.txt
