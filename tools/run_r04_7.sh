set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
V=$PWD/build/variants
for c in single 1080p; do
  echo "== $c stats"; ARUCOHIP_LIB=$V/lib_segstats.so python tools/latency_trace.py $c 2>&1 | tail -2
done
for g in 8 16 32; do
  for c in single 1080p; do
    echo "== $c grid $g"; ARUCOHIP_GRID=$g bash tools/trace_latency.sh $c > gpurun_out/r04_lat3_${c}_g$g.txt 2>&1 || true; grep -E "segment_kernel|cycle_kernel|emit_kernel|span|ms per call" gpurun_out/r04_lat3_${c}_g$g.txt | tail -5
  done
done
