set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_refine.py tests/test_gpu_shim.py tests/test_gpu_rows.py tests/test_gpu_boundary.py tests/test_gpu_robustness.py -x -q > gpurun_out/r04_t10.log 2>&1 || { tail -40 gpurun_out/r04_t10.log; exit 1; }
tail -3 gpurun_out/r04_t10.log
for c in single board chessboard 1080p; do
  echo "== $c"; bash tools/trace_latency.sh $c > gpurun_out/r04_lat6_${c}.txt 2>&1 || true; tail -26 gpurun_out/r04_lat6_${c}.txt | grep -v "rocprofv3\|copyBuffer"
done
for g in 32; do for c in single board 1080p; do echo "== $c grid $g"; ARUCOHIP_GRID=$g bash tools/trace_latency.sh $c 2>&1 | grep -E "segment_k|skip_k|fill_k|cycle|emit|ms per"; done; done
