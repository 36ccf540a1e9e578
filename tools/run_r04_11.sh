set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_refine.py tests/test_gpu_shim.py -x -q > gpurun_out/r04_t11.log 2>&1 || { tail -40 gpurun_out/r04_t11.log; exit 1; }
tail -3 gpurun_out/r04_t11.log
for g in 8 16; do for c in single board chessboard 1080p; do echo "== $c grid $g"; ARUCOHIP_GRID=$g bash tools/trace_latency.sh $c 2>&1 | grep -E "candidates_k|segment_k|skip_k|fill_k|cycle|emit|contour_quad|span|ms per"; done; done
