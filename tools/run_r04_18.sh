set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_refine.py tests/test_gpu_shim.py -x -q > gpurun_out/r04_t18.log 2>&1 || { tail -40 gpurun_out/r04_t18.log; exit 1; }
tail -2 gpurun_out/r04_t18.log
for c in "single pose" "board pose" "chessboard pose" "1080p"; do echo "== $c"; python tools/latency_trace.py $c 2>&1 | grep "ms per call"; done
echo "== 1080p trace"; bash tools/trace_latency.sh 1080p 2>&1 | tail -24 | grep -v rocprofv3
echo "== board pose trace"; bash tools/trace_latency.sh board pose 2>&1 | tail -24 | grep -v rocprofv3
