set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_refine.py tests/test_gpu_shim.py -x -q > gpurun_out/r04_t19.log 2>&1 || { tail -40 gpurun_out/r04_t19.log; exit 1; }
tail -2 gpurun_out/r04_t19.log
for c in "single pose" "1080p"; do echo "== $c"; python tools/latency_trace.py $c 2>&1 | grep "ms per call"; done
echo "== 1080p trace"; bash tools/trace_latency.sh 1080p 2>&1 | grep -E "warp_hist|otsu|frame_cand"
tools/sweep.sh -r 1 -s 20 -w 5 "X=0" > gpurun_out/r04_b19.txt 2>&1; cut -c1-330 gpurun_out/r04_b19.txt
