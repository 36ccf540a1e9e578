set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for mode in "pose" "pose pageable"; do
  echo "== single $mode (no profiler)"; python tools/latency_trace.py single $mode 2>&1 | grep "ms per call"
done
echo "== 1080p pinned / pageable (no profiler)"; python tools/latency_trace.py 1080p 2>&1 | grep "ms per call"; python tools/latency_trace.py 1080p x pageable 2>&1 | grep "ms per call"
echo "== single pose trace"; bash tools/trace_latency.sh single pose > gpurun_out/r04_lat7_single_pose.txt 2>&1 || true; tail -30 gpurun_out/r04_lat7_single_pose.txt | grep -v rocprofv3
echo "== single pose pageable trace"; bash tools/trace_latency.sh single pose pageable > gpurun_out/r04_lat7_single_pose_pg.txt 2>&1 || true; tail -30 gpurun_out/r04_lat7_single_pose_pg.txt | grep -E "COPY|span|fill"
