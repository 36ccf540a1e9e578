set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
V=$PWD/build/variants
python -m pytest tests/test_gpu_parity.py tests/test_gpu_refine.py tests/test_gpu_matrix.py tests/test_gpu_shim.py tests/test_gpu_rows.py -x -q > gpurun_out/r04_t8.log 2>&1 || { tail -40 gpurun_out/r04_t8.log; exit 1; }
tail -3 gpurun_out/r04_t8.log
for c in single 1080p; do
  echo "== $c stats"; ARUCOHIP_LIB=$V/lib_segstats.so python tools/latency_trace.py $c 2>&1 | tail -2
done
for c in single board 1080p; do
  echo "== $c"; bash tools/trace_latency.sh $c > gpurun_out/r04_lat4_${c}.txt 2>&1 || true; tail -24 gpurun_out/r04_lat4_${c}.txt | grep -v rocprofv3
done
echo "== single, laps through memory"; ARUCOHIP_CYCLE_LDS=0 bash tools/trace_latency.sh single 2>&1 | grep -E "cycle|span|ms per"
