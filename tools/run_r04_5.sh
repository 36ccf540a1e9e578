set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python -m pytest tests/test_gpu_matrix.py tests/test_gpu_fullsize.py tests/test_gpu_parity.py -x -q > gpurun_out/r04_t5.log 2>&1 || { tail -40 gpurun_out/r04_t5.log; exit 1; }
tail -3 gpurun_out/r04_t5.log
tools/sweep.sh -r 2 -s 20 -w 5 "X=0" "ARUCOHIP_QUAD_DUAL=0" > gpurun_out/r04_ab_dual.txt 2>&1
cat gpurun_out/r04_ab_dual.txt
tools/sweep.sh -r 2 -s 20 -w 5 -a "--clutter" "X=0" "ARUCOHIP_QUAD_DUAL=0" > gpurun_out/r04_ab_dual_clutter.txt 2>&1
cat gpurun_out/r04_ab_dual_clutter.txt
for c in single 1080p; do
  for m in segments walkers; do
    echo "== $c $m"; ARUCOHIP_CONTOURS=$m bash tools/trace_latency.sh $c > gpurun_out/r04_lat_${c}_$m.txt 2>&1 || true; tail -45 gpurun_out/r04_lat_${c}_$m.txt
  done
done
python - <<'PY'
import numpy as np, subprocess, sys
sys.path.insert(0, '.')
from aruco_amd import synth
fr, _ = synth.make_stream(1, seed=4711, device="cpu")
open('/tmp/f.raw','wb').write(fr[0].numpy().tobytes())
for q in ("8", "16", "32"):
    import os
    env = dict(os.environ, GPU_MAX_HW_QUEUES=q)
    r = subprocess.run(["build/thread_bench", "/tmp/f.raw", "1920", "1080", "1.5", "1", "2", "4", "8", "16"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    print("queues", q, r.stdout.strip())
PY
