set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
V=$PWD/build/variants
tools/sweep.sh -r 2 -s 20 -w 5 "X=0" "ARUCOHIP_LIB=$V/lib_pre.so" > gpurun_out/r04_ab_pre.txt 2>&1
cat gpurun_out/r04_ab_pre.txt
tools/sweep.sh -r 2 -s 20 -w 5 -a "--clutter" "X=0" "ARUCOHIP_LIB=$V/lib_pre.so" "ARUCOHIP_LIB=$V/lib_qp2048.so" "ARUCOHIP_LIB=$V/lib_qp4096.so" "ARUCOHIP_PULL_Q=2" "ARUCOHIP_QUAD_BLOCKS=32" "ARUCOHIP_QUAD_BLOCKS=48" > gpurun_out/r04_sweep_clutter.txt 2>&1
cat gpurun_out/r04_sweep_clutter.txt
python bench.py --steps 20 --warmup 5 --no-legs --no-latency --no-cpu-baseline > gpurun_out/r04_b4.log 2>gpurun_out/r04_b4.err
python -c "
import json; d=json.loads(open('gpurun_out/r04_b4.log').read().splitlines()[-1]); r=d['roofline']; print(d['value'], {k:r[k] for k in ('achieved','frac','traffic','avg_launch_ms','avg_launch_ms_event_pass','isolated_launch_ms','frac_traffic','traffic_over_algorithmic')})"
