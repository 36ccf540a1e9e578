set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python bench.py --steps 20 --warmup 5 > gpurun_out/r04_b12.log 2> gpurun_out/r04_b12.err || { tail -20 gpurun_out/r04_b12.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_b12.log').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'])
r=d['roofline']; print({k:r[k] for k in ('achieved','frac','traffic','traffic_source','avg_launch_ms','avg_launch_ms_event_pass','isolated_launch_ms','frac_isolated')})
print(json.dumps(d['latency'])[:3000])
print(json.dumps(d['other_configs'])[:1500])
print(d['kernel_ms_isolated'])
PY
