set -e
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_matrix.py -x -q > gpurun_out/r04_t2.log 2>&1 || { tail -30 gpurun_out/r04_t2.log; exit 1; }
tail -3 gpurun_out/r04_t2.log
for q in 1 2 4; do
  ARUCOHIP_LIB=$PWD/build/variants/lib_walkstats.so ARUCOHIP_PULL_Q=$q python bench.py --no-latency --no-cpu-baseline --no-legs --steps 10 --warmup 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print('walkstats q=$q', d['value'], d.get('list_fill_per_frame'), {k: round(v,3) for k,v in d['kernel_ms_isolated'].items() if 'walker' in k})"
done > gpurun_out/r04_walkstats.txt 2>&1
cat gpurun_out/r04_walkstats.txt
tools/sweep.sh -r 2 -s 20 -w 5 "X=0" "ARUCOHIP_PULL_Q=1" "ARUCOHIP_PULL_Q=4" "ARUCOHIP_PULL_Q=4 ARUCOHIP_GENS=128,768 ARUCOHIP_FORK_AFTER=2" "ARUCOHIP_PULL_Q=3 ARUCOHIP_GENS=896 ARUCOHIP_FORK_AFTER=1" "ARUCOHIP_PULL_Q=3" > gpurun_out/r04_sweep_pull.txt 2>&1
cat gpurun_out/r04_sweep_pull.txt
