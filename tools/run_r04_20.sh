set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r04_t27.log 2>&1 || { tail -40 gpurun_out/r04_t20.log; exit 1; }
tail -3 gpurun_out/r04_t27.log
tools/profile.sh r04f
python bench.py --steps 20 --warmup 5 > gpurun_out/r04_b27.log 2> gpurun_out/r04_b27.err || { tail -20 gpurun_out/r04_b27.err; exit 1; }
cut -c1-200 gpurun_out/r04_b27.log
