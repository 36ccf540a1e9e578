set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
PASSES=trace tools/profile.sh r04b
python bench.py --steps 20 --warmup 5 > gpurun_out/r04_b15.log 2> gpurun_out/r04_b15.err || { tail -20 gpurun_out/r04_b15.err; exit 1; }
cut -c1-200 gpurun_out/r04_b15.log
