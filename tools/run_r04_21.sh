set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
bash tools/trace_overlap.sh r04c > gpurun_out/r04c_overlap.txt 2>&1 || true
cat gpurun_out/r04c_overlap.txt | tail -30
bash tools/trace_overlap.sh r04c_clutter --clutter > gpurun_out/r04c_overlap_clutter.txt 2>&1 || true
tail -30 gpurun_out/r04c_overlap_clutter.txt
bash tools/stage_cost.sh build/variants/lib_stage.so > gpurun_out/r04c_stage_cost.txt 2>&1 || true
cat gpurun_out/r04c_stage_cost.txt
