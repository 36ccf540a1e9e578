set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_matrix.py tests/test_gpu_robustness.py -x -q > gpurun_out/r04_t23.log 2>&1 || { tail -40 gpurun_out/r04_t23.log; exit 1; }
tail -2 gpurun_out/r04_t23.log
tools/sweep.sh -r 2 -s 20 -w 5 "X=0" "ARUCOHIP_GEN_XCD=0" > gpurun_out/r04_ab_genxcd.txt 2>&1
cut -c1-330 gpurun_out/r04_ab_genxcd.txt
tools/sweep.sh -r 2 -s 20 -w 5 -a "--clutter" "X=0" "ARUCOHIP_GEN_XCD=0" > gpurun_out/r04_ab_genxcd_clutter.txt 2>&1
cut -c1-330 gpurun_out/r04_ab_genxcd_clutter.txt
