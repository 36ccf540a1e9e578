set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
V=$PWD/build/variants
ARUCOHIP_LIB=$V/lib_dual1024.so python -m pytest tests/test_gpu_matrix.py -x -q -k "two_borders or cluttered or parameter_matrix" > gpurun_out/r04_t16.log 2>&1 || { tail -30 gpurun_out/r04_t16.log; exit 1; }
tail -2 gpurun_out/r04_t16.log
tools/sweep.sh -r 2 -s 20 -w 5 "X=0" "ARUCOHIP_LIB=$V/lib_dual1024.so" > gpurun_out/r04_ab_dual1024.txt 2>&1
cut -c1-330 gpurun_out/r04_ab_dual1024.txt
tools/sweep.sh -r 2 -s 20 -w 5 -a "--clutter" "X=0" "ARUCOHIP_LIB=$V/lib_dual1024.so" > gpurun_out/r04_ab_dual1024_clutter.txt 2>&1
cut -c1-330 gpurun_out/r04_ab_dual1024_clutter.txt
