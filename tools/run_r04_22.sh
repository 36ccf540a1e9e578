set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
V=$PWD/build/variants
tools/sweep.sh -r 2 -s 20 -w 5 "X=0" "ARUCOHIP_LIB=$V/lib_ntload.so" > gpurun_out/r04_ab_ntload.txt 2>&1
cut -c1-330 gpurun_out/r04_ab_ntload.txt
tools/sweep.sh -r 1 -s 20 -w 5 -a "--clutter" "X=0" "ARUCOHIP_LIB=$V/lib_ntload.so" > gpurun_out/r04_ab_ntload_clutter.txt 2>&1
cut -c1-330 gpurun_out/r04_ab_ntload_clutter.txt
