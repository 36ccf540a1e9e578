set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
V=$PWD/build/variants
tools/sweep.sh -r 2 -s 20 -w 5 "X=0" "ARUCOHIP_LIB=$V/lib_ww3.so" "ARUCOHIP_LIB=$V/lib_rl4.so" "ARUCOHIP_LIB=$V/lib_occ.so" "ARUCOHIP_LIB=$V/lib_ww4.so" > gpurun_out/r04_ab_occ.txt 2>&1
cut -c1-330 gpurun_out/r04_ab_occ.txt
tools/sweep.sh -r 1 -s 20 -w 5 -a "--clutter" "X=0" "ARUCOHIP_LIB=$V/lib_occ.so" > gpurun_out/r04_ab_occ_clutter.txt 2>&1
cut -c1-330 gpurun_out/r04_ab_occ_clutter.txt
