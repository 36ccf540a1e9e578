#!/bin/bash
# Builds a VARIANT of the library for an A/B or an experiment without touching the product library: objects and the .so go to
# build/variants/<name>/ (git-ignored, but shipped to the GPU box with the snapshot). Load it with ARUCOHIP_LIB=<path> (aruco_amd/capi.py).
#   tools/build_variant.sh stage -DARUCOHIP_STAGE_EXPERIMENT      -> build/variants/lib_stage.so
#   tools/build_variant.sh chunk4 -DCHUNK_N=4
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
NAME=$1; shift
D="$ROOT/build/variants/$NAME"
mkdir -p "$D"
cp "$ROOT"/aruco_amd/csrc/*.hip "$ROOT"/aruco_amd/csrc/*.h "$ROOT"/aruco_amd/csrc/Makefile "$D"/
mkdir -p "$ROOT/build/include" && cp "$ROOT"/include/arucohip.h "$ROOT/build/include/"
# the copied sources include "../../include/arucohip.h": from build/variants/<name>/ that is build/include/
make -C "$D" -j6 -s OUT="$ROOT/build/variants/lib_$NAME.so" EXTRA="$*"
echo "$ROOT/build/variants/lib_$NAME.so"
