#!/bin/bash
# A/B of library variants on one box (tools/build_variant.sh makes them): tools/ab_variants.sh <steps> <variant.so> ...
# "base" = the library as shipped. The variants are loaded through ARUCOHIP_LIB; the product library is never overwritten.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
STEPS=$1; shift
SET=("X=0"); for v in "$@"; do SET+=("ARUCOHIP_LIB=$(realpath $v)"); done
exec tools/sweep.sh -r 2 -s $STEPS -w 6 -a "${AB_ARGS:-}" "${SET[@]}"
