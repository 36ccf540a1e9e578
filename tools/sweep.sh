#!/bin/bash
# One sweep script (round 4; replaces the nine sweep_*.sh of rounds 2-3, whose results are kept in profiles/r0*_kernel_experiments.txt).
# Runs bench.py once per SETTING, the whole list ROUNDS times in alternation (so that drift of the box shows), prints one line per run.
#   tools/sweep.sh [-r rounds] [-s steps] [-w warmup] [-a "bench args for every run"] SETTING...
# A SETTING is a list of environment assignments, optionally followed by " -- " and bench.py arguments of its own:
#   tools/sweep.sh "X=0" "ARUCOHIP_QUAD_BLOCKS=16"                                     (what sweep_env.sh / sweep_gens.sh did)
#   tools/sweep.sh "GPU_MAX_HW_QUEUES=8 -- --depth 3" "GPU_MAX_HW_QUEUES=12 -- --depth 5"   (sweep_depth*.sh)
#   tools/sweep.sh -s 80 "X=0 -- --batch 512 --depth 6" "X=0 -- --batch 256 --depth 8"      (sweep_batch.sh)
#   tools/sweep.sh -a "--clutter --batch 256 --frames 256" "X=0" "ARUCOHIP_GENS=64,64,128,256,512,1024 ARUCOHIP_FORK_AFTER=5"
#   tools/sweep.sh "X=0" "ARUCOHIP_LIB=$PWD/build/variants/lib_foo.so"                  (A/B of a variant build, tools/build_variant.sh)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
ROUNDS=2; STEPS=20; WARM=5; ALL=""
while getopts "r:s:w:a:" o; do case $o in r) ROUNDS=$OPTARG;; s) STEPS=$OPTARG;; w) WARM=$OPTARG;; a) ALL=$OPTARG;; esac; done
shift $((OPTIND - 1))
for r in $(seq 1 $ROUNDS); do
  for setting in "$@"; do
    envs=${setting%% -- *}; args=""; [[ "$setting" == *" -- "* ]] && args=${setting#* -- }
    env $envs python bench.py --no-latency --no-cpu-baseline --no-legs --steps $STEPS --warmup $WARM $ALL $args 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
print('round $r | $setting |', d['value'], 'fps', d['ms_per_step'], 'ms/step', {k: round(v, 3) for k, v in d.get('kernel_ms_isolated', {}).items() if v > 0.05})" || exit 1
  done
done
