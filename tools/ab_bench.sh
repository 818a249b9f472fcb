#!/bin/bash
# A/B of two builds of libasac_hip.so on ONE box (run through gpurun from the repo root): the boxes of the pool differ by
# about +-0.5 % in train steps/s, two runs on one box by about +-0.2 %, so a change worth 0.5 % can only be judged by
# alternating the two libraries inside one call.  The library is chosen through ASAC_HIP_LIB (native.py).
#   build B:  cp advanced-soft-actor-critic_amd/lib/libasac_hip.so advanced-soft-actor-critic_amd/lib/libasac_hip_b.so   (before the change)
#             python __graft_entry__.py                                                                                   (after it)
#   usage:    gpurun -- 'bash tools/ab_bench.sh [config] [rounds] [steps]'
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
CFG=${1:-cfg2}
ROUNDS=${2:-3}
STEPS=${3:-20000}
for i in $(seq $ROUNDS); do
  for v in libasac_hip.so libasac_hip_b.so; do
    printf '%-20s ' $v
    ASAC_HIP_LIB=$R/advanced-soft-actor-critic_amd/lib/$v timeout 300 python $R/bench.py --config $CFG --no-extras --no-cpu-baseline \
      --steps $STEPS --warmup 500 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
  done
done
