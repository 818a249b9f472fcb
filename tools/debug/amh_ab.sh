#!/bin/bash
# Scratch A/B driver used through the round (run as `gpurun -- 'bash tools/debug/amh_ab.sh'`): alternates two builds of the library
# (ASAC_HIP_LIB) on one box — box-to-box variance is 0.5-1 %, same-box runs agree to 0.2 %.   usage: amh_ab.sh [cfg] [old.so]
cd ${GRAFT_REPO_ROOT:-/root/repo}
CFG=${1:-cfg_attn_h64}; OLD=${2:-advanced-soft-actor-critic_amd/lib/libasac_hip_old.so}
for i in 1 2; do for lib in $OLD advanced-soft-actor-critic_amd/lib/libasac_hip.so; do
  printf '%-60s ' $lib
  ASAC_HIP_LIB=$PWD/$lib timeout 600 python bench.py --config $CFG --no-extras --no-cpu-baseline --profile-steps 0 --steps 1500 --warmup 100 \
    --run-length 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done; done
