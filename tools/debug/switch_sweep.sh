#!/bin/bash
# which hip_config switch makes a full-size case pass / fail?  usage: switch_sweep.sh <case> <switch> ...
c=$1; shift
for sw in "$@"; do
  r=$(ASAC_TEST_HIP_CONFIG="{\"$sw\": false}" python -m pytest tests/test_full_size_gpu.py -q -s -k $c 2>&1 | grep -E "step 2: td|passed|failed" | tr '\n' ' ')
  echo "$sw=false: $r"
done
