#!/bin/bash
# In-situ kernel durations of one configuration's replayed train step: rocprofv3 --kernel-trace --stats of
# `bench.py --config <cfg>` (no extras), summarised into gpurun_out/insitu/<cfg>_{kernel_stats.json,kernel_stats_summary.txt,
# step_sequence.txt}.   usage: tools/insitu.sh cfg4 [cfg5 ...]   (env STEPS / WARMUP override 200 / 20)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/insitu
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in "$@"; do
  rm -rf /tmp/prof_$c
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$c -- python $R/bench.py --no-extras --config $c \
      --no-cpu-baseline --profile-steps 0 --steps ${STEPS:-200} --warmup ${WARMUP:-20} --run-length 0 > $O/${c}_bench.log 2>&1
  python $R/tools/step_sequence.py $(find /tmp/prof_$c -name "*kernel_trace.csv" | head -1) > $O/${c}_step_sequence.txt
  python $R/tools/summarize_rocprof.py $(find /tmp/prof_$c -name "*kernel_stats.csv" | head -1) $((${STEPS:-200} + ${WARMUP:-20})) 40 \
      $O/${c}_kernel_stats.json > $O/${c}_kernel_stats_summary.txt
  tail -1 $O/${c}_bench.log | cut -c1-400
  grep -E "prologue|gather|vtrace|td_update|sumtree_update" $O/${c}_kernel_stats_summary.txt
done
