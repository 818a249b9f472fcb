#!/bin/bash
# same-box A/B of this tree against another checkout of the repository (a git worktree inside the repo root, e.g. `_r04`):
#   usage: gpurun -- 'bash tools/ab_rounds.sh _r04 "cfg2 cfg4 cfg5" 2'
R=${GRAFT_REPO_ROOT:-$(pwd)}
OTHER=$1; CFGS=${2:-cfg2}; ROUNDS=${3:-2}
for c in $CFGS; do
  st=3000; [ $c = cfg2 ] && st=20000; [ $c = cfg5 ] && st=600
  for i in $(seq $ROUNDS); do
    for tree in $R $R/$OTHER; do
      printf '%-6s %-28s ' $c $(basename $tree)
      (cd $tree && timeout 600 python bench.py --config $c --no-extras --no-cpu-baseline --profile-steps 0 --steps $st --warmup 200 \
        --run-length 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    done
  done
done
