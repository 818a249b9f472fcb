set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/refresh
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in cfg3_h64 cfg_attn_h64; do
  rm -rf /tmp/pmc_r_$c /tmp/pmc_w_$c
  st="--steps 60 --warmup 10"
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_r_$c -- python $R/bench.py --no-extras --config $c --no-cpu-baseline $st --fill 20000 --profile-steps 0 --run-length 0 > /tmp/r_$c.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w_$c -- python $R/bench.py --no-extras --config $c --no-cpu-baseline $st --fill 20000 --profile-steps 0 --run-length 0 > /tmp/w_$c.log 2>&1
  python $R/tools/summarize_pmc.py $(find /tmp/pmc_r_$c -name "*counter_collection.csv" | head -1) $(find /tmp/pmc_w_$c -name "*counter_collection.csv" | head -1) $O/${c}_pmc_traffic.json > $O/${c}_pmc_traffic_all.txt
  tail -3 /tmp/r_$c.log | cut -c1-200
done
ls -la $O | grep pmc
