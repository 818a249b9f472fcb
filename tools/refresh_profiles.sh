#!/bin/bash
# Regenerates the round's measurement artifacts on a GPU box (run through gpurun from the repo root):
#   bench JSONs (cfg2 default = the BENCH line, cfg3 / cfg4 informational), rocprofv3 kernel-trace stats of
#   the same cfg2 command, the two PMC passes (FETCH_SIZE / WRITE_SIZE) and the kernel sweep.
# Outputs land in gpurun_out/refresh/ ; copy what should be judged into profiles/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/refresh
mkdir -p $O
cd $R
timeout 600 python tools/kernel_sweep.py > $O/kernel_sweep.txt 2> $O/kernel_sweep.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof /tmp/pmc_r /tmp/pmc_w /tmp/spmc_r /tmp/spmc_w
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $R/bench.py --no-extras --no-cpu-baseline --profile-steps 0 --run-length 0 > /tmp/prof.log 2>&1
f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1)
cp $f $O/cfg2_kernel_stats.csv
python $R/tools/summarize_rocprof.py $f 2100 45 $O/cfg2_kernel_stats.json > $O/cfg2_kernel_stats_summary.txt
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_r -- python $R/bench.py --no-extras --no-cpu-baseline --steps 100 --warmup 10 --fill 20000 --profile-steps 0 --run-length 0 > /tmp/r.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w -- python $R/bench.py --no-extras --no-cpu-baseline --steps 100 --warmup 10 --fill 20000 --profile-steps 0 --run-length 0 > /tmp/w.log 2>&1
python $R/tools/summarize_pmc.py $(find /tmp/pmc_r -name "*counter_collection.csv" | head -1) $(find /tmp/pmc_w -name "*counter_collection.csv" | head -1) $O/cfg2_pmc_traffic.json > $O/cfg2_pmc_traffic_all.txt
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/spmc_r -- python $R/tools/kernel_sweep.py > /tmp/sr.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/spmc_w -- python $R/tools/kernel_sweep.py > /tmp/sw.log 2>&1
python $R/tools/summarize_pmc.py $(find /tmp/spmc_r -name "*counter_collection.csv" | head -1) $(find /tmp/spmc_w -name "*counter_collection.csv" | head -1) $O/kernel_sweep_pmc.json --by-grid > $O/kernel_sweep_pmc_all.txt
f=$(find /tmp/prof -name "*kernel_trace.csv" | head -1)
python $R/tools/step_sequence.py $f > $O/cfg2_step_sequence.txt
for c in cfg3 cfg3_h64 cfg4 cfg4_84 cfg5 cfg5_without_prediction cfg_attn_h64; do
  rm -rf /tmp/prof_$c
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$c -- python $R/bench.py --no-extras --config $c --no-cpu-baseline --profile-steps 0 --steps 200 --warmup 20 --run-length 0 > /tmp/prof_$c.log 2>&1
  python $R/tools/step_sequence.py $(find /tmp/prof_$c -name "*kernel_trace.csv" | head -1) > $O/${c}_step_sequence.txt
  python $R/tools/summarize_rocprof.py $(find /tmp/prof_$c -name "*kernel_stats.csv" | head -1) 220 40 $O/${c}_kernel_stats.json > $O/${c}_kernel_stats_summary.txt
done
# HBM traffic of the representation kernels (cfg3 GRU, cfg4 / cfg5 convolution stack and attention): two PMC passes each
for c in ${PMC_CFGS:-cfg3 cfg3_h64 cfg4 cfg4_84 cfg5 cfg_attn_h64}; do
  rm -rf /tmp/pmc_r_$c /tmp/pmc_w_$c
  st="--steps 60 --warmup 10"; if [ $c = cfg5 ]; then st="--steps 12 --warmup 4"; fi; if [ $c = cfg4_84 ]; then st="--steps 30 --warmup 6"; fi      # (cfg5: 200 launches a step)
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_r_$c -- python $R/bench.py --no-extras --config $c --no-cpu-baseline $st --fill $( [ $c = cfg4_84 ] && echo 8000 || echo 20000 ) --profile-steps 0 --run-length 0 > /tmp/r_$c.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w_$c -- python $R/bench.py --no-extras --config $c --no-cpu-baseline $st --fill $( [ $c = cfg4_84 ] && echo 8000 || echo 20000 ) --profile-steps 0 --run-length 0 > /tmp/w_$c.log 2>&1
  python $R/tools/summarize_pmc.py $(find /tmp/pmc_r_$c -name "*counter_collection.csv" | head -1) $(find /tmp/pmc_w_$c -name "*counter_collection.csv" | head -1) $O/${c}_pmc_traffic.json > $O/${c}_pmc_traffic_all.txt
done
# the bench lines last: their `*_in_situ` fields read the kernel-trace summaries just made (profiles/r06_*_kernel_stats.json)
# the headline workload with one batch in flight (hip_config['lookahead'] = 1): its two alternating graphs
rm -rf /tmp/prof_la
ASAC_BENCH_HIP_CONFIG='{"lookahead": 1}' timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_la -- python $R/bench.py --no-extras --no-cpu-baseline --profile-steps 0 --steps 400 --warmup 40 --run-length 0 > /tmp/prof_la.log 2>&1
python $R/tools/step_sequence.py $(find /tmp/prof_la -name "*kernel_trace.csv" | head -1) > $O/cfg2_lookahead_step_sequence.txt
python $R/tools/summarize_rocprof.py $(find /tmp/prof_la -name "*kernel_stats.csv" | head -1) 440 40 $O/cfg2_lookahead_kernel_stats.json > $O/cfg2_lookahead_kernel_stats_summary.txt
for c in cfg2 cfg3 cfg3_h64 cfg4 cfg4_84 cfg5 cfg5_without_prediction cfg_attn_h64; do cp $O/${c}_kernel_stats.json $R/profiles/r06_${c}_kernel_stats.json; [ -f $O/${c}_pmc_traffic.json ] && cp $O/${c}_pmc_traffic.json $R/profiles/r06_${c}_pmc_traffic.json; done
cd $R
timeout 1200 python bench.py > $O/cfg2_bench_line.json 2> $O/cfg2_bench.err
cp $R/bench_details.json $O/cfg2_bench.json      # (the full record; cfg2_bench_line.json: the < 4 KB line the driver parses)
timeout 900 python bench.py --no-extras --config cfg3 --cpu-budget 10 --emit full > $O/cfg3_bench.json 2> $O/cfg3_bench.err
timeout 900 python bench.py --no-extras --config cfg4 --cpu-budget 10 --emit full > $O/cfg4_bench.json 2> $O/cfg4_bench.err
timeout 900 python bench.py --no-extras --config cfg4_84 --cpu-budget 10 --emit full > $O/cfg4_84_bench.json 2> $O/cfg4_84_bench.err
timeout 900 python bench.py --no-extras --config cfg5 --steps 500 --cpu-budget 10 --emit full > $O/cfg5_bench.json 2> $O/cfg5_bench.err
ls -la $O
tail -c 400 $O/cfg2_bench_line.json
