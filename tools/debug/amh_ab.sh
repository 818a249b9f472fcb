cd /root/repo
for c in cfg4 cfg5 cfg4_84; do for v in false true; do
printf "$c rep_branch=$v "
ASAC_BENCH_HIP_CONFIG="{\"rep_branch\": $v}" timeout 600 python bench.py --config $c --no-extras --no-cpu-baseline --profile-steps 0 --steps 800 --warmup 60 --run-length 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done; done
