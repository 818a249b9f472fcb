cd /root/repo
for c in cfg_attn_h64 cfg3_h64 cfg4 cfg3; do
printf "$c "
timeout 600 python bench.py --config $c --no-extras --no-cpu-baseline --profile-steps 0 --steps 1500 --warmup 100 --run-length 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
ASAC_PARITY_RECORD=1 timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -8
