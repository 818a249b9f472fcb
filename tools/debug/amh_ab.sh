cd /root/repo
timeout 900 python -m pytest tests/test_fused_attn_mh_gpu.py -x -q 2>&1 | tail -3
for c in cfg_attn_h64 cfg3_h64 cfg4 cfg5; do printf "$c "; timeout 600 python bench.py --config $c --no-extras --no-cpu-baseline --profile-steps 0 --steps 800 --warmup 60 --run-length 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
ASAC_PARITY_RECORD=1 timeout 2400 python -m pytest tests -q -m gpu -x --durations=25 2>&1 | grep -E "passed|failed|s call|s setup" | head -40
