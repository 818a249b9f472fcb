cd /root/repo
timeout 600 python -m pytest tests/test_fused_attn_mh_gpu.py tests/test_fused_gru_wide_gpu.py -x -q 2>&1 | tail -5
for i in 1 2; do for v in 0 1; do
printf 'XTY_QUEUE=%s ' $v
ASAC_XTY_QUEUE=$v timeout 600 python bench.py --config cfg_attn_h64 --no-extras --no-cpu-baseline --profile-steps 0 --steps 1500 --warmup 100 --run-length 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done; done
python bench.py --config cfg3_h64 --no-extras --no-cpu-baseline --profile-steps 0 --steps 1500 --warmup 100 --run-length 0 2>&1 | tail -1 | cut -c1-120
