cd /root/repo
timeout 900 python -m pytest tests/test_fused_attn_mh_gpu.py -x -q 2>&1 | grep -E "passed|failed"
for i in 1 2; do for lib in libasac_hip_old.so libasac_hip.so; do
printf '%-22s ' $lib
ASAC_HIP_LIB=/root/repo/advanced-soft-actor-critic_amd/lib/$lib timeout 600 python bench.py --config cfg_attn_h64 --no-extras --no-cpu-baseline --profile-steps 0 --steps 1500 --warmup 100 --run-length 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done; done
