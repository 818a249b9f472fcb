cd /root/repo
timeout 1500 python -m pytest tests/test_fused_gru_wide_gpu.py tests/test_sac_step_gpu.py tests/test_full_size_gpu.py -x -q -k "gru or rnn or cfg3" 2>&1 | tail -5
for i in 1 2; do
timeout 600 python bench.py --config cfg3_h64 --no-extras --no-cpu-baseline --profile-steps 0 --steps 1500 --warmup 100 --run-length 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
bash tools/insitu.sh cfg3_h64 > /dev/null 2>&1; grep -i "affine\|Cijk" gpurun_out/insitu/cfg3_h64_step_sequence.txt | cut -c1-100
