cd /root/repo
timeout 600 python -m pytest tests/test_fused_gru_wide_gpu.py -x -q 2>&1 | grep -E "passed|failed"
python tools/debug/gruw_probe.py
ASAC_HIP_LIB=/root/repo/advanced-soft-actor-critic_amd/lib/libasac_hip_old.so python tools/debug/gruw_probe.py
python tools/debug/gruw_probe.py 256 81 128
ASAC_HIP_LIB=/root/repo/advanced-soft-actor-critic_amd/lib/libasac_hip_old.so python tools/debug/gruw_probe.py 256 81 128
