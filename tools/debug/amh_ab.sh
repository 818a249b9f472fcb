cd /root/repo
timeout 600 python -m pytest tests/test_fused_gru_wide_gpu.py -x -q 2>&1 | tail -3
python tools/debug/gruw_probe.py 256 81 128
python tools/debug/gruw_probe.py 100 7 128
