cd /root/repo
ASAC_PARITY_RECORD=1 timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -4
