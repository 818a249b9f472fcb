"""CPU, build container only: the reference loads checkpoint + replay files written by the product
(`tests/golden/product_ckpt/`, see `tests/golden/ref_load_product.py`).  Skipped where `/root/reference` does not
exist (the GPU box)."""
import subprocess
import sys
from pathlib import Path

import pytest

HERE = Path(__file__).resolve().parent


@pytest.mark.skipif(not Path('/root/reference/algorithm/sac_base.py').exists(), reason='needs the reference checkout')
@pytest.mark.skipif(not (HERE / 'golden' / 'product_ckpt' / '3.pth').exists(), reason='product files not generated yet')
def test_reference_restores_product_written_files():
    r = subprocess.run([sys.executable, str(HERE / 'golden' / 'ref_load_product.py')], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert 'reference restored the product files: ok' in r.stdout
