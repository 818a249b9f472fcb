"""Phase clocks of `k_conv2_fwd` (workgroup 0, summed over its groups); stamped variant as tools/debug/conv_bwd_phases.py"""
import ctypes
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import asac_amd  # noqa: E402,F401
from asac_amd import native  # noqa: E402

NAMES = ['dma wait', 'barrier after layer 1', 'barrier after tail sums', 'barrier after layer 2', 'epilogue (stores)', 'next frames requested',
         'tail sums, GELU', 'layer 1 (full tiles + tail partials)', 'layer 2', 'loop top']
lib = native.load()
C, H, W = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (3, 30, 30)))
SIZES = [int(v) for v in sys.argv[4:]] or [2048, 4608, 9216]
desc = native.conv2_desc(C, H, W, 16, 8, 4, 32, 4, 2)
torch.manual_seed(0)
w = [torch.randn(16, 3, 8, 8, device='cuda') * 0.1, torch.zeros(16, device='cuda'), torch.randn(32, 16, 4, 4, device='cuda') * 0.1,
     torch.zeros(32, device='cuda')]
for N, train in [(n, t) for n in SIZES for t in (False, True)]:
    x = torch.randn(N, C, H, W, device='cuda')
    h1_, w1_ = (H - 8) // 4 + 1, (W - 8) // 4 + 1
    y = torch.empty(N, 32 * ((h1_ - 4) // 2 + 1) * ((w1_ - 4) // 2 + 1), device='cuda')
    z1 = torch.empty(native.conv2_z1_floats(desc, N), device='cuda') if train else None
    z2 = torch.empty_like(y) if train else None
    for _ in range(3):
        native.conv2_forward(desc, x, *w, y, z1, z2)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        native.conv2_forward(desc, x, *w, y, z1, z2)
    e1.record()
    torch.cuda.synchronize()
    st = (ctypes.c_ulonglong * 16)()
    if hasattr(lib, "asac_debug_conv_stamps"):
        lib.asac_debug_conv_stamps(st)       # (a library built without -DASAC_CONV_STAMPS: the timing only)
    tot = sum(st[:10]) or 1
    print(f'N = {N} train = {train}: {e0.elapsed_time(e1) * 100:.1f} us per launch; workgroup 0: {tot} clocks in its group loop')
    for k, name in enumerate(NAMES):
        if st[k]:
            print(f'   {name:40s} {st[k]:9d} clocks  {100.0 * st[k] / tot:5.1f} %')
