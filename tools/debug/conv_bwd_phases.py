"""Phase clocks of `k_conv2_bwd<., NC>` (workgroup 0, summed over its groups): build the stamped variant first
    tools/build_variant.sh stamps conv.hip -DASAC_CONV_STAMPS
then   ASAC_HIP_LIB=advanced-soft-actor-critic_amd/lib/libasac_hip_stamps.so python tools/debug/conv_bwd_phases.py"""
import ctypes
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import asac_amd  # noqa: E402,F401
from asac_amd import native  # noqa: E402

NAMES = ['dma wait', 'z1 -> a1, gelu\', dz2', 'dW2 (+ db2)', 'dp = dz2 W2', 'col2im gather -> dz1', 'barrier', 'dW1', 'product tile -> LDS, barrier', '-', 'loop top']
lib = native.load()
N, C, H, W = (int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (9216, 3, 30, 30)))
desc = native.conv2_desc(C, H, W, 16, 8, 4, 32, 4, 2)
torch.manual_seed(0)
w = [torch.randn(16, 3, 8, 8, device='cuda') * 0.1, torch.zeros(16, device='cuda'), torch.randn(32, 16, 4, 4, device='cuda') * 0.1,
     torch.zeros(32, device='cuda')]
x = torch.randn(N, C, H, W, device='cuda')
h1_, w1_ = (H - 8) // 4 + 1, (W - 8) // 4 + 1
y = torch.empty(N, 32 * ((h1_ - 4) // 2 + 1) * ((w1_ - 4) // 2 + 1), device='cuda')
z1 = torch.empty(native.conv2_z1_floats(desc, N), device='cuda')
z2 = torch.empty_like(y)
native.conv2_forward(desc, x, *w, y, z1, z2)
n = native.conv2_param_count(desc)
for nc in range(1, native.conv2_backward_multi_max(desc) + 1):
    gys = [torch.randn_like(y) for _ in range(nc)]
    out = torch.empty(nc, n, device='cuda')
    ws = torch.empty(nc * native.conv2_backward_workspace(desc, N), device='cuda')
    for _ in range(3):
        native.conv2_backward_multi(desc, x, w[2], z1, z2, gys, out, ws)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        native.conv2_backward_multi(desc, x, w[2], z1, z2, gys, out, ws)
    e1.record()
    torch.cuda.synchronize()
    st = (ctypes.c_ulonglong * 16)()
    if hasattr(lib, "asac_debug_conv_stamps"):
        lib.asac_debug_conv_stamps(st)       # (a library built without -DASAC_CONV_STAMPS: the timing only)
    tot = sum(st[:10]) or 1
    print(f'NC = {nc}: {e0.elapsed_time(e1) * 100:.1f} us per launch pair; workgroup 0: {tot} clocks over its groups')
    for k, name in enumerate(NAMES):
        if st[k]:
            print(f'   {name:28s} {st[k]:9d} clocks  {100.0 * st[k] / tot:5.1f} %')
