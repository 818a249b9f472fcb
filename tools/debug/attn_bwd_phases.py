"""Phase clocks of `k_attn_proj_bwd` (workgroup 0): tools/build_variant.sh stamps attn.hip -DASAC_ATTN_STAMPS, then
ASAC_HIP_LIB=advanced-soft-actor-critic_amd/lib/libasac_hip_stamps.so python tools/debug/attn_bwd_phases.py [B L E]"""
import ctypes
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import asac_amd  # noqa: E402,F401
from asac_amd import native  # noqa: E402
import algorithm.nn_models as m  # noqa: E402

B, L, E = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (1024, 9, 8)))
lib = native.load()
torch.manual_seed(0)
attn = m.MultiheadAttention(E, 1, out_dense_depth=1).cuda()
key = torch.randn(B, L, E, device='cuda', requires_grad=True)
out, w = attn(key[:, -L:], key, key)
go = torch.randn_like(out)
params = list(attn.parameters())
for _ in range(3):
    torch.autograd.grad(out, [key] + params, grad_outputs=go, retain_graph=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    torch.autograd.grad(out, [key] + params, grad_outputs=go, retain_graph=True)
e1.record()
torch.cuda.synchronize()
print(f'backward walk (kernel + partial sums + glue): {e0.elapsed_time(e1) * 50:.1f} us')
if hasattr(lib, 'asac_debug_attn_stamps'):
    st = (ctypes.c_ulonglong * 8)()
    lib.asac_debug_attn_stamps(st)
    names = ['staging (weights, w, g_w, rows)', 'projections, output block back', 'phase 1 (queries)', 'phase 2 (keys)',
             'phase 3 (parameter partials)']
    for k, n in enumerate(names):
        print(f'   {n:36s} {st[k + 1] - st[k]:8d} clocks')
    print(f'   {"total":36s} {st[5] - st[0]:8d} clocks')
