"""k_attn_mh forward / backward launch time at the cfg_attn_h64 shape (B 1024, window 9, 8 heads of 8), back-to-back launches
under HIP events.   usage: python tools/debug/amh_probe.py [B L H d]"""
import sys, importlib
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
native = importlib.import_module('advanced-soft-actor-critic_amd.native')

B, L, H, d = (int(x) for x in sys.argv[1:5]) if len(sys.argv) >= 5 else (1024, 9, 8, 8)
E = H * d
dev = 'cuda'
g = torch.Generator(device=dev).manual_seed(0)
q, k, v, go = (torch.randn(B, L, E, device=dev, generator=g) for _ in range(4))
mask = torch.triu(torch.ones(L, L, device=dev, dtype=torch.bool), 1)[None].expand(B, L, L).contiguous()
out, gq, gk, gv = (torch.empty(B, L, E, device=dev) for _ in range(4))
w, gw = torch.empty(B, L, L, device=dev), torch.randn(B, L, L, device=dev, generator=g)
keep = torch.empty(B, L, device=dev)
ph = torch.empty(B, H, L, L, device=dev)
R = 50


def timed(fn):
    fn(); torch.cuda.synchronize()
    native.load().asac_set_launch_repeat(R)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / R)
    native.load().asac_set_launch_repeat(1)
    return best


f = timed(lambda: native.attention_mh_forward(q, k, v, mask, H, out, w, keep, ph))
b = timed(lambda: native.attention_mh_backward(q, k, v, mask, H, ph, go, gw, gq, gk, gv))
print(f'B {B} L {L} H {H} d {d}: forward {f:.2f} us  backward {b:.2f} us')
