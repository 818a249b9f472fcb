"""k_linear_tanh forward / backward launch time at the attention / wide-GRU state head (9 216 rows x 64 -> 8)."""
import sys, importlib
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
native = importlib.import_module('advanced-soft-actor-critic_amd.native')
N, K, O = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (9216, 64, 8)
dev = 'cuda'
x, w, b = torch.randn(N, K, device=dev), torch.randn(O, K, device=dev) * 0.3, torch.randn(O, device=dev)
y, gy, gx = torch.empty(N, O, device=dev), torch.randn(N, O, device=dev), torch.empty(N, K, device=dev)
gp = torch.zeros(O * K + O, device=dev)
ws = torch.zeros(native.linear_tanh_workspace(N, K, O), device=dev)
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


f = timed(lambda: native.linear_tanh_forward(x, w, b, y))
bw = timed(lambda: native.linear_tanh_backward(x, w, y, gy, gx, gp, False, ws))
print(f'N {N} K {K} O {O}: forward {f:.2f} us  backward {bw:.2f} us (main launch x {R}; the reduce launch once)')
