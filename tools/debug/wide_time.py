"""Launch times of the wide Linear's kernels (HIP events over back-to-back launches): python tools/debug/wide_time.py [R K N]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import asac_amd  # noqa: E402,F401
from asac_amd import native  # noqa: E402

R, K, N = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (2304, 2592, 64)))
torch.manual_seed(0)
x = torch.randn(R, K, device='cuda')
w = torch.randn(N, K, device='cuda') * 0.02
b = torch.randn(N, device='cuda')
y, pre, g = torch.empty(R, N, device='cuda'), torch.empty(R, N, device='cuda'), torch.randn(R, N, device='cuda')
dpre, dx = torch.empty(R, N, device='cuda'), torch.empty(R, K, device='cuda')
dw, db = torch.empty(N, K, device='cuda'), torch.empty(N, device='cuda')
ws = native.rows_wide_workspace(x, N)


def timed(name, fn, flops, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / reps
    print(f'{name:28s} {us:7.1f} us   {flops / us / 1e6:6.1f} TFLOP/s')


f = 2.0 * R * K * N
timed('forward (+ finish)', lambda: native.rows_wide_forward(x, w, b, y, pre, True, ws), f)
timed('backward input', lambda: native.rows_wide_backward_input(g, pre, w, dpre, dx), f)
timed('backward params (+ reduce)', lambda: native.rows_wide_backward_params(dpre, x, dw, db, False, ws), f)
ref = torch.nn.functional.gelu(x @ w.t() + b)
print('max |y - ref| =', float((y - ref).abs().max()))
