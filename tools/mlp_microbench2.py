import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import asac_amd  # noqa
from asac_amd import native
import algorithm.nn_models as m
from algorithm.fused import FlatParamGroup
from algorithm.fused_mlp import StockMLP, describe_q
native.load()
def mk(depth):
    class Q(m.ModelQ):
        def _build_model(self):
            super()._build_model(c_dense_depth=depth)
    Q.__name__ = 'ModelQ'
    return Q
import algorithm.fused_mlp as fm
for depth in (1, 2, 3, 4):
    Q = mk(depth)
    q = Q(6, [], 2, False).cuda()
    import algorithm.nn_models.q as cr
    old = cr.ModelQ; cr.ModelQ = Q
    d = describe_q(q); cr.ModelQ = old
    assert d is not None
    g = FlatParamGroup([('m0', list(q.parameters()))], 'cuda')
    mlp = StockMLP(d, g.flat, g.grad, 0, g.segments['m0'][1], 1, torch.device('cuda'))
    N = 256
    x, a = torch.randn(N, 6, device='cuda'), torch.randn(N, 2, device='cuda', requires_grad=True)
    gout = torch.randn(1, N, 1, device='cuda')
    for _ in range(3):
        out = mlp(x, a); (out * gout).sum().backward()
    with native.LaunchProfiler(repeat=50) as prof:
        for _ in range(5):
            out = mlp(x, a); (out * gout).sum().backward()
    s = prof.summary()
    print(f'depth={depth} fwd {s["asac_mlp_forward"]["avg_us"]:7.2f} us   bwd {s["asac_mlp_backward"]["avg_us"]:7.2f} us')
# empty-ish kernels for reference
t = torch.zeros(1024, device='cuda'); s2 = torch.ones(1024, device='cuda')
with native.LaunchProfiler(repeat=50) as prof:
    for _ in range(5):
        native.polyak(t, s2, 0.5)
print('polyak 1024 floats', prof.summary()['asac_polyak']['avg_us'])
