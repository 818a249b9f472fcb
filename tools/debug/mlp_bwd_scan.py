"""launch time of the fused dense stack (forward, backward) over N rows: fixed part vs per-row part"""
import sys
import torch
sys.path.insert(0, '.')
import asac_amd  # noqa
from asac_amd import native
import algorithm.nn_models as m
from algorithm.fused_mlp import fused_dense

torch.manual_seed(0)
for in0 in (128, 64):
    ll = m.LinearLayers(in0, dense_n=64, dense_depth=2, output_size=8).cuda()
    ll.fuse = True
    from algorithm.fused import FlatParamGroup
    group = FlatParamGroup([('m', list(ll.parameters()))], 'cuda')
    for N in (1152, 4608, 9216, 18432, 36864):
        x = torch.randn(N, in0, device='cuda', requires_grad=True)
        g = torch.randn(N, 8, device='cuda')
        for _ in range(3):
            out = fused_dense(ll, x)
            out.backward(g)
        with native.LaunchProfiler(repeat=20) as prof:
            out = fused_dense(ll, x)
            out.backward(g)
        s = prof.summary()
        print(in0, N, {k: round(v['avg_us'], 1) for k, v in s.items()})
