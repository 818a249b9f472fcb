import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import asac_amd  # noqa
from asac_amd import native
from algorithm.fused_linear import LinearTanhHead, fuse_linear_tanh_heads
from torch import nn
torch.manual_seed(0)
for N in (96, 128, 300, 4608):
    m = nn.Sequential(nn.Linear(18, 8), nn.Tanh()).cuda()
    ref = nn.Sequential(nn.Linear(18, 8), nn.Tanh()).cuda()
    ref.load_state_dict(m.state_dict())
    holder = nn.ModuleList([m]); fuse_linear_tanh_heads(holder)
    x = torch.randn(N, 18, device='cuda', requires_grad=True)
    x2 = x.detach().clone().requires_grad_(True)
    y = m(x); y2 = ref(x2)
    for rep in range(2):
        g = torch.autograd.grad(y.square().mean(), [x, *m.parameters()], retain_graph=True)
        g2 = torch.autograd.grad(y2.square().mean(), [x2, *ref.parameters()], retain_graph=True)
        print(N, rep, [float((a - b).abs().max() / b.abs().max()) for a, b in zip(g, g2)])
