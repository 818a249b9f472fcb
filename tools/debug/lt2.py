import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import asac_amd  # noqa
from algorithm.fused_linear import fuse_linear_tanh_heads
from algorithm.fused_mlp import direct_param_grads
from torch import nn
torch.manual_seed(0)
N = 96
m = nn.Sequential(nn.Linear(18, 8), nn.Tanh()).cuda()
ref = nn.Sequential(nn.Linear(18, 8), nn.Tanh()).cuda()
ref.load_state_dict(m.state_dict())
holder = nn.ModuleList([m]); fuse_linear_tanh_heads(holder)
flat = torch.zeros(18 * 8 + 8, device='cuda')
m[0].weight.grad = flat[:144].view(8, 18); m[0].bias.grad = flat[144:]
x = torch.randn(N, 18, device='cuda', requires_grad=True)
x2 = x.detach().clone().requires_grad_(True)
y = m(x); y2 = ref(x2)
g2 = torch.autograd.grad(y2.square().mean(), [x2, *ref.parameters()], retain_graph=True)
with direct_param_grads():
    y.square().mean().backward(retain_graph=True)
print('direct', float((flat[:144].view(8, 18) - g2[1]).abs().max() / g2[1].abs().max()))
for rep in range(3):
    g = torch.autograd.grad(y.square().mean() * (1.0 if rep else -1.0), [x, *m.parameters()], retain_graph=True)
    print(rep, [float((a - b).abs().max() / b.abs().max()) for a, b in zip(g, g2)], float(g[1].abs().max()), float(g2[1].abs().max()))
