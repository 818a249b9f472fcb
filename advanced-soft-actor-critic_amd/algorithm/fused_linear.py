"""The `nn.Sequential(nn.Linear(n, m), nn.Tanh())` state head of a representation plugin as one launch per pass
(csrc/linear.hip) instead of a library GEMM + tanh forward and two GEMMs, a bias reduction, a tanh backward and two
gradient accumulations backward.  The learner re-classes matching sub-modules of the plugin's `ModelRep` in place
(`fuse_linear_tanh_heads`): parameters, `state_dict` keys and results (to f32 rounding) are unchanged, and anything
the kernel does not cover — other devices or dtypes, wider layers — takes the module path."""
import weakref

import torch
from torch import nn

from asac_amd import native

from .fused_mlp import _flat_alias, direct_enabled

__all__ = ['LinearTanhHead', 'fuse_linear_tanh_heads']

_WORKSPACES = weakref.WeakKeyDictionary()
FUSED_LINEAR_TANH = True      # module switch (tests compare against the module path)


class _LinearTanhFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, head):
        rows = x.reshape(-1, x.shape[-1])
        if rows.stride(1) != 1:
            rows = rows.contiguous()
        y = torch.empty(rows.shape[0], weight.shape[0], dtype=x.dtype, device=x.device)
        native.linear_tanh_forward(rows, weight.detach(), bias.detach(), y)
        ctx.save_for_backward(rows, y)
        ctx.weight, ctx.bias, ctx.head, ctx.x_shape = weight, bias, head, x.shape
        return y.view(*x.shape[:-1], weight.shape[0])

    @staticmethod
    def backward(ctx, grad_y):
        rows, y = ctx.saved_tensors
        weight, bias = ctx.weight, ctx.bias
        N, K = rows.shape
        O = weight.shape[0]
        gy = grad_y.reshape(N, O)
        gy = gy if gy.is_contiguous() else gy.contiguous()
        gx = torch.empty(N, K, dtype=rows.dtype, device=rows.device) if ctx.needs_input_grad[0] else None
        ws = ctx.head._workspace(N, K, O, rows.device)
        flat = None
        if (direct_enabled() and weight.requires_grad and bias.requires_grad and weight.grad is not None
                and bias.grad is not None):
            flat = _flat_alias([weight.grad, bias.grad])     # the learner's flat gradient buffer: add in place
        if flat is not None:
            native.linear_tanh_backward(rows, weight.detach(), y, gy, gx, flat, True, ws)
            gw = gb = None
        else:
            g = torch.empty(O * K + O, dtype=rows.dtype, device=rows.device)
            native.linear_tanh_backward(rows, weight.detach(), y, gy, gx, g, False, ws)
            gw, gb = g[:O * K].view(O, K), g[O * K:]
        return (None if gx is None else gx.view(ctx.x_shape)), gw, gb, None


class LinearTanhHead(nn.Sequential):
    """`nn.Sequential(nn.Linear, nn.Tanh)` whose forward is the fused launch when it applies"""

    def _workspace(self, N, K, O, device):
        # zero before first use, left zero by every launch; one per row count (the step's passes differ in rows);
        # kept off the module: not pickled / deep-copied with it
        cache = _WORKSPACES.setdefault(self, {})
        key = (N, str(device))
        if key not in cache:
            cache[key] = torch.zeros(native.linear_tanh_workspace(N, K, O), dtype=torch.float32, device=device)
        return cache[key]

    def forward(self, x):
        lin = self[0]
        if (FUSED_LINEAR_TANH and x.is_cuda and x.dtype == torch.float32 and lin.weight.dtype == torch.float32
                and x.shape[-1] == lin.in_features and x.numel() > 0):
            return _LinearTanhFn.apply(x, lin.weight, lin.bias, self)
        return super().forward(x)


def _is_plain_head(mod) -> bool:
    return (type(mod) is nn.Sequential and len(mod) == 2 and type(mod[0]) is nn.Linear and type(mod[1]) is nn.Tanh
            and mod[0].bias is not None and mod[0].in_features <= native.LINEAR_TANH_MAX_IN
            and mod[0].out_features <= native.LINEAR_TANH_MAX_OUT)


def fuse_linear_tanh_heads(model: nn.Module) -> int:
    """re-classes every plain Linear + Tanh `nn.Sequential` inside `model`; -> how many"""
    n = 0
    for mod in model.modules():
        if _is_plain_head(mod):
            mod.__class__ = LinearTanhHead
            n += 1
    return n
