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


def _as_rows(x):
    rows = x.reshape(-1, x.shape[-1])
    return rows if rows.stride(1) == 1 else rows.contiguous()


# `backward_from_members`: [the _LinearTanhFn node, members [E, B, O], window, position] while that node's backward is
# the next to run
_MEMBERS = None


def is_fused_head_output(t: torch.Tensor) -> bool:
    fn = t.grad_fn
    return fn is not None and type(fn).__name__ == '_LinearTanhFnBackward'


def backward_from_members(out, members, position, placeholder) -> None:
    """Back-propagate d loss / d out[:, position] = sum_e members[e] (zero at the window's other positions) from the
    fused head's own output `out` [B, L, O] (`is_fused_head_output`): the head's backward launch sums the members itself
    (no member-sum launch, no dense [B, L, O] gradient read).  `placeholder`: any tensor of out's shape."""
    global _MEMBERS
    assert is_fused_head_output(out) and out.dim() == 3 and members.shape[1:] == (out.shape[0], out.shape[2])
    _MEMBERS = [out.grad_fn, members.contiguous(), out.shape[1], int(position) % out.shape[1]]
    try:
        torch.autograd.backward([out], [placeholder])
    finally:
        _MEMBERS = None


class _LinearTanhFn(torch.autograd.Function):
    """x0 [..., K0] (| x1 [..., K1]: the two read side by side, `adjacent_cat.DeferredCat`) -> tanh(Linear)"""

    @staticmethod
    def forward(ctx, x0, x1, weight, bias, head, grad_mode=True):
        train = grad_mode and any(ctx.needs_input_grad[:4])
        if (x0.dim() == 3 and x0.stride(2) == 1 and not x0.is_contiguous() and not ctx.needs_input_grad[0]
                and x0.stride(0) != x0.stride(1) * x0.shape[1]):
            # a slice of the sampled windows that does not collapse to uniform rows (vec[:, b:], data): read in place,
            # forward and backward, by (sample, step) addressing
            r1 = None if x1 is None else _as_rows(x1)
            y = torch.empty(x0.shape[0] * x0.shape[1], weight.shape[0], dtype=x0.dtype, device=x0.device)
            native.linear_tanh_forward2(native.WindowRows(x0), r1, weight.detach(), bias.detach(), y)
            if train:
                ctx.window = True
                ctx.save_for_backward(x0, y, *([] if r1 is None else [r1]))
                ctx.weight, ctx.bias, ctx.head = weight, bias, head
                ctx.x0_shape, ctx.x1_shape = x0.shape, None if x1 is None else x1.shape
            return y.view(*x0.shape[:-1], weight.shape[0])
        ctx.window = False
        r0 = _as_rows(x0)
        r1 = None if x1 is None else _as_rows(x1)
        y = torch.empty(r0.shape[0], weight.shape[0], dtype=x0.dtype, device=x0.device)
        native.linear_tanh_forward2(r0, r1, weight.detach(), bias.detach(), y)
        ctx.save_for_backward(r0, y, *([] if r1 is None else [r1]))
        ctx.weight, ctx.bias, ctx.head = weight, bias, head
        ctx.x0_shape, ctx.x1_shape = x0.shape, None if x1 is None else x1.shape
        return y.view(*x0.shape[:-1], weight.shape[0])

    @staticmethod
    def backward(ctx, grad_y):
        r0, y, *rest = ctx.saved_tensors
        r1 = rest[0] if rest else None
        weight, bias = ctx.weight, ctx.bias
        if ctx.window:
            N, K0 = r0.shape[0] * r0.shape[1], r0.shape[2]
            r0 = native.WindowRows(r0)
            dev, dt = y.device, y.dtype
        else:
            N, K0 = r0.shape
            dev, dt = r0.device, r0.dtype
        K = K0 + (0 if r1 is None else r1.shape[1])
        O = weight.shape[0]
        at = _MEMBERS if (_MEMBERS is not None and _MEMBERS[0] is ctx) else None
        assert _MEMBERS is None or at is not None, 'backward_from_members: another node ran first'
        if at is not None:
            gy, members, window, position = at[1], at[1].shape[0], at[2], at[3]
        else:
            gy, members, window, position = grad_y.reshape(N, O), 1, 1, 0
            gy = gy if gy.is_contiguous() else gy.contiguous()
        empty = lambda k: torch.empty(N, k, dtype=dt, device=dev)  # noqa: E731
        gx0 = empty(K0) if ctx.needs_input_grad[0] else None
        gx1 = empty(K - K0) if (r1 is not None and ctx.needs_input_grad[1]) else None
        ws = ctx.head._workspace(N, K, O, dev)
        flat = None
        if (direct_enabled() and weight.requires_grad and bias.requires_grad and weight.grad is not None
                and bias.grad is not None):
            flat = _flat_alias([weight.grad, bias.grad])     # the learner's flat gradient buffer: add in place
        from .fused_mlp import DeferredPartialSums
        later = DeferredPartialSums.active()      # (the workgroups' partials summed with the walk's other second launches)
        if later is not None:
            ws = torch.empty(native.linear_tanh_workspace(N, K, O), dtype=dt, device=dev)      # (kept until the flush)
        if flat is not None:
            native.linear_tanh_backward2(r0, r1, weight.detach(), y, gy, gx0, gx1, flat,
                                         native.SUM_DEFER if later is not None else True, ws, members, window, position)
            gw = gb = None
            if later is not None:
                later.add(ws, (N + 63) // 64, 16, O * K + O, O * K + O, flat, accumulate=True)
        else:
            g = torch.empty(O * K + O, dtype=dt, device=dev)
            native.linear_tanh_backward2(r0, r1, weight.detach(), y, gy, gx0, gx1, g,
                                         native.SUM_DEFER if later is not None else False, ws, members, window, position)
            gw, gb = g[:O * K].view(O, K), g[O * K:]
            if later is not None:       # (the gradients reach the caller through `later.flush()`)
                later.add(ws, (N + 63) // 64, 16, O * K + O, O * K + O, g)
                later.record([weight, bias], [gw, gb])
                gw = gb = None
        return (None if gx0 is None else gx0.view(ctx.x0_shape), None if gx1 is None else gx1.view(ctx.x1_shape),
                gw, gb, None, None)


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
        from .adjacent_cat import DeferredCat
        lin = self[0]
        x1 = None
        if isinstance(x, DeferredCat):
            if FUSED_LINEAR_TANH and len(x.parts) == 2 and x.width == lin.in_features and lin.weight.dtype == torch.float32:
                x, x1 = x.parts                 # the two blocks are read where they are
            else:
                x = x.materialize()
        if (FUSED_LINEAR_TANH and x.is_cuda and x.dtype == torch.float32 and lin.weight.dtype == torch.float32
                and x.shape[-1] + (0 if x1 is None else x1.shape[-1]) == lin.in_features and x.numel() > 0):
            return _LinearTanhFn.apply(x, x1, lin.weight, lin.bias, self, torch.is_grad_enabled())
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
