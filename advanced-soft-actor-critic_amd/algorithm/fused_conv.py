"""One-launch convolution stack (`asac_conv2_forward` / `asac_conv2_backward`).

The plugin layer `nn_models.layers.ConvLayers` (reference `image_layers.py:178-216`) routes its
`conv_layers` here on the device when they are the two-layer pattern Conv2d GELU Conv2d GELU within the
kernel's limits (the reference's `simple` preset: small frames in one piece, frames up to 84 x 84 and beyond in tiles of
the second-layer map, csrc/conv.hip) — the representation pass of
`SAC_Base.get_l_states` (sac_base.py:1117-1146) over all B x L frames of the sampled windows then costs one
launch for the convolutions instead of a dozen MIOpen / elementwise launches with layout transposes.
Frames are data: the backward produces parameter gradients only (an input that requires grad keeps the
generic path).
"""
import torch
from torch import nn

from asac_amd import native

from .fused_mlp import _flat_alias, direct_enabled

__all__ = ['fused_conv_stack', 'conv_stack_desc', 'DeferredConvBackward']

# add the parameter gradients into existing consecutive `.grad` views from the reduction kernel itself
DIRECT_PARAM_GRADS = True


def conv_stack_desc(conv_layers, x):
    """-> the kernel descriptor if `conv_layers` applied to `x` [N, C, H, W] fits `csrc/conv.hip`, else None"""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and not x.requires_grad):
        return None
    mods = list(conv_layers) if isinstance(conv_layers, nn.Sequential) else None
    if mods is None or len(mods) != 4:
        return None
    c1, g1, c2, g2 = mods
    if not (type(c1) is nn.Conv2d and type(c2) is nn.Conv2d and type(g1) is nn.GELU and type(g2) is nn.GELU):
        return None
    if g1.approximate != 'none' or g2.approximate != 'none':
        return None
    for c in (c1, c2):
        if (c.kernel_size[0] != c.kernel_size[1] or c.stride[0] != c.stride[1] or c.padding not in ((0, 0), 0)
                or c.dilation != (1, 1) or c.groups != 1 or c.bias is None or c.padding_mode != 'zeros'):
            return None
    if c1.in_channels != x.shape[1] or c2.in_channels != c1.out_channels:
        return None
    desc = native.conv2_desc(x.shape[1], x.shape[2], x.shape[3], c1.out_channels, c1.kernel_size[0], c1.stride[0],
                             c2.out_channels, c2.kernel_size[0], c2.stride[0])
    return desc if native.conv2_supported(desc) else None


class DeferredConvBackward:
    """Several backward walks of ONE forward pass with the convolution stack's share of them as ONE launch.

    The reference differentiates the representation's graph once per gated auxiliary loss (`calculate_adaptive_weights`,
    sac_base.py:1607-1631); the convolution stack is the LEAF of each of these walks (frames are data), so its backward can
    wait: inside `with DeferredConvBackward() as d:` a `_ConvStackFn.backward` only records its output gradient under the
    current `d.walk` index and hands autograd no parameter gradients; `d.flush()` then runs `asac_conv2_backward_multi` once
    per forward pass that was reached — frames, saved pre-activations and the gathered patch operands shared by the
    cotangents — and returns, per walk, {parameter id: gradient}.  Bit-identical to the walks' own launches."""
    _active = None

    def __init__(self):
        self.walk = 0
        self._pending = {}       # id(ctx) -> (ctx, {walk: grad_y})

    def __enter__(self):
        assert DeferredConvBackward._active is None
        DeferredConvBackward._active = self
        return self

    def __exit__(self, *exc):
        DeferredConvBackward._active = None

    def record(self, ctx, grad_y) -> bool:
        entry = self._pending.setdefault(id(ctx), (ctx, {}))
        if self.walk in entry[1]:          # (the same node twice in one walk: not a case this form handles)
            return False
        entry[1][self.walk] = grad_y.contiguous()
        return True

    def flush(self) -> dict:
        """-> {walk: {id(param): gradient tensor}} (gradients of a parameter reached through several forward passes summed)"""
        out = {}
        for ctx, by_walk in self._pending.values():
            desc = ctx.desc
            x, z1, z2, w2 = ctx.saved_tensors
            n_frames = x.shape[0] * x.shape[1] if ctx.windows else x.shape[0]
            walks = sorted(by_walk)
            from .fused_mlp import DeferredPartialSums
            later = DeferredPartialSums.active()      # (the launch's own second launch — the slab sums — may wait as well)
            most = native.conv2_backward_multi_max(desc) if later is not None else native.CONV2_MAX_COTANGENTS
            for lo in range(0, len(walks), most):
                part = walks[lo:lo + most]
                n = native.conv2_param_count(desc)
                g = torch.empty(len(part), n, dtype=x.dtype, device=x.device)
                slabs = native.conv2_backward_slabs(desc, n_frames, len(part))
                ws = torch.empty(len(part) * native.conv2_backward_workspace(desc, n_frames), dtype=x.dtype, device=x.device)
                native.conv2_backward_multi(desc, x, w2.detach().contiguous(), z1, z2, [by_walk[k] for k in part], g, ws,
                                            native.SUM_DEFER if later is not None else False)
                if later is not None:
                    later.add(ws, slabs, 16, len(part) * n, len(part) * n, g.view(-1))
                for row, k in enumerate(part):
                    off, grads = 0, out.setdefault(k, {})
                    for p_ in ctx.params:
                        cnt = p_.numel()
                        if p_.requires_grad:
                            piece = g[row, off:off + cnt].view(p_.shape)
                            grads[id(p_)] = piece if id(p_) not in grads else grads[id(p_)] + piece
                        off += cnt
        self._pending = {}
        return out


class _ConvStackFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, desc, w1, b1, w2, b2, windows=None, grad_mode=True):
        N = x.shape[0]
        # (`needs_input_grad` mirrors `requires_grad` whatever the caller's grad mode, and inside `forward` the mode is
        # always off: the caller's mode is handed in, so that a no-grad pass over trainable parameters saves nothing)
        train = grad_mode and any(ctx.needs_input_grad[2:6])
        if windows is not None:
            # `windows` [B, T, C, H, W]: the frames as a slice of the sampled windows, read in place forward and backward
            # (x is a stand-in consulted for its shape only; the slice lives in the step's static batch, which outlasts
            # the backward)
            y = torch.empty(N, _out_width(desc), dtype=windows.dtype, device=windows.device)
            wd = [t.detach().contiguous() for t in (w1, b1, w2, b2)]
            z1 = z2 = None
            if train:
                z1 = torch.empty(native.conv2_z1_floats(desc, N), dtype=windows.dtype, device=windows.device)
                z2 = torch.empty_like(y)
            native.conv2_forward_windows(desc, windows, *wd, y, z1, z2)
            if train:
                ctx.desc, ctx.windows = desc, True
                ctx.save_for_backward(windows, z1, z2, w2)
                ctx.params = (w1, b1, w2, b2)
            return y
        ctx.windows = False
        x = x.contiguous()
        h1 = (desc.height - desc.kernel1) // desc.stride1 + 1
        w1o = (desc.width - desc.kernel1) // desc.stride1 + 1
        h2, w2o = (h1 - desc.kernel2) // desc.stride2 + 1, (w1o - desc.kernel2) // desc.stride2 + 1
        out = desc.out2 * h2 * w2o
        y = torch.empty(N, out, dtype=x.dtype, device=x.device)
        z1 = torch.empty(native.conv2_z1_floats(desc, N), dtype=x.dtype, device=x.device) if train else None
        z2 = torch.empty(N, out, dtype=x.dtype, device=x.device) if train else None
        wd = [t.detach().contiguous() for t in (w1, b1, w2, b2)]
        native.conv2_forward(desc, x, *wd, y, z1, z2)
        if train:
            ctx.desc = desc
            ctx.save_for_backward(x, z1, z2, w2)
            ctx.params = (w1, b1, w2, b2)
        return y

    @staticmethod
    def backward(ctx, grad_y):
        desc = ctx.desc
        deferred = DeferredConvBackward._active
        if deferred is not None and deferred.record(ctx, grad_y):
            return (None, None, None, None, None, None, None, None)
        x, z1, z2, w2 = ctx.saved_tensors
        n_frames = x.shape[0] * x.shape[1] if ctx.windows else x.shape[0]
        run = native.conv2_backward_windows if ctx.windows else native.conv2_backward
        ws = torch.empty(native.conv2_backward_workspace(desc, n_frames), dtype=x.dtype, device=x.device)
        params = ctx.params
        # the four gradients as one packed block: inside the learner they are consecutive views of the flat
        # gradient buffer, and the reduction kernel adds into them directly (no AccumulateGrad launches)
        flat = None
        if DIRECT_PARAM_GRADS and direct_enabled() and all(p.requires_grad and p.grad is not None for p in params):
            flat = _flat_alias([p.grad for p in params])
        from .fused_mlp import DeferredPartialSums
        later = DeferredPartialSums.active()          # (the slab sums as one launch with the walk's other second launches)
        n = native.conv2_param_count(desc)
        if flat is not None:
            run(desc, x, w2.detach().contiguous(), z1, z2, grad_y.contiguous(), flat, ws,
                native.SUM_DEFER if later is not None else True)
            if later is not None:
                later.add(ws, native.conv2_backward_slabs(desc, n_frames), 16, n, n, flat, accumulate=True)
            return (None, None, None, None, None, None, None, None)
        g = torch.empty(n, dtype=x.dtype, device=x.device)
        run(desc, x, w2.detach().contiguous(), z1, z2, grad_y.contiguous(), g, ws,
            native.SUM_DEFER if later is not None else False)
        grads, off = [], 0
        for p in params:
            k = p.numel()
            grads.append(g[off:off + k].view(p.shape) if p.requires_grad else None)
            off += k
        if later is not None:           # (the gradients reach the caller through `later.flush()`)
            later.add(ws, native.conv2_backward_slabs(desc, n_frames), 16, n, n, g)
            later.record(params, grads)
            grads = [None] * 4
        return (None, None, *grads, None, None)


def _out_width(desc):
    h1 = (desc.height - desc.kernel1) // desc.stride1 + 1
    w1o = (desc.width - desc.kernel1) // desc.stride1 + 1
    return desc.out2 * ((h1 - desc.kernel2) // desc.stride2 + 1) * ((w1o - desc.kernel2) // desc.stride2 + 1)


def window_slice(x5, desc):
    """x5 [B, T, C, H, W] -> x5 itself when it is a slice of sampled windows the forward can read in place (dense
    frames, a sample's T frames consecutive, T whole workgroup groups), else None"""
    if (x5.dim() == 5 and not x5.is_contiguous() and x5[0].is_contiguous() and x5.stride(0) % 4 == 0
            and x5.shape[1] % native.conv2_group_frames(desc) == 0 and x5.data_ptr() % 16 == 0):
        return x5
    return None


def fused_conv_stack(x, desc, conv_layers, windows=None):
    """x [N, C, H, W] -> [N, out2*H2*W2]: what `conv_layers(x).reshape(N, -1)` returns.  `windows`: see `window_slice`
    (then x may be a non-materialised stand-in of the right shape)"""
    c1, _, c2, _ = list(conv_layers)
    return _ConvStackFn.apply(x, desc, c1.weight, c1.bias, c2.weight, c2.bias, windows, torch.is_grad_enabled())
