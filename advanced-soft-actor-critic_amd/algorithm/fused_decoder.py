"""The observation decoder of the recurrent prediction models on MFMA (`asac_obs_decoder_forward / _backward`).

`nn_models.layers.ConvTransposeLayers` (reference `image_layers.py:231-253`) routes here on the device when it is the
decoder the reference's image plugins build (`envs/roller/nn_visual_hard_attn.py:64-96`, `envs/roller/nn_visual_hard.py:
47-57`, `envs/pyramid/nn_visual.py:50-60`): a one-block dense head of width 64 into [32, 2, 2], then
ConvTranspose2d(32, 32, 4, 2) / (32, 16, 8, 4) / (16, 3, 3, 1), each followed by LeakyReLU — `_train_rpm`'s observation
loss (`sac_base.py:1798-1839`) then costs three launches forward and four backward instead of MIOpen's transposed
convolutions, their layout transposes and one elementwise launch per bias / activation.  Plain autograd semantics: the
Function returns the frames, its backward takes their gradient and returns the gradients of the state and of the ten
parameters (or adds the latter straight into consecutive `.grad` views inside `fused_mlp.direct_param_grads()`).
"""
import torch
from torch import nn

from asac_amd import native

from .fused_mlp import direct_enabled, direct_skips

__all__ = ['decoder_params', 'fused_obs_decoder']

ENABLED = True


def _plain_ct(m, cin, cout, k, s) -> bool:
    return (type(m) is nn.ConvTranspose2d and m.in_channels == cin and m.out_channels == cout
            and m.kernel_size == (k, k) and m.stride == (s, s) and m.padding == (0, 0) and m.output_padding == (0, 0)
            and m.dilation == (1, 1) and m.groups == 1 and m.bias is not None and m.padding_mode == 'zeros')


def decoder_params(ctl):
    """-> the ten parameters of `ctl` (a `ConvTransposeLayers`) in the kernels' order when it is the supported decoder,
    else None"""
    from .nn_models.layers.linear_layers import LinearLayers, ResBlock
    if (ctl._height, ctl._width, ctl._channels) != (2, 2, 32):
        return None
    dense = ctl.dense
    if not isinstance(dense, LinearLayers) or dense.input_size > 16:
        return None
    mods = [m for m in dense.dense if not (isinstance(m, nn.Dropout) and m.p == 0.)]
    if len(mods) != 2 or not isinstance(mods[0], ResBlock) or type(mods[1]) is not nn.Linear:
        return None
    block, final = mods
    if (block.residual or type(block.act) is not nn.GELU or block.act.approximate != 'none'
            or block.linear.out_features != 64 or block.linear.bias is None or final.bias is None
            or final.in_features != 64 or final.out_features != 128):
        return None
    ct = ctl.conv_transpose
    mods = list(ct) if isinstance(ct, nn.Sequential) else None
    if mods is None or len(mods) != 6:
        return None
    c1, a1, c2, a2, c3, a3 = mods
    if not (_plain_ct(c1, 32, 32, 4, 2) and _plain_ct(c2, 32, 16, 8, 4) and _plain_ct(c3, 16, 3, 3, 1)):
        return None
    for act in (a1, a2, a3):
        if type(act) is not nn.LeakyReLU or act.negative_slope != 0.01:
            return None
    return [block.linear.weight, block.linear.bias, final.weight, final.bias, c1.weight, c1.bias, c2.weight, c2.bias,
            c3.weight, c3.bias]


class _ObsDecoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, state, *params):
        N = state.shape[0]
        dev = state.device
        state = state.contiguous()
        frames = torch.empty(N, 3, 30, 30, dtype=torch.float32, device=dev)
        packed = torch.empty(native.obs_decoder_packed_floats(), dtype=torch.float32, device=dev)
        saved = torch.empty(native.obs_decoder_saved_floats(N), dtype=torch.float32, device=dev)
        native.obs_decoder_forward(state, [p.detach().contiguous() for p in params], packed, saved, frames)
        ctx.save_for_backward(state, packed, saved, frames)
        ctx.params = params
        return frames

    @staticmethod
    def backward(ctx, grad_frames):
        state, packed, saved, frames = ctx.saved_tensors
        params = ctx.params
        N, S = state.shape
        dev = state.device
        ws = torch.empty(native.obs_decoder_workspace_floats(N), dtype=torch.float32, device=dev)
        gx = torch.empty(N, S, dtype=torch.float32, device=dev) if ctx.needs_input_grad[0] else None
        grad_frames = grad_frames.contiguous()
        if direct_enabled() and any(ctx.needs_input_grad[1:]) and not direct_skips(*params) \
                and all(p.requires_grad and p.grad is not None for p in params):
            # inside the learner the ten gradients are views of the flat gradient buffer: the reduction adds into them
            grads = [p.grad for p in params]
            if all(g.is_contiguous() and g.dtype == torch.float32 for g in grads):
                native.obs_decoder_backward(state, packed, saved, frames, grad_frames, gx, grads, ws, accumulate=True)
                return (gx, *([None] * 10))
        grads = [torch.empty_like(p, memory_format=torch.contiguous_format) for p in params]
        native.obs_decoder_backward(state, packed, saved, frames, grad_frames, gx, grads, ws)
        return (gx, *[g if p.requires_grad else None for g, p in zip(grads, params)])


def fused_obs_decoder(ctl, x):
    """`ctl(x)` for a supported `ConvTransposeLayers` (see `decoder_params`): x [..., S] -> [..., 3, 30, 30]; None when the
    module or the input does not fit (the caller keeps the module path)"""
    if not (ENABLED and x.is_cuda and x.dtype == torch.float32 and x.dim() >= 2 and x.shape[-1] == ctl.dense.input_size):
        return None
    params = getattr(ctl, '_asac_decoder_params', False)
    if params is False:
        params = decoder_params(ctl)
        ctl.__dict__['_asac_decoder_params'] = params      # (not a registered attribute: parameters stay where they are)
    if params is None or any(p.dtype != torch.float32 or not p.is_cuda for p in params):
        return None
    lead = x.shape[:-1]
    frames = _ObsDecoderFn.apply(x.reshape(-1, x.shape[-1]), *params)
    return frames.reshape(*lead, 3, 30, 30)
