"""Image / 1-D convolution stacks of the model-plugin surface (API-compatible with reference
`algorithm/nn_models/layers/image_layers.py:12-256,360-377`: same constructor arguments, attribute
names (`conv_layers`, `dense`, `conv_transpose`) and pre-defined stacks `small` / `simple` / `nature`,
so `state_dict`s interchange).  Two-layer Conv2d-GELU stacks on small frames (the `simple` preset) run as
one fused launch per pass (`algorithm/fused_conv.py`, `csrc/conv.hip`); other stacks run on MIOpen.  The replay
side feeds them straight from HBM (uint8 frames are widened to float32 / 255 inside the gather kernel).
`VisionTransformer` needs torchvision's encoder, which this image does not ship: importing the name
works, constructing it raises with that explanation.
"""
import math
from typing import Callable

import torch
from torch import nn

from .linear_layers import LinearLayers

__all__ = ['conv1d_output_size', 'conv2d_output_shape', 'pool_out_shape', 'convtranspose_output_shape',
           'default_conv1d', 'small_visual', 'simple_visual', 'nature_visual',
           'Conv1dLayers', 'ConvLayers', 'ConvTransposeLayers', 'VisionTransformer', 'Transform']


def conv1d_output_size(l, kernel_size=1, stride=1, padding=0, dilation=1) -> int:
    return math.floor((l + 2 * padding - dilation * (kernel_size - 1) - 1) / stride + 1)


def _pair(v):
    return v if isinstance(v, tuple) else (int(v), int(v))


def conv2d_output_shape(h_w, kernel_size=1, stride=1, padding=0, dilation=1):
    k = _pair(kernel_size)
    return tuple(math.floor((d + 2 * padding - dilation * (kk - 1) - 1) / stride + 1) for d, kk in zip(h_w, k))


def pool_out_shape(h_w, kernel_size, stride):
    return tuple((d - kernel_size) // stride + 1 for d in h_w)


def convtranspose_output_shape(h_w, kernel_size=1, stride=1, padding=0, output_padding=0, dilation=1):
    k = _pair(kernel_size)
    return tuple((d - 1) * stride - 2 * padding + dilation * (kk - 1) + output_padding + 1 for d, kk in zip(h_w, k))


def _conv_stack(channels, spec, act):
    """spec: list of (out_channels, kernel, stride) -> Sequential(conv, act, conv, act, ...)"""
    layers, c = [], channels
    for out_c, k, s in spec:
        layers += [nn.Conv2d(c, out_c, [k, k], [s, s]), act()]
        c = out_c
    return layers, c


def default_conv1d(l, channels):
    l1 = conv1d_output_size(l, 8, 4)
    l2 = conv1d_output_size(l1, 4, 2)
    return nn.Sequential(nn.Conv1d(channels, 16, 8, 4), nn.LeakyReLU(),
                         nn.Conv1d(16, 32, 4, 2), nn.LeakyReLU()), l2, 32


def small_visual(height, width, channels):
    hw = pool_out_shape(conv2d_output_shape((height, width), 3, 1), 2, 2)
    hw = pool_out_shape(conv2d_output_shape(hw, 3, 1), 2, 2)
    return nn.Sequential(nn.Conv2d(channels, 35, [3, 3], [1, 1]), nn.LeakyReLU(), nn.MaxPool2d(2, 2),
                         nn.Conv2d(35, 144, [3, 3], [1, 1]), nn.LeakyReLU(), nn.MaxPool2d(2, 2)), hw, 144


def simple_visual(height, width, channels):
    hw = conv2d_output_shape(conv2d_output_shape((height, width), 8, 4), 4, 2)
    layers, c = _conv_stack(channels, [(16, 8, 4), (32, 4, 2)], nn.GELU)
    return nn.Sequential(*layers), hw, c


def nature_visual(height, width, channels):
    hw = conv2d_output_shape(conv2d_output_shape(conv2d_output_shape((height, width), 8, 4), 4, 2), 3, 1)
    layers, c = _conv_stack(channels, [(32, 8, 4), (64, 4, 2), (64, 3, 1)], nn.LeakyReLU)
    return nn.Sequential(*layers), hw, c


def _flatten_lead(x, keep):
    lead = x.shape[:-keep]
    return lead, x.reshape(-1, *x.shape[-keep:])


class Conv1dLayers(nn.Module):
    """[..., length, channels] -> conv1d stack -> LinearLayers"""

    def __init__(self, in_l, in_channels, conv, out_dense_n=64, out_dense_depth=0, output_size=None):
        super().__init__()
        if isinstance(conv, str):
            if conv != 'default':
                raise RuntimeError(f'No pre-defined {conv} convolutional layer')
            self.conv_layers, l, out_c = default_conv1d(in_l, in_channels)
        elif isinstance(conv, tuple):
            self.conv_layers, l, out_c = conv
        else:
            raise RuntimeError('Argument conv should a tuple[nn.Module, tuple[int, int], int]')
        self.conv_output_size = l * out_c
        self.dense = LinearLayers(self.conv_output_size, out_dense_n, out_dense_depth, output_size)
        self.output_size = self.dense.output_size

    def forward(self, x):
        assert x.dim() >= 3, 'The dimension of input should be greater than or equal to 3'
        lead, x = _flatten_lead(x, 2)
        h = self.conv_layers(x.permute(0, 2, 1))
        return self.dense(h.reshape(*lead, self.conv_output_size))


class ConvLayers(nn.Module):
    """[..., C, H, W] -> conv2d stack ('small' | 'simple' | 'nature' | custom tuple) -> LinearLayers"""

    _PRESETS = {'small': small_visual, 'simple': simple_visual, 'nature': nature_visual}

    def __init__(self, in_height, in_width, in_channels, conv, out_dense_n=64, out_dense_depth=0,
                 output_size=None):
        super().__init__()
        if isinstance(conv, str):
            if conv not in self._PRESETS:
                raise RuntimeError(f'No pre-defined {conv} convolutional layer')
            self.conv_layers, (h, w), out_c = self._PRESETS[conv](in_height, in_width, in_channels)
        elif isinstance(conv, tuple):
            self.conv_layers, (h, w), out_c = conv
        else:
            raise RuntimeError('Argument conv should a tuple[nn.Module, tuple[int, int], int]')
        self.conv_output_size = h * w * out_c
        self.dense = LinearLayers(self.conv_output_size, out_dense_n, out_dense_depth, output_size)
        self.dense.fuse = True       # ResBlock head over every frame of the sampled windows: fused MLP launches
        self.output_size = self.dense.output_size

    def forward(self, x):
        assert x.dim() >= 4, 'The dimension of input should be greater than or equal to 4'
        if x.is_cuda and x.dim() == 5 and not x.is_contiguous():
            # a slice of the sampled windows ([B, T, C, H, W] views such as frames[:, b:]): the fused stack reads it in
            # place instead of from the copy `reshape` would make
            from algorithm.fused_conv import conv_stack_desc, fused_conv_stack, window_slice
            desc = conv_stack_desc(self.conv_layers, x[0])
            if desc is not None and window_slice(x, desc) is not None:
                stand_in = x.new_empty(1).as_strided((x.shape[0] * x.shape[1], *x.shape[2:]), (0, 0, 0, 0))   # (shape only)
                y = fused_conv_stack(stand_in, desc, self.conv_layers, windows=x)
                return self.dense(y.reshape(*x.shape[:2], self.conv_output_size))
        lead, x = _flatten_lead(x, 3)
        if x.is_cuda:
            from algorithm.fused_conv import conv_stack_desc, fused_conv_stack   # lazy: avoids an import cycle
            desc = conv_stack_desc(self.conv_layers, x)
            if desc is not None:      # Conv2d GELU Conv2d GELU on small frames: one launch (csrc/conv.hip)
                return self.dense(fused_conv_stack(x, desc, self.conv_layers).reshape(*lead, self.conv_output_size))
        return self.dense(self.conv_layers(x).reshape(*lead, self.conv_output_size))


class ConvTransposeLayers(nn.Module):
    """[..., input] -> LinearLayers -> [C, H, W] -> transposed-conv stack (observation decoders)"""

    def __init__(self, input_size, in_dense_n, in_dense_depth, height, width, channels, conv_transpose):
        super().__init__()
        self._height, self._width, self._channels = height, width, channels
        self.dense = LinearLayers(input_size, in_dense_n, in_dense_depth, height * width * channels)
        self.conv_transpose = conv_transpose

    def forward(self, x):
        assert x.dim() >= 2, 'The dimension of input should be greater than or equal to 2'
        if x.is_cuda:
            from algorithm.fused_decoder import fused_obs_decoder      # lazy: avoids an import cycle
            vis = fused_obs_decoder(self, x)      # the reference plugins' 2x2x32 -> 30x30x3 decoder on MFMA (csrc/decoder.hip)
            if vis is not None:
                return vis
        x = self.dense(x)
        lead = x.shape[:-1]
        vis = self.conv_transpose(x.reshape(-1, self._channels, self._height, self._width))
        return vis.reshape(*lead, *vis.shape[1:])


class VisionTransformer(nn.Module):
    def __init__(self, *args, **kwargs):
        super().__init__()
        raise ImportError('VisionTransformer wraps torchvision.models.vision_transformer.Encoder; '
                          'torchvision is not available in this environment')


class Transform(nn.Module):
    """Applies an image augmentation callable over [..., C, H, W] (identity when None)."""

    def __init__(self, transform: Callable[[torch.Tensor], torch.Tensor] | None = None):
        super().__init__()
        self.transform = transform

    def forward(self, x):
        if self.transform is None:
            return x
        assert x.dim() >= 4, 'The dimension of input should be greater than or equal to 4'
        lead, x = _flatten_lead(x, 3)
        x = self.transform(x)
        return x.reshape(*lead, *x.shape[1:])
