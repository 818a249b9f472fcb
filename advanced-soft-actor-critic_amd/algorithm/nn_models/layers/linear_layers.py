"""Dense building blocks of the model-plugin surface.

`LinearLayers` / `ResBlock` keep the constructor signature, attribute names (hence `state_dict`
keys: `dense.<2i>.linear.{weight,bias}`, final `dense.<2d>.{weight,bias}`) and initialisation
(Kaiming-uniform weights, zero bias, GELU) of reference
`algorithm/nn_models/layers/linear_layers.py:24-119`, so user model files and checkpoints are
interchangeable.  The dense contractions run on rocBLAS/hipBLASLt (MFMA); the step's non-GEMM
work is what the HIP kernels of this package fuse.
"""
import torch
from torch import nn

__all__ = ['ResBlock', 'LinearLayers']


def _init_linear(linear: nn.Linear) -> nn.Linear:
    nn.init.kaiming_uniform_(linear.weight.data)
    linear.bias.data.zero_()
    return linear


class ResBlock(nn.Module):
    """act(Linear(x)) (+ x when the widths agree and `residual`)."""

    def __init__(self, input_size, output_size=None, activation=None, residual=True):
        super().__init__()
        output_size = input_size if output_size is None else output_size
        self.residual = bool(residual) and input_size == output_size
        self.linear = _init_linear(nn.Linear(input_size, output_size))
        self.act = (nn.GELU if activation is None else activation)()

    def forward(self, x):
        assert x.shape[-1] == self.linear.in_features
        from algorithm.fused_rows_linear import rows_linear, rows_resblock      # lazy: avoids an import cycle
        y = rows_resblock(self, x)      # thousands of rows on the device: Linear + GELU (+ x) as one launch per pass
        if y is not None:
            return y
        y = self.act(rows_linear(self.linear, x))
        return y + x if self.residual else y


class LinearLayers(nn.Module):
    """`dense_depth` ResBlocks of width `dense_n` (int or explicit list), optional output Linear.

    `output_size` attribute reports the width of what `forward` returns (the input width when the
    stack is empty), which model files use to chain blocks.
    """

    def __init__(self, input_size, dense_n=64, dense_depth=0, output_size=None,
                 activation=None, residual=True, dropout=0.):
        super().__init__()
        self.input_size = input_size
        widths = list(dense_n) if isinstance(dense_n, (list, tuple)) else [dense_n] * dense_depth

        blocks, width = [], input_size
        for w in widths:
            blocks += [ResBlock(width, w, activation=activation, residual=residual), nn.Dropout(dropout)]
            width = w
        if output_size:
            blocks.append(_init_linear(nn.Linear(width, output_size)))
            width = output_size

        self.output_size = width
        self.dense = nn.Sequential(*blocks)
        self.fuse = False      # opt-in (set by the convolution encoders for their heads): see `forward`

    def forward(self, x):
        assert x.shape[-1] == self.input_size
        if self.fuse and x.is_cuda:
            from algorithm.fused_mlp import fused_dense, fused_dense_wide_first     # lazy: avoids an import cycle
            out = fused_dense(self, x)    # one launch per pass (csrc/mlp.hip) when the stack and its buffers fit
            if out is None and self.input_size > 128:
                out = fused_dense_wide_first(self, x)    # a wide first layer on its own launches (csrc/wide.hip), the rest fused
            if out is not None:
                return out
        if x.is_cuda and torch.is_grad_enabled():
            # the stack module by module: a plain Linear over thousands of rows takes its parameter gradients from one
            # launch (`fused_rows_linear`); everything else is called as the Sequential would
            from algorithm.fused_rows_linear import rows_linear
            for mod in self.dense:
                x = rows_linear(mod, x) if type(mod) is nn.Linear else mod(x)
            return x
        return self.dense(x)
