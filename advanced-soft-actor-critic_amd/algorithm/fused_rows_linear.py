"""`nn.Linear` over thousands of rows with its parameter gradients from ONE MFMA launch.

The building blocks of the plugin surface (`LinearLayers`, `ResBlock`, the attention projections) apply `nn.Linear(K <= 128,
O <= 128)` to every row of the sampled windows: 9 216 ... 20 736 rows a pass.  Forward and input gradient are library GEMMs
that suit their shape; the WEIGHT gradient grad_out^T x is a [O x K] product over the rows for which the library picks a
32 x 6-style tile (measured 46-76 us per layer at 9 216 rows), and the bias gradient is an ATen split reduction plus the
memset node that zeroes its semaphores.  `rows_linear` keeps the module (parameters, `state_dict`, the forward GEMM) and
replaces the backward: `asac_xty` forms grad_weight and grad_bias in one launch (fixed summation order) — reference
nn_models/layers/linear_layers.py:24-119 under autograd.
"""
import os

import torch

ENABLED = os.environ.get('ASAC_ROWS_LINEAR', '1') != '0'      # (0: plain nn.Linear — A/B runs)
MIN_ROWS = 2048
WIDE = os.environ.get('ASAC_ROWS_WIDE', '1') != '0'          # (0: the library's GEMMs for wide first layers — A/B runs)
QUEUE = os.environ.get('ASAC_XTY_QUEUE', '1') != '0'      # (0: one launch pair per product — A/B runs)


# Direct mode (the learner's `loss.backward()`: gradients are ADDED into the `.grad` views of the flat buffer, nobody reads them
# before the optimizer): the products are not launched one by one but queued and issued four at a time as one launch pair
# (`asac_xty_multi` — the four Linears of an attention block make exactly one), the rest when the backward pass ends.
_pending = []
_armed = False


def flush_param_grads():
    from asac_amd import native
    jobs = list(_pending)
    del _pending[:]
    if len(jobs) == 1:
        native.xty(*jobs[0], accumulate=True)
    elif jobs:
        native.xty_multi(jobs, accumulate=True)


def reset_queue():
    """drop what a failed backward pass left behind (called when the learner opens a direct-mode backward)"""
    global _armed
    del _pending[:]
    _armed = False


def _end_of_backward():
    global _armed
    _armed = False
    flush_param_grads()


def queue_param_grads(g2, x2, w_grad, b_grad):
    """w_grad += g2^T x2, b_grad += column sums of g2 — some time before this backward pass returns"""
    global _armed
    if not QUEUE:
        from asac_amd import native
        native.xty(g2, x2, w_grad, b_grad, accumulate=True)
        return
    if any(j[2].data_ptr() == w_grad.data_ptr() or (b_grad is not None and j[3] is not None and j[3].data_ptr() == b_grad.data_ptr())
           for j in _pending):
        flush_param_grads()      # (a layer applied twice: its two products add to one gradient, one after the other)
    _pending.append((g2, x2, w_grad, b_grad))
    if len(_pending) == 4:
        flush_param_grads()
    elif not _armed:
        _armed = True
        torch.autograd.Variable._execution_engine.queue_callback(_end_of_backward)


class _RowsLinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        x2 = x.reshape(-1, x.shape[-1])
        ctx.save_for_backward(x2, weight)
        ctx.lead = x.shape[:-1]
        ctx.weight_param, ctx.bias_param = weight, bias
        return torch.addmm(bias, x2, weight.t()).view(*x.shape[:-1], weight.shape[0])

    @staticmethod
    def backward(ctx, g):
        from asac_amd import native
        x2, weight = ctx.saved_tensors
        g2 = g.reshape(-1, g.shape[-1])
        if g2.stride(1) != 1:
            g2 = g2.contiguous()
        gx = (g2 @ weight).view(*ctx.lead, weight.shape[1]) if ctx.needs_input_grad[0] else None
        gw = gb = None
        from algorithm.fused_mlp import direct_enabled, direct_skips
        if (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]) and not direct_skips(ctx.weight_param, ctx.bias_param):
            w_grad, b_grad = ctx.weight_param.grad, ctx.bias_param.grad
            if (direct_enabled() and w_grad is not None and b_grad is not None and w_grad.is_contiguous()
                    and b_grad.is_contiguous()):
                # `loss.backward()` of the learner's step: added straight into the `.grad` views of the flat buffer (no
                # gradient tensors handed to autograd, no accumulation launches)
                queue_param_grads(g2, x2, w_grad, b_grad)
            else:
                gw, gb = torch.empty_like(weight), torch.empty(weight.shape[0], dtype=weight.dtype, device=weight.device)
                native.xty(g2, x2, gw, gb)
        return gx, gw, gb


def rows_linear(linear: torch.nn.Linear, x: torch.Tensor) -> torch.Tensor:
    """`linear(x)`; on the device, over >= MIN_ROWS rows of a trainable layer of <= 128 x 128 features, with the
    one-launch parameter gradients"""
    if (ENABLED and x.is_cuda and x.dtype == torch.float32 and torch.is_grad_enabled() and linear.bias is not None
            and linear.weight.requires_grad and linear.bias.requires_grad and linear.in_features <= 128
            and linear.out_features <= 128 and x.dim() >= 2 and x.shape[-1] == linear.in_features
            and x.numel() // linear.in_features >= MIN_ROWS and x.stride(-1) == 1):
        return _RowsLinearFn.apply(x, linear.weight, linear.bias)
    return linear(x)


class _AffineGeluFn(torch.autograd.Function):
    """gelu(x W^T + b) over the rows of a window batch, narrow input (K <= 64), as one MFMA launch
    (`asac_rows_affine_gelu_forward`, csrc/rows_proj.hip); backward: `gelu_backward` on the saved pre-activation, the
    parameter gradients as a product over the rows (`asac_xty`), the input gradient (if anyone needs it) as the library's GEMM"""

    @staticmethod
    def forward(ctx, x, weight, bias):
        from asac_amd import native
        x2 = x.reshape(-1, x.shape[-1])
        if x2.stride(1) != 1:
            x2 = x2.contiguous()
        N = weight.shape[0]
        y = torch.empty(x2.shape[0], N, dtype=x.dtype, device=x.device)
        pre = torch.empty_like(y)
        native.rows_affine_gelu_forward(x2, weight.detach().contiguous(), bias.detach().contiguous(), y, pre)
        ctx.save_for_backward(x2, pre, weight)
        ctx.params, ctx.lead = (weight, bias), x.shape[:-1]
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, gy):
        from asac_amd import native
        from algorithm.fused_mlp import direct_enabled, direct_skips
        x2, pre, weight = ctx.saved_tensors
        gpre = torch.ops.aten.gelu_backward(gy.reshape(pre.shape).contiguous(), pre, approximate='none')
        gx = (gpre @ weight.detach()).view(*ctx.lead, weight.shape[1]) if ctx.needs_input_grad[0] else None
        gw = gb = None
        w_param, b_param = ctx.params
        if (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]) and not direct_skips(w_param, b_param):
            w_grad, b_grad = w_param.grad, b_param.grad
            if (direct_enabled() and w_grad is not None and b_grad is not None and w_grad.is_contiguous()
                    and b_grad.is_contiguous()):
                queue_param_grads(gpre, x2, w_grad, b_grad)
            else:
                gw, gb = torch.empty_like(weight), torch.empty(weight.shape[0], dtype=weight.dtype, device=weight.device)
                native.xty(gpre, x2, gw, gb)
        return gx, gw, gb


WIDE_MIN_ROWS = 256      # (the wide first layer of a convolution head: 1 024 ... 2 304 frames a pass)


class _WideAffineGeluFn(torch.autograd.Function):
    """gelu(x W^T + b) for an input of hundreds to thousands of features (K > 128: the flattened convolution map in front of
    `ConvLayers.dense`, 2 592 features for 84 x 84 frames) on the MFMA launches of csrc/wide.hip: split-K forward with the
    bias and the GELU in its second launch; backward dpre = g * gelu'(pre) and dx = dpre W in one launch, dW = dpre^T x and db
    in a launch pair — through the library: a GEMM and an elementwise launch forward, two GEMMs, a split reduction and an
    elementwise launch backward, with autograd's accumulation launches behind them."""

    @staticmethod
    def forward(ctx, x, weight, bias, grad_mode):
        from asac_amd import native
        K = x.shape[-1]
        x2 = x.reshape(-1, K)
        if x2.stride(1) != 1 or x2.stride(0) % 4 or x2.data_ptr() % 16:
            x2 = x2.contiguous()
        N = weight.shape[0]
        train = grad_mode and any(ctx.needs_input_grad[:3])
        y = torch.empty(x2.shape[0], N, dtype=x.dtype, device=x.device)
        pre = torch.empty_like(y) if train else None
        native.rows_wide_forward(x2, weight.detach(), bias.detach(), y, pre)
        if train:
            ctx.save_for_backward(x2, pre, weight)
            ctx.params, ctx.lead = (weight, bias), x.shape[:-1]
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, gy):
        from asac_amd import native
        from algorithm.fused_mlp import direct_enabled, direct_skips
        x2, pre, weight = ctx.saved_tensors
        R, K = x2.shape
        g2 = gy.reshape(pre.shape)
        if not g2.is_contiguous() or g2.data_ptr() % 16:
            g2 = g2.contiguous()
        dpre = torch.empty_like(pre)
        dx = torch.empty(R, K, dtype=pre.dtype, device=pre.device) if ctx.needs_input_grad[0] else None
        native.rows_wide_backward_input(g2, pre, weight.detach(), dpre, dx)
        gw = gb = None
        w_param, b_param = ctx.params
        if (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]) and not direct_skips(w_param, b_param):
            w_grad, b_grad = w_param.grad, b_param.grad
            if (direct_enabled() and w_grad is not None and b_grad is not None and w_grad.is_contiguous()
                    and b_grad.is_contiguous()):
                native.rows_wide_backward_params(dpre, x2, w_grad, b_grad, accumulate=True)
            else:
                gw, gb = torch.empty_like(weight), torch.empty(weight.shape[0], dtype=weight.dtype, device=weight.device)
                native.rows_wide_backward_params(dpre, x2, gw, gb)
        return (dx.view(*ctx.lead, K) if dx is not None else None), gw, gb, None


def rows_resblock(block, x):
    """`ResBlock.forward` over >= MIN_ROWS rows on the device as one launch per pass, or None: without a residual path
    (widths differ) from a narrow input — `_AffineGeluFn`; with one at the widths of csrc/rows_proj.hip — the output-block
    function of the attention layers (`seq_layers._OutResRowsFn`)"""
    from torch import nn
    lin = block.linear
    if not (ENABLED and x.is_cuda and x.dtype == torch.float32 and type(block.act) is nn.GELU
            and getattr(block.act, 'approximate', 'none') == 'none' and lin.bias is not None and x.dim() >= 2
            and x.shape[-1] == lin.in_features and x.stride(-1) == 1
            and lin.weight.data_ptr() % 16 == 0 and lin.bias.data_ptr() % 16 == 0):      # (16-byte vector reads of the parameters)
        return None
    from asac_amd import native
    rows = x.numel() // lin.in_features
    if (WIDE and not block.residual and lin.in_features > 128 and rows >= WIDE_MIN_ROWS and lin.weight.is_contiguous()
            and native.rows_wide_supported(rows, lin.in_features, lin.out_features)):
        # the wide first layer of a convolution head (csrc/wide.hip)
        return _WideAffineGeluFn.apply(x, lin.weight, lin.bias, torch.is_grad_enabled())
    if rows < MIN_ROWS:
        return None
    if not block.residual:
        if native.rows_affine_supported(lin.in_features, lin.out_features):
            return _AffineGeluFn.apply(x, lin.weight, lin.bias)
        return None
    if native.rows_proj_supported(lin.in_features):
        from algorithm.nn_models.layers.seq_layers import _OutResRowsFn
        return _OutResRowsFn.apply(x, lin.weight, lin.bias, None)
    return None
