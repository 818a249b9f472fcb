"""Sequence layers of the model-plugin surface (reference `algorithm/nn_models/layers/seq_layers.py`): the
padding-aware `GRU` stack (14-114) and the attention layers (117-851).

-- Recurrent layers of the model-plugin surface.

`GRU` keeps the interface of reference `algorithm/nn_models/layers/seq_layers.py:14-114`
(stack of single-layer batch-first `nn.GRU`s held in `_grus`, per-step hidden states of every
layer returned as `[batch, seq, layers, hidden]`, padding-aware).  The reference packs the
left-aligned sequence with `pack_padded_sequence`, which forces a device->host copy of the valid
lengths on every call (seq_layers.py:71).  Here the valid block is left-aligned with a gather and
the recurrence simply runs over the full window: steps after the valid block cannot influence
earlier outputs and are masked to zero afterwards, so the values at valid positions are the same
and no host synchronisation is needed (the step stays graph-capturable).

On the device, cells that fit `csrc/gru.hip` (input, hidden <= 16, <= 2 layers) run as ONE fused
launch per pass (`algorithm/fused_gru.py`); the cell loop below is the generic path for larger cells
and for CPU tensors (model construction / plugin unit tests — the train step itself is device-only).

-- Attention layers of the model-plugin surface: `MultiheadAttention`, the gate layers and the
episodic (windowed, stateful) `EpisodeMultiheadAttention` stack with absolute / rotary positional
encodings.

API, sub-module names (`q_proj`, `k_proj`, `v_proj`, `out_proj`, `abpe`, `rope`, `attn`, `gatedlayer`,
`layer_norm`, `_attn_list`) and numerics follow reference
`algorithm/nn_models/layers/seq_layers.py:117-851`, so user representation files and checkpoints
interchange.  The math runs as batched GEMMs + softmax on PyTorch-ROCm (MFMA through
rocBLAS/hipBLASLt); nothing here synchronises with the host, so an attention representation stays
inside the captured train step.

Per-layer "hidden state" of the episodic stack = the previous layers' outputs at the query
positions, which lets a window be continued from where the last one stopped:
  * hidden_state None            — run every layer over the full key window
  * is_prev_hidden_state False   — `hidden_state` holds, per layer, outputs for positions BEFORE the
                                   window (acting: history of up to burn_in steps)
  * is_prev_hidden_state True    — `hidden_state` holds the state just before the window's first
                                   element (training: one stored state per sampled window)
"""
import math
import os
from enum import Enum

import torch
from torch import nn

from .linear_layers import LinearLayers

__all__ = ['GRU', 'step_mask_cache', 'POSITIONAL_ENCODING', 'GATE', 'MultiheadAttention', 'GatedResidualLayer', 'GatedOutputLayer',
           'GatedRecurrentLayer', 'GatedCatLayer', 'EpisodeMultiheadAttentionBlock',
           'EpisodeMultiheadAttention', 'AbsolutePositionalEncoding', 'RotaryPositionalEncoding',
           'RotaryPositionalEncoding2']


class GRU(nn.Module):
    def __init__(self, input_size, hidden_size, num_layers=1, bias=True, dropout=0.0,
                 device=None, dtype=None):
        super().__init__()
        self.num_layers = num_layers
        self._fusable = bool(bias) and dropout == 0.0
        self._grus = nn.ModuleList([
            nn.GRU(input_size=input_size if i == 0 else hidden_size, hidden_size=hidden_size,
                   num_layers=1, bias=bias, batch_first=True, dropout=dropout,
                   device=device, dtype=dtype)
            for i in range(num_layers)])

    def forward(self, x, h0=None, padding_mask=None):
        """
        x: [batch, seq, input]; h0: [batch, layers, hidden] or None; padding_mask: bool [batch, seq]
        returns output [batch, seq, hidden], hn [batch, seq, layers, hidden]
        """
        from algorithm.fused_gru import fused_gru, fused_gru_supported   # lazy: avoids an import cycle
        cell = self._grus[0]
        if self._fusable and fused_gru_supported(x, cell.input_size, cell.hidden_size, self.num_layers):
            # one launch for the whole window (csrc/gru.hip); same values as the cell loop below
            return fused_gru(x, h0, padding_mask, list(self._grus), layer=self)     # (top layer [B, L, H], hn)
        if self._fusable and x.is_cuda:
            from algorithm.fused_gru_wide import fused_gru_wide, fused_gru_wide_supported
            if fused_gru_wide_supported(x, list(self._grus)):
                # hidden 32 / 64 / 128: the recurrence of each layer as one MFMA launch (csrc/gru_wide.hip)
                return fused_gru_wide(x, h0, padding_mask, list(self._grus), layer=self)

        if h0 is not None:
            h0 = h0.transpose(0, 1).contiguous()  # [layers, batch, hidden]
        batch, seq_len, _ = x.shape

        if padding_mask is not None:
            lead = padding_mask.long().argmin(dim=1, keepdim=True)  # first valid position
            steps = torch.arange(seq_len, device=x.device).unsqueeze(0)
            fwd_idx = torch.clamp(steps + lead, max=seq_len - 1)  # left-align the valid block
            bwd_idx = torch.clamp(steps - lead, min=0)            # and put results back
            x = x.gather(1, fwd_idx.unsqueeze(-1).expand(-1, -1, x.shape[-1]))

        per_layer = []
        for i, gru in enumerate(self._grus):
            out, _ = gru(x, None if h0 is None else h0[i:i + 1])
            x = out
            if padding_mask is not None:
                out = out.gather(1, bwd_idx.unsqueeze(-1).expand(-1, -1, out.shape[-1]))
                out = out.masked_fill(padding_mask.unsqueeze(-1), 0.0)
            per_layer.append(out)

        return per_layer[-1], torch.stack(per_layer, dim=2)


class POSITIONAL_ENCODING(Enum):
    ABSOLUTE = 1
    ABSOLUTE_CAT = 2
    ROPE = 3
    ROPE2 = 4


class GATE(Enum):
    RESIDUAL = 1
    OUTPUT = 2
    RECURRENT = 3
    CAT = 4


# ------------------------------------------------------------------------------------------------
# positional encodings
# ------------------------------------------------------------------------------------------------
class AbsolutePositionalEncoding(nn.Module):
    def __init__(self, d_model: int, max_seq_len: int = 5000):
        super().__init__()
        self.d_model = d_model
        pos = torch.arange(max_seq_len, dtype=torch.float64).unsqueeze(1)
        i = torch.arange(0, d_model, 2, dtype=torch.float64)
        pe = torch.zeros(max_seq_len, d_model, dtype=torch.float64)
        # even slot i: sin(pos / 10000^(2i/d)); odd slot i+1: cos(pos / 10000^(2(i+1)/d))
        pe[:, 0::2] = torch.sin(pos / torch.pow(10000., 2 * i / d_model))
        pe[:, 1::2] = torch.cos(pos / torch.pow(10000., 2 * (i + 1) / d_model))[:, :d_model // 2]
        self.register_buffer('pe', pe.to(torch.float32))

    @torch.no_grad()
    def forward(self, indexes):
        return self.pe[indexes.type(torch.int64)]


class RotaryPositionalEncoding(nn.Module):
    """Complex-pair rotary encoding: consecutive feature pairs are rotated by index * theta_i."""

    def __init__(self, d_model: int, max_seq_len: int = 5000, theta: float = 10000.0):
        super().__init__()
        freqs = 1.0 / (theta ** (torch.arange(0, d_model, 2)[: (d_model // 2)] / d_model))
        angles = torch.outer(torch.arange(max_seq_len), freqs)
        self.register_buffer('freqs_cis', torch.polar(torch.ones_like(angles), angles))

    def _rotate(self, x, indexes):
        rot = self.freqs_cis[indexes.type(torch.int64)]
        xc = torch.view_as_complex(x.reshape(*x.shape[:-1], -1, 2))
        return torch.view_as_real(xc * rot).flatten(2).type_as(x)

    def forward(self, xq_indexes, xk_indexes, xq, xk):
        return self._rotate(xq, xq_indexes), self._rotate(xk, xk_indexes)


class RotaryPositionalEncoding2(nn.Module):
    """Half-split rotary encoding: feature j is paired with j + d/2."""

    def __init__(self, d_model: int, max_seq_len: int = 5000, base: int = 10_000):
        super().__init__()
        self.d_model = d_model
        theta = 1. / (base ** (torch.arange(0, d_model, 2).float() / d_model))
        idx_theta = torch.einsum('n,d->nd', torch.arange(max_seq_len).float(), theta)
        idx_theta2 = torch.cat([idx_theta, idx_theta], dim=1)
        self.register_buffer('cos_cached', idx_theta2.cos())
        self.register_buffer('sin_cached', idx_theta2.sin())

    def _neg_half(self, x):
        d_2 = self.d_model // 2
        return torch.cat([-x[:, :, d_2:], x[:, :, :d_2]], dim=-1)

    def _rotate(self, x, indexes):
        rope, rest = x[..., :self.d_model], x[..., self.d_model:]
        rope = rope * self.cos_cached[indexes] + self._neg_half(rope) * self.sin_cached[indexes]
        return torch.cat((rope, rest), dim=-1)

    def forward(self, xq_indexes, xk_indexes, xq, xk):
        return self._rotate(xq, xq_indexes), self._rotate(xk, xk_indexes)


# ------------------------------------------------------------------------------------------------
FUSED_PROJECTIONS = True      # q / k / v Linear projections inside the attention launch when they are plain Linears
# several heads: the core of all heads as one MFMA launch (csrc/attn_mh.hip); ASAC_ATTN_MH=0 keeps the module path (A/B runs)
FUSED_MULTIHEAD = os.environ.get('ASAC_ATTN_MH', '1') != '0'
# the Linear layers around the multi-head core as one launch each (csrc/rows_proj.hip)   (0: library GEMMs — A/B runs)
FUSED_ROWS_PROJ = os.environ.get('ASAC_ROWS_PROJ', '1') != '0'
# ... and, for windows of <= 16 positions, the q / k / v projections inside the core's forward launch   (0: a launch of their own)
FUSED_QKV_IN_CORE = os.environ.get('ASAC_QKV_IN_CORE', '1') != '0'
# ... and the block's backward as one launch too   (0: ResBlock backward, core backward, projections' backward)
FUSED_BLOCK_BACKWARD = os.environ.get('ASAC_ATTN_BLOCK_BWD', '1') != '0'


class _AttnCoreFn(torch.autograd.Function):
    """scores / mask / softmax / weighted sum of one attention head as one launch per pass
    (`asac_attention_forward/backward`, csrc/attn.hip) -> (out, weights * keep, keep)"""

    @staticmethod
    def forward(ctx, q, k, v, mask):
        from asac_amd import native
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        B, Lq, D = q.shape
        out = torch.empty(B, Lq, D, dtype=q.dtype, device=q.device)
        weights = torch.empty(B, Lq, k.shape[1], dtype=q.dtype, device=q.device)
        keep = torch.empty(B, Lq, dtype=q.dtype, device=q.device)
        native.attention_forward(q, k, v, mask, out, weights, keep)
        ctx.save_for_backward(q, k, v, weights)
        ctx.mark_non_differentiable(keep)
        ctx.set_materialize_grads(False)
        return out, weights, keep

    @staticmethod
    def backward(ctx, g_out, g_w, _g_keep):
        from asac_amd import native
        q, k, v, weights = ctx.saved_tensors
        if g_out is None and g_w is None:
            return None, None, None, None
        if g_out is None:
            g_out = torch.zeros(q.shape, dtype=q.dtype, device=q.device)
        g_q, g_k, g_v = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        native.attention_backward(q, k, v, weights, g_out.contiguous(), None if g_w is None else g_w.contiguous(),
                                  g_q, g_k, g_v)
        return g_q, g_k, g_v, None


class _AttnMhFn(torch.autograd.Function):
    """scores / mask / softmax / weighted sum of ALL heads and the head-averaged weights as one MFMA launch per pass
    (`asac_attention_mh_forward/backward`, csrc/attn_mh.hip): q, k, v [B, L, heads * d] -> (out [B, Lq, heads * d] with the
    heads concatenated, mean-over-heads weights * keep, keep)"""

    @staticmethod
    def forward(ctx, q, k, v, mask, heads, row_zero=None):
        """`row_zero` bool [B, Lq]: the caller's padded positions — the fourth result is then keep * ~row_zero, the one
        factor the caller's output needs (else it is `keep` again)"""
        from asac_amd import native
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        B, Lq, E = q.shape
        Lk = k.shape[1]
        out = torch.empty(B, Lq, E, dtype=q.dtype, device=q.device)
        weights = torch.empty(B, Lq, Lk, dtype=q.dtype, device=q.device)
        keep = torch.empty(B, Lq, dtype=q.dtype, device=q.device)
        keep_rows = torch.empty(B, Lq, dtype=q.dtype, device=q.device) if row_zero is not None else keep
        p_heads = torch.empty(B, heads, Lq, Lk, dtype=q.dtype, device=q.device) if any(ctx.needs_input_grad[:3]) else None
        native.attention_mh_forward(q, k, v, mask, heads, out, weights, keep, p_heads,
                                    None if row_zero is None else row_zero.contiguous(),
                                    keep_rows if row_zero is not None else None)
        if p_heads is not None:
            ctx.save_for_backward(q, k, v, p_heads, *([mask] if mask is not None else []))
        ctx.heads, ctx.has_mask = heads, mask is not None
        ctx.mark_non_differentiable(keep, keep_rows)
        ctx.set_materialize_grads(False)
        return out, weights, keep, keep_rows

    @staticmethod
    def backward(ctx, g_out, g_w, _g_keep, _g_keep_rows=None):
        from asac_amd import native
        q, k, v, p_heads, *rest = ctx.saved_tensors
        if g_out is None and g_w is None:
            return None, None, None, None, None, None
        if g_out is None:
            g_out = torch.zeros(q.shape, dtype=q.dtype, device=q.device)
        g_q, g_k, g_v = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        native.attention_mh_backward(q, k, v, rest[0] if ctx.has_mask else None, ctx.heads, p_heads, g_out.contiguous(),
                                     None if g_w is None else g_w.contiguous(), g_q, g_k, g_v)
        return g_q, g_k, g_v, None, None, None


class _AttnProjFn(torch.autograd.Function):
    """`_AttnCoreFn` with the q / k / v Linear projections — and, with 8 parameters, the output ResBlock
    y = (GELU(Wo o + bo) + o) * keep — on chip (`asac_attention_proj_*`): x_q, x_k in, (out, weights * keep, keep) out.
    The parameter gradients are added straight into their `.grad` views when those are consecutive slices of one
    buffer (the learner's flat gradient buffer), else returned."""

    @staticmethod
    def forward(ctx, xq, xk, mask, row_zero, *params):
        from asac_amd import native
        xk = xk if xk.stride(-1) == 1 else xk.contiguous()
        ctx.tail = 0
        if isinstance(xq, int):      # the queries are the last `xq` rows of x_k: one gradient, no slice node
            ctx.tail = xq
            xq = xk[:, -xq:]
        xq = xq if xq.stride(-1) == 1 else xq.contiguous()
        B, Lq, E = xq.shape
        pd = [t.detach().contiguous() for t in params]
        out = torch.empty(B, Lq, E, dtype=xq.dtype, device=xq.device)
        weights = torch.empty(B, Lq, xk.shape[1], dtype=xq.dtype, device=xq.device)
        keep = torch.empty(B, Lq, dtype=xq.dtype, device=xq.device)
        attn_out = torch.empty(B, Lq, E, dtype=xq.dtype, device=xq.device) if len(params) == 8 else None
        native.attention_proj_forward(xq, xk, pd, mask, out, weights, keep, attn_out, row_zero)
        ctx.save_for_backward(xk if ctx.tail else xq, xk, weights, keep, *([attn_out] if attn_out is not None else []))
        ctx.params, ctx.row_zero = params, row_zero
        ctx.mark_non_differentiable(keep)
        ctx.set_materialize_grads(False)
        return out, weights, keep

    @staticmethod
    def backward(ctx, g_out, g_w, _g_keep):
        from asac_amd import native
        from algorithm.fused_mlp import _flat_alias, direct_enabled
        xq, xk, weights, keep, *rest = ctx.saved_tensors
        attn_out = rest[0] if rest else None
        params, tail = ctx.params, ctx.tail
        if tail:
            xq = xk[:, -tail:]
        if g_out is None and g_w is None:
            return (None,) * (4 + len(params))
        if g_out is None:
            g_out = torch.zeros(xq.shape[0], xq.shape[1], xq.shape[2], dtype=xq.dtype, device=xq.device)
        B, Lq, E = xq.shape
        Lk = xk.shape[1]
        # (tail: the queries are the last rows of the keys — the launch adds their gradient into those rows of g_xk)
        g_xq = None if tail else torch.empty(B, Lq, E, dtype=xq.dtype, device=xq.device)
        g_xk = torch.empty(B, Lk, E, dtype=xq.dtype, device=xq.device)
        if g_out.stride(2) != 1:
            g_out = g_out.contiguous()
        ws = torch.empty(native.attention_proj_workspace(B, Lq, Lk, E), dtype=xq.dtype, device=xq.device)
        pd = [t.detach().contiguous() for t in params]
        gw = None if g_w is None else g_w.contiguous()
        flat = None
        if direct_enabled() and all(p.requires_grad and p.grad is not None for p in params):
            flat = _flat_alias([p.grad for p in params])
        from algorithm.fused_mlp import DeferredPartialSums
        later = DeferredPartialSums.active()
        if flat is not None:
            native.attention_proj_backward(xq, xk, pd, weights, g_out, gw, g_xq, g_xk, flat,
                                           native.SUM_DEFER if later is not None else True, ws, keep, attn_out, ctx.row_zero)
            if later is not None:
                slabs, n = native.attention_proj_partials(ws, E, len(params) == 8)
                later.add(ws, slabs, 16, n, n, flat, accumulate=True)
            return (g_xq, g_xk, None, None, *([None] * len(params)))
        g = torch.empty(sum(p.numel() for p in params), dtype=xq.dtype, device=xq.device)
        native.attention_proj_backward(xq, xk, pd, weights, g_out, gw, g_xq, g_xk, g,
                                       native.ATTN_SUM_DEFER if later is not None else False, ws, keep, attn_out, ctx.row_zero)
        if later is not None:       # (the workgroups' partials are summed into g with the other walks', one launch)
            slabs, n = native.attention_proj_partials(ws, E, len(params) == 8)
            later.add(ws, slabs, 16, n, n, g)
        grads, off = [], 0
        for p_ in params:
            k = p_.numel()
            grads.append(g[off:off + k].view(p_.shape) if p_.requires_grad else None)
            off += k
        if later is not None:       # (the gradients reach the caller through `later.flush()`)
            later.record(params, grads)
            grads = [None] * len(params)
        return (g_xq, g_xk, None, None, *grads)


def _rows_param_grads(ctx_needs, params, g2, x2):
    """weight / bias gradient of one Linear from its output gradient rows g2 and input rows x2 (`asac_xty`): added straight
    into the `.grad` views under the learner's direct mode, else returned"""
    from asac_amd import native
    from algorithm.fused_mlp import direct_enabled, direct_skips
    weight, bias = params
    if not any(ctx_needs) or direct_skips(weight, bias):
        return None, None
    w_grad, b_grad = weight.grad, bias.grad
    if direct_enabled() and w_grad is not None and b_grad is not None and w_grad.is_contiguous() and b_grad.is_contiguous():
        from algorithm.fused_rows_linear import queue_param_grads
        queue_param_grads(g2, x2, w_grad, b_grad)
        return None, None
    gw, gb = torch.empty_like(weight), torch.empty_like(bias)
    native.xty(g2, x2, gw, gb)
    return gw, gb


class _QkvRowsFn(torch.autograd.Function):
    """`q_proj(x[:, -tail:]), k_proj(x), v_proj(x)` — three nn.Linear(E, E) over the rows of a window batch — as ONE MFMA
    launch; backward: the input gradient of the three as one launch, the parameter gradients as products over the rows
    (`asac_rows_proj_*`, csrc/rows_proj.hip; `asac_xty`)"""

    @staticmethod
    def forward(ctx, x, tail, wq, bq, wk, bk, wv, bv):
        from asac_amd import native
        if x.stride(2) != 1 or (x.stride(0) | x.stride(1) | (x.data_ptr() >> 2)) & 3:
            x = x.contiguous()
        B, L, E = x.shape
        q = torch.empty(B, tail, E, dtype=x.dtype, device=x.device)
        k, v = torch.empty(B, L, E, dtype=x.dtype, device=x.device), torch.empty(B, L, E, dtype=x.dtype, device=x.device)
        native.rows_proj_forward(x, [wq.detach(), wk.detach(), wv.detach()], [bq.detach(), bk.detach(), bv.detach()],
                                 [tail, L, L], [q, k, v])
        ctx.save_for_backward(x)
        ctx.tail, ctx.params = tail, (wq, bq, wk, bk, wv, bv)
        return q, k, v

    @staticmethod
    def backward(ctx, gq, gk, gv):
        from asac_amd import native
        x, = ctx.saved_tensors
        B, L, E = x.shape
        wq, bq, wk, bk, wv, bv = ctx.params
        grads = [g.contiguous() for g in (gq, gk, gv)]
        tails = [ctx.tail, L, L]
        gx = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty(B, L, E, dtype=x.dtype, device=x.device)
            native.rows_proj_backward(grads, tails, [wq.detach(), wk.detach(), wv.detach()], gx)
        x2 = x.reshape(-1, E)
        xq2 = x2 if ctx.tail == L else x[:, -ctx.tail:].reshape(-1, E)
        out = [gx, None]
        for j, (g, xin) in enumerate(zip(grads, (xq2, x2, x2))):
            out.extend(_rows_param_grads(ctx.needs_input_grad[2 + 2 * j:4 + 2 * j], ctx.params[2 * j:2 * j + 2],
                                         g.view(-1, E), xin))
        return tuple(out)


class _QkvAttnMhFn(torch.autograd.Function):
    """`_QkvRowsFn` and `_AttnMhFn` as ONE forward launch for windows of <= 16 positions (`asac_attention_mh_proj_forward`:
    the projections run in front of the scores inside the launch); backward: the core's launch, then the projections' input
    gradient and parameter-gradient products as in `_QkvRowsFn`"""

    @staticmethod
    def forward(ctx, x, tail, wq, bq, wk, bk, wv, bv, mask, heads, row_zero=None, out_block=None):
        """`out_block` = (weight, bias) of the output ResBlock: it runs behind the core in the same launch and two more tensors
        come back (y, pre) — non-differentiable HERE: `_OutResSavedFn` turns them into the block's output and carries its
        backward"""
        from asac_amd import native
        if x.stride(2) != 1 or (x.stride(0) | x.stride(1) | (x.data_ptr() >> 2)) & 3:
            x = x.contiguous()
        B, L, E = x.shape
        dd = dict(dtype=x.dtype, device=x.device)
        q, k, v = torch.empty(B, tail, E, **dd), torch.empty(B, L, E, **dd), torch.empty(B, L, E, **dd)
        out, weights, keep = torch.empty(B, tail, E, **dd), torch.empty(B, tail, L, **dd), torch.empty(B, tail, **dd)
        keep_rows = torch.empty(B, tail, **dd) if row_zero is not None else keep
        need = any(ctx.needs_input_grad[i] for i in (0, 2, 3, 4, 5, 6, 7))
        p_heads = torch.empty(B, heads, tail, L, **dd) if need else None
        y = pre = None
        if out_block is not None:
            y, pre = torch.empty(B, tail, E, **dd), torch.empty(B, tail, E, **dd)
        native.attention_mh_proj_forward(x, [wq.detach(), wk.detach(), wv.detach()], [bq.detach(), bk.detach(), bv.detach()],
                                         mask, heads, q, k, v, out, weights, keep, p_heads,
                                         None if row_zero is None else (row_zero if row_zero.stride(-1) == 1 else row_zero.contiguous()),
                                         keep_rows if row_zero is not None else None,
                                         *(() if out_block is None else (out_block[0].detach().contiguous(),
                                                                         out_block[1].detach().contiguous(), y, pre)))
        if need:
            ctx.save_for_backward(x, q, k, v, p_heads, *([mask] if mask is not None else []))
        ctx.tail, ctx.heads, ctx.has_mask, ctx.params = tail, heads, mask is not None, (wq, bq, wk, bk, wv, bv)
        ctx.set_materialize_grads(False)
        if out_block is None:
            ctx.mark_non_differentiable(keep, keep_rows)
            return out, weights, keep, keep_rows
        ctx.mark_non_differentiable(keep, keep_rows, y, pre)
        return out, weights, keep, keep_rows, y, pre

    @staticmethod
    def backward(ctx, g_out, g_w, _g_keep, _g_keep_rows=None, _g_y=None, _g_pre=None):
        from asac_amd import native
        if g_out is None and g_w is None:
            return (None,) * 12
        x, q, k, v, p_heads, *rest = ctx.saved_tensors
        B, L, E = x.shape
        if g_out is None:
            g_out = torch.zeros(q.shape, dtype=q.dtype, device=q.device)
        g_q, g_k, g_v = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        native.attention_mh_backward(q, k, v, rest[0] if ctx.has_mask else None, ctx.heads, p_heads, g_out.contiguous(),
                                     None if g_w is None else g_w.contiguous(), g_q, g_k, g_v)
        wq, bq, wk, bk, wv, bv = ctx.params
        grads, tails = [g_q, g_k, g_v], [ctx.tail, L, L]
        gx = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty(B, L, E, dtype=x.dtype, device=x.device)
            native.rows_proj_backward(grads, tails, [wq.detach(), wk.detach(), wv.detach()], gx)
        x2 = x.reshape(-1, E)
        xq2 = x2 if ctx.tail == L else x[:, -ctx.tail:].reshape(-1, E)
        out = [gx, None]
        for j, (g, xin) in enumerate(zip(grads, (xq2, x2, x2))):
            out.extend(_rows_param_grads(ctx.needs_input_grad[2 + 2 * j:4 + 2 * j], ctx.params[2 * j:2 * j + 2],
                                         g.view(-1, E), xin))
        return (*out, None, None, None, None)


class _AttnBlockFn(torch.autograd.Function):
    """The whole attention block for windows of <= 16 positions — q / k / v projections, core, output ResBlock with the
    dead-row / padded-row factor — as ONE launch forward (`asac_attention_mh_proj_forward`) and ONE backward
    (`asac_attention_mh_block_backward`), plus the four parameter-gradient products (`asac_xty`, queued)"""

    @staticmethod
    def forward(ctx, x, tail, wq, bq, wk, bk, wv, bv, wo, bo, mask, heads, row_zero, scaled):
        from asac_amd import native
        if x.stride(2) != 1 or (x.stride(0) | x.stride(1) | (x.data_ptr() >> 2)) & 3:
            x = x.contiguous()
        B, L, E = x.shape
        dd = dict(dtype=x.dtype, device=x.device)
        q, k, v = torch.empty(B, tail, E, **dd), torch.empty(B, L, E, **dd), torch.empty(B, L, E, **dd)
        out, weights, keep = torch.empty(B, tail, E, **dd), torch.empty(B, tail, L, **dd), torch.empty(B, tail, **dd)
        keep_rows = torch.empty(B, tail, **dd) if row_zero is not None else keep
        need = any(ctx.needs_input_grad[i] for i in (0, 2, 3, 4, 5, 6, 7, 8, 9))
        p_heads = torch.empty(B, heads, tail, L, **dd) if need else None
        y, pre = torch.empty(B, tail, E, **dd), torch.empty(B, tail, E, **dd)
        native.attention_mh_proj_forward(x, [wq.detach(), wk.detach(), wv.detach()], [bq.detach(), bk.detach(), bv.detach()],
                                         mask, heads, q, k, v, out, weights, keep, p_heads,
                                         None if row_zero is None else (row_zero if row_zero.stride(-1) == 1 else row_zero.contiguous()),
                                         keep_rows if row_zero is not None else None,
                                         wo.detach().contiguous(), bo.detach().contiguous(), y, pre)
        if need:
            # (`scaled`: the block's output was multiplied by keep_rows — with a row mask — or by keep — with an attention mask)
            ctx.save_for_backward(x, q, k, v, p_heads, out, pre, *([keep_rows] if scaled else []), *([mask] if mask is not None else []))
        ctx.tail, ctx.heads, ctx.has_mask, ctx.scaled = tail, heads, mask is not None, scaled
        ctx.params = (wq, bq, wk, bk, wv, bv, wo, bo)
        ctx.mark_non_differentiable(keep, keep_rows)
        ctx.set_materialize_grads(False)
        return y, weights, keep, keep_rows

    @staticmethod
    def backward(ctx, g_y, g_w, _g_keep, _g_keep_rows=None):
        from asac_amd import native
        if g_y is None and g_w is None:
            return (None,) * 14
        x, q, k, v, p_heads, out, pre, *rest = ctx.saved_tensors
        scale = rest.pop(0) if ctx.scaled else None
        mask = rest[0] if ctx.has_mask else None
        B, L, E = x.shape
        if g_y is None:
            g_y = torch.zeros(q.shape, dtype=q.dtype, device=q.device)
        wq, bq, wk, bk, wv, bv, wo, bo = ctx.params
        g_q, g_k, g_v = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        g_pre, g_x = torch.empty_like(pre), torch.empty(B, L, E, dtype=x.dtype, device=x.device)
        if g_y.stride(2) != 1 or (g_y.stride(0) | g_y.stride(1) | (g_y.data_ptr() >> 2)) & 3:
            g_y = g_y.contiguous()      # (otherwise a slice — the gradient of a concatenation's part — is read in place)
        native.attention_mh_block_backward(q, k, v, mask, ctx.heads, p_heads, g_y, pre,
                                           None if scale is None else scale.contiguous(), wo.detach().contiguous(),
                                           None if g_w is None else g_w.contiguous(),
                                           [wq.detach(), wk.detach(), wv.detach()], g_q, g_k, g_v, g_pre, g_x)
        x2 = x.reshape(-1, E)
        xq2 = x2 if ctx.tail == L else x[:, -ctx.tail:].reshape(-1, E)
        grads = [g_x if ctx.needs_input_grad[0] else None, None]
        # (the output block's product first: the order the three-launch form queues them in)
        go = _rows_param_grads(ctx.needs_input_grad[8:10], (wo, bo), g_pre.view(-1, E), out.view(-1, E))
        for j, (g, xin) in enumerate(zip((g_q, g_k, g_v), (xq2, x2, x2))):
            grads.extend(_rows_param_grads(ctx.needs_input_grad[2 + 2 * j:4 + 2 * j], ctx.params[2 * j:2 * j + 2], g.view(-1, E), xin))
        return (*grads, *go, None, None, None, None)


class _OutResRowsFn(torch.autograd.Function):
    """`(x + gelu(linear(x))) * row_scale[..., None]` — the output ResBlock of an attention layer and its dead-row / padded-row
    factor — as one MFMA launch per pass (`asac_rows_resblock_*`, csrc/rows_proj.hip); the parameter gradients from `asac_xty`"""

    @staticmethod
    def forward(ctx, x, weight, bias, row_scale):
        from asac_amd import native
        x2 = x.reshape(-1, x.shape[-1])
        if x2.stride(1) != 1 or (x2.stride(0) | (x2.data_ptr() >> 2)) & 3:
            x2 = x2.contiguous()
        y, pre = torch.empty(x2.shape, dtype=x.dtype, device=x.device), torch.empty(x2.shape, dtype=x.dtype, device=x.device)
        sc = None if row_scale is None else row_scale.reshape(-1).contiguous()
        native.rows_resblock_forward(x2, weight.detach(), bias.detach(), sc, y, pre)
        ctx.save_for_backward(x2, pre, *([sc] if sc is not None else []))
        ctx.params, ctx.lead = (weight, bias), x.shape
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, gy):
        from asac_amd import native
        x2, pre, *rest = ctx.saved_tensors
        weight, bias = ctx.params
        gy2 = gy.reshape(x2.shape).contiguous()
        gx, gpre = torch.empty_like(pre), torch.empty_like(pre)
        native.rows_resblock_backward(gy2, pre, weight.detach(), rest[0] if rest else None, gx, gpre)
        gw, gb = _rows_param_grads(ctx.needs_input_grad[1:3], ctx.params, gpre, x2)
        return gx.view(ctx.lead), gw, gb, None


class _OutResSavedFn(torch.autograd.Function):
    """`_OutResRowsFn` whose forward already ran (inside `asac_attention_mh_proj_forward`): y and pre are handed in, the
    backward is the same"""

    @staticmethod
    def forward(ctx, x, weight, bias, row_scale, y_pre):
        y, pre = y_pre          # (a tuple, not tensor arguments: the result must not be a view of an input)
        x2 = x.reshape(-1, x.shape[-1])
        sc = None if row_scale is None else row_scale.reshape(-1)
        ctx.save_for_backward(x2, pre.view(x2.shape), *([sc] if sc is not None else []))
        ctx.params, ctx.lead = (weight, bias), x.shape
        return y.view_as(x)

    @staticmethod
    def backward(ctx, gy):
        gx, gw, gb, _ = _OutResRowsFn.backward(ctx, gy)
        return gx, gw, gb, None, None


def _rows_proj_ok(module, query, key) -> bool:
    """the q / k / v projections of `module` as one `_QkvRowsFn` launch: plain Linears E -> E over a device window batch whose
    query is the key tensor or its newest positions"""
    if not (FUSED_ROWS_PROJ and key.is_cuda and key.dtype == torch.float32 and key.dim() == 3 and key.stride(2) == 1):
        return False
    if not (query is key or _is_tail_view(query, key)):
        return False
    E = module.embed_dim
    from asac_amd import native
    if not native.rows_proj_supported(E):
        return False
    for ll in (module.q_proj, module.k_proj, module.v_proj):
        lin = _plain_linear(ll)
        if lin is None or lin.in_features != E or lin.out_features != E:
            return False
    return True


def _is_tail_view(query, key):
    """query is `key[:, -q:]` (the episode blocks' cut query): same memory, so one gradient serves both"""
    q = query.shape[1]
    return (q <= key.shape[1] and query.shape[0] == key.shape[0]
            and query.shape[2] == key.shape[2] and query.stride() == key.stride()
            and query.untyped_storage().data_ptr() == key.untyped_storage().data_ptr()
            and query.storage_offset() == key.storage_offset() + (key.shape[1] - q) * key.stride(1))


def _plain_resblock(ll, width):
    """the Linear of a `LinearLayers` stack that is exactly ONE residual GELU ResBlock width -> width, or None"""
    from .linear_layers import ResBlock
    mods = [m for m in ll.dense if not (isinstance(m, nn.Dropout) and m.p == 0)]
    if len(mods) != 1 or not isinstance(mods[0], ResBlock):
        return None
    rb = mods[0]
    if not (rb.residual and type(rb.act) is nn.GELU and getattr(rb.act, 'approximate', 'none') == 'none'
            and rb.linear.bias is not None and rb.linear.in_features == rb.linear.out_features == width
            and _params_aligned16(rb.linear)):
        return None
    return rb.linear


def _params_aligned16(linear) -> bool:
    """The MFMA row kernels read weights and biases as 16-byte vectors and reject anything else (`bad_arg` -> AsacNativeError).
    Parameters are views packed back to back in the learner's flat buffer: only SEGMENT starts are aligned, so a plugin
    whose model holds an earlier parameter of numel % 4 != 0 (a scalar gate, an odd-width Linear) shifts everything behind
    it — such layers take the nn.Linear path instead of crashing in forward."""
    return (linear.weight.data_ptr() % 16 == 0 and linear.weight.is_contiguous()
            and (linear.bias is None or (linear.bias.data_ptr() % 16 == 0 and linear.bias.is_contiguous())))


def _plain_linear(ll):
    """the single nn.Linear (with bias) of a `LinearLayers` stack that is nothing else, or None"""
    mods = [m for m in ll.dense if not (isinstance(m, nn.Dropout) and m.p == 0)]
    if len(mods) == 1 and type(mods[0]) is nn.Linear and mods[0].bias is not None and _params_aligned16(mods[0]):
        return mods[0]
    return None


def _fused_core_ok(q, k, num_heads, dropout_active) -> bool:
    if not (q.is_cuda and q.dtype == torch.float32 and num_heads == 1 and not dropout_active):
        return False
    from asac_amd import native
    return native.attention_supported(q.shape[1], k.shape[1], q.shape[2])


class MultiheadAttention(nn.Module):
    def __init__(self, embed_dim: int, num_heads: int = 1, pe=None, qkv_dense_depth: int = 0,
                 out_dense_depth: int = 0, out_size: int | None = None, dropout: float = 0.) -> None:
        super().__init__()
        self.embed_dim, self.num_heads = embed_dim, num_heads
        self.head_dim = embed_dim // num_heads
        assert self.head_dim * num_heads == embed_dim, 'embed_dim must be divisible by num_heads'
        self.pe, self.dropout = pe, dropout

        in_dim = embed_dim
        if pe == POSITIONAL_ENCODING.ABSOLUTE:
            self.abpe = AbsolutePositionalEncoding(embed_dim)
        elif pe == POSITIONAL_ENCODING.ABSOLUTE_CAT:
            self.abpe = AbsolutePositionalEncoding(embed_dim)
            in_dim = embed_dim * 2
        elif pe == POSITIONAL_ENCODING.ROPE:
            self.rope = RotaryPositionalEncoding(embed_dim)
        elif pe == POSITIONAL_ENCODING.ROPE2:
            self.rope = RotaryPositionalEncoding2(embed_dim)

        proj = lambda: LinearLayers(in_dim, dense_n=embed_dim, dense_depth=qkv_dense_depth,  # noqa: E731
                                    output_size=embed_dim, dropout=dropout)
        self.q_proj, self.k_proj, self.v_proj = proj(), proj(), proj()
        self.out_proj = LinearLayers(embed_dim, dense_n=embed_dim, dense_depth=out_dense_depth,
                                     output_size=out_size, dropout=dropout)

    def _split_heads(self, x):
        """[bsz, len, embed] -> [bsz * heads, len, head_dim], head-major like chunk+cat on dim 0"""
        if self.num_heads == 1:
            return x
        b, l, _ = x.shape
        return x.view(b, l, self.num_heads, self.head_dim).permute(2, 0, 1, 3).reshape(self.num_heads * b, l, self.head_dim)

    def forward(self, query, key, value, query_index=None, key_index=None, key_padding_mask=None, attn_mask=None,
                out_row_mask=None):
        """query [batch, q, E]; key / value [batch, k, E]; *_index [batch, len]; key_padding_mask
        [batch, k] (True = ignore); attn_mask [batch, q, k] or [q, k] (True = blocked); out_row_mask
        [batch, q] (True = zero that output row: the caller's padded positions)
        -> (output [batch, q, E_out], weights [batch, q, k] averaged over heads)"""
        lead = query.shape[:-2]
        same_kv = value is key
        query, key, value = (t.reshape(-1, *t.shape[-2:]) for t in (query, key, value))
        bsz, q_len, k_len = query.shape[0], query.shape[1], key.shape[1]
        if key_padding_mask is not None:
            key_padding_mask = key_padding_mask.reshape(-1, key_padding_mask.shape[-1])
        if attn_mask is not None:
            assert attn_mask.dim() in (2, 3)

        if self.pe:       # (None and False both mean no positional encoding)
            if query_index is None:
                query_index = torch.arange(q_len, device=query.device).unsqueeze(0).expand(bsz, -1)
            if key_index is None:
                key_index = torch.arange(k_len, device=key.device).unsqueeze(0).expand(bsz, -1)
        if self.pe == POSITIONAL_ENCODING.ABSOLUTE:
            query = self.abpe(query_index) + query
            kpe = self.abpe(key_index)
            key, value = kpe + key, kpe + value
        elif self.pe == POSITIONAL_ENCODING.ABSOLUTE_CAT:
            query = torch.cat([query, self.abpe(query_index)], dim=-1)
            kpe = self.abpe(key_index)
            key, value = torch.cat([key, kpe], dim=-1), torch.cat([value, kpe], dim=-1)

        if (self.pe is None or self.pe is False) and same_kv and FUSED_PROJECTIONS and query.dim() == 3 \
                and _fused_core_ok(query, key, self.num_heads, self.training and self.dropout > 0.):
            lq, lk, lv = _plain_linear(self.q_proj), _plain_linear(self.k_proj), _plain_linear(self.v_proj)
            if lq is not None and lk is not None and lv is not None and lq.in_features == lq.out_features == self.embed_dim:
                # projections + scores / mask / softmax / weighted sum: one launch per pass (csrc/attn.hip)
                m = attn_mask
                if key_padding_mask is not None:
                    kpm = key_padding_mask.unsqueeze(1)
                    m = kpm.expand(-1, q_len, -1) if m is None else torch.logical_or(m, kpm)
                if m is not None:
                    m = m.unsqueeze(0) if m.dim() == 2 else m
                    m = m if m.dtype in (torch.bool, torch.uint8) else m != 0
                lo = _plain_resblock(self.out_proj, self.embed_dim)
                rz = out_row_mask
                if rz is not None:
                    rz = rz.reshape(-1, rz.shape[-1])
                    rz = rz if rz.dtype in (torch.bool, torch.uint8) else rz != 0
                xq = q_len if _is_tail_view(query, key) else query
                if lo is not None:      # ... and the output ResBlock with the dead-row rule and the row mask
                    out, weights, keep = _AttnProjFn.apply(xq, key, m, rz, lq.weight, lq.bias, lk.weight, lk.bias,
                                                           lv.weight, lv.bias, lo.weight, lo.bias)
                else:
                    out, weights, keep = _AttnProjFn.apply(xq, key, m, None, lq.weight, lq.bias, lk.weight, lk.bias,
                                                           lv.weight, lv.bias)
                    out = self.out_proj(out)
                    if m is not None:
                        out = out * keep.unsqueeze(-1)
                    if rz is not None:
                        out = out * (~rz).to(out.dtype).unsqueeze(-1)
                return out.reshape(*lead, *out.shape[1:]), weights.reshape(*lead, *weights.shape[1:])

        fused_qkv = None
        if (self.pe is None or self.pe is False) and same_kv and _rows_proj_ok(self, query, key):
            qkv_params = [t for ll in (self.q_proj, self.k_proj, self.v_proj) for t in (_plain_linear(ll).weight, _plain_linear(ll).bias)]
            from asac_amd import native
            if (FUSED_MULTIHEAD and FUSED_QKV_IN_CORE and (self.num_heads > 1 or self.head_dim > 16)
                    and not (self.training and self.dropout > 0.)
                    and native.attention_mh_proj_supported(q_len, k_len, self.num_heads, self.head_dim)):
                fused_qkv = qkv_params      # windows of <= 16 positions: the projections run inside the core's forward launch
                q = k = v = key
            else:
                q, k, v = _QkvRowsFn.apply(key, q_len, *qkv_params)
        else:
            q, k, v = self.q_proj(query), self.k_proj(key), self.v_proj(value)
        if self.pe in (POSITIONAL_ENCODING.ROPE, POSITIONAL_ENCODING.ROPE2):
            q, k = self.rope(query_index, key_index, q, k)
        if (FUSED_MULTIHEAD and (self.num_heads > 1 or self.head_dim > 16) and q.is_cuda and q.dtype == torch.float32
                and not (self.training and self.dropout > 0.)):
            from asac_amd import native
            if native.attention_mh_supported(q_len, k_len, self.num_heads, self.head_dim):
                # several heads (or one wide head): scores, mask, softmax, weighted sum and the head average as one MFMA launch (csrc/attn_mh.hip)
                m = attn_mask
                if key_padding_mask is not None:
                    kpm = key_padding_mask.unsqueeze(1)
                    m = kpm.expand(-1, q_len, -1) if m is None else torch.logical_or(m, kpm)
                if m is not None:
                    m = m.unsqueeze(0) if m.dim() == 2 else m
                    m = m if m.dtype in (torch.bool, torch.uint8) else m != 0
                # the dead-row rule and the caller's padded rows as ONE factor formed by the launch itself (keep * ~row mask:
                # a product with `keep`, a `bitwise_not`, a cast and a second product were four launches per block and pass)
                rz = out_row_mask
                if rz is not None:
                    rz = rz.reshape(-1, rz.shape[-1])
                    rz = rz if rz.dtype in (torch.bool, torch.uint8) else rz != 0
                lo = _plain_resblock(self.out_proj, self.embed_dim) if FUSED_ROWS_PROJ else None
                y_pre = None
                block_done = False
                if fused_qkv is not None and lo is not None and FUSED_BLOCK_BACKWARD:
                    # ... and the output ResBlock behind it: the block is ONE launch forward and ONE backward
                    out, weights, keep, keep_rows = _AttnBlockFn.apply(key, q_len, *fused_qkv, lo.weight, lo.bias, m, self.num_heads,
                                                                       rz, rz is not None or m is not None)
                    block_done = True
                elif fused_qkv is not None and lo is not None:
                    out, weights, keep, keep_rows, *y_pre = _QkvAttnMhFn.apply(key, q_len, *fused_qkv, m, self.num_heads, rz,
                                                                               (lo.weight, lo.bias))
                elif fused_qkv is not None:
                    out, weights, keep, keep_rows = _QkvAttnMhFn.apply(key, q_len, *fused_qkv, m, self.num_heads, rz)
                else:
                    out, weights, keep, keep_rows = _AttnMhFn.apply(q, k, v, m, self.num_heads, rz)
                scale = keep_rows if rz is not None else (keep if m is not None else None)
                if block_done:
                    pass
                elif y_pre:
                    out = _OutResSavedFn.apply(out, lo.weight, lo.bias, scale, tuple(y_pre))
                elif lo is not None and native.rows_proj_supported(self.embed_dim):
                    # the output ResBlock and the dead-row / padded-row factor: one launch
                    out = _OutResRowsFn.apply(out, lo.weight, lo.bias, scale)
                else:
                    out = self.out_proj(out)
                    if scale is not None:
                        out = out * scale.unsqueeze(-1)
                return out.reshape(*lead, *out.shape[1:]), weights.reshape(*lead, *weights.shape[1:])
        q, k, v = self._split_heads(q), self._split_heads(k), self._split_heads(v)

        if key_padding_mask is not None:
            kpm = key_padding_mask.unsqueeze(1)                               # [bsz, 1, k]
            attn_mask = kpm.expand(-1, q_len, -1) if attn_mask is None else torch.logical_or(attn_mask, kpm)

        if _fused_core_ok(q, k, self.num_heads, self.training and self.dropout > 0.):
            # short windows, one head: scores, mask, softmax and the weighted sum as one launch (csrc/attn.hip)
            m = attn_mask
            if m is not None:
                m = m.unsqueeze(0) if m.dim() == 2 else m
                m = m if m.dtype in (torch.bool, torch.uint8) else m != 0
            out, weights, keep = _AttnCoreFn.apply(q, k, v, m)
            out = self.out_proj(out)
            if m is not None:      # fully masked queries produce zeros (the kernel already zeroed their weights)
                out = out * keep.unsqueeze(-1)
            if out_row_mask is not None:
                out = out * (~out_row_mask.reshape(-1, out_row_mask.shape[-1])).to(out.dtype).unsqueeze(-1)
            return out.reshape(*lead, *out.shape[1:]), weights.reshape(*lead, *weights.shape[1:])
        q = q / math.sqrt(self.head_dim)

        dead_rows = None
        if attn_mask is not None:
            if attn_mask.dim() == 2:
                attn_mask = attn_mask.unsqueeze(0).expand(bsz, -1, -1)
            dead_rows = attn_mask.all(dim=-1)                                 # [bsz, q]: nothing to attend to
            bias = torch.zeros(attn_mask.shape, dtype=query.dtype, device=query.device)
            bias = bias.masked_fill(attn_mask & ~dead_rows.unsqueeze(-1), float('-inf'))   # dead rows stay 0
            scores = torch.baddbmm(bias.repeat(self.num_heads, 1, 1), q, k.transpose(-2, -1))
        else:
            scores = torch.bmm(q, k.transpose(-2, -1))
        weights = torch.softmax(scores, dim=-1)                               # [bsz * heads, q, k]
        if self.training and self.dropout > 0.:
            weights = nn.functional.dropout(weights, p=self.dropout)
        out = torch.bmm(weights, v)                                           # [bsz * heads, q, head_dim]

        if self.num_heads > 1:
            weights = weights.view(self.num_heads, bsz, q_len, k_len).mean(0)
            out = out.view(self.num_heads, bsz, q_len, self.head_dim).permute(1, 2, 0, 3).reshape(bsz, q_len, self.embed_dim)
        out = self.out_proj(out)
        if dead_rows is not None:   # fully masked queries produce zeros, not NaN
            keep = ~dead_rows.unsqueeze(-1)
            out, weights = out * keep, weights * keep
        if out_row_mask is not None:
            out = out * (~out_row_mask.reshape(-1, out_row_mask.shape[-1])).to(out.dtype).unsqueeze(-1)
        return out.reshape(*lead, *out.shape[1:]), weights.reshape(*lead, *weights.shape[1:])


# ------------------------------------------------------------------------------------------------
def _kaiming_linear(n, bias):
    lin = nn.Linear(n, n, bias=bias)
    nn.init.kaiming_uniform_(lin.weight.data)
    return lin


class GatedResidualLayer(nn.Module):
    def forward(self, x, y):
        return x + y


class GatedOutputLayer(nn.Module):
    def __init__(self, embed_dim: int):
        super().__init__()
        self.embed_dim = embed_dim
        self.dense = _kaiming_linear(embed_dim, bias=False)

    def forward(self, x, y):
        return x + torch.sigmoid(self.dense(x) * y)


class GatedRecurrentLayer(nn.Module):
    """GRU-style gate (GTrXL): r, z gates from (x, y), candidate from (r*x, y)."""

    def __init__(self, embed_dim: int):
        super().__init__()
        self.embed_dim = embed_dim
        self.dense_x_r = _kaiming_linear(embed_dim, False)
        self.dense_y_r = _kaiming_linear(embed_dim, False)
        self.dense_x_z = _kaiming_linear(embed_dim, True)
        self.dense_y_z = _kaiming_linear(embed_dim, False)
        self.dense_x_g = _kaiming_linear(embed_dim, False)
        self.dense_y_g = _kaiming_linear(embed_dim, False)

    def forward(self, x, y):
        r = torch.sigmoid(self.dense_x_r(x) + self.dense_y_r(y))
        z = torch.sigmoid(self.dense_x_z(x) + self.dense_y_z(y))
        h = torch.tanh(self.dense_x_g(r * x) + self.dense_y_g(y))
        return (1 - z) * x + z * h


class GatedCatLayer(nn.Module):
    def forward(self, x, y):
        return torch.cat([x, y], dim=-1)


# ------------------------------------------------------------------------------------------------
class step_mask_cache:
    """`with step_mask_cache():` — inside, attention blocks reuse the (index / padding / attention) masks they
    built for identical inputs (same tensors, same lengths).  The learner wraps one train step in it: the online,
    the target and the post-update representation pass see the very same window buffers, and the buffers are only
    rewritten between steps (by kernels that do not bump torch's version counters — hence an explicit scope instead
    of version-keyed memoisation)."""
    active = None

    def __enter__(self):
        self._prev = step_mask_cache.active
        step_mask_cache.active = {}
        return self

    def __exit__(self, *exc):
        step_mask_cache.active = self._prev
        return False


_CAUSAL = {}


def _causal_mask(k: int, device):
    """[k, k] bool, True above the diagonal (never written to by its users)"""
    key = (k, str(device))
    m = _CAUSAL.get(key)
    if m is None:
        m = _CAUSAL[key] = torch.triu(torch.ones(k, k, dtype=torch.bool, device=device), diagonal=1)
    return m


def _tkey(t):
    return None if t is None else (t.data_ptr(), tuple(t.shape), tuple(t.stride()), t.dtype)


def _tail(x, n):
    """x[:, -n:] — x itself when that is all of it (no slice node: its backward is a fill and a copy)"""
    return x if x.shape[1] == n else x[:, -n:]


class EpisodeMultiheadAttentionBlock(nn.Module):
    def __init__(self, embed_dim: int, num_heads: int, pe=None, qkv_dense_depth: int = 0,
                 out_dense_depth: int = 1, dropout: float = 0., gate=None, use_layer_norm: bool = False):
        super().__init__()
        self.embed_dim, self.num_heads = embed_dim, num_heads
        self.gate, self.use_layer_norm = gate, use_layer_norm
        self.output_dim = embed_dim
        if use_layer_norm:
            self.layer_norm = nn.LayerNorm(embed_dim)
        self.attn = MultiheadAttention(embed_dim=embed_dim, num_heads=num_heads, pe=pe,
                                       qkv_dense_depth=qkv_dense_depth, out_dense_depth=out_dense_depth,
                                       dropout=dropout)
        if gate == GATE.RESIDUAL:
            self.gatedlayer = GatedResidualLayer()
        elif gate == GATE.OUTPUT:
            self.gatedlayer = GatedOutputLayer(embed_dim)
        elif gate == GATE.RECURRENT:
            self.gatedlayer = GatedRecurrentLayer(embed_dim)
        elif gate == GATE.CAT:
            self.gatedlayer = GatedCatLayer()
            self.output_dim = embed_dim * 2

    def get_attn_mask(self, seq_k_len: int, seq_q_len_only_attend_to_rest_key: int | None = None,
                      key_index=None, key_padding_mask=None, device='cpu'):
        """True = blocked.  Default: causal [k, k].  With `seq_q_len_only_attend_to_rest_key` = q the last
        q positions attend only to themselves and to the earlier ("rest") keys whose index is not
        in their future, and the rest keys only to themselves.  Padded keys are blocked everywhere."""
        if seq_q_len_only_attend_to_rest_key is None:
            mask = _causal_mask(seq_k_len, device)          # constant: built once per (length, device)
        else:
            q = seq_q_len_only_attend_to_rest_key
            rest = seq_k_len - q
            mask = torch.ones(seq_k_len, seq_k_len, dtype=torch.bool, device=device)
            mask[:rest, :rest] = torch.eye(rest, rest, dtype=torch.bool, device=device)
            mask[-q:, -q:] = torch.logical_or(mask[-q:, -q:], ~torch.eye(q, dtype=torch.bool, device=device))
            if key_index is not None:
                mask = mask.repeat(key_index.shape[0], 1, 1)
                q_idx = key_index[:, -q:].unsqueeze(-1)             # [batch, q, 1]
                rest_idx = key_index[:, :rest].unsqueeze(1)         # [batch, 1, rest]
                mask[:, -q:, :rest] = ~(q_idx >= rest_idx)
        if key_padding_mask is not None:      # (a 2-D mask broadcasts over the batch: same values as repeat + or)
            mask = torch.logical_or(mask if mask.dim() == 3 else mask.unsqueeze(0), key_padding_mask.unsqueeze(1))
        return mask

    def forward(self, key, seq_q_len: int, cut_query: bool = True, query_only_attend_to_rest_key: bool = False,
                key_index=None, key_padding_mask=None):
        """key [batch, k, E]; key_index / key_padding_mask may be SHORTER than k (they describe the
        newest positions; the older ones get index -1 / the first mask value)
        -> (output [batch, q or k, output_dim], weights [batch, q or k, k])"""
        seq_k_len = key.shape[1]
        residual_src = _tail(key, seq_q_len) if cut_query else key
        if self.use_layer_norm:
            key = self.layer_norm(key)

        cache = step_mask_cache.active
        ck = None if cache is None else ('masks', seq_k_len, seq_q_len, query_only_attend_to_rest_key,
                                         _tkey(key_index), _tkey(key_padding_mask), str(key.device))
        if ck is not None and ck in cache:
            key_index, key_padding_mask, attn_mask = cache[ck]
        else:
            if key_index is not None:
                short = seq_k_len - key_index.shape[1]
                assert short >= 0
                if not (query_only_attend_to_rest_key or self.attn.pe):
                    key_index = None        # read by the rest-key mask and the positional encodings only
                elif short:
                    key_index = torch.cat([key_index.new_full((key_index.shape[0], short), -1), key_index], dim=1)
            if key_padding_mask is not None:
                short = seq_k_len - key_padding_mask.shape[1]
                assert short >= 0
                if short:
                    key_padding_mask = torch.cat([key_padding_mask[:, :1].expand(-1, short), key_padding_mask], dim=1)
            attn_mask = self.get_attn_mask(seq_k_len, seq_q_len if query_only_attend_to_rest_key else None,
                                           key_index=key_index, key_padding_mask=key_padding_mask, device=key.device)
            if ck is not None:
                cache[ck] = (key_index, key_padding_mask, attn_mask)
        query_index = key_index
        query = key
        if cut_query:
            query = _tail(key, seq_q_len)
            if query_index is not None:
                query_index = query_index[:, -seq_q_len:]
            attn_mask = attn_mask[-seq_q_len:] if attn_mask.dim() == 2 else attn_mask[:, -seq_q_len:]

        # padded positions produce zeros: without a gate in between, the attention layer zeroes those rows itself
        # (inside its fused launch when it has one)
        row_mask = None
        if key_padding_mask is not None and self.gate is None:
            row_mask = key_padding_mask[:, -query.shape[1]:]
        output, weights = self.attn(query, key, key, query_index=query_index, key_index=key_index,
                                    attn_mask=attn_mask, out_row_mask=row_mask)
        if self.gate is not None:
            output = self.gatedlayer(residual_src, output)
            if key_padding_mask is not None:
                output = output * (~key_padding_mask[:, -output.shape[1]:]).to(output.dtype).unsqueeze(-1)
        return output, weights


class EpisodeMultiheadAttention(nn.Module):
    def __init__(self, embed_dim: int, num_layers: int = 2, num_heads=1, pe=False, qkv_dense_depth=0,
                 out_dense_depth=1, dropout=0., gate=None, use_layer_norm=False):
        super().__init__()
        self.num_layers = num_layers

        def per_layer(v):
            v = v if isinstance(v, list) else [v] * num_layers
            assert len(v) == num_layers
            return v

        num_heads, pe, qkv_dense_depth, out_dense_depth, dropout, gate, use_layer_norm = map(
            per_layer, (num_heads, pe, qkv_dense_depth, out_dense_depth, dropout, gate, use_layer_norm))

        self._attn_list = nn.ModuleList()
        dim = embed_dim
        for i in range(num_layers):
            block = EpisodeMultiheadAttentionBlock(dim, num_heads[i], pe=pe[i], qkv_dense_depth=qkv_dense_depth[i],
                                                   out_dense_depth=out_dense_depth[i], dropout=dropout[i],
                                                   gate=gate[i], use_layer_norm=use_layer_norm[i])
            self._attn_list.append(block)
            dim = block.output_dim
        self._output_dim_list = [b.output_dim for b in self._attn_list]
        self.output_dim = dim
        self.output_hidden_state_dim = sum(self._output_dim_list[:-1]) if num_layers > 1 else 1

    def forward(self, key, seq_q_len: int = 1, cut_query: bool = True, hidden_state=None,
                is_prev_hidden_state: bool = False, query_only_attend_to_rest_key: bool = False,
                key_index=None, key_padding_mask=None):
        """-> (encoded [batch, q or k, output_dim], next_hidden_state [batch, q, sum(dims[:-1])],
        list of per-layer attention weights)"""
        seq_k_len = key.shape[1]
        assert seq_q_len <= seq_k_len
        blocks, L = self._attn_list, self.num_layers
        kw = dict(query_only_attend_to_rest_key=query_only_attend_to_rest_key, key_index=key_index,
                  key_padding_mask=key_padding_mask)
        next_hidden, weights = [], []

        def run(block, k, cut):
            out, w = block(k, seq_q_len, cut_query=cut, **kw)
            weights.append(w)
            return out

        if hidden_state is None:
            k = key
            for block in blocks[:-1]:
                k = run(block, k, False)
                next_hidden.append(_tail(k, seq_q_len))
            out = run(blocks[-1], k, cut_query)
        else:
            states = hidden_state.split(self._output_dim_list[:-1], dim=-1) if L > 1 else ()
            if not is_prev_hidden_state:
                out = run(blocks[0], key, False if L > 1 else cut_query)
                for i, block in enumerate(blocks[1:]):
                    next_hidden.append(_tail(out, seq_q_len))
                    out = run(block, torch.cat([states[i], out], dim=1), False if i != L - 2 else cut_query)
            else:
                out = run(blocks[0], key, False)
                next_hidden.append(_tail(out, seq_q_len))
                if L == 1 and cut_query:
                    out = _tail(out, seq_q_len)
                for i, block in enumerate(blocks[1:-1]):
                    out = run(block, torch.cat([states[i], _tail(out, seq_k_len)], dim=1), False)
                    next_hidden.append(_tail(out, seq_q_len))
                if L > 1:
                    out = run(blocks[-1], torch.cat([states[-1], _tail(out, seq_k_len)], dim=1), cut_query)

        if L > 1:
            return out, (next_hidden[0] if len(next_hidden) == 1 else torch.cat(next_hidden, dim=-1)), weights
        return out, torch.zeros(key.shape[0], seq_q_len, 1, device=key.device), weights
