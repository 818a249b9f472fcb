"""GRU stacks of hidden size 32 / 64 / 128 with the time loops on MFMA (`asac_gru_wide_forward / _backward`).

The plugin layer `nn_models.layers.GRU` (reference `seq_layers.py:14-114`) routes here on the device for the hidden sizes the
reference's environments use (`m.GRU(…, 64, 1)` envs/square/memory_corridor/nn.py:19, `m.GRU(…, 128, 1)`
envs/uav/uav_hole/nn.py:22) — `csrc/gru.hip` stops at 16.  Per layer and pass: ONE library GEMM for the input projections of
all steps, ONE launch for the recurrence (MIOpen: one launch per step); backward: one launch for the recurrence through time,
three library GEMMs (dW_ih, dW_hh, dx) and two column sums (as products with a row of ones).  Same values as the cell loop of the module path (the padding rule
of the layer included: steps before a row's first unpadded one are skipped, padded outputs are zero).
"""
import os

import torch

from asac_amd import native

__all__ = ['fused_gru_wide', 'fused_gru_wide_supported']

ENABLED = os.environ.get('ASAC_GRU_WIDE', '1') != '0'      # (0: keep the module path — A/B runs)


def fused_gru_wide_supported(x: torch.Tensor, cells) -> bool:
    H = cells[0].hidden_size
    return (ENABLED and x.is_cuda and x.dtype == torch.float32 and native.gru_wide_supported(H)
            and all(c.hidden_size == H and c.bias and c.num_layers == 1 and not c.bidirectional for c in cells))


def _input_products(x2, w_ih, b_ih, out) -> None:
    """out [rows, 3H] = x2 w_ih^T + b_ih: a narrow input (observation ++ action) as one MFMA launch (`asac_rows_affine_forward`:
    the work is writing the result), a wide one (the layer below) as the library's GEMM"""
    from asac_amd import native
    K, N = x2.shape[1], w_ih.shape[0]
    if native.rows_affine_supported(K, N) and x2.stride(1) == 1 and out.is_contiguous():
        native.rows_affine_forward(x2, w_ih.contiguous(), b_ih.contiguous(), out)
    else:
        torch.addmm(b_ih, x2, w_ih.t(), out=out)


class _GruWideFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, h0, padding_mask, grad_mode, twin, *weights):
        """`twin`: None, or a list holding the target copy's cell weights (one layer): its recurrence runs in the same
        launch over the same windows and its (output, hn) are appended to the list for the caller to park"""
        B, L, _ = x.shape
        layers = len(weights) // 4
        H = weights[1].shape[1]
        ctx.set_materialize_grads(False)
        mask = None
        if padding_mask is not None:
            mask = padding_mask if padding_mask.dtype in (torch.bool, torch.uint8) else padding_mask != 0
            mask = mask.contiguous()
        if h0 is not None and h0.stride(2) != 1:
            h0 = h0.contiguous()
        need_grad = grad_mode and (any(ctx.needs_input_grad[i] for i in (0, 1)) or any(ctx.needs_input_grad[5:]))
        if twin is not None:
            # one layer, two networks: the input products side by side in one [2B, L, 3H] buffer, ONE recurrence launch
            w_ih, w_hh, b_ih, b_hh = (t.detach() for t in weights[:4])
            tw_ih, tw_hh, tb_ih, tb_hh = (t.detach() for t in twin[0])
            x2 = x.reshape(B * L, -1)
            gi2 = torch.empty(2 * B, L, 3 * H, dtype=x.dtype, device=x.device)
            _input_products(x2, w_ih, b_ih, gi2[:B].view(B * L, 3 * H))
            _input_products(x2, tw_ih, tb_ih, gi2[B:].view(B * L, 3 * H))
            hn2 = torch.empty(2 * B, L, 1, H, dtype=x.dtype, device=x.device)
            h_raw = torch.empty(B, L, H, dtype=x.dtype, device=x.device) if need_grad else None
            gates = torch.empty(B, L, 4 * H, dtype=x.dtype, device=x.device) if need_grad else None
            native.gru_wide_forward_twin(gi2, w_hh.contiguous(), b_hh.contiguous(), tw_hh.contiguous(), tb_hh.contiguous(),
                                         None if h0 is None else h0[:, 0], mask, hn2[:, :, 0], h_raw, gates)
            hn, t_hn = hn2[:B], hn2[B:]
            twin += [t_hn[:, :, 0], t_hn]
            if need_grad:
                ctx.layers, ctx.H = 1, H
                ctx.has_h0, ctx.has_mask = h0 is not None, mask is not None
                ctx.save_for_backward(x, h_raw, gates, *([h0] if h0 is not None else []), *([mask] if mask is not None else []),
                                      *weights)
            return hn[:, :, 0], hn
        hn = torch.empty(B, L, layers, H, dtype=x.dtype, device=x.device)
        saved, inp = [], x
        for l in range(layers):
            w_ih, w_hh, b_ih, b_hh = (t.detach() for t in weights[4 * l:4 * l + 4])
            gi = torch.empty(B, L, 3 * H, dtype=x.dtype, device=x.device)
            _input_products(inp.reshape(B * L, -1), w_ih, b_ih, gi.view(B * L, 3 * H))
            h_raw = torch.empty(B, L, H, dtype=x.dtype, device=x.device) if need_grad else None
            gates = torch.empty(B, L, 4 * H, dtype=x.dtype, device=x.device) if need_grad else None
            native.gru_wide_forward(gi, w_hh.contiguous(), b_hh.contiguous(), None if h0 is None else h0[:, l], mask,
                                    hn[:, :, l], h_raw, gates)
            if need_grad:
                saved += [inp, h_raw, gates]
            inp = hn[:, :, l]
        if need_grad:
            ctx.layers, ctx.H = layers, H
            ctx.has_h0, ctx.has_mask = h0 is not None, mask is not None
            ctx.save_for_backward(*saved, *([h0] if h0 is not None else []), *([mask] if mask is not None else []), *weights)
        return hn[:, :, layers - 1], hn

    @staticmethod
    def backward(ctx, grad_out, grad_hn):
        layers, H = ctx.layers, ctx.H
        if grad_out is None and grad_hn is None:
            return (None,) * (5 + 4 * layers)
        saved = list(ctx.saved_tensors)
        per_layer = [saved[3 * l:3 * l + 3] for l in range(layers)]
        rest = saved[3 * layers:]
        h0 = rest.pop(0) if ctx.has_h0 else None
        mask = rest.pop(0) if ctx.has_mask else None
        weights = rest
        x = per_layer[0][0]
        B, L = x.shape[:2]
        dev, dt = x.device, x.dtype
        g_w = [None] * (4 * layers)
        g_h0 = torch.zeros(B, layers, H, dtype=dt, device=dev) if (h0 is not None and ctx.needs_input_grad[1]) else None
        from_above = None        # d loss / d (this layer's masked output) coming from the layer above's input gradient
        for l in reversed(range(layers)):
            inp, h_raw, gates = per_layer[l]
            w_ih, w_hh = weights[4 * l].detach(), weights[4 * l + 1].detach()
            g = from_above
            if grad_hn is not None:
                g = grad_hn[:, :, l] if g is None else g + grad_hn[:, :, l]
            if l == layers - 1 and grad_out is not None:
                g = grad_out if g is None else g + grad_out
            if g is None:
                g = torch.zeros(B, L, H, dtype=dt, device=dev)
            if g.stride(2) != 1 or g.stride(0) % 4 or g.stride(1) % 4 or g.data_ptr() % 16:
                g = g.contiguous()
            dgi = torch.empty(B, L, 3 * H, dtype=dt, device=dev)
            dgh = torch.empty(B, L, 3 * H, dtype=dt, device=dev)
            dh0 = torch.empty(B, H, dtype=dt, device=dev) if g_h0 is not None else None
            native.gru_wide_backward(g, w_hh.t().contiguous(), gates, h_raw, None if h0 is None else h0[:, l], mask,
                                     dgi, dgh, dh0)
            if dh0 is not None:
                g_h0[:, l] = dh0
            dgi2, dgh2 = dgi.view(B * L, 3 * H), dgh.view(B * L, 3 * H)
            # h_{t-1} of every step: the state one step earlier (held at the initial state before a row's first step)
            h_first = (h0[:, l] if h0 is not None else torch.zeros(B, H, dtype=dt, device=dev)).unsqueeze(1)
            h_prev = torch.cat([h_first, h_raw[:, :-1]], dim=1).reshape(B * L, H)
            inp2 = inp.reshape(B * L, -1)
            if native.xty_supported(B * L, 3 * H, max(inp2.shape[1], H)):
                # weight and bias gradients as products over the B * L rows on MFMA, both in one launch pair (`asac_xty_multi`);
                # under the learner's direct mode added straight into the `.grad` views (no accumulation launches)
                from .fused_mlp import direct_enabled, direct_skips
                from .fused_rows_linear import queue_param_grads
                params = weights[4 * l:4 * l + 4]
                direct = (direct_enabled() and not direct_skips(*params)
                          and all(p_.requires_grad and p_.grad is not None and p_.grad.is_contiguous() for p_ in params))
                jobs = []
                if ctx.needs_input_grad[5 + 4 * l]:
                    xin = inp2 if inp2.stride(1) == 1 else inp2.contiguous()
                    if direct:
                        queue_param_grads(dgi2, xin, params[0].grad, params[2].grad)
                    else:
                        g_w[4 * l] = torch.empty(3 * H, inp2.shape[1], dtype=dt, device=dev)
                        g_w[4 * l + 2] = torch.empty(3 * H, dtype=dt, device=dev)
                        jobs.append((dgi2, xin, g_w[4 * l], g_w[4 * l + 2]))
                if ctx.needs_input_grad[5 + 4 * l + 1]:
                    if direct:
                        queue_param_grads(dgh2, h_prev, params[1].grad, params[3].grad)
                    else:
                        g_w[4 * l + 1] = torch.empty(3 * H, H, dtype=dt, device=dev)
                        g_w[4 * l + 3] = torch.empty(3 * H, dtype=dt, device=dev)
                        jobs.append((dgh2, h_prev, g_w[4 * l + 1], g_w[4 * l + 3]))
                if len(jobs) == 2:
                    native.xty_multi(jobs)
                elif jobs:
                    native.xty(*jobs[0])
            else:
                ones = torch.ones(1, B * L, dtype=dt, device=dev)      # column sums as library products (fixed order)
                if ctx.needs_input_grad[5 + 4 * l]:
                    g_w[4 * l] = dgi2.t() @ inp2
                    g_w[4 * l + 2] = (ones @ dgi2).view(-1)
                if ctx.needs_input_grad[5 + 4 * l + 1]:
                    g_w[4 * l + 1] = dgh2.t() @ h_prev
                    g_w[4 * l + 3] = (ones @ dgh2).view(-1)
            from_above = None
            if l > 0 or ctx.needs_input_grad[0]:
                from_above = (dgi2 @ w_ih).view(B, L, -1)
        g_x = from_above if ctx.needs_input_grad[0] else None
        return (g_x, g_h0, None, None, None, *g_w)


def fused_gru_wide(x, h0, padding_mask, cells, layer=None):
    """x [B, L, I]; h0 [B, layers, H] | None; padding_mask bool [B, L] | None -> (output [B, L, H], hn [B, L, layers, H]).
    `layer` (the GRU module itself): inside the learner's `TwinPass` the online module's one-layer recurrence runs its
    target namesake over the same windows in the same launch and parks the result (`fused_gru.TwinPass`)."""
    from .fused_gru import TwinPass, _cell_weights
    weights = _cell_weights(cells)
    if x.stride(2) != 1:
        x = x.contiguous()
    tp = TwinPass._active
    if tp is not None and layer is not None and len(cells) == 1:
        got = tp.claim(layer, x, h0, padding_mask,
                       lambda: _GruWideFn.apply(x, h0, padding_mask, torch.is_grad_enabled(), None, *weights))
        if got is not None:
            return got
        other = tp.wants(layer, x, h0)
        if (other is not None and len(other._grus) == 1 and other._fusable and x.shape[0] % 16 == 0
                and (other._grus[0].input_size, other._grus[0].hidden_size) == (cells[0].input_size, cells[0].hidden_size)):
            twin = [_cell_weights(other._grus)]
            res = _GruWideFn.apply(x, h0, padding_mask, torch.is_grad_enabled(), twin, *weights)
            tp.parked[id(other)] = (x, h0, padding_mask, twin[1], twin[2])
            return res
    return _GruWideFn.apply(x, h0, padding_mask, torch.is_grad_enabled(), None, *weights)
