"""One-launch GRU stack over a padded window (`asac_gru_forward` / `asac_gru_backward`).

The plugin layer `nn_models.layers.GRU` (reference `seq_layers.py:14-114`) routes here on the device
when the cell fits the kernel (input, hidden <= 16, layers <= 2, bias, no dropout) — the R2D2 burn-in
of `SAC_Base.get_l_states` (sac_base.py:1117-1146) then costs one launch per pass instead of one
MIOpen launch per time step and layer.

Autograd glue kept off the device: the function returns the top layer's output as its own dense
tensor (no select-backward zero-fill + copy), takes `h0` with any batch stride (no `.contiguous()` of the
sampled window's first hidden state), and — when the cell parameters' `.grad` tensors already exist as
dense buffers (the learner's flat gradient buffer) — adds the parameter gradients into them from the
reduction kernel itself instead of handing eight tensors to eight `AccumulateGrad` add kernels.
"""
import torch

from asac_amd import native

from .fused_mlp import direct_enabled

__all__ = ['fused_gru', 'fused_gru_supported', 'TwinPass']

# accumulate parameter gradients into existing dense `.grad` tensors from the kernel (see above), inside the
# learner's `fused_mlp.direct_param_grads()` regions
DIRECT_PARAM_GRADS = True


def fused_gru_supported(x: torch.Tensor, input_size: int, hidden: int, layers: int) -> bool:
    return (x.is_cuda and x.dtype == torch.float32
            and native.gru_supported(input_size, hidden, layers))


class TwinPass:
    """The learner evaluates the online representation and then the target representation on the very same
    window (reference sac_base.py:2066-2079).  Inside `with twin:` a fused GRU layer of the online module whose
    input does not depend on trainable parameters also runs its namesake in the target module over the same
    input in the same launch (`asac_gru_forward_twin`) and parks the result; the target module's layer
    then picks it up instead of launching again.

    Nothing tells the layer that the target module will feed its GRU the same values, so the shortcut has to
    earn trust first: during the first `verify_steps` eager passes the target layer still computes its own
    result and compares input and output bit for bit with the parked ones (a host sync — which is why this
    only happens outside graph capture).  One mismatch switches the shortcut off for good."""

    _active = None

    def __init__(self, online_root, target_root, verify_steps=2):
        from algorithm.nn_models.layers.seq_layers import GRU
        tg = dict(target_root.named_modules())
        self.partner = {id(m): tg[name] for name, m in online_root.named_modules()
                        if isinstance(m, GRU) and isinstance(tg.get(name), GRU)}
        self.verify_steps = verify_steps
        self.verified, self.failed = 0, False
        self.parked = {}

    @property
    def trusted(self):
        return not self.failed and self.verified >= self.verify_steps

    def __bool__(self):
        return bool(self.partner) and not self.failed

    def __enter__(self):
        self.parked.clear()
        self._checked = self._ok = 0
        TwinPass._active = self
        return self

    def __exit__(self, *exc):
        TwinPass._active = None
        self.parked.clear()
        if not self.trusted and self._checked:
            if self._ok == self._checked:
                self.verified += 1
            else:
                self.failed = True
        return False

    # the online layer's side -------------------------------------------------------------------
    def wants(self, layer, x, h0):
        """-> the target layer to run beside `layer`, or None"""
        if self.failed or x.requires_grad or (h0 is not None and h0.requires_grad):
            return None
        if not self.trusted and torch.cuda.is_current_stream_capturing():
            return None
        return self.partner.get(id(layer))

    # the target layer's side -------------------------------------------------------------------
    @staticmethod
    def _same_view(a, b):
        if a is None or b is None:
            return a is b
        return a.data_ptr() == b.data_ptr() and a.shape == b.shape and a.stride() == b.stride() and a.dtype == b.dtype

    def claim(self, layer, x, h0, mask, compute):
        """-> (out, hn) parked for `layer`, or None.  `compute()` evaluates the layer on its own (verification)."""
        hit = self.parked.pop(id(layer), None)
        if hit is None or torch.is_grad_enabled() and any(p.requires_grad for p in layer.parameters()):
            return None
        px, ph0, pmask, out, hn = hit
        if x.shape != px.shape or not self._same_view(h0, ph0) or not self._same_view(mask, pmask):
            return None
        if self.trusted:
            return out, hn
        own = compute()
        self._checked += 1
        self._ok += int(torch.equal(x, px) and torch.equal(own[0], out) and torch.equal(own[1], hn))
        return own


# `backward_from_position`: [the _GruFn node, members [E, B, H], position, adam epilogue | None, epilogue ran] while
# that node's backward is the next to run
_TOP_AT = None


def is_fused_top(t: torch.Tensor) -> bool:
    """is `t` the top-layer output of a fused GRU launch itself (no module stands between the two)?"""
    fn = t.grad_fn
    return fn is not None and type(fn).__name__ == '_GruFnBackward' and t.output_nr == 0


def backward_from_position(top, members, position, placeholder, adam=None) -> bool:
    """Back-propagate d loss / d top[:, position] = sum_e members[e] (zero at every other position) from the fused
    GRU's own output `top` (`is_fused_top`): the launch sums the members itself and starts its recursion at the position
    (`asac_gru_backward_at`) — no member-sum launch, no [B, L, H] gradient read.  `placeholder`: any [B, L, H] tensor
    (autograd wants a gradient object of the output's shape; its values are not read).  `adam`
    (`native.adam_epilogue` over the learner's flat buffers): the launch that finishes the cell parameters' gradients
    also takes their optimizer step, when those gradients are written in place -> True (else the caller still has
    to step them)."""
    global _TOP_AT
    assert is_fused_top(top) and members.shape[1:] == (top.shape[0], top.shape[2])
    state = _TOP_AT = [top.grad_fn, members.contiguous(), int(position) % top.shape[1], adam, False]
    try:
        torch.autograd.backward([top], [placeholder])
    finally:
        _TOP_AT = None
    return state[4]


class _GruFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, h0, padding_mask, desc, twin, *weights):
        # `desc` may come as (desc, the caller's grad mode): inside `forward` the mode is always off, and
        # `needs_input_grad` mirrors `requires_grad` whatever the mode — a no-grad pass over trainable cells must not
        # save its gate activations
        desc, grad_mode = desc if isinstance(desc, tuple) else (desc, True)
        B, L, _ = x.shape
        H, layers = desc.hidden, desc.layers
        ctx.set_materialize_grads(False)      # an unused output (usually hn) arrives as None, not as a zero-fill
        if x.stride(2) != 1:
            x = x.contiguous()
        if h0 is not None and not (h0.stride(2) == 1 and h0.stride(1) == H):
            h0 = h0.contiguous()
        mask = None
        if padding_mask is not None:
            mask = padding_mask if padding_mask.dtype in (torch.bool, torch.uint8) else padding_mask != 0
            if mask.stride(1) != 1:
                mask = mask.contiguous()
        w = [tuple(t.detach() for t in weights[4 * l:4 * l + 4]) for l in range(layers)]
        need_grad = grad_mode and (any(ctx.needs_input_grad[i] for i in (0, 1)) or any(ctx.needs_input_grad[5:]))
        hn = torch.empty(B, L, layers, H, dtype=x.dtype, device=x.device)
        out = torch.empty(B, L, H, dtype=x.dtype, device=x.device)
        gates = torch.empty(B, L, layers, 5 * H, dtype=x.dtype, device=x.device) if need_grad else None
        if twin is None:
            native.gru_forward(desc, w, x, h0, mask, hn, out, gates)
        else:       # twin = [the target copy's weights]; its outputs are appended for the caller to park
            t_hn, t_out = torch.empty_like(hn), torch.empty_like(out)
            tw = [tuple(t.detach() for t in twin[0][4 * l:4 * l + 4]) for l in range(layers)]
            native.gru_forward_twin(desc, w, tw, x, h0, mask, hn, out, gates, t_hn, t_out)
            twin += [t_out, t_hn]
        if need_grad:
            ctx.desc = desc
            ctx.has_h0, ctx.has_mask = h0 is not None, mask is not None
            ctx.save_for_backward(x, hn, gates, *([h0] if h0 is not None else []),
                                  *([mask] if mask is not None else []), *weights)
        return out, hn

    @staticmethod
    def backward(ctx, grad_out, grad_hn):
        desc = ctx.desc
        if grad_out is None and grad_hn is None:
            return (None,) * (5 + 4 * desc.layers)
        saved = list(ctx.saved_tensors)
        x, hn, gates = saved[:3]
        rest = saved[3:]
        h0 = rest.pop(0) if ctx.has_h0 else None
        mask = rest.pop(0) if ctx.has_mask else None
        weights = rest
        layers = desc.layers
        w = [tuple(t.detach() for t in weights[4 * l:4 * l + 4]) for l in range(layers)]
        B = x.shape[0]
        g_x = torch.empty(x.shape, dtype=x.dtype, device=x.device) if ctx.needs_input_grad[0] else None
        g_h0 = torch.empty((B, layers, desc.hidden), dtype=x.dtype, device=x.device) \
            if (h0 is not None and ctx.needs_input_grad[1]) else None
        ws = torch.empty(native.gru_backward_workspace(desc, B), dtype=x.dtype, device=x.device)
        at = _TOP_AT if (_TOP_AT is not None and _TOP_AT[0] is ctx and grad_hn is None) else None
        assert _TOP_AT is None or at is not None, 'backward_from_position: another node ran first'
        grad_out = None if grad_out is None else grad_out.contiguous()
        grad_hn = None if grad_hn is None else grad_hn.contiguous()

        def launch(g_params, gt, accumulate):
            if at is not None:
                adam = at[3] if gt is not None else None
                native.gru_backward_at(desc, w, x, h0, mask, hn, gates, at[1], at[2], g_x, g_h0, g_params, gt,
                                       accumulate, ws, adam=adam)
                at[4] = adam is not None
            else:
                native.gru_backward(desc, w, x, h0, mask, hn, gates, grad_hn, grad_out, g_x, g_h0, g_params, gt,
                                    accumulate, ws)
        direct = DIRECT_PARAM_GRADS and direct_enabled() and all(
            t.grad is not None and t.grad.is_contiguous() and t.grad.dtype == torch.float32 and t.grad.is_cuda
            for t in weights)
        if direct:
            gt = [tuple(t.grad for t in weights[4 * l:4 * l + 4]) for l in range(layers)]
            launch(None, gt, True)
            return (g_x, g_h0, None, None, None, *([None] * len(weights)))
        g_params = torch.empty(native.gru_param_count(desc), dtype=x.dtype, device=x.device)
        launch(g_params, None, False)
        g_w, off = [], 0
        for t in weights:
            k = t.numel()
            g_w.append(g_params[off:off + k].view(t.shape))
            off += k
        return (g_x, g_h0, None, None, None, *g_w)


def _cell_weights(cells):
    weights = []
    for c in cells:
        weights += [c.weight_ih_l0, c.weight_hh_l0, c.bias_ih_l0, c.bias_hh_l0]
    return weights


def fused_gru(x, h0, padding_mask, cells, layer=None):
    """x [B, L, I]; h0 [B, layers, H] | None; padding_mask bool [B, L] | None; `cells` = the layer's
    single-layer nn.GRU modules (`layer` itself, for the twin pass).  Returns (output [B, L, H] = the top layer,
    hn [B, L, layers, H])."""
    layers = len(cells)
    desc = native.gru_desc(cells[0].input_size, cells[0].hidden_size, layers)
    weights = _cell_weights(cells)
    tp = TwinPass._active
    if tp is not None and layer is not None:
        got = tp.claim(layer, x, h0, padding_mask, lambda: _GruFn.apply(x, h0, padding_mask, (desc, torch.is_grad_enabled()), None, *weights))
        if got is not None:
            return got
        other = tp.wants(layer, x, h0)
        if other is not None and len(other._grus) == layers and other._fusable \
                and (other._grus[0].input_size, other._grus[0].hidden_size) == (desc.input, desc.hidden):
            twin = [_cell_weights(other._grus)]
            res = _GruFn.apply(x, h0, padding_mask, (desc, torch.is_grad_enabled()), twin, *weights)
            tp.parked[id(other)] = (x, h0, padding_mask, twin[1], twin[2])
            return res
    return _GruFn.apply(x, h0, padding_mask, (desc, torch.is_grad_enabled()), None, *weights)
