"""Stock-network fast path: when the Q ensemble / policy are the reference's stock `ModelQ` /
`ModelPolicy` compositions of `LinearLayers` (continuous head only), a whole forward or backward
pass of the network — or of all E ensemble members at once — is ONE `asac_mlp_*` launch
(`csrc/mlp.hip`, MFMA f32) on the parameters where they already live in the flat buffer.
Anything else (user-defined models, discrete heads, other activations, widths > 64) keeps the
generic module path; `describe_*` returns None and the caller falls back.
"""
import contextlib
import os
import weakref

import torch
from torch import nn

from asac_amd import native

from .nn_models.layers.linear_layers import LinearLayers, ResBlock

__all__ = ['StockMLP', 'describe_q', 'describe_policy', 'describe_dense', 'fused_dense', 'gauss_head']

MAX_WIDTH, MAX_HEAD = 64, 16
MAX_INPUT = 128         # a first layer may be up to 128 inputs wide when the stack has <= 3 blocks (two K halves)
FUSED_DENSE = True      # `LinearLayers` stacks that opt in (`fuse = True`: the conv encoders' heads) run fused

# How the fused autograd Functions (this module, fused_conv, fused_gru, fused_linear, layers.seq_layers) deliver
# PARAMETER gradients.  Default: they are returned to autograd like any op's, so `autograd.grad`, `backward(inputs=
# ...)` and gradient gating (`sac_aux.calculate_adaptive_weights`) see exactly what PyTorch semantics promise.
# Inside `direct_param_grads()` the backward kernels instead ADD them straight into the parameters' `.grad` views of
# the learner's flat gradient buffer (no per-parameter AccumulateGrad launches) and hand autograd nothing: only valid
# around a backward pass that is meant to accumulate into every parameter it reaches — the learner wraps exactly
# those (the Q loss, the policy objective, the curiosity loss).  A plain global: the autograd engine runs backward
# nodes on its own device thread while the caller blocks inside the `with`.
# `only`: the backward is restricted to these tensors (`backward(inputs=...)`): `ctx.needs_input_grad` is fixed at
# forward time and does not know about that restriction, so a direct-mode backward consults `direct_skips` and leaves
# the parameters outside the set alone (no launch, no accumulation) — what autograd does with their gradients anyway.
_direct_depth = 0
_direct_only = []


@contextlib.contextmanager
def direct_param_grads(only=None):
    global _direct_depth
    if _direct_depth == 0:
        # (a backward pass that raised may have left queued products and an armed end-of-backward callback behind)
        from .fused_rows_linear import reset_queue
        reset_queue()
    _direct_depth += 1
    _direct_only.append(None if only is None else {id(t) for t in only})
    try:
        yield
    finally:
        _direct_only.pop()
        _direct_depth -= 1


def direct_enabled() -> bool:
    return _direct_depth > 0


def direct_skips(*params) -> bool:
    """inside a restricted direct-mode backward: none of `params` is among the tensors the backward may reach"""
    only = _direct_only[-1] if _direct_only else None
    return only is not None and not any(id(p) in only for p in params)


def _blocks_of(ll: LinearLayers):
    """-> (list of ResBlock, final nn.Linear | None) or None when the stack is not fusable"""
    blocks, final = [], None
    for m in ll.dense:
        if isinstance(m, ResBlock):
            if final is not None or not isinstance(m.act, nn.GELU) or getattr(m.act, 'approximate', 'none') != 'none':
                return None
            blocks.append(m)
        elif isinstance(m, nn.Dropout):
            if m.p != 0:
                return None
        elif isinstance(m, nn.Linear):
            if final is not None:
                return None
            final = m
        else:
            return None
    return blocks, final


def _is_identity(ll) -> bool:
    return isinstance(ll, LinearLayers) and len(ll.dense) == 0


def _fill_blocks(desc, blocks, offsets, in_width):
    if not 1 <= len(blocks) <= 4:
        return None
    prev = in_width
    for l, b in enumerate(blocks):
        w = b.linear.out_features
        if b.linear.in_features != prev or w > MAX_WIDTH or b.linear.bias is None:
            return None
        desc.width[l], desc.residual[l] = w, int(b.residual)
        desc.w_off[l], desc.b_off[l] = offsets[id(b.linear.weight)], offsets[id(b.linear.bias)]
        prev = w
    desc.n_blocks = len(blocks)
    return prev


def _offsets(module: nn.Module) -> dict:
    off, table = 0, {}
    for p in module.parameters():
        table[id(p)] = off
        off += p.numel()
    return table


def describe_q(q) -> 'native.MlpDesc | None':
    """Stock continuous-action Q: [state | action] -> c_dense blocks -> Linear(., 1)."""
    from .nn_models.q import ModelQ
    if type(q) is not ModelQ or q.d_action_sizes or not q.c_action_size:
        return None
    if not (_is_identity(q.dense) and _is_identity(q.c_state_dense) and _is_identity(q.c_action_dense)):
        return None
    parsed = _blocks_of(q.c_dense)
    if parsed is None or parsed[1] is None or parsed[1].out_features != 1:
        return None
    blocks, final = parsed
    desc, offs = native.MlpDesc(), _offsets(q)
    desc.in0, desc.in1 = q.state_size, q.c_action_size
    k0 = desc.in0 + desc.in1
    # (a first layer of up to 128 inputs — critics on a 64-wide state + the action — with <= 3 blocks, the first not residual:
    # the wide instantiations of the forward / backward kernels, one launch per network; the one-launch chains need <= 64)
    if k0 > MAX_INPUT or (k0 > MAX_WIDTH and (len(blocks) > 3 or blocks[0].residual)):
        return None
    if _fill_blocks(desc, blocks, offs, k0) is None:
        return None
    desc.head_cols[0], desc.head_cols[1] = 1, 0
    desc.head_w_off[0], desc.head_b_off[0] = offs[id(final.weight)], offs[id(final.bias)]
    return desc


def _tail_params(ll, skip: int) -> list:
    """the parameters of `ll` behind its first `skip` ResBlocks, in `parameters()` order"""
    if skip == 0:
        return list(ll.parameters())
    parsed = _blocks_of(ll)
    blocks, final = parsed
    mods = [b.linear for b in blocks[skip:]] + ([final] if final is not None else [])
    return [p for m in mods for p in m.parameters()]


def describe_dense(ll, skip: int = 0) -> 'native.MlpDesc | None':
    """A `LinearLayers` stack on its own: input -> ResBlocks -> output Linear (<= 16 columns), offsets relative
    to the stack's own parameters in `parameters()` order.  `skip`: describe the stack BEHIND its first `skip` ResBlocks
    (a wide first layer runs on its own launches, csrc/wide.hip): offsets relative to the first parameter behind them."""
    if not isinstance(ll, LinearLayers):
        return None
    parsed = _blocks_of(ll)
    if parsed is None or parsed[1] is None or len(parsed[0]) <= skip:
        return None
    blocks, final = parsed
    in0 = ll.input_size if skip == 0 else blocks[skip - 1].linear.out_features
    blocks = blocks[skip:]
    if final.bias is None or final.out_features > MAX_HEAD or in0 > MAX_INPUT:
        return None
    if in0 > MAX_WIDTH and (len(blocks) > 3 or blocks[0].residual):
        return None
    desc = native.MlpDesc()
    offs, off = {}, 0
    for p in _tail_params(ll, skip):
        offs[id(p)] = off
        off += p.numel()
    desc.in0, desc.in1 = in0, 0
    if _fill_blocks(desc, blocks, offs, in0) is None:
        return None
    desc.head_cols[0], desc.head_cols[1] = final.out_features, 0
    desc.head_w_off[0], desc.head_b_off[0] = offs[id(final.weight)], offs[id(final.bias)]
    return desc


def _flat_alias(tensors):
    """-> a 1-D tensor aliasing `tensors` if they sit back to back, in order, in one storage (the learner's flat
    parameter / gradient buffers), else None"""
    t0 = tensors[0]
    base, pos = t0.untyped_storage().data_ptr(), t0.storage_offset()
    first = pos
    for t in tensors:
        if (t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous()
                or t.untyped_storage().data_ptr() != base or t.storage_offset() != pos):
            return None
        pos += t.numel()
    return torch.empty(0, dtype=torch.float32, device=t0.device).set_(t0.untyped_storage(), first, (pos - first,))


_DENSE_LAUNCHERS = weakref.WeakKeyDictionary()


def fused_dense(ll, x, skip: int = 0):
    """`ll(x)` for a `LinearLayers` stack as ONE launch per pass on the parameters where they live, when the stack
    fits `describe_dense`, its parameters (and gradients, when it trains) are consecutive views of flat buffers
    — true for every module of a `SAC_Base` — and `x` is f32 on the device.  Returns None otherwise (the caller
    keeps the module path).  Parameter gradients: see `direct_param_grads`.  `skip`: `x` is the output of the stack's
    first `skip` ResBlocks (see `fused_dense_wide_first`), the launch covers the rest."""
    in0 = ll.input_size
    if skip:
        parsed = _blocks_of(ll)
        if parsed is None or len(parsed[0]) <= skip:
            return None
        in0 = parsed[0][skip - 1].linear.out_features
    if not (FUSED_DENSE and x.is_cuda and x.dtype == torch.float32 and x.shape[-1] == in0):
        return None
    params = _tail_params(ll, skip)
    if not params:
        return None
    train = torch.is_grad_enabled() and any(p.requires_grad for p in params)
    if train and not all(p.requires_grad and p.grad is not None for p in params):
        return None
    # one launcher per (parameter buffer, gradient buffer | inference): a stack alternates between its training
    # pass and no-grad passes within a step
    key = (params[0].data_ptr(), params[0].grad.data_ptr() if train else 0, x.device, skip)
    cache = _DENSE_LAUNCHERS.setdefault(ll, {})     # kept off the module: not copied / pickled with it
    if key not in cache:
        if len(cache) > 8:
            cache.clear()
        desc = describe_dense(ll, skip)
        flat = _flat_alias([p.data for p in params]) if desc is not None else None
        gflat = _flat_alias([p.grad for p in params]) if (flat is not None and train) else None
        ok = flat is not None and (gflat is not None or not train) and flat.data_ptr() % 16 == 0
        cache[key] = StockMLP(desc, flat, gflat, 0, flat.numel(), 1, x.device, params) if ok else None
    mlp = cache[key]
    if mlp is None:
        return None
    rows = x.reshape(-1, in0)
    if not rows.is_contiguous():
        rows = rows.contiguous()
    out = mlp(rows, None, param_grads=train)       # [1, rows, cols]
    return out.view(*x.shape[:-1], mlp.out_cols)


def fused_dense_wide_first(ll, x):
    """`ll(x)` for a stack whose FIRST ResBlock has a wide input (more than 128 features: the flattened convolution map in
    front of `ConvLayers.dense`, reference image_layers.py:188-227) — that block on the launches of csrc/wide.hip
    (`fused_rows_linear.rows_resblock`), everything behind it as one fused dense-stack launch per pass; None where the
    wide launches do not apply (the caller keeps the module path)."""
    parsed = _blocks_of(ll)
    if parsed is None or not parsed[0] or parsed[0][0].linear.in_features <= MAX_INPUT or parsed[0][0].residual:
        return None
    from .fused_rows_linear import rows_resblock
    first = parsed[0][0]
    y = rows_resblock(first, x)
    if y is None:
        return None
    out = fused_dense(ll, y, skip=1)
    if out is not None:
        return out
    mods = list(ll.dense)
    for mod in mods[mods.index(first) + 1:]:
        y = mod(y)
    return y


def describe_policy(pi) -> 'native.MlpDesc | None':
    """Stock continuous policy: state -> c_dense blocks -> (mean Linear | logstd Linear)."""
    from .nn_models.policy import ModelPolicy
    if type(pi) is not ModelPolicy or pi.d_action_sizes or not pi.c_action_size:
        return None
    if not _is_identity(pi.dense):
        return None
    trunk, mean, logstd = _blocks_of(pi.c_dense), _blocks_of(pi.mean_dense), _blocks_of(pi.logstd_dense)
    if trunk is None or trunk[1] is not None or mean is None or logstd is None:
        return None
    if mean[0] or logstd[0] or mean[1] is None or logstd[1] is None:
        return None
    A = pi.c_action_size
    if 2 * A > MAX_HEAD or pi.state_size > MAX_WIDTH:
        return None
    desc, offs = native.MlpDesc(), _offsets(pi)
    desc.in0, desc.in1 = pi.state_size, 0
    if _fill_blocks(desc, trunk[0], offs, pi.state_size) is None:
        return None
    desc.head_cols[0], desc.head_cols[1] = A, A
    desc.head_w_off[0], desc.head_b_off[0] = offs[id(mean[1].weight)], offs[id(mean[1].bias)]
    desc.head_w_off[1], desc.head_b_off[1] = offs[id(logstd[1].weight)], offs[id(logstd[1].bias)]
    desc.head_transform = 1     # outputs are (loc | scale) of the policy's Normal, not (mean | logstd)
    return desc


class DeferredPartialSums:
    """While active, the backward launches below (and the attention blocks', seq_layers._AttnProjFn) that would hand
    parameter gradients BACK to autograd — `autograd.grad` walks, not the learner's direct mode — return None for them and
    leave the second launch of each (the per-workgroup partials summed in workgroup order) to `flush()`, which runs up to
    sixteen of them as ONE launch (`native.sum_partials_multi`; per job the order, hence the bits, of the launch it stands
    for) and returns {walk: {id(parameter): gradient}} for the caller to put where autograd left None — like
    fused_conv.DeferredConvBackward, and for the same walks: those of `calculate_adaptive_weights` / `_train_rpm` (reference
    sac_base.py:1607-1631, 1798-1839), which only collect their parameter gradients.  `walk`: set by the caller in front of
    each `autograd.grad`.  A parameter other nodes of the graph use as well still gets those nodes' contributions from
    autograd; the caller adds."""
    _active = None

    def __init__(self):
        self.jobs, self.grads, self.walk, self._outer = [], {}, 0, None

    @classmethod
    def active(cls):
        return cls._active if SUM_PARTIALS_LATER else None

    def __enter__(self):
        self._outer, DeferredPartialSums._active = DeferredPartialSums._active, self
        return self

    def __exit__(self, exc_type, exc, tb):
        DeferredPartialSums._active = self._outer
        return False

    def add(self, partial, slabs, slices, slab_stride, n, out, accumulate=False):
        self.jobs.append((partial, slabs, slices, slab_stride, n, out, bool(accumulate)))

    def record(self, params, grads):
        """the gradients (views of buffers `add`ed above) this walk owes `params`"""
        mine = self.grads.setdefault(self.walk, {})
        for p, g in zip(params, grads):
            if g is not None:
                mine.setdefault(id(p), []).append(g)

    def flush(self):
        """-> {walk: {id(parameter): gradient}}; the sums run here (one launch per sixteen recorded backwards)"""
        jobs, self.jobs = self.jobs, []
        for k in range(0, len(jobs), native.SUM_PARTIALS_MAX_JOBS):
            native.sum_partials_multi(jobs[k:k + native.SUM_PARTIALS_MAX_JOBS])
        taken, self.grads = self.grads, {}
        # (a parameter two recorded nodes of one walk share: their gradients added, as autograd would have)
        return {walk: {pid: gl[0] if len(gl) == 1 else sum(gl[1:], gl[0]) for pid, gl in mine.items()}
                for walk, mine in taken.items()}


SUM_PARTIALS_LATER = os.environ.get('ASAC_SUM_PARTIALS_LATER', '1') != '0'    # (A/B switch)


def _param_grad_target(mlp, param_grads, n_params):
    """-> (kernel target | None, gradients to return to autograd): the flat `.grad` views in direct mode, else a
    scratch buffer with the parameters' layout, returned as one view per parameter"""
    if not param_grads or direct_enabled() or n_params == 0:
        return None, [None] * n_params
    scratch = torch.empty_like(mlp.grad_params)      # (every parameter's span is overwritten: MLP_REDUCE_OVERWRITE)
    base = mlp.params.storage_offset()
    views = [scratch[p.data.storage_offset() - base:p.data.storage_offset() - base + p.numel()].view(p.shape)
             for p in mlp.param_tensors]
    return scratch, views


class _MlpFn(torch.autograd.Function):
    """inputs: (anchor, x0, x1, mlp, param_grads, *the network's parameters) — the parameters are inputs so that the
    node is part of every backward / `autograd.grad` that asks for them"""

    @staticmethod
    def forward(ctx, anchor, x0, x1, mlp, param_grads, *params):
        out = mlp._launch_forward(x0, x1)
        ctx.mlp, ctx.param_grads, ctx.n_params = mlp, param_grads, len(params)
        ctx.save_for_backward(x0, x1 if x1 is not None else x0.new_empty(0))
        ctx.has_x1 = x1 is not None
        return out

    @staticmethod
    def backward(ctx, grad_out):
        x0, x1 = ctx.saved_tensors
        x1 = x1 if ctx.has_x1 else None
        mlp = ctx.mlp
        need0, need1 = ctx.needs_input_grad[1], ctx.has_x1 and ctx.needs_input_grad[2]
        target, pg = _param_grad_target(mlp, ctx.param_grads, ctx.n_params)
        later = DeferredPartialSums.active() if ctx.param_grads else None
        g0, g1 = mlp._launch_backward(x0, x1, grad_out.contiguous(), need0, need1, ctx.param_grads,
                                      grad_target=target, later=later)
        if later is not None and target is not None:       # (the gradients reach the caller through `later.flush()`)
            later.record(mlp.param_tensors, pg)
            pg = [None] * ctx.n_params
        return (None, g0, g1, None, None, *pg)


class _MlpSelectFn(torch.autograd.Function):
    """`mlp(base[:, t], x1)` for a [B, L, in0] window `base`: the rows are read in place (row stride L*in0)
    and the backward hands autograd ONE zero-initialised gradient of `base` with the member-summed input
    gradient reduced straight into its [:, t] slice — instead of sum + select-backward + slice-backward
    (five tiny launches)."""

    @staticmethod
    def forward(ctx, anchor, base, t, x1, mlp, param_grads, *params):
        x0 = base[:, t]
        out = mlp._launch_forward(x0, x1)
        ctx.mlp, ctx.param_grads, ctx.t, ctx.n_params = mlp, param_grads, t, len(params)
        ctx.save_for_backward(base, x1 if x1 is not None else base.new_empty(0))
        ctx.has_x1 = x1 is not None
        return out

    @staticmethod
    def backward(ctx, grad_out):
        base, x1 = ctx.saved_tensors
        x1 = x1 if ctx.has_x1 else None
        mlp, t = ctx.mlp, ctx.t
        need0, need1 = ctx.needs_input_grad[1], ctx.has_x1 and ctx.needs_input_grad[3]
        target, pg = _param_grad_target(mlp, ctx.param_grads, ctx.n_params)
        later = DeferredPartialSums.active() if ctx.param_grads else None
        g0, g1 = mlp._launch_backward(base[:, t], x1, grad_out.contiguous(), need0, need1, ctx.param_grads,
                                      reduce_members=False, grad_target=target, later=later)
        if later is not None and target is not None:       # (the gradients reach the caller through `later.flush()`)
            later.record(mlp.param_tensors, pg)
            pg = [None] * ctx.n_params
        g_base = None
        if g0 is not None:
            # (a window gradient that is zero outside position t: the zeros are kept between calls — only the slice is written,
            # no fill launch; the buffer is never handed out exclusively, so autograd does not accumulate into it in place)
            key = (tuple(base.shape), t % base.shape[1], base.device)
            g_base = mlp._window_grads.get(key)
            if g_base is None:
                if len(mlp._window_grads) > 4:
                    mlp._window_grads.clear()
                g_base = mlp._window_grads[key] = torch.zeros_like(base)
            torch.sum(g0, dim=0, out=g_base[:, t])
        if g1 is not None and x1.dim() == 2:
            g1 = g1.sum(0) if mlp.E > 1 else g1[0]
        return (None, g_base, None, g1, None, None, *pg)


class StockMLP:
    """E structurally identical stock networks whose parameter segments sit `member_stride` floats
    apart starting at `flat[start]` (gradients at the same offsets of `grad_flat`)."""

    def __init__(self, desc, flat, grad_flat, start, member_stride, E, device, param_tensors=()):
        """`param_tensors`: the networks' `nn.Parameter`s (views of `flat`, member by member) — the autograd inputs
        of the differentiable calls; may stay empty for inference-only instances"""
        self.desc, self.E, self.member_stride = desc, E, member_stride
        self._window_grads = {}     # (_MlpSelectFn.backward)
        self.wide = desc.in0 + desc.in1 > MAX_WIDTH      # first layer wider than 64 inputs: single-network launches only
        self.param_tensors = list(param_tensors)
        self.params = flat[start:start + E * member_stride]
        self.grad_params = None if grad_flat is None else grad_flat[start:start + E * member_stride]
        self.in0, self.in1 = desc.in0, desc.in1
        self.out_cols = desc.head_cols[0] + desc.head_cols[1]
        self._anchor = torch.zeros(1, device=device, requires_grad=True)   # keeps the node in the graph
        self.accumulate = True     # False: parameter gradients overwrite the flat gradient buffer
        self._workspace = None
        self._start, self._deferred_rows = start, None
        self.device = device

    @staticmethod
    def _rows(x, width):
        """[..., width] -> 2-D [N, width] (or 3-D [E, N, width]) with a dense inner dim, no copy if possible"""
        assert x.shape[-1] == width
        if x.dim() == 2 and x.stride(-1) == 1:
            return x
        return x.reshape(-1, width) if x.is_contiguous() else x.contiguous().view(-1, width)

    @staticmethod
    def _rows_in_place(x, width):
        """Like `_rows`, but a [samples, T, width] window view that does not collapse to uniformly
        strided rows (e.g. states[:, b:]) comes back as `native.WindowRows` — read in place by the forward
        kernels — instead of a contiguous copy."""
        assert x.shape[-1] == width
        if x.dim() == 3 and not x.is_contiguous() and x.stride(2) == 1 and x.stride(0) != x.stride(1) * x.shape[1]:
            return native.WindowRows(x)
        return StockMLP._rows(x, width)

    def _launch_forward(self, x0, x1, out=None):
        if isinstance(x0, native.WindowRows):
            job, out = self.job(x0, x1, out)
            native.mlp_forward_multi([job])
            return out
        N = x0.shape[-2]
        if out is None:
            out = torch.empty((self.E, N, self.out_cols), dtype=torch.float32, device=self.device)
        assert out.shape == (self.E, N, self.out_cols) and out.is_contiguous()
        native.mlp_forward(self.desc, self.params, self.member_stride, self.E, x0, x1, N, out)
        return out

    def _workspace_for(self, N):
        need = native.mlp_backward_workspace(self.member_stride, self.E, N)
        if self._workspace is None or self._workspace.numel() < need:
            self._workspace = torch.zeros(need, dtype=torch.float32, device=self.device)
        return self._workspace

    def _reduce_mode(self, defer):
        if defer:
            return native.MLP_REDUCE_DEFER
        return native.MLP_REDUCE_ACCUMULATE if self.accumulate else native.MLP_REDUCE_OVERWRITE

    def backward_qloss(self, x0, x1, target_q, y, weights, clip_eps, loss_out, defer=False, state_grads=False):
        """Scalar-head ensemble: clipped double-Q loss + backward in one launch.  With `defer` the
        parameter gradients stay per-tile partial sums for `adam_partials`.  `state_grads`: -> [E, N, in0], the
        members' gradients w.r.t. x0 (a trainable representation continues the backward from their sum)."""
        N = x0.shape[-2]
        g0 = torch.empty((self.E, N, self.in0), dtype=torch.float32, device=self.device) if state_grads else None
        native.mlp_backward_qloss(self.desc, self.params, self.member_stride, self.E, x0, x1, N, target_q, y, weights,
                                  clip_eps, loss_out, self.grad_params, self._workspace_for(N), self._reduce_mode(defer),
                                  grad_x0=g0)
        self._deferred_rows = N if defer else None
        return g0

    def backward_qloss_return_ok(self, N: int, ret) -> bool:
        return native.mlp_backward_qloss_return_ok(self.desc, self.params, self.member_stride, self.E, N, ret)

    def backward_qloss_return(self, x0, x1, target_q, ret, weights, clip_eps, loss_out, defer=False, state_grads=False):
        """`backward_qloss` with the return target `ret` (native.VtraceArgs, its launch not issued) formed inside."""
        N = x0.shape[-2]
        g0 = torch.empty((self.E, N, self.in0), dtype=torch.float32, device=self.device) if state_grads else None
        native.mlp_backward_qloss_return(self.desc, self.params, self.member_stride, self.E, x0, x1, N, target_q, ret,
                                         weights, clip_eps, loss_out, self.grad_params, self._workspace_for(N),
                                         self._reduce_mode(defer), grad_x0=g0)
        self._deferred_rows = N if defer else None
        return g0

    def backward_policy_q(self, x0, x1, q_table, subset, E_sample):
        """-> [E, N, in1] action gradients of mean_b(-min_{e in subset} q_e) (`q_table` [E, N] from the
        forward on the same inputs)."""
        N = x0.shape[-2]
        g1 = torch.empty((self.E, N, self.in1), dtype=torch.float32, device=self.device)
        native.mlp_backward_policy_q(self.desc, self.params, self.member_stride, self.E, x0, x1, N, q_table, subset,
                                     E_sample, g1)
        return g1

    def backward_policy_sample(self, x0, eps, grad_a, log_alpha, defer=False):
        """Gaussian-head policy (E = 1): sampling backward + network backward in one launch; `grad_a`
        [members, N, A] are the action gradients of the ensemble members."""
        N = x0.shape[-2]
        native.mlp_backward_policy_sample(self.desc, self.params, self.member_stride, x0, N, eps, grad_a, log_alpha,
                                          self.grad_params, self._workspace_for(N), self._reduce_mode(defer))
        self._deferred_rows = N if defer else None

    def policy_step_fused_ok(self, critics: 'StockMLP', N: int) -> bool:
        """may `policy_step_fused` replace the critics' forward + `backward_policy_q` + `backward_policy_sample`?"""
        return (self.E == 1 and critics.E >= 2 and self.grad_params is not None and
                native.policy_step_fused_ok(critics.desc, critics.params, critics.member_stride, self.desc, self.params,
                                            self.member_stride, N))

    def policy_step_fused(self, critics: 'StockMLP', x0, action, eps, log_alpha, q_out=None, defer=False, subset=None,
                          sample_out=None):
        """Gaussian-head policy (E = 1) against the TWO critics the objective samples (`subset`: device i32[2], None =
        members 0 and 1 of a two-member ensemble): the whole policy step in one launch (`asac_policy_step_fused`);
        `q_out` [E, N, 1] receives those critics' values of (x0, action).  `action` None: the action is sampled inside
        the launch (no policy-forward / sampling launch in front of it) and lands in `sample_out` = (a_tanh [N, A],
        logp [N], ls [N, 2A] | None)."""
        N = x0.shape[-2]
        assert subset is not None or critics.E == 2
        a_out, logp_out, ls_out = sample_out if action is None else (None, None, None)
        native.policy_step_fused(critics.desc, critics.params, critics.member_stride, self.desc, self.params,
                                 self.member_stride, x0, N, action, eps, log_alpha, q_out, self.grad_params,
                                 self._workspace_for(N), self._reduce_mode(defer), subset=subset, a_out=a_out,
                                 logp_out=logp_out, ls_out=ls_out)
        self._deferred_rows = N if defer else None

    def adam_partials(self, opt, loss_out=None):
        """The deferred tile reduction + Adam over this network's segment(s) in one launch (`opt`: the
        FlatAdam whose moment buffers cover the same flat layout)."""
        N = self._deferred_rows
        assert N is not None, 'no deferred backward pending'
        tiles = native.mlp_backward_tiles(N, self.E)
        s0 = self._start
        s1 = s0 + self.E * self.member_stride
        native.adam_step_partials(self.params, self.grad_params, opt.exp_avg[s0:s1], opt.exp_avg_sq[s0:s1], opt.lr,
                                  opt.betas[0], opt.betas[1], opt.eps, opt.steps_done, self._workspace, tiles,
                                  self.E, self.member_stride, native.mlp_param_extent(self.desc),
                                  self.accumulate, loss_out, N if loss_out is not None else 0)
        self._deferred_rows = None

    def job(self, x0, x1, out=None):
        """A forward pass of this network as one job of `native.mlp_forward_multi` -> (job, out)."""
        N = x0.shape[0] if isinstance(x0, native.WindowRows) else x0.shape[-2]
        if self.wide and isinstance(x0, native.WindowRows):      # (the wide forward has no window addressing)
            x0 = x0.t.reshape(N, x0.t.shape[-1])
        if out is None:
            out = torch.empty((self.E, N, self.out_cols), dtype=torch.float32, device=self.device)
        assert out.shape == (self.E, N, self.out_cols) and out.is_contiguous()
        return native.mlp_job(self.desc, self.params, self.member_stride, self.E, x0, x1, N, out), out

    def _launch_backward(self, x0, x1, grad_out, need0, need1, param_grads, reduce_members=True, defer=False,
                         grad_target=None, later=None):
        """-> (grad_x0, grad_x1).  An input shared by the E members ([N, in]) gets the sum of the members'
        gradients unless `reduce_members` is False (then [E, N, in] comes back for a consumer kernel
        that sums itself).  `defer`: see `backward_qloss`.  `grad_target`: a buffer laid out like the parameters
        that receives (is overwritten with) the parameter gradients instead of the flat gradient buffer."""
        N = x0.shape[-2]
        E = self.E
        g0 = torch.empty((E, N, self.in0), dtype=torch.float32, device=self.device) if need0 else None
        g1 = torch.empty((E, N, self.in1), dtype=torch.float32, device=self.device) if need1 else None
        gp = ws = None
        mode = self._reduce_mode(defer and param_grads)
        if param_grads:
            gp, ws = self.grad_params, self._workspace_for(N)
            self._deferred_rows = N if defer else None
            if grad_target is not None:
                assert not defer
                gp, mode = grad_target, native.MLP_REDUCE_OVERWRITE
            if later is not None and not defer:
                # the partials stay in a workspace of this call's own until `later.flush()` sums them into the target — the
                # scratch buffer autograd would have been handed, or (direct mode) the flat gradient views
                # (`later`: DeferredPartialSums; the cached workspace would be overwritten by the network's next pass)
                ws = torch.empty(native.mlp_backward_workspace(self.member_stride, E, N), dtype=torch.float32,
                                 device=self.device)
                tiles, used = native.mlp_backward_tiles(N, E), native.mlp_param_extent(self.desc)
                for e in range(E):
                    later.add(ws[e * self.member_stride:], tiles, 16 if tiles >= 64 else 1, E * self.member_stride, used,
                              gp[e * self.member_stride:], accumulate=mode == native.MLP_REDUCE_ACCUMULATE)
                mode = native.MLP_REDUCE_DEFER
        native.mlp_backward(self.desc, self.params, self.member_stride, E, x0, x1, N, grad_out, g0, g1, gp, ws,
                            reduce_mode=mode)
        if reduce_members:
            if g0 is not None and x0.dim() == 2:
                g0 = g0.sum(0) if E > 1 else g0[0]     # input shared by the ensemble
            if g1 is not None and x1.dim() == 2:
                g1 = g1.sum(0) if E > 1 else g1[0]
        return g0, g1

    def call_select(self, base, t, x1=None, param_grads=True):
        """`self(base[:, t], x1)` with the window-aware backward (`_MlpSelectFn`)."""
        if torch.is_grad_enabled() and (param_grads or base.requires_grad or (x1 is not None and x1.requires_grad)):
            pg = param_grads and self.grad_params is not None
            return _MlpSelectFn.apply(self._anchor, base, t, x1, self, pg, *(self.param_tensors if pg else ()))
        return self._launch_forward(base[:, t], x1)

    def __call__(self, x0, x1=None, param_grads=True):
        """x0: [N, in0] (shared by all members) or [E, N, in0]; x1 likewise -> [E, N, out_cols]"""
        if isinstance(x0, native.WindowRows):     # inference-only addressing mode
            assert not (torch.is_grad_enabled() and (param_grads or x0.t.requires_grad))
            return self._launch_forward(x0, x1)
        if torch.is_grad_enabled() and (param_grads or x0.requires_grad or (x1 is not None and x1.requires_grad)):
            pg = param_grads and self.grad_params is not None
            return _MlpFn.apply(self._anchor, x0, x1, self, pg, *(self.param_tensors if pg else ()))
        return self._launch_forward(x0, x1)


class _GaussHeadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, raw, A):
        raw = raw.contiguous()
        loc = torch.empty((*raw.shape[:-1], A), dtype=raw.dtype, device=raw.device)
        scale = torch.empty_like(loc)
        native.gauss_head_fwd(raw, A, loc, scale)
        ctx.save_for_backward(raw)
        ctx.A = A
        return loc, scale

    @staticmethod
    def backward(ctx, g_loc, g_scale):
        (raw,) = ctx.saved_tensors
        g_raw = torch.empty_like(raw)
        native.gauss_head_bwd(raw, None if g_loc is None else g_loc.contiguous(),
                              None if g_scale is None else g_scale.contiguous(), ctx.A, g_raw)
        return g_raw, None


def gauss_head(raw: torch.Tensor, A: int):
    """raw [..., 2A] (mean | logstd) -> (loc, scale) of the stock policy's Normal (policy.py:170-172)"""
    return _GaussHeadFn.apply(raw, A)
