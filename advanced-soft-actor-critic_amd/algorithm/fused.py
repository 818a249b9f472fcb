"""Device-side building blocks of the fused SAC step (host Python over `asac_amd.native`).

  * `FlatParamGroup`  re-homes module parameters (and their .grad) into one flat f32 buffer so the
                      Polyak update and Adam are single streaming launches over contiguous HBM
  * `FlatAdam`        torch.optim.Adam-compatible optimizer driving `asac_adam_step` on a segment
  * `squash_sample`   differentiable rsample + tanh + squash-corrected log-prob (one kernel each way)
  * `clipped_q_loss`  clipped double-Q loss whose forward launch also produces the gradient
  * noise sources     `DeviceNoise` (Philox, graph-capturable) and `RecordedNoise` (parity tests)
"""
import numpy as np
import contextlib

import torch

from asac_amd import native

__all__ = ['FlatParamGroup', 'FlatAdam', 'squash_sample', 'squash_sample_ls', 'clipped_q_loss', 'DeviceNoise',
           'RecordedNoise']


class FlatParamGroup:
    """A set of named parameter lists laid out back to back in one flat buffer.

    `segments[name] = (start, stop)` in elements.  Every segment starts on a 16-byte boundary so
    the kernels take their float4 path.  After construction each parameter's `.data` is a view of
    `flat` and its `.grad` a view of `grad` (autograd accumulates in place into it), so
    "zero_grad" is one memset and an optimizer step one launch.
    """

    def __init__(self, named_params: list[tuple[str, list[torch.nn.Parameter]]], device, with_grad=True):
        self.segments: dict[str, tuple[int, int]] = {}
        self.params: dict[str, list[torch.nn.Parameter]] = {}
        total = 0
        for name, ps in named_params:
            start = total
            for p in ps:
                total += p.numel()
            total = (total + 3) // 4 * 4   # keep the next segment 16-byte aligned
            self.segments[name] = (start, total)
            self.params[name] = list(ps)
        self.numel = total
        self.flat = torch.zeros(max(total, 4), dtype=torch.float32, device=device)
        self.grad = torch.zeros(max(total, 4), dtype=torch.float32, device=device) if with_grad else None
        for name, ps in named_params:
            off = self.segments[name][0]
            for p in ps:
                n = p.numel()
                assert p.dtype == torch.float32, 'the fused update kernels are f32'
                view = self.flat[off:off + n].view(p.shape)
                view.copy_(p.data)
                p.data = view
                if with_grad and p.requires_grad:
                    p.grad = self.grad[off:off + n].view(p.shape)
                off += n

    def span(self, first: str, last: str | None = None) -> tuple[int, int]:
        return self.segments[first][0], self.segments[last or first][1]

    def rebind(self) -> None:
        """Re-point .data/.grad at the flat buffers (after load_state_dict-style replacements)."""
        for name, ps in self.params.items():
            off = self.segments[name][0]
            for p in ps:
                n = p.numel()
                view = self.flat[off:off + n].view(p.shape)
                if p.data.data_ptr() != view.data_ptr():
                    view.copy_(p.data)
                    p.data = view
                if self.grad is not None and p.requires_grad:
                    p.grad = self.grad[off:off + n].view(p.shape)
                off += n


class FlatAdam:
    """Adam (torch defaults: betas (0.9, 0.999), eps 1e-8, no weight decay / amsgrad) on one or
    more segments of a `FlatParamGroup`.  `steps_done` is a device counter shared by every optimizer
    of the learner and advanced once per train step by the owner."""

    def __init__(self, group: FlatParamGroup, names: list[str], lr: float, steps_done: torch.Tensor,
                 exp_avg: torch.Tensor, exp_avg_sq: torch.Tensor, betas=(0.9, 0.999), eps=1e-8):
        self.group, self.names = group, names
        self.lr, self.betas, self.eps = lr, betas, eps
        self.steps_done = steps_done
        self.exp_avg, self.exp_avg_sq = exp_avg, exp_avg_sq     # full-length buffers shared by all
        self.start, self.stop = group.span(names[0], names[-1])

    def step(self, start: int | None = None, stop: int | None = None) -> None:
        s = self.start if start is None else start
        e = self.stop if stop is None else stop
        if e <= s:
            return
        g = self.group
        native.adam_step(g.flat[s:e], g.grad[s:e], self.exp_avg[s:e], self.exp_avg_sq[s:e],
                         self.lr, self.betas[0], self.betas[1], self.eps, self.steps_done)

    def zero_grad(self) -> None:
        self.group.grad[self.start:self.stop].zero_()

    # -- torch.optim.Adam-compatible checkpoint format ------------------------------------------
    def _param_list(self):
        return [p for n in self.names for p in self.group.params[n]]

    def state_dict(self) -> dict:
        state, off_map = {}, {}
        step = self.steps_done.detach().to('cpu', torch.float32).reshape(())
        idx = 0
        for n in self.names:
            off = self.group.segments[n][0]
            for p in self.group.params[n]:
                k = p.numel()
                if int(step.item()) > 0:
                    state[idx] = {'step': step.clone(),
                                  'exp_avg': self.exp_avg[off:off + k].view(p.shape).clone(),
                                  'exp_avg_sq': self.exp_avg_sq[off:off + k].view(p.shape).clone()}
                off += k
                idx += 1
        return {'state': state,
                'param_groups': [{'lr': self.lr, 'betas': self.betas, 'eps': self.eps, 'weight_decay': 0,
                                  'amsgrad': False, 'maximize': False, 'foreach': None, 'capturable': False,
                                  'differentiable': False, 'fused': None, 'decoupled_weight_decay': False,
                                  'params': list(range(idx))}]}

    def load_state_dict(self, sd: dict) -> None:
        idx = 0
        for n in self.names:
            off = self.group.segments[n][0]
            for p in self.group.params[n]:
                k = p.numel()
                st = sd['state'].get(idx)
                if st is not None:
                    self.exp_avg[off:off + k].copy_(st['exp_avg'].reshape(-1))
                    self.exp_avg_sq[off:off + k].copy_(st['exp_avg_sq'].reshape(-1))
                    self.steps_done.fill_(int(float(st['step'])))
                off += k
                idx += 1


# ------------------------------------------------------------------------------------------------
class _SquashSampleLSFn(torch.autograd.Function):
    """Same op on the fused policy network's output `ls` = [..., 2A] (loc | scale): one gradient
    tensor of the same shape comes back, so no slicing / concatenation kernels run in backward."""

    @staticmethod
    def forward(ctx, ls, eps):
        A = ls.shape[-1] // 2
        loc, scale = ls[..., :A], ls[..., A:]
        a = torch.empty(loc.shape, dtype=ls.dtype, device=ls.device)
        logp = torch.empty(loc.shape[:-1], dtype=ls.dtype, device=ls.device)
        native.squash_sample_fwd(loc, scale, eps, a, logp)
        ctx.save_for_backward(ls, eps)
        return a, logp

    @staticmethod
    def backward(ctx, grad_a, grad_logp):
        ls, eps = ctx.saved_tensors
        A = ls.shape[-1] // 2
        g = torch.empty_like(ls)
        native.squash_sample_bwd(ls[..., :A], ls[..., A:], eps,
                                 None if grad_a is None else grad_a.contiguous(),
                                 None if grad_logp is None else grad_logp.contiguous(), g[..., :A], g[..., A:])
        return g, None


def squash_sample_ls(ls: torch.Tensor, eps: torch.Tensor):
    assert ls.is_contiguous()
    return _SquashSampleLSFn.apply(ls, eps)


class _SquashSampleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, loc, scale, eps):
        loc, scale, eps = loc.contiguous(), scale.contiguous(), eps.contiguous()
        a = torch.empty_like(loc)
        logp = torch.empty(loc.shape[:-1], dtype=loc.dtype, device=loc.device)
        native.squash_sample_fwd(loc, scale, eps, a, logp)
        ctx.save_for_backward(loc, scale, eps)
        return a, logp

    @staticmethod
    def backward(ctx, grad_a, grad_logp):
        loc, scale, eps = ctx.saved_tensors
        g_loc, g_scale = torch.empty_like(loc), torch.empty_like(scale)
        native.squash_sample_bwd(loc, scale, eps,
                                 None if grad_a is None else grad_a.contiguous(),
                                 None if grad_logp is None else grad_logp.contiguous(), g_loc, g_scale)
        return g_loc, g_scale, None


def squash_sample(loc: torch.Tensor, scale: torch.Tensor, eps: torch.Tensor):
    """x = loc + eps*scale; returns (tanh(x), squash-corrected log-prob summed over the action
    dim) — reference `Normal.rsample` + operators.py:12-14,22-24 — differentiable w.r.t. loc/scale."""
    return _SquashSampleFn.apply(loc, scale, eps)


class _ClippedQLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, tq, y, w, clip_eps):
        E, B = q.shape
        loss = torch.empty(E, dtype=q.dtype, device=q.device)
        grad = torch.empty_like(q)
        native.q_loss_fwd_bwd(q.contiguous(), tq.contiguous(), y.contiguous(),
                              None if w is None else w.contiguous(), clip_eps, loss, grad)
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    def backward(ctx, grad_loss):
        (grad,) = ctx.saved_tensors
        # d(sum_e l_e)/dq was produced by the forward launch; scale by the incoming per-e gradient
        return grad * grad_loss.unsqueeze(-1), None, None, None, None


def clipped_q_loss(q, tq, y, w, clip_eps):
    """q, tq: [E, B]; y, w: [B] -> per-ensemble losses [E] (reference sac_base.py:1539-1561)."""
    return _ClippedQLossFn.apply(q, tq, y, w, clip_eps)


# ------------------------------------------------------------------------------------------------
_ZERO_BLOCKS = {}


def _zero_block(shape, like):
    """a cached all-zero tensor (never written): the padding of `time_slice`'s backward"""
    key = (tuple(shape), like.dtype, like.device)
    z = _ZERO_BLOCKS.get(key)
    if z is None:
        z = torch.zeros(shape, dtype=like.dtype, device=like.device)
        # a tensor first made DURING capture lives in that graph's private pool and is filled by a captured node only:
        # it serves this graph (whose replays re-run the fill) and is not cached for others or for eager steps
        if not (like.is_cuda and torch.cuda.is_current_stream_capturing()):
            _ZERO_BLOCKS[key] = z
    return z


class _TimeSliceFn(torch.autograd.Function):
    """x[:, a:b] of a [B, L, ...] tensor.  The slice node's own backward is a fill and a copy (two launches of a few KB
    each, several times per backward walk through the representation); this one is ONE concatenation with cached zeros."""

    @staticmethod
    def forward(ctx, x, a, b):
        ctx.full, ctx.a, ctx.b = x.shape, a, b
        return x[:, a:b]

    @staticmethod
    def backward(ctx, g):
        B, L, *rest = ctx.full
        parts = []
        if ctx.a > 0:
            parts.append(_zero_block((B, ctx.a, *rest), g))
        parts.append(g)
        if ctx.b < L:
            parts.append(_zero_block((B, L - ctx.b, *rest), g))
        return torch.cat(parts, dim=1), None, None


def time_slice(x: torch.Tensor, a: int = 0, b: int | None = None) -> torch.Tensor:
    """x[:, a:b] (non-negative bounds) — x itself when that is all of it; differentiable inputs on the device get the
    one-launch backward of `_TimeSliceFn`"""
    L = x.shape[1]
    b = L if b is None else (b + L if b < 0 else b)
    a = a + L if a < 0 else a
    if a == 0 and b == L:
        return x
    if x.is_cuda and x.requires_grad and torch.is_grad_enabled() and x.dim() >= 2:
        return _TimeSliceFn.apply(x, a, b)
    return x[:, a:b]


class _ScaledMseFn(torch.autograd.Function):
    """mse_loss(pred, target) / divisor with its gradient formed by the loss launch itself (`asac_masked_mse`: one launch
    for ATen's subtract / square / mean, one for the division; the backward is one scaling) — reference sac_base.py:1817"""

    @staticmethod
    def forward(ctx, pred, target, divisor):
        from asac_amd import native
        grad = torch.empty_like(pred)
        loss = torch.empty((), dtype=pred.dtype, device=pred.device)
        native.masked_mse(pred.detach(), target, None, grad, loss)
        ctx.save_for_backward(grad)
        ctx.divisor = divisor
        return loss / divisor

    @staticmethod
    def backward(ctx, g_loss):
        (grad,) = ctx.saved_tensors
        from .sac_aux import _is_unit
        if _is_unit(g_loss):      # the root gradient of `autograd.grad(loss, ...)`: 1 / divisor is a host constant
            return grad / ctx.divisor, None, None
        return grad * (g_loss / ctx.divisor), None, None


class _MseMeanFn(torch.autograd.Function):
    """mse_loss(pred, target) over millions of elements, value and gradient from one launch (`asac_mse_mean_grad`)"""

    @staticmethod
    def forward(ctx, pred, target, workspace, grad_scale):
        from asac_amd import native
        grad = torch.empty_like(pred)
        loss = torch.empty((), dtype=pred.dtype, device=pred.device)
        native.mse_mean_grad(pred.detach(), target, grad, loss, workspace, grad_scale)
        ctx.save_for_backward(grad)
        ctx.grad_scale = float(grad_scale)
        return loss

    @staticmethod
    def backward(ctx, g_loss):
        (grad,) = ctx.saved_tensors
        from .sac_aux import _const_value
        # the stored gradient already carries `grad_scale`: a root gradient that IS that factor (the cached constant the
        # caller's division hands down) needs nothing more; anything else is one more pass
        if _const_value(g_loss) == ctx.grad_scale:
            return grad, None, None, None
        return grad * (g_loss / ctx.grad_scale), None, None, None


class _SmallMseFn(torch.autograd.Function):
    """mse_loss(pred, target) for a small tensor (the vector part of a plugin's observation loss): value and gradient from
    one launch (`asac_masked_mse`), the backward one scaling — ATen: subtract / square, a reduction, and backwards a fill and
    an elementwise launch"""

    @staticmethod
    def forward(ctx, pred, target):
        from asac_amd import native
        grad = torch.empty_like(pred)
        loss = torch.empty((), dtype=pred.dtype, device=pred.device)
        native.masked_mse(pred.detach(), target, None, grad, loss)
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    def backward(ctx, g_loss):
        (grad,) = ctx.saved_tensors
        from .sac_aux import _const_value
        v = _const_value(g_loss)
        if v == 1.0:
            return grad, None
        return grad * (v if v is not None else g_loss), None


MSE_INTERCEPT_MIN = 1 << 20       # below this ATen's chain is a few launches of microseconds each


@contextlib.contextmanager
def fused_mse_loss(workspace: torch.Tensor, grad_scale: float = 1.0):
    """While active, `torch.nn.functional.mse_loss(input, target)` — the call a plugin's `ModelObservation.get_loss` makes
    on decoded frames (reference envs/*/nn*.py under sac_base.py:1817) — with default arguments, a contiguous f32 device
    `input` of >= 2^20 elements that requires grad and a same-shaped `target` that does not runs as ONE launch for value and
    gradient (`asac_mse_mean_grad`; ATen: an elementwise pass writing the squared differences, a split reduction, and
    backwards a fill and another elementwise pass).  Every other call goes to the original function.  `workspace`: the
    caller's zeroed `native.mse_mean_grad_workspace()` floats; `grad_scale`: the constant the caller will multiply the
    loss with (`sac_aux._DivConstFn`), folded into the stored gradient."""
    import threading
    from asac_amd import native
    F = torch.nn.functional
    orig = F.mse_loss
    owner = threading.get_ident()      # the module attribute is process-global: only the learner's own thread is rerouted —
    #                                    an agent / evaluation thread calling `F.mse_loss` meanwhile gets the original (it must
    #                                    not touch the learner's workspace or inherit its grad_scale)

    def mse_loss(input, target, *args, **kwargs):
        if (threading.get_ident() == owner and not args and not kwargs and isinstance(input, torch.Tensor) and isinstance(target, torch.Tensor)
                and input.dim() >= 3 and input.shape == target.shape and input.numel() >= MSE_INTERCEPT_MIN
                and input.requires_grad and not target.requires_grad and torch.is_grad_enabled() and input.is_contiguous()):
            B, T = input.shape[:2]
            pred3 = input.view(B, T, -1)
            try:
                target3 = target.view(B, T, -1)
            except RuntimeError:
                target3 = None
            if target3 is not None and native.mse_mean_grad_ok(pred3, target3):
                return _MseMeanFn.apply(pred3, target3, workspace, grad_scale)
        elif (threading.get_ident() == owner and not args and not kwargs and isinstance(input, torch.Tensor)
              and isinstance(target, torch.Tensor) and input.is_cuda and input.dim() == 3 and input.shape == target.shape
              and 0 < input.numel() <= native.MASKED_MSE_MAX and input.dtype == torch.float32 and target.dtype == torch.float32
              and input.requires_grad and not target.requires_grad and torch.is_grad_enabled() and input.is_contiguous()
              and target.stride(-1) == 1):
            # the small companion of the frame loss (vector observations: tens of thousands of elements): one launch each way
            return _SmallMseFn.apply(input, target)
        return orig(input, target, *args, **kwargs)

    F.mse_loss = mse_loss
    try:
        yield
    finally:
        F.mse_loss = orig


def scaled_mse(pred: torch.Tensor, target: torch.Tensor, divisor: float) -> torch.Tensor:
    from asac_amd import native
    if (pred.is_cuda and pred.dim() == 3 and pred.dtype == torch.float32 and pred.is_contiguous() and target.shape == pred.shape
            and target.dtype == torch.float32 and target.stride(-1) == 1 and not target.requires_grad
            and 0 < pred.numel() <= native.MASKED_MSE_MAX):
        return _ScaledMseFn.apply(pred, target, divisor)
    return torch.nn.functional.mse_loss(pred, target) / divisor


class DeviceNoise:
    """Draws on the device; every call is graph-capturable.  `begin_step` produces every uniform and
    Gaussian draw of a train step with ONE `asac_noise_fill` launch (Philox keyed by `seed`, counter =
    the learner's device step counter, so a replayed graph draws fresh numbers) into the flat buffers
    the per-use buffers are views of; the following `uniform_` / `normal_` calls on those views are
    then no-ops.  Anything else falls back to torch's generator (also capturable)."""

    def __init__(self, seed: int | None = None):
        self.seed = torch.initial_seed() if seed is None else seed
        self._prefilled = []

    def _covered(self, buf: torch.Tensor) -> bool:
        return any(lo <= buf.data_ptr() < hi for lo, hi in self._prefilled)

    def begin_step(self, step_counter: torch.Tensor, u: torch.Tensor | None, flat: torch.Tensor | None,
                   subsets: torch.Tensor | None = None, ensemble: int = 0, polyak=None, zero=None) -> None:
        """`subsets` i32 [k, E_sample]: the step's ensemble subsets (only drawn when E_sample < ensemble).
        `polyak` = (target_flat, source_flat, tau): the step's target update rides in the same launch, and so
        does the memset of `zero` (the flat gradient buffer, where gradients are accumulated)."""
        if subsets is not None and subsets.shape[1] == ensemble:
            subsets = None       # whole ensemble: order-free, the buffers keep arange
        if polyak is not None or zero is not None:
            native.step_prologue(polyak, zero, self.seed, step_counter, u, flat, subsets, ensemble)
        else:
            native.noise_fill(self.seed, step_counter, u, flat, subsets, ensemble)
        self._prefilled = [(t.data_ptr(), t.data_ptr() + t.numel() * t.element_size())
                           for t in (u, flat, subsets) if t is not None]

    def begin_step_with_sample(self, step_counter, rb, flat, subsets=None, ensemble: int = 0, polyak=None,
                               zero=None, defer_weights: bool = False) -> int:
        """`begin_step` and the replay buffer's stratified sample (`rb.sample_into_static`'s tree walk) as ONE launch
        when that form applies (this source feeds the sampler, batch <= 1024); with a sharded replay
        (`rb.min_ratio_reducer`) the launch leaves the IS weights to `rb.sample_into_static`, which needs the MIN over
        ranks first;  -> 0 (not applicable), 1 (sampled: the caller runs only the buffer's weights / gather) or, with
        `defer_weights` and a batch of 257 .. 1 024 on one GPU, 3: sampled, and the IS weights are left to the gather's
        launch (`rb.sample_into_static(sampled=3)`: `asac_window_gather_pad_w`)."""
        if (rb.uniform_source is not self or rb.sharded is not None
                or rb.batch_size > native.PROLOGUE_SAMPLE_MAX_BATCH):
            return 0
        if subsets is not None and subsets.shape[1] == ensemble:
            subsets = None
        if defer_weights and rb.min_ratio_reducer is None and rb.batch_size > 256 and rb._gather_keys is not None:
            native.step_prologue_sample_partial(polyak, zero, self.seed, step_counter, rb._u, flat, subsets, ensemble, rb._tree,
                                                rb.capacity, rb.batch_size, rb._slot_ids, rb._leaf, rb._p, rb._ids, rb._min_p)
            self._prefilled = [(t.data_ptr(), t.data_ptr() + t.numel() * t.element_size())
                               for t in (rb._u, flat, subsets) if t is not None]
            return 3
        native.step_prologue_sample(polyak, zero, self.seed, step_counter, rb._u, flat, subsets, ensemble, rb._tree,
                                    rb.capacity, rb.batch_size, rb._slot_ids, rb._beta, rb.beta_increment_per_sampling,
                                    rb._leaf, rb._p, rb._ids, rb._w if rb.min_ratio_reducer is None else None,
                                    rb._min_p)
        self._prefilled = [(t.data_ptr(), t.data_ptr() + t.numel() * t.element_size())
                           for t in (rb._u, flat, subsets) if t is not None]
        return 1

    def uniform_(self, buf: torch.Tensor) -> None:
        if not self._covered(buf):
            buf.uniform_()

    def prefill(self, flat: torch.Tensor) -> None:
        if not self._covered(flat):
            flat.normal_()
            self._prefilled.append((flat.data_ptr(), flat.data_ptr() + flat.numel() * flat.element_size()))

    def normal_(self, buf: torch.Tensor) -> None:
        if not self._covered(buf):
            buf.normal_()

    def subset_(self, buf: torch.Tensor, ensemble: int) -> None:
        """buf: i32[E_sample] <- a uniformly random subset of range(ensemble) (the first E_sample
        entries of a random permutation, reference sac_base.py:1434)."""
        k = buf.numel()
        if k == ensemble or self._covered(buf):
            return   # whole ensemble: order-free, buf keeps arange / drawn by begin_step
        keys = torch.rand(ensemble, device=buf.device)
        buf.copy_(torch.topk(keys, k).indices.to(torch.int32))

    fill = uniform_   # replay-buffer uniform source protocol


class RecordedNoise:
    """Replays host-recorded draws (golden fixtures); eager mode only."""

    def __init__(self, u=(), eps=(), perm=()):
        self.u, self.eps, self.perm = list(u), list(eps), list(perm)

    def uniform_(self, buf):
        u = np.asarray(self.u.pop(0), dtype=np.float64)
        assert u.shape == tuple(buf.shape)
        buf.copy_(torch.from_numpy(u))

    fill = uniform_

    def prefill(self, flat):
        pass   # recorded draws are consumed one use at a time

    def begin_step(self, step_counter, u, flat, subsets=None, ensemble=0, polyak=None, zero=None):
        if polyak is not None:
            native.polyak(*polyak)
        if zero is not None:
            zero.zero_()

    def begin_step_with_sample(self, *a, **k) -> bool:
        return False

    def normal_(self, buf):
        e = np.asarray(self.eps.pop(0), dtype=np.float32)
        assert e.shape == tuple(buf.shape), (e.shape, tuple(buf.shape))
        buf.copy_(torch.from_numpy(e))

    def subset_(self, buf, ensemble):
        p = np.asarray(self.perm.pop(0))
        assert p.shape == (ensemble,)
        buf.copy_(torch.from_numpy(p[:buf.numel()].astype(np.int32)))

    def exhausted(self) -> bool:
        return not (self.u or self.eps or self.perm)
