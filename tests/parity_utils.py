"""Shared helpers of the GPU step-parity tests and `__graft_entry__.smoke()`:
build the product learner (HIP path) and the oracle learner (CPU) with identical weights,
episodes and random draws, and compare their observables."""
import numpy as np
import torch

from oracle import sac_ref


def copy_weights_to_oracle(agent, oracle):
    """product (cuda) -> oracle (cpu) for every module the oracle tracks + alphas"""
    mods = {'model_rep': agent.model_rep, 'model_target_rep': agent.model_target_rep,
            'model_policy': agent.model_policy}
    for i in range(agent.ensemble_q_num):
        mods[f'model_q_{i}'] = agent.model_q_list[i]
        mods[f'model_target_q_{i}'] = agent.model_target_q_list[i]
    for name, mod in oracle.named_modules().items():
        if name in mods:
            mod.load_state_dict({k: v.detach().cpu().clone() for k, v in mods[name].state_dict().items()})
    with torch.no_grad():
        oracle.log_c_alpha.copy_(agent.log_c_alpha.detach().cpu())
        oracle.log_d_alpha.copy_(agent.log_d_alpha.detach().cpu())


def load_golden_weights(agent, g, prefix='w0', only=None):
    """golden npz -> product learner: every module of the learner's ckpt_dict (named like the reference's) that the
    fixture holds weights for (`only`: restrict to these module names); -> {name: module}"""
    mods = {name: m for name, m in agent.ckpt_dict.items() if isinstance(m, torch.nn.Module)}
    with torch.no_grad():
        for name, mod in mods.items():
            if only is not None and name not in only:
                continue
            for k, p in mod.state_dict().items():
                key = f'{prefix}/{name}/{k}'
                if key in g.files:
                    p.copy_(torch.from_numpy(g[key].copy()))   # in place: parameters stay views of the flat buffer
        if f'{prefix}/log_c_alpha' in g.files:
            agent.log_c_alpha.copy_(torch.from_numpy(g[f'{prefix}/log_c_alpha'].copy()))
            agent.log_d_alpha.copy_(torch.from_numpy(g[f'{prefix}/log_d_alpha'].copy()))
    return mods


def golden_obs(g, i, j):
    """observation j of episode i; image observations are stored as their 8-bit pixel values"""
    if f'ep{i}/obs_{j}_u8' in g.files:
        return g[f'ep{i}/obs_{j}_u8'].astype(np.float32) / np.float32(255.)
    return g[f'ep{i}/obs_{j}']


def golden_episodes(g, n_obs=1):
    for i in range(int(g['n_episodes'])):
        yield dict(ep_indexes=g[f'ep{i}/ep_indexes'],
                   ep_obses_list=[golden_obs(g, i, j) for j in range(n_obs)],
                   ep_actions=g[f'ep{i}/ep_actions'], ep_rewards=g[f'ep{i}/ep_rewards'],
                   ep_dones=g[f'ep{i}/ep_dones'], ep_probs=g[f'ep{i}/ep_probs'],
                   ep_pre_seq_hidden_states=g[f'ep{i}/ep_pre_seq_hidden_states'])


def synthetic_episode(rng, obs_shapes, d_action_sizes, c_action_size, hidden_shape, T):
    parts = [np.eye(s, dtype=np.float32)[rng.integers(0, s, T)] for s in d_action_sizes]
    if c_action_size:
        parts.append(rng.random((T, c_action_size)).astype(np.float32))
    return dict(ep_indexes=np.arange(T, dtype=np.int32)[None],
                ep_obses_list=[rng.standard_normal((1, T, *s)).astype(np.float32) for s in obs_shapes],
                ep_actions=np.concatenate(parts, -1)[None],
                ep_rewards=rng.standard_normal((1, T)).astype(np.float32),
                ep_dones=(rng.random((1, T)) < 0.5),
                ep_probs=rng.random((1, T, sum(d_action_sizes) + c_action_size)).astype(np.float32),
                ep_pre_seq_hidden_states=rng.standard_normal((1, T, *hidden_shape)).astype(np.float32))


def host_draws(rng, B, n, A, E, has_d=False):
    """One train step's random draws in the reference's consumption order (continuous head)."""
    u = [rng.random(B)]
    eps = [rng.standard_normal((B, n + 1, A)).astype(np.float32),   # _get_y (train)
           rng.standard_normal((B, A)).astype(np.float32),          # _train_policy rsample
           rng.standard_normal((B, A)).astype(np.float32),          # _train_alpha sample
           rng.standard_normal((B, n + 1, A)).astype(np.float32)]   # _get_y (td error)
    n_perm = 5 + (5 if has_d else 0)
    perm = [rng.permutation(E) for _ in range(n_perm)]
    return u, eps, perm


def snapshot_tree(agent):
    return agent.replay_buffer._tree.cpu().numpy()


# ------------------------------------------------------------------------------------------------
# whole-step golden cases (tests/golden/f6_step_<case>.npz), shared by the oracle (CPU) and product (GPU) tests
# ------------------------------------------------------------------------------------------------
VEC = dict(obs_names=['vector'], obs_shapes=[(6,)], c_action_size=2, batch_size=32, capacity=512)
IMG = dict(obs_names=['vector', 'image'], obs_shapes=[(10,), (3, 30, 30)], c_action_size=4, batch_size=16, capacity=256)
VEC16 = dict(obs_names=['vector'], obs_shapes=[(6,)], c_action_size=2, batch_size=16, capacity=256)
IMG84 = dict(obs_names=['vector', 'image'], obs_shapes=[(10,), (3, 84, 84)], c_action_size=4, batch_size=8, capacity=128)
# case -> (plugin module name under tests.plugins, learner keywords, discrete action sizes, observation / size set)
STEP_CASES = {
    'cfg1': ('nn_vec', dict(n_step=1, use_priority=False), (), VEC),
    'cfg2': ('nn_vec', dict(n_step=4), (), VEC),
    'cfg3': ('nn_rnn', dict(n_step=3, burn_in_step=3, seq_encoder='RNN'), (), VEC),
    'attn': ('nn_attn', dict(n_step=3, burn_in_step=4, seq_encoder='ATTN'), (), VEC),
    'attn_tanh': ('nn_attn_tanh', dict(n_step=3, burn_in_step=4, seq_encoder='ATTN'), (), VEC),
    'hybrid': ('nn_vec', dict(n_step=3, ensemble_q_num=3, ensemble_q_sample=2), (3, 2), VEC),
    # BASELINE configs[3] / configs[4] compositions (reference tests/nn_conv_vanilla.py, tests/nn_conv_attn.py)
    'conv': ('nn_conv', dict(n_step=3, burn_in_step=5, ensemble_q_num=4, ensemble_q_sample=2), (), IMG),
    'conv_attn_cur': ('nn_conv_attn', dict(n_step=3, burn_in_step=5, seq_encoder='ATTN', curiosity='FORWARD'), (), IMG),
    # the frame size of the reference's environments (ConvLayers(84, 84, 3, 'simple')): the tiled convolution kernels
    'conv84': ('nn_conv84_small', dict(n_step=3, burn_in_step=2), (), IMG84),
    # ... their recurrent width (GRU(64): the wide-GRU MFMA recurrence) and their attention core (64 channels, 8 heads, 2 layers)
    'rnn_h64': ('nn_rnn_h64', dict(n_step=3, burn_in_step=3, seq_encoder='RNN'), (), VEC16),
    'attn_h64': ('nn_attn_h64', dict(n_step=3, burn_in_step=4, seq_encoder='ATTN'), (), VEC16),
}


def plugin(name):
    import importlib
    return importlib.import_module(f'tests.plugins.{name}')


# module of the reference's ckpt_dict -> the optimizer that owns its parameters
OPTIMIZER_OF = {'model_rep': 'optimizer_rep', 'model_policy': 'optimizer_policy',
                'model_forward_dynamic': 'optimizer_curiosity', 'model_inverse_dynamic': 'optimizer_curiosity',
                'model_rnd': 'optimizer_rnd'}


def product_first_moments(agent) -> dict:
    """{optimizer name (reference ckpt_dict naming): [Adam first-moment view per parameter, `parameters()` order]}"""
    out = {}
    for name, opt in agent.ckpt_dict.items():
        if not name.startswith('optimizer') or opt is None:
            continue
        views = []
        for seg in opt.names:
            off = opt.group.segments[seg][0]
            for p in opt.group.params[seg]:
                views.append(opt.exp_avg[off:off + p.numel()].view(p.shape))
                off += p.numel()
        out[name] = views
    return out


def _optimizer_scales(g) -> dict:
    """{optimizer: max |gradient| over all of its tensors in the golden}"""
    scale = {}
    for key in g.files:
        if key.startswith('g0/'):
            oname = key.split('/')[1]
            scale[oname] = max(scale.get(oname, 0.), float(np.abs(g[key]).max()))
    return scale


ZERO_GRAD_REL = 1e-6     # a tensor whose gradient is below this fraction of its optimizer's largest gradient is
#                          rounding noise around an analytic zero (softmax is invariant to the key-projection bias)


def zero_gradient_tensors(g, mods) -> list:
    """names (`module/parameter`) of the parameters whose reference gradient is numerically zero"""
    scale, names = _optimizer_scales(g), []
    for name, mod in mods.items():
        oname = OPTIMIZER_OF.get(name) or ('optimizer_q_' + name.rsplit('_', 1)[1] if name.startswith('model_q_') else None)
        for j, (k, _) in enumerate(mod.named_parameters()):
            key = f'g0/{oname}/{j}'
            if oname in scale and key in g.files and np.abs(g[key]).max() < ZERO_GRAD_REL * scale[oname]:
                names.append(f'{name}/{k}')
    return names


def assert_first_step_gradients(agent, g, rtol, atol_frac, skip=(), log_key=None):
    """After the FIRST train step Adam's first moment is (1 - beta1) * gradient: compares every gradient of the
    step with the reference's (`g0/<optimizer>/<j>`), entry by entry, within rtol * |want| + atol_frac * max|tensor|
    + ZERO_GRAD_REL * max|any gradient of the same optimizer| (the last term is the rounding floor of sums whose
    terms are as large as the sibling gradients; it is all that is left to compare for analytically zero ones)."""
    moments = product_first_moments(agent)
    opt_scale = _optimizer_scales(g)
    checked = 0
    for key in g.files:
        if not key.startswith('g0/'):
            continue
        _, oname, j = key.split('/')
        if oname in skip:
            continue
        want = g[key]
        got = moments[oname][int(j)].cpu().numpy()
        atol0 = atol_frac * float(np.abs(want).max()) + ZERO_GRAD_REL * opt_scale[oname]
        f = _tolerance_scale(f'{log_key}/{oname}', rtol) if log_key is not None else 1.
        rt, atol = rtol * f, atol0 * f
        if log_key is not None:     # observed: worst entry error relative to the tensor's largest entry, per optimizer
            rec = PARITY_LOG.setdefault(f'{log_key}/{oname}', {'max_err_over_tensor_max': 0., 'used': 0., 'used_of_default': 0.,
                                                               'rtol': rt, 'atol_frac': atol_frac * f, 'default_rtol': rtol,
                                                               'default_atol_frac': atol_frac, 'tensors': 0})
            err = np.abs(got.astype(np.float64) - want)
            rec['max_err_over_tensor_max'] = max(rec['max_err_over_tensor_max'],
                                                 float(err.max() / max(float(np.abs(want).max()), ZERO_GRAD_REL * opt_scale[oname], 1e-300)))
            rec['used'] = max(rec['used'], float((err / (atol + rt * np.abs(want) + 1e-300)).max()))
            rec['used_of_default'] = max(rec['used_of_default'], float((err / (atol0 + rtol * np.abs(want) + 1e-300)).max()))
            rec['tensors'] += 1
        np.testing.assert_allclose(got, want, rtol=rt, atol=atol, err_msg=key)
        checked += 1
    assert checked > 0
    return checked


def assert_weights_close(mods, g, n_steps, lr, rtol, atol, small_frac=1e-3, prefix='w1', only=None, log_key=None):
    """Post-training weights against the reference's.  Adam's first updates are sign-like (-lr * g / (|g| + eps)),
    so an entry whose reference gradient is analytically zero or at rounding level (|g0| < small_frac * max|g0| of
    its tensor, or the whole tensor is a `zero_gradient_tensors` one) may move by +-lr per step with a
    device-dependent sign: those entries — and only those — get 2 * lr * n_steps of slack.  Returns {tensor:
    fraction of slack entries} for the tensors that have any."""
    slack = {}
    rtol0, atol0 = rtol, atol
    f = _tolerance_scale(log_key, rtol) if log_key is not None else 1.
    rtol, atol = rtol * f, atol * f
    zero = set(zero_gradient_tensors(g, mods))
    for name, mod in mods.items():
        if only is not None and name not in only:
            continue
        params = [k for k, _ in mod.named_parameters()]
        oname = OPTIMIZER_OF.get(name) or ('optimizer_q_' + name.rsplit('_', 1)[1] if name.startswith('model_q_') else None)
        for k, v in mod.state_dict().items():
            key = f'{prefix}/{name}/{k}'
            if key not in g.files:
                continue
            got, want = v.detach().cpu().numpy(), g[key]
            loose = np.zeros(want.shape, dtype=bool)
            gkey = f'g0/{oname}/{params.index(k)}' if oname is not None and k in params else None
            if gkey is not None and gkey in g.files:
                g0 = np.abs(g[gkey])
                loose = g0 < small_frac * max(float(g0.max()), 1e-30)
                if f'{name}/{k}' in zero:
                    loose[...] = True
            err = np.abs(got - want)
            bound = np.where(loose, 2.2 * lr * n_steps, atol) + rtol * np.abs(want)
            if log_key is not None:     # observed: strict entries in the norm atol + rtol |want|; slack entries in units of lr
                rec = PARITY_LOG.setdefault(log_key, {'strict_max_abs': 0., 'used': 0., 'used_of_default': 0.,
                                                      'slack_max_over_lr_steps': 0., 'slack_entries': 0, 'entries': 0,
                                                      'rtol': rtol, 'atol': atol, 'default_rtol': rtol0, 'default_atol': atol0})
                if (~loose).any():
                    rec['strict_max_abs'] = max(rec['strict_max_abs'], float(err[~loose].max()))
                    rec['used'] = max(rec['used'], float((err[~loose] / bound[~loose]).max()))
                    rec['used_of_default'] = max(rec['used_of_default'],
                                                 float((err[~loose] / (atol0 + rtol0 * np.abs(want[~loose]))).max()))
                if loose.any():
                    rec['slack_max_over_lr_steps'] = max(rec['slack_max_over_lr_steps'], float(err[loose].max() / (lr * n_steps)))
                rec['slack_entries'] += int(loose.sum())
                rec['entries'] += int(err.size)
            assert (err <= bound).all(), (f'{name}/{k}: {int((err > bound).sum())} of {err.size} entries off, worst '
                                          f'{float(err.max()):.3g} (strict entries: {float(err[~loose].max()) if (~loose).any() else 0:.3g})')
            if loose.any():
                slack[f'{name}/{k}'] = float(loose.mean())
    return slack


# ------------------------------------------------------------------------------------------------
# measured parity errors: every float comparison of the step-parity tests goes through `check`,
# which records the OBSERVED error beside the tolerance it was tested under.  The record is written to
# gpurun_out/parity_errors.json when the test process ends (tools/install_profiles.py copies its summary to
# profiles/<round>_parity_errors.json); tolerances in the tests are set to <= 4x what this file shows.
#   max_abs   max |got - want|
#   max_rel   max |got - want| / |want| over the entries with |want| >= 1e-3 * max|want| (the rest is judged by max_abs)
#   used      max |got - want| / (atol + rtol |want|): the fraction of the tolerance the worst entry consumed
# ------------------------------------------------------------------------------------------------
# Tolerances: every call site names a DEFAULT bound (the fp32 bound one would write down without measuring: device
# libm / MFMA accumulation order against the host).  `tests/parity_tolerances.json` (written by tools/set_tolerances.py
# from a recorded run) replaces it per key by  default x max(4 x used, floor)  — i.e. 4x the worst error observed on
# MI355X in the same norm, never looser than the default, never tighter than 2 ulp.  ASAC_PARITY_RECORD=1 runs the
# tests under the defaults (to record after a kernel change).
# ------------------------------------------------------------------------------------------------
PARITY_LOG = {}
ULP2 = 2.4e-7
_TOL_TABLE = None


def _tolerance_scale(key: str, rtol: float) -> float:
    global _TOL_TABLE
    import os
    if os.environ.get('ASAC_PARITY_RECORD'):
        return 1.
    if _TOL_TABLE is None:
        import json
        from pathlib import Path
        path = Path(__file__).resolve().parent / 'parity_tolerances.json'
        _TOL_TABLE = json.loads(path.read_text()) if path.exists() else {}
    rec = _TOL_TABLE.get(key)
    if rec is None:
        # a comparison without an entry would silently run at the call site's (loose) default: refuse it.  New keys are
        # recorded first (ASAC_PARITY_RECORD=1 on the GPU box, then tools/set_tolerances.py --merge)
        raise AssertionError(f'parity key {key!r} has no entry in tests/parity_tolerances.json: record it '
                             f'(ASAC_PARITY_RECORD=1 python -m pytest tests -m gpu; python tools/set_tolerances.py --merge)')
    return min(1., max(4. * rec['used_of_default'], ULP2 / max(rtol, 1e-300)))


def check(key: str, got, want, rtol: float, atol: float = 0., enforce: bool = True):
    """np.testing.assert_allclose(got, want, rtol, atol) that also records the observed error under `key`
    ('<test>/<case>/<observable>'; several calls under one key — steps of a run — keep the worst).  `rtol` / `atol`
    are the call site's DEFAULT bound; the enforced one is scaled by the tolerance table (see above).
    `enforce=False`: record only."""
    rtol0, atol0 = rtol, atol
    f = _tolerance_scale(key, rtol)
    rtol, atol = rtol * f, atol * f
    got = np.asarray(got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else got, dtype=np.float64)
    want = np.asarray(want.detach().cpu().numpy() if isinstance(want, torch.Tensor) else want, dtype=np.float64)
    got, want = np.broadcast_arrays(got, want)
    err = np.abs(got - want)
    finite = np.isfinite(want) & np.isfinite(got)
    same_inf = (~finite) & (got == want)
    err = np.where(same_inf, 0., err)
    big = np.abs(want) >= 1e-3 * max(float(np.abs(np.where(finite, want, 0.)).max(initial=0.)), 1e-300)
    max_abs = float(err.max(initial=0.))
    max_rel = float((err[big & finite] / np.abs(want[big & finite])).max(initial=0.)) if (big & finite).any() else 0.
    aw = np.abs(np.where(finite, want, 0.))
    used = float((err / (atol + rtol * aw + 1e-300)).max(initial=0.)) if err.size else 0.
    used0 = float((err / (atol0 + rtol0 * aw + 1e-300)).max(initial=0.)) if err.size else 0.
    rec = PARITY_LOG.setdefault(key, {'max_abs': 0., 'max_rel': 0., 'used': 0., 'used_of_default': 0., 'rtol': rtol, 'atol': atol,
                                      'default_rtol': rtol0, 'default_atol': atol0, 'calls': 0, 'entries': 0,
                                      'enforced': bool(enforce)})
    rec['max_abs'], rec['max_rel'], rec['used'] = max(rec['max_abs'], max_abs), max(rec['max_rel'], max_rel), max(rec['used'], used)
    rec['used_of_default'] = max(rec['used_of_default'], used0)
    rec['rtol'], rec['atol'] = rtol, atol
    rec['calls'] += 1
    rec['entries'] = int(err.size)
    if enforce:
        np.testing.assert_allclose(got, want, rtol=rtol, atol=atol, err_msg=key)


def _dump_parity_log():
    if not PARITY_LOG:
        return
    import json
    import os
    from pathlib import Path
    out = Path(os.environ.get('ASAC_PARITY_LOG', Path(__file__).resolve().parent.parent / 'gpurun_out' / 'parity_errors.json'))
    try:
        out.parent.mkdir(parents=True, exist_ok=True)
        old = json.loads(out.read_text()) if out.exists() else {}
        old.update(PARITY_LOG)
        out.write_text(json.dumps(old, indent=1, sort_keys=True))
    except OSError:
        pass


import atexit  # noqa: E402

atexit.register(_dump_parity_log)


def hooked_learner():
    """-> a TEST-SIDE subclass of the product's `SAC_Base` with an `after_rep_q_update` hook: called (eager steps only)
    right after the representation / critic update of a step returns — the step-parity tests read the freshly updated
    weights there, or align them with the reference's so that what the step computes afterwards is compared from identical
    weights.  The product class carries no such hook."""
    import torch
    from algorithm.sac_base import SAC_Base

    class HookedSAC(SAC_Base):
        after_rep_q_update = None

        def _train_rep_q(self, *args, **kwargs):
            out = super()._train_rep_q(*args, **kwargs)
            if self.after_rep_q_update is not None and not torch.cuda.is_current_stream_capturing():
                self.after_rep_q_update()
            return out

    return HookedSAC
