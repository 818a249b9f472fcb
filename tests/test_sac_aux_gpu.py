"""GPU parity of the optional learner heads (SURVEY.md §8a row a21 + a14): curiosity, RND,
observation normalisation, DQN-like discrete targets, siamese ATC / BYOL (with the Q-consistency and
adaptive-gradient variants) — the product learner against steps recorded from the reference
(`tests/golden/f6_step_aux_*.npz`, minted with the same plugin file `tests/plugins/nn_vec_full.py`).
PER ids must be bit-exact; float observables within fp32 tolerance (1e-3); the first step's gradients of EVERY
optimizer (representation incl. the sign-gated auxiliary parts, critics, policy, siamese / RND / curiosity heads)
entry by entry against the reference's; weights after the steps under the sign-aware bound of
`parity_utils.assert_weights_close` (slack only where the reference gradient is at rounding level)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests import parity_utils as pu  # noqa: E402
from tests.plugins import nn_vec_full  # noqa: E402

CASES = {
    'aux_curiosity': (dict(curiosity='FORWARD'), (), 2),
    'aux_rnd': (dict(use_rnd=True), (3,), 2),
    'aux_norm': (dict(use_normalization=True), (), 2),
    'aux_dqn': (dict(discrete_dqn_like=True), (3, 2), 0),
    'aux_atc': (dict(siamese='ATC', siamese_use_q=True, burn_in_step=2), (), 2),
    'aux_byol': (dict(siamese='BYOL', siamese_use_q=True, siamese_use_adaptive=True, burn_in_step=2), (), 2),
}


# `step<s>/w_rq` = the reference's representation / critic weights when its `_train_rep_q` returns, i.e. after the
# siamese heads added their (gated) gradients and every optimizer of that block stepped (reference 1577-1603)
NO_ALIGN = ()


@pytest.mark.parametrize('case', list(CASES))
def test_optional_heads_vs_reference_golden(golden_dir, case):
    import asac_amd  # noqa: F401
    from algorithm.fused import RecordedNoise
    SAC_Base = pu.hooked_learner()
    from algorithm.utils.enums import convert_config_to_enum
    g = np.load(golden_dir / f'f6_step_{case}.npz')
    kw, d_sizes, c_size = CASES[case]
    kw = dict(kw)
    convert_config_to_enum(kw)
    torch.manual_seed(0)
    agent = SAC_Base(['vector'], [(6,)], list(d_sizes), c_size, None, nn_vec_full, device='cuda:0', batch_size=16,
                     n_step=3, replay_config={'capacity': 256}, hip_config={'use_graph': False}, **kw)
    with torch.no_grad():
        for name, obj in agent.ckpt_dict.items():
            if isinstance(obj, torch.nn.Module):
                for k, p in obj.state_dict().items():
                    p.copy_(torch.from_numpy(g[f'w0/{name}/{k}'].copy()))
            elif isinstance(obj, torch.Tensor) and f'w0/t/{name}' in g.files:
                obj.copy_(torch.from_numpy(g[f'w0/t/{name}'].copy()))
        agent.log_c_alpha.copy_(torch.from_numpy(g['w0/log_c_alpha'].copy()))
        agent.log_d_alpha.copy_(torch.from_numpy(g['w0/log_d_alpha'].copy()))
    for ep in pu.golden_episodes(g):
        agent.put_episode(**ep)
    rb = agent.replay_buffer
    if case == 'aux_norm':     # normaliser statistics after ingesting the episodes feed every forward
        np.testing.assert_allclose(agent.running_means[0].cpu().numpy(), g['w1/t/running_means_0'], rtol=1e-4, atol=1e-5)
    n_steps = int(g['n_steps'])
    mods = {name: m for name, m in agent.ckpt_dict.items() if isinstance(m, torch.nn.Module)}
    step_box = [0]

    def align_with_reference():     # see tests/test_sac_step_gpu.py: compare the fresh update, then align
        pu.assert_weights_close(mods, g, 1, 3e-4, rtol=1e-3, atol=2e-5, prefix=f'step{step_box[0]}/w_rq', log_key=f'aux/{case}/w_rq')
        pu.load_golden_weights(agent, g, prefix=f'step{step_box[0]}/w_rq')

    if 'step0/w_rq/model_q_0/' + next(iter(agent.model_q_list[0].state_dict())) in g.files and case not in NO_ALIGN:
        agent.after_rep_q_update = align_with_reference
    for s in range(n_steps):
        step_box[0] = s
        eps = [g[f'step{s}/eps{j}'] for j in range(int(g[f'step{s}/n_eps']))]
        agent.noise = RecordedNoise([g[f'step{s}/u']], eps, list(g[f'step{s}/perm']))
        rb.uniform_source = agent.noise
        assert agent.train() == s + 1
        assert agent.noise.exhausted(), 'every recorded draw must be consumed, in order'
        assert np.array_equal(rb._ids.cpu().numpy(), g[f'step{s}/sample_ids']), f'step {s}: PER index selection'
        pu.check(f'aux/{case}/loss_q', agent._stats['loss_q'].item(), g[f'step{s}/loss_q'], rtol=2e-4)
        pu.check(f'aux/{case}/td_error', agent._td_error.cpu().numpy()[:, None], g[f'step{s}/td_error'], rtol=2e-4, atol=2e-5)
        pu.check(f'aux/{case}/tree', rb._tree.cpu().numpy(), g[f'step{s}/tree'], rtol=2e-4, atol=1e-6)
        if s == 0:
            checked = pu.assert_first_step_gradients(agent, g, rtol=2e-3, atol_frac=5e-5, log_key=f'aux/{case}/grad0')
            print(f'{case}: {checked} gradient tensors of step 0 match the reference')
    slack = pu.assert_weights_close(mods, g, n_steps, 3e-4, rtol=1e-3, atol=2e-5, log_key=f'aux/{case}/weights')
    print(f'{case}: entries given +-lr slack {slack}')
    w_atol = 2 * n_steps * 3e-4 * 1.1
    for name, obj in agent.ckpt_dict.items():
        if isinstance(obj, torch.Tensor) and f'w1/t/{name}' in g.files:     # contrastive weights, normaliser statistics
            pu.check(f'aux/{case}/tensor_{name}', obj.detach().cpu().numpy(), g[f'w1/t/{name}'], rtol=1e-3, atol=w_atol)
    rb.check_health()
    agent.close()


@pytest.mark.parametrize('fused_loss', [True, False], ids=['one_launch_loss', 'aten_loss'])
@pytest.mark.parametrize('tag', ['a', 'b', 'c'])
def test_rpm_vs_reference_golden(golden_dir, tag, fused_loss):
    """`use_prediction` (BASELINE configs[4]): the product's `_train_rpm` — transition / reward / observation losses,
    the cosine-sign gating of their gradients into the representation (`calculate_adaptive_weights`), the prediction
    models' Adam step — against the reference's OWN `_train_rpm` called on the same inputs (`f11_rpm.npz`).  The
    reference's whole step raises with `use_prediction` (its `_train_rep_q` has freed the graph `_train_rpm`
    differentiates again), so the function is pinned in isolation, the way f4 pins `_get_y`.  Variants: main gradient g
    (gates 1, 0, 1), -g (gates 0, 1, 0), other transition_kl without extra data (gates 0, 0, 1)."""
    import asac_amd  # noqa: F401
    SAC_Base = pu.hooked_learner()
    g = np.load(golden_dir / 'f11_rpm.npz')
    B, n, kl, extra = g[f'{tag}/cfg']
    torch.manual_seed(0)
    agent = SAC_Base(['vector'], [(6,)], [], 2, None, nn_vec_full, device='cuda:0', batch_size=int(B), n_step=int(n),
                     replay_config={'capacity': 256}, use_prediction=True, transition_kl=float(kl),
                     use_extra_data=bool(extra), hip_config={'use_graph': False, 'fused_rpm_loss': fused_loss})
    heads = ('model_rep', 'model_target_rep', 'model_transition', 'model_reward', 'model_observation')
    with torch.no_grad():
        for name in heads:
            for k, p in getattr(agent, name).state_dict().items():
                p.copy_(torch.from_numpy(g[f'{tag}/w0/{name}/{k}'].copy()))
    cu = lambda key: torch.from_numpy(g[f'{tag}/{key}'].copy()).cuda()  # noqa: E731
    obs = [cu('obs')]
    nx_states, _ = agent.model_rep(obs, None, None)
    with torch.no_grad():
        nx_target_states, _ = agent.model_target_rep(obs, None, None)
    K = f'aux/rpm_{tag}' if fused_loss else f'aux/rpm_{tag}_aten_loss'
    pu.check(f'{K}/nx_states', nx_states, g[f'{tag}/nx_states'], rtol=1e-5, atol=1e-6)
    main = float(g[f'{tag}/flip']) * torch.mean(torch.square(torch.sum(nx_states * cu('coef'), dim=-1)))
    agent._params.grad.zero_()
    main.backward(retain_graph=True)
    rep_params = list(agent.model_rep.parameters())
    for j, p in enumerate(rep_params):
        pu.check(f'{K}/g_main', p.grad, g[f'{tag}/g_main/{j}'], rtol=1e-4, atol=1e-5 * float(np.abs(g[f'{tag}/g_main/{j}']).max()))
    grads_main = [p.grad.detach() for p in rep_params]
    ret = agent._train_rpm(grads_main, obs, nx_states, nx_target_states, cu('actions'), cu('rewards'))
    pu.check(f'{K}/returned_entropy_lossreward_lossobs', torch.stack(list(ret)), g[f'{tag}/ret'], rtol=1e-5)
    # the gates: a wrong one adds or drops a whole auxiliary gradient, far outside any rounding bound
    for j, p in enumerate(rep_params):
        want = g[f'{tag}/g_rep_after/{j}']
        pu.check(f'{K}/g_rep_after_gating', p.grad, want, rtol=1e-4, atol=1e-5 * float(np.abs(want).max()))
    assert any(not np.array_equal(g[f'{tag}/g_rep_after/{j}'], g[f'{tag}/g_main/{j}']) for j in range(len(rep_params)))
    moments = pu.product_first_moments(agent)['optimizer_prediction']
    n_pred = len(moments)
    assert f'{tag}/g_pred/{n_pred - 1}' in g.files and f'{tag}/g_pred/{n_pred}' not in g.files
    for j, m_ in enumerate(moments):    # Adam's first moment after the first step = (1 - beta1) g
        want = g[f'{tag}/g_pred/{j}']
        pu.check(f'{K}/g_prediction_models', m_ / 0.1, want, rtol=1e-4, atol=1e-5 * float(np.abs(want).max()))
    # post-Adam weights: the first update is -lr g / (|g| + eps), sign-like — entries whose gradient is at rounding level
    # (< 1e-3 of their tensor's largest) may move by +-lr with a device-dependent sign and get 2.2 lr; the rest is strict
    j = 0
    for name in heads[2:]:
        for k, v in getattr(agent, name).named_parameters():
            want, g0 = g[f'{tag}/w1/{name}/{k}'], np.abs(g[f'{tag}/g_pred/{j}'])
            j += 1
            strict = g0 >= 1e-3 * g0.max()
            got = v.detach().cpu().numpy()
            pu.check(f'{K}/weights_after_adam', got[strict], want[strict], rtol=1e-5, atol=1e-7)
            assert np.abs(got - want)[~strict].max(initial=0.) <= 2.2 * 3e-4
            assert np.abs(want - g[f'{tag}/w0/{name}/{k}']).max() > 0
    assert j == n_pred
    agent.close()


def test_prediction_heads_inside_the_captured_step():
    """... and the head inside the whole step (which the reference cannot run, see above): the step captures, the
    prediction models train, nothing diverges.  Step-level parity with `use_prediction` is against the oracle, which
    restates the product's graph-retaining order: tests/test_full_size_gpu.py (cfg5)."""
    import asac_amd  # noqa: F401
    SAC_Base = pu.hooked_learner()
    rng = np.random.default_rng(0)
    torch.manual_seed(0)
    agent = SAC_Base(['vector'], [(6,)], [], 2, None, nn_vec_full, device='cuda:0', batch_size=16, n_step=3,
                     replay_config={'capacity': 256}, use_prediction=True, curiosity=None)
    for T in (40, 30, 50):
        agent.put_episode(**pu.synthetic_episode(rng, [(6,)], [], 2, (0,), T))
    before = agent._params.flat.clone()
    for _ in range(6):
        agent.train()
    seg = agent._params.segments['prediction']
    assert not torch.equal(before[seg[0]:seg[1]], agent._params.flat[seg[0]:seg[1]])
    assert torch.isfinite(agent._params.flat).all() and agent._graph is not None
    agent.close()


def test_adaptive_gating_with_fused_layers():
    """`calculate_adaptive_weights` (reference 1607-1631) on a representation whose layers run as fused launches
    (convolution stack, wide-input MLP head, Linear + tanh): an auxiliary loss that opposes the main gradient must
    leave `.grad` untouched (gate 0), one aligned with it must add its gradient once (gate 1) — the fused backward
    kernels may not add anything on their own while `autograd.grad` runs."""
    import asac_amd  # noqa: F401
    from asac_amd import native
    from algorithm.fused_mlp import direct_param_grads
    SAC_Base = pu.hooked_learner()
    from tests.plugins import nn_conv
    torch.manual_seed(0)
    agent = SAC_Base(['vector', 'image'], [(10,), (3, 30, 30)], [], 4, None, nn_conv, device='cuda:0', batch_size=16,
                     n_step=3, burn_in_step=2, replay_config={'capacity': 256}, hip_config={'use_graph': False})
    rep = agent.model_rep
    obs = [torch.randn(16, 6, 10, device='cuda'), torch.rand(16, 6, 3, 30, 30, device='cuda')]
    with native.LaunchProfiler(repeat=1) as prof:      # (the default re-issues every launch 20x for timing)
        state, _ = rep(obs, torch.zeros(16, 6, 4, device='cuda'), None)
        main = state.square().mean()
        agent._params.grad.zero_()
        with direct_param_grads():
            main.backward(retain_graph=True)
        g_main = [p.grad.clone() for p in rep.parameters()]
        assert all(g.abs().max() > 0 for g in g_main)
        agent.calculate_adaptive_weights([g.clone() for g in g_main], [-main], rep)          # cosine -1: gate 0
        for p, g in zip(rep.parameters(), g_main):
            assert torch.equal(p.grad, g), 'an opposing auxiliary loss changed the gradient'
        agent.calculate_adaptive_weights([g.clone() for g in g_main], [main * 1.0], rep)     # cosine +1: gate 1
        for p, g in zip(rep.parameters(), g_main):
            np.testing.assert_allclose(p.grad.cpu().numpy(), 2 * g.cpu().numpy(), rtol=1e-5, atol=1e-9)
    seen = prof.summary()
    assert seen['asac_conv2_backward']['calls'] == 3 and seen['asac_linear_tanh_backward2']['calls'] == 3
    agent.close()


@pytest.mark.parametrize('n,K', [(25_000, 3), (1000, 1), (70_001, 4)])
def test_cosine_gate_add_matches_torch(n, K):
    """`asac_cosine_gate_add` against the reference's arithmetic (sac_base.py:1619-1631): cosine similarity of the flat
    main gradient with each auxiliary one, sign().clamp(min=0) gates, `grad += gate * aux` loss by loss."""
    import asac_amd  # noqa: F401
    from asac_amd import native
    gen = torch.Generator().manual_seed(n)
    main = torch.randn(n, generator=gen)
    aux = [torch.randn(n, generator=gen) for _ in range(K)]
    aux[0] = 0.3 * main + 0.1 * aux[0]          # clearly aligned: gate 1
    if K > 1:
        aux[1] = -0.5 * main + 0.1 * aux[1]     # clearly opposed: gate 0
    if K > 3:
        aux[3] = torch.zeros(n)                 # zero gradient: cos 0, gate 0
    grad = torch.randn(n, generator=gen)
    want = grad.clone()
    gates = []
    for a in aux:
        cos = torch.nn.functional.cosine_similarity(main.reshape(1, -1), a.reshape(1, -1))
        gate = torch.sign(cos).clamp(min=0)
        gates.append(float(gate))
        want += gate * a
    d_grad, d_gates = grad.cuda(), torch.empty(K, device='cuda')
    native.cosine_gate_add(main.cuda(), [a.cuda() for a in aux], d_grad, d_gates)
    assert d_gates.cpu().tolist() == gates
    assert torch.equal(d_grad.cpu(), want)      # same products and additions, entry by entry
