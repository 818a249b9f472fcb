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
    from algorithm.sac_base import SAC_Base
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
        pu.assert_weights_close(mods, g, 1, 3e-4, rtol=1e-3, atol=2e-5, prefix=f'step{step_box[0]}/w_rq')
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
        np.testing.assert_allclose(agent._stats['loss_q'].item(), g[f'step{s}/loss_q'], rtol=1e-3)
        np.testing.assert_allclose(agent._td_error.cpu().numpy()[:, None], g[f'step{s}/td_error'], rtol=1e-3, atol=1e-4)
        np.testing.assert_allclose(rb._tree.cpu().numpy(), g[f'step{s}/tree'], rtol=1e-3, atol=1e-5)
        if s == 0:
            checked = pu.assert_first_step_gradients(agent, g, rtol=2e-3, atol_frac=5e-5)
            print(f'{case}: {checked} gradient tensors of step 0 match the reference')
    slack = pu.assert_weights_close(mods, g, n_steps, 3e-4, rtol=1e-3, atol=2e-5)
    print(f'{case}: entries given +-lr slack {slack}')
    w_atol = 2 * n_steps * 3e-4 * 1.1
    for name, obj in agent.ckpt_dict.items():
        if isinstance(obj, torch.Tensor) and f'w1/t/{name}' in g.files:     # contrastive weights, normaliser statistics
            np.testing.assert_allclose(obj.detach().cpu().numpy(), g[f'w1/t/{name}'], rtol=1e-3, atol=w_atol,
                                       err_msg=name)
    rb.check_health()
    agent.close()


def test_prediction_heads_run():
    """`use_prediction` cannot be pinned against the reference: with a trainable representation the
    reference raises 'Trying to backward through the graph a second time' (its _train_rpm differentiates
    the representation graph the Q loss already freed).  Here the head runs; check it trains."""
    import asac_amd  # noqa: F401
    from algorithm.sac_base import SAC_Base
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
    from algorithm.sac_base import SAC_Base
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
    assert seen['asac_conv2_backward']['calls'] == 3 and seen['asac_linear_tanh_backward']['calls'] == 3
    agent.close()
