"""GPU: the stock-network kernel chains (Q loss inside the backward, gradients folded into Adam, batched
sampling jobs, window-shared policy forward, on-chip policy objective, ...) against the GENERIC learner path
of the same `SAC_Base` — user-module forward + PyTorch-ROCm autograd + the per-purpose kernels
(`hip_config={'fused_mlp': False}`) — on shapes the reference-minted goldens do not cover: odd batch
sizes, ensemble subsets, n_step 1, no priority, no importance sampling, different clip widths.

Both learners start from the same weights, see the same episodes and draw the same noise (the step's
Philox launch is a pure function of seed and step counter), so after a few steps their sampled ids must
be identical and their parameters / priorities / written-back probabilities must agree to fp32 tolerance."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests import parity_utils as pu  # noqa: E402
from tests.plugins import nn_vec  # noqa: E402

CASES = {
    'b33_n3_e4s2': dict(batch_size=33, n_step=3, ensemble_q_num=4, ensemble_q_sample=2),
    'b256_n1': dict(batch_size=256, n_step=1),
    'b64_n4_noprio': dict(batch_size=64, n_step=4, use_priority=False),
    'b40_n2_nois': dict(batch_size=40, n_step=2, use_n_step_is=False),
    'b96_n5_e3_clip': dict(batch_size=96, n_step=5, ensemble_q_num=3, ensemble_q_sample=3, clip_epsilon=0.05),
    'b50_n2_e1': dict(batch_size=50, n_step=2, ensemble_q_num=1, ensemble_q_sample=1),
}


def _agent(kw, fused, graph):
    import asac_amd  # noqa: F401
    from algorithm.sac_base import SAC_Base
    torch.manual_seed(7)
    return SAC_Base(['vector'], [(6,)], [], 2, None, nn_vec, device='cuda:0', replay_config={'capacity': 1024},
                    hip_config={'use_graph': graph, 'graph_warmup': 1, 'fused_mlp': fused}, **kw)


@pytest.mark.parametrize('case', list(CASES))
@pytest.mark.parametrize('graph', [False, True])
def test_stock_chain_matches_generic_path(case, graph):
    kw = CASES[case]
    stock, generic = _agent(kw, True, graph), _agent(kw, False, False)
    assert stock._stock_c_only() and not generic._stock_c_only()
    generic._params.flat.copy_(stock._params.flat)
    generic._target_params.flat.copy_(stock._target_params.flat)
    rng = np.random.default_rng(3)
    for T in (90, 61, 130, 77, 45, 150):
        ep = pu.synthetic_episode(rng, [(6,)], [], 2, (0,), T)
        stock.put_episode(**ep)
        generic.put_episode(**ep)
    for step in range(5):
        stock.train()
        generic.train()
        assert torch.equal(stock.replay_buffer._ids, generic.replay_buffer._ids), f'step {step}: sampled ids differ'
    torch.cuda.synchronize()
    np.testing.assert_allclose(stock._params.flat.cpu().numpy(), generic._params.flat.cpu().numpy(),
                               rtol=2e-3, atol=3e-5)
    np.testing.assert_allclose(stock._target_params.flat.cpu().numpy(), generic._target_params.flat.cpu().numpy(),
                               rtol=2e-3, atol=3e-5)
    np.testing.assert_allclose(stock.replay_buffer._tree.cpu().numpy(), generic.replay_buffer._tree.cpu().numpy(),
                               rtol=2e-3, atol=1e-5)
    np.testing.assert_allclose(stock.replay_buffer._columns['mu_prob'].cpu().numpy(),
                               generic.replay_buffer._columns['mu_prob'].cpu().numpy(), rtol=5e-3, atol=1e-5)
    np.testing.assert_allclose(stock._stats['loss_q'].item(), generic._stats['loss_q'].item(), rtol=1e-3)
    np.testing.assert_allclose(stock.log_c_alpha.item(), generic.log_c_alpha.item(), rtol=1e-4, atol=1e-6)
    stock.replay_buffer.check_health()
    stock.close()
    generic.close()


def test_fused_visual_encoder_matches_module_path(monkeypatch):
    """The convolution stack + its ResBlock head and the plugin's Linear + tanh state head as fused launches
    (`asac_conv2_*`, wide-input `asac_mlp_*`, `asac_linear_tanh_*`) against the same learner running those layers as
    PyTorch modules (MIOpen / hipBLASLt): same episodes, same noise, a few train steps with a trainable image
    representation."""
    import asac_amd  # noqa: F401
    from asac_amd import native
    from algorithm import fused_conv, fused_linear, fused_mlp
    from algorithm.sac_base import SAC_Base
    from tests.plugins import nn_conv

    def agent():
        torch.manual_seed(11)
        return SAC_Base(['vector', 'image'], [(10,), (3, 30, 30)], [], 4, None, nn_conv, device='cuda:0', batch_size=48,
                        n_step=3, burn_in_step=2, ensemble_q_num=2, ensemble_q_sample=2,
                        replay_config={'capacity': 1024}, hip_config={'use_graph': False})

    fused = agent()
    monkeypatch.setattr(fused_mlp, 'FUSED_DENSE', False)
    monkeypatch.setattr(fused_linear, 'FUSED_LINEAR_TANH', False)
    monkeypatch.setattr(fused_conv, 'conv_stack_desc', lambda *a, **k: None)
    plain = agent()
    plain._params.flat.copy_(fused._params.flat)
    plain._target_params.flat.copy_(fused._target_params.flat)
    rng = np.random.default_rng(5)
    episodes = [pu.synthetic_episode(rng, [(10,), (3, 30, 30)], [], 4, (0,), T) for T in (60, 45, 80, 70)]
    for ep in episodes:
        plain.put_episode(**ep)
    with native.LaunchProfiler() as prof:
        for _ in range(3):
            plain.train()
    assert 'asac_conv2_forward' not in prof.summary() and 'asac_linear_tanh_forward' not in prof.summary()
    monkeypatch.undo()
    for ep in episodes:
        fused.put_episode(**ep)
    with native.LaunchProfiler() as prof:
        for _ in range(3):
            fused.train()
    seen = prof.summary()
    # (per step: the online and the target pass over the positions behind the burn-in, the pass over the window under the
    # updated representation; with n + 1 = 4 positions the online pass is the differentiable one)
    assert seen['asac_conv2_forward']['calls'] + seen.get('asac_conv2_forward_windows', {'calls': 0})['calls'] == 9
    assert seen.get('asac_conv2_backward', {'calls': 0})['calls'] + seen.get('asac_conv2_backward_windows', {'calls': 0})['calls'] == 3
    assert seen['asac_linear_tanh_forward2']['calls'] == 9 and seen['asac_linear_tanh_backward2']['calls'] == 3
    torch.cuda.synchronize()
    assert torch.equal(fused.replay_buffer._ids, plain.replay_buffer._ids)
    np.testing.assert_allclose(fused._params.flat.cpu().numpy(), plain._params.flat.cpu().numpy(), rtol=3e-3, atol=5e-5)
    np.testing.assert_allclose(fused.replay_buffer._tree.cpu().numpy(), plain.replay_buffer._tree.cpu().numpy(),
                               rtol=3e-3, atol=2e-5)
    fused.close()
    plain.close()
