"""GPU parity of the whole train step: the product `SAC_Base` (HIP kernels + PyTorch-ROCm) fed
with the reference's recorded weights / episodes / random draws must reproduce the reference's
observables (golden `f6_step_*.npz`): PER ids bit-exact, float chains within fp32 tolerance
(device GEMM + libm vs host: rtol 2e-4 on losses / td-errors / priorities, 5e-4 on post-Adam weights)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests import parity_utils as pu  # noqa: E402
from tests.plugins import nn_vec  # noqa: E402

LR = 3e-4


def make_agent(case, use_graph=False):
    import asac_amd  # noqa: F401
    from algorithm.sac_base import SAC_Base
    from algorithm.utils.enums import convert_config_to_enum
    plugin_name, kw, d_sizes, io = pu.STEP_CASES[case]
    kw = dict(kw)
    convert_config_to_enum(kw)
    return SAC_Base(io['obs_names'], io['obs_shapes'], list(d_sizes), io['c_action_size'], None, pu.plugin(plugin_name),
                    device='cuda:0', batch_size=io['batch_size'], replay_config={'capacity': io['capacity']},
                    hip_config={'use_graph': use_graph}, **kw)


# Tolerances (fp32; device GEMM / MFMA accumulation order and device libm against the host):
#   observables of a step (losses, td-errors, priorities, hidden states, alpha)   rtol 2e-4
#   first-step gradients, entry by entry                                          rtol 2e-3 + 2e-5 * max|tensor|
#   weights after the steps                                                        rtol 5e-4 + atol 2e-5, except entries whose
#       reference gradient is (analytically or numerically) zero, see parity_utils.assert_weights_close
# Cases whose representation is itself trained by the step carry the rounding of one more Adam update into the
# observables taken after it (td-errors, written-back probabilities): 1e-3 there.
TRAINED_REP = ('attn', 'conv', 'conv_attn_cur')


@pytest.mark.parametrize('case', list(pu.STEP_CASES))
def test_full_step_vs_reference_golden(golden_dir, case):
    from algorithm.fused import RecordedNoise
    g = np.load(golden_dir / f'f6_step_{case}.npz')
    io = pu.STEP_CASES[case][3]
    agent = make_agent(case)
    mods = pu.load_golden_weights(agent, g)
    for ep in pu.golden_episodes(g, len(io['obs_shapes'])):
        agent.put_episode(**ep)
    rb = agent.replay_buffer
    rt = 1e-3 if case in TRAINED_REP else 2e-4
    n_steps = int(g['n_steps'])
    for s in range(n_steps):
        eps = [g[f'step{s}/eps{j}'] for j in range(int(g[f'step{s}/n_eps']))]
        agent.noise = RecordedNoise([g[f'step{s}/u']], eps, list(g[f'step{s}/perm']))
        rb.uniform_source = agent.noise
        alpha_before = agent.log_c_alpha.detach().clone()
        assert agent.train() == s + 1
        assert agent.noise.exhausted(), 'every recorded draw must be consumed, in order'
        assert np.array_equal(rb._ids.cpu().numpy(), g[f'step{s}/sample_ids']), f'step {s}: PER index selection'
        np.testing.assert_allclose(rb._w.cpu().numpy()[:, None], g[f'step{s}/is_weights'], rtol=2e-6)
        np.testing.assert_allclose(agent._stats['loss_q'].item(), g[f'step{s}/loss_q'], rtol=rt)
        # the policy objective and the entropy the reference returns from _train_policy (sac_base.py:1903-1911)
        agent._refresh_policy_stats(alpha_before)
        np.testing.assert_allclose(agent._stats['loss_policy'].item(), g[f'step{s}/loss_policy'], rtol=rt, atol=2e-5)
        if f'step{s}/c_entropy' in g.files:
            np.testing.assert_allclose(agent._stats['c_entropy'].item(), g[f'step{s}/c_entropy'], rtol=rt, atol=2e-5)
        if f'step{s}/d_entropy' in g.files:
            np.testing.assert_allclose(agent._stats['d_entropy'].item(), g[f'step{s}/d_entropy'], rtol=rt, atol=2e-5)
        if f'step{s}/loss_curiosity' in g.files:
            np.testing.assert_allclose(agent._stats['loss_curiosity'].item(), g[f'step{s}/loss_curiosity'], rtol=rt)
        if s == 0:
            pu.assert_first_step_gradients(agent, g, rtol=2e-3, atol_frac=2e-5)
        if f'step{s}/td_error' in g.files:
            np.testing.assert_allclose(agent._td_error.cpu().numpy()[:, None], g[f'step{s}/td_error'],
                                       rtol=rt, atol=2e-5)
            np.testing.assert_allclose(rb._tree.cpu().numpy(), g[f'step{s}/tree'], rtol=rt, atol=1e-6)
        # stored-action probabilities are exp() of a log-density with 1/sigma^2 gain on f32 noise of loc: 1e-3
        np.testing.assert_allclose(rb._columns['mu_prob'].cpu().numpy(), g[f'step{s}/mu_prob'],
                                   rtol=5e-3 if case in TRAINED_REP else 1e-3, atol=1e-6)
        if rb._columns['pre_seq_hidden_state'].shape[-1]:
            np.testing.assert_allclose(rb._columns['pre_seq_hidden_state'].cpu().numpy(), g[f'step{s}/hidden'],
                                       rtol=rt, atol=2e-5)
        np.testing.assert_allclose(agent.log_c_alpha.item(), g[f'step{s}/log_c_alpha'], rtol=1e-5 if case not in TRAINED_REP else 2e-4)
    if agent.curiosity is not None:
        mods['model_forward_dynamic'] = agent.model_forward_dynamic
    slack = pu.assert_weights_close(mods, g, n_steps, LR, rtol=5e-4, atol=2e-5)
    print(f'{case}: tensors with zero-gradient entries (fraction given +-lr slack): {slack}')
    rb.check_health()
    assert rb.check_tree_invariant() == 0
    agent.close()


def test_smoke_against_oracle():
    from tests.smoke_step import run_smoke
    run_smoke(steps=3, verbose=False)


@pytest.mark.parametrize('case', ['cfg2', 'cfg3'])
def test_graph_replay_matches_eager(case):
    """The captured hipGraph must do exactly what the eager step does: two learners with the same
    seed, one eager and one graph-replayed, end with identical trees and weights."""
    import random
    rng = np.random.default_rng(1)
    hidden = (2, 8) if case == 'cfg3' else (0,)
    eps_list = [pu.synthetic_episode(rng, [(6,)], [], 2, hidden, T) for T in (60, 45, 70, 80, 33)]
    results = []
    for use_graph in (False, True):
        torch.manual_seed(3), np.random.seed(3), random.seed(3)
        agent = make_agent(case, use_graph=use_graph)
        for ep in eps_list:
            agent.put_episode(**ep)
        torch.manual_seed(4)
        for _ in range(8):
            agent.train()
        assert (agent._graph is not None) == use_graph, 'graph capture must succeed for stock models'
        results.append((agent.replay_buffer._tree.cpu().numpy().copy(), agent._params.flat.cpu().numpy().copy(),
                        agent.replay_buffer._columns['mu_prob'].cpu().numpy().copy()))
        agent.close()
    for a, b in zip(*results):
        np.testing.assert_allclose(a, b, rtol=1e-5, atol=1e-6)


def test_data_parallel_path_single_rank_nccl():
    """The RCCL exchange steps (gradient mean all-reduce, global-min IS normalisation, weight
    broadcast) inside the eager AND the graph-captured step: with world_size 1 they change nothing
    (`always=True` makes the context issue them anyway), so a dist-enabled learner has to reproduce a
    plain one bit-for-bit-ish."""
    import random
    import socket
    import torch.distributed as dist
    import asac_amd  # noqa: F401
    from algorithm.parallel import DataParallelContext
    from algorithm.sac_base import SAC_Base
    with socket.socket() as s_:
        s_.bind(('127.0.0.1', 0))
        port = s_.getsockname()[1]
    dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1,
                            device_id=torch.device('cuda:0'))
    try:
        rng = np.random.default_rng(2)
        eps_list = [pu.synthetic_episode(rng, [(6,)], [], 2, (0,), T) for T in (60, 45, 70, 80, 33)]
        results = []
        for use_dist, use_graph in ((False, True), (True, False), (True, True)):
            torch.manual_seed(5), np.random.seed(5), random.seed(5)
            hip = {'use_graph': use_graph, 'dist': DataParallelContext(always=True) if use_dist else None}
            agent = SAC_Base(['vector'], [(6,)], [], 2, None, nn_vec, device='cuda:0', batch_size=32, n_step=4,
                             replay_config={'capacity': 512}, hip_config=hip)
            for ep in eps_list:
                agent.put_episode(**ep)
            torch.manual_seed(6)
            for _ in range(8):
                agent.train()
            assert (agent._graph is not None) == use_graph
            results.append((agent.replay_buffer._tree.cpu().numpy().copy(), agent._params.flat.cpu().numpy().copy()))
            agent.close()
        for other in results[1:]:
            for a, b in zip(results[0], other):
                np.testing.assert_allclose(a, b, rtol=1e-5, atol=1e-6)
    finally:
        dist.destroy_process_group()
