"""CPU: pin the oracle (`oracle/`) against golden vectors minted from the imported reference
(`tests/golden/make_golden.py`).  Integer / index / tree results must be bit-exact; float chains
use the tolerances stated inline."""
import numpy as np
import pytest
import torch

from oracle import sac_ref
from oracle.per_ref import PrioritizedReplayRef, SumTreeRef
from tests import parity_utils as pu


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('tag,C', [('c16', 16), ('c1024', 1024), ('c524288', 2 ** 19)])
def test_f1_sumtree_bit_exact(golden_dir, tag, C):
    g = np.load(golden_dir / 'f1_sumtree.npz')
    t = SumTreeRef(C)
    t.update(g[f'{tag}_idx1'], g[f'{tag}_p1'])
    t.update(g[f'{tag}_idx2'], g[f'{tag}_p2'])
    if C <= 1024:
        assert np.array_equal(t.tree.view(np.uint32), g[f'{tag}_tree'].view(np.uint32))
    else:
        assert np.array_equal(t.tree[:4095].view(np.uint32), g[f'{tag}_tree_top'].view(np.uint32))
        assert np.bitwise_xor.reduce(t.tree.view(np.uint32)) == g[f'{tag}_tree_xor']
        assert t.tree.astype(np.float64).sum() == g[f'{tag}_tree_sum64']
    B = int(g[f'{tag}_batch'])
    for u_key, leaf_key, p_key in [('u', 'leaf', 'p'), ('ub', 'leaf_b', 'p_b')]:
        leaf, p = t.sample(B, g[f'{tag}_{u_key}'])
        assert leaf.dtype == np.int32
        assert np.array_equal(leaf, g[f'{tag}_{leaf_key}'])
        assert np.array_equal(p.view(np.uint32), g[f'{tag}_{p_key}'].view(np.uint32))
    assert t.leaf_max() == g[f'{tag}_max']


def test_f2_per_session_bit_exact(golden_dir):
    g = np.load(golden_dir / 'f2_per.npz')
    B, prev_n, post_n, C = (int(x) for x in g['config'])
    rb = PrioritizedReplayRef(B, prev_n, post_n, capacity=C)
    for step, op in enumerate(g['script']):
        op = str(op)
        if op.startswith('add'):
            ep = {k[len(f's{step}_add_'):]: g[k] for k in g.files if k.startswith(f's{step}_add_')}
            ep = {k: ep[k] for k in ('index', 'obs_vec', 'reward', 'done', 'mu_prob')}
            rb.add(ep, ignore_size=1)
            assert np.array_equal(rb.tree.tree.view(np.uint32), g[f's{step}_tree'].view(np.uint32))
            assert np.array_equal(rb.storage.columns['_id'], g[f's{step}_ids'])
        elif op == 'sample:none':
            assert rb.sample(np.zeros(B)) is None
        elif op == 'sample':
            ids, win, w = rb.sample(g[f's{step}_u'])
            assert np.array_equal(ids, g[f's{step}_sample_ids'])
            assert np.array_equal(w.view(np.uint32), g[f's{step}_w'].view(np.uint32))
            assert float(rb.beta) == float(g[f's{step}_beta'])
            for k, v in win.items():
                assert np.array_equal(v, g[f's{step}_win_{k}']), k
        elif op == 'update':
            rb.update(g[f's{step}_upd_ids'], g[f's{step}_td'])
            assert np.array_equal(rb.tree.tree.view(np.uint32), g[f's{step}_tree'].view(np.uint32))
        elif op == 'update_transitions':
            rb.update_transitions(g[f's{step}_ut_ids'], 'mu_prob', g[f's{step}_ut_data'])
            assert np.array_equal(rb.storage.columns['mu_prob'], g[f's{step}_mu_prob'])
        else:
            raise AssertionError(op)


@pytest.mark.parametrize('n', [1, 4, 40])
@pytest.mark.parametrize('use_is', [True, False])
def test_f3_vtrace(golden_dir, n, use_is):
    g = np.load(golden_dir / 'f3_vtrace.npz')
    gamma, lam, rho, c = g['params']
    tag = f'n{n}_is{int(use_is)}'
    t = lambda k: torch.from_numpy(g[f'{tag}_{k}'].copy())  # noqa: E731
    y = sac_ref.v_trace(gamma=float(gamma), gamma_ratio=t('gamma_ratio'), lambda_ratio=t('lambda_ratio'),
                        v_rho=torch.tensor(float(rho)), v_c=torch.tensor(float(c)), use_n_step_is=use_is,
                        n_last_masks=t('n_last_masks'), n_padding_masks=t('n_padding_masks'),
                        n_rewards=t('n_rewards'), n_dones=t('n_dones'), n_mu_probs=t('n_mu_probs'),
                        n_pi_probs=t('n_pi_probs'), n_vs=t('n_vs'), next_n_vs=t('next_n_vs'))
    # same eager ops on the same host: identical bits
    assert np.array_equal(y.numpy().view(np.uint32), g[f'{tag}_y'].view(np.uint32))


# ------------------------------------------------------------------------------------------------
def _load_weights(agent, g, prefix):
    for name, mod in agent.named_modules().items():
        sd = {k[len(f'{prefix}/{name}/'):]: torch.from_numpy(g[k].copy())
              for k in g.files if k.startswith(f'{prefix}/{name}/')}
        if not sd:
            assert not list(mod.state_dict()), name   # parameter-free module (e.g. ModelSimpleRep)
            continue
        mod.load_state_dict(sd)


# same eager ops on the same host give identical bits; these cases run other kernels for the same math: the GRU
# un-packed (nn_models/layers/seq_layers.py), attention heads as batched GEMMs instead of chunk / cat
INEXACT = ('cfg3', 'attn', 'attn_tanh', 'conv_attn_cur', 'rnn_h64', 'attn_h64')


@pytest.mark.parametrize('case', list(pu.STEP_CASES))
def test_f6_full_step(golden_dir, case):
    torch.set_num_threads(1)
    g = np.load(golden_dir / f'f6_step_{case}.npz')
    plugin_name, kw, d_sizes, io = pu.STEP_CASES[case]
    nn_mod = pu.plugin(plugin_name)
    agent = sac_ref.SacRef(io['obs_names'], io['obs_shapes'], list(d_sizes), io['c_action_size'], nn_mod,
                           batch_size=io['batch_size'], replay_config={'capacity': io['capacity']}, **kw)
    _load_weights(agent, g, 'w0')
    with torch.no_grad():
        agent.log_c_alpha.copy_(torch.from_numpy(g['w0/log_c_alpha']))
        agent.log_d_alpha.copy_(torch.from_numpy(g['w0/log_d_alpha']))
    for ep in pu.golden_episodes(g, len(io['obs_shapes'])):
        agent.put_episode(**ep)

    exact = case not in INEXACT
    for s in range(int(g['n_steps'])):
        eps = [g[f'step{s}/eps{j}'] for j in range(int(g[f'step{s}/n_eps']))]
        agent.noise = sac_ref.RecordedNoise(u=[g[f'step{s}/u']], eps=eps, perm=list(g[f'step{s}/perm']))
        out = agent.train()
        assert np.array_equal(out['ids'], g[f'step{s}/sample_ids']), f'step {s}: PER index selection'
        assert np.array_equal(out['is_weights'], g[f'step{s}/is_weights'])
        tol = dict(rtol=0, atol=0) if exact else dict(rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(out['loss_q'].numpy(), g[f'step{s}/loss_q'], **tol)
        np.testing.assert_allclose(out['loss_policy'].numpy(), g[f'step{s}/loss_policy'], **tol)
        if f'step{s}/c_entropy' in g.files:
            np.testing.assert_allclose(out['c_entropy'].numpy(), g[f'step{s}/c_entropy'], **tol)
        if f'step{s}/loss_curiosity' in g.files:
            np.testing.assert_allclose(out['loss_curiosity'].numpy(), g[f'step{s}/loss_curiosity'], **tol)
        if s == 0:      # the first step's gradients: Adam's first moment / (1 - beta1)
            for oname, opt in agent.named_optimizers().items():
                for j, p in enumerate(opt.param_groups[0]['params']):
                    if f'g0/{oname}/{j}' in g.files:
                        want = g[f'g0/{oname}/{j}']
                        np.testing.assert_allclose(opt.state[p]['exp_avg'].numpy(), want, rtol=tol['rtol'] * 50,
                                                   atol=0 if exact else 1e-6 * np.abs(want).max() + 1e-12,
                                                   err_msg=f'{oname}/{j}')
        if f'step{s}/td_error' in g.files:
            np.testing.assert_allclose(out['td_error'], g[f'step{s}/td_error'], **tol)
            if exact:
                assert np.array_equal(agent.replay_buffer.tree.tree.view(np.uint32),
                                      g[f'step{s}/tree'].view(np.uint32))
            else:
                np.testing.assert_allclose(agent.replay_buffer.tree.tree, g[f'step{s}/tree'], rtol=2e-5, atol=1e-6)
        np.testing.assert_allclose(agent.replay_buffer.storage.columns['mu_prob'], g[f'step{s}/mu_prob'], **tol)
        np.testing.assert_allclose(agent.replay_buffer.storage.columns['pre_seq_hidden_state'],
                                   g[f'step{s}/hidden'], **tol)
        np.testing.assert_allclose(agent.log_c_alpha.detach().numpy(), g[f'step{s}/log_c_alpha'], **tol)
        assert not agent.noise.eps and not agent.noise.perm, 'every recorded draw must be consumed'
    wtol = dict(rtol=0, atol=0) if exact else dict(rtol=1e-4, atol=1e-6)
    for name, mod in agent.named_modules().items():
        for k, v in mod.state_dict().items():
            assert f'w1/{name}/{k}' in g.files
            np.testing.assert_allclose(v.numpy(), g[f'w1/{name}/{k}'], err_msg=f'{name}/{k}', **wtol)


# ------------------------------------------------------------------------------------------------
# recurrent prediction models: `_train_rpm` + `calculate_adaptive_weights` (f11_rpm.npz: the reference's functions
# called on a fresh graph — its whole step raises with `use_prediction`)
# ------------------------------------------------------------------------------------------------
def rpm_oracle(g, tag):
    """oracle learner with the fixture's weights; -> (oracle, inputs for train_rpm)"""
    B, n, kl, extra = g[f'{tag}/cfg']
    oracle = sac_ref.SacRef(['vector'], [(6,)], [], 2, pu.plugin('nn_vec_full'), batch_size=int(B), n_step=int(n),
                            use_prediction=True, transition_kl=float(kl), use_extra_data=bool(extra),
                            replay_config={'capacity': 256})
    for name, mod in oracle.named_modules().items():
        sd = {k: torch.from_numpy(g[f'{tag}/w0/{name}/{k}'].copy()) for k in mod.state_dict() if f'{tag}/w0/{name}/{k}' in g.files}
        if sd:
            mod.load_state_dict(sd)
    return oracle


@pytest.mark.parametrize('tag', ['a', 'b', 'c'])
def test_f11_rpm_bit_exact(golden_dir, tag):
    g = np.load(golden_dir / 'f11_rpm.npz')
    torch.set_num_threads(1)
    oracle = rpm_oracle(g, tag)
    obs = [torch.from_numpy(g[f'{tag}/obs'].copy())]
    nx_states, _ = oracle.model_rep(obs, None, None)
    with torch.no_grad():
        nx_target_states, _ = oracle.model_target_rep(obs, None, None)
    assert np.array_equal(nx_states.detach().numpy(), g[f'{tag}/nx_states'])
    main = float(g[f'{tag}/flip']) * torch.mean(torch.square(torch.sum(nx_states * torch.from_numpy(g[f'{tag}/coef']), dim=-1)))
    oracle.optimizer_rep.zero_grad()
    main.backward(retain_graph=True)
    grads_main = [p.grad.detach() for p in oracle.model_rep.parameters()]
    for j, gm in enumerate(grads_main):
        assert np.array_equal(gm.numpy(), g[f'{tag}/g_main/{j}'])
    out = oracle.train_rpm(grads_main, obs, nx_states, nx_target_states, torch.from_numpy(g[f'{tag}/actions'].copy()),
                           torch.from_numpy(g[f'{tag}/rewards'].copy()))
    assert np.array_equal(out['losses'].numpy(), g[f'{tag}/losses'])
    assert np.array_equal(out['gates'].numpy(), g[f'{tag}/gate'])
    assert np.array_equal(np.array([float(out['entropy']), float(out['losses'][1]), float(out['losses'][2])], dtype=np.float32),
                          g[f'{tag}/ret'])
    for j, p in enumerate(oracle.model_rep.parameters()):
        assert np.array_equal(p.grad.numpy(), g[f'{tag}/g_rep_after/{j}']), f'representation gradient {j} after gating'
    for j, p in enumerate(oracle.prediction_parameters()):
        assert np.array_equal(p.grad.numpy(), g[f'{tag}/g_pred/{j}']), f'prediction gradient {j}'
    for name in ('model_transition', 'model_reward', 'model_observation'):
        for k, v in getattr(oracle, name).state_dict().items():
            assert np.array_equal(v.numpy(), g[f'{tag}/w1/{name}/{k}']), f'{name}/{k} after Adam'
