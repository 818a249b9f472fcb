"""GPU: `hip_config['lookahead'] = 1` — one sampled batch in flight, the reference's own schedule (`Queue(maxsize=1)` +
prefetch thread, /root/reference/algorithm/replay_buffer.py:275, 339-396): batch k + 1 is drawn and gathered from the
tree and the rows as step k - 1 left them, while the learner trains on batch k.  The oracle (`SacRef(lookahead=True)`)
follows the same delayed-update schedule; PER ids must agree bit for bit, eager and as the two alternating hipGraphs.
Protocol otherwise that of tests/test_full_size_gpu.py (the product's draws are read back and replayed on the oracle;
after a compared step the oracle's replay state is set to the product's)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import bench  # noqa: E402
from oracle import sac_ref  # noqa: E402
from tests import parity_utils as pu  # noqa: E402
from tests.test_full_size_gpu import SUBSET_ROWS, TOL, _episode, _full_perm  # noqa: E402


@pytest.mark.parametrize('name,fill', [('cfg2', 2 ** 15), ('cfg3', 2 ** 13), ('cfg4', 2 ** 12), ('cfg5', 2 ** 12)])
def test_one_batch_in_flight_matches_the_delayed_oracle(name, fill):
    import asac_amd  # noqa: F401
    from algorithm.sac_base import SAC_Base
    from algorithm.utils.enums import CURIOSITY, SEQ_ENCODER
    cfg = bench.CONFIGS[name]
    plugin = pu.plugin(cfg['plugin'])
    B, n, A, E = cfg['batch_size'], cfg['n_step'], cfg['c_action_size'], cfg['ensemble_q_num']
    common = dict(n_step=n, burn_in_step=cfg['burn_in_step'], batch_size=B, ensemble_q_num=E,
                  ensemble_q_sample=cfg['ensemble_q_sample'], use_prediction=cfg.get('use_prediction', False),
                  replay_config={'capacity': cfg['capacity']})
    torch.manual_seed(0)
    agent = SAC_Base(cfg['obs_names'], cfg['obs_shapes'], [], A, None, plugin, device='cuda:0',
                     seq_encoder=SEQ_ENCODER[cfg['seq_encoder']] if cfg['seq_encoder'] else None,
                     curiosity=CURIOSITY[cfg['curiosity']] if cfg.get('curiosity') else None,
                     hip_config={'use_graph': True, 'graph_warmup': 2, 'lookahead': 1}, **common)
    oracle = sac_ref.SacRef(cfg['obs_names'], cfg['obs_shapes'], [], A, plugin, seq_encoder=cfg['seq_encoder'],
                            curiosity=cfg.get('curiosity'), lookahead=True, **common)
    pu.copy_weights_to_oracle(agent, oracle)
    for mname in (['model_forward_dynamic'] if cfg.get('curiosity') else []) + \
            (['model_transition', 'model_reward', 'model_observation'] if cfg.get('use_prediction') else []):
        getattr(oracle, mname).load_state_dict({k: v.detach().cpu().clone() for k, v in getattr(agent, mname).state_dict().items()})
    trained_rep = agent.optimizer_rep is not None
    rng = np.random.default_rng(11)
    T = cfg['episode_len']
    for _ in range(fill // T):
        ep = _episode(rng, cfg, T)
        agent.put_episode(**ep)
        oracle.put_episode(**ep)
    rb, orb = agent.replay_buffer, oracle.replay_buffer
    ids = torch.arange(rb.size, device=rb.device, dtype=torch.int64)
    td = torch.from_numpy(np.abs(rng.standard_normal(rb.size)).astype(np.float32)).to(rb.device)
    for s in range(0, rb.size, 4096):
        rb.update(ids[s:s + 4096], td[s:s + 4096])
    last = ids[T - 1::T]
    rb._update_ids(last, torch.zeros(last.numel(), device=rb.device), stale_check=False, mode=1)
    orb.tree.tree[:] = rb._tree.cpu().numpy()

    trained = []
    from asac_amd import native
    for step in range(6 if name == 'cfg3' else 7):      # (cfg3: ~10 s of CPU oracle per step)
        if step == 1:      # (eager) the next batch's gather: extra workgroups of the first policy / critic launch where the
            with native.LaunchProfiler(repeat=1) as prof:      # stock chain runs, a launch of its own before the write-backs otherwise
                agent.train()
            assert ('asac_window_gather_pad' in prof.summary()) == (name != 'cfg2'), sorted(prof.summary())
        else:
            agent.train()
        torch.cuda.synchronize()
        # steps 0, 1 eager; 2 and 3 capture one arrangement of the two batch sets each; 4 .. 6 replay them in turn
        assert (agent._graph is not None) == (step >= 2)
        u = [rb.next_uniforms().cpu().numpy()]
        if step == 0:      # two batches are drawn before the first step: the one it trains on, then the one in flight
            u = [rb._u.cpu().numpy()] + u
        eps = [b.cpu().numpy().copy() for b in (agent._eps_y, agent._eps_pi, agent._eps_alpha, agent._eps_td)]
        perm = [_full_perm(agent._subsets[k].cpu().numpy(), E) for k in SUBSET_ROWS]
        oracle.noise = sac_ref.RecordedNoise(u, eps, perm)
        out = oracle.train()
        assert not oracle.noise.eps and not oracle.noise.perm and not oracle.noise.u
        got = rb._ids.cpu().numpy()
        assert np.array_equal(got, out['ids']), f'{name} step {step}: {int((got != out["ids"]).sum())} of {B} ids differ'
        trained.append(got.copy())
        pu.check(f'lookahead/{name}/is_weights', rb._w.cpu().numpy()[:, None], out['is_weights'], *TOL['is_weights'])
        pu.check(f'lookahead/{name}/loss_q', agent._stats['loss_q'].item(), float(out['loss_q']), 1e-3, 0.)
        if cfg.get('curiosity'):
            pu.check(f'lookahead/{name}/loss_curiosity', agent._stats['loss_curiosity'].item(), float(out['loss_curiosity']), 1e-3, 0.)
        pu.check(f'lookahead/{name}/td_error', agent._td_error.cpu().numpy(), out['td_error'].reshape(-1), 1e-3, 5e-5)
        pu.check(f'lookahead/{name}/tree', rb._tree.cpu().numpy(), orb.tree.tree, 1e-3, 1e-5)
        # the batch in flight is the oracle's queued one: ids now, everything else when it is trained on
        assert np.array_equal(rb._alt['_ids'].cpu().numpy(), oracle._queued[0])
        orb.tree.tree[:] = rb._tree.cpu().numpy()
        for key in ('mu_prob', 'pre_seq_hidden_state'):
            orb.storage.columns[key][...] = rb._columns[key].cpu().numpy()
    # the schedule is really the delayed one: a synchronous sampler would have drawn other batches from the same uniforms
    assert any(not np.array_equal(trained[i], trained[i + 1]) for i in range(len(trained) - 1))
    rb.check_health()
    assert rb.check_tree_invariant() == 0
    agent.close()
