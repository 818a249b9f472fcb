"""GPU: every BASELINE.json configuration at its REAL size (bench.CONFIGS: batch 256 / 256 / 512 / 1024, capacity
2^19 for the vector configurations, window 81 for the RNN one, 4 608 / 9 216 frames per representation pass for the
image ones) against the CPU oracle, through the eager step AND the captured hipGraph.

Protocol: product learner and `oracle.sac_ref.SacRef` start from identical weights, episodes and tree bytes.  The
product draws its own numbers on the device (Philox, the graph-capturable source the bench runs with); after each
step the draws are read back from the step's static buffers and replayed on the oracle, so both sides consume the
same uniforms, Gaussians and ensemble subsets.  Step 0 runs eager, step 1 is the capture + first replay, step 2 a
replay.  After a step has been compared, the oracle's replay state (tree bytes, written-back probabilities and
hidden states) is set to the product's: index selection is a discontinuous function of the priorities, so a
1e-4 difference in one td-error would otherwise move a later stratum boundary across a sampled value — every
step's sampling is compared from identical replay contents, its results within the float tolerances.  Required: PER ids bit-exact, IS weights to 2e-6, losses / td-errors / priorities within the fp32
tolerances of tests/test_sac_step_gpu.py.  This covers what the small goldens cannot: the multi-workgroup return
kernel, the fused sampler at 256..1024 strata over 19 tree levels, convolution group tails at 4 608 and 9 216
frames, the GRU at 256 x 81."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import bench  # noqa: E402
from oracle import sac_ref  # noqa: E402
from tests import parity_utils as pu  # noqa: E402

FILL = {'cfg1': 2 ** 15, 'cfg2': 2 ** 15, 'cfg3': 2 ** 14, 'cfg3_h64': 2 ** 13, 'cfg4': 4096, 'cfg4_84': 2048, 'cfg5': 4096, 'cfg_attn_h64': 4096}
# observable -> (rtol, atol), set from the observed errors (profiles/r03_parity_errors.json; <= 4x the worst seen)
TOL = {'is_weights': (2e-6, 0.), 'loss_q': (2e-4, 0.), 'loss_curiosity': (2e-4, 0.), 'td_error': (2e-4, 5e-5),
       'tree': (2e-4, 1e-5), 'mu_prob': (5e-3, 1e-6), 'hidden': (2e-4, 5e-5), 'log_c_alpha': (2e-4, 0.)}
TOL_TRAINED_REP = {'loss_q': (1e-3, 0.), 'loss_curiosity': (1e-3, 0.), 'td_error': (1e-3, 5e-5), 'tree': (1e-3, 1e-5),
                   'hidden': (1e-3, 5e-5)}
SUBSET_ROWS = ('y_cn', 'y_cnext', 'pi_c', 'td_cn', 'td_cnext')      # the oracle's consumption order (continuous head)


def _full_perm(subset, E):
    """a permutation of range(E) whose first entries are `subset` (the reference keeps `randperm(E)[:E_sample]`)"""
    subset = [int(x) for x in subset]
    return np.array(subset + [e for e in range(E) if e not in subset], dtype=np.int64)


def _episode(rng, cfg, T):
    A = cfg['c_action_size']
    return dict(ep_indexes=np.arange(T, dtype=np.int32)[None],
                ep_obses_list=[rng.standard_normal((1, T, *s)).astype(np.float32) for s in cfg['obs_shapes']],
                ep_actions=rng.random((1, T, A)).astype(np.float32),
                ep_rewards=rng.standard_normal((1, T)).astype(np.float32),
                ep_dones=(rng.random((1, T)) < 0.5),
                ep_probs=rng.random((1, T, A)).astype(np.float32),
                ep_pre_seq_hidden_states=rng.standard_normal((1, T, *cfg['hidden'])).astype(np.float32))


@pytest.mark.parametrize('name', ['cfg1', 'cfg2', 'cfg3', 'cfg3_h64', 'cfg4', 'cfg4_84', 'cfg5', 'cfg_attn_h64'])
def test_baseline_config_full_size_vs_oracle(name):
    import asac_amd  # noqa: F401
    from algorithm.sac_base import SAC_Base
    from algorithm.utils.enums import CURIOSITY, SEQ_ENCODER
    cfg = bench.CONFIGS[name]
    plugin = pu.plugin(cfg['plugin'])
    B, n, A, E = cfg['batch_size'], cfg['n_step'], cfg['c_action_size'], cfg['ensemble_q_num']
    common = dict(n_step=n, burn_in_step=cfg['burn_in_step'], batch_size=B, ensemble_q_num=E,
                  ensemble_q_sample=cfg['ensemble_q_sample'], use_priority=cfg.get('use_priority', True),
                  use_prediction=cfg.get('use_prediction', False), replay_config={'capacity': cfg['capacity']})
    torch.manual_seed(0)
    agent = SAC_Base(cfg['obs_names'], cfg['obs_shapes'], [], A, None, plugin, device='cuda:0',
                     seq_encoder=SEQ_ENCODER[cfg['seq_encoder']] if cfg['seq_encoder'] else None,
                     curiosity=CURIOSITY[cfg['curiosity']] if cfg.get('curiosity') else None,
                     hip_config={'use_graph': not os.environ.get('ASAC_TEST_EAGER'), 'graph_warmup': 1,
                                 **json.loads(os.environ.get('ASAC_TEST_HIP_CONFIG', '{}'))}, **common)
    oracle = sac_ref.SacRef(cfg['obs_names'], cfg['obs_shapes'], [], A, plugin, seq_encoder=cfg['seq_encoder'],
                            curiosity=cfg.get('curiosity'), **common)
    pu.copy_weights_to_oracle(agent, oracle)
    extra = (['model_forward_dynamic'] if cfg.get('curiosity') else []) + \
        (['model_transition', 'model_reward', 'model_observation'] if cfg.get('use_prediction') else [])
    for mname in extra:
        getattr(oracle, mname).load_state_dict({k: v.detach().cpu().clone() for k, v in getattr(agent, mname).state_dict().items()})

    rng = np.random.default_rng(7)
    T = cfg['episode_len']
    for _ in range(FILL[name] // T):
        ep = _episode(rng, cfg, T)
        agent.put_episode(**ep)
        oracle.put_episode(**ep)
    rb, orb = agent.replay_buffer, oracle.replay_buffer
    # non-uniform priorities: one |N(0,1)| update pass on the device; the oracle starts from the same tree bytes
    ids = torch.arange(rb.size, device=rb.device, dtype=torch.int64)
    td = torch.from_numpy(np.abs(rng.standard_normal(rb.size)).astype(np.float32)).to(rb.device)
    for s in range(0, rb.size, 4096):
        rb.update(ids[s:s + 4096], td[s:s + 4096])
    last = ids[T - 1::T]
    rb._update_ids(last, torch.zeros(last.numel(), device=rb.device), stale_check=False, mode=1)
    orb.tree.tree[:] = rb._tree.cpu().numpy()
    assert np.array_equal(orb.storage.columns['_id'], rb._slot_ids.cpu().numpy())

    trained_rep = agent.optimizer_rep is not None

    def chk(observable, got, want):
        rt, at = (TOL_TRAINED_REP if trained_rep else TOL).get(observable, TOL[observable])
        pu.check(f'full_size/{name}/{observable}', got, want, rt, at)

    tree_before = rb._tree.clone()
    for step in range(3):
        agent.train()
        torch.cuda.synchronize()
        assert os.environ.get('ASAC_TEST_EAGER') or (agent._graph is not None) == (step >= 1), 'step 0 eager, step 1 captures and replays'
        # the step's draws, read back from its static buffers
        u = [rb._u.cpu().numpy()]
        eps = [b.cpu().numpy().copy() for b in (agent._eps_y, agent._eps_pi, agent._eps_alpha, agent._eps_td)]
        perm = [_full_perm(agent._subsets[k].cpu().numpy(), E) for k in SUBSET_ROWS]
        if not cfg.get('use_priority', True):
            # no TD error is formed (reference sac_base.py:2571-2584): its draws stay unconsumed on both sides
            eps, perm = eps[:3], perm[:3]
        oracle.noise = sac_ref.RecordedNoise(u, eps, perm)
        out = oracle.train()
        assert not oracle.noise.eps and not oracle.noise.perm and not oracle.noise.u
        assert np.array_equal(rb._ids.cpu().numpy(), out['ids']), f'{name} step {step}: PER index selection differs in {int((rb._ids.cpu().numpy() != out["ids"]).sum())} of {B} rows'
        if cfg.get('use_priority', True):
            chk('is_weights', rb._w.cpu().numpy()[:, None], out['is_weights'])
        print(f'{name} step {step}: loss_q {agent._stats["loss_q"].item():.6f} / {float(out["loss_q"]):.6f}')
        chk('loss_q', agent._stats['loss_q'].item(), float(out['loss_q']))
        if cfg.get('curiosity'):
            chk('loss_curiosity', agent._stats['loss_curiosity'].item(), float(out['loss_curiosity']))
        if cfg.get('use_prediction'):
            # the prediction models after their Adam step (sign-like updates: entries move by lr per step, so the weights
            # themselves are compared) and, through the next steps' losses / td-errors, the gated representation update
            for mname in ('model_transition', 'model_reward', 'model_observation'):
                for (k, v), vo in zip(getattr(agent, mname).state_dict().items(), getattr(oracle, mname).state_dict().values()):
                    pu.check(f'full_size/{name}/prediction_weights', v, vo, rtol=0., atol=2.2 * 3e-4 * (step + 1))
            assert out['rpm'] is not None and np.isfinite(out['rpm']['losses'].numpy()).all()
        if trained_rep:     # diagnostics: where the two learners' weights stand after this step's Adam updates (units of lr)
            mods = {'model_rep': agent.model_rep, 'model_policy': agent.model_policy, 'model_target_rep': agent.model_target_rep,
                    **{f'model_q_{i}': q for i, q in enumerate(agent.model_q_list)},
                    **{f'model_target_q_{i}': q for i, q in enumerate(agent.model_target_q_list)}}
            omods = oracle.named_modules()
            worst = []
            for mn, mod in mods.items():
                for (k, v), vo in zip(mod.state_dict().items(), omods[mn].state_dict().values()):
                    dv = (v.detach().cpu() - vo).abs()
                    worst.append((float(dv.max()) / 3e-4, int((dv > 1.5e-4).sum()), dv.numel(), f'{mn}.{k}'))
            worst.sort(reverse=True)
            print(f'{name} step {step}: weights vs oracle, max|diff|/lr (entries off by > lr/2 of n): ' +
                  '; '.join(f'{w[3]} {w[0]:.2f} ({w[1]}/{w[2]})' for w in worst[:40] if w[0] > 0.02))
        if cfg.get('use_priority', True):
            _t, _o = agent._td_error.cpu().numpy(), out['td_error'].reshape(-1)
            print(f'{name} step {step}: td max|diff| {np.abs(_t - _o).max():.3e} of {np.abs(_o).max():.3f}; log_alpha {agent.log_c_alpha.item():.6f} / {oracle.log_c_alpha.item():.6f}')
            if os.environ.get('ASAC_TEST_DUMP'):
                os.makedirs('gpurun_out', exist_ok=True)
                np.savez(f'gpurun_out/td_{name}_{step}_{os.environ["ASAC_TEST_DUMP"]}.npz', t=_t, o=_o, ids=out['ids'],
                         **{f'w_{mn}.{k}': v.detach().cpu().numpy() for mn, mod in mods.items() for k, v in mod.state_dict().items()},
                         hid=rb._columns['pre_seq_hidden_state'].cpu().numpy(), mu=rb._columns['mu_prob'].cpu().numpy(),
                         u=u[0], **{f'eps{i}': e for i, e in enumerate(eps)}, **{f'perm{i}': e for i, e in enumerate(perm)})
            if trained_rep:
                _bad = np.argsort(-np.abs(_t - _o))[:8]
                print(f'{name} step {step}: worst td rows: ' + ', '.join(
                    f'id%T {int(out["ids"][i]) % T} {_t[i]:.4f}/{_o[i]:.4f}' for i in _bad))
            chk('td_error', agent._td_error.cpu().numpy(), out['td_error'].reshape(-1))
        else:       # the tree is frozen: sampled, never updated (reference sac_base.py:2571-2584)
            assert torch.equal(rb._tree, tree_before)
        chk('tree', rb._tree.cpu().numpy(), orb.tree.tree)
        chk('mu_prob', rb._columns['mu_prob'].cpu().numpy(), orb.storage.columns['mu_prob'])
        if rb._columns['pre_seq_hidden_state'].shape[-1]:
            chk('hidden', rb._columns['pre_seq_hidden_state'].cpu().numpy(), orb.storage.columns['pre_seq_hidden_state'])
        chk('log_c_alpha', agent.log_c_alpha.item(), oracle.log_c_alpha.item())
        orb.tree.tree[:] = rb._tree.cpu().numpy()
        for key in ('mu_prob', 'pre_seq_hidden_state'):
            orb.storage.columns[key][...] = rb._columns[key].cpu().numpy()
    rb.check_health()
    assert rb.check_tree_invariant() == 0
    agent.close()
