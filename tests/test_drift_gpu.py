"""GPU: how far product and oracle drift apart when NOTHING re-aligns them.

220 (cfg2) / 110 (cfg4) consecutive train steps of a BASELINE configuration at its real size: the product draws on the device, the draws of
every step are read back and replayed on the CPU oracle (`oracle.sac_ref.SacRef`), and the oracle's replay state (tree,
written-back probabilities / hidden states, weights) is NEVER set to the product's after step 0's common start.  PER
index selection is a discontinuous function of priorities that differ at rounding level, so at some step a stratum
boundary crosses a sampled value and the two sides train on different rows from there on.  Recorded per configuration
(gpurun_out/drift_<cfg>.json, committed as profiles/r04_drift.json): the step of the first id mismatch, the fraction of
differing ids per step afterwards, and how far the losses / TD errors / temperature of the two runs are apart as
distributions over the last 100 steps.  Asserted: ids agree for the first steps; both runs stay finite; the
distribution distances stay within bounds set from the recorded run (4x observed)."""
import json
import os
from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import bench  # noqa: E402
from oracle import sac_ref  # noqa: E402
from tests import parity_utils as pu  # noqa: E402
from tests.test_full_size_gpu import SUBSET_ROWS, _episode, _full_perm  # noqa: E402

STEPS = {'cfg2': 220, 'cfg4': 110}      # (round 4: 500 / 300; shortened to keep the GPU suite inside its time budget)
LONG_STEPS = {'cfg2': 500, 'cfg4': 300}  # the long variant: ASAC_LONG_DRIFT=1 (run once per round, record under profiles/)
FILL = {'cfg2': 2 ** 15, 'cfg4': 4096}
# (relative difference of the means over the last 100 steps): loss_q, mean |td|, log alpha
BOUNDS = {'cfg2': (0.25, 0.25, 0.05), 'cfg4': (0.25, 0.25, 0.05)}


@pytest.mark.slow
@pytest.mark.skipif(not os.environ.get('ASAC_LONG_DRIFT'), reason='long drift run: set ASAC_LONG_DRIFT=1')
@pytest.mark.parametrize('name', ['cfg2', 'cfg4'])
def test_unaligned_drift_at_full_size_long(name):
    test_unaligned_drift_at_full_size(name, LONG_STEPS[name], '_long')


@pytest.mark.parametrize('name', ['cfg2', 'cfg4'])
def test_unaligned_drift_at_full_size(name, n_steps=None, tag=''):
    import asac_amd  # noqa: F401
    from algorithm.sac_base import SAC_Base
    cfg = bench.CONFIGS[name]
    plugin = pu.plugin(cfg['plugin'])
    B, n, A, E = cfg['batch_size'], cfg['n_step'], cfg['c_action_size'], cfg['ensemble_q_num']
    common = dict(n_step=n, burn_in_step=cfg['burn_in_step'], batch_size=B, ensemble_q_num=E,
                  ensemble_q_sample=cfg['ensemble_q_sample'], replay_config={'capacity': cfg['capacity']})
    torch.manual_seed(0)
    agent = SAC_Base(cfg['obs_names'], cfg['obs_shapes'], [], A, None, plugin, device='cuda:0',
                     hip_config={'use_graph': True, 'graph_warmup': 1}, **common)
    oracle = sac_ref.SacRef(cfg['obs_names'], cfg['obs_shapes'], [], A, plugin, **common)
    pu.copy_weights_to_oracle(agent, oracle)
    rng = np.random.default_rng(21)
    T = cfg['episode_len']
    for _ in range(FILL[name] // T):
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
    orb.tree.tree[:] = rb._tree.cpu().numpy()          # the common start; nothing is copied after this line

    n_steps = STEPS[name] if n_steps is None else n_steps
    first_mismatch, differing = None, []
    series = {k: ([], []) for k in ('loss_q', 'td_abs_mean', 'log_alpha')}
    for step in range(n_steps):
        agent.train()
        torch.cuda.synchronize()
        u = [rb._u.cpu().numpy()]
        eps = [b.cpu().numpy().copy() for b in (agent._eps_y, agent._eps_pi, agent._eps_alpha, agent._eps_td)]
        perm = [_full_perm(agent._subsets[k].cpu().numpy(), E) for k in SUBSET_ROWS]
        oracle.noise = sac_ref.RecordedNoise(u, eps, perm)
        out = oracle.train()
        got = rb._ids.cpu().numpy()
        frac = float((got != out['ids']).mean())
        differing.append(frac)
        if frac > 0 and first_mismatch is None:
            first_mismatch = step
        series['loss_q'][0].append(agent._stats['loss_q'].item())
        series['loss_q'][1].append(float(out['loss_q']))
        series['td_abs_mean'][0].append(float(agent._td_error.abs().mean()))
        series['td_abs_mean'][1].append(float(np.abs(out['td_error']).mean()))
        series['log_alpha'][0].append(agent.log_c_alpha.item())
        series['log_alpha'][1].append(oracle.log_c_alpha.item())
    rb.check_health()
    assert rb.check_tree_invariant() == 0

    tail = slice(n_steps - 100, n_steps)
    dist = {}
    for k, (p, o) in series.items():
        p, o = np.asarray(p), np.asarray(o)
        assert np.isfinite(p).all() and np.isfinite(o).all(), k
        mp, mo = p[tail].mean(), o[tail].mean()
        dist[k] = {'product_mean_last100': float(mp), 'oracle_mean_last100': float(mo),
                   'rel_diff_of_means': float(abs(mp - mo) / max(abs(mo), 1e-12)),
                   'product_std_last100': float(p[tail].std()), 'oracle_std_last100': float(o[tail].std()),
                   'max_rel_diff_while_ids_agree': float(np.max(np.abs(p[:first_mismatch] - o[:first_mismatch])
                                                                 / np.maximum(np.abs(o[:first_mismatch]), 1e-12)))
                   if first_mismatch != 0 else None}
    after = differing[first_mismatch:] if first_mismatch is not None else []
    record = {'config': cfg['desc'], 'steps': n_steps, 'batch': B, 'rows_resident': FILL[name],
              'first_step_with_an_id_mismatch': first_mismatch,
              'mean_fraction_of_differing_ids_after_it': float(np.mean(after)) if after else 0.0,
              'fraction_of_differing_ids_last_step': differing[-1],
              'fraction_differing_by_step_every_25': [round(x, 4) for x in differing[::25]],
              'observables': dist}
    out_dir = Path(__file__).resolve().parents[1] / 'gpurun_out'
    out_dir.mkdir(exist_ok=True)
    (out_dir / f'drift_{name}{tag}.json').write_text(json.dumps(record, indent=1))
    print(json.dumps(record))
    # The runs separate when a rounding difference of one priority moves a stratum boundary across a sampled value: WHICH
    # step that is is a matter of the kernels' summation orders (cfg4: step 20 in round 5, step 2 after the convolution
    # forward's reduction order changed in round 6).  What must hold: the first draw — from identical trees — is the
    # oracle's; the first mismatch is a boundary crossing (a few samples), not a systematic difference; and while the ids
    # agree the two runs compute the same numbers.
    assert first_mismatch is None or first_mismatch >= 1, 'the first draw comes from identical trees'
    if first_mismatch is not None:
        assert differing[first_mismatch] <= 0.05, f'{differing[first_mismatch]:.3f} of the ids differ at the first mismatch'
        for k, v in dist.items():
            assert v['max_rel_diff_while_ids_agree'] is None or v['max_rel_diff_while_ids_agree'] < 1e-4, (k, v)
    b_loss, b_td, b_alpha = BOUNDS[name]
    assert dist['loss_q']['rel_diff_of_means'] < b_loss
    assert dist['td_abs_mean']['rel_diff_of_means'] < b_td
    assert abs(dist['log_alpha']['product_mean_last100'] - dist['log_alpha']['oracle_mean_last100']) < b_alpha * abs(dist['log_alpha']['oracle_mean_last100'])
    agent.close()
