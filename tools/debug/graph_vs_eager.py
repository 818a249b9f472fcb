"""Which parameters differ between an eager and a graph-replayed learner after k steps?  (python tools/debug/graph_vs_eager.py cfg k)"""
import json
import os
import random
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from tests import parity_utils as pu  # noqa: E402


READS = sys.argv[3].split(',') if len(sys.argv) > 3 else []


def run(name, steps, use_graph):
    import asac_amd  # noqa: F401
    from algorithm.sac_base import SAC_Base
    from algorithm.utils.enums import CURIOSITY, SEQ_ENCODER
    cfg = bench.CONFIGS[name]
    torch.manual_seed(3), np.random.seed(3), random.seed(3)
    agent = SAC_Base(cfg['obs_names'], cfg['obs_shapes'], [], cfg['c_action_size'], None, pu.plugin(cfg['plugin']), device='cuda:0',
                     seq_encoder=SEQ_ENCODER[cfg['seq_encoder']] if cfg['seq_encoder'] else None,
                     curiosity=CURIOSITY[cfg['curiosity']] if cfg.get('curiosity') else None,
                     n_step=cfg['n_step'], burn_in_step=cfg['burn_in_step'], batch_size=cfg['batch_size'],
                     ensemble_q_num=cfg['ensemble_q_num'], ensemble_q_sample=cfg['ensemble_q_sample'],
                     use_prediction=cfg.get('use_prediction', False), replay_config={'capacity': cfg['capacity']},
                     hip_config={'use_graph': use_graph, 'graph_warmup': 1, **json.loads(os.environ.get('ASAC_TEST_HIP_CONFIG', '{}'))})
    saved = {'go': [], 'gb': [], 'gw': [], 'x': []}
    if os.environ.get('ASAC_DEBUG_HOOK'):
        mod = agent.model_rep
        for part in os.environ['ASAC_DEBUG_HOOK'].split('.'):
            mod = mod[int(part)] if part.isdigit() else getattr(mod, part)

        def fwd_hook(m_, inp, out_):
            if out_.requires_grad:
                saved['x'].append(inp[0].detach().clone())
                out_.register_hook(lambda g: saved['go'].append(g.clone()) and None)
        mod.register_forward_hook(fwd_hook)
        mod.bias.register_hook(lambda g: saved['gb'].append(g.clone()) and None)
        mod.weight.register_hook(lambda g: saved['gw'].append(g.clone()) and None)
    counts = None
    rng = np.random.default_rng(7)
    A = cfg['c_action_size']
    for _ in range(40):
        T = cfg['episode_len']
        agent.put_episode(ep_indexes=np.arange(T, dtype=np.int32)[None],
                          ep_obses_list=[rng.standard_normal((1, T, *s)).astype(np.float32) for s in cfg['obs_shapes']],
                          ep_actions=rng.random((1, T, A)).astype(np.float32), ep_rewards=rng.standard_normal((1, T)).astype(np.float32),
                          ep_dones=(rng.random((1, T)) < 0.5), ep_probs=rng.random((1, T, A)).astype(np.float32),
                          ep_pre_seq_hidden_states=rng.standard_normal((1, T, *cfg['hidden'])).astype(np.float32))
    torch.manual_seed(4)
    rb = agent.replay_buffer
    for _ in range(steps):
        agent.train()
        if counts is None:
            counts = {k: len(v) for k, v in saved.items()}
        if 'sync' in READS:
            torch.cuda.synchronize()
        if 'noise' in READS:
            rb._u.cpu().numpy()
            [b.cpu().numpy().copy() for b in (agent._eps_y, agent._eps_pi, agent._eps_alpha, agent._eps_td)]
        if 'subsets' in READS:
            [agent._subsets[k].cpu().numpy() for k in ('y_cn', 'y_cnext', 'pi_c', 'td_cn', 'td_cnext')]
        if 'item' in READS:
            agent._stats['loss_q'].item(), agent.log_c_alpha.item()
        if 'cols' in READS:
            rb._ids.cpu().numpy(), rb._w.cpu().numpy(), agent._td_error.cpu().numpy()
            rb._tree.cpu().numpy(), rb._columns['mu_prob'].cpu().numpy(), rb._columns['pre_seq_hidden_state'].cpu().numpy()
        if 'sleep' in READS:
            import time
            time.sleep(0.05)
    torch.cuda.synchronize()
    out = {}
    for mname in ('model_rep', 'model_q_list', 'model_policy'):
        mod = getattr(agent, mname)
        for i, sub in enumerate(mod if isinstance(mod, list) else [mod]):
            for n_, p in sub.named_parameters():
                out[f'{mname}{i}.{n_}'] = p.detach().cpu().numpy().copy()
    for k, v in saved.items():
        for i, t in enumerate(v[len(v) - counts[k]:]):
            out[f'_hook_{k}_{i}'] = t.cpu().numpy().copy()
    out['_td_error'] = agent._td_error.cpu().numpy().copy()
    out['_tree'] = agent.replay_buffer._tree.cpu().numpy().copy()
    for c in ('mu_prob', 'pre_seq_hidden_state'):
        out['_col_' + c] = agent.replay_buffer._columns[c].cpu().numpy().copy()
    agent.close()
    return out


if __name__ == '__main__':
    name, steps = sys.argv[1], int(sys.argv[2])
    a, b = run(name, steps, False), run(name, steps, True)
    for k in a:
        d = np.abs(a[k] - b[k]).max()
        print(f'{k:60s} max|diff| {d:.3e}  max|w| {np.abs(a[k]).max():.3e}' + ('   <<<' if d > 1e-4 else ''))
