"""debug: first-step gradient errors vs golden per tensor, fused stock MLP path vs module path"""
import sys
from pathlib import Path
import numpy as np
import torch
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import asac_amd  # noqa
from tests import parity_utils as pu
from algorithm.fused import RecordedNoise
from tests import parity_utils as _pu_h
SAC_Base = _pu_h.hooked_learner()
from algorithm.utils.enums import convert_config_to_enum

case = sys.argv[1] if len(sys.argv) > 1 else 'attn'
g = np.load(ROOT / f'tests/golden/f6_step_{case}.npz')
for fused in (True, False):
    plugin_name, kw, d_sizes, io = pu.STEP_CASES[case]
    kw = dict(kw); convert_config_to_enum(kw)
    agent = SAC_Base(io['obs_names'], io['obs_shapes'], list(d_sizes), io['c_action_size'], None, pu.plugin(plugin_name),
                     device='cuda:0', batch_size=io['batch_size'], replay_config={'capacity': io['capacity']},
                     hip_config={'use_graph': False, 'fused_mlp': fused}, **kw)
    pu.load_golden_weights(agent, g)
    for ep in pu.golden_episodes(g, len(io['obs_shapes'])):
        agent.put_episode(**ep)
    eps = [g[f'step0/eps{j}'] for j in range(int(g['step0/n_eps']))]
    agent.noise = RecordedNoise([g['step0/u']], eps, list(g['step0/perm']))
    agent.replay_buffer.uniform_source = agent.noise
    agent.after_rep_q_update = lambda: pu.load_golden_weights(agent, g, prefix='step0/w_rq')
    seen = {}
    orig = agent._policy
    agent.train()
    m = pu.product_first_moments(agent)
    if fused:
        q = agent._pi_q.view(agent.ensemble_q_num, -1).cpu().numpy()
        gap = np.abs(q[0] - q[1])
        order = np.argsort(gap)[:5]
        print('   smallest critic gaps |q0-q1|:', [(int(i), float(gap[i]), float(q[0][i])) for i in order])
    print('fused_mlp', fused)
    for key in g.files:
        if key.startswith('g0/optimizer_policy') or key.startswith('g0/optimizer_alpha'):
            _, o, j = key.split('/')
            want, got = g[key], m[o][int(j)].cpu().numpy()
            print(f'   {key}: max|want| {np.abs(want).max():.3g}  max err {np.abs(got - want).max():.3g}  rel-to-max {np.abs(got - want).max() / np.abs(want).max():.3g}')
    agent.close()
    if fused:
        pass
