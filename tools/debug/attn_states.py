"""debug: product vs oracle representation outputs per pass of one golden `attn` step"""
import sys
from pathlib import Path
import numpy as np
import torch
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import asac_amd  # noqa
from oracle import sac_ref
from tests import parity_utils as pu
from tests.test_sac_step_gpu import make_agent
from algorithm.fused import RecordedNoise

case = sys.argv[1] if len(sys.argv) > 1 else 'attn'
g = np.load(ROOT / f'tests/golden/f6_step_{case}.npz')
plugin_name, kw, d_sizes, io = pu.STEP_CASES[case]
agent = make_agent(case)
mods = pu.load_golden_weights(agent, g)
oracle = sac_ref.SacRef(io['obs_names'], io['obs_shapes'], list(d_sizes), io['c_action_size'], pu.plugin(plugin_name),
                        batch_size=io['batch_size'], replay_config={'capacity': io['capacity']}, **kw)
for name, mod in oracle.named_modules().items():
    sd = {k[len(f'w0/{name}/'):]: torch.from_numpy(g[k].copy()) for k in g.files if k.startswith(f'w0/{name}/')}
    if sd:
        mod.load_state_dict(sd)
with torch.no_grad():
    oracle.log_c_alpha.copy_(torch.from_numpy(g['w0/log_c_alpha']))
for ep in pu.golden_episodes(g, len(io['obs_shapes'])):
    agent.put_episode(**ep)
    oracle.put_episode(**ep)
eps = [g[f'step0/eps{j}'] for j in range(int(g['step0/n_eps']))]
agent.noise = RecordedNoise([g['step0/u']], eps, list(g['step0/perm']))
agent.replay_buffer.uniform_source = agent.noise
oracle.noise = sac_ref.RecordedNoise(u=[g['step0/u']], eps=eps, perm=list(g['step0/perm']))

rec_p, rec_o = [], []
orig = agent.get_l_states
def gl(*a, **k):
    r = orig(*a, **k)
    rec_p.append((r[0].detach().cpu().numpy().copy(), r[1].detach().cpu().numpy().copy(), a[1].cpu().numpy().copy()))
    return r
agent.get_l_states = gl
orig_o = oracle.l_states
def ol(*a, **k):
    r = orig_o(*a, **k)
    rec_o.append((r[0].detach().numpy().copy(), r[1].detach().numpy().copy()))
    return r
oracle.l_states = ol
agent.after_rep_q_update = lambda: pu.load_golden_weights(agent, g, prefix='step0/w_rq')
agent.train()
oracle.train()
# oracle's weights after its own update vs golden
for name in ('model_rep',):
    for k, v in oracle.named_modules()[name].state_dict().items():
        d = np.abs(v.numpy() - g[f'step0/w_rq/{name}/{k}']).max()
        if d > 1e-6:
            print('oracle w_rq diff', k, d)
for i, ((ps, ph, pad), (os_, oh)) in enumerate(zip(rec_p, rec_o)):
    ds = np.abs(ps - os_)
    print(f'pass {i}: state max diff {ds.max():.3g} (|state| max {np.abs(os_).max():.3g}); hidden max diff {np.abs(ph - oh).max():.3g}')
    bad = np.argwhere(ds.max(-1) > 1e-4 * max(1, np.abs(os_).max()))
    print('   rows off:', len(bad), 'of', ds.shape[0] * ds.shape[1], bad[:10].tolist(), 'padded there:', [bool(pad[b, t]) for b, t in bad[:10]])
