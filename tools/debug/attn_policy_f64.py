"""debug: policy-step gradient of the golden `attn` step 0 in float64 from the aligned weights, vs product and golden"""
import copy, sys
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

case = 'attn'
g = np.load(ROOT / f'tests/golden/f6_step_{case}.npz')
agent = make_agent(case)
pu.load_golden_weights(agent, g)
io = pu.STEP_CASES[case][3]
for ep in pu.golden_episodes(g, 1):
    agent.put_episode(**ep)
eps = [g[f'step0/eps{j}'] for j in range(int(g['step0/n_eps']))]
agent.noise = RecordedNoise([g['step0/u']], eps, list(g['step0/perm']))
agent.replay_buffer.uniform_source = agent.noise
agent.after_rep_q_update = lambda: pu.load_golden_weights(agent, g, prefix='step0/w_rq')
box = {}
orig = agent._train_policy
def tp(obs_list, state, action, mu, ls=None):
    box['state'] = state.detach().cpu().double().clone()
    box['pi'] = copy.deepcopy(agent.model_policy).cpu().double()
    box['q'] = [copy.deepcopy(q).cpu().double() for q in agent.model_q_list]
    box['alpha'] = agent.log_c_alpha.detach().cpu().double().exp()
    return orig(obs_list, state, action, mu, ls=ls)
agent._train_policy = tp
agent.train()
m = pu.product_first_moments(agent)
pi, qs, state = box['pi'], box['q'], box['state']
for dt in (torch.float64, torch.float32):
    pi_, qs_, st = copy.deepcopy(pi).to(dt), [copy.deepcopy(q).to(dt) for q in qs], state.to(dt)
    d, c = pi_(st, [None])
    e = torch.from_numpy(g['step0/eps1']).to(dt)
    x = c.loc + e * c.scale
    cq = torch.stack([q(st, torch.tanh(x), [None])[1] for q in qs_])
    logp = sac_ref.masked_sum_log_prob(sac_ref.squash_log_prob(c, x), keepdim=True)
    loss = torch.mean(box['alpha'].to(dt) * logp - cq.min(0)[0])
    grads = torch.autograd.grad(loss, list(pi_.parameters()))
    print(dt, "loss", float(loss.detach()), 'golden', float(g['step0/loss_policy']))
    for j, gr in enumerate(grads):
        want = g[f'g0/optimizer_policy/{j}'] / 0.1
        got = m['optimizer_policy'][j].cpu().numpy() / 0.1
        ref = gr.numpy()
        sc = np.abs(ref).max()
        print(f'   param {j}: golden-vs-host {np.abs(want - ref).max() / sc:.2e}   product-vs-host {np.abs(got - ref).max() / sc:.2e}')
    print('   loc range', float(c.loc.abs().max()), 'scale range', float(c.scale.min()), float(c.scale.max()), '|x| max', float(x.abs().max()))
