"""Cost of one environment step on the acting side: `AgentManager.get_action` +
`end_episode` + `put_episode` of the device-resident assembly (algorithm/agent.py), per configuration and number
of agents; the native launches of a step are counted with the library's launch profiler.

    python tools/agent_bench.py [--steps 300]
"""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import asac_amd  # noqa: E402,F401
from algorithm.agent import AgentManager  # noqa: E402
from algorithm.sac_base import SAC_Base  # noqa: E402
from algorithm.utils.enums import convert_config_to_enum  # noqa: E402
from tests import parity_utils as pu  # noqa: E402

CASES = {
    'vec': ('nn_vec', ['vector'], [(6,)], 2, dict(n_step=4)),
    'rnn': ('nn_rnn', ['vector'], [(6,)], 2, dict(n_step=4, burn_in_step=8, seq_encoder='RNN')),
    'conv': ('nn_conv', ['vector', 'image'], [(10,), (3, 30, 30)], 4, dict(n_step=3, burn_in_step=5)),
    'conv_attn': ('nn_conv_attn', ['vector', 'image'], [(10,), (3, 30, 30)], 4,
                  dict(n_step=3, burn_in_step=5, seq_encoder='ATTN')),
}


def run(tag, n_agents, steps):
    plugin, names, shapes, A, kw = CASES[tag]
    kw = dict(kw)
    convert_config_to_enum(kw)
    sac = SAC_Base(names, shapes, [], A, None, pu.plugin(plugin), device='cuda:0', batch_size=64,
                   replay_config={'capacity': 16384}, **kw)
    mgr = AgentManager('bench', names, shapes, [np.float32] * len(shapes), [], A, max_episode_length=256)
    mgr.set_rl(sac)
    rng = np.random.default_rng(0)
    ids = np.arange(n_agents)
    obs = [rng.standard_normal((n_agents, *s)).astype(np.float32) for s in shapes]
    reward = rng.standard_normal(n_agents).astype(np.float32)

    def step(t):
        mgr.get_action(ids, obs, reward)
        term = (np.arange(n_agents) + t) % 50 == 49       # every agent finishes an episode every 50 steps
        if term.any():
            mgr.end_episode(ids[term], [o[term] for o in obs], reward[term], np.zeros(int(term.sum()), dtype=bool))
            mgr.put_episode()

    for t in range(60):
        step(t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(60, 60 + steps):
        step(t)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    out = dict(case=tag, n_agents=n_agents, us_per_env_step=round(dt * 1e6, 1),
               agent_steps_per_s=round(n_agents / dt, 1), replay_rows=int(sac.replay_buffer.size))
    sac.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=300)
    args = ap.parse_args()
    for tag in CASES:
        for n in (8, 64):
            print(json.dumps(run(tag, n, args.steps)), flush=True)


if __name__ == '__main__':
    main()
