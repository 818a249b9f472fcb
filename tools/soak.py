"""Long run of one bench configuration with episodes arriving between train steps (the ring wraps, slots go stale
under sampled windows): every 500 steps the replay invariants are checked and the parameters screened for
non-finite values.   usage: soak.py cfg4 6000 [episodes-every-N-steps]"""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402


def main():
    cfg, steps = sys.argv[1], int(sys.argv[2])
    every = int(sys.argv[3]) if len(sys.argv) > 3 else 7
    bench.CFG.clear()
    bench.CFG.update(bench.CONFIGS[cfg])
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    agent = bench.build_agent(dev, None, bench.CFG['capacity'] // 8, seed=0)      # small ring: wraps often
    rng = np.random.default_rng(7)
    bench.fill_buffer(agent, rng, bench.CFG['capacity'] // 16)
    for step in range(1, steps + 1):
        agent.train()
        if step % every == 0:
            agent.put_episode(**bench.synthetic_episode(rng, int(rng.integers(20, bench.CFG['episode_len'] + 1))))
        if step % 500 == 0:
            agent.replay_buffer.check_health()
            flat = agent._params.flat
            assert torch.isfinite(flat).all(), f'non-finite parameters at step {step}'
            g = getattr(agent, '_g_state_base', None)
            if g is not None and g.shape[1] in (1, agent.burn_in_step + agent.n_step + 1):
                # the state-gradient buffer must stay zero outside the slice the Q step writes (a pass over the positions
                # behind the burn-in writes position 0 of a shorter buffer: not checked here)
                b = agent.burn_in_step if g.shape[1] > 1 else 0      # (a one-position differentiable pass: [B, 1, S])
                assert float(g[:, :b].abs().sum()) == 0.0 and float(g[:, b + 1:].abs().sum()) == 0.0, 'state-gradient buffer'
            print(step, 'ok; |theta| =', float(flat.norm()), 'graph' if agent._graph is not None else 'eager', flush=True)
    agent.close()
    print('soak ok', cfg, steps)


if __name__ == '__main__':
    main()
