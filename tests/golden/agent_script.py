"""The environment side of the agent-assembly fixture (`f10_agent.npz`): who decides, who terminates, observations
and rewards per step.  Shared by the generator (`make_golden.py`, build container) and the GPU parity test, which
regenerates the script instead of storing it."""
import numpy as np


def agent_script(tag, T=26):
    """The environment side of the agent-assembly fixture: who decides, who terminates, observations and rewards
    per step.  Pure function of `tag` (the GPU test regenerates it instead of storing it)."""
    rng = np.random.default_rng({'vec': 10, 'rnn': 11, 'attn': 12}[tag])
    steps = []
    for t in range(T):
        ids = [0, 1, 2, 3]
        if t >= 6:
            ids.append(9)               # an agent that appears later
        if 10 <= t < 14:
            ids.remove(3)               # ... and one that pauses
        ids = np.asarray(ids)
        n = len(ids)
        term = rng.random(n) < 0.22
        if t == 1:
            term[0] = True              # an "empty" first episode (<= NON_EMPTY_STEPS steps)
        if t < 3:
            term[1:] = False
        steps.append(dict(
            agent_ids=ids, obs=rng.standard_normal((n, 6)).astype(np.float32),
            last_reward=(rng.standard_normal(n) * (t > 0)).astype(np.float32),
            term=term, term_obs=rng.standard_normal((n, 6)).astype(np.float32),
            term_reward=rng.standard_normal(n).astype(np.float32), term_max=rng.random(n) < 0.3))
    return steps


# tag -> (reference plugin file, this repo's test plugin, learner keywords, max_episode_length)
AGENT_CASES = {'vec': ('envs/test/nn.py', 'nn_vec', dict(), 8),      # length 8: the buffer fills and shifts
               'rnn': ('envs/test/nn_rnn.py', 'nn_rnn', dict(seq_encoder='RNN', burn_in_step=3), 40),
               'attn': ('envs/test/nn_attn.py', 'nn_attn', dict(seq_encoder='ATTN', burn_in_step=4), 40)}
