"""GPU: the device-resident agent-side episode assembly (`algorithm/agent.py`, `asac_rows_move`) against the
reference's `AgentManager` (reference algorithm/agent.py:21-690), SURVEY.md §8f rank 2.

`tests/golden/f10_agent.npz` holds what the reference did when driven by `tests/golden/agent_script.py` — 26
environment steps of five agents that join, pause, finish "empty" first episodes, overflow an 8-row episode buffer
— for a vector, a recurrent and an attention learner: the continuous action it returned every step (with the
recorded Gaussian draw) and EVERY episode it handed to `put_episode`.  The product replays the same script:
indexes, observations, rewards and done flags of the episodes must be bit-identical, actions / probabilities /
hidden states within the acting tolerances (`test_surface_parity_gpu.py`; probabilities 1e-3), the agents' statistics equal.
Also: the row mover against NumPy on every addressing mode, and slab -> replay ring without a host copy."""
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

sys.path.insert(0, str(Path(__file__).resolve().parent / 'golden'))
from agent_script import AGENT_CASES, agent_script  # noqa: E402
from tests import parity_utils as pu  # noqa: E402


def _learner(plugin_name, **kw):
    import asac_amd  # noqa: F401
    from algorithm.sac_base import SAC_Base
    from algorithm.utils.enums import convert_config_to_enum
    kw = dict(kw)
    convert_config_to_enum(kw)
    return SAC_Base(['vector'], [(6,)], [], 2, None, pu.plugin(plugin_name), device='cuda:0',
                    hip_config={'use_graph': False}, batch_size=16, n_step=3, replay_config={'capacity': 64}, **kw)


def _assert_probs(got, want, what):
    """Densities of the stored actions.  Where the policy's log-std sits on its clamp (scale ~1e-9 beside |loc| ~ 5,
    the attention fixture: see DESIGN.md §5) `(loc + eps * scale) - loc` cancels in f32 and the density is 0 or ~1e9
    depending on the last bit, in the reference as much as here: those entries (density > 1e6) are only required
    to be degenerate on both sides.  Below that the density's condition number still grows like the density itself
    (one ulp of the pre-squash action is 6e-8 |u| / scale standard deviations), and the density is evaluated at
    atanh(clamp(tanh(u))), which loses digits as tanh saturates: rtol 1e-3 (the tolerance of the written-back
    probabilities in the step tests) + 1e-8 * density.
    -> number of degenerate entries"""
    sane = want < 1e6
    err = np.abs(got[sane] - want[sane])
    bound = 1e-6 + (1e-3 + 1e-8 * want[sane]) * np.abs(want[sane])
    assert np.all(err <= bound), (what, got[sane][err > bound], want[sane][err > bound])
    assert np.all((got[~sane] > 1e6) | (got[~sane] == 0)), what
    return int((~sane).sum())


def test_rows_move_modes():
    import asac_amd  # noqa: F401
    from asac_amd import native
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(0)
    S, R, n = 6, 9, 11
    for width, dtype in ((3, np.float32), (4, np.float32), (5, np.uint8), (16, np.uint8), (1, np.int32)):
        slab = rng.integers(0, 200, (S, R, width)).astype(dtype)
        slot = rng.integers(0, S, n).astype(np.int32)
        row = rng.integers(-3, R - 1, n).astype(np.int32)
        d_slab, d_slot, d_row = (torch.from_numpy(x).to(dev) for x in (slab, slot, row))
        rb = width * slab.itemsize
        # gather with padding in front and a row offset
        out = torch.zeros((n, width), dtype=d_slab.dtype, device=dev)
        native.rows_move(native.make_row_moves([dict(
            src=d_slab, dst=out, row_bytes=rb, src_mode=native.ROW_SLOT_ROW, src_stride0=R * rb, src_stride1=rb,
            dst_mode=native.ROW_ITEM, dst_stride0=rb, src_row_offset=1, pad_word=0xffffffff)]), d_slot, d_row, None, n)
        want = np.where((row + 1 >= 0)[:, None], slab[slot, np.maximum(row + 1, 0)],
                        np.frombuffer(b'\xff' * rb, dtype=dtype)[None])
        assert np.array_equal(out.cpu().numpy().view(np.uint8), want.view(np.uint8)), (width, dtype)
        # scatter items -> (slot, row); negative rows skipped; broadcast source into per-slot rows
        rows_in = rng.integers(0, 200, (n, width)).astype(dtype)
        uniq = np.unique(np.stack([slot, np.maximum(row, 0)]), axis=1, return_index=True)[1]   # no duplicate targets
        keep = np.zeros(n, dtype=bool)
        keep[uniq] = True
        row_w = np.where(keep, row, -1).astype(np.int32)
        target = d_slab.clone()
        per_slot = torch.zeros((S, width), dtype=d_slab.dtype, device=dev)
        bc = torch.from_numpy(rng.integers(0, 200, (width,)).astype(dtype)).to(dev)
        d_in = torch.from_numpy(rows_in).to(dev)
        native.rows_move(native.make_row_moves([
            dict(src=d_in, dst=target, row_bytes=rb, src_mode=native.ROW_ITEM, src_stride0=rb,
                 dst_mode=native.ROW_SLOT_ROW, dst_stride0=R * rb, dst_stride1=rb),
            dict(src=bc, dst=per_slot, row_bytes=rb, src_mode=native.ROW_BROADCAST, dst_mode=native.ROW_SLOT,
                 dst_stride0=rb)]), d_slot, None, torch.from_numpy(row_w).to(dev), n)
        want = slab.copy()
        for i in range(n):
            if row_w[i] >= 0:
                want[slot[i], row_w[i]] = rows_in[i]
        assert np.array_equal(target.cpu().numpy(), want), (width, dtype)
        want_ps = np.zeros((S, width), dtype=dtype)
        want_ps[slot] = bc.cpu().numpy()
        assert np.array_equal(per_slot.cpu().numpy(), want_ps)
    with pytest.raises(native.AsacNativeError):
        native.rows_move(native.make_row_moves([dict(src=out, dst=out, row_bytes=4, src_mode=native.ROW_SLOT,
                                                     dst_mode=native.ROW_ITEM)]), None, None, None, 3)


@pytest.mark.parametrize('tag', list(AGENT_CASES))
def test_agent_manager_matches_reference(golden_dir, tag):
    from algorithm.agent import AgentManager, episode_trans_to_numpy
    from algorithm.fused import RecordedNoise
    g = np.load(golden_dir / 'f10_agent.npz')
    _, plugin_name, kw, max_len = AGENT_CASES[tag]
    sac = _learner(plugin_name, **kw)
    pu.load_golden_weights(sac, g, prefix=f'{tag}/w0')
    mgr = AgentManager('test', ['vector'], [(6,)], [np.float32], [], 2, max_episode_length=max_len, hit_reward=1)
    mgr.set_rl(sac)
    episodes = []

    def drain(t):
        for ep in mgr.get_tmp_episode_trans_list():
            assert all(v.is_cuda for v in ep.values() if isinstance(v, torch.Tensor)), 'episodes stay in HBM'
            episodes.append((t, episode_trans_to_numpy(ep)))
        mgr.clear_tmp_episode_trans_list()

    script = agent_script(tag)
    for t, st in enumerate(script):
        sac.noise = RecordedNoise((), [g[f'{tag}/t{t}/eps']], ())
        d_action, c_action = mgr.get_action(st['agent_ids'], [st['obs'].copy()], st['last_reward'].copy())
        assert sac.noise.exhausted()
        assert d_action.shape == (len(st['agent_ids']), 0) and c_action.dtype == np.float32
        np.testing.assert_allclose(c_action, g[f'{tag}/t{t}/c_action'], rtol=1e-5, atol=2e-6, err_msg=f'{tag} step {t}')
        m = st['term']
        mgr.end_episode(st['agent_ids'][m], [st['term_obs'][m]], st['term_reward'][m], st['term_max'][m])
        drain(t)
    last = script[-1]
    mgr.end_episode(last['agent_ids'], [last['term_obs']], last['term_reward'],
                    np.ones(len(last['agent_ids']), dtype=bool), force_terminated=True)
    mgr.force_end_all_episodes()
    drain(len(script))

    assert len(episodes) == int(g[f'{tag}/n_episodes'])
    degenerate = n_probs = 0
    for i, (t, ep) in enumerate(episodes):
        want = {k[len(f'{tag}/ep{i}/'):]: g[k] for k in g.files if k.startswith(f'{tag}/ep{i}/')}
        assert t == int(want['step']), f'episode {i} ended at another step'
        for k in ('ep_indexes', 'ep_rewards', 'ep_dones'):
            assert ep[k].dtype == want[k].dtype and np.array_equal(ep[k], want[k]), f'{tag} episode {i} {k}'
        assert np.array_equal(ep['ep_obses_list'][0], want['ep_obses_list']), f'{tag} episode {i} observations'
        np.testing.assert_allclose(ep['ep_actions'], want['ep_actions'], rtol=1e-5, atol=2e-6)
        degenerate += _assert_probs(ep['ep_probs'], want['ep_probs'], f'{tag} episode {i}')
        n_probs += want['ep_probs'].size
        np.testing.assert_allclose(ep['ep_pre_seq_hidden_states'], want['ep_pre_seq_hidden_states'], rtol=1e-5, atol=2e-6)
    assert degenerate <= 0.1 * n_probs, (degenerate, n_probs)
    ids = sorted(mgr.agents_dict)
    assert ids == list(g[f'{tag}/final/agent_ids'])
    for f in ('steps', 'done', 'max_reached', 'force_terminated', 'hit', 'current_step'):
        assert [float(getattr(mgr.agents_dict[i], f)) for i in ids] == list(g[f'{tag}/final/{f}']), f
    np.testing.assert_allclose([mgr.agents_dict[i].reward for i in ids], g[f'{tag}/final/reward'], rtol=1e-6)
    assert [mgr.agents_liveness[i] for i in ids] == list(g[f'{tag}/final/liveness'])
    sac.close()


def test_episodes_enter_the_replay_ring_without_the_host(golden_dir):
    """`AgentManager.put_episode` hands device tensors to `SAC_Base.put_episode`; the ring must equal what the
    NumPy path (the reference's callers) stores for the same episodes."""
    from algorithm.agent import AgentManager, episode_trans_to_numpy
    tag = 'rnn'
    _, plugin_name, kw, max_len = AGENT_CASES[tag]
    torch.manual_seed(3)
    a, b = _learner(plugin_name, **kw), _learner(plugin_name, **kw)
    mgr = AgentManager('test', ['vector'], [(6,)], [np.float32], [], 2, max_episode_length=max_len)
    mgr.set_rl(a)
    for st in agent_script(tag):
        mgr.get_action(st['agent_ids'], [st['obs']], st['last_reward'])
        m = st['term']
        mgr.end_episode(st['agent_ids'][m], [st['term_obs'][m]], st['term_reward'][m], st['term_max'][m])
        for ep in mgr.get_tmp_episode_trans_list():
            b.put_episode(**episode_trans_to_numpy(ep))
        mgr.put_episode()
        assert not mgr.get_tmp_episode_trans_list()
    ra, rb = a.replay_buffer, b.replay_buffer
    assert ra.size == rb.size > 64 // 2 and ra._next_id == rb._next_id
    assert np.array_equal(ra._tree.cpu().numpy().view(np.uint32), rb._tree.cpu().numpy().view(np.uint32))
    assert np.array_equal(ra._slot_ids.cpu().numpy(), rb._slot_ids.cpu().numpy())
    for k in ra._columns:
        assert torch.equal(ra._columns[k], rb._columns[k]), k
    assert a.train() == 1     # and the learner trains from it
    a.close()
    b.close()


def test_test_actions_without_a_learner():
    """`get_test_action` (agent.py:576-618): random actions, episodes still assembled."""
    from algorithm.agent import AgentManager
    np.random.seed(0)
    mgr = AgentManager('test', ['vector', 'flag'], [(6,), (2,)], [np.float32, np.uint8], [3, 2], 2, max_episode_length=16)
    ids = np.arange(3)
    for t in range(6):
        d_action, c_action = mgr.get_action(ids, [np.random.randn(3, 6).astype(np.float32),
                                                  np.random.randint(0, 255, (3, 2)).astype(np.uint8)],
                                            np.zeros(3, dtype=np.float32))
        assert d_action.shape == (3, 5) and c_action.shape == (3, 2)
        assert np.all(d_action[:, :3].sum(-1) == 1) and np.all(d_action[:, 3:].sum(-1) == 1)
    mgr.end_episode(ids, [np.zeros((3, 6), np.float32), np.zeros((3, 2), np.uint8)], np.ones(3, np.float32),
                    np.zeros(3, dtype=bool))
    eps = mgr.get_tmp_episode_trans_list()
    assert len(eps) == 3
    for ep in eps:
        assert ep['ep_indexes'].cpu().numpy().tolist() == [list(range(7))]
        assert ep['ep_obses_list'][1].dtype == torch.uint8 and ep['ep_obses_list'][1].shape == (1, 7, 2)
        assert ep['ep_dones'].cpu().numpy().tolist() == [[False] * 5 + [True, False]]
        assert np.array_equal(ep['ep_actions'][0, -1].cpu().numpy(), np.asarray([1, 0, 0, 1, 0, 0, 0], np.float32))


def test_multi_agents_manager_runs_two_behaviours(tmp_path):
    """`MultiAgentsManager` (agent.py:691-890) over two behaviour names with their own learners: the reference's
    training-loop calls (`sac_main._run`, 388-532) in order — actions per behaviour, episodes ended, put, trained."""
    from algorithm.agent import MultiAgentsManager
    names = ['a?team=0', 'b?team=1']
    mgr = MultiAgentsManager({n: ['vector'] for n in names}, {n: [(6,)] for n in names}, {n: [np.float32] for n in names},
                             {n: [] for n in names}, {n: 2 for n in names}, inference_ma_names=set(), model_abs_dir=tmp_path,
                             max_episode_length=32)
    assert len(mgr) == 2 and all((tmp_path / d).is_dir() for d in ('a-team=0', 'b-team=1'))
    for n, m in mgr:
        m.set_rl(_learner('nn_vec'))
    mgr.set_train_mode(True)
    rng = np.random.default_rng(0)
    ids = {n: np.arange(3) for n in names}
    trained = 0
    for t in range(40):
        obs = {n: [rng.standard_normal((3, 6)).astype(np.float32)] for n in names}
        rew = {n: rng.standard_normal(3).astype(np.float32) for n in names}
        d_act, c_act = mgr.get_ma_action(ids, obs, rew)
        assert all(c_act[n].shape == (3, 2) and d_act[n].shape == (3, 0) for n in names)
        if t % 9 == 8:
            mgr.end_episode(ids, obs, rew, {n: np.zeros(3, dtype=bool) for n in names})
            assert all(len(m.get_tmp_episode_trans_list()) == 3 for _, m in mgr)
            mgr.log_episode()
            mgr.put_episode()
            trained = mgr.train(trained)
    assert (tmp_path / 'episodes_info.json').exists()
    # 4 rounds x 3 agents x 10 rows (9 transitions + the closing row) through a ring of 64 slots
    assert trained >= 1 and all(m.rl.replay_buffer.size == 64 and m.rl.replay_buffer._next_id == 120 for _, m in mgr)
    assert mgr.done is False or mgr.done is True
    mgr.reset_and_continue()
    mgr.force_end_all_episode()
    mgr.close()
