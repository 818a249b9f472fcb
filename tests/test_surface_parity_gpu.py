"""GPU: the rows either side of the train step against the reference.

* acting — `choose_action` / `choose_attn_action` (reference sac_base.py:968-1086) on the reference's recorded
  inputs, weights and Gaussian draw (`tests/golden/f9_acting.npz`): action, probability, next hidden state;
* interop — the product restores a checkpoint and replay files WRITTEN BY THE REFERENCE
  (`tests/golden/interop/2.pth`, `2-rb_tree.npy`, `2-rb_storage.npz`; sac_base.py:568-668, replay_buffer.py:96-111,
  220-227, 436-446) and its next train step equals the step the reference took after restoring the same files
  (`tests/golden/f8_interop.npz`); it also writes the files the CPU test `test_interop_cpu.py` feeds to the reference;
* `add_with_td_error` with `ignore_size` (replay_buffer.py:317-337) against the oracle on a wrapping ring."""
import shutil

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle.per_ref import PrioritizedReplayRef  # noqa: E402
from tests import parity_utils as pu  # noqa: E402


def _agent(plugin_name, model_abs_dir=None, **kw):
    import asac_amd  # noqa: F401
    from algorithm.sac_base import SAC_Base
    from algorithm.utils.enums import convert_config_to_enum
    kw = dict(kw)
    convert_config_to_enum(kw)
    return SAC_Base(['vector'], [(6,)], [], 2, model_abs_dir, pu.plugin(plugin_name), device='cuda:0',
                    hip_config={'use_graph': False}, **kw)


@pytest.mark.parametrize('tag,plugin_name,kw', [
    ('vec', 'nn_vec', dict()),
    ('rnn', 'nn_rnn', dict(seq_encoder='RNN', burn_in_step=3)),
    ('attn', 'nn_attn', dict(seq_encoder='ATTN', burn_in_step=4))])
def test_acting_matches_reference(golden_dir, tag, plugin_name, kw):
    from algorithm.fused import RecordedNoise
    g = np.load(golden_dir / 'f9_acting.npz')
    agent = _agent(plugin_name, batch_size=16, n_step=3, replay_config={'capacity': 64}, **kw)
    pu.load_golden_weights(agent, g, prefix=f'{tag}/w0')
    inputs = {k[len(f'{tag}/in/'):]: g[k] for k in g.files if k.startswith(f'{tag}/in/')}
    for mode, extra in (('sample', {}), ('deter', dict(disable_sample=True))):
        eps = [g[f'{tag}/{mode}/eps{j}'] for j in range(int(g[f'{tag}/{mode}/n_eps']))]
        assert len(eps) == (1 if mode == 'sample' else 0)
        agent.noise = RecordedNoise((), eps, ())
        if tag == 'attn':
            got = agent.choose_attn_action(inputs['ep_indexes'].copy(), inputs['ep_padding_masks'].copy(),
                                           [inputs['ep_obses_list'].copy()], inputs['ep_pre_actions'].copy(),
                                           inputs['ep_pre_attn_states'].copy(), **extra)
        else:
            got = agent.choose_action([inputs['obs_list'].copy()], inputs['pre_action'].copy(),
                                      inputs['pre_seq_hidden_state'].copy(), **extra)
        assert agent.noise.exhausted()
        action, prob, hidden = got
        assert action.dtype == np.float32 and prob.dtype == np.float32
        # device tanh / exp against the host's: 1e-5 on actions and states, 2e-4 on the density (exp of a log-density)
        np.testing.assert_allclose(action, g[f'{tag}/{mode}/action'], rtol=1e-5, atol=2e-6, err_msg=f'{tag}/{mode} action')
        np.testing.assert_allclose(prob, g[f'{tag}/{mode}/prob'], rtol=2e-4, atol=1e-6, err_msg=f'{tag}/{mode} prob')
        np.testing.assert_allclose(hidden, g[f'{tag}/{mode}/hidden'], rtol=1e-5, atol=2e-6, err_msg=f'{tag}/{mode} hidden')
    agent.close()


INTEROP_KW = dict(batch_size=16, n_step=3, burn_in_step=2, seq_encoder='RNN', replay_config={'capacity': 128})


def test_product_restores_reference_files_and_continues(golden_dir, tmp_path):
    from algorithm.fused import RecordedNoise
    g = np.load(golden_dir / 'f8_interop.npz')
    run = tmp_path / 'run'
    (run / 'model').mkdir(parents=True)
    for f in (golden_dir / 'interop').iterdir():
        shutil.copy(f, run / 'model' / f.name)
    agent = _agent('nn_rnn', run, **INTEROP_KW)
    rb = agent.replay_buffer
    assert agent.get_global_step() == 2
    saved = np.load(golden_dir / 'interop' / '2-rb_storage.npz')
    assert rb.size == int(saved['p_size']) == 128 and rb._next_id == int(saved['p_id']) == 165
    assert np.array_equal(rb._tree.cpu().numpy().view(np.uint32), np.load(golden_dir / 'interop' / '2-rb_tree.npy').view(np.uint32))
    assert np.array_equal(rb._slot_ids.cpu().numpy(), saved['_id'])
    for k in saved.files:
        if k not in ('p_size', 'p_id', '_id'):
            assert np.array_equal(rb._columns[k].cpu().numpy(), saved[k]), k
    assert int(agent._opt_steps.item()) == 2, "Adam's step count comes from the reference's optimizer state"

    eps = [g[f'eps{j}'] for j in range(int(g['n_eps']))]
    agent.noise = RecordedNoise([g['u']], eps, list(g['perm']))
    rb.uniform_source = agent.noise
    mods = {name: m for name, m in agent.ckpt_dict.items() if isinstance(m, torch.nn.Module)}
    assert agent.train() == 3
    assert agent.noise.exhausted()
    assert np.array_equal(rb._ids.cpu().numpy(), g['sample_ids']), 'PER index selection after the restore'
    np.testing.assert_allclose(rb._w.cpu().numpy()[:, None], g['is_weights'], rtol=2e-6)
    np.testing.assert_allclose(agent._stats['loss_q'].item(), g['loss_q'], rtol=2e-4)
    np.testing.assert_allclose(agent._td_error.cpu().numpy()[:, None], g['td_error'], rtol=1e-3, atol=2e-5)
    np.testing.assert_allclose(rb._tree.cpu().numpy(), g['tree'], rtol=1e-3, atol=1e-6)
    np.testing.assert_allclose(rb._columns['mu_prob'].cpu().numpy(), g['mu_prob'], rtol=2e-3, atol=1e-6)
    np.testing.assert_allclose(rb._columns['pre_seq_hidden_state'].cpu().numpy(), g['hidden'], rtol=1e-3, atol=2e-5)
    np.testing.assert_allclose(agent.log_c_alpha.item(), g['log_c_alpha'], rtol=1e-5)
    # third Adam update on the restored moments: m / sqrt(v) is no longer sign-like, so no +-lr slack is needed
    for name, mod in mods.items():
        for k, v in mod.state_dict().items():
            np.testing.assert_allclose(v.cpu().numpy(), g[f'w1/{name}/{k}'], rtol=1e-3, atol=1e-4, err_msg=f'{name}/{k}')
    agent.close()


def test_product_writes_files_for_the_reference(tmp_path):
    """The other direction: a product run's checkpoint + replay files, kept under `gpurun_out/product_ckpt/`; the
    committed copy (`tests/golden/product_ckpt/`) is what `tests/test_interop_cpu.py` has the reference load."""
    import os
    from pathlib import Path
    run = tmp_path / 'run'
    torch.manual_seed(4)
    agent = _agent('nn_rnn', run, **INTEROP_KW)
    rng = np.random.default_rng(4)
    for T in (40, 30, 50, 45):
        agent.put_episode(**pu.synthetic_episode(rng, [(6,)], [], 2, tuple(agent.seq_hidden_state_shape), T))
    for _ in range(3):
        agent.train()
    agent.save_model(save_replay_buffer=True)
    files = sorted(p.name for p in (run / 'model').iterdir())
    assert files == ['0.pth', '3-rb_storage.npz', '3-rb_tree.npy', '3.pth']      # (step 0 saves, like the reference)
    saved = torch.load(run / 'model' / '3.pth', weights_only=True)
    st = saved['optimizer_q_0']['state']
    assert set(st[0]) == {'step', 'exp_avg', 'exp_avg_sq'} and float(st[0]['step']) == 3.
    assert saved['optimizer_q_0']['param_groups'][0]['params'] == list(range(len(st)))
    out = Path(os.environ.get('GRAFT_REPO_ROOT', Path(__file__).resolve().parent.parent)) / 'gpurun_out' / 'product_ckpt'
    out.mkdir(parents=True, exist_ok=True)
    for f in (run / 'model').glob('3*'):
        shutil.copy(f, out / f.name)
    # what the restored learner must reproduce: a deterministic action for a fixed observation
    obs = np.linspace(-1, 1, 12, dtype=np.float32).reshape(2, 6)
    a, p, h = agent.choose_action([obs], np.zeros((2, 2), np.float32),
                                  np.zeros((2, *agent.seq_hidden_state_shape), np.float32), disable_sample=True)
    np.savez(out / 'expect.npz', obs=obs, action=a, prob=p, hidden=h, tree=agent.replay_buffer._tree.cpu().numpy(),
             log_c_alpha=agent.log_c_alpha.detach().cpu().numpy())
    agent.close()


@pytest.mark.parametrize('ignore_size', [0, 1, 2])
def test_add_with_td_error_matches_oracle(ignore_size):
    import asac_amd  # noqa: F401
    from algorithm.replay_buffer import PrioritizedReplayBuffer
    C, B = 64, 8
    rb = PrioritizedReplayBuffer(B, 1, 2, torch.device('cuda:0'), capacity=C)
    ref = PrioritizedReplayRef(B, 1, 2, capacity=C)
    rng = np.random.default_rng(11 + ignore_size)
    for it in range(12):                  # 12 episodes of 3..29 rows: the ring wraps several times
        T = int(rng.integers(3, 30))
        ep = {'index': np.arange(T, dtype=np.int32), 'obs_vec': rng.standard_normal((T, 3)).astype(np.float32),
              'reward': rng.standard_normal(T).astype(np.float32)}
        td = (np.abs(rng.standard_normal((T, 1))) * 0.7).astype(np.float32)
        td[rng.integers(0, T)] = 5.0      # clipped to td_error_max
        td[rng.integers(0, T)] = 1e-4     # clipped to td_error_min
        rb.add_with_td_error(td, ep, ignore_size=ignore_size)
        ref.add_with_td_error(td, ep, ignore_size=ignore_size)
        assert np.array_equal(rb._slot_ids.cpu().numpy(), ref.storage.columns['_id']), f'add {it}: id map'
        for k in ep:
            assert np.array_equal(rb._columns[k].cpu().numpy(), ref.storage.columns[k]), f'add {it}: {k}'
        tree, want = rb._tree.cpu().numpy(), ref.tree.tree
        assert np.array_equal(tree == 0, want == 0), f'add {it}: zero-priority rows (episode tail / ring tail)'
        np.testing.assert_allclose(tree, want, rtol=2e-6, atol=0, err_msg=f'add {it}')    # device powf vs host: 1 ulp
        assert rb.size == ref.size and rb.get_curr_id() == ref.get_curr_id()
    assert rb.check_tree_invariant() == 0
    with pytest.raises(Exception, match='td_error has nan'):
        rb.add_with_td_error(np.array([0.5, np.nan, 0.2], np.float32), {'index': np.arange(3, dtype=np.int32),
                             'obs_vec': np.zeros((3, 3), np.float32), 'reward': np.zeros(3, np.float32)})
        rb.check_health()
    rb._nan_flag.zero_()
    rb.close()


def test_packed_ingress_equals_per_key_copies_and_the_oracle():
    """NumPy episodes enter through one pinned copy + one `asac_rows_move` scatter (`_store_rows_packed`): same ring
    bytes, id map and tree as the per-key copies and as the oracle — across the ring seam, for an episode longer than
    the ring, odd row sizes (bool / uint8 / 3-byte rows) and torch inputs mixed in (which keep the per-key path)."""
    import asac_amd  # noqa: F401
    from algorithm.replay_buffer import PrioritizedReplayBuffer
    from asac_amd import native
    C, B = 64, 8
    packed = PrioritizedReplayBuffer(B, 1, 2, torch.device('cuda:0'), capacity=C)
    plain = PrioritizedReplayBuffer(B, 1, 2, torch.device('cuda:0'), capacity=C)
    plain.packed_ingress = False
    ref = PrioritizedReplayRef(B, 1, 2, capacity=C)
    rng = np.random.default_rng(3)
    lengths = [5, 29, 17, 64, 3, 150, 41, 7, 63, 2]          # 150 > C: only the last C rows survive
    for it, T in enumerate(lengths):
        ep = {'index': np.arange(T, dtype=np.int32), 'obs_vec': rng.standard_normal((T, 3)).astype(np.float32),
              'obs_img': rng.integers(0, 255, (T, 3, 5, 5)).astype(np.uint8),
              'flag': rng.integers(0, 2, (T, 3)).astype(bool), 'done': rng.integers(0, 2, T).astype(bool),
              'reward': rng.standard_normal(T).astype(np.float32),
              'hidden': rng.standard_normal((T, 2, 4)).astype(np.float32), 'none': np.zeros((T, 0), np.float32)}
        with native.LaunchProfiler(repeat=1) as prof:
            packed.add(ep, ignore_size=1)
        assert prof.summary()['asac_rows_move']['calls'] == 1
        plain.add(ep, ignore_size=1)
        ref.add(ep, ignore_size=1)
        for k in ep:
            got = packed._columns[k].cpu().numpy()
            assert got.dtype == ep[k].dtype and np.array_equal(got, plain._columns[k].cpu().numpy()), (it, k)
            assert np.array_equal(got, ref.storage.columns[k]), (it, k, 'oracle')
        assert np.array_equal(packed._slot_ids.cpu().numpy(), ref.storage.columns['_id'])
        assert np.array_equal(packed._tree.cpu().numpy().view(np.uint32), plain._tree.cpu().numpy().view(np.uint32))
        assert packed.size == ref.size and packed.get_curr_id() == ref.get_curr_id()
    # a torch tensor among the values: the per-key path (no packing), same result
    T = 6
    ep = {'index': np.arange(T, dtype=np.int32), 'obs_vec': torch.randn(T, 3), 'obs_img': np.zeros((T, 3, 5, 5), np.uint8),
          'flag': np.zeros((T, 3), bool), 'done': np.zeros(T, bool), 'reward': np.zeros(T, np.float32),
          'hidden': np.zeros((T, 2, 4), np.float32), 'none': np.zeros((T, 0), np.float32)}
    with native.LaunchProfiler(repeat=1) as prof:
        packed.add(ep, ignore_size=1)
    assert 'asac_rows_move' not in prof.summary()
    plain.add(ep, ignore_size=1)
    assert torch.equal(packed._columns['obs_vec'], plain._columns['obs_vec'])
    assert packed.check_tree_invariant() == 0
    packed.close()
    plain.close()


def test_packed_ingress_converts_dtypes_and_rejects_other_row_shapes():
    """A later episode whose key arrives in another dtype (float64 rewards after a float32 first episode: `put_episode`
    does no casting) is converted to the ring's dtype like the per-key `copy_` did — not scattered as rows of another
    width — and a different row shape / an unknown key raises instead of writing out of bounds."""
    import asac_amd  # noqa: F401
    from algorithm.replay_buffer import PrioritizedReplayBuffer
    C, B = 32, 4
    packed = PrioritizedReplayBuffer(B, 1, 2, torch.device('cuda:0'), capacity=C)
    plain = PrioritizedReplayBuffer(B, 1, 2, torch.device('cuda:0'), capacity=C)
    plain.packed_ingress = False
    rng = np.random.default_rng(11)

    def episode(T, reward_dtype=np.float32, mu_dtype=np.float32, width=3):
        return {'index': np.arange(T, dtype=np.int32), 'obs_vec': rng.standard_normal((T, width)).astype(np.float32),
                'reward': rng.standard_normal(T).astype(reward_dtype), 'mu_prob': rng.random((T, 2)).astype(mu_dtype),
                'done': np.zeros(T, bool)}
    for ep in (episode(7), episode(9, np.float64, np.float64), episode(30, np.float16), episode(5, np.float64)):
        packed.add(ep, ignore_size=1)
        plain.add(ep, ignore_size=1)
        for k in ep:
            assert packed._columns[k].dtype == plain._columns[k].dtype
            assert torch.equal(packed._columns[k], plain._columns[k]), k
    with pytest.raises(ValueError):
        packed.add(episode(4, width=5))
    bad = episode(4)
    bad['extra'] = np.zeros(4, np.float32)
    with pytest.raises(KeyError):
        packed.add(bad)
    ragged = episode(4)
    ragged['reward'] = ragged['reward'][:3]
    with pytest.raises(ValueError):
        packed.add(ragged)
    assert packed.check_tree_invariant() == 0
    packed.close()
    plain.close()


def test_option_critic_replay_fields_and_random_reads():
    """the option-critic variant's storage dict (reference oc/option_selector_base.py:2029-2086:
    `option_index` int8, `option_changed_index` int32, `pre_low_seq_hidden_state` beside the usual keys) through
    `add`, window sampling and `get_storage_data` / `get_storage_data_ids` — the random reads of its key-transition
    walk (2205, 2223: ids - 1, ids - delta, negative and stale ids included) — on a ring that wraps: every key
    bit-exact against `oracle.per_ref.RingStorageRef.rows_at`, in ONE gather launch per call."""
    import asac_amd  # noqa: F401
    from asac_amd import native
    from algorithm.replay_buffer import PrioritizedReplayBuffer
    C, B = 128, 16
    rb = PrioritizedReplayBuffer(B, 2, 3, torch.device('cuda:0'), capacity=C)
    ref = PrioritizedReplayRef(B, 2, 3, capacity=C)
    rng = np.random.default_rng(21)
    for _ in range(9):                       # 9 episodes of 20..59 rows on 128 slots
        T = int(rng.integers(20, 60))
        ep = {'index': np.arange(T, dtype=np.int32), 'last_mask': np.arange(T) == T - 1,
              'obs_vector': rng.standard_normal((T, 5)).astype(np.float32),
              'obs_image': rng.integers(0, 256, (T, 3, 6, 6)).astype(np.uint8),
              'option_index': rng.integers(-1, 4, T).astype(np.int8),
              'option_changed_index': np.maximum.accumulate(np.where(rng.random(T) < 0.3, np.arange(T), 0)).astype(np.int32),
              'action': rng.random((T, 3)).astype(np.float32), 'reward': rng.standard_normal(T).astype(np.float32),
              'done': rng.random(T) < 0.1, 'mu_prob': rng.random((T, 3)).astype(np.float32),
              'pre_seq_hidden_state': rng.standard_normal((T, 4)).astype(np.float32),
              'pre_low_seq_hidden_state': rng.standard_normal((T, 2, 3)).astype(np.float32)}
        rb.add(ep, ignore_size=1)
        ref.add(ep, ignore_size=1)
    assert set(rb._columns) == set(ep)
    assert rb._columns['option_index'].dtype == torch.int8 and rb._columns['obs_image'].dtype == torch.uint8
    u = rng.random(B)
    rb._u.copy_(torch.from_numpy(u))
    rb.uniform_source = type('U', (), {'fill': staticmethod(lambda buf: None)})()
    ids, windows, w = rb.sample()
    ids_ref, win_ref, w_ref = ref.sample(u)
    assert np.array_equal(ids.cpu().numpy(), ids_ref)
    for k, v in win_ref.items():             # no padding configured: plain windows of every key
        assert np.array_equal(windows[k].cpu().numpy(), v), k
    pointers = ids_ref.copy()
    for hop in range(4):                     # the key-transition walk's reads
        for probe in (pointers - 1, pointers - rng.integers(0, 40, B), pointers + 10 * C, -pointers):
            with native.LaunchProfiler(repeat=1) as prof:
                got = rb.get_storage_data(probe)
            assert prof.summary()['asac_gather_rows']['calls'] == 1, 'one launch for all keys'
            want = ref.storage.rows_at(probe)
            assert set(got) == set(want)
            for k, v in want.items():
                g = got[k].cpu().numpy()
                assert g.dtype == v.dtype and np.array_equal(g, v), f'hop {hop}: {k}'
            assert np.array_equal(rb.get_storage_data_ids(probe).cpu().numpy(), ref.storage.ids_at(probe))
        pointers = pointers - 1 - hop
    assert rb.get_storage_data(np.zeros(0, np.int64))['reward'].shape == (0,)
    rb.close()
