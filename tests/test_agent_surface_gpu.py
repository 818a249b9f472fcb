"""GPU: the learner's outward surface with the fused representation layers in place — acting
(`choose_action` / `choose_attn_action`, reference sac_base.py:968-1086) on image and attention agents gives the
same actions through the fused launches as through the PyTorch module path, and a checkpoint written by one
agent restores parameters, replay contents and behaviour in another (reference 654-668, replay_buffer.py:436-446)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests import parity_utils as pu  # noqa: E402


def _agent(plugin, obs_shapes, tmp_path=None, seq_encoder=None, **kw):
    import asac_amd  # noqa: F401
    from algorithm.sac_base import SAC_Base
    from algorithm.utils.enums import SEQ_ENCODER
    torch.manual_seed(3)
    names = ['vector', 'image'][:len(obs_shapes)]
    return SAC_Base(names, obs_shapes, [], 4, str(tmp_path) if tmp_path else None, plugin, device='cuda:0',
                    batch_size=32, n_step=3, burn_in_step=5 if seq_encoder else 2,
                    seq_encoder=SEQ_ENCODER[seq_encoder] if seq_encoder else None,
                    replay_config={'capacity': 512}, hip_config={'use_graph': False}, **kw)


def test_acting_through_fused_encoder_layers_matches_module_path(monkeypatch):
    from asac_amd import native
    from algorithm import fused_conv, fused_mlp
    from algorithm.nn_models.layers import seq_layers as attention
    from tests.plugins import nn_conv, nn_conv_attn
    rng = np.random.default_rng(0)
    n_env = 7
    vec, img = rng.standard_normal((n_env, 10)).astype(np.float32), rng.standard_normal((n_env, 3, 30, 30)).astype(np.float32)
    conv = _agent(nn_conv, [(10,), (3, 30, 30)])
    pre_a = np.zeros((n_env, 4), np.float32)
    hidden = np.zeros((n_env, *conv.seq_hidden_state_shape), np.float32)
    attn = _agent(nn_conv_attn, [(10,), (3, 30, 30)], seq_encoder='ATTN')
    T = 6
    ep_idx = np.tile(np.arange(T, dtype=np.int32), (n_env, 1))
    ep_pad = np.zeros((n_env, T), bool)
    ep_pad[:, 0] = True
    ep_obs = [rng.standard_normal((n_env, T, 10)).astype(np.float32),
              rng.standard_normal((n_env, T, 3, 30, 30)).astype(np.float32)]
    ep_pre_a = rng.random((n_env, T, 4)).astype(np.float32)
    ep_state = rng.standard_normal((n_env, T, *attn.seq_hidden_state_shape)).astype(np.float32)

    def act():
        with native.LaunchProfiler() as prof:
            a1 = conv.choose_action([vec, img], pre_a, hidden, disable_sample=True)
            a2 = attn.choose_attn_action(ep_idx, ep_pad, ep_obs, ep_pre_a, ep_state, disable_sample=True)
        return a1, a2, prof.summary()

    f1, f2, seen = act()
    assert seen['asac_conv2_forward']['calls'] == 2 and 'asac_attention_proj_forward' in seen
    monkeypatch.setattr(fused_mlp, 'FUSED_DENSE', False)
    monkeypatch.setattr(fused_conv, 'conv_stack_desc', lambda *a, **k: None)
    monkeypatch.setattr(attention, '_fused_core_ok', lambda *a, **k: False)
    p1, p2, seen = act()
    assert 'asac_conv2_forward' not in seen and 'asac_attention_proj_forward' not in seen
    for got, want in ((f1, p1), (f2, p2)):
        for g, w in zip(got, want):
            assert g.shape == w.shape and np.isfinite(g).all()
            np.testing.assert_allclose(g, w, rtol=2e-4, atol=2e-5)
    assert f1[0].shape == (n_env, 4) and np.abs(f1[0]).max() <= 1.0
    conv.close()
    attn.close()


def test_checkpoint_round_trip_with_fused_layers(tmp_path):
    from tests.plugins import nn_conv
    a = _agent(nn_conv, [(10,), (3, 30, 30)], tmp_path / 'run')
    rng = np.random.default_rng(1)
    for T in (40, 55, 70):
        a.put_episode(**pu.synthetic_episode(rng, [(10,), (3, 30, 30)], [], 4, (0,), T))
    for _ in range(3):
        a.train()
    a.save_model(save_replay_buffer=True)
    vec, img = rng.standard_normal((3, 10)).astype(np.float32), rng.standard_normal((3, 3, 30, 30)).astype(np.float32)
    pre_a, hidden = np.zeros((3, 4), np.float32), np.zeros((3, *a.seq_hidden_state_shape), np.float32)
    want = a.choose_action([vec, img], pre_a, hidden, disable_sample=True)
    b = _agent(nn_conv, [(10,), (3, 30, 30)], tmp_path / 'run', last_ckpt=str(a.get_global_step()))
    assert b.get_global_step() == a.get_global_step()
    assert torch.equal(b._params.flat, a._params.flat) and torch.equal(b._target_params.flat, a._target_params.flat)
    assert b.replay_buffer.size == a.replay_buffer.size
    assert torch.equal(b.replay_buffer._tree, a.replay_buffer._tree)
    got = b.choose_action([vec, img], pre_a, hidden, disable_sample=True)
    for g, w in zip(got, want):
        np.testing.assert_array_equal(g, w)
    b.train()           # the restored learner keeps training (fused layers, optimizer state in place)
    torch.cuda.synchronize()
    b.replay_buffer.check_health()
    a.close()
    b.close()
