"""CPU: the attention layers of the plugin surface reproduce the reference's outputs (golden
`f7_attention.npz`: reference weights + inputs -> outputs for the three hidden-state modes, every
positional encoding and gate type, multi-head, padded and fully padded rows)."""
import numpy as np
import pytest
import torch

import algorithm.nn_models as m
from algorithm.nn_models.layers.seq_layers import GATE, POSITIONAL_ENCODING

CASES = {
    'plain': dict(embed_dim=8),
    'rope_res_ln': dict(embed_dim=8, num_layers=3, num_heads=2, pe=POSITIONAL_ENCODING.ROPE, gate=GATE.RESIDUAL,
                        use_layer_norm=True),
    'rope2_out': dict(embed_dim=8, num_layers=2, num_heads=[1, 4], pe=POSITIONAL_ENCODING.ROPE2, gate=GATE.OUTPUT),
    'abs_rec': dict(embed_dim=6, num_layers=2, num_heads=2, pe=POSITIONAL_ENCODING.ABSOLUTE, gate=GATE.RECURRENT,
                    qkv_dense_depth=1),
    'abscat_cat': dict(embed_dim=4, num_layers=2, pe=POSITIONAL_ENCODING.ABSOLUTE_CAT, gate=GATE.CAT),
    'single': dict(embed_dim=8, num_layers=1, num_heads=2, pe=POSITIONAL_ENCODING.ROPE),
}
TOL = dict(rtol=1e-5, atol=5e-6)      # (host BLAS differs by a few 1e-6 between CPUs; SURVEY 8c: ATTN 2e-5)


def _load(mod, g, prefix):
    params = dict(mod.named_parameters())
    want = {k[len(prefix):]: g[k] for k in g.files if k.startswith(prefix)}
    assert set(params) == set(want), 'parameter names must match the reference'
    with torch.no_grad():
        for k, v in want.items():
            params[k].copy_(torch.from_numpy(v))


@pytest.mark.parametrize('tag', list(CASES))
def test_episode_attention_matches_reference(golden_dir, tag):
    g = np.load(golden_dir / 'f7_attention.npz')
    attn = m.EpisodeMultiheadAttention(**CASES[tag])
    _load(attn, g, f'{tag}/w/')
    key, index, pad = (torch.from_numpy(g[f'{tag}/{k}']) for k in ('key', 'index', 'pad'))
    K, Q = key.shape[1], 3
    with torch.no_grad():
        y, h, w = attn(key, seq_q_len=Q, key_index=index, key_padding_mask=pad)
        np.testing.assert_allclose(y.numpy(), g[f'{tag}/A/y'], **TOL)
        np.testing.assert_allclose(h.numpy(), g[f'{tag}/A/h'], **TOL)
        for i, wi in enumerate(w):
            np.testing.assert_allclose(wi.numpy(), g[f'{tag}/A/w{i}'], **TOL)
        assert torch.isfinite(y).all() and torch.all(y[2] == 0), 'fully padded rows give zeros, not NaN'
        y, h, _ = attn(key, seq_q_len=K, cut_query=True, key_index=index, key_padding_mask=pad)
        np.testing.assert_allclose(y.numpy(), g[f'{tag}/A_full/y'], **TOL)
        np.testing.assert_allclose(h.numpy(), g[f'{tag}/A_full/h'], **TOL)
        y, h, _ = attn(key, seq_q_len=K, hidden_state=torch.from_numpy(g[f'{tag}/C/hs']), is_prev_hidden_state=True,
                       key_index=index, key_padding_mask=pad)
        np.testing.assert_allclose(y.numpy(), g[f'{tag}/C/y'], **TOL)
        np.testing.assert_allclose(h.numpy(), g[f'{tag}/C/h'], **TOL)
        y, h, _ = attn(key, seq_q_len=1, hidden_state=torch.from_numpy(g[f'{tag}/B/hs']), is_prev_hidden_state=False,
                       key_index=index, key_padding_mask=pad)
        np.testing.assert_allclose(y.numpy(), g[f'{tag}/B/y'], **TOL)
        np.testing.assert_allclose(h.numpy(), g[f'{tag}/B/h'], **TOL)
        y, h, _ = attn(key, seq_q_len=Q, query_only_attend_to_rest_key=True, key_index=index)
        np.testing.assert_allclose(y.numpy(), g[f'{tag}/R/y'], **TOL)
        np.testing.assert_allclose(h.numpy(), g[f'{tag}/R/h'], **TOL)


def test_multihead_attention_matches_reference(golden_dir):
    g = np.load(golden_dir / 'f7_attention.npz')
    mha = m.MultiheadAttention(8, num_heads=2, pe=POSITIONAL_ENCODING.ROPE2, out_dense_depth=1, out_size=5)
    _load(mha, g, 'mha/w/')
    with torch.no_grad():
        y, w = mha(torch.from_numpy(g['mha/q']), torch.from_numpy(g['mha/k']), torch.from_numpy(g['mha/k']),
                   key_padding_mask=torch.from_numpy(g['mha/kpm']))
    np.testing.assert_allclose(y.numpy(), g['mha/y'], **TOL)
    np.testing.assert_allclose(w.numpy(), g['mha/wts'], **TOL)
