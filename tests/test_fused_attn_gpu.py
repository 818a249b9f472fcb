"""GPU: the short-window attention core (`asac_attention_forward/backward`) inside `MultiheadAttention` /
`EpisodeMultiheadAttention` against the same modules' PyTorch path on the CPU: outputs, returned weights and
the gradients of inputs and parameters, with causal / per-batch / padding masks and fully masked ("dead") rows."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _mask(kind, B, Lq, Lk, gen):
    if kind == 'none':
        return None, None
    if kind == 'causal2d':
        return torch.triu(torch.ones(Lq, Lk, dtype=torch.bool), diagonal=1 + Lk - Lq), None
    m = torch.rand(B, Lq, Lk, generator=gen) < 0.4
    m[0, 0] = True                       # a dead row
    m[1] = True                          # a dead batch entry
    kpm = torch.rand(B, Lk, generator=gen) < 0.3
    return m, kpm


@pytest.mark.parametrize('B,Lq,Lk,E,kind', [
    (1024, 9, 9, 8, 'batch'),       # the cfg5 first block
    (37, 9, 18, 8, 'batch'),        # second block: keys = previous states ++ outputs
    (5, 1, 32, 16, 'causal2d'),
    (64, 32, 32, 4, 'none'),
    (3, 7, 5, 12, 'batch'),
])
def test_attention_core_matches_module_path(B, Lq, Lk, E, kind, monkeypatch):
    import asac_amd  # noqa: F401
    from asac_amd import native
    import algorithm.nn_models as m
    from algorithm.nn_models.layers import seq_layers as attention
    monkeypatch.setattr(attention, 'FUSED_PROJECTIONS', False)      # the core alone (projections by the modules)
    torch.manual_seed(0)
    ref = m.MultiheadAttention(E, 1, out_dense_depth=1)
    dev = copy.deepcopy(ref).cuda()
    gen = torch.Generator().manual_seed(1)
    q, k = torch.randn(B, Lq, E, generator=gen), torch.randn(B, Lk, E, generator=gen)
    mask, kpm = _mask(kind, B, Lq, Lk, gen)
    g_out, g_w = torch.randn(B, Lq, E, generator=gen), torch.randn(B, Lq, Lk, generator=gen) * 0.2

    def run(layer, device):
        qd = q.clone().to(device).requires_grad_(True)
        kd = k.clone().to(device).requires_grad_(True)
        out, w = layer(qd, kd, kd, key_padding_mask=None if kpm is None else kpm.to(device),
                       attn_mask=None if mask is None else mask.to(device))
        ((out * g_out.to(device)).sum() + (w * g_w.to(device)).sum()).backward()
        return [t.detach().cpu().numpy() for t in (out, w, qd.grad, kd.grad, *(p.grad for p in layer.parameters()))]

    want = run(ref, 'cpu')
    with native.LaunchProfiler() as prof:
        got = run(dev, 'cuda')
    assert prof.summary()['asac_attention_forward']['calls'] == 1 and prof.summary()['asac_attention_backward']['calls'] == 1
    for n_, (a, b) in enumerate(zip(got, want)):
        assert np.isfinite(a).all()
        # parameter gradients are sums over B * L rows of O(1) terms (the key projection's bias gradient is a sum
        # that cancels exactly in exact arithmetic): absolute tolerance scaled by the number of summands
        atol = 2e-5 if n_ < 4 else 2e-7 * B * Lq * max(1.0, float(np.abs(b).max()) ** 0.5) + 2e-5
        np.testing.assert_allclose(a, b, rtol=2e-4, atol=atol)


def test_episode_attention_stack_matches_cpu_path_and_heads_fall_back():
    import asac_amd  # noqa: F401
    from asac_amd import native
    import algorithm.nn_models as m
    torch.manual_seed(2)
    ref = m.EpisodeMultiheadAttention(8)
    dev = copy.deepcopy(ref).cuda()
    gen = torch.Generator().manual_seed(3)
    B, L = 50, 9
    key = torch.randn(B, L, 8, generator=gen)
    hidden = torch.randn(B, L, ref.output_hidden_state_dim, generator=gen)
    index = torch.arange(L).repeat(B, 1) + torch.randint(0, 5, (B, 1), generator=gen)
    pad = torch.arange(L).unsqueeze(0) < torch.randint(0, 4, (B, 1), generator=gen)
    g_o = torch.randn(B, L, 8, generator=gen)
    kc = key.clone().requires_grad_(True)
    out_c, hn_c, w_c = ref(kc, seq_q_len=L, hidden_state=hidden[:, :1], is_prev_hidden_state=True, key_index=index,
                           key_padding_mask=pad)
    ((out_c * g_o).sum() + hn_c.sum()).backward()
    kg = key.clone().cuda().requires_grad_(True)
    with native.LaunchProfiler() as prof:
        out_g, hn_g, w_g = dev(kg, seq_q_len=L, hidden_state=hidden[:, :1].cuda(), is_prev_hidden_state=True,
                               key_index=index.cuda(), key_padding_mask=pad.cuda())
        ((out_g * g_o.cuda()).sum() + hn_g.sum()).backward()
    assert prof.summary()['asac_attention_proj_forward']['calls'] == 2
    assert prof.summary()['asac_attention_proj_backward']['calls'] == 2
    # padded positions (zeroed inside the launch) and their gradients
    assert not out_g[pad.cuda()].any()
    np.testing.assert_allclose(kg.grad.cpu().numpy(), kc.grad.numpy(), rtol=3e-4, atol=3e-5)
    for pr, pd in zip(ref.parameters(), dev.parameters()):
        np.testing.assert_allclose(pd.grad.cpu().numpy(), pr.grad.numpy(), rtol=3e-4, atol=1e-4)
    np.testing.assert_allclose(out_g.detach().cpu().numpy(), out_c.detach().numpy(), rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(hn_g.detach().cpu().numpy(), hn_c.detach().numpy(), rtol=2e-4, atol=2e-5)
    for a, b in zip(w_g, w_c):
        np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().numpy(), rtol=2e-4, atol=2e-5)
    two_heads = m.MultiheadAttention(8, 2).cuda()
    with native.LaunchProfiler() as prof:
        two_heads(key.cuda(), key.cuda(), key.cuda())
    assert 'asac_attention_forward' not in prof.summary()


@pytest.mark.parametrize('out_depth', [1, 0, 2])      # 1: the output ResBlock rides in the launch too
@pytest.mark.parametrize('B,Lq,Lk,E,flat', [(1024, 9, 9, 8, True), (37, 9, 18, 8, False), (6, 4, 32, 16, True), (3, 7, 7, 5, False)])
def test_attention_with_projections_on_chip(B, Lq, Lk, E, flat, out_depth):
    """self-attention (value is key, query = the last Lq key rows read in place): q / k / v projections + core as one
    launch per pass; parameter gradients added into flat `.grad` views or returned."""
    import asac_amd  # noqa: F401
    from asac_amd import native
    from algorithm.fused import FlatParamGroup
    import algorithm.nn_models as m
    torch.manual_seed(0)
    ref = m.MultiheadAttention(E, 1, out_dense_depth=out_depth)
    with torch.no_grad():
        for p_ in ref.parameters():
            if p_.dim() == 1:
                p_.normal_(0, 0.3)
    dev = copy.deepcopy(ref).cuda()
    group = FlatParamGroup([('attn', list(dev.parameters()))], 'cuda') if flat else None
    gen = torch.Generator().manual_seed(1)
    key = torch.randn(B, Lk, E, generator=gen)
    mask, kpm = _mask('batch', B, Lq, Lk, gen)
    g_out, g_w = torch.randn(B, Lq, E, generator=gen), torch.randn(B, Lq, Lk, generator=gen) * 0.2

    def run(layer, device):
        kd = key.clone().to(device).requires_grad_(True)
        out, w = layer(kd[:, -Lq:], kd, kd, key_padding_mask=kpm.to(device), attn_mask=mask.to(device))
        ((out * g_out.to(device)).sum() + (w * g_w.to(device)).sum()).backward()
        return [t.detach().cpu().numpy() for t in (out, w, kd.grad, *(p.grad for p in layer.parameters()))]

    want = run(ref, 'cpu')
    if group is not None:
        group.grad.zero_()
    with native.LaunchProfiler() as prof:
        got = run(dev, 'cuda')
    seen = prof.summary()
    assert seen['asac_attention_proj_forward']['calls'] == 1 and seen['asac_attention_proj_backward']['calls'] == 1
    assert 'asac_attention_forward' not in seen
    for n_, (a, b) in enumerate(zip(got, want)):
        assert np.isfinite(a).all()
        atol = 2e-5 if n_ < 3 else 2e-7 * B * Lq * max(1.0, float(np.abs(b).max()) ** 0.5) + 2e-5
        np.testing.assert_allclose(a, b, rtol=3e-4, atol=atol)
