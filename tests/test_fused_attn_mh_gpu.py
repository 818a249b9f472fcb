"""GPU: the multi-head attention core on MFMA (`asac_attention_mh_forward/backward`, csrc/attn_mh.hip) inside
`MultiheadAttention` / `EpisodeMultiheadAttention` against the same modules' PyTorch path on the CPU (reference
nn_models/layers/seq_layers.py:239-333): outputs, head-averaged weights and the gradients of inputs and parameters, at the
widths of the reference's environments (embed 64 with 2 - 8 heads) and at ragged ones, with causal / per-batch / padding masks,
fully masked ("dead") rows and rotary position encoding in front of the core."""
import copy
import time

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
    if kind == 'padding':
        return None, kpm
    return m, kpm


@pytest.mark.parametrize('B,Lq,Lk,E,H,kind,pe', [
    (256, 9, 9, 64, 8, 'batch', None),         # EpisodeMultiheadAttention(64, num_heads 8) over a window of 9
    (64, 9, 18, 64, 2, 'batch', None),         # second block: keys = previous states ++ outputs
    (33, 16, 16, 64, 4, 'causal2d', 'rope'),
    (5, 1, 32, 64, 1, 'padding', None),        # one head of 64 channels (too wide for csrc/attn.hip)
    (7, 32, 32, 128, 2, 'none', None),         # head_dim 64, both tile pairs
    (3, 7, 5, 24, 2, 'batch', None),           # ragged: head_dim 12
    (4, 20, 27, 40, 8, 'batch', 'rope2'),      # ragged tiles, head_dim 5
])
def test_multihead_core_matches_module_path(B, Lq, Lk, E, H, kind, pe):
    import asac_amd  # noqa: F401
    from asac_amd import native
    import algorithm.nn_models as m
    from algorithm.nn_models.layers.seq_layers import POSITIONAL_ENCODING
    pe = {None: None, 'rope': POSITIONAL_ENCODING.ROPE, 'rope2': POSITIONAL_ENCODING.ROPE2}[pe]
    torch.manual_seed(0)
    ref = m.MultiheadAttention(E, H, pe=pe, out_dense_depth=1)
    dev = copy.deepcopy(ref).cuda()
    gen = torch.Generator().manual_seed(1)
    q, k, v = (torch.randn(B, L, E, generator=gen) for L in (Lq, Lk, Lk))
    mask, kpm = _mask(kind, B, Lq, Lk, gen)
    g_out, g_w = torch.randn(B, Lq, E, generator=gen), torch.randn(B, Lq, Lk, generator=gen) * 0.2

    def run(layer, device):
        qd, kd, vd = (t.clone().to(device).requires_grad_(True) for t in (q, k, v))
        out, w = layer(qd, kd, vd, key_padding_mask=None if kpm is None else kpm.to(device),
                       attn_mask=None if mask is None else mask.to(device))
        ((out * g_out.to(device)).sum() + (w * g_w.to(device)).sum()).backward()
        return [t.detach().cpu().numpy() for t in (out, w, qd.grad, kd.grad, vd.grad, *(p.grad for p in layer.parameters()))]

    want = run(ref, 'cpu')
    with native.LaunchProfiler() as prof:
        got = run(dev, 'cuda')
    seen = prof.summary()
    assert seen['asac_attention_mh_forward']['calls'] == 1 and seen['asac_attention_mh_backward']['calls'] == 1
    for n_, (a, b) in enumerate(zip(got, want)):
        assert np.isfinite(a).all()
        atol = 3e-5 if n_ < 5 else 2e-7 * B * Lq * max(1.0, float(np.abs(b).max()) ** 0.5) + 3e-5
        np.testing.assert_allclose(a, b, rtol=3e-4, atol=atol, err_msg=f'output {n_}')


def test_only_the_weights_or_only_the_output_are_used():
    """either result of the module may stay unused (its gradient is then None): no NaN, same gradients as the module path"""
    import asac_amd  # noqa: F401
    import algorithm.nn_models as m
    torch.manual_seed(0)
    ref = m.MultiheadAttention(64, 8)
    dev = copy.deepcopy(ref).cuda()
    gen = torch.Generator().manual_seed(1)
    x = torch.randn(10, 9, 64, generator=gen)
    mask, kpm = _mask('batch', 10, 9, 9, gen)
    for use in (0, 1):
        xc, xg = x.clone().requires_grad_(True), x.clone().cuda().requires_grad_(True)
        ref(xc, xc, xc, attn_mask=mask, key_padding_mask=kpm)[use].square().sum().backward()
        dev(xg, xg, xg, attn_mask=mask.cuda(), key_padding_mask=kpm.cuda())[use].square().sum().backward()
        np.testing.assert_allclose(xg.grad.cpu().numpy(), xc.grad.numpy(), rtol=3e-4, atol=3e-5)
    with torch.no_grad():
        o_c, w_c = ref(x, x, x, attn_mask=mask)
        o_g, w_g = dev(x.cuda(), x.cuda(), x.cuda(), attn_mask=mask.cuda())
    np.testing.assert_allclose(o_g.cpu().numpy(), o_c.numpy(), rtol=3e-4, atol=3e-5)
    np.testing.assert_allclose(w_g.cpu().numpy(), w_c.numpy(), rtol=3e-4, atol=3e-5)


def test_episode_attention_of_the_reference_environments_and_its_speed():
    """`EpisodeMultiheadAttention(64, num_layers 2, num_heads 8)` (envs/gym/toy_queue/nn_attn.py:28-45) over windows of 9:
    GPU against CPU, then the device time of forward + backward with and without the MFMA core."""
    import asac_amd  # noqa: F401
    import algorithm.nn_models as m
    from algorithm.nn_models.layers import seq_layers
    torch.manual_seed(2)
    ref = m.EpisodeMultiheadAttention(64, num_layers=2, num_heads=8)
    dev = copy.deepcopy(ref).cuda()
    gen = torch.Generator().manual_seed(3)
    B, L = 48, 9
    key = torch.randn(B, L, 64, generator=gen)
    hidden = torch.randn(B, L, ref.output_hidden_state_dim, generator=gen)
    index = torch.arange(L).repeat(B, 1) + torch.randint(0, 5, (B, 1), generator=gen)
    pad = torch.arange(L).unsqueeze(0) < torch.randint(0, 4, (B, 1), generator=gen)
    g_o = torch.randn(B, L, 64, generator=gen)
    kc = key.clone().requires_grad_(True)
    out_c, hn_c, w_c = ref(kc, seq_q_len=L, hidden_state=hidden[:, :1], is_prev_hidden_state=True, key_index=index,
                           key_padding_mask=pad)
    ((out_c * g_o).sum() + hn_c.sum()).backward()
    kg = key.clone().cuda().requires_grad_(True)
    out_g, hn_g, w_g = dev(kg, seq_q_len=L, hidden_state=hidden[:, :1].cuda(), is_prev_hidden_state=True,
                           key_index=index.cuda(), key_padding_mask=pad.cuda())
    ((out_g * g_o.cuda()).sum() + hn_g.sum()).backward()
    np.testing.assert_allclose(kg.grad.cpu().numpy(), kc.grad.numpy(), rtol=5e-4, atol=5e-5)
    for pr, pd in zip(ref.parameters(), dev.parameters()):
        np.testing.assert_allclose(pd.grad.cpu().numpy(), pr.grad.numpy(), rtol=5e-4, atol=2e-4)
    np.testing.assert_allclose(out_g.detach().cpu().numpy(), out_c.detach().numpy(), rtol=3e-4, atol=3e-5)
    np.testing.assert_allclose(hn_g.detach().cpu().numpy(), hn_c.detach().numpy(), rtol=3e-4, atol=3e-5)
    for a, b in zip(w_g, w_c):
        np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().numpy(), rtol=3e-4, atol=3e-5)

    # speed at the learner's batch (1024 windows of 9)
    B = 1024
    key = torch.randn(B, L, 64, device='cuda')
    index = torch.arange(L, device='cuda').repeat(B, 1)
    pad = torch.zeros(B, L, dtype=torch.bool, device='cuda')
    h0 = torch.zeros(B, 1, ref.output_hidden_state_dim, device='cuda')

    def step():
        kk = key.clone().requires_grad_(True)
        o, hn, _ = dev(kk, seq_q_len=L, hidden_state=h0, is_prev_hidden_state=True, key_index=index, key_padding_mask=pad)
        (o.sum() + hn.sum()).backward()

    def timed():
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / 30 * 1e3

    fused = timed()
    seq_layers.FUSED_MULTIHEAD = False
    try:
        generic = timed()
    finally:
        seq_layers.FUSED_MULTIHEAD = True
    print(f'\nEpisodeMultiheadAttention(64, 2 layers, 8 heads), 1024 x 9, forward + backward (eager, wall): MFMA core '
          f'{fused:.3f} ms, module path {generic:.3f} ms')


@pytest.mark.parametrize('B,L,E,tail', [(1024, 9, 64, 9), (37, 9, 64, 3), (5, 18, 32, 1), (3, 7, 128, 7), (1, 1, 64, 1)])
def test_rows_proj_kernels_against_f64(B, L, E, tail):
    """`asac_rows_proj_forward/backward` (csrc/rows_proj.hip): q / k / v = three Linears of one input (the query on the newest
    `tail` positions), read through a strided view, and the summed input gradient — against float64"""
    import asac_amd  # noqa: F401
    from asac_amd import native
    gen = torch.Generator().manual_seed(B + L)
    big = torch.randn(B, L + 2, E + 8, generator=gen)
    x = big[:, 1:L + 1, 4:E + 4]                      # strides (L + 2)(E + 8), E + 8, 1: multiples of 4, 16-byte aligned start
    ws = [torch.randn(E, E, generator=gen) / E ** 0.5 for _ in range(3)]
    bs = [torch.randn(E, generator=gen) for _ in range(3)]
    tails = [tail, L, L]
    xd = big.cuda()[:, 1:L + 1, 4:E + 4]
    outs = [torch.full((B, n, E), float('nan'), device='cuda') for n in tails]
    native.rows_proj_forward(xd, [w.cuda() for w in ws], [b.cuda() for b in bs], tails, outs)
    for o, w, b, n in zip(outs, ws, bs, tails):
        want = x[:, L - n:].double() @ w.double().t() + b.double()
        np.testing.assert_allclose(o.cpu().numpy(), want.numpy(), rtol=2e-5, atol=2e-5)
    gs = [torch.randn(B, n, E, generator=gen) for n in tails]
    gx = torch.full((B, L, E), float('nan'), device='cuda')
    native.rows_proj_backward([g.cuda() for g in gs], tails, [w.cuda() for w in ws], gx)
    want = torch.zeros(B, L, E, dtype=torch.float64)
    for g, w, n in zip(gs, ws, tails):
        want[:, L - n:] += g.double() @ w.double()
    np.testing.assert_allclose(gx.cpu().numpy(), want.numpy(), rtol=2e-5, atol=3e-5)


@pytest.mark.parametrize('rows,E,scaled', [(9216, 64, True), (50, 64, False), (17, 32, True), (33, 128, True)])
def test_rows_resblock_kernels_against_f64(rows, E, scaled):
    """`asac_rows_resblock_forward/backward`: y = (x + gelu(x W^T + b)) * row_scale and its backward against float64 autograd"""
    import asac_amd  # noqa: F401
    from asac_amd import native
    gen = torch.Generator().manual_seed(rows)
    x = torch.randn(rows, E, generator=gen)
    w, b = torch.randn(E, E, generator=gen) / E ** 0.5, torch.randn(E, generator=gen)
    sc = (torch.rand(rows, generator=gen) < 0.8).float() if scaled else None
    gy = torch.randn(rows, E, generator=gen)
    xr = x.double().requires_grad_(True)
    pre_r = xr @ w.double().t() + b.double()
    pre_r.retain_grad()
    yr = (xr + torch.nn.functional.gelu(pre_r)) * (1.0 if sc is None else sc.double().unsqueeze(-1))
    yr.backward(gy.double())
    y, pre = torch.empty(rows, E, device='cuda'), torch.empty(rows, E, device='cuda')
    scd = None if sc is None else sc.cuda()
    native.rows_resblock_forward(x.cuda(), w.cuda(), b.cuda(), scd, y, pre)
    np.testing.assert_allclose(pre.cpu().numpy(), pre_r.detach().numpy(), rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(y.cpu().numpy(), yr.detach().numpy(), rtol=2e-5, atol=2e-5)
    gx, gpre = torch.empty(rows, E, device='cuda'), torch.empty(rows, E, device='cuda')
    native.rows_resblock_backward(gy.cuda(), pre, w.cuda(), scd, gx, gpre)
    np.testing.assert_allclose(gpre.cpu().numpy(), pre_r.grad.numpy(), rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(gx.cpu().numpy(), xr.grad.numpy(), rtol=2e-5, atol=3e-5)


@pytest.mark.parametrize('tail', [9, 4])
def test_projections_and_output_block_around_the_core_are_one_launch_each(tail):
    """self-attention of a window batch (value is key, query = its newest positions): the three projections as one launch,
    the output ResBlock with the padded-row factor as one launch, per pass — values and every gradient as the CPU module"""
    import asac_amd  # noqa: F401
    from asac_amd import native
    import algorithm.nn_models as m
    torch.manual_seed(0)
    ref = m.MultiheadAttention(64, 8, out_dense_depth=1)
    dev = copy.deepcopy(ref).cuda()
    gen = torch.Generator().manual_seed(1)
    B, L = 300, 9
    x = torch.randn(B, L, 64, generator=gen)
    mask, kpm = _mask('batch', B, tail, L, gen)
    rowm = torch.rand(B, tail, generator=gen) < 0.2
    g_out, g_w = torch.randn(B, tail, 64, generator=gen), torch.randn(B, tail, L, generator=gen) * 0.2

    def run(layer, device):
        xd = x.clone().to(device).requires_grad_(True)
        key = xd * 1.0
        out, w = layer(key[:, -tail:] if tail != L else key, key, key, key_padding_mask=kpm.to(device), attn_mask=mask.to(device),
                       out_row_mask=rowm.to(device))
        ((out * g_out.to(device)).sum() + (w * g_w.to(device)).sum()).backward()
        return [t.detach().cpu().numpy() for t in (out, w, xd.grad, *(p.grad for p in layer.parameters()))]

    want = run(ref, 'cpu')
    with native.LaunchProfiler() as prof:
        got = run(dev, 'cuda')
    seen = prof.summary()
    # (windows of <= 16 positions: the projections run inside the core's forward launch, `asac_attention_mh_proj_forward`)
    # (the block is one launch forward and one backward)
    for name in ('asac_attention_mh_proj_forward', 'asac_attention_mh_block_backward'):
        assert seen[name]['calls'] == 1, (name, seen.keys())
    for name in ('asac_rows_resblock_forward', 'asac_rows_proj_forward', 'asac_rows_proj_backward', 'asac_rows_resblock_backward',
                 'asac_attention_mh_backward', 'asac_attention_mh_forward'):
        assert name not in seen, name
    for n_, (a, b) in enumerate(zip(got, want)):
        assert np.isfinite(a).all()
        atol = 3e-5 if n_ < 3 else 2e-7 * B * L * max(1.0, float(np.abs(b).max()) ** 0.5) + 3e-5
        np.testing.assert_allclose(a, b, rtol=3e-4, atol=atol, err_msg=f'output {n_}')


def test_xty_multi_is_the_single_products():
    """`asac_xty_multi`: up to four products over rows in one launch pair — bit-identical to the single launches (same
    partition of the rows, same summation order), overwriting and accumulating, with ragged shapes side by side"""
    import asac_amd  # noqa: F401
    from asac_amd import native
    gen = torch.Generator(device='cuda').manual_seed(5)
    shapes = [(9216, 64, 64), (3072, 64, 64), (9216, 192, 40), (777, 8, 64)]
    jobs, single = [], []
    for R, M, N in shapes:
        x, y = torch.randn(R, M, device='cuda', generator=gen), torch.randn(R, N + 3, device='cuda', generator=gen)[:, :N]
        jobs.append((x, y, torch.randn(M, N, device='cuda', generator=gen), torch.randn(M, device='cuda', generator=gen)))
        single.append((jobs[-1][2].clone(), jobs[-1][3].clone()))
    for acc in (False, True):
        for (x, y, _, _), (o, c) in zip(jobs, single):
            native.xty(x, y, o, c, accumulate=acc)
        native.xty_multi(jobs[:3] + [(jobs[3][0], jobs[3][1], jobs[3][2], None)], accumulate=acc)
        native.xty(jobs[3][0], jobs[3][1], torch.empty_like(jobs[3][2]), jobs[3][3], accumulate=acc)      # (its column sums)
        for (x, y, o, c), (so, sc) in zip(jobs, single):
            assert torch.equal(o, so) and torch.equal(c, sc)
    want = jobs[0][0].double().t() @ jobs[0][1].double()
    native.xty_multi(jobs[:2])
    np.testing.assert_allclose(jobs[0][2].cpu().numpy(), want.cpu().numpy(), rtol=1e-4, atol=1e-3)
    with pytest.raises(RuntimeError):
        native.xty_multi([jobs[0], jobs[0]])         # one output twice


def test_resblock_over_rows_is_one_launch_per_pass():
    """`ResBlock.forward` over thousands of rows on the device: Linear + GELU from a narrow input (no residual path: the
    embedding in front of a sequence encoder) through `asac_rows_affine_gelu_forward`, Linear + GELU + x at 64 channels through
    `asac_rows_resblock_*` — values and every gradient as the CPU modules"""
    import asac_amd  # noqa: F401
    from asac_amd import native
    import algorithm.nn_models as m
    torch.manual_seed(0)
    ref = m.LinearLayers(8, dense_n=64, dense_depth=2)            # ResBlock(8, 64) without, ResBlock(64, 64) with a residual path
    dev = copy.deepcopy(ref).cuda()
    gen = torch.Generator().manual_seed(1)
    x, g = torch.randn(512, 9, 8, generator=gen), torch.randn(512, 9, 64, generator=gen)

    def run(layer, device):
        xd = x.clone().to(device).requires_grad_(True)
        out = layer(xd)
        (out * g.to(device)).sum().backward()
        return [t.detach().cpu().numpy() for t in (out, xd.grad, *(p.grad for p in layer.parameters()))]

    want = run(ref, 'cpu')
    with native.LaunchProfiler() as prof:
        got = run(dev, 'cuda')
    seen = prof.summary()
    for name in ('asac_rows_affine_gelu_forward', 'asac_rows_resblock_forward', 'asac_rows_resblock_backward'):
        assert seen[name]['calls'] == 1, (name, sorted(seen))
    for n_, (a, b) in enumerate(zip(got, want)):
        atol = 3e-5 if n_ < 2 else 2e-7 * x.shape[0] * x.shape[1] * max(1.0, float(np.abs(b).max()) ** 0.5) + 3e-5
        np.testing.assert_allclose(a, b, rtol=3e-4, atol=atol, err_msg=f'output {n_}')


@pytest.mark.parametrize('B,L,tail,E,H', [(300, 9, 9, 64, 8), (300, 10, 3, 64, 8), (64, 16, 16, 128, 4), (33, 5, 1, 32, 2)])
def test_projections_inside_the_core_forward(B, L, tail, E, H):
    """windows of <= 16 positions: q / k / v are projected INSIDE the attention core's forward launch
    (`asac_attention_mh_proj_forward`) — outputs, weights and every gradient as the CPU module; and bit for bit what the
    two-launch form (`ASAC_QKV_IN_CORE=0`) gives, since the arithmetic is the same"""
    import asac_amd  # noqa: F401
    from asac_amd import native
    import algorithm.nn_models as m
    from algorithm.nn_models.layers import seq_layers
    torch.manual_seed(0)
    ref = m.MultiheadAttention(E, H, out_dense_depth=1)
    dev = copy.deepcopy(ref).cuda()
    gen = torch.Generator().manual_seed(1)
    x = torch.randn(B, L, E, generator=gen)
    mask, kpm = _mask('batch', B, tail, L, gen)
    rowm = torch.rand(B, tail, generator=gen) < 0.2
    g_out, g_w = torch.randn(B, tail, E, generator=gen), torch.randn(B, tail, L, generator=gen) * 0.2

    def run(layer, device):
        for p_ in layer.parameters():
            p_.grad = None
        xd = x.clone().to(device).requires_grad_(True)
        key = xd * 1.0
        out, w = layer(key[:, -tail:] if tail != L else key, key, key, key_padding_mask=kpm.to(device), attn_mask=mask.to(device),
                       out_row_mask=rowm.to(device))
        ((out * g_out.to(device)).sum() + (w * g_w.to(device)).sum()).backward()
        return [t.detach().cpu().numpy() for t in (out, w, xd.grad, *(p.grad for p in layer.parameters()))]

    want = run(ref, 'cpu')
    with native.LaunchProfiler() as prof:
        got = run(dev, 'cuda')
    seen = prof.summary()
    assert seen['asac_attention_mh_proj_forward']['calls'] == 1 and 'asac_rows_proj_forward' not in seen and 'asac_attention_mh_forward' not in seen
    assert 'asac_rows_resblock_forward' not in seen and seen['asac_attention_mh_block_backward']['calls'] == 1
    assert 'asac_rows_proj_backward' not in seen and 'asac_attention_mh_backward' not in seen and 'asac_rows_resblock_backward' not in seen
    for n_, (a, b) in enumerate(zip(got, want)):
        assert np.isfinite(a).all()
        atol = 3e-5 if n_ < 3 else 2e-7 * B * L * max(1.0, float(np.abs(b).max()) ** 0.5) + 3e-5
        np.testing.assert_allclose(a, b, rtol=3e-4, atol=atol, err_msg=f'output {n_}')
    # ... and bit for bit the launches they replace: one-launch forward with the three-launch backward, and three + three
    seq_layers.FUSED_BLOCK_BACKWARD = False
    try:
        mixed = run(dev, 'cuda')
        seq_layers.FUSED_QKV_IN_CORE = False
        two = run(dev, 'cuda')
    finally:
        seq_layers.FUSED_QKV_IN_CORE = seq_layers.FUSED_BLOCK_BACKWARD = True
    for a, b, c in zip(got, mixed, two):
        assert np.array_equal(a, b) and np.array_equal(a, c)
