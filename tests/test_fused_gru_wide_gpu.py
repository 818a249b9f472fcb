"""GPU: the MFMA recurrence for GRU hidden sizes 32 / 64 / 128 (`asac_gru_wide_forward / _backward`, csrc/gru_wide.hip +
algorithm/fused_gru_wide.py) against the plugin layer's own cell loop (`nn_models.layers.GRU` on the CPU = the stack of
`nn.GRU` cells of reference seq_layers.py:14-114): outputs, per-layer states, and the gradients of the input, the initial
state and every cell parameter; leading / trailing padding, ragged batch, input widths that are not multiples of 4."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _mask(B, L, gen):
    """leading padding of 0 .. L/3 steps and trailing padding of 0 .. L/4 steps per row; one row fully valid"""
    lead = torch.randint(0, L // 3 + 1, (B,), generator=gen)
    tail = torch.randint(0, L // 4 + 1, (B,), generator=gen)
    lead[0] = tail[0] = 0
    t = torch.arange(L).unsqueeze(0)
    return (t < lead.unsqueeze(1)) | (t >= (L - tail).unsqueeze(1))


@pytest.mark.parametrize('B,L,I,H,layers,masked,with_h0', [
    (32, 12, 16, 64, 1, False, False),     # envs/square/memory_corridor: m.GRU(_, 64, 1)
    (20, 9, 10, 64, 2, True, True),        # ragged batch, input width 10, two layers, padding, initial state
    (16, 17, 24, 32, 2, True, False),
    (48, 7, 36, 128, 1, True, True),       # envs/uav/uav_hole: m.GRU(_, 128, 1)
    (256, 81, 8, 64, 1, True, True),       # the cfg3 window (burn-in 40 + n-step 40) at hidden 64
])
def test_wide_gru_matches_the_cell_loop(B, L, I, H, layers, masked, with_h0):
    import asac_amd  # noqa: F401
    import algorithm.nn_models as m
    from asac_amd import native
    torch.manual_seed(H + layers)
    ref = m.GRU(I, H, layers)
    dev = copy.deepcopy(ref).cuda()
    gen = torch.Generator().manual_seed(B + L)
    x = torch.randn(B, L, I, generator=gen, requires_grad=True)
    h0 = torch.randn(B, layers, H, generator=gen, requires_grad=True) if with_h0 else None
    mask = _mask(B, L, gen) if masked else None
    g_out = torch.randn(B, L, H, generator=gen)
    g_hn = torch.randn(B, L, layers, H, generator=gen)
    out, hn = ref(x, h0, mask)
    ((out * g_out).sum() + (hn * g_hn).sum()).backward()

    xd = x.detach().cuda().requires_grad_(True)
    h0d = None if h0 is None else h0.detach().cuda().requires_grad_(True)
    with native.LaunchProfiler(repeat=1) as prof:
        out_d, hn_d = dev(xd, h0d, None if mask is None else mask.cuda())
        ((out_d * g_out.cuda()).sum() + (hn_d * g_hn.cuda()).sum()).backward()
    calls = prof.summary()
    assert calls['asac_gru_wide_forward']['calls'] == layers and calls['asac_gru_wide_backward']['calls'] == layers
    tol = dict(rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(out_d.detach().cpu().numpy(), out.detach().numpy(), **tol)
    np.testing.assert_allclose(hn_d.detach().cpu().numpy(), hn.detach().numpy(), **tol)
    if mask is not None:
        assert float(out_d.detach()[mask.cuda()].abs().max()) == 0.0          # padded outputs are exactly zero

    def close(got, want, name):
        scale = float(want.abs().max())
        np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=1e-3, atol=3e-5 * max(scale, 1.0), err_msg=name)
    close(xd.grad, x.grad, 'd x')
    if h0 is not None:
        close(h0d.grad, h0.grad, 'd h0')
    for (name, pr), pd in zip(ref.named_parameters(), dev.parameters()):
        close(pd.grad, pr.grad, name)
    # a no-grad pass gives the same values and saves nothing
    with torch.no_grad():
        out_n, hn_n = dev(xd, h0d, None if mask is None else mask.cuda())
    assert torch.equal(out_n, out_d.detach()) and torch.equal(hn_n, hn_d.detach())


def test_other_hidden_sizes_keep_their_paths():
    import asac_amd  # noqa: F401
    import algorithm.nn_models as m
    from algorithm.fused_gru_wide import fused_gru_wide_supported
    x = torch.randn(4, 5, 8, device='cuda')
    assert not fused_gru_wide_supported(x, list(m.GRU(8, 48, 1).cuda()._grus))       # module path
    assert not fused_gru_wide_supported(x, list(m.GRU(8, 8, 2).cuda()._grus))        # csrc/gru.hip
    out, hn = m.GRU(8, 48, 1).cuda()(x)
    assert out.shape == (4, 5, 48) and hn.shape == (4, 5, 1, 48)


@pytest.mark.parametrize('H,B,L', [(64, 256, 81), (32, 48, 7), (128, 16, 5)])
def test_wide_twin_pass_equals_two_passes(H, B, L):
    """online + target representation over the same windows as ONE recurrence launch (`asac_gru_wide_forward_twin`):
    bit-identical to the two separate launches, gradients of the online pass unchanged, taken only after verification; a
    batch that is not a multiple of 16 rows keeps two launches."""
    import asac_amd  # noqa: F401
    from asac_amd import native
    from algorithm.fused_gru import TwinPass
    from tests.test_fused_gru_gpu import _Rep
    torch.manual_seed(3)
    online, target = _Rep(9, H, 1).cuda(), _Rep(9, H, 1).cuda()
    a, b = torch.randn(B, L, 5, device='cuda'), torch.randn(B, L, 4, device='cuda')
    h0 = torch.randn(B, 1, H, device='cuda')
    mask = torch.arange(L, device='cuda').unsqueeze(0) < torch.randint(0, L - 1, (B, 1), device='cuda')

    def both(twin, aa=a, bb=b, hh=h0, mm=mask):
        online.zero_grad()
        ctx = twin if twin is not None else __import__('contextlib').nullcontext()
        with native.LaunchProfiler(repeat=1) as prof:
            with ctx:
                out, hn = online(aa, bb, hh, mm)
                with torch.no_grad():
                    t_out, t_hn = target(aa, bb, hh, mm)
            (out.sum() + (hn * 0.5).sum()).backward()
        grads = [p.grad.clone() for p in online.parameters()]
        return (out, hn, t_out, t_hn, *grads), prof.summary()

    want, _ = both(None)
    twin = TwinPass(online, target, verify_steps=2)
    for step in range(4):
        got, launches = both(twin)
        for w, g in zip(want, got):
            assert torch.equal(w, g)
        if step < 2:
            assert launches['asac_gru_wide_forward_twin']['calls'] == 1 and launches['asac_gru_wide_forward']['calls'] == 1
        else:
            assert twin.trusted and 'asac_gru_wide_forward' not in launches
            assert launches['asac_gru_wide_forward_twin']['calls'] == 1
    got, launches = both(twin, a[:B - 3], b[:B - 3], h0[:B - 3], mask[:B - 3])
    assert 'asac_gru_wide_forward_twin' not in launches and launches['asac_gru_wide_forward']['calls'] == 2


@pytest.mark.parametrize('rows,K,N', [(20736, 8, 192), (100, 5, 96), (17, 64, 384), (33, 1, 16)])
def test_input_products_of_a_narrow_input(rows, K, N):
    """`asac_rows_affine_forward`: x W_ih^T + b_ih for every step in front of the recurrence, x read through a row stride"""
    import asac_amd  # noqa: F401
    from asac_amd import native
    gen = torch.Generator().manual_seed(rows)
    big = torch.randn(rows, K + 3, generator=gen)
    w, b = torch.randn(N, K, generator=gen), torch.randn(N, generator=gen)
    y = torch.full((rows, N), float('nan'), device='cuda')
    native.rows_affine_forward(big.cuda()[:, 1:K + 1], w.cuda(), b.cuda(), y)
    want = big[:, 1:K + 1].double() @ w.double().t() + b.double()
    np.testing.assert_allclose(y.cpu().numpy(), want.numpy(), rtol=2e-5, atol=2e-5)
