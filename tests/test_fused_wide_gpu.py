"""GPU: the wide-input Linear + GELU launches (csrc/wide.hip through the C ABI: `asac_rows_wide_*`) against plain PyTorch f32
(float64 where the summation order matters) — forward, pre-activations, input / weight / bias gradients, ragged row counts,
strided rows, every supported width — and through `ResBlock` / `ConvLayers` against the module path.  Tolerances: f32
rounding of a K-term dot product (K up to 4 096) and of sums over up to 2 304 rows."""
import numpy as np
import pytest
import torch
from torch import nn

pytestmark = pytest.mark.gpu


def _ref64(x, w, b, gy, act=True):
    x, w, b = (t.detach().double().clone().requires_grad_(True) for t in (x, w, b))
    pre = torch.nn.functional.linear(x, w, b)
    y = torch.nn.functional.gelu(pre) if act else pre
    y.backward(gy.double())
    return y.detach(), pre.detach(), x.grad, w.grad, b.grad


@pytest.mark.parametrize('R,K,N', [(1024, 2592, 64), (2304, 2592, 64), (37, 2592, 64), (1, 16, 32), (300, 4096, 128),
                                   (513, 800, 32), (2048, 2592, 128), (4608, 1152, 64)])
def test_wide_linear_gelu_kernels(R, K, N):
    from asac_amd import native
    assert native.rows_wide_supported(R, K, N)
    torch.manual_seed(R + K + N)
    dev = 'cuda:0'
    x = torch.randn(R, K, device=dev)
    w = torch.randn(N, K, device=dev) * (1.0 / K ** 0.5)
    b = torch.randn(N, device=dev) * 0.1
    gy = torch.randn(R, N, device=dev)
    y_ref, pre_ref, gx_ref, gw_ref, gb_ref = _ref64(x, w, b, gy)
    y, pre = torch.empty(R, N, device=dev), torch.empty(R, N, device=dev)
    with native.LaunchProfiler() as prof:
        native.rows_wide_forward(x, w, b, y, pre)
    assert list(prof.summary()) == ['asac_rows_wide_forward']
    np.testing.assert_allclose(pre.cpu().numpy(), pre_ref.cpu().numpy(), rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(y.cpu().numpy(), y_ref.cpu().numpy(), rtol=2e-5, atol=2e-5)
    y2 = torch.empty_like(y)
    native.rows_wide_forward(x, w, b, y2)                      # inference: no pre-activations, same bits
    assert torch.equal(y2, y)
    # backward
    dpre, dx = torch.empty(R, N, device=dev), torch.empty(R, K, device=dev)
    native.rows_wide_backward_input(gy, pre, w, dpre, dx)
    dpre_ref = (gy.double() * torch.ops.aten.gelu_backward(torch.ones_like(pre_ref), pre.double(), approximate='none'))
    np.testing.assert_allclose(dpre.cpu().numpy(), dpre_ref.cpu().numpy(), rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(dx.cpu().numpy(), gx_ref.cpu().numpy(), rtol=2e-4, atol=2e-5)
    dpre_only = torch.empty_like(dpre)
    native.rows_wide_backward_input(gy, pre, w, dpre_only, None)
    assert torch.equal(dpre_only, dpre)
    dw, db = torch.full((N, K), 7.0, device=dev), torch.full((N,), 7.0, device=dev)
    native.rows_wide_backward_params(dpre, x, dw, db)
    scale = max(R, 16) ** 0.5
    np.testing.assert_allclose(dw.cpu().numpy(), gw_ref.cpu().numpy(), rtol=2e-4, atol=4e-6 * scale)
    np.testing.assert_allclose(db.cpu().numpy(), gb_ref.cpu().numpy(), rtol=2e-4, atol=4e-6 * scale)
    # accumulate form; determinism
    base_w, base_b = torch.randn_like(dw), torch.randn_like(db)
    aw, ab = base_w.clone(), base_b.clone()
    native.rows_wide_backward_params(dpre, x, aw, ab, accumulate=True)
    torch.testing.assert_close(aw - base_w, dw, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(ab - base_b, db, rtol=1e-4, atol=1e-4)
    dw2, db2 = torch.empty_like(dw), torch.empty_like(db)
    native.rows_wide_backward_params(dpre, x, dw2, db2)
    assert torch.equal(dw2, dw) and torch.equal(db2, db)


def test_wide_linear_strided_rows_and_limits():
    from asac_amd import native
    dev = 'cuda:0'
    torch.manual_seed(3)
    big = torch.randn(200, 2600, device=dev)
    x = big[:, 8:]                                           # row stride 2 600, 32-byte offset
    assert x.shape == (200, 2592)
    w, b = torch.randn(64, 2592, device=dev) * 0.02, torch.randn(64, device=dev)
    y = torch.empty(200, 64, device=dev)
    native.rows_wide_forward(x, w, b, y)
    torch.testing.assert_close(y, torch.nn.functional.gelu(torch.nn.functional.linear(x, w, b)), rtol=2e-5, atol=2e-5)
    for R, K, N in ((10, 2592, 48), (10, 2590, 64), (10, 8, 64), (0, 64, 64), (10, 20000, 64)):
        assert not native.rows_wide_supported(R, K, N)
    with pytest.raises(native.AsacNativeError):
        native.rows_wide_forward(big[:, 1:2593], w, b, y)      # rows not 16-byte aligned


@pytest.mark.parametrize('rows', [(256, 4), (2304,)])
def test_resblock_and_conv_head_take_the_wide_launches(rows):
    """`ResBlock(2592, 64)` (no residual path: the widths differ) over the frames of a step's windows runs on the wide
    launches — values and every gradient as the module path (`ASAC_ROWS_WIDE`-style switch off), in direct mode (gradients
    added into flat `.grad` views) and through plain autograd"""
    import asac_amd  # noqa: F401
    from asac_amd import native
    from algorithm import fused_rows_linear as frl
    from algorithm.nn_models.layers.linear_layers import ResBlock
    torch.manual_seed(7)
    block = ResBlock(2592, 64).cuda()
    with torch.no_grad():
        block.linear.bias.normal_(0, 0.1)
    x = torch.randn(*rows, 2592, device='cuda', requires_grad=True)
    gy = torch.randn(*rows, 64, device='cuda')
    with native.LaunchProfiler() as prof:
        y = block(x)
        y.backward(gy)
    s = prof.summary()
    assert s['asac_rows_wide_forward']['calls'] == 1 and s['asac_rows_wide_backward_input']['calls'] == 1
    assert s['asac_rows_wide_backward_params']['calls'] == 1
    got = [y.detach().clone(), x.grad.clone(), block.linear.weight.grad.clone(), block.linear.bias.grad.clone()]
    x.grad = None
    block.zero_grad(set_to_none=True)
    frl.WIDE = False
    try:
        y_ref = block(x)
        y_ref.backward(gy)
    finally:
        frl.WIDE = True
    want = [y_ref.detach(), x.grad, block.linear.weight.grad, block.linear.bias.grad]
    n_rows = int(np.prod(rows))
    for name, g, w, atol in (('y', got[0], want[0], 2e-5), ('dx', got[1], want[1], 2e-5),
                             ('dw', got[2], want[2], 4e-6 * n_rows ** 0.5 * 4), ('db', got[3], want[3], 4e-6 * n_rows ** 0.5 * 4)):
        np.testing.assert_allclose(g.cpu().numpy(), w.cpu().numpy(), rtol=3e-4, atol=atol, err_msg=name)
    with torch.no_grad(), native.LaunchProfiler() as prof:
        y_inf = block(x)
    assert list(prof.summary()) == ['asac_rows_wide_forward'] and torch.equal(y_inf, got[0])
