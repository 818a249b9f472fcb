"""GPU: `asac_xty` (csrc/xty.hip) — out = x^T y over the rows and the column sums of x — against float64 products, at the
shapes of its callers (Linear 64 x 64 over 9 216 rows, GRU gates 192 x 64 / 192 x 8 over 20 736 rows, 384 x 128) and at ragged
ones, with row strides and accumulation; `rows_linear` (the `nn.Linear` of `LinearLayers` / `ResBlock` with that backward)
against the plain module; bit-determinism."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('R,M,N', [(9216, 64, 64), (20736, 192, 64), (20736, 192, 8), (4096, 384, 128), (1000, 100, 37),
                                   (17, 5, 3), (1, 16, 16), (70000, 48, 96)])
def test_xty_matches_float64(R, M, N):
    import asac_amd  # noqa: F401
    from asac_amd import native
    gen = torch.Generator(device='cuda').manual_seed(R + M + N)
    xb = torch.randn(R, M + 5, device='cuda', generator=gen)
    yb = torch.randn(R, N + 3, device='cuda', generator=gen)
    x, y = xb[:, 2:2 + M], yb[:, 1:1 + N]                       # row strides, unaligned starts
    out, cs = torch.empty(M, N, device='cuda'), torch.empty(M, device='cuda')
    native.xty(x, y, out, cs)
    want = (x.double().t() @ y.double())
    want_cs = x.double().sum(0)
    tol = 3e-6 * np.sqrt(R) + 1e-6
    assert float((out.double() - want).abs().max()) <= tol * float(want.abs().max() + 1)
    assert float((cs.double() - want_cs).abs().max()) <= tol * float(want_cs.abs().max() + 1)
    out2, cs2 = torch.empty(M, N, device='cuda'), torch.empty(M, device='cuda')
    native.xty(x, y, out2, cs2)
    assert torch.equal(out, out2) and torch.equal(cs, cs2), 'fixed summation order: bit-identical across launches'
    base, base_cs = torch.randn(M, N, device='cuda', generator=gen), torch.randn(M, device='cuda', generator=gen)
    acc, acc_cs = base.clone(), base_cs.clone()
    native.xty(x, y, acc, acc_cs, accumulate=True)
    assert torch.equal(acc, base + out) and torch.equal(acc_cs, base_cs + cs)
    only = torch.empty(M, N, device='cuda')
    native.xty(x, y, only)                                       # no column sums asked for
    assert torch.equal(only, out)


def test_rows_linear_is_the_module_with_another_backward():
    import asac_amd  # noqa: F401
    from asac_amd import native
    import algorithm.nn_models as m
    from algorithm import fused_rows_linear
    torch.manual_seed(0)
    ref = m.LinearLayers(40, dense_n=64, dense_depth=2, output_size=24).cuda()
    dev = copy.deepcopy(ref)
    x = torch.randn(1024, 9, 40, device='cuda')
    g = torch.randn(1024, 9, 24, device='cuda')

    def run(layer, enabled):
        fused_rows_linear.ENABLED = enabled
        try:
            xi = x.clone().requires_grad_(True)
            with native.LaunchProfiler(repeat=1) as prof:
                out = layer(xi)
                (out * g).sum().backward()
            return out.detach(), xi.grad, [p.grad for p in layer.parameters()], prof.summary()
        finally:
            fused_rows_linear.ENABLED = True

    o_r, gx_r, gp_r, seen_r = run(ref, False)
    o_d, gx_d, gp_d, seen_d = run(dev, True)
    assert 'asac_xty' not in seen_r and seen_d['asac_xty']['calls'] == 3
    # (the two ResBlocks of the stack run as one launch per pass now — `asac_rows_affine_gelu_forward`, `asac_rows_resblock_*`:
    # the library's own GELU arithmetic and MFMA order, not the library product bit for bit)
    assert seen_d['asac_rows_affine_gelu_forward']['calls'] == 1 and seen_d['asac_rows_resblock_forward']['calls'] == 1
    np.testing.assert_allclose(o_d.cpu().numpy(), o_r.cpu().numpy(), rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(gx_d.cpu().numpy(), gx_r.cpu().numpy(), rtol=1e-4, atol=2e-5)
    for a, b in zip(gp_d, gp_r):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=2e-4, atol=2e-4 * float(b.abs().max()))
    # short inputs and inference keep the plain module
    with native.LaunchProfiler(repeat=1) as prof:
        dev(torch.randn(64, 40, device='cuda', requires_grad=True)).sum().backward()
        with torch.no_grad():
            dev(x)
    assert 'asac_xty' not in prof.summary()
