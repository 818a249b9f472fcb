"""GPU: the Linear + Tanh state head (csrc/linear.hip through the C ABI) against plain PyTorch f32 on the same
inputs — forward, input gradient, parameter gradients (returned and added into a flat gradient buffer), ragged row
counts, strided inputs, the widest supported layer — and the re-classed plugin module against the module path.
Tolerances: f32 rounding of a <= 64-term dot product and of tanh (rtol 1e-5 / atol 1e-6 forward; the parameter
gradients sum up to 10^4 rows, so their atol scales with the row count)."""
import pytest
import torch
from torch import nn

pytestmark = pytest.mark.gpu


def _ref(x, w, b, gy):
    x = x.detach().clone().requires_grad_(True)
    w = w.detach().clone().requires_grad_(True)
    b = b.detach().clone().requires_grad_(True)
    y = torch.tanh(torch.nn.functional.linear(x, w, b))
    y.backward(gy)
    return y.detach(), x.grad, w.grad, b.grad


@pytest.mark.parametrize('N,K,O', [(1, 1, 1), (127, 18, 8), (128, 8, 8), (129, 5, 3), (4608, 18, 8), (9216, 8, 8),
                                   (1000, 64, 16), (333, 33, 16), (9216, 64, 8), (20736, 64, 16), (9217, 63, 8)])
def test_linear_tanh_kernels(N, K, O):
    from asac_amd import native
    torch.manual_seed(N + K)
    dev = 'cuda:0'
    x = torch.randn(N, K, device=dev)
    w = torch.randn(O, K, device=dev) * 0.3
    b = torch.randn(O, device=dev) * 0.1
    gy = torch.randn(N, O, device=dev)
    y_ref, gx_ref, gw_ref, gb_ref = _ref(x, w, b, gy)

    y = torch.empty(N, O, device=dev)
    native.linear_tanh_forward(x, w, b, y)
    torch.testing.assert_close(y, y_ref, rtol=1e-5, atol=2e-6)

    ws = torch.zeros(native.linear_tanh_workspace(N, K, O), device=dev)
    gx = torch.empty(N, K, device=dev)
    g = torch.full((O * K + O,), 7.0, device=dev)
    native.linear_tanh_backward(x, w, y, gy, gx, g, False, ws)
    atol = 2e-6 * max(N, 16) ** 0.5 * 4
    torch.testing.assert_close(gx, gx_ref, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(g[:O * K].view(O, K), gw_ref, rtol=1e-4, atol=atol)
    torch.testing.assert_close(g[O * K:], gb_ref, rtol=1e-4, atol=atol)
    assert float(ws[-1].view(torch.int32)) == 0          # the arrival counter is back at zero

    # accumulate into an existing buffer, no input gradient, second use of the same workspace
    base = torch.randn(O * K + O, device=dev)
    g2 = base.clone()
    native.linear_tanh_backward(x, w, y, gy, None, g2, True, ws)
    torch.testing.assert_close(g2 - base, g, rtol=1e-4, atol=atol)
    # same inputs -> same bits (the partial sums are combined in workgroup order)
    g3 = torch.empty_like(g)
    native.linear_tanh_backward(x, w, y, gy, None, g3, False, ws)
    assert torch.equal(g3, g)


def test_linear_tanh_strided_rows_and_limits():
    from asac_amd import native
    dev = 'cuda:0'
    torch.manual_seed(0)
    wide = torch.randn(300, 40, device=dev)
    x = wide[:, 5:23]                                   # row stride 40, 18 columns
    w, b = torch.randn(8, 18, device=dev), torch.randn(8, device=dev)
    y = torch.empty(300, 8, device=dev)
    native.linear_tanh_forward(x, w, b, y)
    torch.testing.assert_close(y, torch.tanh(x @ w.t() + b), rtol=1e-5, atol=2e-6)
    assert native.linear_tanh_workspace(10, 65, 8) == -1 and native.linear_tanh_workspace(10, 8, 17) == -1
    with pytest.raises(native.AsacNativeError):
        native.linear_tanh_forward(torch.randn(4, 65, device=dev), torch.randn(8, 65, device=dev), b, torch.empty(4, 8, device=dev))


@pytest.mark.parametrize('flat_grads', [False, True])
def test_reclassed_head_matches_module_path(flat_grads):
    import asac_amd  # noqa: F401
    import algorithm.fused_linear as fl

    class Rep(nn.Module):
        def __init__(self):
            super().__init__()
            self.enc = nn.Linear(10, 12)
            self.dense = nn.Sequential(nn.Linear(18, 8), nn.Tanh())
            self.other = nn.Sequential(nn.Linear(8, 8), nn.ReLU())          # not a match
            self.wide = nn.Sequential(nn.Linear(100, 8), nn.Tanh())         # too wide: stays as it is

        def forward(self, vec, extra):
            return self.other(self.dense(torch.cat([extra, self.enc(vec)], dim=-1)))

    torch.manual_seed(3)
    dev = 'cuda:0'
    rep = Rep().to(dev)
    keys = list(rep.state_dict())
    assert fl.fuse_linear_tanh_heads(rep) == 1
    assert type(rep.dense) is fl.LinearTanhHead and type(rep.wide) is nn.Sequential and list(rep.state_dict()) == keys
    params = list(rep.parameters())
    if flat_grads:          # the learner's layout: every .grad a view of one zeroed flat buffer
        flat = torch.zeros(sum(p.numel() for p in params), device=dev)
        off = 0
        for p in params:
            p.grad = flat[off:off + p.numel()].view_as(p)
            off += p.numel()
    vec, extra = torch.randn(64, 9, 10, device=dev), torch.randn(64, 9, 6, device=dev, requires_grad=True)
    results = []
    for fused in (True, False):
        fl.FUSED_LINEAR_TANH = fused
        try:
            for p in params:
                if p.grad is not None:
                    p.grad.zero_()
            extra.grad = None
            out = rep(vec, extra)
            (out * torch.linspace(-1, 1, 8, device=dev)).sum().backward()
            results.append([out.detach().clone(), extra.grad.clone()] + [p.grad.clone() for p in params if p.grad is not None])
        finally:
            fl.FUSED_LINEAR_TANH = True
    assert len(results[0]) == len(results[1]) >= 6
    for a, b in zip(*results):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=2e-5)
    with torch.no_grad():
        assert rep.dense(torch.randn(0, 18, device=dev)).shape == (0, 8)     # empty input: module path
    import copy
    clone = copy.deepcopy(rep)
    assert type(clone.dense) is fl.LinearTanhHead
    torch.testing.assert_close(clone(vec, extra), rep(vec, extra))


@pytest.mark.parametrize('B,L,position,E,K0,K1,O', [(512, 1, 0, 4, 10, 8, 8), (100, 9, 5, 2, 8, 0, 8), (33, 4, 3, 3, 5, 7, 6),
                                                    (257, 1, 0, 1, 40, 24, 16)])
def test_two_block_input_and_member_gradients_equal_the_materialised_forms(B, L, position, E, K0, K1, O):
    """`asac_linear_tanh_forward2 / _backward2`: the input read as two blocks side by side, the output gradient given as
    E members at one window position and summed in the launch == the one-input launches on the concatenation and on the
    dense gradient (sum_e members at the position, zero elsewhere): every output bit for bit."""
    from asac_amd import native
    torch.manual_seed(B + L)
    dev, N = 'cuda:0', B * L
    wide0 = torch.randn(N, K0 + 3, device=dev)
    x0 = wide0[:, 1:1 + K0]                                   # (a row stride)
    x1 = torch.randn(N, K1, device=dev) if K1 else None
    w = torch.randn(O, K0 + K1, device=dev) * 0.3
    b = torch.randn(O, device=dev) * 0.1
    members = torch.randn(E, B, O, device=dev)
    x = torch.cat([x0, x1], dim=-1) if K1 else x0.contiguous()
    y, y2 = torch.empty(N, O, device=dev), torch.empty(N, O, device=dev)
    native.linear_tanh_forward(x, w, b, y)
    native.linear_tanh_forward2(x0, x1, w, b, y2)
    assert torch.equal(y, y2)
    dense = torch.zeros(B, L, O, device=dev)
    acc = members[0].clone()
    for e in range(1, E):
        acc = acc + members[e]
    dense[:, position] = acc
    K = K0 + K1
    ws = torch.zeros(native.linear_tanh_workspace(N, K, O), device=dev)
    gx, g = torch.empty(N, K, device=dev), torch.empty(O * K + O, device=dev)
    native.linear_tanh_backward(x, w, y, dense.view(N, O), gx, g, False, ws)
    gx0, gx1 = torch.empty(N, K0, device=dev), (torch.empty(N, K1, device=dev) if K1 else None)
    g2 = torch.empty(O * K + O, device=dev)
    native.linear_tanh_backward2(x0, x1, w, y, members, gx0, gx1, g2, False, ws, members=E, window=L, position=position)
    assert torch.equal(g, g2) and torch.equal(gx[:, :K0], gx0) and (not K1 or torch.equal(gx[:, K0:], gx1))
    if K1:      # only the second block's gradient asked for (the first is data)
        g3, gx1b = torch.empty(O * K + O, device=dev), torch.empty(N, K1, device=dev)
        native.linear_tanh_backward2(x0, x1, w, y, members, None, gx1b, g3, False, ws, members=E, window=L, position=position)
        assert torch.equal(g, g3) and torch.equal(gx1, gx1b)


def test_deferred_concatenation_reaches_the_fused_head_in_two_blocks():
    """Inside `AdjacentCat(defer_width)` the plugin's `self.dense(torch.cat([vec, features], -1))` runs the two-block
    launch (no concatenation is formed), with the gradients of the module path; anything else done to the deferred
    object gets the real concatenation."""
    import asac_amd  # noqa: F401
    from algorithm import fused_linear as fl
    from algorithm.adjacent_cat import AdjacentCat, DeferredCat
    torch.manual_seed(0)
    dev = 'cuda:0'
    head = nn.Sequential(nn.Linear(18, 8), nn.Tanh()).to(dev)
    ref = nn.Sequential(nn.Linear(18, 8), nn.Tanh()).to(dev)
    ref.load_state_dict(head.state_dict())
    holder = nn.Module()
    holder.dense = head
    assert fl.fuse_linear_tanh_heads(holder) == 1
    vec = torch.randn(64, 9, 10, device=dev)
    feat = torch.randn(64, 9, 8, device=dev, requires_grad=True)
    feat_ref = feat.detach().clone().requires_grad_(True)
    gy = torch.randn(64, 9, 8, device=dev)
    with AdjacentCat(64):
        cat = torch.cat([vec, feat], dim=-1)
        assert isinstance(cat, DeferredCat) and cat._value is None
        out = holder.dense(cat)
        assert cat._value is None                       # never materialised
        other = torch.cat([vec, feat], dim=-1)
        assert other.shape == (64, 9, 18) and other._value is not None          # an attribute: the real tensor
        assert torch.equal(other + 0, torch.cat([vec.clone(), feat.detach()], -1))
        assert torch.equal(torch.relu(torch.cat([vec, feat], -1)), torch.relu(torch.cat([vec.clone(), feat.detach()], -1)))
        wide = torch.cat([torch.randn(4, 60, device=dev), torch.randn(4, 60, device=dev)], -1)
        assert isinstance(wide, torch.Tensor)           # too wide for the head: ATen's concatenation
    out.backward(gy)
    want = ref(torch.cat([vec, feat_ref], dim=-1))
    want.backward(gy)
    torch.testing.assert_close(out, want, rtol=1e-5, atol=2e-6)
    torch.testing.assert_close(feat.grad, feat_ref.grad, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(head[0].weight.grad, ref[0].weight.grad, rtol=1e-4, atol=2e-4)
    torch.testing.assert_close(head[0].bias.grad, ref[0].bias.grad, rtol=1e-4, atol=2e-4)


def test_no_grad_head_reads_a_window_slice_in_place():
    """A no-grad call of the fused head on [vec[:, b:] | features] (vec a non-collapsible slice of the sampled windows)
    == the same call on contiguous copies, without a copy launch (`asac_linear_tanh_forward2w`)."""
    import asac_amd  # noqa: F401
    from asac_amd import native
    from algorithm import fused_linear as fl
    from algorithm.adjacent_cat import AdjacentCat
    torch.manual_seed(1)
    dev = 'cuda:0'
    holder = nn.Module()
    holder.dense = nn.Sequential(nn.Linear(18, 8), nn.Tanh()).to(dev)
    assert fl.fuse_linear_tanh_heads(holder) == 1
    joint = torch.randn(33, 9, 14, device=dev)
    vec = joint[:, 5:, :10]                                   # [33, 4, 10], rows 14 apart, samples 126 apart
    feat = torch.randn(33, 4, 8, device=dev)
    with torch.no_grad():
        want = holder.dense(torch.cat([vec.contiguous(), feat], dim=-1))
        with AdjacentCat(64), native.LaunchProfiler() as prof:
            got = holder.dense(torch.cat([vec, feat], dim=-1))
        assert torch.equal(got, want) and prof.summary()['asac_linear_tanh_forward2']['calls'] == 1
        assert torch.equal(holder.dense(vec.expand(33, 4, 10)[..., :10].new_zeros(33, 4, 18)), holder.dense(torch.zeros(33, 4, 18, device=dev)))
    # with gradients wanted: the same slice addressing forward and backward (asac_linear_tanh_backward2w), same values
    feat_g = feat.clone().requires_grad_(True)
    with AdjacentCat(64):
        out = holder.dense(torch.cat([vec, feat_g], dim=-1))
    assert torch.equal(out, want)
    out.sum().backward()
    got_w, got_f = holder.dense[0].weight.grad.clone(), feat_g.grad.clone()
    holder.zero_grad(set_to_none=True)
    feat_c = feat.clone().requires_grad_(True)
    holder.dense(torch.cat([vec.contiguous(), feat_c], dim=-1)).sum().backward()
    assert torch.equal(got_w, holder.dense[0].weight.grad) and torch.equal(got_f, feat_c.grad)
