"""GPU: the one-launch GRU stack (`asac_gru_forward/backward`) against the cell loop it replaces —
the plugin layer's generic `nn.GRU` path run on the CPU in f32 — for outputs, per-step hidden
states, and the gradients of the input, the initial state and every cell parameter."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _layers(I, H, layers, seed=0):
    import asac_amd  # noqa: F401
    import algorithm.nn_models as m
    torch.manual_seed(seed)
    ref = m.GRU(I, H, layers)
    with torch.no_grad():
        for p in ref.parameters():
            p.uniform_(-0.6, 0.6)
    dev = copy.deepcopy(ref).cuda()
    return ref, dev


def _mask(B, L, kind, gen):
    if kind == 'none':
        return None
    lead = torch.randint(0, L - 1, (B,), generator=gen)
    lead[0] = 0
    m = torch.arange(L).unsqueeze(0) < lead.unsqueeze(1)
    if kind == 'lead+tail':       # episode ended early: padding behind the valid block too
        tail = torch.randint(0, 3, (B,), generator=gen)
        m |= torch.arange(L).unsqueeze(0) >= (L - tail).unsqueeze(1)
        m[1] = True               # a fully padded row
    return m


@pytest.mark.parametrize('I,H,layers,B,L,with_h0,mask_kind', [
    (8, 8, 2, 256, 81, True, 'lead'),        # the cfg3 burn-in shape
    (8, 8, 2, 37, 13, False, 'none'),
    (12, 6, 1, 70, 9, True, 'lead+tail'),    # hidden not a power of two, ragged block count
    (16, 16, 2, 33, 21, True, 'lead'),
    (3, 16, 2, 5, 4, False, 'lead+tail'),
])
def test_fused_gru_matches_cell_loop(I, H, layers, B, L, with_h0, mask_kind):
    from algorithm.fused_gru import fused_gru_supported
    ref, dev = _layers(I, H, layers)
    gen = torch.Generator().manual_seed(1)
    x = torch.randn(B, L, I, generator=gen)
    h0 = torch.randn(B, layers, H, generator=gen) * 0.5 if with_h0 else None
    mask = _mask(B, L, mask_kind, gen)
    g_out = torch.randn(B, L, H, generator=gen)
    g_hn = torch.randn(B, L, layers, H, generator=gen) * 0.3

    def run(layer, device):
        xd = x.detach().clone().to(device).requires_grad_(True)
        hd = None if h0 is None else h0.detach().clone().to(device).requires_grad_(True)
        md = None if mask is None else mask.to(device)
        out, hn = layer(xd, hd, md)
        loss = (out * g_out.to(device)).sum() + (hn * g_hn.to(device)).sum()
        loss.backward()
        return out, hn, xd.grad, None if hd is None else hd.grad

    assert fused_gru_supported(x.cuda(), I, H, layers)
    o_ref, hn_ref, gx_ref, gh_ref = run(ref, 'cpu')
    o_dev, hn_dev, gx_dev, gh_dev = run(dev, 'cuda')

    tol = dict(rtol=2e-4, atol=2e-5)
    torch.testing.assert_close(o_dev.cpu(), o_ref, **tol)
    torch.testing.assert_close(hn_dev.cpu(), hn_ref, **tol)
    gtol = dict(rtol=1e-3, atol=2e-4)
    torch.testing.assert_close(gx_dev.cpu(), gx_ref, **gtol)
    if with_h0:
        torch.testing.assert_close(gh_dev.cpu(), gh_ref, **gtol)
    for (n, p_ref), (_, p_dev) in zip(ref.named_parameters(), dev.named_parameters()):
        scale = max(1.0, float(p_ref.grad.abs().max()))
        torch.testing.assert_close(p_dev.grad.cpu(), p_ref.grad, rtol=1e-3, atol=2e-4 * scale, msg=lambda s: f'{n}: {s}')


def test_fused_gru_is_deterministic_and_inference_skips_gates():
    ref, dev = _layers(8, 8, 2)
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(300, 40, 8, generator=gen).cuda()
    mask = _mask(300, 40, 'lead', gen).cuda()
    grads = []
    for _ in range(2):
        dev.zero_grad(set_to_none=True)
        out, hn = dev(x, None, mask)
        (out.square().sum() + hn.sum()).backward()
        grads.append([p.grad.clone() for p in dev.parameters()])
    for a, b in zip(*grads):
        assert torch.equal(a, b)
    with torch.no_grad():
        out_ng, hn_ng = dev(x, None, mask)
    assert torch.equal(out_ng, out.detach()) and torch.equal(hn_ng, hn.detach())


def test_fused_gru_adds_parameter_gradients_in_place_when_grad_buffers_exist():
    """With dense `.grad` tensors already present (the learner's flat gradient buffer) the reduction kernel
    accumulates into them directly; the result equals the autograd-accumulated one."""
    ref, dev = _layers(8, 8, 2)
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(70, 17, 8, generator=gen).cuda()
    h0_win = torch.randn(70, 17, 2, 8, generator=gen).cuda()       # the sampled window's hidden states
    h0 = h0_win[:, 0]                                               # non-contiguous batch stride, no copy needed
    def run():
        out, hn = dev(x, h0, None)
        (out.square().sum() + 0.5 * hn.sum()).backward()
    dev.zero_grad(set_to_none=True)
    run()
    once = [p.grad.clone() for p in dev.parameters()]              # packed path (no .grad yet)
    pre = [torch.full_like(p, 0.25) for p in dev.parameters()]
    for p, g in zip(dev.parameters(), pre):
        p.grad = g.clone()
    run()                                                           # direct path: adds onto 0.25
    for p, g1 in zip(dev.parameters(), once):
        torch.testing.assert_close(p.grad, g1 + 0.25, rtol=1e-6, atol=1e-6)


def test_large_cells_use_the_generic_path():
    from algorithm.fused_gru import fused_gru_supported
    x = torch.zeros(2, 3, 64, device='cuda')
    assert not fused_gru_supported(x, 64, 64, 1)
    assert not fused_gru_supported(x, 8, 8, 3)
    assert not fused_gru_supported(x.cpu(), 8, 8, 2)


class _Rep(torch.nn.Module):
    """a representation in the plugin's shape: parameter-free glue in front of a GRU layer"""
    def __init__(self, I, H, layers, scale=1.0):
        super().__init__()
        import algorithm.nn_models as m
        self.rnn = m.GRU(I, H, layers)
        self.scale = scale

    def forward(self, a, b, h0, mask):
        return self.rnn(torch.cat([a, b], dim=-1) * self.scale, h0, mask)


@pytest.mark.parametrize('H,layers,B,L', [(8, 2, 256, 81), (6, 1, 45, 7), (16, 2, 33, 12)])
def test_twin_pass_equals_two_passes(H, layers, B, L):
    """online + target representation over the same window in one launch (`asac_gru_forward_twin`):
    bit-identical to the two separate launches, gradients of the online pass unchanged, and the shortcut is
    only taken after the verification passes."""
    import asac_amd  # noqa: F401
    from asac_amd import native
    from algorithm.fused_gru import TwinPass
    torch.manual_seed(3)
    online, target = _Rep(9, H, layers).cuda(), _Rep(9, H, layers).cuda()
    a, b = torch.randn(B, L, 5, device='cuda'), torch.randn(B, L, 4, device='cuda')
    h0 = torch.randn(B, layers, H, device='cuda')
    mask = torch.arange(L, device='cuda').unsqueeze(0) < torch.randint(0, L - 1, (B, 1), device='cuda')

    def both(twin):
        online.zero_grad()
        ctx = twin if twin is not None else __import__('contextlib').nullcontext()
        with native.LaunchProfiler() as prof:
            with ctx:
                out, hn = online(a, b, h0, mask)
                with torch.no_grad():
                    t_out, t_hn = target(a, b, h0, mask)
            (out.sum() + (hn * 0.5).sum()).backward()
        grads = [p.grad.clone() for p in online.parameters()]
        return (out, hn, t_out, t_hn, *grads), prof.summary()

    want, _ = both(None)
    twin = TwinPass(online, target, verify_steps=2)
    assert twin and not twin.trusted
    for step in range(4):
        got, launches = both(twin)
        for w, g in zip(want, got):
            assert torch.equal(w, g)
        if step < 2:      # verification: the paired launch AND the target's own
            assert launches['asac_gru_forward_twin']['calls'] == 1 and launches['asac_gru_forward']['calls'] == 1
        else:
            assert twin.trusted and 'asac_gru_forward' not in launches
            assert launches['asac_gru_forward_twin']['calls'] == 1
    # a target module that feeds its GRU something else is caught during verification and never paired again
    other = _Rep(9, H, layers, scale=0.5).cuda()
    twin = TwinPass(online, other, verify_steps=2)
    for step in range(3):
        with twin:
            online(a, b, h0, mask)
            with torch.no_grad():
                t_out, _ = other(a, b, h0, mask)
        with torch.no_grad():
            assert torch.equal(t_out, other(a, b, h0, mask)[0])
    assert twin.failed and not twin
    # an input that depends on trainable parameters is never paired
    twin = TwinPass(online, target, verify_steps=0)
    a_param = a.clone().requires_grad_(True)
    with native.LaunchProfiler() as prof, twin:
        online(a_param, b, h0, mask)
        with torch.no_grad():
            target(a, b, h0, mask)
    assert 'asac_gru_forward_twin' not in prof.summary()


@pytest.mark.parametrize('I,H,layers,B,L,position,E,with_h0,mask_kind,want_gx', [
    (8, 8, 2, 256, 81, 40, 2, True, 'lead', False),      # the cfg3 step: state at burn_in_step of a window of 81
    (8, 8, 2, 37, 13, 12, 3, False, 'none', True),       # the last position: the whole window is walked
    (12, 6, 1, 70, 9, 0, 1, True, 'lead+tail', True),    # the first position: one step
    (16, 16, 2, 33, 21, 7, 2, True, 'lead', True),       # position at a chunk's last slot
    (3, 16, 1, 5, 17, 8, 4, False, 'lead+tail', False),  # ... and at a chunk's first
])
def test_backward_from_one_position_equals_dense_backward(I, H, layers, B, L, position, E, with_h0, mask_kind, want_gx):
    """`asac_gru_backward_at` (members summed in the launch, recursion started at the position) == `asac_gru_backward`
    on the dense gradient that is sum_e members[e] at the position and zero elsewhere: every output bit for bit."""
    from asac_amd import native
    _, dev = _layers(I, H, layers)
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(B, L, I, generator=gen).cuda()
    h0 = (torch.randn(B, layers, H, generator=gen) * 0.5).cuda() if with_h0 else None
    mask = _mask(B, L, mask_kind, gen)
    mask = None if mask is None else mask.cuda()
    members = torch.randn(E, B, H, generator=gen).cuda()
    desc = native.gru_desc(I, H, layers)
    w = [tuple(getattr(c, n).detach().contiguous() for n in ('weight_ih_l0', 'weight_hh_l0', 'bias_ih_l0', 'bias_hh_l0'))
         for c in dev._grus]
    hn = torch.empty(B, L, layers, H, device='cuda')
    out = torch.empty(B, L, H, device='cuda')
    gates = torch.empty(B, L, layers, 5 * H, device='cuda')
    native.gru_forward(desc, w, x, h0, mask, hn, out, gates)

    def run(at):
        g_x = torch.full((B, L, I), 7.0, device='cuda') if want_gx else None
        g_h0 = torch.empty(B, layers, H, device='cuda') if with_h0 else None
        g_p = torch.empty(native.gru_param_count(desc), device='cuda')
        ws = torch.empty(native.gru_backward_workspace(desc, B), device='cuda')
        if at:
            native.gru_backward_at(desc, w, x, h0, mask, hn, gates, members, position, g_x, g_h0, g_p, None, False, ws)
        else:
            dense = torch.zeros(B, L, H, device='cuda')
            acc = members[0].clone()
            for e in range(1, E):
                acc = acc + members[e]
            dense[:, position] = acc
            native.gru_backward(desc, w, x, h0, mask, hn, gates, None, dense, g_x, g_h0, g_p, None, False, ws)
        torch.cuda.synchronize()
        return g_x, g_h0, g_p

    for name, a, b in zip(('grad_x', 'grad_h0', 'grad_params'), run(True), run(False)):
        if a is not None:
            assert torch.equal(a, b), name
    assert run(True)[2].abs().sum() > 0


def test_autograd_route_of_backward_from_position():
    """`fused_gru.backward_from_position` through autograd == `torch.autograd.backward` with the dense gradient"""
    import asac_amd  # noqa: F401
    from algorithm import fused_gru
    _, dev = _layers(8, 8, 2)
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(64, 21, 8, generator=gen).cuda()
    members = torch.randn(2, 64, 8, generator=gen).cuda()
    grads = []
    for at in (True, False):
        dev.zero_grad(set_to_none=True)
        top, _ = dev(x)
        assert fused_gru.is_fused_top(top) and not fused_gru.is_fused_top(top * 1.0)
        dense = torch.zeros_like(top)
        if at:
            fused_gru.backward_from_position(top, members, 9, dense)
        else:
            dense[:, 9] = members[0] + members[1]
            torch.autograd.backward([top], [dense])
        grads.append([p.grad.clone() for p in dev.parameters()])
    assert all(torch.equal(a, b) for a, b in zip(*grads))


def test_adam_epilogue_of_the_gradient_reduction_equals_a_separate_adam_launch():
    """`asac_gru_backward_at(..., adam)`: the launch that finishes the cell parameters' gradients steps them —
    parameters, moments and gradients bit-equal to the same launch followed by `asac_adam_step` over the flat buffers."""
    from asac_amd import native
    I, H, layers, B, L, position = 8, 8, 2, 96, 21, 9
    _, dev = _layers(I, H, layers)
    gen = torch.Generator().manual_seed(11)
    x = torch.randn(B, L, I, generator=gen).cuda()
    members = torch.randn(2, B, H, generator=gen).cuda()
    desc = native.gru_desc(I, H, layers)
    n = native.gru_param_count(desc)
    cells = [[getattr(c, k).detach() for k in ('weight_ih_l0', 'weight_hh_l0', 'bias_ih_l0', 'bias_hh_l0')] for c in dev._grus]
    steps = torch.full((1,), 6, dtype=torch.int64, device='cuda')
    results = []
    for fused in (True, False):
        flat, grad = torch.zeros(n + 5, device='cuda'), torch.full((n + 5,), 0.25, device='cuda')
        m, v = torch.full((n + 5,), 0.01, device='cuda'), torch.full((n + 5,), 0.002, device='cuda')
        w, gt, off = [], [], 3                         # the cells' tensors as views of flat buffers, 3 floats in
        for cell in cells:
            ws_, gs_ = [], []
            for t in cell:
                k = t.numel()
                flat[off:off + k] = t.reshape(-1)
                ws_.append(flat[off:off + k].view(t.shape))
                gs_.append(grad[off:off + k].view(t.shape))
                off += k
            w.append(tuple(ws_))
            gt.append(tuple(gs_))
        hn = torch.empty(B, L, layers, H, device='cuda')
        gates = torch.empty(B, L, layers, 5 * H, device='cuda')
        native.gru_forward(desc, w, x, None, None, hn, None, gates)
        ws = torch.empty(native.gru_backward_workspace(desc, B), device='cuda')
        ep = native.adam_epilogue(flat, grad, m, v, 3e-4, 0.9, 0.999, 1e-8, steps) if fused else None
        native.gru_backward_at(desc, w, x, None, None, hn, gates, members, position, None, None, None, gt, True, ws, adam=ep)
        if not fused:
            native.adam_step(flat[3:3 + n], grad[3:3 + n], m[3:3 + n], v[3:3 + n], 3e-4, 0.9, 0.999, 1e-8, steps)
        torch.cuda.synchronize()
        results.append((flat.clone(), grad.clone(), m.clone(), v.clone()))
    for name, a, b in zip(('param', 'grad', 'exp_avg', 'exp_avg_sq'), *results):
        assert torch.equal(a[3:3 + n], b[3:3 + n]), name
        assert torch.equal(a[:3], b[:3]) and torch.equal(a[3 + n:], b[3 + n:]), name
    assert int(steps.item()) == 6 and not torch.equal(results[0][0][3:3 + n], torch.cat([t.reshape(-1) for c in cells for t in c]))
