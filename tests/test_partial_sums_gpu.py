"""GPU: `asac_sum_partials_multi` / `fused_mlp.DeferredPartialSums` — the second launches of several backwards (per-workgroup
parameter-gradient partials summed in workgroup order) as ONE launch: per job the bits of the launch it stands for."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dense(in_size, widths, out, seed=0):
    import asac_amd  # noqa: F401
    import algorithm.nn_models as m
    from algorithm.fused import FlatParamGroup
    torch.manual_seed(seed)
    dev = m.LinearLayers(in_size, widths, len(widths), out).cuda()
    group = FlatParamGroup([('dense', list(dev.parameters()))], 'cuda')
    dev.fuse = True
    return dev, group


def test_sum_partials_multi_is_the_single_reductions_bit_for_bit():
    """sliced (16 slices of slabs) and sequential jobs of ragged sizes beside each other, one accumulating"""
    import asac_amd  # noqa: F401
    from asac_amd import native
    gen = torch.Generator(device='cuda').manual_seed(0)
    jobs, want = [], []
    for slabs, slices, n, stride, acc in [(288, 16, 8776, 8776, False), (17, 1, 300, 512, False), (64, 16, 288, 288, True),
                                         (1, 1, 5, 5, False), (129, 16, 64, 70, False), (63, 1, 1000, 1000, True)]:
        partial = torch.randn(slabs * stride, generator=gen, device='cuda')
        out = torch.randn(n + 3, generator=gen, device='cuda')
        slab = partial.view(slabs, stride)[:, :n]
        if slices == 1:
            s = torch.zeros(n, device='cuda')
            for t in range(slabs):
                s = s + slab[t]
        else:
            per = (slabs + 15) // 16
            s = torch.zeros(n, device='cuda')
            for sl in range(16):
                part = torch.zeros(n, device='cuda')
                for t in range(sl * per, min((sl + 1) * per, slabs)):
                    part = part + slab[t]
                s = s + part
        want.append(torch.cat([out[:n] + s if acc else s, out[n:]]))
        jobs.append((partial, slabs, slices, stride, n, out, acc))
    native.sum_partials_multi(jobs)
    for (_, _, _, _, _, out, _), w in zip(jobs, want):
        assert torch.equal(out, w)


@pytest.mark.parametrize('N', [4608, 300])          # 144 / 10 row tiles: the sliced and the sequential reduction
def test_deferred_sums_of_dense_stack_and_attention_are_the_undeferred_gradients(N):
    """three `autograd.grad` walks through a fused dense stack and an attention block with their second launches deferred:
    the gradients `flush()` hands back equal the ones the undeferred walks return, bit for bit, from ONE launch"""
    import asac_amd  # noqa: F401
    from asac_amd import native
    import algorithm.nn_models as m
    from algorithm.fused_mlp import DeferredPartialSums
    from algorithm.fused_conv import conv_stack_desc, fused_conv_stack
    dense, _ = _dense(128, [64, 64], 8)
    torch.manual_seed(1)
    attn = m.MultiheadAttention(8, 1, out_dense_depth=1).cuda()
    conv = torch.nn.Sequential(torch.nn.Conv2d(3, 16, 8, 4), torch.nn.GELU(), torch.nn.Conv2d(16, 32, 4, 2), torch.nn.GELU()).cuda()
    B = N // 9
    frames = torch.randn(N // 4, 3, 30, 30, device='cuda')
    x = torch.randn(N, 128, device='cuda')
    key = torch.randn(B, 9, 8, device='cuda', requires_grad=True)
    y = dense(x)
    o, w = attn(key[:, -9:], key, key)
    f = fused_conv_stack(frames, conv_stack_desc(conv, frames), conv)
    cots = [(torch.randn_like(y), torch.randn_like(o), torch.randn_like(f)) for _ in range(3)]
    dp, ap = list(dense.parameters()), list(attn.parameters()) + list(conv.parameters())

    def walks():
        got = []
        for k, (gy, go, gf) in enumerate(cots):
            later = DeferredPartialSums.active()
            if later is not None:
                later.walk = k
            got.append(list(torch.autograd.grad([y, o, f], dp + ap, grad_outputs=[gy, go, gf], retain_graph=True,
                                                allow_unused=True)))
        return got

    want = walks()
    assert all(g is not None for g in want[0])
    with native.LaunchProfiler(repeat=1) as prof:
        with DeferredPartialSums() as later:
            got = walks()
        assert all(g is None for walk in got for g in walk)          # (nothing handed to autograd: nothing read early)
        late = later.flush()
    seen = prof.summary()
    assert seen['asac_sum_partials_multi']['calls'] == 1 and seen['asac_mlp_backward']['calls'] == 3
    assert seen['asac_attention_proj_backward']['calls'] == 3 and seen['asac_conv2_backward']['calls'] == 3
    for k in range(3):
        for p, g in zip(dp + ap, want[k]):
            assert torch.equal(late[k][id(p)], g), (k, tuple(p.shape))
    # the switch off: the walks return their gradients themselves
    import algorithm.fused_mlp as fm
    fm.SUM_PARTIALS_LATER = False
    try:
        with DeferredPartialSums() as later:
            got = walks()
        assert later.flush() == {} and all(torch.equal(a, b) for a, b in zip(got[2], want[2]))
    finally:
        fm.SUM_PARTIALS_LATER = True


def test_direct_mode_backward_with_deferred_sums_adds_the_same_gradients():
    """the learner's direct mode (parameter gradients ADDED into the flat `.grad` views by the backward launches) with the
    second launches deferred: one launch, the same sums"""
    import asac_amd  # noqa: F401
    from asac_amd import native
    import algorithm.nn_models as m
    from algorithm.fused import FlatParamGroup
    from algorithm.fused_conv import conv_stack_desc, fused_conv_stack
    from algorithm.fused_mlp import DeferredPartialSums, direct_param_grads
    torch.manual_seed(0)
    dense = m.LinearLayers(128, [64, 64], 2, 8).cuda()
    attn = m.MultiheadAttention(8, 1, out_dense_depth=1).cuda()
    conv = torch.nn.Sequential(torch.nn.Conv2d(3, 16, 8, 4), torch.nn.GELU(), torch.nn.Conv2d(16, 32, 4, 2), torch.nn.GELU()).cuda()
    group = FlatParamGroup([('dense', list(dense.parameters())), ('attn', list(attn.parameters())),
                            ('conv', list(conv.parameters()))], 'cuda')
    dense.fuse = True
    frames = torch.randn(1152, 3, 30, 30, device='cuda')
    key = torch.randn(512, 9, 8, device='cuda')

    def loss():
        f = fused_conv_stack(frames, conv_stack_desc(conv, frames), conv)          # [1152, 128]
        o, _ = attn(key[:, -9:], key, key)
        return dense(f).square().mean() + o.square().mean()

    group.grad.normal_()                  # (the launches ADD: something to add to)
    start = group.grad.clone()
    with direct_param_grads():
        loss().backward()
    want = group.grad.clone()
    assert not torch.equal(want, start)
    group.grad.copy_(start)
    with native.LaunchProfiler(repeat=1) as prof:
        with direct_param_grads(), DeferredPartialSums() as later:
            loss().backward()
        mid = group.grad.clone()
        later.flush()
    assert torch.equal(mid, start), 'a deferred backward wrote parameter gradients before the flush'
    assert torch.equal(group.grad, want)
    assert prof.summary()['asac_sum_partials_multi']['calls'] == 1


def test_a_stack_used_twice_in_one_walk_gets_both_contributions():
    """a parameter two recorded nodes of one walk share: `flush()` hands back the sum autograd would have formed"""
    import asac_amd  # noqa: F401
    from algorithm.fused_mlp import DeferredPartialSums
    dense, _ = _dense(40, [64], 8)
    x1, x2 = torch.randn(300, 40, device='cuda'), torch.randn(300, 40, device='cuda')
    y = dense(x1) + 2.0 * dense(x2)
    gy = torch.randn_like(y)
    params = list(dense.parameters())
    want = torch.autograd.grad(y, params, grad_outputs=gy, retain_graph=True)
    with DeferredPartialSums() as later:
        got = torch.autograd.grad(y, params, grad_outputs=gy, retain_graph=True, allow_unused=True)
    assert all(g is None for g in got)
    late = later.flush()[0]
    for p, w in zip(params, want):
        np.testing.assert_allclose(late[id(p)].cpu().numpy(), w.cpu().numpy(), rtol=1e-6, atol=1e-7)


def test_linear_tanh_head_defers_its_sums_too():
    """the fused Linear + Tanh state head under `DeferredPartialSums`: no reduction of its own, the flushed gradients equal the
    undeferred ones to rounding (another grouping of the same workgroup partials)"""
    import asac_amd  # noqa: F401
    from asac_amd import native
    from algorithm.fused_linear import fuse_linear_tanh_heads
    from algorithm.fused_mlp import DeferredPartialSums
    torch.manual_seed(0)
    model = torch.nn.ModuleList([torch.nn.Sequential(torch.nn.Linear(18, 8), torch.nn.Tanh())]).cuda()
    assert fuse_linear_tanh_heads(model) == 1
    head = model[0]
    x = torch.randn(9216, 18, device='cuda', requires_grad=True)
    y = head(x)
    gy = torch.randn_like(y)
    params = list(head.parameters())
    want = torch.autograd.grad(y, [x] + params, grad_outputs=gy, retain_graph=True)
    with native.LaunchProfiler(repeat=1) as prof:
        with DeferredPartialSums() as later:
            got = torch.autograd.grad(y, [x] + params, grad_outputs=gy, retain_graph=True, allow_unused=True)
        late = later.flush()[0]
    assert torch.equal(got[0], want[0]) and got[1] is None and got[2] is None
    assert prof.summary()['asac_sum_partials_multi']['calls'] == 1
    for p, w in zip(params, want[1:]):
        np.testing.assert_allclose(late[id(p)].cpu().numpy(), w.cpu().numpy(), rtol=2e-5, atol=2e-5)
