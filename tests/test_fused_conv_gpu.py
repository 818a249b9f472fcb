"""GPU: the one-launch convolution stack (`asac_conv2_forward/backward`) against the module stack it
replaces — Conv2d GELU Conv2d GELU run by PyTorch on the CPU in f32 — for the flattened activations and
the gradients of the four parameter tensors."""
import copy

import numpy as np
import pytest
import torch
from torch import nn

pytestmark = pytest.mark.gpu


def _stack(C, o1, k1, s1, o2, k2, s2, seed=0):
    torch.manual_seed(seed)
    ref = nn.Sequential(nn.Conv2d(C, o1, [k1, k1], [s1, s1]), nn.GELU(), nn.Conv2d(o1, o2, [k2, k2], [s2, s2]), nn.GELU())
    return ref, copy.deepcopy(ref).cuda()


@pytest.mark.parametrize('N,C,H,W,o1,k1,s1,o2,k2,s2,tiles', [
    (4608, 3, 30, 30, 16, 8, 4, 32, 4, 2, 1),   # the cfg4 representation pass: 512 x 9 frames, `simple` preset
    (37, 3, 30, 30, 16, 8, 4, 32, 4, 2, 1),     # ragged last group
    (3, 3, 30, 30, 16, 8, 4, 32, 4, 2, 1),      # less than one group
    (50, 4, 20, 24, 12, 4, 2, 20, 4, 3, 0),     # 9x11 -> 2x3 = 6 positions, strides 2 x 3: crops not 16-byte granular (generic path)
    (41, 1, 28, 28, 16, 4, 4, 24, 4, 3, 1),     # 7x7 -> 2x2, K1 = 16, out2 not a multiple of 16
    (29, 4, 16, 16, 8, 4, 2, 32, 4, 1, 1),      # 7x7 -> 4x4 = 16 positions: one frame per group, 8 channels (K2 = 128)
    # TILED: the second-layer map has more than 16 positions
    (37, 3, 84, 84, 16, 8, 4, 32, 4, 2, -1),    # the reference environments' frames (ConvLayers(84, 84, 3, 'simple')):
                                                # 20x20 -> 9x9 in blocks of <= 16 positions (six of 3 x 5: the cheapest cover)
    (300, 3, 84, 84, 16, 8, 4, 32, 4, 2, -1),   # ... more virtual frames than workgroups
    (21, 1, 64, 64, 16, 8, 4, 32, 4, 2, -1),    # 15x15 -> 6x6: blocks that do not tile the map are moved inside, duplicates masked
    (18, 2, 52, 68, 8, 4, 4, 16, 4, 2, -1),     # 13x17 -> 5x7, non-square frame, fewer channels
    (10, 3, 48, 48, 16, 8, 4, 32, 4, 1, -1),    # 11x11 -> 8x8 with second stride 1 (s1 * s2 = 4)
    (23, 4, 30, 30, 16, 8, 4, 32, 4, 2, 1),     # four input channels: K1 = 256, sixteen quads of operand constants in registers
    (9, 4, 44, 44, 16, 8, 4, 32, 4, 2, 1),      # ... 10x10 -> 4x4 = 16 positions: one frame per group
    (9, 4, 52, 52, 16, 8, 4, 32, 4, 2, -1),     # ... 12x12 -> 5x5: tiled
    (9, 4, 84, 84, 16, 8, 4, 32, 4, 2, -1),     # four channels at 84 x 84: a row of blocks' crop exceeds the DMA pieces -> block by block
    (25, 3, 30, 30, 16, 8, 4, 32, 3, 1, 1),     # 3 x 3 second layer, stride 1: 9 col2im contributions a position, K2 = 144
    (17, 3, 32, 32, 16, 4, 4, 32, 2, 2, 1),     # 2 x 2 second layer, stride 2: patches do not overlap (one contribution)
])
def test_fused_conv_stack_matches_modules(N, C, H, W, o1, k1, s1, o2, k2, s2, tiles):
    import asac_amd  # noqa: F401
    from asac_amd import native
    from algorithm.fused_conv import conv_stack_desc, fused_conv_stack
    ref, dev = _stack(C, o1, k1, s1, o2, k2, s2)
    gen = torch.Generator().manual_seed(1)
    x = torch.randn(N, C, H, W, generator=gen)
    want = ref(x).reshape(N, -1)
    gy = torch.randn(want.shape, generator=gen)
    (want * gy).sum().backward()
    xd = x.cuda()
    desc = conv_stack_desc(dev, xd)
    h1, w1 = (H - k1) // s1 + 1, (W - k1) // s1 + 1
    h2, w2 = (h1 - k2) // s2 + 1, (w1 - k2) // s2 + 1
    assert (desc is not None) == (tiles != 0)
    if desc is None:
        return
    assert h2 * w2 > 0 and (native.conv2_tiles(desc) == tiles if tiles > 0 else native.conv2_tiles(desc) > 1)
    with native.LaunchProfiler() as prof:
        got = fused_conv_stack(xd, desc, dev)
        (got * gy.cuda()).sum().backward()
    assert prof.summary()['asac_conv2_forward']['calls'] == 1 and prof.summary()['asac_conv2_backward']['calls'] == 1
    np.testing.assert_allclose(got.detach().cpu().numpy(), want.detach().numpy(), rtol=1e-4, atol=2e-5)
    for pr, pd in zip(ref.parameters(), dev.parameters()):
        scale = float(pr.grad.abs().max())
        np.testing.assert_allclose(pd.grad.cpu().numpy(), pr.grad.numpy(), rtol=2e-4, atol=2e-5 * max(scale, 1.0))
    # inference: no saved activations, same values; deterministic
    with torch.no_grad():
        again = fused_conv_stack(xd, desc, dev)
    assert torch.equal(again, got.detach())
    dev.zero_grad()
    (fused_conv_stack(xd, desc, dev) * gy.cuda()).sum().backward()
    g1 = [p.grad.clone() for p in dev.parameters()]
    dev.zero_grad()
    (fused_conv_stack(xd, desc, dev) * gy.cuda()).sum().backward()
    assert all(torch.equal(a, b.grad) for a, b in zip(g1, dev.parameters()))


def test_conv_layers_route_to_the_fused_stack():
    """`ConvLayers(..., 'simple')` on device frames uses the fused launch (with leading batch dims) and matches
    its own generic path; other presets and inputs that need gradients keep the generic path."""
    import asac_amd  # noqa: F401
    from asac_amd import native
    import algorithm.nn_models as m
    torch.manual_seed(0)
    layer = m.ConvLayers(30, 30, 3, 'simple', out_dense_depth=2, output_size=8).cuda()
    x = torch.randn(5, 9, 3, 30, 30, device='cuda')
    with native.LaunchProfiler() as prof:
        got = layer(x)
    assert prof.summary()['asac_conv2_forward']['calls'] == 1 and got.shape == (5, 9, 8)
    want = layer.dense(layer.conv_layers(x.reshape(-1, 3, 30, 30)).reshape(5, 9, -1))
    np.testing.assert_allclose(got.detach().cpu().numpy(), want.detach().cpu().numpy(), rtol=1e-4, atol=1e-5)
    xg = x.clone().requires_grad_(True)
    with native.LaunchProfiler() as prof:
        layer(xg).sum().backward()
    assert 'asac_conv2_forward' not in prof.summary() and xg.grad is not None
    nature = m.ConvLayers(84, 84, 3, 'nature', out_dense_depth=1, output_size=8).cuda()
    with native.LaunchProfiler() as prof:
        nature(torch.randn(2, 3, 84, 84, device='cuda'))
    assert 'asac_conv2_forward' not in prof.summary()


def test_conv_stack_adds_parameter_gradients_in_place_inside_flat_buffers():
    """With the four parameters' `.grad`s consecutive views of one buffer (as inside SAC_Base), the reduction kernel
    adds into them itself: same values as the returned-gradient path, no autograd accumulation launches."""
    import asac_amd  # noqa: F401
    from algorithm.fused import FlatParamGroup
    from algorithm.fused_conv import conv_stack_desc, fused_conv_stack
    ref, dev = _stack(3, 16, 8, 4, 32, 4, 2)
    free = copy.deepcopy(dev)
    group = FlatParamGroup([('conv', list(dev.parameters()))], 'cuda')
    x = torch.randn(203, 3, 30, 30, device='cuda')
    gy = torch.randn(203, 128, device='cuda')
    desc = conv_stack_desc(dev, x)
    (fused_conv_stack(x, desc, free) * gy).sum().backward()
    group.grad.fill_(1.0)
    (fused_conv_stack(x, desc, dev) * gy).sum().backward()
    for pf, pd in zip(free.parameters(), dev.parameters()):
        np.testing.assert_allclose(pd.grad.cpu().numpy(), 1.0 + pf.grad.cpu().numpy(), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize('B,L,b', [(64, 9, 5), (7, 12, 4), (33, 8, 0)])
def test_forward_over_a_window_slice_read_in_place(B, L, b):
    """`asac_conv2_forward_windows` on frames[:, b:] of a [B, L, C, H, W] batch == `asac_conv2_forward` on its
    contiguous copy, bit for bit; through `ConvLayers` a no-grad call on the slice takes that launch (no copy)."""
    import asac_amd  # noqa: F401
    import algorithm.nn_models as m
    from asac_amd import native
    torch.manual_seed(B)
    conv = m.ConvLayers(30, 30, 3, 'simple', out_dense_depth=1, output_size=8).cuda()
    frames = torch.rand(B, L, 3, 30, 30, device='cuda')
    view = frames[:, b:]
    desc = native.conv2_desc(3, 30, 30, 16, 8, 4, 32, 4, 2)
    G = native.conv2_group_frames(desc)
    assert G == 4
    c1, _, c2, _ = list(conv.conv_layers)
    wd = [t.detach().contiguous() for t in (c1.weight, c1.bias, c2.weight, c2.bias)]
    T = L - b
    want = torch.empty(B * T, 128, device='cuda')
    native.conv2_forward(desc, view.reshape(B * T, 3, 30, 30).contiguous(), *wd, want)
    if T % G == 0 and b > 0:
        got = torch.empty(B * T, 128, device='cuda')
        native.conv2_forward_windows(desc, view, *wd, got)
        assert torch.equal(got, want)
        with native.LaunchProfiler() as prof, torch.no_grad():
            out = conv(view)
        assert prof.summary()['asac_conv2_forward_windows']['calls'] == 1 and 'asac_conv2_forward' not in prof.summary()
    else:
        with torch.no_grad():
            out = conv(view)
    with torch.no_grad():
        ref = conv(view.contiguous())
    assert out.shape == (B, T, 8) and torch.equal(out, ref)
    if T % G == 0 and b > 0:
        # ... and with gradients: the slice is read in place forward and backward, same parameter gradients as on a copy
        grads = []
        for src in (view, view.contiguous()):
            conv.zero_grad(set_to_none=True)
            with native.LaunchProfiler() as prof:
                conv(src).square().sum().backward()
            grads.append([p.grad.clone() for p in conv.parameters()])
            assert ('asac_conv2_backward_windows' in prof.summary()) == (src is view)
        assert all(torch.equal(a, c) for a, c in zip(*grads))
        with pytest.raises(native.AsacNativeError):        # a slice whose samples do not hold whole groups
            native.conv2_forward_windows(desc, frames[:, 1:4], *wd, torch.empty(B * 3, 128, device='cuda'))


@pytest.mark.parametrize('N,C,H,W,nc,windows', [
    (4608, 3, 30, 30, 3, False),    # cfg5's three gated walks over the 512 x 9 frames of the windows
    (4608, 3, 30, 30, 4, True),     # ... four cotangents, frames read in place as a slice of the windows
    (37, 3, 30, 30, 2, False),      # ragged last group
    (300, 3, 84, 84, 3, False),     # tiled mode (the reference environments' frames)
    (64, 3, 84, 84, 4, True),
    (23, 4, 30, 30, 4, False),      # K1 = 256
])
def test_conv_backward_for_several_cotangents_is_the_single_launches_bit_for_bit(N, C, H, W, nc, windows):
    """`asac_conv2_backward_multi`: nc backward walks of ONE forward pass as one launch == nc `asac_conv2_backward(_windows)`
    launches, every packed gradient bit for bit (the per-cotangent operations keep their order; only what does not depend
    on the cotangent is shared)."""
    import asac_amd  # noqa: F401
    from asac_amd import native
    torch.manual_seed(N + nc)
    dev = _stack(C, 16, 8, 4, 32, 4, 2)[1]
    c1, _, c2, _ = list(dev)
    desc = native.conv2_desc(C, H, W, 16, 8, 4, 32, 4, 2)
    assert native.conv2_supported(desc)
    wd = [t.detach().contiguous() for t in (c1.weight, c1.bias, c2.weight, c2.bias)]
    if windows:
        T = 4
        frames = torch.randn(N // T, T + 3, C, H, W, device='cuda')
        x = frames[:, 3:]
        n_frames = (N // T) * T
    else:
        x = torch.randn(N, C, H, W, device='cuda')
        n_frames = N
    h1, w1 = (H - 8) // 4 + 1, (W - 8) // 4 + 1
    out = 32 * ((h1 - 4) // 2 + 1) * ((w1 - 4) // 2 + 1)
    y = torch.empty(n_frames, out, device='cuda')
    z1 = torch.empty(native.conv2_z1_floats(desc, n_frames), device='cuda')
    z2 = torch.empty_like(y)
    (native.conv2_forward_windows if windows else native.conv2_forward)(desc, x, *wd, y, z1, z2)
    gys = [torch.randn_like(y) * (0.3 + c) for c in range(nc)]
    n = native.conv2_param_count(desc)
    ws1 = torch.empty(native.conv2_backward_workspace(desc, n_frames), device='cuda')
    want = torch.zeros(nc, n, device='cuda')
    single = native.conv2_backward_windows if windows else native.conv2_backward
    for c in range(nc):
        single(desc, x, wd[2], z1, z2, gys[c], want[c], ws1)
    got = torch.full((nc, n), 7.0, device='cuda')
    ws = torch.empty(nc * ws1.numel(), device='cuda')
    with native.LaunchProfiler() as prof:
        native.conv2_backward_multi(desc, x, wd[2], z1, z2, gys, got, ws)
    assert list(prof.summary()) == ['asac_conv2_backward_multi']
    # a workgroup's arithmetic per cotangent is the single launch's; where both forms run the same number of workgroups the
    # final sums see the same slabs in the same order: bit for bit.  (A single launch that fits two workgroups per CU — one
    # frame buffer, 30 x 30 — has twice the slabs of the multi form: the same numbers summed in another grouping.)
    per_launch = min(nc, native.conv2_backward_multi_max(desc))
    same_slabs = native.conv2_backward_slabs(desc, n_frames, 1) == native.conv2_backward_slabs(desc, n_frames, per_launch)

    def check(a, b):
        if same_slabs:
            assert torch.equal(a, b)
        else:
            np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=2e-5, atol=2e-6 * float(b.abs().max()))
    check(got, want)
    assert float(want.abs().sum()) > 0
    # three cotangents of 30 x 30 frames are ONE kernel launch (the fourth set of LDS buffers does not fit beside the
    # double-buffered frames: 3 + 1)
    if (C, H, W) == (3, 30, 30):
        assert native.conv2_backward_multi_max(desc) == 3
    # accumulate form
    before = got.clone()
    native.conv2_backward_multi(desc, x, wd[2], z1, z2, gys, got, ws, accumulate=True)
    assert torch.equal(got, before + before)


def test_deferred_conv_backward_equals_the_walks_own_launches():
    """`fused_conv.DeferredConvBackward`: three `autograd.grad` walks of one forward pass with the convolution stack's
    backward recorded and run as ONE launch afterwards — the gradients each walk would have returned, bit for bit; a walk
    outside the context launches as always."""
    import asac_amd  # noqa: F401
    from asac_amd import native
    from algorithm.fused_conv import DeferredConvBackward, conv_stack_desc, fused_conv_stack
    dev = _stack(3, 16, 8, 4, 32, 4, 2)[1]
    head = nn.Linear(128, 8).cuda()
    x = torch.randn(400, 3, 30, 30, device='cuda')
    desc = conv_stack_desc(dev, x)
    params = [*dev.parameters(), *head.parameters()]
    y = head(fused_conv_stack(x, desc, dev))
    cots = [torch.randn_like(y) for _ in range(3)]
    want = [torch.autograd.grad(y, params, grad_outputs=c, retain_graph=True) for c in cots]
    later = DeferredConvBackward()
    got = []
    with native.LaunchProfiler() as prof:
        for k, c in enumerate(cots):
            later.walk = k
            with later:
                got.append(list(torch.autograd.grad(y, params, grad_outputs=c, retain_graph=True, allow_unused=True)))
        conv_grads = later.flush()
    assert 'asac_conv2_backward' not in prof.summary() and prof.summary()['asac_conv2_backward_multi']['calls'] == 1
    for k in range(3):
        for j, p in enumerate(params):
            g = conv_grads[k].get(id(p)) if got[k][j] is None else got[k][j]
            assert g is not None and torch.equal(g, want[k][j]), (k, j)
        assert all(got[k][j] is None for j in range(4)) and set(conv_grads[k]) == {id(p) for p in dev.parameters()}
    assert DeferredConvBackward._active is None
