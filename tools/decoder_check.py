"""Stage-by-stage check of the fused observation decoder (csrc/decoder.hip) against the PyTorch modules on the same
device: every saved activation (unpacked from the kernels' tile layout), the frames, and every gradient.  Prints the
largest absolute / relative error per stage, so that one GPU run localises an indexing mistake.
    python tools/decoder_check.py [N] [S]
"""
import sys
from pathlib import Path

import numpy as np
import torch
from torch import nn

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import asac_amd  # noqa: F401,E402
from asac_amd import native  # noqa: E402
import algorithm.nn_models as m  # noqa: E402
from algorithm.fused_decoder import decoder_params  # noqa: E402


def tiles(flat, G, n):
    """[G * n * 256] -> [G, n, 16 channels, 16 states]"""
    a = flat.reshape(G, n, 4, 16, 4)          # q, x, r
    return a.permute(0, 1, 2, 4, 3).reshape(G, n, 16, 16)


def report(name, got, want):
    got, want = got.double().cpu(), want.double().cpu()
    err = (got - want).abs()
    scale = want.abs().max().item()
    print(f'{name:12s} max|err| {err.max().item():.3e}   max|ref| {scale:.3e}   rel {err.max().item() / max(scale, 1e-30):.3e}'
          f'   shape {tuple(want.shape)}')
    return err.max().item() / max(scale, 1e-30)


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    torch.manual_seed(0)
    dev = torch.device('cuda')
    ctl = m.ConvTransposeLayers(S, 64, 1, 2, 2, 32, conv_transpose=nn.Sequential(
        nn.ConvTranspose2d(32, 32, 4, 2), nn.LeakyReLU(), nn.ConvTranspose2d(32, 16, 8, 4), nn.LeakyReLU(),
        nn.ConvTranspose2d(16, 3, 3, 1), nn.LeakyReLU())).to(dev)
    for p in ctl.parameters():       # biases are initialised to zero: make them matter
        if p.dim() == 1:
            p.data.normal_(0, 0.1)
    params = decoder_params(ctl)
    assert params is not None
    x = torch.randn(N, S, device=dev, requires_grad=True)
    G = (N + 15) // 16

    # reference with intermediates
    lin1, lin2 = ctl.dense.dense[0].linear, ctl.dense.dense[2]
    c1, _, c2, _, c3, _ = list(ctl.conv_transpose)
    z1 = lin1(x)
    h1 = nn.functional.gelu(z1)
    h0 = lin2(h1)
    a1 = nn.functional.leaky_relu(c1(h0.reshape(N, 32, 2, 2)))
    a2 = nn.functional.leaky_relu(c2(a1))
    out = nn.functional.leaky_relu(c3(a2))
    gout = torch.randn_like(out)
    for t in (z1, h0, a1, a2):
        t.retain_grad()
    (out * gout).sum().backward()
    ref_grads = [p.grad.clone() for p in params]
    ref_gx = x.grad.clone()

    frames = torch.empty(N, 3, 30, 30, device=dev)
    packed = torch.empty(native.obs_decoder_packed_floats(), device=dev)
    saved = torch.full((native.obs_decoder_saved_floats(N),), float('nan'), device=dev)
    native.obs_decoder_forward(x.detach(), [p.detach() for p in params], packed, saved, frames)
    torch.cuda.synchronize()
    o = 0
    z1_t = tiles(saved[o:o + G * 4 * 256], G, 4); o += G * 4 * 256
    h0_t = tiles(saved[o:o + G * 8 * 256], G, 8); o += G * 8 * 256
    a1_t = tiles(saved[o:o + G * 72 * 256], G, 72); o += G * 72 * 256
    a2_t = tiles(saved[o:o + G * 784 * 256], G, 784)

    def pad(t):      # [N, ...] -> [G, 16, ...]
        full = torch.zeros(G * 16, *t.shape[1:], device=dev)
        full[:N] = t.detach()
        return full.reshape(G, 16, *t.shape[1:])
    valid = torch.zeros(G * 16, dtype=torch.bool, device=dev)
    valid[:N] = True
    valid = valid.reshape(G, 16)

    def masked(t_tiles_state_last):   # zero invalid states (the kernels compute a clamped copy there)
        return t_tiles_state_last * valid.reshape(G, *([1] * (t_tiles_state_last.dim() - 2)), 16)
    # z1: tiles [G, 4, 16 ch, 16 st] -> [G, 64, 16]
    report('z1', masked(z1_t.reshape(G, 64, 16)), pad(z1).permute(0, 2, 1))
    # h0: tile (p, t): channel ic = 16 t + c; feature f = ic * 4 + p
    h0_got = h0_t.reshape(G, 4, 2, 16, 16).permute(0, 2, 3, 1, 4).reshape(G, 32, 4, 16)     # [G, ic, p, st]
    report('h0', masked(h0_got), pad(h0).reshape(G, 16, 32, 4).permute(0, 2, 3, 1))
    a1_got = a1_t.reshape(G, 36, 2, 16, 16).permute(0, 2, 3, 1, 4).reshape(G, 32, 36, 16)   # [G, oc, pix, st]
    report('act1', masked(a1_got), pad(a1).reshape(G, 16, 32, 36).permute(0, 2, 3, 1))
    a2_got = a2_t.reshape(G, 784, 16, 16).permute(0, 2, 1, 3)                               # [G, oc, pix, st]
    report('act2', masked(a2_got), pad(a2).reshape(G, 16, 16, 784).permute(0, 2, 3, 1))
    worst = report('frames', frames, out)

    ws = torch.full((native.obs_decoder_workspace_floats(N),), float('nan'), device=dev)
    gx = torch.empty(N, S, device=dev)
    grads = [torch.full_like(p, float('nan')) for p in params]
    native.obs_decoder_backward(x.detach(), packed, saved, frames, gout, gx, grads, ws)
    torch.cuda.synchronize()
    dz2_t = tiles(ws[:G * 784 * 256], G, 784).reshape(G, 784, 16, 16).permute(0, 2, 1, 3)
    dz2_ref = a2.grad * torch.where(a2 > 0, 1.0, 0.01)
    # LeakyReLU has a kink: an activation at rounding level may land on either side of zero in the two implementations,
    # and its derivative then differs by a factor of 100 — count those and compare the gradient in front of the kink
    a2_mine = pad(a2).reshape(G, 16, 16, 784).permute(0, 2, 3, 1)
    flips = ((a2_got > 0) != (a2_mine > 0)) & valid.reshape(G, 1, 1, 16)
    print(f'act2 sign flips against the reference: {int(flips.sum())} of {flips.numel()}')
    dact2_got = dz2_t / torch.where(a2_got > 0, 1.0, 0.01)
    report('d act2', masked(dact2_got), pad(a2.grad).reshape(G, 16, 16, 784).permute(0, 2, 3, 1))
    report('dz2', masked(dz2_t), pad(dz2_ref).reshape(G, 16, 16, 784).permute(0, 2, 3, 1))
    names = ['dense1.w', 'dense1.b', 'dense2.w', 'dense2.b', 'ct1.w', 'ct1.b', 'ct2.w', 'ct2.b', 'ct3.w', 'ct3.b']
    for n, g, r in zip(names, grads, ref_grads):
        worst = max(worst, report('d ' + n, g, r))
    worst = max(worst, report('d state', gx, ref_gx))
    # accumulate mode adds
    grads2 = [g.clone() for g in grads]
    native.obs_decoder_backward(x.detach(), packed, saved, frames, gout, gx, grads2, ws, accumulate=True)
    torch.cuda.synchronize()
    print('accumulate: ', max(((g2 - 2 * g).abs().max().item()) for g, g2 in zip(grads, grads2)))
    print('WORST', worst)
    if '--time' in sys.argv:
        for fn, label in ((lambda: native.obs_decoder_forward(x.detach(), [p.detach() for p in params], packed, saved, frames), 'forward'),
                          (lambda: native.obs_decoder_backward(x.detach(), packed, saved, frames, gout, gx, grads, ws), 'backward')):
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 100
            fl = (2 if label == 'forward' else 4) * N * native.OBS_DECODER_MACS
            print(f'{label}: {us:.1f} us   {fl / us / 1e6:.1f} TFLOP/s')


if __name__ == '__main__':
    main()
