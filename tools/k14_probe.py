"""K1-K3 inside a graph, cold: the step prologue's sampler (B = 256 / 512 / 1024, C = 2^16) and the window gather at cfg4 /
cfg5 shape with a DIFFERENT id set per launch (the kernel sweep re-issues one id set back to back: its 50 - 100 MB of rows
stay in the 256 MB Infinity Cache, which a train step's fresh draw never sees).  N dependent launches captured in one
hipGraph, replayed; per-launch time = replay time / N.

    python tools/k14_probe.py [--ring 65536]
"""
import argparse
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import asac_amd  # noqa: E402,F401
from asac_amd import native  # noqa: E402

dev = torch.device('cuda')


def graph_time(fn_list, replays=10):
    """fn_list: launches captured in order into one graph -> microseconds per launch (mean over replays)"""
    for f in fn_list[:2]:
        f()
    torch.cuda.synchronize()
    stream = torch.cuda.Stream()
    stream.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(stream):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            for f in fn_list:
                f()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(replays):
            g.replay()
        e1.record(stream)
        torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / replays / len(fn_list)


def gather_probe(C, B, L=9, n_sets=24):
    img = torch.randn(C, 3, 30, 30, device=dev)
    vec = torch.randn(C, 10, device=dev)
    index = (torch.arange(C, device=dev, dtype=torch.int32) % 100)
    out_img = torch.empty(B, L, 3, 30, 30, device=dev)
    out_vec = torch.empty(B, L, 10, device=dev)
    out_idx = torch.empty(B, L, dtype=torch.int32, device=dev)
    mask = torch.empty(B, L, dtype=torch.bool, device=dev)
    keys = native.make_gather_keys([
        dict(src=img, dst=out_img, row_bytes=10800, pad_mode=native.PAD_KEEP),
        dict(src=vec, dst=out_vec, row_bytes=40, pad_mode=native.PAD_KEEP),
        dict(src=index, dst=out_idx, row_bytes=4, pad_mode=native.PAD_WORD, pad_word=0xffffffff),
        dict(src=None, dst=mask, pad_mode=native.PAD_EMIT_MASK)])
    T = 10800 + 40 + 4
    by = 8 * B + 2 * B * L * T
    sets = [torch.randint(10, C - 10, (B,), device=dev, dtype=torch.int64) for _ in range(n_sets)]
    warm = graph_time([lambda: native.window_gather_pad(keys, sets[0], B, 5, 3, C, index)] * n_sets)
    cold = graph_time([(lambda s=s: native.window_gather_pad(keys, s, B, 5, 3, C, index)) for s in sets])
    for name, us in (('same ids every launch', warm), ('fresh ids every launch', cold)):
        print(f'window_gather_pad B={B} L={L} ring {C} rows ({C * T / 1e6:.0f} MB), {name:24s} {by / 1e6:8.2f} MB '
              f'{us:8.2f} us {by / us / 1e3:8.1f} GB/s = {by / us / 1e3 / 80:5.1f} % of 8 TB/s')


def sampler_probe(C, B, n=32):
    tree = torch.zeros(2 * C - 1, device=dev)
    winner = torch.full((C + 2 * C,), -1, dtype=torch.int32, device=dev)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    for s in range(0, C, 4096):
        native.sumtree_update(tree, C, torch.arange(s, s + 4096, device=dev), None, torch.rand(4096, device=dev) + 0.01,
                              0.9, 0.01, 1.0, 1, winner, flag)
    slot_ids = torch.arange(C, device=dev, dtype=torch.int64)
    beta = torch.tensor([0.4], dtype=torch.float64, device=dev)
    minp = torch.zeros(528, device=dev)
    u = torch.zeros(B, dtype=torch.float64, device=dev)
    z = torch.zeros(20000, device=dev)
    leaf = torch.empty(B, dtype=torch.int32, device=dev)
    p = torch.empty(B, device=dev)
    ids = torch.empty(B, dtype=torch.int64, device=dev)
    w = torch.empty(B, device=dev)
    step = torch.zeros(1, dtype=torch.int64, device=dev)
    target, source, grad = torch.zeros(60000, device=dev), torch.ones(60000, device=dev), torch.ones(60000, device=dev)
    us = graph_time([lambda: native.step_prologue_sample((target, source, 0.005), grad, 7, step, u, z, None, 0, tree, C, B,
                                                         slot_ids, beta, 0.001, leaf, p, ids, w, minp)] * n)
    print(f'step_prologue_sample C={C} B={B} (Polyak 60k, zero 60k, 20k draws): {us:6.2f} us per launch')


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--ring', type=int, default=65536)
    a = ap.parse_args()
    native.load()
    print(f'# {torch.cuda.get_device_name(0)}')
    for B in (256, 512, 1024):
        sampler_probe(a.ring, B)
    sampler_probe(2 ** 19, 256)
    for B in (512, 1024):
        gather_probe(a.ring, B)
