"""Saturating-size sweep of the HBM-bound hot-path kernels: achieved GB/s vs the 8 TB/s HBM3E peak
(6.3 TB/s measured copy ceiling, MI355X_MICROARCH.md).  At the BASELINE batch (256) these launches move
<= 0.5 MB and are latency-bound; this sweep shows what the same kernels reach when one launch has
enough work.  Timing: HIP events around `repeat` back-to-back launches (asac_set_launch_repeat).

    python tools/kernel_sweep.py > profiles/r01_kernel_sweep.txt
"""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import asac_amd  # noqa: E402,F401
from asac_amd import native  # noqa: E402

PEAK = 8000.0
dev = torch.device('cuda')


RECORDS = []        # what `timed` measured, for callers that want the numbers (bench.py's `sweep` section)
QUIET = False
ONLY_SATURATING = False     # bench.py: only the sizes where the HBM roofline is meaningful


def timed(name, fn, bytes_per_launch, repeat=20, rounds=5):
    for _ in range(2):
        fn()
    with native.LaunchProfiler(repeat=repeat) as prof:
        for _ in range(rounds):
            fn()
    s = prof.summary()
    # mean over the rounds (each a HIP-event bracket around `repeat` back-to-back launches)
    us = sum(v['avg_us'] for v in s.values())
    gbs = bytes_per_launch / (us * 1e-6) / 1e9
    RECORDS.append({'kernel': name, 'alg_bytes_per_launch': int(bytes_per_launch), 'avg_launch_us': round(us, 3),
                    'achieved_GBs': round(gbs, 1), 'frac_of_hbm_peak': round(gbs / PEAK, 4)})
    if not QUIET:
        print(f'{name:58s} {bytes_per_launch / 1e6:10.2f} MB {us:10.2f} us {gbs:9.1f} GB/s  {100 * gbs / PEAK:5.1f}% of HBM peak')


def sweep_gather():
    # cfg4 shape: vector(10) + image(3,30,30) f32 rows (10.9 KB), L = 9 (b=5, n=3), B = 512
    # (ring of 2^16 rows = 711 MB: larger than the Infinity Cache)
    for B, L, C in (((1024, 9, 2 ** 16), (4096, 9, 2 ** 16)) if ONLY_SATURATING else
                    ((512, 9, 2 ** 16), (1024, 9, 2 ** 16), (4096, 9, 2 ** 16))):
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
        ids = torch.randint(10, C - 10, (B,), device=dev, dtype=torch.int64)
        T = 10800 + 40 + 4
        # WARM: the same ids re-issued back to back — a launch's 50-400 MB of rows are still in the 256 MB Infinity Cache when
        # the next one reads them (what rounds 1-5 quoted); COLD: fresh ids every launch, what a train step's draw sees
        timed(f'window_gather_pad cfg4 rows B={B} L={L}  WARM (same ids)',
              lambda: native.window_gather_pad(keys, ids, B, 5, 3, C, index), 8 * B + 2 * B * L * T)
        pool = [torch.randint(10, C - 10, (B,), device=dev, dtype=torch.int64) for _ in range(16)]
        turn = [0]

        def cold():
            turn[0] = (turn[0] + 1) % len(pool)
            native.window_gather_pad(keys, pool[turn[0]], B, 5, 3, C, index)
        timed(f'window_gather_pad cfg4 rows B={B} L={L}  COLD (fresh ids)', cold, 8 * B + 2 * B * L * T, repeat=1, rounds=12)


def sweep_sample():
    C = 2 ** 19
    tree = torch.zeros(2 * C - 1, device=dev)
    winner = torch.full((C + 2 * C,), -1, dtype=torch.int32, device=dev)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    native.sumtree_update(tree, C, torch.arange(C, device=dev), None, torch.rand(C, device=dev) + 0.01,
                          0.9, 0.01, 1.0, 1, winner, flag)
    slot_ids = torch.arange(C, device=dev, dtype=torch.int64)
    beta = torch.tensor([0.4], dtype=torch.float64, device=dev)
    minp = torch.zeros(2, device=dev)
    for B in ((2 ** 18, 2 ** 20) if ONLY_SATURATING else (256, 2 ** 14, 2 ** 18, 2 ** 20)):
        u = torch.rand(B, dtype=torch.float64, device=dev)
        leaf = torch.empty(B, dtype=torch.int32, device=dev)
        p = torch.empty(B, device=dev)
        ids = torch.empty(B, dtype=torch.int64, device=dev)
        w = torch.empty(B, device=dev)
        timed(f'sumtree_sample (+ids +IS weights) C=2^19 B={B}',
              lambda: native.sumtree_sample(tree, C, B, u, slot_ids, beta, 0.0, leaf, p, ids, w, minp),
              B * (8 + 8 * 19 + 8) + 8 * B)
    out = torch.zeros(1, device=dev)
    for Cx in ((2 ** 26,) if ONLY_SATURATING else (2 ** 19, 2 ** 24, 2 ** 26)):
        t2 = torch.rand(2 * Cx - 1, device=dev)
        timed(f'sumtree_leaf_max C={Cx}', lambda: native.sumtree_leaf_max(t2, Cx, out), 4 * Cx)


def sweep_return():
    E, A = 2, 2
    for B, n in (((2 ** 20, 4), (2 ** 16, 40)) if ONLY_SATURATING else ((256, 4), (2 ** 16, 4), (2 ** 20, 4), (2 ** 16, 40))):
        q = torch.randn(E, B, n + 1, device=dev)
        logp = torch.randn(B, n + 1, device=dev)
        la = torch.tensor([-2.3], device=dev)
        r = torch.randn(B, n, device=dev)
        done = torch.rand(B, n, device=dev) < 0.3
        last = torch.rand(B, n, device=dev) < 0.1
        pad = torch.rand(B, n, device=dev) < 0.1
        mu = torch.rand(B, n, A, device=dev) + 0.1
        pi = torch.rand(B, n + 1, A, device=dev) + 0.1
        gr = torch.logspace(0, n - 1, n, 0.99, device=dev)
        lr = torch.ones(n, device=dev)
        y = torch.empty(B, device=dev)
        a = native.VtraceArgs()
        a.q, a.q_stride_e, a.q_stride_b, a.q_stride_t = q.data_ptr(), q.stride(0), q.stride(1), q.stride(2)
        a.E_sample, a.logp, a.log_alpha = E, logp.data_ptr(), la.data_ptr()
        a.reward, a.reward_stride = r.data_ptr(), r.stride(0)
        a.done, a.last_mask, a.padding_mask, a.mask_stride = done.data_ptr(), last.data_ptr(), pad.data_ptr(), n
        a.mu_prob, a.mu_stride_b, a.mu_stride_t, a.mu_offset = mu.data_ptr(), mu.stride(0), mu.stride(1), 0
        a.pi_prob, a.pi_stride_b, a.pi_stride_t, a.A = pi.data_ptr(), pi.stride(0), pi.stride(1), A
        a.gamma_ratio, a.lambda_ratio = gr.data_ptr(), lr.data_ptr()
        a.gamma, a.v_rho, a.v_c, a.use_n_step_is, a.B, a.n = 0.99, 1.0, 1.0, 1, B, n
        a.y_out = y.data_ptr()
        by = B * (n * (4 + 1 + 1 + 1 + 4 * A + 4 * A) + E * (n + 1) * 4 + (n + 1) * 4 + 4)
        timed(f'vtrace_return_min E={E} A={A} n={n} B={B}', lambda: native.vtrace_return_min(a), by)
        loc, scale, eps = torch.randn(B, n + 1, A, device=dev), torch.rand(B, n + 1, A, device=dev) + 0.1, torch.randn(B, n + 1, A, device=dev)
        at, lp = torch.empty_like(loc), torch.empty(B, n + 1, device=dev)
        timed(f'squash_sample_fwd A={A} rows={B * (n + 1)}', lambda: native.squash_sample_fwd(loc, scale, eps, at, lp),
              B * (n + 1) * (16 * A + 4))


def sweep_params():
    for P in (27_000, 2 ** 20, 2 ** 26):
        t, s = torch.zeros(P, device=dev), torch.ones(P, device=dev)
        timed(f'polyak P={P}', lambda: native.polyak(t, s, 0.005), 12 * P)
        g, m, v = torch.randn(P, device=dev), torch.zeros(P, device=dev), torch.zeros(P, device=dev)
        steps = torch.zeros(1, dtype=torch.int64, device=dev)
        timed(f'adam_step P={P}', lambda: native.adam_step(t, g, m, v, 3e-4, 0.9, 0.999, 1e-8, steps), 28 * P)


def saturating_sweep() -> list:
    """K1-K4 (+K8) at sizes where one launch has enough work for the HBM roofline to mean something -> records"""
    global QUIET, ONLY_SATURATING
    QUIET, ONLY_SATURATING = True, True
    RECORDS.clear()
    native.load()
    sweep_gather()
    sweep_sample()
    sweep_return()
    torch.cuda.empty_cache()
    return list(RECORDS)


if __name__ == '__main__':
    native.load()
    print(f'# kernel sweep on {torch.cuda.get_device_name(0)}; peak {PEAK:.0f} GB/s (spec), algorithmic bytes per SURVEY.md §8d')
    sweep_gather()
    sweep_sample()
    sweep_return()
    sweep_params()
