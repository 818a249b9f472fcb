"""GPU, ONE device: the world-2 PRODUCT learner.  Two processes share `cuda:0` and talk over gloo (CUDA tensors:
all-reduce SUM / MIN, broadcast, all-gather, all-to-all), `use_graph=False` — everything rank-dependent in `SAC_Base`
(B / G rows per rank, shards of unequal fill meeting in `all_ready`, per-rank draws, the weight broadcast, the all-to-all
of parity mode) runs here without a second GPU.  north_star: "the replay buffer shards and gradients partition across
the GPUs of one node"; global normalisation of the IS weights: reference `replay_buffer.py:352-354`.

* throughput mode (every rank stratifies its own shard): each rank's IS weights equal the reference formula with the
  minimum ratio over BOTH shards; every reduced gradient segment equals the mean of the two ranks' own gradients bit for
  bit; the replicas stay bit-identical; a rank whose shard is short keeps BOTH ranks from stepping.
* parity mode (one stratified sample over the union): per rank, ids / windows / IS weights equal — bit for bit — the
  slice of ONE product buffer holding the union of the shards (`[shard 0 | shard 1]` as leaves of one tree), and the
  step's gradients, priorities and row write-backs equal those of ONE product learner training the whole global batch on
  that union buffer (float tolerance: mean of two half-batch means against a full-batch mean)."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests import parity_utils as pu  # noqa: E402

WORLD = 2
PER_RANK, N_STEP, BURN_IN, A = 16, 3, 2, 2
SHARD_C = 256
HIDDEN = (2, 8)


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _learner(ctx, sampling, batch, capacity, seed):
    import asac_amd  # noqa: F401
    from algorithm.sac_base import SAC_Base
    from algorithm.utils.enums import SEQ_ENCODER
    torch.manual_seed(seed)
    return SAC_Base(['vector'], [(6,)], [], A, None, pu.plugin('nn_rnn'), device='cuda:0', batch_size=batch, n_step=N_STEP,
                    burn_in_step=BURN_IN, seq_encoder=SEQ_ENCODER.RNN, replay_config={'capacity': capacity}, seed=seed,
                    hip_config={'use_graph': False, 'dist': ctx, 'dist_sampling': sampling})


def _gather(t: torch.Tensor) -> list:
    """the tensor of every rank (same shape everywhere); bool travels as bytes (gloo has no bool)"""
    import torch.distributed as dist
    src = t.contiguous()
    wire = src.view(torch.uint8) if src.dtype == torch.bool else src
    out = [torch.empty_like(wire) for _ in range(WORLD)]
    dist.all_gather(out, wire)
    return [o.view(torch.bool) if src.dtype == torch.bool else o for o in out]


def _episode(rng, T):
    return pu.synthetic_episode(rng, [(6,)], [], A, HIDDEN, T)


def _set_priorities(rb, rng):
    """distinct priorities on every stored row (fresh rows all carry the maximum: a flat tree samples by position)"""
    n = rb.size
    ids = torch.arange(n, dtype=torch.int64, device='cuda')
    td = torch.from_numpy((rng.random(n) * 0.9 + 0.05).astype(np.float32)).cuda()
    keep = rb._tree[rb.capacity - 1:rb.capacity - 1 + n] > 0          # (an episode's last row keeps priority 0)
    rb._update_ids(ids[keep], td[keep], stale_check=True)


# ------------------------------------------------------------------------------------------------------------------
# throughput mode
# ------------------------------------------------------------------------------------------------------------------
def _throughput(rank):
    import torch.distributed as dist
    from algorithm.parallel import DataParallelContext

    class Recording(DataParallelContext):
        """the product's context, keeping what went into and came out of every collective of a step"""

        def __init__(self):
            super().__init__()
            self.grads, self.mins = [], []

        def all_reduce_grads(self, flat_grad, start, stop):
            before = flat_grad[start:stop].clone()
            super().all_reduce_grads(flat_grad, start, stop)
            self.grads.append((int(start), int(stop), before, flat_grad[start:stop].clone()))

        def all_reduce_min_(self, scalar):
            before = scalar.clone()
            super().all_reduce_min_(scalar)
            self.mins.append((before, scalar.clone()))

    ctx = Recording()
    assert ctx.world_size == WORLD and ctx.rank == rank and ctx._live and not ctx._avg
    agent = _learner(ctx, 'throughput', PER_RANK, SHARD_C, seed=3 + rank)      # per-rank initialisation and draws
    rb = agent.replay_buffer
    assert rb.min_ratio_reducer is not None and rb.sharded is None
    flats = _gather(agent._params.flat)
    assert torch.equal(flats[0], flats[1]), 'rank 0\'s weights reach every replica (sac_base.py broadcast)'
    assert float(flats[0].abs().sum()) > 0

    # unequal shards: rank 1 starts with ONE short episode (fewer rows than a batch) -> no rank may step
    rng = np.random.default_rng(100 + rank)
    for T in ((60, 45, 70) if rank == 0 else (12,)):
        agent.put_episode(**_episode(rng, T))
    step0 = agent.get_global_step()
    assert agent.train() == step0 and agent.get_global_step() == step0, 'a short shard anywhere holds every rank'
    assert not ctx.grads and not ctx.mins, 'no collective of the step was issued'
    if rank == 1:
        agent.put_episode(**_episode(rng, 33))
        agent.put_episode(**_episode(rng, 21))
    else:
        agent.put_episode(**_episode(rng, 80))
    _set_priorities(rb, rng)
    sizes = _gather(torch.tensor([rb.size], device='cuda'))
    assert int(sizes[0]) != int(sizes[1])

    for step in range(5):
        ctx.grads.clear()
        ctx.mins.clear()
        total = rb._tree[0:1].clone()                # the root the sample sees
        agent.train()
        torch.cuda.synchronize()
        assert agent.get_global_step() == step0 + step + 1
        # -- IS weights: the reference's formula with the minimum ratio over BOTH shards (replay_buffer.py:352-354)
        assert len(ctx.mins) == 1
        local_min, global_min = (float(x) for x in ctx.mins[0])
        p = rb._p.cpu().numpy()
        tot = np.float32(total.item())
        assert p.min() > 0
        np.testing.assert_allclose(local_min, np.float32(p.min()) / tot, rtol=1.2e-7, err_msg='this shard\'s min ratio')
        mins = [float(x) for x in _gather(ctx.mins[0][0])]
        assert global_min == min(mins), 'MIN over the ranks, exact'
        ratio = p / tot
        w_ref = np.power(ratio / np.float32(global_min), -np.float64(rb.beta)).astype(np.float32)
        np.testing.assert_allclose(rb._w.cpu().numpy(), w_ref, rtol=2e-6, atol=0, err_msg=f'step {step}: IS weights')
        has_one = [bool((x == 1).any()) for x in _gather(rb._w)]
        assert has_one == [m == global_min for m in mins], 'the globally rarest row has weight 1, on its shard only'
        ids = _gather(rb._ids)
        assert not torch.equal(ids[0], ids[1]), 'per-rank draws from per-rank shards'
        # -- every reduced segment = mean of the two ranks' own gradients, bit for bit
        assert len(ctx.grads) >= 3, 'representation + critics, policy, temperature (+ the averaged log-probabilities)'
        spans = _gather(torch.tensor([[s, e] for s, e, _, _ in ctx.grads], device='cuda'))
        assert torch.equal(spans[0], spans[1]), 'the same collectives in the same order on every rank'
        differ = 0
        for start, stop, before, after in ctx.grads:
            b = _gather(before)
            assert torch.equal(after, (b[0] + b[1]) * 0.5), f'step {step}: segment [{start}, {stop})'
            assert torch.isfinite(after).all()
            differ += int(not torch.equal(b[0], b[1]))
        assert differ == len(ctx.grads), 'the two ranks trained on different rows'
        # -- replicas identical after the step
        for name, t in (('weights', agent._params.flat), ('targets', agent._target_params.flat),
                        ('adam m', agent._exp_avg), ('adam v', agent._exp_avg_sq)):
            both = _gather(t)
            assert torch.equal(both[0], both[1]), f'step {step}: {name} diverged'
        trees = _gather(rb._tree)
        assert not torch.equal(trees[0], trees[1]), 'priorities stay per shard'
    rb.check_health()
    assert rb.check_tree_invariant() == 0
    agent.close()
    dist.barrier()


# ------------------------------------------------------------------------------------------------------------------
# parity mode
# ------------------------------------------------------------------------------------------------------------------
def _union_tree(trees, Cs):
    """array heap over 2 Cs leaves whose two subtrees are the shard heaps (root = left + right in f32)"""
    out = np.zeros(4 * Cs - 1, np.float32)
    out[0] = np.float32(trees[0][0]) + np.float32(trees[1][0])
    j = 0
    while (1 << j) <= Cs:
        lo, n = (1 << j) - 1, 1 << j
        base = (1 << (j + 1)) - 1
        out[base:base + n] = trees[0][lo:lo + n]
        out[base + n:base + 2 * n] = trees[1][lo:lo + n]
        j += 1
    return out


def _assemble_union(union, agent):
    """ONE product learner / buffer holding what the two ranks hold together: replay rows `[shard 0 | shard 1]`, the
    union tree, ids shifted by the shard's base, and this replica's weights / targets / optimizer state"""
    rb, urb = agent.replay_buffer, union.replay_buffer
    Cs = rb.capacity
    assert urb.capacity == WORLD * Cs and rb._tree.numel() == 2 * Cs - 1
    trees = [t.cpu().numpy() for t in _gather(rb._tree)]
    urb._tree.copy_(torch.from_numpy(_union_tree(trees, Cs)))
    for k, col in rb._columns.items():
        urb._columns[k].copy_(torch.cat(_gather(col)))
    sizes = [int(x) for x in _gather(torch.tensor([rb.size], device='cuda'))]
    slot_ids = _gather(rb._slot_ids)
    for g in range(WORLD):
        shifted = slot_ids[g].clone()
        shifted[:sizes[g]] += g * Cs
        urb._slot_ids[g * Cs:(g + 1) * Cs].copy_(shifted)
    urb._size, urb._next_id = Cs + sizes[1], Cs + sizes[1]
    urb._beta.copy_(rb._beta)
    union._params.flat.copy_(agent._params.flat)
    union._target_params.flat.copy_(agent._target_params.flat)
    union._exp_avg.copy_(agent._exp_avg)
    union._exp_avg_sq.copy_(agent._exp_avg_sq)
    union._opt_steps.copy_(agent._opt_steps)
    return sizes


def _parity(rank):
    import torch.distributed as dist
    from algorithm.fused import RecordedNoise
    from algorithm.parallel import DataParallelContext
    ctx = DataParallelContext()
    B = WORLD * PER_RANK
    agent = _learner(ctx, 'parity', PER_RANK, SHARD_C, seed=13 + rank)
    union = _learner(None, 'throughput', B, WORLD * SHARD_C, seed=99)
    rb, urb = agent.replay_buffer, union.replay_buffer
    assert rb.sharded is not None and rb.sharded.B == B and rb.sharded.per == PER_RANK

    # rank 1 holds nothing yet (it does not even know the transition layout): nobody steps
    rng = np.random.default_rng(200 + rank)
    if rank == 0:
        for T in (60, 45, 70):
            agent.put_episode(**_episode(rng, T))
    step0 = agent.get_global_step()
    assert agent.train() == step0, 'a rank without rows holds every rank'
    # unequal fill, and rank 1's shard alone is SHORTER than the global batch: the union decides
    for T in ((31,) if rank == 0 else (25,)):
        agent.put_episode(**_episode(rng, T))
    _set_priorities(rb, rng)
    union.put_episode(**_episode(np.random.default_rng(0), 40))       # (allocates the union's columns; overwritten below)
    lr = 3e-4

    draw_rng = np.random.default_rng(7)          # the SAME draws on both ranks: u of the global batch, noise per row
    for step in range(4):
        sizes = _assemble_union(union, agent)
        assert sizes[1] < B < sizes[0]
        u, eps, perm = pu.host_draws(draw_rng, B, N_STEP, A, 2)
        lo, hi = rank * PER_RANK, (rank + 1) * PER_RANK
        agent.noise = RecordedNoise(list(u), [e[lo:hi].copy() for e in eps], list(perm))
        union.noise = RecordedNoise(list(u), [e.copy() for e in eps], list(perm))
        rb.uniform_source, urb.uniform_source = agent.noise, union.noise
        m0 = agent._exp_avg.clone()
        agent.train()
        union.train()
        torch.cuda.synchronize()
        assert agent.get_global_step() == step0 + step + 1
        # -- PER index selection: the owners' ids, shifted to the union's numbering, are the union buffer's draw
        owner_mine, ids_own = rb.sharded._plan
        both = _gather(ids_own)
        owned = torch.stack([b >= 0 for b in both])
        assert bool((owned.sum(0) == 1).all()), 'every sample of the global batch has exactly one owner'
        union_ids = torch.where(both[0] >= 0, both[0], both[1] + SHARD_C)
        assert torch.equal(union_ids, urb._ids), f'step {step}: ids'
        assert bool(owned[1].any()) and bool(owned[0].any()), 'both shards are drawn from'
        assert torch.equal(owner_mine, (urb._ids[lo:hi] >= SHARD_C).long())
        # -- IS weights and windows of MY rows: bit for bit the union's slice
        assert torch.equal(rb._w, urb._w[lo:hi]), f'step {step}: IS weights'
        assert float(rb._beta) == float(urb._beta)
        for k in rb._batch:
            assert torch.equal(rb._batch[k], urb._batch[k][lo:hi]), f'step {step}: window {k}'
        # -- the step: reduced gradients (Adam first moment: m' - b1 m = (1 - b1) g) against the full-batch learner's
        b1 = 0.9
        g_dp = ((agent._exp_avg - b1 * m0) / (1 - b1)).cpu().numpy()
        g_un = ((union._exp_avg - b1 * m0) / (1 - b1)).cpu().numpy()
        for name, (s, e) in agent._params.segments.items():
            if e == s:
                continue
            scale = float(np.abs(g_un[s:e]).max())
            pu.check(f'world2/parity/grad/{name}', g_dp[s:e], g_un[s:e], rtol=1e-3, atol=2e-5 * scale + 1e-12)
        both_w = _gather(agent._params.flat)
        assert torch.equal(both_w[0], both_w[1]), f'step {step}: replicas diverged'
        w_dp, w_un = agent._params.flat.cpu().numpy(), union._params.flat.cpu().numpy()
        assert np.abs(w_dp - w_un).max() <= 2.2 * lr, 'weights: one Adam step from identical state'
        # -- write-backs reached the OWNING shards: priorities, behaviour probabilities, hidden states
        trees = [t.cpu().numpy() for t in _gather(rb._tree)]
        leaves_dp = np.concatenate([t[SHARD_C - 1:] for t in trees])
        leaves_un = urb._tree.cpu().numpy()[WORLD * SHARD_C - 1:]
        assert np.array_equal(leaves_dp > 0, leaves_un > 0)
        pu.check('world2/parity/priorities', leaves_dp, leaves_un, rtol=2e-3, atol=1e-5)
        for k in ('mu_prob', 'pre_seq_hidden_state'):
            got = torch.cat(_gather(rb._columns[k])).cpu().numpy()
            pu.check(f'world2/parity/{k}', got, urb._columns[k].cpu().numpy(), rtol=1e-3, atol=1e-6)
        assert rb.check_tree_invariant() == 0
    rb.check_health()
    agent.close()
    union.close()
    dist.barrier()


def _worker(rank, port, mode):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.cuda.set_device(0)                          # both ranks on the one device
    dist.init_process_group('gloo', rank=rank, world_size=WORLD)
    try:
        (_throughput if mode == 'throughput' else _parity)(rank)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('mode', ['throughput', 'parity'])
def test_world2_product_learner_on_one_gpu(mode):
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(_free_port(), mode), nprocs=WORLD, join=True)


def test_bench_two_ranks_host_path_on_one_gpu(tmp_path):
    """`bench.py --gpus 2` end to end on a one-GPU box (`ASAC_BENCH_ONE_DEVICE=1`: both ranks on cuda:0 over gloo, no graph):
    the launcher, the per-rank shard fill, strong scaling's 128 rows a rank, the barrier + max-over-ranks timing and rank 0's
    ONE parseable line — the host code the driver's N = 2 / 4 / 8 runs go through.  Not a measurement (the line says so)."""
    import json
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    env = dict(os.environ, ASAC_BENCH_ONE_DEVICE='1', MASTER_ADDR='127.0.0.1')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, str(root / 'bench.py'), '--gpus', '2', '--steps', '24', '--warmup', '6', '--no-cpu-baseline',
           '--no-extras', '--profile-steps', '4', '--run-length', '0', '--fill', '20000']
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=tmp_path)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, 'rank 0 prints ONE JSON line'
    d = json.loads(lines[-1])
    assert d['n_gpus'] == 2 and d['steps'] == 24 and d['warmup'] == 6 and d['scaling'] == 'strong' and d['value'] > 0
    assert d['config']['per_gpu_batch'] == 128 and d['config']['global_batch'] == 256 and d['config']['ranks'] == 2
    assert d['config']['replay_shard_capacity'] == 262144 and 'ONE device' in d['config']['collectives']
    assert d['config']['hipgraph'] is False and d['cpu_baseline'] is None
    assert abs(d['ms_per_step'] * d['value'] - 1000.0) < 1.0          # value = steps / max-over-ranks time
