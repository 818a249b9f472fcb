"""CPU, world_size 2 over gloo: the data-parallel exchange steps of the path
(`algorithm/parallel.py`) — gradient mean over the flat segment, global-min normalisation of the
importance weights for sharded replay, weight broadcast."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle.per_ref import PrioritizedReplayRef


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import asac_amd  # noqa: F401
    from algorithm.parallel import DataParallelContext, shard_of_episode
    ctx = DataParallelContext()
    assert ctx.world_size == world and ctx.rank == rank

    # (1) gradient mean over a flat segment only
    flat = torch.arange(10, dtype=torch.float32) * (rank + 1)
    ctx.all_reduce_grads(flat, 2, 8)
    want = torch.arange(10, dtype=torch.float32) * (rank + 1)
    want[2:8] = torch.arange(2, 8, dtype=torch.float32) * 1.5
    assert torch.equal(flat, want)

    # (2) broadcast of rank 0's weights
    w = torch.full((5,), float(rank + 7))
    ctx.broadcast_(w)
    assert torch.all(w == 7)

    # (2b) whether a step runs is decided by all ranks together (a step holds collectives: a rank with a short shard
    # must not return early while the others wait inside an all-reduce)
    assert ctx.all_ready(rank == 0, size=10 + rank) == (False, 21)
    assert ctx.all_ready(True, size=3) == (True, 6)

    # (3) sharded replay: episodes round-robin over shards; IS weights normalised by the GLOBAL min ratio
    rng = np.random.default_rng(0)                 # same stream on both ranks
    shard = PrioritizedReplayRef(batch_size=8, capacity=64)
    full_p, full_tot = [], []
    for ep in range(6):
        T = int(rng.integers(5, 12))
        rows = {'x': rng.standard_normal(T).astype(np.float32)}
        pr = rng.random(T).astype(np.float32) + 0.05
        if shard_of_episode(ep, world) == rank:
            first = shard.storage.next_id
            shard.add(rows, ignore_size=0)
            shard.tree.update(np.arange(first, first + T), pr)
    leaf, p = shard.tree.sample(8, np.random.default_rng(rank).random(8))
    ratio = torch.from_numpy(p / shard.tree.total)
    m = ratio.min().reshape(1).clone()
    ctx.all_reduce_min_(m)
    gathered = [torch.zeros(8) for _ in range(world)]
    dist.all_gather(gathered, ratio)
    assert m.item() == torch.cat(gathered).min().item()
    w_is = (ratio / m) ** -0.4
    assert w_is.max().item() <= 1.0 + 1e-6           # the globally rarest sample has weight 1
    torch.save({'w': w_is, 'min': m}, os.path.join(out_dir, f'r{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_context_world2(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = (torch.load(tmp_path / f'r{r}.pt') for r in range(2))
    assert a['min'].item() == b['min'].item()
    assert min(a['w'].max().item(), b['w'].max().item()) <= 1.0 + 1e-6
    assert max(a['w'].max().item(), b['w'].max().item()) == 1.0   # exactly one shard holds the global minimum


# ------------------------------------------------------------------------------------------------
# "parity" sharded sampling: two ranks, each with an oracle shard; the union must sample exactly like
# ONE reference tree whose leaves are [shard 0 | shard 1]
# ------------------------------------------------------------------------------------------------
class _OracleShard:
    """`ShardedParityReplay` backend over the NumPy oracle buffer: the plan is the host statement
    (`parallel.plan_global_sample`) of what the product's backend does with three launches"""

    def __init__(self, rb, beta=0.4, beta_increment=0.001):
        self.rb, self.beta, self.beta_increment = rb, beta, beta_increment

    def root_tensor(self):
        return torch.tensor([self.rb.tree.total], dtype=torch.float32)

    def plan_and_descend(self, roots, u, rank):
        from algorithm.parallel import plan_global_sample
        owner, v, total = plan_global_sample(roots.numpy(), len(u), u.numpy())
        mine = owner == rank
        p, ids = np.zeros(len(u), np.float32), np.full(len(u), -1, np.int64)
        leaf, p[mine] = self.rb.tree.descend(v[mine])
        ids[mine] = self.rb.storage.ids_at(leaf - (self.rb.capacity - 1))
        return (torch.from_numpy(owner.astype(np.int64)), torch.from_numpy(p), torch.from_numpy(ids),
                torch.tensor([total], dtype=torch.float32))

    def is_weights(self, p_all, total, first, count):
        self.beta = min(1., self.beta + self.beta_increment)
        ratio = p_all.numpy() / np.float32(total.item())
        w = np.power(ratio / np.min(ratio), -np.float64(self.beta)).astype(np.float32)
        return torch.from_numpy(w[first:first + count])

    def windows(self, ids):
        ids = np.where(ids.numpy() < 0, 0, ids.numpy())          # (rows of other shards' samples: anything)
        off = np.arange(-self.rb.prev_n, self.rb.post_n + 1)
        rows = self.rb.storage.rows_at((ids[:, None] + off[None, :]).reshape(-1))
        L = len(off)
        return {k: torch.from_numpy(v.reshape(len(ids), L, *v.shape[1:])) for k, v in rows.items()}

    def update(self, ids, td):
        mine = ids.numpy() >= 0
        if mine.any():
            self.rb.update(ids.numpy()[mine], td.numpy()[mine])

    def update_windows(self, ids, first_off, count, mask, key, rows):
        mine = ids.numpy() >= 0
        ids, mask, rows = ids.numpy()[mine], mask.numpy()[mine], rows.numpy()[mine]
        tgt = (ids[:, None] + first_off + np.arange(count)[None, :]).reshape(-1)
        keep = ~mask[:, :count].reshape(-1)
        self.rb.update_transitions(tgt[keep], key, rows[:, :count].reshape(len(tgt), *rows.shape[2:])[keep])


def _parity_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import asac_amd  # noqa: F401
    from algorithm.parallel import DataParallelContext, ShardedParityReplay, shard_of_episode
    from oracle.per_ref import SumTreeRef
    ctx = DataParallelContext()
    Cs, B = 64, 16
    rng = np.random.default_rng(5)                  # the same stream on every rank: every rank knows every episode
    shards = [PrioritizedReplayRef(batch_size=B, sample_prev_n=1, sample_post_n=2, capacity=Cs) for _ in range(world)]
    for ep in range(10):
        T = int(rng.integers(6, 14))
        rows = {'index': np.arange(T, dtype=np.int32), 'x': rng.standard_normal((T, 3)).astype(np.float32),
                'mu_prob': rng.random((T, 2)).astype(np.float32)}
        pr = (rng.random(T) + 0.05).astype(np.float32)
        s = shards[shard_of_episode(ep, world)]
        first = s.storage.next_id
        s.add(rows, ignore_size=1)
        s.tree.update(np.arange(first, first + T - 1) % Cs, pr[:-1])
    mine = shards[rank]
    sharded = ShardedParityReplay(ctx, _OracleShard(mine), B, 'cpu')
    # the single reference tree over the union of the shards
    union = SumTreeRef(world * Cs)
    union.update(np.arange(world * Cs), np.concatenate([s.tree.tree[Cs - 1:] for s in shards]))
    for it in range(3):
        u = rng.random(B)
        windows, w, plan_owner = sharded.sample(torch.from_numpy(u))
        gidx = np.arange(rank * (B // world), (rank + 1) * (B // world))
        leaf, p = union.sample(B, u)
        owner, slot = (leaf - (world * Cs - 1)) // Cs, (leaf - (world * Cs - 1)) % Cs
        assert np.array_equal(plan_owner.numpy(), owner), 'every sample is owned by the shard that holds the union tree\'s leaf'
        ratio = p / union.total
        w_ref = np.power(ratio / np.min(ratio), -np.float64(0.4 + 0.001 * (it + 1))).astype(np.float32)
        assert np.array_equal(w.numpy(), w_ref[gidx]), 'IS weights of the global batch (bit-exact)'
        assert len(windows['x']) == B // world
        for j, i in enumerate(gidx):                # my rows are the owner's windows around the union tree's leaf
            src = shards[owner[i]]
            sid = src.storage.ids_at(np.array([slot[i]]))[0]
            want = src.storage.rows_at(sid + np.arange(-1, 3))
            for k in ('index', 'x', 'mu_prob'):
                assert np.array_equal(windows[k][j].numpy(), want[k]), (it, i, k)
        # write-backs travel to the owners: td-errors -> priorities, new mu rows -> the owner's ring
        td = torch.from_numpy((np.abs(rng.standard_normal(B)) * 0.5).astype(np.float32))[gidx]
        before = mine.tree.tree.copy()
        sharded.update(td)
        new_mu = torch.full((len(gidx), 3, 2), float(100 + it)) + torch.from_numpy(gidx.astype(np.float32))[:, None, None]
        mask = torch.zeros((len(gidx), 3), dtype=torch.bool)
        ring_before = mine.storage.columns['mu_prob'].copy()
        sharded.update_windows(-1, 3, mask, 'mu_prob', new_mu)
        # every rank replays ALL updates on its copy of the shards it does not own... through the union oracle instead:
        td_full = torch.zeros(B)
        td_full[torch.from_numpy(gidx)] = td
        dist.all_reduce(td_full)
        pri = np.power(np.clip(td_full.numpy(), 0.01, 1.0), np.float32(0.9)).astype(np.float32)
        own_rows = np.nonzero(owner == rank)[0]
        ref_tree = SumTreeRef(Cs)
        ref_tree.tree[:] = before
        ref_tree.update(slot[own_rows], pri[own_rows])
        assert np.array_equal(mine.tree.tree.view(np.uint32), ref_tree.tree.view(np.uint32)), 'priorities landed on their owner'
        for i in own_rows:                         # the new mu rows of the samples I own (last writer wins on overlaps)
            sid = mine.storage.ids_at(np.array([slot[i]]))[0]
            got = mine.storage.rows_at(np.array([sid]))['mu_prob'][0]
            assert got[0] >= 100, 'the write-back reached the owning shard'
        # ... and the WHOLE ring of my shard is what a plain oracle shard holds after the same rows were written to it in
        # ascending sample order (the exchange delivers them in that order: last writer wins on overlapping windows)
        want_ring = PrioritizedReplayRef(batch_size=B, sample_prev_n=1, sample_post_n=2, capacity=Cs)
        want_ring.storage = type(mine.storage).__new__(type(mine.storage))
        want_ring.storage.__dict__.update({k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in mine.storage.__dict__.items()})
        want_ring.storage.columns = {k: v.copy() for k, v in mine.storage.columns.items()}
        want_ring.storage.columns['mu_prob'] = ring_before
        sids_own = mine.storage.ids_at(slot[own_rows])
        tgt_own = (sids_own[:, None] - 1 + np.arange(3)[None, :]).reshape(-1)
        vals_own = np.repeat((100 + it + own_rows).astype(np.float32), 3)[:, None] * np.ones((1, 2), np.float32)
        if len(own_rows):
            want_ring.update_transitions(tgt_own, 'mu_prob', vals_own)
        assert np.array_equal(mine.storage.columns['mu_prob'].view(np.uint32), want_ring.storage.columns['mu_prob'].view(np.uint32)), \
            'my shard\'s mu_prob ring after the write-backs == the oracle shard\'s'
        # keep this rank's copies of the OTHER shards (and the union tree) in step for the next round: the same
        # priorities and mu rows their owners just received (rows arrive in ascending sample order: last writer wins)
        for s_i, s in enumerate(shards):
            rows_s = np.nonzero(owner == s_i)[0]
            if s_i != rank:
                s.tree.update(slot[rows_s], pri[rows_s])
                sids = s.storage.ids_at(slot[rows_s])
                tgt = (sids[:, None] - 1 + np.arange(3)[None, :]).reshape(-1)
                vals = np.repeat((100 + it + rows_s).astype(np.float32), 3)[:, None] * np.ones((1, 2), np.float32)
                s.update_transitions(tgt, 'mu_prob', vals)
        union.update(np.arange(world * Cs), np.concatenate([s.tree.tree[Cs - 1:] for s in shards]))
    dist.barrier()
    dist.destroy_process_group()


def test_parity_sharded_sampling_world2(tmp_path):
    mp.spawn(_parity_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)


def test_parity_plan_equals_single_tree():
    """host part alone, G = 1 / 2 / 4 / 8 shards: owner + residual of `plan_global_sample` followed by the owner's
    descent is the single reference tree's sample, leaf for leaf and bit for bit"""
    import asac_amd  # noqa: F401
    from algorithm.parallel import plan_global_sample
    from oracle.per_ref import SumTreeRef
    rng = np.random.default_rng(3)
    for G in (1, 2, 4, 8):
        Cs, B = 128, 64
        leaves = (rng.random(G * Cs) * (rng.random(G * Cs) < 0.7)).astype(np.float32)     # zero-priority leaves too
        union = SumTreeRef(G * Cs)
        union.update(np.arange(G * Cs), leaves)
        shard_trees = []
        for g in range(G):
            t = SumTreeRef(Cs)
            t.update(np.arange(Cs), leaves[g * Cs:(g + 1) * Cs])
            shard_trees.append(t)
        u = rng.random(B)
        u[0], u[-1] = 0.0, np.nextafter(1.0, 0.0)
        owner, v, total = plan_global_sample(np.array([t.total for t in shard_trees], np.float32), B, u)
        assert total == union.total
        leaf_ref, p_ref = union.sample(B, u)
        for g in range(G):
            rows = np.nonzero(owner == g)[0]
            leaf, p = shard_trees[g].descend(v[rows])
            assert np.array_equal(leaf - (Cs - 1) + g * Cs, leaf_ref[rows] - (G * Cs - 1))
            assert np.array_equal(p.view(np.uint32), p_ref[rows].view(np.uint32))
