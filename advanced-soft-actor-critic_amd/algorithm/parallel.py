"""Data-parallel context for the SAC step: one process per GPU, `torch.distributed` over
RCCL/xGMI (backend "nccl"), replay sharded by episode, models replicated.

The reference has no distributed code at all; this is new functionality with
exactly the exchange steps the path needs:
  * gradient mean all-reduce over the flat gradient segment of the optimizers about to step
    (108 KB for the stock MLPs: latency-bound, so one flat bucket per exchange, no bucketing loop)
  * one-scalar MIN all-reduce so the importance-sampling weights are normalised by the global
    minimum sampling probability, as the single-buffer formula does (replay_buffer.py:352-354).
    With per-rank stratified sampling of B/G... rows from a shard, P(i) = p_i / (G * sum_shard p), the
    1/G cancels in P(i)/min_j P(j), so only the minimum ratio has to cross ranks.
Everything here is device-agnostic torch.distributed code (covered on CPU with gloo, world 2).
"""
import torch
import torch.distributed as dist

__all__ = ['DataParallelContext', 'shard_of_episode', 'top_heap', 'plan_global_sample', 'ShardedParityReplay',
           'ProductShard']


def shard_of_episode(episode_counter: int, world_size: int) -> int:
    """Episodes land whole on one shard (windows need contiguous neighbours): round-robin."""
    return episode_counter % world_size


class DataParallelContext:
    def __init__(self, process_group=None, always: bool = False):
        """`always`: issue the collectives even in a world of one rank (where they change nothing), so a
        single-GPU box exercises the RCCL calls, eagerly and inside a captured graph."""
        if not dist.is_initialized():
            raise RuntimeError('torch.distributed is not initialised')
        self.group = process_group
        self.world_size = dist.get_world_size(process_group)
        self.rank = dist.get_rank(process_group)
        self._inv_world = 1.0 / self.world_size
        self._live = always or self.world_size > 1
        # RCCL averages inside the collective; gloo (the CPU tests) has no AVG: sum, then scale
        self._avg = dist.get_backend(process_group) == 'nccl'

    def all_reduce_grads(self, flat_grad: torch.Tensor, start: int, stop: int) -> None:
        """Mean over ranks of flat_grad[start:stop], in place, as ONE collective."""
        if stop <= start or not self._live:
            return
        seg = flat_grad[start:stop]
        if self._avg:
            dist.all_reduce(seg, op=dist.ReduceOp.AVG, group=self.group)
            return
        dist.all_reduce(seg, op=dist.ReduceOp.SUM, group=self.group)
        seg.mul_(self._inv_world)

    def all_reduce_min_(self, scalar: torch.Tensor) -> None:
        if self._live:
            dist.all_reduce(scalar, op=dist.ReduceOp.MIN, group=self.group)

    def broadcast_(self, flat: torch.Tensor, src: int = 0) -> None:
        """Replicate initial weights from rank `src` (models are constructed per rank)."""
        if self._live:
            dist.broadcast(flat, src=src, group=self.group)

    def barrier(self) -> None:
        dist.barrier(group=self.group)


# ================================================================================================
# "Parity" sharded sampling: the G shard trees are the subtrees of ONE sum tree whose top log2 G
# levels are formed from the shard roots — so a batch drawn over the sharded replay is exactly the reference's
# stratified sample (replay_buffer.py:185-205) over the union of the shards, G = 1 being the plain buffer.
#   1. all-gather the G shard roots (f32)                                    -> every rank builds the same top heap
#   2. every rank draws the SAME B stratified values (shared uniforms) and walks the top levels with the
#      reference's comparisons (f64 value against f32 node sums)              -> owner shard + residual value per sample
#   3. the owner finishes the descent in its own tree (asac_sumtree_descend)  -> leaf priority, local id
#   4. priorities are summed into a [B] vector (one all-reduce: every sample has one owner) -> IS weights with the
#      global total and the global minimum, the reference's formula (replay_buffer.py:352-354)
#   5. the sampled windows travel by ONE all-to-all per key so every rank trains on B / G rows (sample i is trained
#      by rank i // (B / G)); td-errors and row write-backs travel back to the owners the same way.
# Device-agnostic torch.distributed code; the shard backend is duck-typed (product buffer or the NumPy oracle).
# ================================================================================================
import math

import numpy as np


def top_heap(roots: np.ndarray) -> np.ndarray:
    """array heap (root 0) over G = 2^k f32 leaves, parent = left + right in f32 (replay_buffer.py:172-183)"""
    G = len(roots)
    assert G & (G - 1) == 0, 'the number of shards must be a power of two'
    heap = np.zeros(2 * G - 1, dtype=np.float32)
    heap[G - 1:] = np.asarray(roots, dtype=np.float32)
    for node in range(G - 2, -1, -1):
        heap[node] = heap[2 * node + 1] + heap[2 * node + 2]
    return heap


def plan_global_sample(roots: np.ndarray, batch: int, u: np.ndarray):
    """-> (owner shard i64 [B], residual value f64 [B], total f32): steps 1-2 above, identical on every rank"""
    heap = top_heap(roots)
    G = len(roots)
    seg = np.float32(heap[0] / batch)
    k = np.arange(batch)
    lo, hi = k * seg, (k + 1) * seg                       # int64 * float32 -> float64
    v = lo + (hi - lo) * np.asarray(u, dtype=np.float64)
    node = np.zeros(batch, dtype=np.int64)
    for _ in range(int(math.log2(G))):
        left, right = 2 * node + 1, 2 * node + 2
        go_left = (v <= heap[left]) | (heap[right] == 0)
        v = np.where(go_left, v, v - heap[left])
        node = np.where(go_left, left, right)
    return node - (G - 1), v, heap[0]


class ShardedParityReplay:
    """One instance per rank around that rank's replay shard.  `shard` must offer
         root() -> float,  descend(v f64 ndarray) -> (p f32 [n], ids i64 [n]) as torch tensors on `device`,
         windows(ids) -> {key: tensor [n, L, ...]},  update(ids, td),  update_windows(ids, first_off, count, mask, key, rows)
    (`ProductShard` / the oracle adapter of the tests)."""

    def __init__(self, ctx: DataParallelContext, shard, batch_size: int, device, beta=0.4, beta_increment=0.001):
        assert batch_size % ctx.world_size == 0
        self.ctx, self.shard, self.B, self.device = ctx, shard, batch_size, torch.device(device)
        self.G, self.rank, self.per = ctx.world_size, ctx.rank, batch_size // ctx.world_size
        self.beta, self.beta_increment = beta, beta_increment
        self._plan = None

    # -- exchange helpers: rows of the samples I OWN (ascending sample index) <-> rows of the samples I TRAIN ------------
    def _splits(self, owner):
        target = np.arange(self.B) // self.per
        mine = owner == self.rank
        send = [int(np.sum(mine & (target == d))) for d in range(self.G)]                 # owned by me, trained by d
        recv = [int(np.sum((owner == s) & (target == self.rank))) for s in range(self.G)]  # owned by s, trained by me
        # position of every received row inside my local batch: rows arrive grouped by source, ascending sample index
        order = np.concatenate([np.nonzero((owner == s) & (target == self.rank))[0] for s in range(self.G)]) - self.rank * self.per
        return send, recv, torch.from_numpy(order.astype(np.int64)).to(self.device)

    def _to_trainers(self, rows_owned: torch.Tensor, send, recv, order) -> torch.Tensor:
        out = torch.empty((sum(recv), *rows_owned.shape[1:]), dtype=rows_owned.dtype, device=self.device)
        if self.G > 1:
            dist.all_to_all_single(out, rows_owned.contiguous(), recv, send, group=self.ctx.group)
        else:
            out.copy_(rows_owned)
        local = torch.empty_like(out)
        local[order] = out
        return local

    def _to_owners(self, rows_local: torch.Tensor, send, recv, order) -> torch.Tensor:
        grouped = rows_local[order].contiguous()          # back into (source, ascending sample index) order
        out = torch.empty((sum(send), *rows_local.shape[1:]), dtype=rows_local.dtype, device=self.device)
        if self.G > 1:
            dist.all_to_all_single(out, grouped, send, recv, group=self.ctx.group)
        else:
            out.copy_(grouped)
        return out

    def sample(self, u: np.ndarray):
        """`u`: the batch's B uniforms, the SAME on every rank -> (windows {key: [B/G, L, ...]}, IS weights f32 [B/G],
        global sample indexes of my rows)"""
        roots = torch.tensor([self.shard.root()], dtype=torch.float32, device=self.device)
        gathered = [torch.zeros_like(roots) for _ in range(self.G)]
        if self.G > 1:
            dist.all_gather(gathered, roots, group=self.ctx.group)
        else:
            gathered = [roots]
        owner, v, total = plan_global_sample(torch.cat(gathered).cpu().numpy(), self.B, u)
        mine = np.nonzero(owner == self.rank)[0]
        p_own, ids_own = self.shard.descend(v[mine])
        p_all = torch.zeros(self.B, dtype=torch.float32, device=self.device)
        p_all[torch.from_numpy(mine).to(self.device)] = p_own
        if self.G > 1:
            dist.all_reduce(p_all, group=self.ctx.group)      # every entry has exactly one non-zero contributor
        self.beta = min(1., self.beta + self.beta_increment)
        ratio = p_all / float(total)
        w = torch.pow((ratio / ratio.min()).double(), -self.beta).float()
        send, recv, order = self._splits(owner)
        windows = {k: self._to_trainers(rows, send, recv, order) for k, rows in self.shard.windows(ids_own).items()}
        self._plan = (send, recv, order, ids_own)
        lo = self.rank * self.per
        return windows, w[lo:lo + self.per], np.arange(lo, lo + self.per)

    def sample_into(self, rb) -> None:
        """`rb.sample_into_static()` in parity mode: rank 0's uniforms for the GLOBAL batch reach every rank, the
        rows this rank trains on land in the buffer's static batch tensors (`rb.sharded = self` routes the step's
        write-backs through `update` / `update_windows`)."""
        u = torch.empty(self.B, dtype=torch.float64, device=self.device)
        rb.uniform_source.fill(u)
        if self.G > 1:
            dist.broadcast(u, src=0, group=self.ctx.group)
        windows, w, _ = self.sample(u.cpu().numpy())
        for k, v in windows.items():
            rb._batch[k].copy_(v)
        rb._w.copy_(w)
        rb._ids.fill_(-1)        # the sampled ids live on their owners (kept in the exchange plan)

    def update(self, td_local: torch.Tensor) -> None:
        """td-errors of my B / G rows -> priorities on the shards that own them (reference PER.update, 412-427)"""
        send, recv, order, ids_own = self._plan
        self.shard.update(ids_own, self._to_owners(td_local.reshape(-1, 1), send, recv, order).reshape(-1))

    def update_windows(self, first_off: int, count: int, padding_mask: torch.Tensor, key: str, rows: torch.Tensor) -> None:
        """rows[s, j] of my B / G samples -> id(s) + first_off + j on the owning shards (update_transitions, 429-434)"""
        send, recv, order, ids_own = self._plan
        self.shard.update_windows(ids_own, first_off, count, self._to_owners(padding_mask, send, recv, order), key,
                                  self._to_owners(rows, send, recv, order))


class ProductShard:
    """`ShardedParityReplay` backend over this rank's HBM-resident `PrioritizedReplayBuffer`."""

    def __init__(self, rb):
        self.rb = rb

    def root(self) -> float:
        return float(self.rb._tree[0].item())

    def descend(self, v: np.ndarray):
        from asac_amd import native
        rb, n = self.rb, len(v)
        p = torch.empty(n, dtype=torch.float32, device=rb.device)
        ids = torch.empty(n, dtype=torch.int64, device=rb.device)
        if n:
            leaf = torch.empty(n, dtype=torch.int32, device=rb.device)
            native.sumtree_descend(rb._tree, rb.capacity, torch.from_numpy(np.ascontiguousarray(v)).to(rb.device),
                                   rb._slot_ids, leaf, p, ids)
        return p, ids

    def windows(self, ids: torch.Tensor) -> dict:
        return self.rb.gather_windows(ids)

    def update(self, ids, td) -> None:
        if ids.numel():
            self.rb.update(ids, td)

    def update_windows(self, ids, first_off, count, padding_mask, key, rows) -> None:
        if ids.numel():
            self.rb.update_window_transitions(ids, first_off, count, padding_mask.contiguous(), key, rows.contiguous())
