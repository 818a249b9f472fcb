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

    def all_ready(self, ready: bool, size: int = 0, device='cpu'):
        """-> (every rank is ready, sum of `size` over the ranks): ONE collective, called by every rank at the same
        point of its loop (`SAC_Base.train` before its first step).  A step of the data-parallel learner holds
        collectives, so whether it runs has to be decided by all ranks together: a rank whose shard is still short may
        not return early while the others wait for it inside an all-reduce."""
        if not self._live:
            return bool(ready), int(size)
        t = torch.tensor([1 if ready else 0, int(size)], dtype=torch.int64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        flag, total = (int(x) for x in t.cpu())
        return flag == self.world_size, total


# ================================================================================================
# "Parity" sharded sampling: the G shard trees are the subtrees of ONE sum tree whose top log2 G
# levels are formed from the shard roots — so a batch drawn over the sharded replay is exactly the reference's
# stratified sample (replay_buffer.py:185-205) over the union of the shards, G = 1 being the plain buffer.
#   1. all-gather the G shard roots (f32)                                    -> every rank builds the same top heap
#   2. every rank draws the SAME B stratified values (shared uniforms) and walks the top levels with the
#      reference's comparisons (f64 value against f32 node sums)              -> owner shard + residual value per sample
#   3. the owner finishes the descent in its own tree                         -> leaf priority, local id; priority 0 and
#      id -1 for the samples of other shards
#   4. priorities are summed into a [B] vector (one all-reduce: every sample has one owner) -> IS weights with the
#      global total and the global minimum, the reference's formula (replay_buffer.py:352-354)
#   5. every rank gathers the windows of ALL B ids (its own where it owns the sample) and ONE equal-split all-to-all per
#      key hands rank r the B / G rows it trains (sample i is trained by rank i // (B / G)) from every shard; it keeps,
#      per row, the copy of the owner.  td-errors and row write-backs travel back by all-gather; each owner applies
#      them to its ids (id -1: skipped like a stale row).
# Every buffer has a fixed shape and nothing is read on the host: on the product backend (`ProductShard`: steps 2 - 4
# are the launches asac_sumtree_plan_top / asac_sumtree_descend_owned / asac_per_is_weights_slice) the sharded step is
# captured into the hipGraph like the plain one.  `plan_global_sample` is the host statement of steps 1 - 2 (the
# oracle backend of the CPU tests and the specification the kernel is tested against).
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
    """One instance per rank around that rank's replay shard.  `shard` must offer (`ProductShard` / the oracle adapter
    of the tests; B = the global batch, tensors on `device`):
         root_tensor() -> f32 [1]
         plan_and_descend(roots f32 [G], u f64 [B], rank) -> (owner int [B], p f32 [B], ids i64 [B], total f32 [1]);
                                                  p = 0 and id = -1 where owner != rank
         is_weights(p_all f32 [B], total f32 [1], first, count) -> f32 [count]   (advances beta first)
         windows(ids i64 [B]) -> {key: tensor [B, L, ...]}   (rows of id -1: anything)
         update(ids i64 [B], td f32 [B]),  update_windows(ids, first_off, count, mask [B, count], key, rows [B, count, ...]):
                                                  ids == -1 are skipped"""

    def __init__(self, ctx: DataParallelContext, shard, batch_size: int, device):
        assert batch_size % ctx.world_size == 0
        self.ctx, self.shard, self.B, self.device = ctx, shard, batch_size, torch.device(device)
        self.G, self.rank, self.per = ctx.world_size, ctx.rank, batch_size // ctx.world_size
        assert self.G & (self.G - 1) == 0, 'the number of shards must be a power of two'
        self._roots = torch.zeros(self.G, dtype=torch.float32, device=self.device)
        self._cols = torch.arange(self.per, device=self.device)
        self._plan = None       # (owner of my rows i64 [B / G], ids of the samples I own i64 [B])
        self._live = ctx._live  # (a world of one rank with `always=True` still issues every collective)

    # -- fixed-size exchanges -------------------------------------------------------------------------------------------
    @staticmethod
    def _wire(t: torch.Tensor) -> torch.Tensor:
        return t.view(torch.uint8) if t.dtype == torch.bool else t

    def _to_trainers(self, rows_all: torch.Tensor, owner_mine: torch.Tensor) -> torch.Tensor:
        """rows_all [B, ...]: this shard's rows for every sample of the global batch -> [B / G, ...]: the rows I train,
        each taken from the shard that owns it"""
        src = self._wire(rows_all.contiguous())
        if not self._live:
            return rows_all
        out = torch.empty_like(src)
        dist.all_to_all_single(out, src, group=self.ctx.group)            # [G (source shard), B / G, ...]
        out = out.view(self.G, self.per, *src.shape[1:])[owner_mine, self._cols]
        return out.view(torch.bool) if rows_all.dtype == torch.bool else out

    def _to_owners(self, rows_local: torch.Tensor) -> torch.Tensor:
        """rows_local [B / G, ...] of the samples I train -> [B, ...] of the whole batch (sample order)"""
        src = self._wire(rows_local.contiguous())
        if not self._live:
            return rows_local
        out = torch.empty((self.B, *src.shape[1:]), dtype=src.dtype, device=self.device)
        dist.all_gather_into_tensor(out, src, group=self.ctx.group)
        return out.view(torch.bool) if rows_local.dtype == torch.bool else out

    def sample(self, u: torch.Tensor):
        """`u`: the batch's B uniforms (f64 tensor), the SAME on every rank -> (windows {key: [B/G, L, ...]}, IS weights
        f32 [B/G], owner of every sample [B])"""
        root = self.shard.root_tensor()
        if self._live:
            dist.all_gather_into_tensor(self._roots, root, group=self.ctx.group)
        else:
            self._roots.copy_(root)
        owner, p_all, ids_own, total = self.shard.plan_and_descend(self._roots, u, self.rank)
        if self._live:
            dist.all_reduce(p_all, group=self.ctx.group)      # every entry has exactly one non-zero contributor
        lo = self.rank * self.per
        w = self.shard.is_weights(p_all, total, lo, self.per)
        owner_mine = owner[lo:lo + self.per].long()
        windows = {k: self._to_trainers(rows, owner_mine) for k, rows in self.shard.windows(ids_own).items()}
        self._plan = (owner_mine, ids_own)
        return windows, w, owner

    def sample_into(self, rb) -> None:
        """`rb.sample_into_static()` in parity mode: rank 0's uniforms for the GLOBAL batch reach every rank, the
        rows this rank trains on land in the buffer's static batch tensors (`rb.sharded = self` routes the step's
        write-backs through `update` / `update_windows`).  No host synchronisation: graph-capturable."""
        u = rb._u            # [B] in this mode: the step's noise launch draws the global batch's uniforms into it
        assert u.numel() == self.B
        rb.uniform_source.fill(u)
        if self._live:
            dist.broadcast(u, src=0, group=self.ctx.group)
        windows, w, _ = self.sample(u)
        for k, v in windows.items():
            rb._batch[k].copy_(v)
        rb._w.copy_(w)
        rb._ids.fill_(-1)        # the sampled ids live on their owners (kept in the exchange plan)

    def update(self, td_local: torch.Tensor) -> None:
        """td-errors of my B / G rows -> priorities on the shards that own them (reference PER.update, 412-427)"""
        _, ids_own = self._plan
        self.shard.update(ids_own, self._to_owners(td_local.reshape(-1)))

    def update_windows(self, first_off: int, count: int, padding_mask: torch.Tensor, key: str, rows: torch.Tensor) -> None:
        """rows[s, j] of my B / G samples -> id(s) + first_off + j on the owning shards (update_transitions, 429-434)"""
        _, ids_own = self._plan
        self.shard.update_windows(ids_own, first_off, count, self._to_owners(padding_mask), key, self._to_owners(rows))


class ProductShard:
    """`ShardedParityReplay` backend over this rank's HBM-resident `PrioritizedReplayBuffer`: three launches for the
    plan, nothing on the host."""

    def __init__(self, rb, global_batch: int):
        self.rb, B, dev = rb, global_batch, rb.device
        self._owner = torch.zeros(B, dtype=torch.int32, device=dev)
        self._v = torch.zeros(B, dtype=torch.float64, device=dev)
        self._total = torch.zeros(1, dtype=torch.float32, device=dev)
        self._leaf = torch.zeros(B, dtype=torch.int32, device=dev)
        self._p = torch.zeros(B, dtype=torch.float32, device=dev)
        self._ids = torch.zeros(B, dtype=torch.int64, device=dev)

    def root_tensor(self) -> torch.Tensor:
        return self.rb._tree[0:1]

    def plan_and_descend(self, roots, u, rank):
        from asac_amd import native
        rb = self.rb
        native.sumtree_plan_top(roots, u.numel(), u, self._owner, self._v, self._total)
        native.sumtree_descend_owned(rb._tree, rb.capacity, self._v, self._owner, rank, rb._slot_ids, self._leaf, self._p,
                                     self._ids)
        return self._owner, self._p, self._ids, self._total

    def is_weights(self, p_all, total, first, count):
        from asac_amd import native
        rb = self.rb
        w = torch.empty(count, dtype=torch.float32, device=rb.device)
        native.per_is_weights_slice(p_all, first, count, total, rb._beta, rb.beta_increment_per_sampling, w)
        return w

    def windows(self, ids: torch.Tensor) -> dict:
        return self.rb.gather_windows(ids)

    def update(self, ids, td) -> None:
        self.rb.update(ids, td)                 # (not the step's `rb._ids`: applied to this shard; id -1 = stale)

    def update_windows(self, ids, first_off, count, padding_mask, key, rows) -> None:
        self.rb.update_window_transitions(ids, first_off, count, padding_mask.contiguous(), key, rows.contiguous())
