"""ORACLE (test infrastructure, not product code) — CPU/NumPy restatement of the reference's
prioritized replay: sum-tree, ring storage and the PER front-end.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
module; the product path (`advanced-soft-actor-critic_amd/`) never does.

Parity pinning: checked bit-for-bit against golden vectors minted from the imported reference
(`tests/golden/make_golden.py` -> `tests/golden/f1_sumtree.npz`, `f2_per.npz`), see
`tests/test_oracle_golden.py`.  The reference's own tests hold no vectors for this path
(SURVEY.md §4), so these fixtures are the pin.

What it follows (reference `algorithm/replay_buffer.py`):
  * `SumTreeRef`         <- SumTree, lines 145-242 (array heap, root 0, leaves [C-1, 2C-1))
  * `RingStorageRef`     <- DataStorage, lines 21-142 (ring of per-key arrays, ids mod 10*C)
  * `PrioritizedReplayRef` <- PrioritizedReplayBuffer, lines 245-477, driven *synchronously*
    (the reference's prefetch thread, lines 339-375, is replaced by a direct call, which is the
    deterministic schedule SURVEY.md §8c verified)

The only deliberate interface difference: every random draw is an explicit input (`u`), so the
same uniforms can be fed to the HIP kernels.  `np.random.uniform(lo, hi)` is `lo + (hi-lo)*u`
with `u = random_sample()`, reproduced here.
"""
import math

import numpy as np


class SumTreeRef:
    def __init__(self, capacity: int):
        capacity = int(capacity)
        assert capacity > 0 and capacity & (capacity - 1) == 0, 'capacity must be a power of two'
        self.capacity = capacity
        self.levels = int(math.log2(capacity))  # edges between root and a leaf
        self.tree = np.zeros(2 * capacity - 1, dtype=np.float32)

    # --- replay_buffer.py:172-183 -------------------------------------------------------------
    def update(self, data_idx, p) -> None:
        node = np.asarray(data_idx, dtype=np.int64) + (self.capacity - 1)
        # duplicate indices: NumPy fancy assignment keeps the last occurrence
        self.tree[node] = np.asarray(p, dtype=np.float32)
        for _ in range(self.levels):
            node = np.unique((node - 1) // 2)
            self.tree[node] = self.tree[2 * node + 1] + self.tree[2 * node + 2]  # f32 left+right

    # --- replay_buffer.py:185-205 -------------------------------------------------------------
    def stratified_values(self, batch: int, u: np.ndarray) -> np.ndarray:
        """v_i = lo_i + (hi_i - lo_i) * u_i with lo_i = i*seg, hi_i = (i+1)*seg; `seg` is the
        float32 quotient root/batch, products are taken in float64 (int64 x float32 promotes)."""
        seg = np.float32(self.tree[0] / batch)
        assert seg.dtype == np.float32
        k = np.arange(batch)
        lo = k * seg
        hi = (k + 1) * seg
        assert lo.dtype == np.float64
        return lo + (hi - lo) * np.asarray(u, dtype=np.float64)

    def sample(self, batch: int, u: np.ndarray):
        """-> (leaf index int32 [B] into the tree array, leaf priority f32 [B])"""
        return self.descend(self.stratified_values(batch, u))

    def descend(self, v: np.ndarray):
        """the descent of `sample` for explicit f64 values (lines 196-205) -> (leaf index int32, priority f32)"""
        v = np.array(v, dtype=np.float64)
        batch = len(v)
        node = np.zeros(batch, dtype=np.int32)
        for _ in range(self.levels):
            left = node * 2 + 1
            right = left + 1
            go_left = (v <= self.tree[left]) | (self.tree[right] == 0)
            v = np.where(go_left, v, v - self.tree[left])  # float64 minus float32
            node = np.where(go_left, left, right).astype(np.int32)
        return node, self.tree[node]

    @property
    def total(self) -> np.float32:
        return self.tree[0]

    def leaf_max(self) -> np.float32:
        return self.tree[self.capacity - 1:].max()

    def clear(self):
        self.tree[:] = 0


class RingStorageRef:
    def __init__(self, capacity: int):
        self.capacity = capacity
        self.max_id = 10 * capacity
        self.size = 0
        self.next_id = 0
        self.columns = None  # key -> array [capacity, *shape]; '_id' -> int64 [capacity]

    def add(self, rows: dict) -> np.ndarray:
        n = next(iter(rows.values())).shape[0]
        if self.columns is None:
            self.columns = {'_id': np.zeros(self.capacity, dtype=np.int64)}
            for k, v in rows.items():
                self.columns[k] = np.zeros((self.capacity, *v.shape[1:]), dtype=v.dtype)
        ids = (np.arange(n) + self.next_id) % self.max_id
        slots = ids % self.capacity
        self.columns['_id'][slots] = ids
        for k, v in rows.items():
            self.columns[k][slots] = v
        self.size = min(self.size + n, self.capacity)
        self.next_id = int(ids[-1] + 1)
        if self.next_id == self.max_id:
            self.next_id = 0
        return slots

    def ids_at(self, ids):
        return self.columns['_id'][np.asarray(ids) % self.capacity]

    def rows_at(self, ids) -> dict:
        slots = np.asarray(ids) % self.capacity
        return {k: v[slots] for k, v in self.columns.items() if k != '_id'}

    def write(self, ids, key, data):
        self.columns[key][np.asarray(ids) % self.capacity] = data


class PrioritizedReplayRef:
    def __init__(self, batch_size=256, sample_prev_n=0, sample_post_n=0,
                 capacity=524288, alpha=0.9, beta=0.4, beta_increment_per_sampling=0.001,
                 td_error_min=0.01, td_error_max=1.):
        self.batch_size = batch_size
        self.prev_n, self.post_n = sample_prev_n, sample_post_n
        self.capacity = int(2 ** math.floor(math.log2(capacity)))  # rounded DOWN (line 264)
        self.alpha, self.beta = alpha, beta
        self.beta_increment = beta_increment_per_sampling
        self.td_error_min, self.td_error_max = td_error_min, td_error_max
        self.tree = SumTreeRef(self.capacity)
        self.storage = RingStorageRef(self.capacity)

    # --- replay_buffer.py:293-307 -------------------------------------------------------------
    def add(self, transitions: dict, ignore_size=0) -> None:
        max_p = self.td_error_max if self.storage.size == 0 else self.tree.leaf_max()
        slots = self.storage.add(transitions)
        p = np.full(len(slots), max_p, dtype=np.float32)
        if ignore_size > 0:
            p[slots >= self.capacity - ignore_size] = 0  # ring tail is never a window start
            p[-ignore_size:] = 0                          # last rows of the episode
        self.tree.update(slots, p)

    def priorities_from_td(self, td_error) -> np.ndarray:
        clipped = np.clip(np.asarray(td_error).flatten(), self.td_error_min, self.td_error_max)
        if np.isnan(np.min(clipped)):
            raise Exception('td_error has nan')
        return np.power(clipped, self.alpha)

    def add_with_td_error(self, td_error, transitions: dict, ignore_size=0) -> None:
        slots = self.storage.add(transitions)
        p = self.priorities_from_td(td_error)
        if ignore_size > 0:
            p[slots >= self.capacity - ignore_size] = 0
            p[-ignore_size:] = 0
        self.tree.update(slots, p)

    # --- replay_buffer.py:345-364, 377-396 ----------------------------------------------------
    @property
    def is_lg_batch_size(self) -> bool:
        return self.storage.size > self.batch_size

    def sample(self, u: np.ndarray):
        """-> None | (data ids int64 [B], {key: array [B, L, *]}, IS weights f32 [B, 1])"""
        if not self.is_lg_batch_size:
            return None
        leaf, p = self.tree.sample(self.batch_size, u)
        ids = self.storage.ids_at(leaf - (self.capacity - 1))

        w = p / self.tree.total                                           # float32
        self.beta = np.min([1., self.beta + self.beta_increment])        # np.float64 from here on
        w = np.power(w / np.min(w), -self.beta).astype(np.float32)       # float64 power (NumPy 2)

        offsets = np.arange(-self.prev_n, self.post_n + 1, dtype=np.int64)
        window_ids = (ids[:, None] + offsets[None, :]).reshape(-1)
        rows = self.storage.rows_at(window_ids)
        L = self.prev_n + 1 + self.post_n
        windows = {k: v.reshape(self.batch_size, L, *v.shape[1:]) for k, v in rows.items()}
        return ids, windows, w[:, None]

    # --- replay_buffer.py:412-434 -------------------------------------------------------------
    def update(self, ids, td_error) -> None:
        ids = np.asarray(ids)
        p = self.priorities_from_td(td_error)
        live = self.storage.ids_at(ids) == ids   # slot not overwritten since the sample
        self.tree.update(ids[live] % self.capacity, p[live])

    def update_transitions(self, ids, key, data) -> None:
        ids = np.asarray(ids)
        live = self.storage.ids_at(ids) == ids
        self.storage.write(ids[live], key, data[live])

    @property
    def size(self):
        return self.storage.size

    def get_curr_id(self):
        return self.storage.next_id % self.capacity
