"""HBM-resident prioritized replay: the drop-in for the reference's NumPy
`PrioritizedReplayBuffer` (reference `algorithm/replay_buffer.py:245-477`) on MI355X.

What lives where
  * sum tree        f32[2C-1] in HBM, the reference's array-heap layout (so `*-rb_tree.npy` files
                    are interchangeable); sampled / updated by `asac_sumtree_*` kernels
  * ring storage    one device tensor [C, *shape] per transition key, dtype preserved (uint8 images
                    stay uint8 and are widened during the gather), plus the id map i64[C]
  * sampled batch   static device tensors [B, L, *shape] per key, rewritten in place every sample
                    (stable addresses: the whole train step is replayed as one hipGraph)
There is no prefetch thread, no pinned staging and no H2D copy per step: `sample()` is two kernel
launches on the caller's stream (`sumtree_sample`, `window_gather_pad`) and returns views of the
static batch.  Consequently a sample always sees the newest priorities (the reference's batches are
1-2 steps stale because of its queue, replay_buffer.py:275,339-375).

API kept from the reference: constructor arguments, `add`, `add_with_td_error`, `sample`,
`update`, `update_transitions`, `get_storage_data(_ids)`, `get_curr_id`, `save`/`load`, `clear`,
`copy`, `size`/`is_full`/`is_lg_batch_size`, `close`.  Differences, all forced by device residency:
  * `sample()` returns the ids as a device int64 tensor (not a NumPy array) and `update*` accept
    device tensors — no D2H round trip on the hot path; NumPy inputs are still accepted
  * a NaN td-error cannot raise synchronously: the update kernel skips the batch and raises a
    device flag which `check_health()` turns into the reference's `Exception('td_error has nan')`;
    `SAC_Base.train()` calls it every `write_summary_per_step` steps and before every checkpoint,
    `close()` / `save()` call it too
The HIP library is mandatory (`asac_amd.native`): there is no CPU fallback.
"""
import logging
import math
from pathlib import Path

import numpy as np
import torch

from asac_amd import native

# ring column dtype -> the NumPy dtype an incoming episode's key is converted to before its bytes are scattered
_NP_DTYPES = {torch.float32: np.dtype(np.float32), torch.float64: np.dtype(np.float64), torch.float16: np.dtype(np.float16),
              torch.int64: np.dtype(np.int64), torch.int32: np.dtype(np.int32), torch.int16: np.dtype(np.int16),
              torch.int8: np.dtype(np.int8), torch.uint8: np.dtype(np.uint8), torch.bool: np.dtype(np.bool_)}

__all__ = ['PrioritizedReplayBuffer']


def _f32_bits(x: float) -> int:
    return int(np.float32(x).view(np.uint32))


class _DeviceUniform:
    """Default source of the stratified uniforms: torch's Philox stream (graph-capturable)."""

    def fill(self, u: torch.Tensor) -> None:
        u.uniform_()


class PrioritizedReplayBuffer:
    def __init__(self,
                 batch_size=256,
                 sample_prev_n=0,
                 sample_post_n=0,
                 device: torch.device | None = None,

                 capacity=524288,
                 alpha=0.9,
                 beta=0.4,
                 beta_increment_per_sampling=0.001,
                 td_error_min=0.01,
                 td_error_max=1.,
                 logger_parent_name=''):
        self.batch_size = batch_size
        self.prev_n = sample_prev_n
        self.post_n = sample_post_n
        self.window = sample_prev_n + 1 + sample_post_n
        self.device = torch.device(device) if device is not None else torch.device('cuda', torch.cuda.current_device())
        if self.device.type != 'cuda':
            raise native.AsacNativeError(
                'PrioritizedReplayBuffer is HBM-resident: it needs a cuda (ROCm) device, got '
                f'{self.device}.  There is no CPU fallback; the CPU restatement lives in oracle/ for tests only.')
        native.load()

        self.capacity = int(2 ** math.floor(math.log2(capacity)))  # rounded down, replay_buffer.py:264
        self.max_id = 10 * self.capacity
        self.alpha = alpha
        self.beta_increment_per_sampling = beta_increment_per_sampling
        self.td_error_min = td_error_min
        self.td_error_max = td_error_max

        name = f'{logger_parent_name}.replay_buffer' if logger_parent_name else 'replay_buffer'
        self._logger = logging.getLogger(name)

        C, dev = self.capacity, self.device
        with torch.cuda.device(dev):
            self._tree = torch.zeros(2 * C - 1, dtype=torch.float32, device=dev)
            self._slot_ids = torch.zeros(C, dtype=torch.int64, device=dev)
            # scratch: winner map for last-writer-wins (+ spill for > 1024-item updates)
            self._winner = torch.full((C + 2 * max(4096, batch_size),), -1, dtype=torch.int32, device=dev)
            # election scratch of the row write-backs (they may overlap the priority update), and one more for a
            # write-back issued on a second stream beside another write-back (`side=True`)
            self._winner_rows = torch.full((C,), -1, dtype=torch.int32, device=dev)
            self._winner_rows_side = None
            self._nan_flag = torch.zeros(1, dtype=torch.int32, device=dev)
            self._beta = torch.tensor([beta], dtype=torch.float64, device=dev)
            self._max_p = torch.zeros(1, dtype=torch.float32, device=dev)
            self._min_p = torch.zeros(528, dtype=torch.float32, device=dev)    # [2..9], [16 (1 + f)]: exchange of the step prologue's workgroups (kept zero)
            B = batch_size
            self._u = torch.zeros(B, dtype=torch.float64, device=dev)
            self._leaf = torch.zeros(B, dtype=torch.int32, device=dev)
            self._p = torch.zeros(B, dtype=torch.float32, device=dev)
            self._ids = torch.zeros(B, dtype=torch.int64, device=dev)
            self._w = torch.ones(B, dtype=torch.float32, device=dev)
        self._init_beta = beta

        self._columns: dict[str, torch.Tensor] | None = None   # key -> ring [C, *shape]
        self._batch: dict[str, torch.Tensor] | None = None     # key -> [B, L, *shape]
        self._gather_keys = None
        self._join_action_width, self.joint_pre_action, self.derived = None, None, None
        self._size = 0
        self._next_id = 0

        self._pad_action: torch.Tensor | None = None           # enables the fused window padding
        self.uniform_source = _DeviceUniform()
        self.min_ratio_reducer = None  # callable(f32[1] tensor) -> in-place MIN over ranks (parallel.py)
        self.sharded = None            # parallel.ShardedParityReplay: sampling / write-backs over all ranks' shards
        self.lookahead, self.next_valid, self.parity, self._alt = False, False, 0, None
        self._gather_sidecar = None
        self._closed = False

    # ------------------------------------------------------------------------------------------
    # configuration used by SAC_Base
    # ------------------------------------------------------------------------------------------
    def set_window_padding(self, padding_action: torch.Tensor) -> None:
        """Fuse SAC_Base's episode-continuity padding (reference sac_base.py:2435-2453) into the
        gather: rows of a window that do not belong to the centre row's episode get index -1,
        padding_mask True, action = padding_action, reward 0, done True, mu_prob 1, hidden 0;
        uint8 / bool observations are widened to float32 (783-788)."""
        self._pad_action = padding_action.to(self.device, torch.float32).contiguous()
        self._gather_keys = None

    @property
    def beta(self) -> float:
        return float(self._beta.item())

    # ------------------------------------------------------------------------------------------
    # ingress
    # ------------------------------------------------------------------------------------------
    def _to_device(self, v) -> torch.Tensor:
        if isinstance(v, torch.Tensor):
            return v.to(self.device, non_blocking=True)
        return torch.from_numpy(np.ascontiguousarray(v)).to(self.device, non_blocking=True)

    def _store_rows(self, transitions: dict) -> tuple[int, int]:
        """Ring write of an episode (reference DataStorage.add, replay_buffer.py:30-56).
        -> (first_id, count)"""
        if (self.packed_ingress and len(transitions) <= native.MAX_GATHER_KEYS
                and all(isinstance(v, np.ndarray) for v in transitions.values())):
            return self._store_rows_packed(transitions)
        rows = {k: self._to_device(v) for k, v in transitions.items()}
        count = next(iter(rows.values())).shape[0]
        C = self.capacity
        if self._columns is None:
            self._columns = {k: torch.zeros((C, *v.shape[1:]), dtype=v.dtype, device=self.device)
                             for k, v in rows.items()}
            self._batch, self._gather_keys = None, None
        first_id = self._next_id
        # only the last C rows of an over-long episode survive (later rows overwrite earlier ones)
        skip = max(0, count - C)
        start = (first_id + skip) % C
        n1 = min(count - skip, C - start)
        for k, v in rows.items():
            col = self._columns[k]
            col[start:start + n1].copy_(v[skip:skip + n1], non_blocking=True)
            if skip + n1 < count:
                col[:count - skip - n1].copy_(v[skip + n1:], non_blocking=True)
        self._size = min(self._size + count, C)
        self._next_id = (first_id + count) % self.max_id
        return first_id, count

    packed_ingress = True      # NumPy episodes: ONE host-to-device copy and ONE scatter launch per `add`

    def _store_rows_packed(self, transitions: dict) -> tuple[int, int]:
        """`_store_rows` for an episode of NumPy arrays (the reference-style caller: `SAC_Base.put_episode` from an
        environment loop): every key's rows and the ring slots they go to are packed into one pinned staging buffer, cross
        the bus as ONE copy, and `asac_rows_move` scatters all keys into their rings in one launch — instead of a pageable
        copy and one or two ring copies per key (~14 copies an episode; 36 791 `copyBuffer` launches = 26 % of the traced
        device time of the round-4 cfg2 profile, all in the fill)."""
        arrs = {k: np.ascontiguousarray(v) for k, v in transitions.items()}
        count = next(iter(arrs.values())).shape[0]
        C = self.capacity
        if self._columns is None:
            self._columns = {k: torch.zeros((C, *v.shape[1:]), dtype=torch.from_numpy(v[:0]).dtype, device=self.device)
                             for k, v in arrs.items()}
            self._batch, self._gather_keys = None, None
        # the scatter moves raw bytes: every key must arrive in its ring's own dtype and row shape (the per-key path's
        # `copy_` converted and raised; a later episode with float64 rewards after a float32 first one would otherwise
        # write rows of another width into the ring)
        for k, v in arrs.items():
            col = self._columns.get(k)
            if col is None:
                raise KeyError(f'transition key {k!r} is not a column of this replay buffer ({list(self._columns)})')
            if tuple(v.shape[1:]) != tuple(col.shape[1:]):
                raise ValueError(f'transition key {k!r}: row shape {tuple(v.shape[1:])} does not match the ring\'s '
                                 f'{tuple(col.shape[1:])}')
            if v.shape[0] != count:
                raise ValueError(f'transition key {k!r}: {v.shape[0]} rows, the episode has {count}')
            want = _NP_DTYPES.get(col.dtype)
            if want is not None and v.dtype != want:
                arrs[k] = np.ascontiguousarray(v.astype(want))
        first_id = self._next_id
        skip = max(0, count - C)             # only the last C rows of an over-long episode survive
        live = count - skip
        if live > 0:
            offs, total = {}, 0
            for k, v in arrs.items():
                col = self._columns[k]
                rb_ = col.element_size() * int(np.prod(col.shape[1:], dtype=np.int64))      # the RING's row bytes
                offs[k] = (total, rb_)
                total += (live * rb_ + 15) & ~15
            slot_off = total
            total += 4 * live
            st = self._ingress_staging(total)
            host = st['host_np']
            for k, v in arrs.items():
                o, rb_ = offs[k]
                host[o:o + live * rb_] = v[skip:].reshape(-1).view(np.uint8)
            host[slot_off:slot_off + 4 * live].view(np.int32)[:] = (first_id + skip + np.arange(live, dtype=np.int64)) % C
            dev = st['dev']
            dev[:total].copy_(st['host'][:total], non_blocking=True)
            st['event'].record()
            specs = [dict(src=dev[offs[k][0]:], dst=self._columns[k], row_bytes=offs[k][1], src_mode=native.ROW_ITEM,
                          dst_mode=native.ROW_SLOT, src_stride0=offs[k][1], dst_stride0=offs[k][1])
                     for k in arrs if offs[k][1] > 0]          # (zero-width keys — no hidden state — have no bytes to move)
            if specs:
                native.rows_move(native.make_row_moves(specs), dev[slot_off:slot_off + 4 * live].view(torch.int32), None, None,
                                 live)
        self._size = min(self._size + count, C)
        self._next_id = (first_id + count) % self.max_id
        return first_id, count

    def _ingress_staging(self, nbytes: int) -> dict:
        """pinned host buffer + its device twin, grown as needed; the host side is reused only after the previous copy out
        of it has completed (normally long ago: one wait on its event)"""
        st = getattr(self, '_ingress', None)
        if st is None or st['host'].numel() < nbytes:
            size = max(1 << 16, 1 << (int(nbytes) - 1).bit_length())
            host = torch.empty(size, dtype=torch.uint8).pin_memory()
            st = self._ingress = {'host': host, 'host_np': host.numpy(), 'event': torch.cuda.Event(),
                                  'dev': torch.empty(size, dtype=torch.uint8, device=self.device)}
        else:
            st['event'].synchronize()
        return st

    def add(self, transitions: dict, ignore_size=0) -> None:
        """New rows enter with the current max priority; the episode's last `ignore_size` rows and
        the ring's last `ignore_size` slots get 0 (replay_buffer.py:293-307)."""
        was_empty = self._size == 0
        with torch.cuda.device(self.device):
            if not was_empty:
                native.sumtree_leaf_max(self._tree, self.capacity, self._max_p)
            first_id, count = self._store_rows(transitions)
            native.per_add(self._tree, self.capacity, first_id, count, ignore_size,
                           None if was_empty else self._max_p, self.td_error_max, self._slot_ids)

    def add_with_td_error(self, td_error, transitions: dict, ignore_size: int = 0) -> None:
        """replay_buffer.py:317-337."""
        with torch.cuda.device(self.device):
            first_id, count = self._store_rows(transitions)
            native.per_add(self._tree, self.capacity, first_id, count, 0, None, 0.0, self._slot_ids)
            td = self._to_device(np.asarray(td_error, dtype=np.float32).reshape(-1)
                                 if not isinstance(td_error, torch.Tensor) else td_error.reshape(-1).float())
            ids = (torch.arange(count, device=self.device, dtype=torch.int64) + first_id) % self.max_id
            if ignore_size > 0:
                td = td.clone()
                slots = ids % self.capacity
                keep = torch.ones(count, dtype=torch.bool, device=self.device)
                keep[-ignore_size:] = False
                keep &= slots < self.capacity - ignore_size
                self._update_ids(ids[keep], td[keep], stale_check=False)
            else:
                self._update_ids(ids, td, stale_check=False)

    # ------------------------------------------------------------------------------------------
    # sample
    # ------------------------------------------------------------------------------------------
    def join_vector_obs_with_pre_action(self, action_width: int | None) -> None:
        """Lay the static batch's vector observations out as column blocks of ONE [B, L, sum(widths) + action_width]
        tensor whose last block (`joint_pre_action`) the learner fills with the previous actions: the concatenation a
        recurrent representation starts with (reference envs/*/nn*.py: `torch.cat([obs, pre_action], dim=-1)`) then
        already exists in memory and `adjacent_cat.AdjacentCat` hands it out as a view.  The gather then also delivers the
        representation's derived window inputs (`derived`: index_x, padding_mask_x, pre_action — SAC_Base.get_bnx_data,
        reference sac_base.py:1090-1115) as derived keys of the same launch.  None: dense tensors per key, no derived keys."""
        self._join_action_width = action_width
        self._batch, self._gather_keys, self.joint_pre_action, self.derived = None, None, None, None

    def _out_kind(self, k, col, pad):
        if pad and k.startswith('obs_') and col.dtype in (torch.uint8, torch.bool):
            return torch.float32, native.CVT_U8_TO_F32_UNIT if col.dtype == torch.uint8 else native.CVT_BOOL_TO_F32
        return col.dtype, native.CVT_NONE

    def _window_specs(self, n: int, joined: bool = False):
        """destination tensors [n, L, *shape] per key (+ `padding_mask` with the fused padding) and the gather key
        table that fills them"""
        L, dev = self.window, self.device
        pad = self._pad_action is not None
        batch, specs = {}, []
        joint, joint_at = None, {}
        if joined and self._join_action_width:
            members = [(k, col.shape[1]) for k, col in self._columns.items()
                       if k.startswith('obs_') and col.dim() == 2 and col.shape[1] > 0
                       and self._out_kind(k, col, pad)[0] == torch.float32]
            if members:
                width = sum(w for _, w in members)
                joint = torch.zeros((n, L, width + self._join_action_width), dtype=torch.float32, device=dev)
                off = 0
                for k, w in members:
                    joint_at[k] = off
                    off += w
                self.joint_pre_action = joint[..., width:]
        for k, col in self._columns.items():
            shape = tuple(col.shape[1:])
            row_bytes = int(np.prod(shape, dtype=np.int64)) * col.element_size() if shape else col.element_size()
            out_dtype, convert = self._out_kind(k, col, pad)
            pitch = 0
            if k in joint_at:      # a column block of the joint tensor
                out = joint[..., joint_at[k]:joint_at[k] + shape[0]]
                pitch = joint.shape[-1] * 4
            else:
                out = torch.zeros((n, L, *shape), dtype=out_dtype, device=dev)
            batch[k] = out
            mode, word, pad_row = native.PAD_KEEP, 0, None
            if pad:
                if k == 'index':
                    mode, word = native.PAD_WORD, 0xffffffff
                elif k == 'action':
                    mode, pad_row = native.PAD_ROW, self._pad_action
                    assert self._pad_action.numel() * 4 == row_bytes, 'padding action width'
                elif k == 'reward':
                    mode, word = native.PAD_WORD, _f32_bits(0.)
                elif k == 'done':
                    mode, word = native.PAD_BYTE, 1
                elif k == 'mu_prob':
                    mode, word = native.PAD_WORD, _f32_bits(1.)
                elif k == 'pre_seq_hidden_state':
                    mode, word = native.PAD_WORD, _f32_bits(0.)
            if row_bytes == 0:
                continue   # e.g. pre_seq_hidden_state of shape [C, 0]: nothing to move
            specs.append(dict(src=col, dst=out, row_bytes=row_bytes, pad_mode=mode, pad_word=word,
                              pad_row=pad_row, convert=convert, dst_row_pitch=pitch))
        if pad:
            batch['padding_mask'] = torch.zeros((n, L), dtype=torch.bool, device=dev)
            specs.append(dict(src=None, dst=batch['padding_mask'], pad_mode=native.PAD_EMIT_MASK))
        if (joined and pad and self._join_action_width and L >= 2 and 'index' in self._columns
                and 'action' in self._columns and self._columns['action'].dtype == torch.float32
                and tuple(self._columns['action'].shape[1:]) == (self._join_action_width,)
                and len(specs) + 3 <= native.MAX_GATHER_KEYS):
            A = self._join_action_width
            pre = self.joint_pre_action if joint is not None else torch.zeros((n, L, A), dtype=torch.float32, device=dev)
            self.derived = dict(index_x=torch.zeros((n, L), dtype=torch.int32, device=dev),
                                padding_mask_x=torch.zeros((n, L), dtype=torch.bool, device=dev), pre_action=pre)
            specs.append(dict(src=self._columns['index'], dst=self.derived['index_x'], row_bytes=4, pad_mode=native.PAD_WORD,
                              pad_word=0xffffffff, derive=native.DERIVE_HOLD_LAST_NEXT))
            specs.append(dict(src=None, dst=self.derived['padding_mask_x'], pad_mode=native.PAD_EMIT_MASK,
                              derive=native.DERIVE_HOLD_LAST))
            specs.append(dict(src=self._columns['action'], dst=pre, row_bytes=4 * A, pad_mode=native.PAD_ROW,
                              pad_row=self._pad_action, derive=native.DERIVE_PREVIOUS,
                              dst_row_pitch=pre.stride(1) * 4 if joint is not None else 0))
        assert len(specs) <= native.MAX_GATHER_KEYS, 'too many transition keys for one gather launch'
        return batch, specs

    def _build_batch(self) -> None:
        self._batch, specs = self._window_specs(self.batch_size, joined=True)
        self._gather_keys = native.make_gather_keys(specs)
        self._gather_refs = specs   # keep tensors alive
        if self.lookahead:
            self._build_second_set()

    # ------------------------------------------------------------------------------------------
    # one batch in flight (the reference's schedule, replay_buffer.py:275, 339-396)
    # ------------------------------------------------------------------------------------------
    # The reference's prefetch thread keeps ONE sampled batch queued (`Queue(maxsize=1)`) while the learner trains on the
    # one before it: batch k + 1 is drawn and gathered before step k's priorities and row write-backs land.  `lookahead`
    # offers that schedule deterministically: two static batch sets; `sample_next_into_static` draws into the one the
    # learner is not training on, `swap_sets` exchanges their roles between two steps.
    _SET_ATTRS = ('_u', '_leaf', '_p', '_ids', '_w', '_min_p', '_batch', '_gather_keys', '_gather_refs',
                  'joint_pre_action', 'derived', '_gather_sidecar')

    def enable_lookahead(self) -> None:
        assert self.sharded is None and self.min_ratio_reducer is None, 'one batch in flight: single-shard replay only'
        self.lookahead, self.next_valid, self.parity = True, False, 0
        self._alt = None
        if self._gather_keys is not None:
            self._build_second_set()

    def _build_second_set(self) -> None:
        first = {a: getattr(self, a) for a in self._SET_ATTRS}
        with torch.cuda.device(self.device):
            for a in ('_u', '_leaf', '_p', '_ids', '_w', '_min_p'):
                setattr(self, a, first[a].clone())
            self._batch, specs = self._window_specs(self.batch_size, joined=True)   # (sets joint_pre_action / derived)
            self._gather_keys = native.make_gather_keys(specs)
            self._gather_refs = specs
            # each set's window gather as a sidecar job (its launch description in device memory: a blocking copy, so here)
            self._gather_sidecar = self._make_gather_sidecar()
        self._alt = {a: getattr(self, a) for a in self._SET_ATTRS}
        for a, v in first.items():
            setattr(self, a, v)
        with torch.cuda.device(self.device):
            self._gather_sidecar = self._make_gather_sidecar()
        self.next_valid = False

    def _make_gather_sidecar(self):
        return native.sidecar_window_gather(self._gather_keys, self._ids, self.batch_size, self.prev_n, self.post_n,
                                            self.capacity, self._index_ring())

    def next_gather_sidecar(self):
        """the NEXT set's window gather (its ids already drawn by the step's prologue) as a sidecar job of a launch of
        the current step; `gather_next_now` if no launch took it"""
        self.next_valid = True
        return self._alt['_gather_sidecar']

    def gather_next_now(self) -> None:
        self.swap_sets()
        try:
            self.sample_into_static(sampled=True)
        finally:
            self.swap_sets()

    def swap_sets(self) -> None:
        """the batch drawn last becomes the one `_ids` / `_batch` / ... name (host bookkeeping only)"""
        for a in self._SET_ATTRS:
            mine = getattr(self, a)
            setattr(self, a, self._alt[a])
            self._alt[a] = mine
        self.parity ^= 1

    def next_uniforms(self) -> torch.Tensor:
        return self._alt['_u']

    def sample_next_into_static(self, sampled: bool = False) -> None:
        """`sample_into_static` into the set the learner is NOT training on (device work only: capturable)"""
        self.swap_sets()
        try:
            self.sample_into_static(sampled=sampled)
        finally:
            self.swap_sets()
        self.next_valid = True

    def _index_ring(self):
        index_ring = self._columns.get('index') if self._pad_action is not None else None
        if self._pad_action is not None and index_ring is None:
            raise KeyError("window padding needs an 'index' column")
        return self._leaf if index_ring is None else index_ring   # plain gather: any i32 ring satisfies the argument

    def gather_windows(self, ids: torch.Tensor) -> dict:
        """{key: [n, L, *shape]} windows (padded like a sampled batch) around arbitrary resident ids — what a shard
        hands to the ranks that train its samples (`parallel.ShardedParityReplay`)."""
        ids = ids.reshape(-1).to(self.device, torch.int64).contiguous()
        out, specs = self._window_specs(ids.numel())
        if ids.numel():
            with torch.cuda.device(self.device):
                native.window_gather_pad(native.make_gather_keys(specs), ids, ids.numel(), self.prev_n, self.post_n,
                                         self.capacity, self._index_ring())
        return out

    def sample(self):
        """-> None | (ids i64[B] (device), {key: tensor [B, L, *]}, IS weights f32 [B, 1])
        (replay_buffer.py:345-364, 377-396).  The tensors are views of static buffers that the
        next `sample()` overwrites."""
        if not self.is_lg_batch_size:
            return None
        if self._gather_keys is None:
            self._build_batch()
        self.sample_into_static()
        return self._ids, self._batch, self._w.unsqueeze(-1)

    def sample_into_static(self, sampled: int = 0) -> None:
        """The device part of `sample()` (no host logic; safe inside graph capture).  `sampled`: 1 — the tree walk
        (leaf, p, ids, IS weights) has already been done by the caller's fused prologue launch; 2 — and the window gather
        too (`NoiseSource.begin_step_with_sample(..., gather=True)`); 3 — the tree walk has been done, the IS weights have
        not: the gather's launch forms them (`asac_window_gather_pad_w`)."""
        B, C = self.batch_size, self.capacity
        if self.sharded is not None:       # "parity" mode: the batch is drawn over every rank's shard (host logic)
            self.sharded.sample_into(self)
            return
        reducer = self.min_ratio_reducer
        if not sampled:
            self.uniform_source.fill(self._u)
            native.sumtree_sample(self._tree, C, B, self._u, self._slot_ids, self._beta,
                                  self.beta_increment_per_sampling, self._leaf, self._p, self._ids,
                                  self._w if reducer is None else None, self._min_p)
        if reducer is not None:
            # sharded replay: weights are normalised by the GLOBAL minimum sampling ratio
            # (parallel.py): local min(p)/sum(p) -> MIN all-reduce -> stand-alone weight kernel
            if not sampled:       # (the step prologue's sampler stores the ratio itself)
                torch.div(self._min_p[0:1], self._tree[0:1], out=self._min_p[1:2])
            reducer(self._min_p[1:2])
            native.per_is_weights(self._p, B, self._tree, self._min_p[1:2], self._beta,
                                  self.beta_increment_per_sampling, self._w)
        if sampled == 3:
            native.window_gather_pad_w(self._gather_keys, self._ids, B, self.prev_n, self.post_n, C, self._index_ring(),
                                       self._p, self._tree, self._beta, self.beta_increment_per_sampling, self._w, self._min_p)
        elif sampled != 2:
            native.window_gather_pad(self._gather_keys, self._ids, B, self.prev_n, self.post_n, C, self._index_ring())

    # ------------------------------------------------------------------------------------------
    # priority / transition write-backs
    # ------------------------------------------------------------------------------------------
    def _update_ids(self, ids: torch.Tensor, td: torch.Tensor, stale_check=True, mode=0, sidecars=None) -> None:
        k = ids.numel()
        if k == 0:
            return
        # the election scratch is allocated once (a captured hipGraph holds its address): longer updates go in
        # pieces, which is equivalent (later writers win, piece after piece)
        piece = (self._winner.numel() - self.capacity) // 2
        for s in range(0, k, piece):
            native.sumtree_update(self._tree, self.capacity, ids[s:s + piece], self._slot_ids if stale_check else None,
                                  td[s:s + piece], self.alpha, self.td_error_min, self.td_error_max, mode,
                                  self._winner, self._nan_flag, sidecars=sidecars if s == 0 else None)

    def update(self, data_ids, td_error, sidecars=None) -> None:
        """priority <- clip(td, min, max)^alpha for ids still resident (replay_buffer.py:412-427).  `sidecars`: small
        jobs of the caller's step that ride as extra workgroups of the update launch (`native.Sidecar`)."""
        ids = data_ids if isinstance(data_ids, torch.Tensor) else self._to_device(np.asarray(data_ids, np.int64))
        td = td_error if isinstance(td_error, torch.Tensor) else self._to_device(np.asarray(td_error, np.float32))
        if self.sharded is not None and ids is self._ids:      # the step's batch: priorities go to the owning shards
            assert not sidecars
            self.sharded.update(td.reshape(-1))
            return
        with torch.cuda.device(self.device):
            self._update_ids(ids.reshape(-1), td.reshape(-1).contiguous(), stale_check=True, sidecars=sidecars)

    def td_update_ok(self, ids, n: int) -> bool:
        return (self.sharded is None and isinstance(ids, torch.Tensor) and ids.dim() == 1 and ids.is_contiguous()
                and native.td_update_ok(ids.numel(), n))

    def td_update(self, vtrace_args, ids: torch.Tensor, sidecars=None, alpha_step=None) -> None:
        """`update(ids, td)` where td is the TD error `vtrace_args` describes (native.VtraceArgs with q_online and
        td_error_out set): return and priority update in one launch (reference sac_base.py:2182-2245 +
        replay_buffer.py:412-427).  `alpha_step`: the temperature step to run first, if it is still pending."""
        with torch.cuda.device(self.device):
            native.td_update(vtrace_args, self._tree, self.capacity, ids, self._slot_ids, self.alpha, self.td_error_min,
                             self.td_error_max, self._winner, self._nan_flag, sidecars=sidecars, alpha_step=alpha_step)

    def update_transitions(self, data_ids, key: str, data) -> None:
        """Overwrite `key` rows for ids still resident (replay_buffer.py:429-434); later duplicates
        win, like NumPy fancy assignment."""
        ids = data_ids if isinstance(data_ids, torch.Tensor) else self._to_device(np.asarray(data_ids, np.int64))
        rows = (data if isinstance(data, torch.Tensor) else self._to_device(data)).contiguous()
        k = ids.numel()
        if k == 0:
            return
        col = self._columns[key]
        row_bytes = col[0].numel() * col.element_size()
        assert rows.dtype == col.dtype and rows.numel() * rows.element_size() == k * row_bytes
        with torch.cuda.device(self.device):
            native.scatter_rows_if_id_match(col, row_bytes, self.capacity, ids.reshape(-1).contiguous(), k, 0, 1,
                                            self._slot_ids, None, 0, rows, row_bytes, row_bytes, self._winner_rows)

    def update_window_transitions(self, sample_ids: torch.Tensor, first_off: int, count: int,
                                  padding_mask: torch.Tensor, key: str, rows: torch.Tensor, side=False) -> None:
        """Fused form used by SAC_Base.train: rows[s, j] -> id = sample_ids[s] + first_off + j for
        j < count, skipping padded positions and overwritten slots (reference sac_base.py:2589-2605
        builds those id lists on the host).  `rows` is [B, >=count, *shape] (a view is fine)."""
        col = self._columns[key]
        row_bytes = col[0].numel() * col.element_size()
        if row_bytes == 0:
            return
        assert rows.dtype == col.dtype
        if self.sharded is not None and sample_ids is self._ids:
            # (target j of a sample is masked by padding_mask[:, j], whatever first_off is: sac_base.py:2589-2605)
            self.sharded.update_windows(first_off, count, padding_mask[:, :count], key, rows[:, :count])
            return
        es = rows.element_size()
        scratch = self._winner_rows
        if side:    # concurrent with a write-back on another stream: two elections must not share their map
            if self._winner_rows_side is None:
                self._winner_rows_side = torch.full_like(self._winner_rows, -1)
            scratch = self._winner_rows_side
        native.scatter_rows_if_id_match(col, row_bytes, self.capacity, sample_ids, sample_ids.numel(),
                                        first_off, count, self._slot_ids, padding_mask,
                                        padding_mask.stride(0), rows, rows.stride(0) * es,
                                        rows.stride(1) * es, scratch)

    def window_scatter_sidecars(self, sample_ids: torch.Tensor, first_off: int, count: int,
                                padding_mask: torch.Tensor, key: str, rows: torch.Tensor, side: bool = False):
        """`update_window_transitions` as two sidecar jobs (`native.Sidecar`: elect, write) for launches that
        already sit on the step's critical path; the write job must ride a LATER launch than the elect job.
        `side`: the second election scratch (two write-backs in flight at once must not share their map)."""
        col = self._columns[key]
        row_bytes = col[0].numel() * col.element_size()
        assert row_bytes > 0 and rows.dtype == col.dtype
        es = rows.element_size()
        scratch = self._winner_rows
        if side:
            if self._winner_rows_side is None:
                self._winner_rows_side = torch.full_like(self._winner_rows, -1)
            scratch = self._winner_rows_side
        args = (col, row_bytes, self.capacity, sample_ids, sample_ids.numel(), first_off, count, self._slot_ids,
                padding_mask, padding_mask.stride(0), rows, rows.stride(0) * es, rows.stride(1) * es, scratch)
        return (native.sidecar_scatter(native.SIDECAR_SCATTER_ELECT, *args),
                native.sidecar_scatter(native.SIDECAR_SCATTER_WRITE, *args))

    # ------------------------------------------------------------------------------------------
    # random access (used by the option-critic variant) and bookkeeping
    # ------------------------------------------------------------------------------------------
    def get_curr_id(self) -> int:
        return self._next_id % self.capacity

    def get_storage_data(self, data_ids) -> dict:
        """Rows at `data_ids % C` for every key, without residency check (replay_buffer.py:401-406): one gather
        launch for all keys (the option-critic variant calls this once per key-transition hop,
        oc/option_selector_base.py:2205, 2223).  Any transition keys are carried (`option_index`,
        `option_changed_index`, `pre_low_seq_hidden_state`, ...): the rings are created from whatever `add` is given."""
        ids = data_ids if isinstance(data_ids, torch.Tensor) else self._to_device(np.asarray(data_ids, np.int64))
        ids = ids.reshape(-1).to(torch.int64).contiguous()
        k = ids.numel()
        out, specs = {}, []
        for key, col in self._columns.items():
            out[key] = torch.empty((k, *col.shape[1:]), dtype=col.dtype, device=self.device)
            row_bytes = col[0].numel() * col.element_size()
            if row_bytes and k:
                specs.append(dict(src=col, dst=out[key], row_bytes=row_bytes, pad_mode=native.PAD_KEEP))
        with torch.cuda.device(self.device):
            for s0 in range(0, len(specs), native.MAX_GATHER_KEYS):
                native.gather_rows(native.make_gather_keys(specs[s0:s0 + native.MAX_GATHER_KEYS]), ids, self.capacity)
        return out

    def get_storage_data_ids(self, data_ids):
        ids = data_ids if isinstance(data_ids, torch.Tensor) else self._to_device(np.asarray(data_ids, np.int64))
        return self._slot_ids.index_select(0, torch.remainder(ids, self.capacity))

    def check_health(self) -> None:
        """Synchronising: raise the reference's NaN error if an update kernel flagged one."""
        if int(self._nan_flag.item()) != 0:
            self._logger.error('td_error has nan')
            raise Exception('td_error has nan')

    def check_tree_invariant(self) -> int:
        """Debug: number of internal nodes with node != left + right (0 for a healthy tree)."""
        out = torch.zeros(1, dtype=torch.int32, device=self.device)
        native.sumtree_check(self._tree, self.capacity, out)
        return int(out.item())

    # on-disk format of the reference: `<ckpt>-rb_tree.npy`, `<ckpt>-rb_storage.npz`
    def save(self, save_dir: Path, ckpt: int) -> None:
        self.check_health()
        save_dir = Path(save_dir)
        np.save(save_dir.joinpath(f'{ckpt}-rb_tree.npy'), self._tree.cpu().numpy())
        cols = {k: v.cpu().numpy() for k, v in (self._columns or {}).items()}
        np.savez(save_dir.joinpath(f'{ckpt}-rb_storage.npz'), _id=self._slot_ids.cpu().numpy(), **cols,
                 p_size=self._size, p_id=self._next_id)

    def load(self, save_dir: Path, ckpt: int) -> None:
        save_dir = Path(save_dir)
        tree_path = save_dir.joinpath(f'{ckpt}-rb_tree.npy')
        if tree_path.exists():
            tree = np.load(tree_path)
            assert tree.shape[0] == 2 * self.capacity - 1, 'replay capacity differs from the checkpoint'
            self._tree.copy_(torch.from_numpy(tree))
        storage_path = save_dir.joinpath(f'{ckpt}-rb_storage.npz')
        if storage_path.exists():
            saved = np.load(storage_path)
            self._size, self._next_id = int(saved['p_size']), int(saved['p_id'])
            self._columns = {}
            for k in saved.files:
                if k in ('p_size', 'p_id'):
                    continue
                if k == '_id':
                    self._slot_ids.copy_(torch.from_numpy(saved[k]))
                else:
                    self._columns[k] = torch.from_numpy(saved[k]).to(self.device)
            self._batch, self._gather_keys = None, None

    def clear(self) -> None:
        self._tree.zero_()
        self._slot_ids.zero_()
        self._columns, self._batch, self._gather_keys = None, None, None
        self._size, self._next_id = 0, 0

    def copy(self, src: 'PrioritizedReplayBuffer') -> None:
        self._tree.copy_(src._tree)
        self._slot_ids.copy_(src._slot_ids)
        self._columns = {k: v.to(self.device, copy=True) for k, v in (src._columns or {}).items()}
        self._batch, self._gather_keys = None, None
        self._size, self._next_id = src._size, src._next_id

    @property
    def size(self) -> int:
        return self._size

    @property
    def is_full(self) -> bool:
        return self._size == self.capacity

    @property
    def is_lg_batch_size(self) -> bool:
        return self._size > self.batch_size

    def close(self):
        if self._closed:
            return
        self._closed = True
        try:
            self.check_health()
        finally:
            self._columns, self._batch, self._gather_keys = None, None, None
