"""Device-resident agent-side episode assembly: the drop-in for the reference's `Agent`, `AgentManager` and
`MultiAgentsManager` (reference `algorithm/agent.py:21-890`), SURVEY.md §8f rank 2.

The reference keeps every live agent's running episode as a dict of NumPy arrays, moves one transition per
agent and environment step through Python (`set_tmp_obs_action` 87-97, `_add_transition` 191-235), stacks the
agents' pending actions / hidden states on the host for `choose_action` (474-485), copies the policy's outputs
back to the host, and finally ships the finished episode host -> device in `put_episode`.  Here

  * every transition key is ONE slab `[slots, max_episode_length + 1, *shape]` in HBM (`EpisodeSlab`), an agent
    is a slot of it plus a handful of host scalars (cursor, pending index, the episode statistics the training
    loop reads); pending values (`_tmp_*` of the reference) are per-slot rows `[slots, *shape]`;
  * an environment step is three launches of one row-mover kernel (`asac_rows_move`, csrc/episode.hip) over
    all agents and all keys — commit the pending transitions with this step's rewards, collect the policy's
    inputs, stage its outputs — around `SAC_Base.choose_action_device`; the only host <-> device traffic is
    what the environment itself produces and consumes (observations + rewards in, actions out);
  * the attention agents' episode window (`get_episode_trans(force_length)` 258-316 and the concatenations of
    `get_action` 536-552) is gathered from the slabs by the same kernel, restricted to the `burn_in_step`
    positions `choose_attn_action` keeps (sac_base.py:1049-1053);
  * a finished episode is handed to `SAC_Base.put_episode` as device tensors: slab -> replay ring inside HBM.

Semantics kept from the reference, quirks included: the terminal row repeats the last transition's
pre-hidden-state (agent.py:172); an "empty" first episode (<= NON_EMPTY_STEPS steps) resets the counters but
keeps the pending transition and the running index (143-150); a full buffer drops its oldest row (205-214);
`ep_dones = done & ~max_reached` (296-298); liveness counting and zombie agents (440-472).

`get_tmp_episode_trans_list()` returns the reference's dicts with device tensors as values
(`episode_trans_to_numpy` converts one for host consumers).  There is no CPU fallback: the slabs need the HIP
library (`asac_amd.native`).
"""
import json
import logging
from copy import deepcopy
from pathlib import Path
from typing import Iterator

import numpy as np
import torch

from asac_amd import native

from .sac_base import SAC_Base
from .utils.enums import SEQ_ENCODER
from .utils.operators import ma_name2path_name

AGENT_MAX_LIVENESS = 20
NON_EMPTY_STEPS = 2
DEFAULT_MAX_EPISODE_LENGTH = 2000

_TORCH_DTYPE = {np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64,
                np.dtype(np.uint8): torch.uint8, np.dtype(np.bool_): torch.bool,
                np.dtype(np.int32): torch.int32, np.dtype(np.int64): torch.int64,
                np.dtype(np.float16): torch.float16}


def episode_trans_to_numpy(episode_trans: dict) -> dict:
    """an episode dict of this module (device tensors) as the reference's NumPy dict"""
    out = {}
    for k, v in episode_trans.items():
        out[k] = [o.cpu().numpy() for o in v] if isinstance(v, list) else v.cpu().numpy()
    return out


class EpisodeSlab:
    """HBM storage of the running episodes of up to `slots` agents (grown by doubling)."""

    SCALARS = ('index', 'reward', 'done', 'max_reached')

    def __init__(self, obs_shapes, obs_dtypes, action_size, seq_hidden_state_shape, max_episode_length, device,
                 padding_action: np.ndarray, slots: int = 8):
        native.load()
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise native.AsacNativeError('EpisodeSlab is HBM-resident: it needs a cuda (ROCm) device; the reference\'s '
                                         'host path is restated in oracle/ for tests only')
        self.M = int(max_episode_length)
        self.rows = self.M + 1           # one spare row: the attention agents' open position
        self.n_obs = len(obs_shapes)
        self.action_size = int(action_size)
        self.hidden_shape = tuple(seq_hidden_state_shape)
        self._spec = {'index': ((), torch.int32)}
        for i, (s, d) in enumerate(zip(obs_shapes, obs_dtypes)):
            self._spec[f'obs_{i}'] = (tuple(s), _TORCH_DTYPE[np.dtype(d)])
        self._spec.update({'action': ((self.action_size,), torch.float32), 'reward': ((), torch.float32),
                           'done': ((), torch.bool), 'max_reached': ((), torch.bool),
                           'prob': ((self.action_size,), torch.float32),
                           'pre_seq_hidden_state': (self.hidden_shape, torch.float32)})
        # pending rows (reference `_tmp_obs_list`, `_tmp_action`, `_tmp_prob`, `_tmp_pre_seq_hidden_state`,
        # `_tmp_seq_hidden_state`)
        self._pending_spec = {k: self._spec[k] for k in self._spec if k not in self.SCALARS}
        self._pending_spec['seq_hidden_state'] = (self.hidden_shape, torch.float32)
        self._row_bytes_of = {k: int(np.prod(shape, dtype=np.int64)) * torch.empty((), dtype=dtype).element_size()
                              for k, (shape, dtype) in (self._spec | self._pending_spec).items()}
        self.padding_action = torch.from_numpy(np.ascontiguousarray(padding_action, dtype=np.float32)).to(self.device)
        self.ones_prob = torch.ones(self.action_size, dtype=torch.float32, device=self.device)
        self.slots = 0
        self.slab: dict[str, torch.Tensor] = {}
        self.pending: dict[str, torch.Tensor] = {}
        self._free: list[int] = []
        self._grow(slots)

    # -- storage ---------------------------------------------------------------------------------------------
    def _row_bytes(self, key) -> int:
        return self._row_bytes_of[key]

    def _grow(self, slots: int) -> None:
        old, self.slots = self.slots, slots
        for k, (shape, dtype) in self._spec.items():
            new = torch.zeros((slots, self.rows, *shape), dtype=dtype, device=self.device)
            if old:
                new[:old] = self.slab[k]
            self.slab[k] = new
        for k, (shape, dtype) in self._pending_spec.items():
            new = torch.zeros((slots, *shape), dtype=dtype, device=self.device)
            if old:
                new[:old] = self.pending[k]
            self.pending[k] = new
        self.pending['action'][old:] = self.padding_action
        self._free.extend(range(slots - 1, old - 1, -1))

    def acquire(self) -> int:
        if not self._free:
            self._grow(self.slots * 2)
        slot = self._free.pop()
        self.clear_pending(slot)
        return slot

    def release(self, slot: int) -> None:
        self._free.append(slot)

    def clear_pending(self, slot: int) -> None:
        """`_tmp_* = None` of the reference: what `get_tmp_action` / `get_tmp_seq_hidden_state` then return"""
        self.pending['action'][slot] = self.padding_action
        self.pending['seq_hidden_state'][slot].zero_()
        self.pending['pre_seq_hidden_state'][slot].zero_()

    # -- per-step scalars ------------------------------------------------------------------------------------
    def upload(self, columns: list[np.ndarray]) -> list[torch.Tensor]:
        """a few i32 / f32 / bool host columns of equal length -> device in ONE copy (every column 4 B per item);
        the pinned staging block comes from torch's caching host allocator, which does not hand it out again
        before the asynchronous copy has completed"""
        n = len(columns[0])
        host = torch.empty((len(columns), n), dtype=torch.int32, pin_memory=True)
        for r, c in enumerate(columns):
            c = np.asarray(c)
            if c.dtype == np.float32:
                host[r].view(torch.float32).copy_(torch.from_numpy(np.ascontiguousarray(c)))
            else:
                host[r].copy_(torch.from_numpy(np.ascontiguousarray(c, dtype=np.int32)))
        dev = host.to(self.device, non_blocking=True)
        return [dev[r] for r in range(len(columns))]

    # -- movements -------------------------------------------------------------------------------------------
    def _slab_strides(self, key):
        t = self.slab[key]
        return t.stride(0) * t.element_size(), t.stride(1) * t.element_size()

    def commit(self, slots, dst_rows, index, reward, done, max_reached) -> None:
        """pending rows + this step's scalars -> slab[slot, dst_row]   (`Agent._add_transition`)"""
        n = len(slots)
        if n == 0:
            return
        d_slot, d_row, d_index, d_reward, d_done, d_max = self.upload(
            [slots, dst_rows, index, np.asarray(reward, dtype=np.float32), np.asarray(done, dtype=np.int32),
             np.asarray(max_reached, dtype=np.int32)])
        specs = []
        # the flags travel as i32 columns; the slab keeps one byte per flag: copy the low byte of every word
        for k, src in (('index', d_index), ('reward', d_reward), ('done', d_done), ('max_reached', d_max)):
            s0, s1 = self._slab_strides(k)
            specs.append(dict(src=src, dst=self.slab[k], row_bytes=self._row_bytes(k), src_mode=native.ROW_ITEM,
                              src_stride0=4, dst_mode=native.ROW_SLOT_ROW, dst_stride0=s0, dst_stride1=s1))
        for k in self._pending_spec:
            rb = self._row_bytes(k)
            if k == 'seq_hidden_state' or rb == 0:
                continue
            s0, s1 = self._slab_strides(k)
            specs.append(dict(src=self.pending[k], dst=self.slab[k], row_bytes=rb, src_mode=native.ROW_SLOT,
                              src_stride0=rb, dst_mode=native.ROW_SLOT_ROW, dst_stride0=s0, dst_stride1=s1))
        native.rows_move(native.make_row_moves(specs), d_slot, None, d_row, n)

    def append_terminal(self, slot: int, row: int, index: int, next_obs_list) -> None:
        """the closing row of an episode (`Agent._end_episode`, agent.py:161-173): next observation, padding action,
        reward 0, done and max_reached set, probability 1, the LAST transition's pre-hidden-state"""
        s = self.slab
        s['index'][slot, row] = index
        for i, o in enumerate(next_obs_list):
            s[f'obs_{i}'][slot, row] = o
        s['action'][slot, row] = self.padding_action
        s['reward'][slot, row] = 0.
        s['done'][slot, row] = True
        s['max_reached'][slot, row] = True
        s['prob'][slot, row] = self.ones_prob
        s['pre_seq_hidden_state'][slot, row] = self.pending['pre_seq_hidden_state'][slot]

    def collect(self, d_slot, n):
        """pending action / hidden state of the listed slots -> dense [n, ...] (`_get_merged_action` / `_seq_hidden_state`)"""
        pre_action = torch.empty((n, self.action_size), dtype=torch.float32, device=self.device)
        hidden = torch.empty((n, *self.hidden_shape), dtype=torch.float32, device=self.device)
        specs = []
        for k, dst in (('action', pre_action), ('seq_hidden_state', hidden)):
            rb = self._row_bytes(k)
            if rb == 0:
                continue
            specs.append(dict(src=self.pending[k], dst=dst, row_bytes=rb, src_mode=native.ROW_SLOT, src_stride0=rb,
                              dst_mode=native.ROW_ITEM, dst_stride0=rb))
        if specs:
            native.rows_move(native.make_row_moves(specs), d_slot, None, None, n)
        return pre_action, hidden

    def stage(self, d_slot, n, obs_list, action, prob, pre_hidden, hidden) -> None:
        """the policy's outputs and the observations they answer -> pending rows (`Agent.set_tmp_obs_action`);
        `pre_hidden` is the state the step started from (= the previous pending `seq_hidden_state`)"""
        srcs = {f'obs_{i}': o for i, o in enumerate(obs_list)}
        srcs.update(action=action, prob=prob, pre_seq_hidden_state=pre_hidden, seq_hidden_state=hidden)
        specs = []
        for k, src in srcs.items():
            rb = self._row_bytes(k)
            if rb == 0:
                continue
            src = src.contiguous()
            srcs[k] = src     # keep alive until the launch is issued
            specs.append(dict(src=src, dst=self.pending[k], row_bytes=rb, src_mode=native.ROW_ITEM, src_stride0=rb,
                              dst_mode=native.ROW_SLOT, dst_stride0=rb))
        native.rows_move(native.make_row_moves(specs), d_slot, None, None, n)

    def window(self, slots, cursors, ep_length, width, open_index, obs_list):
        """Attention agents: the last `width` positions of the sequence the reference builds in
        `AgentManager.get_action` (agent.py:536-552) — every agent's last `ep_length` rows left-padded, then the
        position being decided — as device tensors [n, W, ...]; `ep_pre_attn_states` has no entry for the open
        position, so its window covers the last min(width, ep_length) rows."""
        n = len(slots)
        slots = np.asarray(slots, dtype=np.int32)
        cursors = np.asarray(cursors, dtype=np.int32)
        # the open position becomes slab row `cursor` (scratch: the next commit overwrites it)
        d_slot, d_cur, d_open = self.upload([slots, cursors, np.asarray(open_index, dtype=np.int32)])
        specs = []
        srcs = {'index': d_open, **{f'obs_{i}': o.contiguous() for i, o in enumerate(obs_list)}}
        for k, src in srcs.items():
            s0, s1 = self._slab_strides(k)
            rb = self._row_bytes(k)
            specs.append(dict(src=src, dst=self.slab[k], row_bytes=rb, src_mode=native.ROW_ITEM, src_stride0=rb,
                              dst_mode=native.ROW_SLOT_ROW, dst_stride0=s0, dst_stride1=s1))
        native.rows_move(native.make_row_moves(specs), d_slot, None, d_cur, n)

        W = min(width, ep_length + 1)
        Wh = min(width, ep_length)
        # sequence position s in [ep_length + 1 - W, ep_length] -> slab row cursor - ep_length + s
        pos = np.arange(ep_length + 1 - W, ep_length + 1, dtype=np.int32)
        rows = (cursors[:, None] - ep_length + pos[None, :]).astype(np.int32).reshape(-1)
        item_slot = np.repeat(slots, W)
        d_islot, d_rows = self.upload([item_slot, rows])
        out = {}
        specs = []
        for k in ('index', *[f'obs_{i}' for i in range(self.n_obs)], 'action'):
            shape, dtype = self._spec[k]
            dst = torch.empty((n, W, *shape), dtype=dtype, device=self.device)
            out[k] = dst
            s0, s1 = self._slab_strides(k)
            rb = self._row_bytes(k)
            # pre-actions: position s carries the action of position s - 1, zeros in front (gen_n_pre_actions)
            specs.append(dict(src=self.slab[k], dst=dst, row_bytes=rb, src_mode=native.ROW_SLOT_ROW, src_stride0=s0,
                              src_stride1=s1, dst_mode=native.ROW_ITEM, dst_stride0=rb,
                              src_row_offset=-1 if k == 'action' else 0,
                              pad_word=0xffffffff if k == 'index' else 0))
        native.rows_move(native.make_row_moves(specs), d_islot, d_rows, None, n * W)
        hidden = torch.empty((n, Wh, *self.hidden_shape), dtype=torch.float32, device=self.device)
        rb = self._row_bytes('pre_seq_hidden_state')
        if Wh > 0 and rb > 0:
            pos_h = np.arange(ep_length - Wh, ep_length, dtype=np.int32)
            rows_h = (cursors[:, None] - ep_length + pos_h[None, :]).astype(np.int32).reshape(-1)
            d_hslot, d_hrows = self.upload([np.repeat(slots, Wh), rows_h])
            s0, s1 = self._slab_strides('pre_seq_hidden_state')
            native.rows_move(native.make_row_moves([dict(
                src=self.slab['pre_seq_hidden_state'], dst=hidden, row_bytes=rb, src_mode=native.ROW_SLOT_ROW,
                src_stride0=s0, src_stride1=s1, dst_mode=native.ROW_ITEM, dst_stride0=rb)]), d_hslot, d_hrows, None,
                n * Wh)
        return out, hidden

    def shift_left(self, slot: int) -> None:
        """a full episode buffer drops its oldest row (agent.py:205-214)"""
        for t in self.slab.values():
            t[slot, :self.M - 1] = t[slot, 1:self.M].clone()

    def move_row(self, slot: int, src_row: int, dst_row: int) -> None:
        for t in self.slab.values():
            t[slot, dst_row] = t[slot, src_row]

    def episode(self, slot: int, length: int) -> dict:
        """`Agent.get_episode_trans()` of a finished episode: copies, so the slot can start its next episode"""
        s = self.slab
        take = lambda k: s[k][slot, :length].clone().unsqueeze(0)  # noqa: E731
        done = s['done'][slot, :length] & ~s['max_reached'][slot, :length]
        return {'ep_indexes': take('index'),
                'ep_obses_list': [take(f'obs_{i}') for i in range(self.n_obs)],
                'ep_actions': take('action'),
                'ep_rewards': take('reward'),
                'ep_dones': done.unsqueeze(0),
                'ep_probs': take('prob'),
                'ep_pre_seq_hidden_states': take('pre_seq_hidden_state')}


class Agent:
    """Host view of one agent: the scalars of the reference's `Agent` (agent.py:21-331); its arrays live in the
    manager's `EpisodeSlab` under `slot`."""
    reward = 0
    steps = 0
    done = False
    max_reached = False
    force_terminated = False
    hit_reward: int | None = None
    hit = 0
    current_reward = 0
    current_step = 0

    def __init__(self, agent_id: int, slab: EpisodeSlab, slot: int, max_episode_length: int,
                 hit_reward: int | None = None):
        self.agent_id = agent_id
        self.slab = slab
        self.slot = slot
        self.max_episode_length = max_episode_length
        self.hit_reward = hit_reward
        self._tmp_index = -1
        self._has_tmp = False
        self._logger = logging.getLogger(f'agent.{agent_id}')

    @property
    def episode_length(self) -> int:
        return self.current_step

    @property
    def is_empty(self) -> bool:
        return self.steps <= NON_EMPTY_STEPS or self.force_terminated

    def get_tmp_index(self) -> int:
        return self._tmp_index

    def _next_row(self) -> int:
        """row a new transition goes to (`_add_transition`: a full buffer first drops its oldest row)"""
        if self.current_step == self.max_episode_length:
            self._logger.warning(f'_tmp_episode_trans is full {self.max_episode_length}')
            self.slab.shift_left(self.slot)
            self.current_step -= 1
        row = self.current_step
        self.current_step += 1
        return row

    def account(self, reward: float) -> None:
        """the bookkeeping of `end_transition` after the row was added (agent.py:135-141)"""
        self.current_reward += reward
        if not self.done:
            self.steps += 1
            self.reward += reward
            if self.hit_reward is not None and reward >= self.hit_reward:
                self.hit += 1

    def force_done(self) -> None:
        self.done = True
        self.max_reached = True
        self.force_terminated = True

    def reset(self) -> None:
        self.reward = self.current_reward
        self.steps = 0
        self.done = False
        self.max_reached = False
        self.hit = 0


class AgentManager:
    def __init__(self, name: str, obs_names: list[str], obs_shapes: list[tuple[int]], obs_dtypes: list[np.dtype],
                 d_action_sizes: list[int], c_action_size: int, max_episode_length: int = -1,
                 hit_reward: int | None = None):
        self.name = name
        self.obs_names = obs_names
        self.obs_shapes = obs_shapes
        self.obs_dtypes = obs_dtypes
        self.d_action_sizes = d_action_sizes
        self.d_action_summed_size = sum(d_action_sizes)
        self.c_action_size = c_action_size
        self.action_size = self.d_action_summed_size + self.c_action_size
        self.max_episode_length = max_episode_length
        self.hit_reward = hit_reward

        self.agents_dict: dict[int, Agent] = {}
        self.agents_liveness: dict[int, int] = {}

        self.rl: SAC_Base | None = None
        self.seq_encoder = None
        self.il = None
        self.slab: EpisodeSlab | None = None

        self._logger = logging.getLogger(f'agent_mgr.{name}')
        self._tmp_episode_trans_list = []
        self._data = {}

    # -- per-manager scratch the environment loops hang on the manager (`mgr['key']`), and views over its agents
    #    (reference agent.py:372-400: same names, same meaning)
    def __getitem__(self, key: str):
        return self._data[key]

    def __setitem__(self, key: str, value):
        self._data[key] = value

    def _agents_where(self, empty: bool | None = None) -> list[Agent]:
        members = self.agents_dict.values()
        if empty is None:
            return [*members]
        return [ag for ag in members if bool(ag.is_empty) == empty]

    agents = property(lambda self: self._agents_where())
    non_empty_agents = property(lambda self: self._agents_where(empty=False))
    empty_agents = property(lambda self: self._agents_where(empty=True))

    @property
    def done(self) -> bool:
        """every agent of the manager has finished its episode"""
        for ag in self.agents_dict.values():
            if not ag.done:
                return False
        return True

    @property
    def max_reached(self) -> bool:
        """some running (non-empty) agent has hit its step limit"""
        for ag in self._agents_where(empty=False):
            if ag.max_reached:
                return True
        return False

    def set_config(self, config) -> None:
        self.config = deepcopy(config)

    def set_model_abs_dir(self, model_abs_dir: Path) -> None:
        self.model_abs_dir = model_abs_dir
        self.model_abs_dir.mkdir(parents=True, exist_ok=True)

    def set_rl(self, rl: SAC_Base) -> None:
        self.rl = rl
        self.seq_encoder = rl.seq_encoder
        self._ensure_slab()

    def set_il(self, il) -> None:
        self.il = il

    def _ensure_slab(self, device=None) -> None:
        if self.slab is not None:
            return
        hidden_shape = tuple(self.rl.seq_hidden_state_shape) if self.rl is not None else (0,)
        device = self.rl.device if self.rl is not None else (device or torch.device('cuda', torch.cuda.current_device()))
        d_action_list = [np.eye(s, dtype=np.float32)[0] for s in self.d_action_sizes]
        padding_action = np.concatenate(d_action_list + [np.zeros(self.c_action_size, dtype=np.float32)], axis=-1)
        M = self.max_episode_length if self.max_episode_length != -1 else DEFAULT_MAX_EPISODE_LENGTH
        self._M = M
        self.slab = EpisodeSlab(self.obs_shapes, self.obs_dtypes, self.action_size, hidden_shape, M, device,
                                padding_action)

    # -- agent population (agent.py:410-472) -----------------------------------------------------------------
    def reset(self) -> None:
        for a in self.agents:
            self.slab.release(a.slot)
        self.agents_dict.clear()
        self.agents_liveness.clear()
        self.clear_tmp_episode_trans_list()

    def reset_dead_agents(self) -> None:
        dead_agent_ids = {agent_id for agent_id, liveness in self.agents_liveness.items() if liveness <= 0}
        dead_agent_ids.union({agent.agent_id for agent in self.empty_agents})   # (the reference discards this union)
        for agent_id in dead_agent_ids:
            self.slab.release(self.agents_dict[agent_id].slot)
            del self.agents_dict[agent_id]
            del self.agents_liveness[agent_id]

    def reset_and_continue(self) -> None:
        self.reset_dead_agents()
        for agent in self.agents:
            agent.reset()
        self.clear_tmp_episode_trans_list()

    def _verify_agents(self, agent_ids: np.ndarray):
        assert self.rl is not None or self.slab is not None
        for agent_id in self.agents_liveness:
            self.agents_liveness[agent_id] -= 1
        for agent_id in agent_ids:
            if agent_id not in self.agents_dict:
                self.agents_dict[agent_id] = Agent(agent_id, self.slab, self.slab.acquire(), self._M, self.hit_reward)
            self.agents_liveness[agent_id] = AGENT_MAX_LIVENESS
        for agent_id in self.agents_liveness:
            agent = self.agents_dict[agent_id]
            if self.agents_liveness[agent_id] <= 0 and not agent.done:
                agent.force_done()

    # -- transitions -----------------------------------------------------------------------------------------
    def _end_transitions(self, agents: list[Agent], rewards, done: bool, max_reached=None, force_terminated=False,
                         next_obs_list=None):
        """`Agent.end_transition` (agent.py:117-159) for a list of agents: ONE commit launch for their pending
        transitions, then the per-agent episode endings."""
        live = [(i, a) for i, a in enumerate(agents) if a._has_tmp]
        if not live:
            return
        rows = [a._next_row() for _, a in live]
        rew = np.asarray([rewards[i] for i, _ in live], dtype=np.float32)
        mx = np.asarray([bool(max_reached[i]) if max_reached is not None else False for i, _ in live])
        self.slab.commit(np.asarray([a.slot for _, a in live], dtype=np.int32), np.asarray(rows, dtype=np.int32),
                         np.asarray([a._tmp_index for _, a in live], dtype=np.int32), rew,
                         np.full(len(live), done), mx)
        for (i, a), r, m in zip(live, rew, mx):
            a.account(rewards[i])
            if not done:
                continue
            if not a.done and a.is_empty and not force_terminated:
                # an "empty" first episode: counters restart, the pending transition and the running index stay
                a.steps = 0
                a.reward = 0
                a.hit = 0
                a.current_step = 0
                a.current_reward = 0
                continue
            if not a.done:
                a.done = True
                a.max_reached = bool(m)
                a.force_terminated = force_terminated
            ep = self._end_episode(a, [o[i] for o in next_obs_list] if next_obs_list is not None else None)
            if ep is not None:
                yield ep

    def _end_episode(self, a: Agent, next_obs):
        slab = self.slab
        row = a._next_row()
        if next_obs is None:
            next_obs = [torch.zeros(s, dtype=slab.slab[f'obs_{i}'].dtype, device=slab.device)
                        for i, s in enumerate(self.obs_shapes)]
        slab.append_terminal(a.slot, row, a._tmp_index + 1, next_obs)
        ep = slab.episode(a.slot, a.episode_length) if a.episode_length > 1 else None
        a.current_reward = 0
        a.current_step = 0
        a._tmp_index = -1
        a._has_tmp = False
        slab.clear_pending(a.slot)
        return ep

    def _obs_to_device(self, obs_list):
        return [o if isinstance(o, torch.Tensor) else
                torch.from_numpy(np.ascontiguousarray(o)).to(self.slab.device, non_blocking=True) for o in obs_list]

    def get_action(self, agent_ids: np.ndarray, obs_list: list[np.ndarray], last_reward: np.ndarray,
                   offline_action: np.ndarray | None = None, disable_sample: bool = False,
                   force_rnd_if_available: bool = False):
        assert len(agent_ids) == obs_list[0].shape[0]
        if self.rl is None:
            return self.get_test_action(agent_ids=agent_ids, obs_list=obs_list)
        self._verify_agents(agent_ids)
        agents = [self.agents_dict[i] for i in agent_ids]
        for _ in self._end_transitions(agents, last_reward, done=False):
            pass
        slab, n = self.slab, len(agents)
        obs = self._obs_to_device(obs_list)
        off = torch.from_numpy(offline_action).to(slab.device) if offline_action is not None else None
        slots = np.asarray([a.slot for a in agents], dtype=np.int32)

        if self.seq_encoder in (None, SEQ_ENCODER.RNN):
            d_slot, = slab.upload([slots])
            pre_action, pre_hidden = slab.collect(d_slot, n)
            action, prob, hidden = self.rl.choose_action_device(
                obs, pre_action, pre_hidden, off, disable_sample, force_rnd_if_available)
        else:
            ep_length = min(512, max(a.episode_length for a in agents))
            cursors = [a.episode_length for a in agents]
            # agent.py:545-548: the open position's index is the last one + 1 (0 for an agent without history:
            # its window is all padding, -1 + 1)
            open_index = [(a._tmp_index + 1) if a.episode_length > 0 else 0 for a in agents]
            win, attn_states = slab.window(slots, cursors, ep_length, self.rl.burn_in_step, open_index, obs)
            d_slot, = slab.upload([slots])
            _, pre_hidden = slab.collect(d_slot, n)
            action, prob, hidden = self.rl.choose_attn_action_device(
                win['index'], win['index'] == -1, [win[f'obs_{i}'] for i in range(slab.n_obs)], win['action'],
                attn_states, off, disable_sample, force_rnd_if_available)

        slab.stage(d_slot, n, obs, action, prob, pre_hidden, hidden)
        for a in agents:
            a._tmp_index += 1
            a._has_tmp = True
        action = action.cpu().numpy()
        return action[..., :self.d_action_summed_size], action[..., self.d_action_summed_size:]

    def get_test_action(self, agent_ids: np.ndarray, obs_list: list[np.ndarray]):
        """Random actions without a learner (agent.py:576-618)."""
        assert len(agent_ids) == obs_list[0].shape[0]
        self._ensure_slab()
        self._verify_agents(agent_ids)
        agents = [self.agents_dict[i] for i in agent_ids]
        n_agents = len(agent_ids)
        for _ in self._end_transitions(agents, np.zeros(n_agents, dtype=np.float32), done=False):
            pass
        action = np.zeros((n_agents, self.action_size), dtype=np.float32)
        prob = np.random.rand(n_agents, self.action_size)
        if self.d_action_sizes:
            d_action_list = [np.random.randint(0, s, size=n_agents) for s in self.d_action_sizes]
            d_action_list = [np.eye(s, dtype=np.int32)[d] for d, s in zip(d_action_list, self.d_action_sizes)]
            action[:, :self.d_action_summed_size] = np.concatenate(d_action_list, axis=-1)
        if self.c_action_size:
            action[:, self.d_action_summed_size:] = np.tanh(np.random.randn(n_agents, self.c_action_size))
        slab = self.slab
        d_slot, = slab.upload([np.asarray([a.slot for a in agents], dtype=np.int32)])
        _, pre_hidden = slab.collect(d_slot, n_agents)
        dev = lambda x: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(slab.device)  # noqa: E731
        slab.stage(d_slot, n_agents, self._obs_to_device(obs_list), dev(action), dev(prob), pre_hidden,
                   torch.zeros_like(pre_hidden))
        for a in agents:
            a._tmp_index += 1
            a._has_tmp = True
        return action[..., :self.d_action_summed_size], action[..., self.d_action_summed_size:]

    def end_episode(self, agent_ids: np.ndarray, obs_list: list[np.ndarray], last_reward: np.ndarray,
                    max_reached: np.ndarray, force_terminated: bool = False):
        keep = [i for i, agent_id in enumerate(agent_ids) if agent_id in self.agents_dict]
        if not keep:
            return
        agents = [self.agents_dict[agent_ids[i]] for i in keep]
        obs = [o[keep] for o in self._obs_to_device(obs_list)] if any(a._has_tmp for a in agents) else None
        for ep in self._end_transitions(agents, [last_reward[i] for i in keep], done=True,
                                        max_reached=[max_reached[i] for i in keep],
                                        force_terminated=force_terminated, next_obs_list=obs):
            self._tmp_episode_trans_list.append(ep)

    def force_end_all_episodes(self):
        agents = [a for a in self.agents if not a.done]
        for _ in self._end_transitions(agents, [0.] * len(agents), done=True, max_reached=[True] * len(agents),
                                       force_terminated=True):
            pass   # the reference discards these episodes too (agent.py:640-650)

    def put_episode(self):
        for episode_trans in self._tmp_episode_trans_list:
            self.rl.put_episode(**episode_trans)
        self.clear_tmp_episode_trans_list()

    def train(self) -> int:
        self.rl.set_train_mode(True)
        return self.rl.train()

    def get_tmp_episode_trans_list(self) -> list[dict]:
        return self._tmp_episode_trans_list

    def clear_tmp_episode_trans_list(self) -> None:
        self._tmp_episode_trans_list.clear()

    def log_episode(self, force: bool = False) -> None:
        for episode_trans in self._tmp_episode_trans_list:
            self.rl.log_episode(force, **episode_trans)


class MultiAgentsManager:
    """The reference's `MultiAgentsManager` (agent.py:691-890): one `AgentManager` per behaviour name; building
    the learners stays with the caller (`mgr.set_rl`), as `sac_main` does."""

    def __init__(self, ma_obs_names, ma_obs_shapes, ma_obs_dtypes, ma_d_action_sizes, ma_c_action_size,
                 inference_ma_names: set[str], model_abs_dir: Path, max_episode_length: int = -1,
                 hit_reward: int | None = None):
        self._inference_ma_names = inference_ma_names
        self.model_abs_dir = model_abs_dir
        self._ma_manager: dict[str, AgentManager] = {}
        for n in ma_obs_shapes:
            self._ma_manager[n] = AgentManager(n, ma_obs_names[n], ma_obs_shapes[n], ma_obs_dtypes[n],
                                               ma_d_action_sizes[n], ma_c_action_size[n],
                                               max_episode_length=max_episode_length, hit_reward=hit_reward)
            if len(ma_obs_shapes) == 1:
                self._ma_manager[n].set_model_abs_dir(model_abs_dir)
            else:
                self._ma_manager[n].set_model_abs_dir(model_abs_dir / ma_name2path_name(n))

    def __iter__(self) -> Iterator[tuple[str, AgentManager]]:
        return iter(self._ma_manager.items())

    def __getitem__(self, k) -> AgentManager:
        return self._ma_manager[k]

    def __len__(self) -> int:
        return len(self._ma_manager)

    @property
    def done(self) -> bool:
        return all([mgr.done for _, mgr in self])

    @property
    def max_reached(self) -> bool:
        return any([mgr.max_reached for _, mgr in self])

    def reset(self) -> None:
        for _, mgr in self:
            mgr.reset()

    def reset_dead_agents(self) -> None:
        for _, mgr in self:
            mgr.reset_dead_agents()

    def reset_and_continue(self) -> None:
        for _, mgr in self:
            mgr.reset_and_continue()

    def set_train_mode(self, train_mode: bool = True):
        for n, mgr in self:
            if n in self._inference_ma_names or mgr.rl is None:
                continue
            mgr.rl.set_train_mode(train_mode)

    def get_ma_action(self, ma_agent_ids, ma_obs_list, ma_last_reward, ma_offline_action=None,
                      disable_sample: bool = False, force_rnd_if_available: bool = False):
        ma_d_action, ma_c_action = {}, {}
        if ma_offline_action is None:
            ma_offline_action = {}
        for n, mgr in self:
            if len(ma_agent_ids[n]) == 0:
                ma_d_action[n] = ma_c_action[n] = None
                continue
            ma_d_action[n], ma_c_action[n] = mgr.get_action(
                agent_ids=ma_agent_ids[n], obs_list=ma_obs_list[n], last_reward=ma_last_reward[n],
                offline_action=ma_offline_action[n] if n in ma_offline_action else None,
                disable_sample=disable_sample, force_rnd_if_available=force_rnd_if_available)
        return ma_d_action, ma_c_action

    def get_test_ma_action(self, ma_agent_ids, ma_obs_list, ma_last_reward=None, disable_sample=None):
        ma_d_action, ma_c_action = {}, {}
        for n, mgr in self:
            ma_d_action[n], ma_c_action[n] = mgr.get_test_action(agent_ids=ma_agent_ids[n], obs_list=ma_obs_list[n])
        return ma_d_action, ma_c_action

    def end_episode(self, ma_agent_ids, ma_obs_list, ma_last_reward, ma_max_reached, force_terminated: bool = False):
        for n, mgr in self:
            mgr.end_episode(agent_ids=ma_agent_ids[n], obs_list=ma_obs_list[n], last_reward=ma_last_reward[n],
                            max_reached=ma_max_reached[n], force_terminated=force_terminated)

    def force_end_all_episode(self):
        for _, mgr in self:
            mgr.force_end_all_episodes()

    def put_episode(self):
        for _, mgr in self:
            mgr.put_episode()

    def train(self, trained_steps: int) -> int:
        for n, mgr in self:
            if n in self._inference_ma_names:
                continue
            trained_steps = max(mgr.train(), trained_steps)
        return trained_steps

    def log_episode(self, force: bool = False) -> None:
        ma_episodes_info = {}
        for n, mgr in self:
            ma_episodes_info[n] = {'obs_names': mgr.obs_names, 'obs_shapes': mgr.obs_shapes,
                                   'd_action_sizes': mgr.d_action_sizes, 'c_action_size': mgr.c_action_size}
            mgr.log_episode(force)
        episodes_info_f = self.model_abs_dir / 'episodes_info.json'
        if not episodes_info_f.exists():
            with open(episodes_info_f, 'w') as f:
                json.dump(ma_episodes_info, f, indent=4)

    def save_model(self, save_replay_buffer=False) -> None:
        for n, mgr in self:
            if n in self._inference_ma_names or mgr.rl is None:
                continue
            mgr.rl.save_model(save_replay_buffer)

    def clear_tmp_episode_trans_list(self) -> None:
        for _, mgr in self:
            mgr.clear_tmp_episode_trans_list()

    def close(self) -> None:
        for _, mgr in self:
            if mgr.rl is None:
                continue
            mgr.rl.close()
