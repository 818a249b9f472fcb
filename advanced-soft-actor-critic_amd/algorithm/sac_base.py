"""`SAC_Base` — the drop-in learner for reference `algorithm/sac_base.SAC_Base`
(`/root/reference/algorithm/sac_base.py:19-2614`) with its training step re-built for MI355X.

Same constructor keywords (every `sac_config` key of the reference's `default_config.yaml`), same
model-plugin protocol (`nn.ModelRep / ModelQ / ModelPolicy ...`), same public methods
(`train`, `put_episode`, `choose_action`, `choose_attn_action`, `save_model`, ...).

How one `train()` runs here (reference call stack: SURVEY.md §3.1):
  1. PER sample + window gather + padding      2 HIP launches, data never leaves HBM
  2. Polyak soft update                         1 launch over the flat target / online buffers
  3. representation / Q / policy forward+backward on PyTorch-ROCm (rocBLAS / hipBLASLt -> MFMA)
  4. target value: rsample + tanh-squash log-prob, stored-action probabilities, ensemble subset
     min, V = minQ - alpha*logpi, V-trace scan      3 fused launches (`asac_squash_*`, `asac_vtrace_*`)
  5. clipped double-Q loss + its gradient       1 launch;  Adam per optimizer: 1 launch per segment
  6. TD error -> priority update, mu-prob / hidden-state write-back   fused launches, no D2H
The device work of a whole step has no host synchronisation, so after a few eager steps it is
captured into one hipGraph (torch.cuda.CUDAGraph) and replayed; `hip_config={'use_graph': False}`
keeps it eager.  Every random draw comes from `self.noise` (device Philox by default; tests inject
recorded draws to compare against the reference bit-for-bit on index selection).

Not carried over: `use_replay_buffer=False` (BatchBuffer path) — outside the hot path.
"""
import contextlib
import logging
import random
from collections import defaultdict
from pathlib import Path

import numpy as np
import torch
from torch import distributions, nn
from torch.nn import functional

from asac_amd import native

from . import fused_gru, fused_linear
from .fused import DeviceNoise, FlatAdam, FlatParamGroup, squash_sample, squash_sample_ls, time_slice
from .fused_mlp import DeferredPartialSums, StockMLP, describe_policy, describe_q, direct_param_grads
from .nn_models import *  # noqa: F401,F403
from .nn_models.layers.seq_layers import step_mask_cache
from .nn_models.representation import ModelSimpleRep
from .replay_buffer import PrioritizedReplayBuffer
from .sac_aux import AuxHeadsMixin
from .utils import *  # noqa: F401,F403
from .utils.enums import CURIOSITY, SEQ_ENCODER, SIAMESE
from .utils.elapse_timer import UnifiedElapsedTimer, unified_elapsed_timer
from .utils.operators import (gen_n_pre_actions, squash_correction_log_prob,
                              squash_correction_prob, sum_entropy, sum_log_prob)

try:  # tensorboard is optional (absent on the GPU box image)
    from torch.utils.tensorboard import SummaryWriter
except Exception:  # pragma: no cover
    SummaryWriter = None


def _masked_mse_backward(pred, target, padding_mask, loss_out, fused=True):
    """loss = mean over ALL elements of (pred - target)^2 with padded rows zeroed (`mse_loss(reduction='none') *
    ~mask` then `mean`, reference sac_base.py:1962-1964) -> `loss_out`, and `loss.backward()` through `pred`:
    the loss is the root, so its gradient 2 d / N is handed to autograd directly (six launches in all)."""
    with torch.no_grad():
        if (fused and pred.is_cuda and pred.dim() == 3 and pred.dtype == torch.float32 and pred.is_contiguous()
                and target.stride(-1) == 1 and pred.numel() <= native.MASKED_MSE_MAX and padding_mask.dtype == torch.bool
                and (padding_mask.stride(1) == 1 or padding_mask.shape[1] == 1)):
            d = torch.empty_like(pred)
            native.masked_mse(pred.detach(), target, padding_mask, d, loss_out)       # one launch
        else:
            d = (pred - target).mul_((~padding_mask).unsqueeze(-1))
            flat = d.reshape(-1)
            torch.div(torch.dot(flat, flat), flat.numel(), out=loss_out)
            d.mul_(2. / flat.numel())
    with direct_param_grads():      # detached inputs: the backward reaches the model's own parameters only
        pred.backward(d)


def _real(x):
    """a concatenation that left the representation un-consumed (a plugin whose returned state IS a `torch.cat`) is
    formed here: nothing outside the representation ever sees an `adjacent_cat.DeferredCat`"""
    from .adjacent_cat import DeferredCat
    return x.materialize() if isinstance(x, DeferredCat) else x


class _Window:
    """views of the step's static batch tensors and what the phases of `_device_step_body` hand to each other"""


class _AfterPolicy:
    """what the stock networks' pass of the UPDATED policy over the window leaves for the temperature step, the TD
    error and the write-backs (all None: the generic path computes each where it is needed)"""
    ls_win = alpha_logp = probs_win = td_sample = None      # (loc | scale) [B, L, 2A]; log pi of the temperature sample;
    #                                                           pi(stored actions) [B, L, A]; the TD target's (action, log pi)
    sc_write = sc_alpha = None                               # sidecar jobs still waiting for a host launch
    side_cq = td_q_table = ls_td = td_pi = None              # online Q(s_b, a_b); target Q on the TD sample; the TD
    #                                                           target's own policy pass (over target states) and its pi
    mu_written = False                                       # the mu-probability write-back is issued / riding


class SAC_Base(AuxHeadsMixin):
    _closed = False

    def __init__(self,
                 obs_names: list[str],
                 obs_shapes: list[tuple[int]],
                 d_action_sizes: list[int],
                 c_action_size: int,
                 model_abs_dir: Path | None,
                 nn,

                 device: str | None = None,
                 ma_name: str | None = None,
                 summary_path: str | None = 'log',
                 train_mode: bool = True,
                 last_ckpt: str | None = None,

                 nn_config: dict | None = None,

                 seed: float | None = None,
                 write_summary_per_step: float = 1e3,
                 save_model_per_step: float = 1e5,

                 use_replay_buffer: bool = True,
                 use_priority: bool = True,

                 ensemble_q_num: int = 2,
                 ensemble_q_sample: int = 2,

                 burn_in_step: int = 0,
                 n_step: int = 1,
                 seq_encoder: SEQ_ENCODER | None = None,

                 batch_size: int = 256,
                 tau: float = 0.005,
                 update_target_per_step: int = 1,
                 init_log_alpha: float = -2.3,
                 use_auto_alpha: bool = True,
                 target_d_alpha: float = 0.98,
                 target_c_alpha: float = 1.,
                 d_policy_entropy_penalty: float = 0.5,

                 learning_rate: float = 3e-4,

                 gamma: float = 0.99,
                 v_lambda: float = 1.,
                 v_rho: float = 1.,
                 v_c: float = 1.,
                 clip_epsilon: float = 0.2,

                 discrete_dqn_like: bool = False,
                 discrete_dqn_epsilon: float = 0.2,
                 use_n_step_is: bool = True,

                 siamese: SIAMESE | None = None,
                 siamese_use_q: bool = False,
                 siamese_use_adaptive: bool = False,

                 use_prediction: bool = False,
                 transition_kl: float = 0.8,
                 use_extra_data: bool = True,

                 curiosity: CURIOSITY | None = None,
                 curiosity_strength: float = 1.,
                 use_rnd: bool = False,
                 rnd_n_sample: int = 10,

                 use_normalization: bool = False,

                 offline_enabled: bool = False,
                 offline_loss: bool = False,

                 action_noise: list[float] | None = None,

                 replay_config: dict | None = None,
                 hip_config: dict | None = None):
        """Arguments as in the reference (`sac_base.py:22-164`).  `hip_config` is the one new,
        optional section: {'use_graph': bool (default True), 'graph_warmup': int (default 3),
        'fused_mlp': bool (default True: stock ModelQ / ModelPolicy run as fused MFMA kernels),
        'twin_rep': bool (default True: the online and the target representation's fused GRU passes over the
        sampled window share one launch, see `fused_gru.TwinPass`)}."""
        self._kwargs = {k: v for k, v in locals().items() if k != 'self'}

        self.obs_names = obs_names
        self.obs_shapes = obs_shapes
        self.d_action_sizes = d_action_sizes
        self.d_action_summed_size = sum(d_action_sizes)
        self.d_action_branch_size = len(d_action_sizes)
        self.c_action_size = c_action_size
        self.model_abs_dir = model_abs_dir
        self.ma_name = ma_name
        self.train_mode = train_mode

        self.use_replay_buffer = use_replay_buffer
        self.use_priority = use_priority
        self.ensemble_q_num = ensemble_q_num
        self.ensemble_q_sample = ensemble_q_sample
        self.burn_in_step = burn_in_step
        self.n_step = n_step
        self.seq_encoder = seq_encoder
        self.write_summary_per_step = int(write_summary_per_step)
        self.save_model_per_step = int(save_model_per_step)
        self.batch_size = batch_size
        self.tau = tau
        self.update_target_per_step = update_target_per_step
        self.use_auto_alpha = use_auto_alpha
        self.target_d_alpha = target_d_alpha
        self.target_c_alpha = target_c_alpha
        self.d_policy_entropy_penalty = d_policy_entropy_penalty
        self.learning_rate = learning_rate
        self.gamma = gamma
        self.v_lambda = v_lambda
        self.v_rho = v_rho
        self.v_c = v_c
        self.clip_epsilon = clip_epsilon
        self.discrete_dqn_like = discrete_dqn_like
        self.discrete_dqn_epsilon = discrete_dqn_epsilon
        self.use_n_step_is = use_n_step_is
        self.siamese = siamese
        self.siamese_use_q = siamese_use_q
        self.siamese_use_adaptive = siamese_use_adaptive
        self.use_prediction = use_prediction
        self.transition_kl = transition_kl
        self.use_extra_data = use_extra_data
        self.curiosity = curiosity
        self.curiosity_strength = curiosity_strength
        self.use_rnd = use_rnd
        self.rnd_n_sample = rnd_n_sample
        self.use_normalization = use_normalization
        self.offline_enabled = offline_enabled
        self.offline_loss = offline_loss
        self.action_noise = action_noise

        hip_config = dict(hip_config or {})
        self._use_graph = bool(hip_config.get('use_graph', True))
        self._graph_warmup = int(hip_config.get('graph_warmup', 3))
        self._direct_graph_launch = bool(hip_config.get('direct_graph_launch', True))   # (False: every replay through torch)
        self._dist = hip_config.get('dist')     # parallel.DataParallelContext or None
        self._use_fused_mlp = bool(hip_config.get('fused_mlp', True))
        self._graph_collectives = bool(hip_config.get('graph_collectives', True))
        self._twin_rep = bool(hip_config.get('twin_rep', True))
        self._fuse_linear_tanh = bool(hip_config.get('fused_linear_tanh', True))
        self._use_sidecars = bool(hip_config.get('sidecars', True))
        # batches of 257 .. 1 024: the IS weights in an extra workgroup of the gather's launch instead of behind an exchange
        # between the sampler's workgroups (asac_step_prologue_sample_partial + asac_window_gather_pad_w)
        self._defer_is_weights = bool(hip_config.get('defer_is_weights', True))
        self._fused_policy_step = bool(hip_config.get('fused_policy_step', True))
        self._fused_forward_chain = bool(hip_config.get('fused_forward_chain', True))
        self._fused_td_chain = bool(hip_config.get('fused_td_chain', True))
        self._fused_td_update = bool(hip_config.get('fused_td_update', True))
        self._fused_q_return = bool(hip_config.get('fused_q_return', True))
        self._defer_return = False      # `_get_y`: hand the return's arguments back (`_deferred_return`) instead of launching
        self._deferred_return = None
        self._td_update_with = None     # (replay buffer, ids): the TD error's return launch also updates the priorities
        self._fused_q_state_grads = bool(hip_config.get('fused_q_state_grads', True))
        self._gru_backward_at = bool(hip_config.get('gru_backward_at', True))
        self._adjacent_cat = bool(hip_config.get('adjacent_cat', True))
        self._fold_rep_q_adam = bool(hip_config.get('fold_rep_q_adam', True))
        self._rep_grad_one_position = bool(hip_config.get('rep_grad_one_position', True))
        self._deferred_cat = bool(hip_config.get('deferred_cat', True))
        self._rep_from_burn_in = bool(hip_config.get('rep_from_burn_in', True))
        self._fused_curiosity = bool(hip_config.get('fused_curiosity', True))
        self._fused_rpm_loss = bool(hip_config.get('fused_rpm_loss', True))
        # one backward walk per prediction model (gates and model gradients from it): sac_aux._train_rpm
        self._rpm_single_backward = bool(hip_config.get('rpm_single_backward', True))
        self._fused_gating = bool(hip_config.get('fused_gating', True))          # asac_cosine_gate_add
        # the convolution stack's share of the three per-loss walks of `_train_rpm` as one launch (asac_conv2_backward_multi)
        self._rpm_conv_one_launch = bool(hip_config.get('rpm_conv_one_launch', True))
        # one sampled batch in flight, the reference's schedule (replay_buffer.py:275, 339-396): see `_step_sample`
        self._lookahead = int(hip_config.get('lookahead', 0))
        assert self._lookahead in (0, 1), 'hip_config lookahead: 0 (synchronous sampling) or 1 (one batch in flight)'
        # (graph branches — the next batch's draw, the target representation's pass — on a second stream were measured
        # slower than the launches in line and are gone: a fork / join inside a replayed hipGraph costs more than the small
        # launches it hides, NOTES.md section 1 rounds 4 / 5)
        self._la_gather_sidecar = bool(hip_config.get('lookahead_gather_sidecar', True)) and self._use_sidecars
        self._la_gather = None
        self._fuse_prediction_dense = bool(hip_config.get('fuse_prediction_dense', True))
        self._fused_q_loss_with_aux = bool(hip_config.get('fused_q_loss_with_aux', True))
        self._head_sums_members = bool(hip_config.get('head_sums_members', True))
        self._rep_epilogue, self._rep_epilogue_hp = False, None      # (False: not looked at yet; None: not applicable)
        self._cat_mode = None
        self._g_state_base = None
        self._vtrace_sidecars = self._pending_alpha = None
        self._dist_sampling = hip_config.get('dist_sampling', 'throughput')     # 'throughput' | 'parity' 
        self._dist_ready = False        # (collective decision, `_ready_to_train`)
        assert self._dist_sampling in ('throughput', 'parity')

        self._set_logger()

        if not use_replay_buffer:
            raise NotImplementedError('use_replay_buffer=False (BatchBuffer) is outside the MI355X hot path')

        if self.use_n_step_is and c_action_size == 0 and len(d_action_sizes) != 0 and discrete_dqn_like:
            self.use_n_step_is = False

        if device is None:
            if not torch.cuda.is_available():
                raise native.AsacNativeError('SAC_Base needs an MI355X (cuda/ROCm device); there is no CPU path')
            self.device = torch.device(f'cuda:{random.randint(0, torch.cuda.device_count() - 1)}')
        else:
            self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise native.AsacNativeError(f'SAC_Base needs a cuda/ROCm device, got {self.device}; no CPU fallback')
        if self.device.index is None:
            self.device = torch.device('cuda', torch.cuda.current_device())
        torch.cuda.set_device(self.device)
        native.load()
        self._logger.info(f'Device: {self.device.type}:{self.device.index}')

        if seed is not None:
            torch.manual_seed(seed)
            torch.cuda.manual_seed_all(seed)

        distributions.Distribution.set_default_validate_args(False)

        self.summary_writer = None
        self.summary_available = False
        if summary_path and self.model_abs_dir and self.train_mode and SummaryWriter is not None:
            self.summary_writer = SummaryWriter(str(Path(self.model_abs_dir).joinpath(summary_path)))
            self.summary_available = True

        self._profiler = UnifiedElapsedTimer(self._logger)
        self.noise = DeviceNoise()
        self._graph = None
        self._graph_hp, self._all_optimizers = None, None     # `_optimizer_hp()` the captured graphs were made with
        self._graph_runs = {}           # run length -> (the step graph it was captured beside, graph, exec handle)
        self._la_graphs = {}      # hip_config['lookahead']
        self._graph_failed = False
        self._eager_steps = 0

        self._build_model(nn, nn_config, init_log_alpha, learning_rate)
        self._build_ckpt()
        self._init_replay_buffer(replay_config)
        self._init_or_restore(int(last_ckpt) if last_ckpt is not None else None)

    def _set_logger(self):
        self._logger = logging.getLogger('sac.base' if self.ma_name is None else f'sac.base.{self.ma_name}')

    # ==========================================================================================
    # construction (reference sac_base.py:271-491)
    # ==========================================================================================
    def _build_model(self, nn, nn_config, init_log_alpha, learning_rate) -> None:
        nn_config = defaultdict(dict, nn_config or {})
        rep_kw = nn_config['rep'] or {}
        pol_kw = nn_config['policy'] or {}
        dev = self.device

        self.global_step = torch.tensor(0, dtype=torch.int64, requires_grad=False, device='cpu')
        self._gamma_ratio = torch.logspace(0, self.n_step - 1, self.n_step, self.gamma, device=dev)
        self._lambda_ratio = torch.logspace(0, self.n_step - 1, self.n_step, self.v_lambda, device=dev)
        self._v_rho_f, self._v_c_f = float(self.v_rho), float(self.v_c)
        self.v_rho = torch.tensor(self.v_rho, device=dev)
        self.v_c = torch.tensor(self.v_c, device=dev)

        pad = [np.eye(s, dtype=np.float32)[0] for s in self.d_action_sizes]
        self._np_padding_action = np.concatenate(pad + [np.zeros(self.c_action_size, dtype=np.float32)], axis=-1)
        self._padding_action = torch.from_numpy(self._np_padding_action).to(dev)

        B = self.batch_size
        A_all = self.d_action_summed_size + self.c_action_size

        # -- representation ----------------------------------------------------------------------
        rep_args = (self.obs_names, self.obs_shapes, self.d_action_sizes, self.c_action_size)
        ModelRep = self._wrap_normalized_rep(nn.ModelRep) if self.use_normalization else nn.ModelRep
        self.model_rep = ModelRep(*rep_args, False, self.model_abs_dir, **rep_kw).to(dev)
        self.model_target_rep = ModelRep(*rep_args, True, self.model_abs_dir, **rep_kw).to(dev)
        if self._fuse_linear_tanh:     # the plugins' Linear + Tanh state heads: one launch per pass
            from .fused_linear import fuse_linear_tanh_heads
            fuse_linear_tanh_heads(self.model_rep), fuse_linear_tanh_heads(self.model_target_rep)
        test_obs = [torch.rand(B, 1, *s, device=dev) for s in self.obs_shapes]
        test_pre_action = torch.rand(B, 1, A_all, device=dev)
        with torch.no_grad():
            if self.seq_encoder == SEQ_ENCODER.ATTN:
                test_index = torch.zeros((B, 1), dtype=torch.int32, device=dev)
                st, hs, _ = self.model_rep(1, test_index, test_obs, test_pre_action, None)
            else:
                st, hs = self.model_rep(test_obs, test_pre_action, None)
        self.state_size, self.seq_hidden_state_shape = st.shape[-1], hs.shape[2:]
        for p in self.model_target_rep.parameters():
            p.requires_grad = False
        self._logger.info(f'State size: {self.state_size}')
        self._logger.info(f'Seq hidden state shape: {tuple(self.seq_hidden_state_shape)}')

        # -- Q ensemble and policy -----------------------------------------------------------------
        mk_q = lambda tgt: nn.ModelQ(self.state_size, self.d_action_sizes, self.c_action_size,  # noqa: E731
                                     tgt, self.model_abs_dir).to(dev)
        self.model_q_list = [mk_q(False) for _ in range(self.ensemble_q_num)]
        self.model_target_q_list = [mk_q(True) for _ in range(self.ensemble_q_num)]
        for q in self.model_target_q_list:
            for p in q.parameters():
                p.requires_grad = False
        self.model_policy = nn.ModelPolicy(self.state_size, self.d_action_sizes, self.c_action_size,
                                           self.model_abs_dir, **pol_kw).to(dev)

        # -- alpha -------------------------------------------------------------------------------------
        self.log_d_alpha = nn_parameter_scalar(init_log_alpha, dev)
        self.log_c_alpha = nn_parameter_scalar(init_log_alpha, dev)
        if self.d_action_sizes:
            sizes = torch.tensor(self.d_action_sizes, device=dev)
            sizes = torch.repeat_interleave(sizes.type(torch.float32), sizes)
            self.target_d_alpha = self.target_d_alpha * (-torch.log(1 / sizes))

        # -- curiosity -----------------------------------------------------------------------------------
        self.model_forward_dynamic = self.model_inverse_dynamic = None
        if self.curiosity == CURIOSITY.FORWARD:
            self.model_forward_dynamic = nn.ModelForwardDynamic(self.state_size, A_all).to(dev)
        elif self.curiosity == CURIOSITY.INVERSE:
            self.model_inverse_dynamic = nn.ModelInverseDynamic(self.state_size, A_all).to(dev)

        # -- flat parameter / gradient / moment buffers ------------------------------------------------------
        named = [('rep', list(self.model_rep.parameters()))]
        named += [(f'q_{i}', list(q.parameters())) for i, q in enumerate(self.model_q_list)]
        named += [('policy', list(self.model_policy.parameters())),
                  ('alpha', [self.log_d_alpha, self.log_c_alpha])]
        cur = self.model_forward_dynamic or self.model_inverse_dynamic
        if cur is not None:
            named.append(('curiosity', list(cur.parameters())))
        named += self._build_aux(nn, test_obs)      # siamese / prediction / RND heads (sac_aux.py)
        self._params = FlatParamGroup(named, dev, with_grad=True)
        tnamed = [('rep', list(self.model_target_rep.parameters()))]
        tnamed += [(f'q_{i}', list(q.parameters())) for i, q in enumerate(self.model_target_q_list)]
        self._target_params = FlatParamGroup(tnamed, dev, with_grad=False)
        self._polyak_len = self._params.span('rep', f'q_{self.ensemble_q_num - 1}')[1]
        assert self._polyak_len == self._target_params.numel
        if self._dist is not None:   # replicas start from rank 0's initialisation
            self._dist.broadcast_(self._params.flat)

        self._opt_steps = torch.zeros(1, dtype=torch.int64, device=dev)
        self._exp_avg = torch.zeros_like(self._params.flat)
        self._exp_avg_sq = torch.zeros_like(self._params.flat)

        def adam(names):
            if all(len(self._params.params[n]) == 0 for n in names):
                return None
            return FlatAdam(self._params, names, learning_rate, self._opt_steps, self._exp_avg, self._exp_avg_sq)

        self.optimizer_rep = adam(['rep'])
        self.optimizer_q_list = [adam([f'q_{i}']) for i in range(self.ensemble_q_num)]
        self.optimizer_policy = adam(['policy'])
        if self.use_auto_alpha:
            self.optimizer_alpha = adam(['alpha'])
        if cur is not None:
            self.optimizer_curiosity = adam(['curiosity'])
        if self.siamese is not None:
            self.optimizer_siamese = adam(['siamese'])
        if self.use_prediction:
            self.optimizer_prediction = adam(['prediction'])
        if self.use_rnd:
            self.optimizer_rnd = adam(['rnd'])

        # -- stock-network fast path: one MFMA launch per pass of the Q ensemble / policy ------------------------
        self._fq = self._ftq = self._fpi = None
        if self._use_fused_mlp and self.c_action_size and not self.d_action_sizes:
            dq = [describe_q(q) for q in self.model_q_list + self.model_target_q_list]
            seg = self._params.segments
            stride = seg['q_0'][1] - seg['q_0'][0]
            consecutive = all(seg[f'q_{i}'][0] == seg['q_0'][0] + i * stride for i in range(self.ensemble_q_num))
            if all(d is not None for d in dq) and consecutive:
                tseg = self._target_params.segments
                self._fq = StockMLP(dq[0], self._params.flat, self._params.grad, seg['q_0'][0], stride,
                                    self.ensemble_q_num, dev, [p for q in self.model_q_list for p in q.parameters()])
                self._ftq = StockMLP(dq[0], self._target_params.flat, None, tseg['q_0'][0], stride,
                                     self.ensemble_q_num, dev)
            dp = describe_policy(self.model_policy)
            if dp is not None:
                self._fpi = StockMLP(dp, self._params.flat, self._params.grad, seg['policy'][0],
                                     seg['policy'][1] - seg['policy'][0], 1, dev, list(self.model_policy.parameters()))
        self._wide_critics = self._fq is not None and self._fq.wide
        if self._wide_critics:
            # critics whose first layer is wider than 64 inputs (64-wide state + action): one launch per network pass — the
            # chains that pack several networks into one launch are not taken and no critic launch can host a sidecar job.
            # The jobs that do not need one still ride: the sampler in the step prologue, the write-backs' passes and the
            # temperature step in the TD error's return launch and the priority update (as after the fused policy chain)
            self._la_gather_sidecar = False
        self._logger.info(f'fused stock MLP path: Q={self._fq is not None} policy={self._fpi is not None}')
        # When the stock networks and the temperatures are the only trainable parameters, every gradient
        # slot is written exactly once per step by a kernel: overwrite instead of memset + accumulate.
        stock_only = {f'q_{i}' for i in range(self.ensemble_q_num)} | {'policy', 'alpha'}
        self._grads_overwrite = (self._fq is not None and self._fpi is not None and not self.offline_enabled
                                 and all(stop == start or name in stock_only
                                         for name, (start, stop) in self._params.segments.items()))
        if self._grads_overwrite:
            self._fq.accumulate = self._fpi.accumulate = False

        # -- static step buffers (stable addresses for graph replay) --------------------------------------------
        n, A, E, Es = self.n_step, self.c_action_size, self.ensemble_q_num, self.ensemble_q_sample
        f32 = dict(dtype=torch.float32, device=dev)
        A1 = max(A, 1)
        sizes = [B * (n + 1) * A1, B * A1, B * A1, B * (n + 1) * A1]     # consumption order of a step
        self._eps_all = torch.zeros(sum(sizes), **f32)
        o = np.cumsum([0] + sizes)
        self._eps_y = self._eps_all[o[0]:o[1]].view(B, n + 1, A1)
        self._eps_pi = self._eps_all[o[1]:o[2]].view(B, A1)
        self._eps_alpha = self._eps_all[o[2]:o[3]].view(B, A1)
        self._eps_td = self._eps_all[o[3]:o[4]].view(B, n + 1, A1)
        names = ('y_dn', 'y_dnext', 'y_cn', 'y_cnext', 'pi_d', 'pi_c', 'td_dn', 'td_dnext', 'td_cn', 'td_cnext')
        self._subsets_all = torch.arange(Es, dtype=torch.int32, device=dev).repeat(len(names), 1).contiguous()
        self._subsets = {k: self._subsets_all[i] for i, k in enumerate(names)}   # views of one buffer
        self._y_buf = torch.zeros(B, **f32)
        self._y_td_buf = torch.zeros(B, **f32)
        self._td_error = torch.zeros(B, **f32)
        self._stats = {k: torch.zeros((), **f32) for k in ('d_entropy', 'c_entropy', 'loss_curiosity', 'loss_policy')}
        self._loss_q_e = torch.zeros(E, **f32)             # per-ensemble Q losses of the last step
        self._stats['loss_q'] = self._loss_q_e[0]
        self._grad_q = torch.zeros(E, B, **f32)            # d loss / d q written by the loss kernels
        # zeroed exchange words of `asac_mse_mean_grad` (the observation model's frame loss, sac_aux._train_rpm)
        self._mse_big_ws = torch.zeros(native.mse_mean_grad_workspace(), **f32) if self.use_prediction else None
        self._grad_logp = torch.zeros(B, **f32)
        self._ls_y = None
        self._cq_buf, self._tq_buf, self._cq_td_buf = (torch.zeros(E, B, 1, **f32) for _ in range(3))
        self._pi_q, self._pi_stats_src = torch.zeros(E, B, 1, **f32), None
        self._pi_a, self._pi_logp, self._pi_sampled = torch.zeros(B, A1, **f32), torch.zeros(B, **f32), False
        self._graph_exec, self._graph_exec_checked, self._graph_stats_src = None, False, None
        # online + target representation over the same window: fused GRU layers pair up in one launch
        self._rep_twin = None
        if self._twin_rep and type(self.model_rep) is not ModelSimpleRep:
            from .fused_gru import TwinPass
            self._rep_twin = TwinPass(self.model_rep, self.model_target_rep)

    def _build_ckpt(self) -> None:
        """name -> module / optimizer / tensor, same keys as the reference (sac_base.py:493-566)."""
        ck = self.ckpt_dict = {'global_step': self.global_step}
        if self.optimizer_rep is not None:
            ck['model_rep'], ck['model_target_rep'], ck['optimizer_rep'] = \
                self.model_rep, self.model_target_rep, self.optimizer_rep
        for i in range(self.ensemble_q_num):
            ck[f'model_q_{i}'] = self.model_q_list[i]
            ck[f'model_target_q_{i}'] = self.model_target_q_list[i]
            ck[f'optimizer_q_{i}'] = self.optimizer_q_list[i]
        ck['model_policy'], ck['optimizer_policy'] = self.model_policy, self.optimizer_policy
        ck['log_d_alpha'], ck['log_c_alpha'] = self.log_d_alpha, self.log_c_alpha
        if self.use_auto_alpha:
            ck['optimizer_alpha'] = self.optimizer_alpha
        if self.curiosity == CURIOSITY.FORWARD:
            ck['model_forward_dynamic'] = self.model_forward_dynamic
        elif self.curiosity == CURIOSITY.INVERSE:
            ck['model_inverse_dynamic'] = self.model_inverse_dynamic
        if self.curiosity is not None:
            ck['optimizer_curiosity'] = self.optimizer_curiosity
        self._aux_ckpt(ck)
        total = sum(p.numel() for m in ck.values() if isinstance(m, nn.Module) for p in m.parameters())
        self._logger.info(f'Parameters: {total}')

    def _init_or_restore(self, last_ckpt: int | None) -> None:
        """sac_base.py:568-629."""
        self.ckpt_dir = None
        if not self.model_abs_dir:
            self._update_target_variables()
            return
        self.ckpt_dir = ckpt_dir = Path(self.model_abs_dir).joinpath('model')
        ckpts = sorted(int(p.stem) for p in ckpt_dir.glob('*.pth')) if ckpt_dir.exists() else []
        ckpt_dir.mkdir(parents=True, exist_ok=True)
        if not ckpts:
            self._logger.info('Initializing from scratch')
            self._update_target_variables()
            return
        if last_ckpt is None or last_ckpt not in ckpts:
            if last_ckpt is not None:
                self._logger.warning(f'{last_ckpt} NOT IN {ckpts}, using {ckpts[-1]}')
            last_ckpt = ckpts[-1]
        path = ckpt_dir.joinpath(f'{last_ckpt}.pth')
        restored = torch.load(path, map_location=self.device, weights_only=True)
        failed = False
        for name, obj in self.ckpt_dict.items():
            if name not in restored:
                self._logger.warning(f'{name} not in {last_ckpt}.pth')
                continue
            if isinstance(obj, torch.Tensor):
                if name == 'global_step':
                    self.global_step.copy_(restored[name].to('cpu'))
                else:
                    with torch.no_grad():
                        obj.copy_(restored[name])
                continue
            try:
                if failed and name.startswith('optimizer'):
                    continue
                obj.load_state_dict(restored[name])
            except RuntimeError as e:
                failed = True
                self._logger.error(e)
            if isinstance(obj, nn.Module):
                obj.train(self.train_mode)
        self._params.rebind()
        self._target_params.rebind()
        self._logger.info(f'Restored from {path}')
        if self.train_mode and self.use_replay_buffer:
            self.replay_buffer.load(ckpt_dir, last_ckpt)
            self._logger.info('Replay buffer restored')

    def _init_replay_buffer(self, replay_config: dict | None = None) -> None:
        if not self.train_mode:
            return
        self.replay_buffer = PrioritizedReplayBuffer(batch_size=self.batch_size,
                                                     sample_prev_n=self.burn_in_step,
                                                     sample_post_n=self.n_step,
                                                     device=self.device,
                                                     logger_parent_name=self._logger.name,
                                                     **(replay_config or {}))
        self.replay_buffer.set_window_padding(self._padding_action)
        self.replay_buffer.uniform_source = self.noise
        self._cat_mode = None
        if self._adjacent_cat and type(self.model_rep) is not ModelSimpleRep:
            # a user representation receives (obs, previous actions): keep them side by side in the static batch so
            # that the concatenation sequence modules start with is a view (adjacent_cat.py); the gather then also
            # delivers the derived window inputs (index_x, padding_mask_x, pre_action) in its own launch
            from .adjacent_cat import AdjacentCat
            self.replay_buffer.join_vector_obs_with_pre_action(self.d_action_summed_size + self.c_action_size)
            width = native.LINEAR_TANH_MAX_IN if (self._deferred_cat and self._fuse_linear_tanh) else 0
            # (deferred only at the widths a fused head of this representation takes: nothing else can use the two blocks)
            from .fused_linear import LinearTanhHead
            widths = {m[0].in_features for m in self.model_rep.modules() if isinstance(m, LinearTanhHead)}
            self._cat_mode = lambda: AdjacentCat(width if widths else 0, widths)
        if self._dist is not None and self._dist_sampling == 'parity':
            # SURVEY 8e "parity": every batch is the reference's stratified sample over the UNION of the ranks' shards
            # (global batch = world_size * batch_size, this rank trains on batch_size rows of it)
            from .parallel import ProductShard, ShardedParityReplay
            rb = self.replay_buffer
            global_batch = self.batch_size * self._dist.world_size
            rb.sharded = ShardedParityReplay(self._dist, ProductShard(rb, global_batch), global_batch, self.device)
            rb._u = torch.zeros(global_batch, dtype=torch.float64, device=self.device)    # the GLOBAL batch's uniforms
        elif self._dist is not None:
            self.replay_buffer.min_ratio_reducer = self._dist.all_reduce_min_

    # ==========================================================================================
    # small public surface (reference sac_base.py:648-743)
    # ==========================================================================================
    def set_train_mode(self, train_mode=True):
        self.train_mode = train_mode
        for m in self.ckpt_dict.values():
            if isinstance(m, nn.Module):
                m.train(mode=train_mode)

    def save_model(self, save_replay_buffer=False) -> None:
        if self.ckpt_dir is None:
            return
        if self.use_replay_buffer and self.train_mode:
            self.replay_buffer.check_health()      # never write a checkpoint of a run whose td-errors went NaN
        step = self.get_global_step()
        path = self.ckpt_dir.joinpath(f'{step}.pth')
        torch.save({k: (v.detach().clone() if isinstance(v, torch.Tensor) else v.state_dict())
                    for k, v in self.ckpt_dict.items()}, path)
        self._logger.info(f'Model saved at {path}')
        if self.use_replay_buffer and save_replay_buffer:
            self.replay_buffer.save(self.ckpt_dir, step)

    def write_constant_summaries(self, constant_summaries: list[dict], iteration=None) -> None:
        if self.summary_writer is None:
            return
        for s in constant_summaries:
            self.summary_writer.add_scalar(s['tag'], s['simple_value'],
                                           self.get_global_step() if iteration is None else iteration)
        self.summary_writer.flush()

    def write_histogram_summaries(self, histograms, iteration=None) -> None:
        if self.summary_writer is None:
            return
        for s in histograms:
            self.summary_writer.add_histogram(s['tag'], s['histogram'],
                                              self.get_global_step() if iteration is None else iteration)
        self.summary_writer.flush()

    def log_episode(self, force: bool = False, **episode_trans) -> None:
        if not force and (self.summary_writer is None or not self.summary_available):
            return
        self.summary_available = False

    def increase_global_step(self) -> int:
        self.global_step.add_(1)
        return self.global_step.item()

    def set_global_step(self, global_step):
        if isinstance(global_step, torch.Tensor):
            global_step = global_step.item()
        if global_step == self.get_global_step():
            return
        self._logger.warning(f'Global step {self.get_global_step()} -> {global_step}')
        self.global_step.fill_(global_step)

    def get_global_step(self) -> int:
        return self.global_step.item()

    def get_initial_action(self, batch_size, get_numpy=True):
        if get_numpy:
            parts = [np.eye(s, dtype=np.float32)[np.random.randint(0, s, size=batch_size)]
                     for s in self.d_action_sizes]
            parts.append(np.zeros([batch_size, self.c_action_size], dtype=np.float32))
            return np.concatenate(parts, axis=-1)
        parts = [functional.one_hot(torch.randint(0, s, (batch_size,), device=self.device), num_classes=s).float()
                 for s in self.d_action_sizes]
        parts.append(torch.zeros((batch_size, self.c_action_size), device=self.device))
        return torch.cat(parts, dim=-1)

    def get_initial_seq_hidden_state(self, batch_size, get_numpy=True):
        if get_numpy:
            return np.zeros([batch_size, *self.seq_hidden_state_shape], dtype=np.float32)
        return torch.zeros([batch_size, *self.seq_hidden_state_shape], device=self.device)

    @torch.no_grad()
    def _update_target_variables(self, tau=1.) -> None:
        """Polyak over the flat [rep | q_0 .. q_E-1] buffers: one launch (reference 745-764 loops
        over parameters with three kernels each)."""
        if self._polyak_len > 0:
            native.polyak(self._target_params.flat[:self._polyak_len], self._params.flat[:self._polyak_len], tau)

    def _process_torch_obs_list(self, obs_list):
        for i, o in enumerate(obs_list):
            if o.dtype == torch.uint8:
                obs_list[i] = o.type(torch.float32) / 255.
            elif o.dtype == torch.bool:
                obs_list[i] = o.type(torch.float32)

    # ==========================================================================================
    # acting (reference sac_base.py:858-1086) — eager torch, NumPy in/out, not on the train path
    # ==========================================================================================
    @torch.no_grad()
    def _random_action(self, d_action, c_action):
        if self.action_noise is None:
            return d_action, c_action
        batch = max(d_action.shape[0], c_action.shape[0])
        noise = torch.linspace(*self.action_noise, steps=batch, device=self.device)
        if self.d_action_sizes:
            rnd = [functional.one_hot(torch.argmax(torch.rand(batch, s, device=self.device), dim=-1), s)
                   for s in self.d_action_sizes]
            rnd = torch.cat(rnd, dim=-1).type(torch.float32)
            mask = torch.rand(batch, device=self.device) < noise
            d_action[mask] = rnd[mask]
        if self.c_action_size:
            c_action = torch.tanh(torch.atanh(c_action)
                                  + torch.randn(batch, self.c_action_size, device=self.device) * noise.unsqueeze(1))
        return d_action, c_action

    @torch.no_grad()
    def _choose_action(self, obs_list, state, offline_action=None, disable_sample=False,
                       force_rnd_if_available=False):
        batch = state.shape[0]
        use_rnd = self.use_rnd and (self.train_mode or force_rnd_if_available)
        if (offline_action is None and not use_rnd and self.action_noise is None and self._stock_c_only()
                and state.dim() == 2):
            # stock Gaussian policy, continuous actions only: the policy's forward is ONE launch (`asac_mlp_forward`),
            # the sample and the probability of the chosen action one elementwise launch each — instead of ~25 eager
            # module / distribution launches (0.45 ms of host time per environment step)
            A = self.c_action_size
            ls = self._fpi._launch_forward(StockMLP._rows(state, self.state_size), None)[0]      # [batch, 2A] (loc | scale)
            loc, scale = ls[:, :A], ls[:, A:]
            if disable_sample:
                c_action = torch.tanh(loc)
            else:
                eps = torch.empty((batch, A), dtype=torch.float32, device=self.device)
                self.noise.normal_(eps)
                c_action = torch.empty((batch, A), dtype=torch.float32, device=self.device)
                native.squash_sample_fwd(loc, scale, eps, c_action, torch.empty(batch, dtype=torch.float32, device=self.device))
            prob = torch.empty((batch, A), dtype=torch.float32, device=self.device)
            win = lambda t: t.as_strided((batch, 1, A), (t.stride(0), t.stride(0), 1))  # noqa: E731
            native.squash_prob(win(loc), win(scale), win(c_action), 0, win(prob), 0)
            return c_action, prob
        d_policy, c_policy = self.model_policy(state, obs_list)
        if offline_action is None:
            if self.d_action_sizes and self.discrete_dqn_like:
                # greedy w.r.t. the first critic, epsilon / RND-novelty random (reference 904-930)
                d_qs, _ = self.model_q_list[0](state, c_policy.sample() if self.c_action_size else None, obs_list)
                d_action = torch.cat([functional.one_hot(torch.argmax(part, dim=-1), size).type(torch.float32)
                                      for part, size in zip(d_qs.split(self.d_action_sizes, dim=-1),
                                                            self.d_action_sizes)], dim=-1)
                if self.train_mode:
                    if use_rnd:
                        s_rnd = torch.sigmoid(self.model_rnd.cal_s_rnd(state))
                        t_rnd = torch.sigmoid(self.model_target_rnd.cal_s_rnd(state))
                        mask = torch.rand(batch).to(self.device) < torch.mean(torch.abs(s_rnd - t_rnd), dim=-1)
                    else:
                        mask = (torch.rand(batch) < self.discrete_dqn_epsilon).to(self.device)
                    rnd_d = torch.cat([distributions.OneHotCategorical(
                        logits=torch.ones((batch, size), device=self.device), validate_args=False).sample()
                        for size in self.d_action_sizes], dim=-1)
                    d_action[mask] = rnd_d[mask]
            elif self.d_action_sizes:
                if disable_sample:
                    d_action = d_policy.sample_deter()
                elif use_rnd:
                    d_action = self.rnd_sample_d_action(state, d_policy)
                else:
                    d_action = d_policy.sample()
                d_action = d_action.type(torch.float32)
            else:
                d_action = torch.zeros(0, device=self.device)
            if self.c_action_size:
                if disable_sample:
                    c_action = torch.tanh(c_policy.mean)
                elif use_rnd:
                    c_action = self.rnd_sample_c_action(state, c_policy)
                else:   # `c_policy.sample()` (reference 944) with the Gaussian draw taken from `self.noise`
                    eps = torch.empty_like(c_policy.loc)
                    self.noise.normal_(eps)
                    c_action = torch.tanh(self._rsample(c_policy, eps))
            else:
                c_action = torch.zeros(0, device=self.device)
            d_action, c_action = self._random_action(d_action, c_action)
        else:
            d_action = offline_action[..., :self.d_action_summed_size]
            c_action = offline_action[..., self.d_action_summed_size:]
        prob = torch.ones((batch, self.d_action_summed_size + self.c_action_size), device=self.device)
        if self.d_action_sizes and not self.discrete_dqn_like:
            prob[:, :self.d_action_summed_size] = d_policy.probs
        if self.c_action_size:
            prob[:, self.d_action_summed_size:] = squash_correction_prob(
                c_policy, torch.atanh(torch.clamp(c_action, -0.999, 0.999)))
        if not self.d_action_sizes:
            return c_action, prob
        if not self.c_action_size:
            return d_action, prob
        return torch.cat([d_action, c_action], dim=-1), prob

    @torch.no_grad()
    def choose_action_device(self, obs_list, pre_action, pre_seq_hidden_state, offline_action=None,
                             disable_sample=False, force_rnd_if_available=False):
        """`choose_action` on tensors that already live in HBM (the device-resident `AgentManager`,
        algorithm/agent.py): same computation, no host round trip; returns device tensors."""
        obs_list = list(obs_list)
        self._process_torch_obs_list(obs_list)
        state, next_hidden = self.model_rep([o.unsqueeze(1) for o in obs_list], pre_action.unsqueeze(1),
                                            pre_seq_hidden_state.unsqueeze(1))
        action, prob = self._choose_action(obs_list, state.squeeze(1), offline_action, disable_sample,
                                           force_rnd_if_available)
        return action, prob, next_hidden.squeeze(1)

    @torch.no_grad()
    def choose_action(self, obs_list, pre_action, pre_seq_hidden_state, offline_action=None,
                      disable_sample=False, force_rnd_if_available=False):
        dev = lambda x: torch.from_numpy(x).to(self.device)  # noqa: E731
        action, prob, next_hidden = self.choose_action_device(
            [dev(o) for o in obs_list], dev(pre_action), dev(pre_seq_hidden_state),
            dev(offline_action) if offline_action is not None else None, disable_sample, force_rnd_if_available)
        return action.cpu().numpy(), prob.cpu().numpy(), next_hidden.cpu().numpy()

    @torch.no_grad()
    def choose_attn_action_device(self, ep_indexes, ep_padding_masks, ep_obses_list, ep_pre_actions,
                                  ep_pre_attn_states, offline_action=None, disable_sample=False,
                                  force_rnd_if_available=False):
        """`choose_attn_action` on device tensors (windows of at most `burn_in_step` positions, which is all the
        reference keeps of what it is handed, sac_base.py:1049-1053); returns device tensors."""
        w = self.burn_in_step
        cut = lambda x: x[:, -w:]  # noqa: E731
        obs = [cut(o) for o in ep_obses_list]
        self._process_torch_obs_list(obs)
        state, attn_state, _ = self.model_rep(1, cut(ep_indexes), obs, cut(ep_pre_actions),
                                              pre_seq_hidden_state=cut(ep_pre_attn_states),
                                              is_prev_hidden_state=False, padding_mask=cut(ep_padding_masks))
        action, prob = self._choose_action([o[:, -1] for o in obs], state.squeeze(1), offline_action,
                                           disable_sample, force_rnd_if_available)
        return action, prob, attn_state.squeeze(1)

    @torch.no_grad()
    def choose_attn_action(self, ep_indexes, ep_padding_masks, ep_obses_list, ep_pre_actions,
                           ep_pre_attn_states, offline_action=None, disable_sample=False,
                           force_rnd_if_available=False):
        w = self.burn_in_step
        to = lambda x: torch.from_numpy(x[:, -w:]).to(self.device)  # noqa: E731
        action, prob, attn_state = self.choose_attn_action_device(
            to(ep_indexes), to(ep_padding_masks), [to(o) for o in ep_obses_list], to(ep_pre_actions),
            to(ep_pre_attn_states),
            torch.from_numpy(offline_action).to(self.device) if offline_action is not None else None,
            disable_sample, force_rnd_if_available)
        return action.cpu().numpy(), prob.cpu().numpy(), attn_state.cpu().numpy()

    # ==========================================================================================
    # states (reference sac_base.py:1090-1189)
    # ==========================================================================================
    def get_bnx_data(self, bn_indexes, bn_padding_masks, bn_actions, pre_action_out=None):
        if (bn_indexes.is_cuda and bn_indexes.dtype == torch.int32 and bn_indexes.shape[1] >= 1
                and bn_padding_masks.dtype == torch.bool and bn_actions.dtype == torch.float32
                and bn_indexes.stride(1) == 1 and bn_padding_masks.stride(1) == 1 and bn_actions.stride(2) == 1):
            B, Lm1 = bn_indexes.shape      # one launch instead of six concatenation / fill kernels
            index_x = torch.empty((B, Lm1 + 1), dtype=torch.int32, device=bn_indexes.device)
            pad_x = torch.empty((B, Lm1 + 1), dtype=torch.bool, device=bn_indexes.device)
            pre_action = pre_action_out      # (the static batch's column block beside the vector observations)
            if pre_action is None or pre_action.shape != (B, Lm1 + 1, bn_actions.shape[-1]):
                pre_action = torch.empty((B, Lm1 + 1, bn_actions.shape[-1]), dtype=torch.float32, device=bn_indexes.device)
            native.window_aux(bn_indexes, bn_padding_masks, bn_actions, index_x, pad_x, pre_action)
            return index_x, pad_x, pre_action
        bnx_indexes = torch.concat([bn_indexes, bn_indexes[:, -1:] + (bn_indexes[:, -1:] != -1)], dim=1)
        bnx_padding_masks = torch.concat([bn_padding_masks, bn_padding_masks[:, -1:]], dim=1)
        return bnx_indexes, bnx_padding_masks, gen_n_pre_actions(bn_actions, keep_last_action=True)

    def get_l_states(self, l_indexes, l_padding_masks, l_obses_list, l_pre_actions, l_pre_seq_hidden_states,
                     is_target=False):
        rep = self.model_target_rep if is_target else self.model_rep
        if self.seq_encoder == SEQ_ENCODER.ATTN:
            st, attn, _ = rep(l_indexes.shape[1], l_indexes, l_obses_list, l_pre_actions,
                              l_pre_seq_hidden_states[:, :1], is_prev_hidden_state=True,
                              padding_mask=l_padding_masks)
            return _real(st), _real(attn)
        out = rep(l_obses_list, l_pre_actions, l_pre_seq_hidden_states, padding_mask=l_padding_masks)
        return tuple(_real(o) for o in out) if isinstance(out, tuple) else _real(out)

    # ------------------------------------------------------------------------------------------
    # network evaluation: fused stock path or the user modules
    # ------------------------------------------------------------------------------------------
    def _c_q_values(self, target: bool, state, c_action, obs_list, param_grads=True, select=None):
        """Continuous Q of every ensemble member -> [E, *state.shape[:-1]].  `select` = (window [B, L, S], t)
        when `state` is `window[:, t]`: lets the fused path differentiate into the window directly."""
        fused = self._ftq if target else self._fq
        if fused is not None:
            lead = state.shape[:-1]
            a_rows = StockMLP._rows(c_action, self.c_action_size)
            if select is not None and not target and select[0].dim() == 3 and select[0].is_contiguous():
                out = fused.call_select(select[0], select[1], a_rows, param_grads=param_grads)
            else:
                rows = StockMLP._rows if torch.is_grad_enabled() else StockMLP._rows_in_place
                out = fused(rows(state, self.state_size), a_rows, param_grads=param_grads and not target)
            return out.view(self.ensemble_q_num, *lead)
        models = self.model_target_q_list if target else self.model_q_list
        return torch.stack([q(state, c_action, obs_list)[1] for q in models]).squeeze(-1)

    def _policy(self, state, obs_list):
        """-> (d_policy, c_policy, loc, scale, plain): `plain` says (loc, scale) fully describe the
        continuous head (torch Normal or the fused stock policy), so the fused squash kernels apply;
        c_policy is None on the fused path."""
        if self._fpi is not None:
            lead, A = state.shape[:-1], self.c_action_size
            rows = StockMLP._rows if torch.is_grad_enabled() else StockMLP._rows_in_place
            self._ls = self._fpi(rows(state, self.state_size))[0].view(*lead, 2 * A)   # (loc | scale)
            return None, None, self._ls[..., :A], self._ls[..., A:], True
        self._ls = None
        d_policy, c_policy = self.model_policy(state, obs_list)
        if c_policy is None:
            return d_policy, None, None, None, False
        plain = type(c_policy) is distributions.Normal
        return d_policy, c_policy, c_policy.loc, c_policy.scale, plain

    @staticmethod
    def _rsample(c_policy, eps):
        """reference `Normal.rsample` / `NormalWithPadding.rsample` with externally drawn noise"""
        if hasattr(c_policy, 'padding_mask'):
            keep = ~c_policy.padding_mask
            return c_policy.loc * keep + eps * (c_policy.scale * keep)
        return c_policy.loc + eps * c_policy.scale

    @torch.no_grad()
    def get_l_probs(self, l_obses_list, l_states, l_actions):
        """pi-probability of the stored actions over the window (new mu for the next visit)."""
        d_policy, c_policy, loc, scale, plain = self._policy(l_states, l_obses_list)
        A_all = self.d_action_summed_size + self.c_action_size
        probs = torch.empty((*l_states.shape[:2], A_all), dtype=torch.float32, device=self.device)
        if self.d_action_sizes:
            probs[..., :self.d_action_summed_size] = d_policy.probs
        if self.c_action_size:
            if plain and l_actions.stride(-1) == 1:
                native.squash_prob(loc, scale, l_actions, self.d_action_summed_size, probs,
                                   self.d_action_summed_size)
            else:
                c_act = l_actions[..., self.d_action_summed_size:]
                probs[..., self.d_action_summed_size:] = squash_correction_prob(
                    c_policy, torch.atanh(torch.clamp(c_act, -0.999, 0.999)))
        return probs

    def _return_sample_epilogue(self, jobs, k, ls, nx_actions, eps_buf, policy_sample):
        """What `_get_y` does with the policy's output over `nx_states` right after the forward — rsample / tanh / log-prob,
        pi(stored actions) under `use_n_step_is`, with `policy_sample` the policy step's draw at t = 0 (reference
        sac_base.py:1346-1351, 1430, 1452) — as the EPILOGUE of that forward's launch (`jobs[k]` of
        `native.mlp_forward_multi_sampled`: no elementwise launch behind it).  Where the form applies the step's noise is
        drawn as `_get_y` would have and the launch is issued;  -> ((a_tanh, logp), c_pi) to hand to
        `_get_y(sample=, stored_pi=)`, or None: nothing drawn, nothing launched."""
        A = self.c_action_size
        if not (native.SAMPLE_EPILOGUE and A and 2 * A <= 16 and ls.dim() == 3 and nx_actions.stride(-1) == 1):
            return None
        f32 = dict(dtype=torch.float32, device=self.device)
        a_tanh, logp = torch.empty((*ls.shape[:2], A), **f32), torch.empty(ls.shape[:2], **f32)
        c_pi = torch.empty((*ls.shape[:2], A), **f32) if self.use_n_step_is else None
        epis = [native.sample_epilogue() for _ in jobs]
        epis[k] = native.sample_epilogue(jobs[k], eps_buf, a_tanh, logp, ls.shape[1],
                                         action=nx_actions if self.use_n_step_is else None,
                                         action_offset=self.d_action_summed_size, prob_out=c_pi,
                                         eps2=self._eps_pi if policy_sample else None, t2=0,
                                         a2_out=self._pi_a if policy_sample else None,
                                         logp2_out=self._pi_logp if policy_sample else None)
        if not native.mlp_forward_multi_sampled_ok(jobs, epis):
            return None
        self.noise.normal_(eps_buf)
        if policy_sample:
            self.noise.normal_(self._eps_pi)
            self._pi_sampled = True
        native.mlp_forward_multi_sampled(jobs, epis)
        return (a_tanh, logp), c_pi

    # ==========================================================================================
    # target value (reference _get_y 1297-1466 + _v_trace 1244-1295)
    # ==========================================================================================
    def _vtrace_args(self, n_rewards, n_dones, n_last, n_pad, y_out):
        a = native.VtraceArgs()
        a.reward, a.reward_stride = n_rewards.data_ptr(), n_rewards.stride(0)
        a.done, a.last_mask, a.padding_mask = n_dones.data_ptr(), n_last.data_ptr(), n_pad.data_ptr()
        assert n_dones.stride(0) == n_last.stride(0) == n_pad.stride(0)
        a.mask_stride = n_dones.stride(0)
        a.gamma_ratio, a.lambda_ratio = self._gamma_ratio.data_ptr(), self._lambda_ratio.data_ptr()
        a.gamma, a.v_rho, a.v_c = self.gamma, self._v_rho_f, self._v_c_f
        a.use_n_step_is, a.B, a.n = int(self.use_n_step_is), n_rewards.shape[0], self.n_step
        a.y_out = y_out.data_ptr()
        return a

    @torch.no_grad()
    def _get_y(self, n_last_masks, n_padding_masks, nx_obses_list, nx_states, nx_actions, n_rewards,
               n_dones, n_mu_probs, *, eps_buf, subset_prefix, y_out, q_online=None, td_out=None, ls=None,
               sample=None, stored_pi=None, policy_sample=False, q_table=None):
        """-> (d_y [B,1] | None, c_y [B,1] | None).

        Stock continuous-only networks may hand over work they already did: `ls` = the policy's
        [B, n+1, 2A] (loc | scale) output for `nx_states`; `sample` = (a_tanh [B, n+1, A], logp [B, n+1])
        already drawn from it with `eps_buf`; `stored_pi` = pi(stored actions) [B, n+1, A].  With
        `policy_sample` the launch that samples here also draws the policy step's action for t = 0
        (`self._pi_a`, `self._pi_logp`; same policy parameters, same state).  `q_table` = the target
        ensemble's values [E, B, n+1] on (nx_states, a_tanh) if already computed.

        `nx_actions` is the stored-action window [B, n+1, A] (the reference appends a zero row
        instead, 1329: the extra row's probability is discarded either way).  With `q_online`
        ([E, B], continuous-only action spaces) the TD error mean_e|q_e - y| is produced by the
        same launch into `td_out`.
        """
        dsum = self.d_action_summed_size
        n_actions = nx_actions[:, :-1]
        if ls is None and sample is None and self._fpi is not None and self.c_action_size and nx_states.dim() == 3:
            # the stock policy over the window and the window's sample as ONE launch (`_return_sample_epilogue`)
            job_pi, ls_out = self._fpi.job(StockMLP._rows_in_place(nx_states, self.state_size), None)
            ls_try = ls_out[0].view(*nx_states.shape[:2], 2 * self.c_action_size)
            pre = self._return_sample_epilogue([job_pi], 0, ls_try, nx_actions, eps_buf, policy_sample)
            if pre is not None:
                ls, (sample, stored_pi) = ls_try, pre
        if ls is not None:
            d_policy = c_policy = None
            loc, scale, plain = ls[..., :self.c_action_size], ls[..., self.c_action_size:], True
        else:
            d_policy, c_policy, loc, scale, plain = self._policy(nx_states, nx_obses_list)
            ls = self._ls
        self._ls_y = ls    # kept for the policy step (same parameters, state at t = 0 of this window)

        if self.curiosity is not None:   # 1333-1343: augments the sampled reward window in place
            n_states, next_n_states = nx_states[:, :-1], nx_states[:, 1:]
            if self.curiosity == CURIOSITY.FORWARD:
                approx, actual = self.model_forward_dynamic(n_states, n_actions), next_n_states
            else:
                approx, actual = self.model_inverse_dynamic(n_states, next_n_states), n_actions
            if (self._fused_curiosity and approx.is_cuda and approx.dim() == 3 and approx.dtype == torch.float32
                    and actual.stride(-1) == 1 and n_rewards.dim() == 2 and n_rewards.dtype == torch.float32
                    and (n_rewards.stride(1) == 1 or n_rewards.shape[1] == 1) and not torch.is_grad_enabled()):
                native.curiosity_bonus(approx.contiguous(), actual, n_rewards, self.curiosity_strength)   # one launch
            else:
                d = approx - actual
                bonus = torch.sum(d.mul_(d), dim=-1).mul_(0.5)       # 0.5 * sum (approx - actual)^2
                n_rewards.add_(bonus, alpha=self.curiosity_strength)

        logp = None
        if self.c_action_size and sample is not None:
            a_tanh, logp = sample
            c_pi = stored_pi
        elif self.c_action_size:
            self.noise.normal_(eps_buf)
            c_pi = None
            if plain:   # one launch: rsample, tanh, log-prob and the stored-action probabilities
                a_tanh = torch.empty(loc.shape, dtype=torch.float32, device=self.device)
                logp = torch.empty(loc.shape[:-1], dtype=torch.float32, device=self.device)
                if policy_sample and ls is not None and ls.dim() == 3:
                    A = self.c_action_size
                    self.noise.normal_(self._eps_pi)
                    if self.use_n_step_is:
                        c_pi = torch.empty(loc.shape, dtype=torch.float32, device=self.device)
                    native.squash_multi([
                        native.squash_job(loc, scale, eps_buf, a_tanh, logp, nx_actions if self.use_n_step_is else None,
                                          dsum, c_pi, 0),
                        native.squash_job(ls[:, 0, :A], ls[:, 0, A:], self._eps_pi, self._pi_a, self._pi_logp)])
                    self._pi_sampled = True
                elif self.use_n_step_is:
                    c_pi = torch.empty(loc.shape, dtype=torch.float32, device=self.device)
                    native.squash_sample_fwd(loc, scale, eps_buf, a_tanh, logp, None, nx_actions, dsum, c_pi, 0)
                else:
                    native.squash_sample_fwd(loc, scale, eps_buf, a_tanh, logp)
            else:
                sampled = self._rsample(c_policy, eps_buf)
                a_tanh = torch.tanh(sampled)
                logp = sum_log_prob(squash_correction_log_prob(c_policy, sampled))
        else:
            a_tanh = torch.zeros(0, device=self.device)

        d_y = c_y = None
        E, Es = self.ensemble_q_num, self.ensemble_q_sample
        nx_qs = None
        if self.d_action_sizes:
            nx_qs = [q(nx_states, a_tanh, nx_obses_list) for q in self.model_target_q_list]

        if self.d_action_sizes and self.discrete_dqn_like:   # 1363-1382: double-DQN target, no policy
            sub_next, sub_eval = self._subsets[subset_prefix + '_dnext'], self._subsets[subset_prefix + '_dn']
            self.noise.subset_(sub_next, E)
            target_next = torch.stack([q[0][:, 1:] for q in nx_qs]).index_select(0, sub_next.long())
            next_c = a_tanh[:, 1:] if self.c_action_size else a_tanh
            next_obs = [o[:, 1:] for o in nx_obses_list]
            eval_next = torch.stack([q(nx_states[:, 1:], next_c, next_obs)[0] for q in self.model_q_list])
            self.noise.subset_(sub_eval, E)
            d_y = self.get_dqn_like_d_y(n_last_masks, n_padding_masks, n_rewards, n_dones,
                                        eval_next.index_select(0, sub_eval.long()), target_next)
        elif self.d_action_sizes:   # 1383-1421, policy-based branch, eager ops + the scan kernel
            sub_next, sub_n = self._subsets[subset_prefix + '_dnext'], self._subsets[subset_prefix + '_dn']
            self.noise.subset_(sub_next, E)
            self.noise.subset_(sub_n, E)
            stacked = torch.stack([q[0] for q in nx_qs])                      # [E, B, n+1, D]
            mean_next = stacked[:, :, 1:].index_select(0, sub_next.long()).mean(0)
            mean_n = stacked[:, :, :-1].index_select(0, sub_n.long()).mean(0)
            probs = d_policy.probs
            n_p, next_p = probs[:, :-1], probs[:, 1:]
            d_alpha = torch.exp(self.log_d_alpha)
            v_n = torch.sum(n_p * (mean_n - d_alpha * torch.log(n_p.clamp(min=1e-8))), -1) / self.d_action_branch_size
            v_next = torch.sum(next_p * (mean_next - d_alpha * torch.log(next_p.clamp(min=1e-8))), -1) \
                / self.d_action_branch_size
            mu = pi = None
            if self.use_n_step_is:
                mu = n_mu_probs[..., :dsum] * n_actions[..., :dsum]
                mu = torch.where(mu == 0., torch.ones_like(mu), mu).prod(-1).contiguous()
                pi = torch.exp(d_policy.log_prob(nx_actions[..., :dsum]).sum(-1))[:, :-1].contiguous()
            d_y = torch.empty(n_rewards.shape[0], dtype=torch.float32, device=self.device)
            args = self._vtrace_args(n_rewards, n_dones, n_last_masks, n_padding_masks, d_y)
            native.vtrace_return_direct(args, v_n.contiguous(), v_next.contiguous(), pi, mu)
            d_y = d_y.unsqueeze(-1)

        if self.c_action_size:    # 1423-1464, fused
            sub_n, sub_next = self._subsets[subset_prefix + '_cn'], self._subsets[subset_prefix + '_cnext']
            self.noise.subset_(sub_n, E)
            self.noise.subset_(sub_next, E)
            if q_table is not None:
                q_tab = q_table
            elif nx_qs is not None:
                q_tab = torch.stack([q[1] for q in nx_qs]).squeeze(-1)        # [E, B, n+1]
            else:
                q_tab = self._c_q_values(True, nx_states, a_tanh, nx_obses_list)
            args = self._vtrace_args(n_rewards, n_dones, n_last_masks, n_padding_masks, y_out)
            args.q = q_tab.data_ptr()
            args.q_stride_e, args.q_stride_b, args.q_stride_t = q_tab.stride(0), q_tab.stride(1), q_tab.stride(2)
            # (the whole ensemble: the minimum does not depend on the order, no index loads in front of the values)
            args.subset_n, args.subset_next, args.E_sample = (sub_n.data_ptr(), sub_next.data_ptr(), Es) if Es != E else (None, None, Es)
            logp = logp.contiguous()
            args.logp, args.log_alpha = logp.data_ptr(), self.log_c_alpha.data_ptr()
            if self.use_n_step_is:
                if not plain:
                    c_pi = squash_correction_prob(
                        c_policy, torch.atanh(torch.clamp(nx_actions[..., dsum:], -0.999, 0.999))).contiguous()
                args.pi_prob, args.pi_stride_b, args.pi_stride_t = c_pi.data_ptr(), c_pi.stride(0), c_pi.stride(1)
                args.mu_prob, args.mu_stride_b, args.mu_stride_t = \
                    n_mu_probs.data_ptr(), n_mu_probs.stride(0), n_mu_probs.stride(1)
                args.mu_offset, args.A = dsum, self.c_action_size
            if q_online is not None and not self.d_action_sizes:
                args.q_online, args.E_online, args.td_error_out = q_online.data_ptr(), q_online.shape[0], td_out.data_ptr()
            sidecars = None
            if q_online is not None and self._vtrace_sidecars:     # the TD error's launch hosts the pending write-backs
                sidecars, self._vtrace_sidecars = self._vtrace_sidecars, None
            if self._defer_return and q_online is None and sidecars is None:
                # (the caller's Q-loss backward forms this return itself; everything `args` points to is the caller's)
                self._deferred_return = (args, (q_tab, logp, c_pi, sub_n, sub_next))
            elif q_online is not None and self._td_update_with is not None and args.td_error_out:
                (rb, ids), self._td_update_with = self._td_update_with, None
                rb.td_update(args, ids, sidecars=sidecars, alpha_step=self._pending_alpha)
                self._pending_alpha = None
            else:
                native.vtrace_return_min(args, sidecars=sidecars,
                                         pending_alpha=self._pending_alpha if q_online is not None else None)
            c_y = y_out.unsqueeze(-1)
        return d_y, c_y

    # ==========================================================================================
    # losses / updates (reference _train_rep_q 1468-1605, _train_policy 1841-1911, _train_alpha 1913-1949)
    # ==========================================================================================
    @torch.no_grad()
    def _train_rep_q_stock(self, n_last_masks, n_padding_masks, nx_obses_list, nx_states, nx_actions, n_rewards,
                           n_dones, n_mu_probs, priority_is, policy_sample):
        """Q step (reference 1468-1605) for stock networks under a parameter-free representation as an
        explicit kernel chain: target Q of the stored (s0, a0) pair -> return target -> [online Q, clipped
        double-Q loss, backward] in one launch -> [tile reduction + Adam] in one launch."""
        E, B = self.ensemble_q_num, nx_states.shape[0]
        x0 = StockMLP._rows(nx_states[:, 0], self.state_size)
        a0 = StockMLP._rows(nx_actions[:, 0], self.c_action_size)
        xs = StockMLP._rows(nx_states, self.state_size)
        job_tq, _ = self._ftq.job(x0, a0, out=self._tq_buf)
        job_pi, ls = self._fpi.job(xs, None)
        T, A = nx_states.shape[1], self.c_action_size
        fused = None
        if self._fused_forward_chain and self.curiosity is None:
            # policy over the window -> target sample (+ pi(stored actions), + the policy step's sample at t = 0) ->
            # target critics on the sampled actions, with the target Q of the stored pair riding along: ONE launch
            # (bit-identical to the three it replaces)
            f32 = dict(dtype=torch.float32, device=self.device)
            a_y, logp_y = torch.empty((B, T, A), **f32), torch.empty((B, T), **f32)
            c_pi = torch.empty((B, T, A), **f32) if self.use_n_step_is else None
            job_q, q_tab = self._ftq.job(xs, a_y.view(-1, A))
            fused = native.pi_q_job(job_pi, job_q, self._eps_y, a_y, logp_y, T,
                                    action=nx_actions if self.use_n_step_is else None,
                                    action_offset=self.d_action_summed_size, prob_out=c_pi,
                                    eps2=self._eps_pi if policy_sample else None, t2=0,
                                    a2_out=self._pi_a, logp2_out=self._pi_logp)
            if not native.policy_sample_q_forward_ok(fused):
                fused = None
        if fused is not None:
            self.noise.normal_(self._eps_y)
            if policy_sample:
                self.noise.normal_(self._eps_pi)
                self._pi_sampled = True
            native.policy_sample_q_forward(fused, [job_tq], sidecars=self._take_la_gather())
            ls = ls[0].view(B, T, 2 * A)
            # the return target is formed by the Q-loss backward's own workgroups where that form applies: its
            # arguments are assembled as always, its launch is not issued
            self._defer_return, self._deferred_return = self._fused_q_return, None
            _, c_y = self._get_y(n_last_masks, n_padding_masks, nx_obses_list, nx_states, nx_actions, n_rewards,
                                 n_dones, n_mu_probs if self.use_n_step_is else None, eps_buf=self._eps_y,
                                 subset_prefix='y', y_out=self._y_buf, ls=ls, sample=(a_y, logp_y), stored_pi=c_pi,
                                 q_table=q_tab.view(E, B, T))
            self._defer_return = False
        else:
            # one launch: target Q of the stored pair (for the clipped loss) beside the policy over the window — whose
            # lanes also draw the window's sample (`_return_sample_epilogue`; otherwise an elementwise launch in `_get_y`)
            ls = ls[0].view(*nx_states.shape[:2], 2 * self.c_action_size)
            pre = self._return_sample_epilogue([job_tq, job_pi], 1, ls, nx_actions, self._eps_y, policy_sample)
            if pre is not None:
                _, c_y = self._get_y(n_last_masks, n_padding_masks, nx_obses_list, nx_states, nx_actions, n_rewards,
                                     n_dones, n_mu_probs if self.use_n_step_is else None, eps_buf=self._eps_y,
                                     subset_prefix='y', y_out=self._y_buf, ls=ls, sample=pre[0], stored_pi=pre[1])
            else:
                native.mlp_forward_multi([job_tq, job_pi])
                _, c_y = self._get_y(n_last_masks, n_padding_masks, nx_obses_list, nx_states, nx_actions, n_rewards,
                                     n_dones, n_mu_probs if self.use_n_step_is else None, eps_buf=self._eps_y,
                                     subset_prefix='y', y_out=self._y_buf, policy_sample=policy_sample, ls=ls)
        w = priority_is.reshape(-1).contiguous() if priority_is is not None else None
        # loss + backward in one launch (the backward recomputes the forward on chip anyway); on a single
        # GPU the tile reduction of the parameter gradients is folded into the Adam launch
        opt = self.optimizer_q_list[0]
        fold = self._dist is None and opt.start == self._fq._start and self._params.span('rep')[0] == self._params.span('rep')[1]
        ret, self._deferred_return = self._deferred_return, None
        if ret is not None and self._fq.backward_qloss_return_ok(B, ret[0]):
            self._fq.backward_qloss_return(x0, a0, self._tq_buf.view(E, B), ret[0], w, self.clip_epsilon, self._loss_q_e,
                                           defer=fold)
        else:
            if ret is not None:
                native.vtrace_return_min(ret[0])
            self._fq.backward_qloss(x0, a0, self._tq_buf.view(E, B), c_y.reshape(-1), w, self.clip_epsilon,
                                    self._loss_q_e, defer=fold)
        if fold:
            self._fq.adam_partials(opt, loss_out=self._loss_q_e)
        else:
            self._finish_rep_q(None, None)

    def _train_rep_q(self, n_last_masks, n_padding_masks, nx_obses_list, nx_states, nx_actions, n_rewards,
                     n_dones, n_mu_probs, priority_is, aux=None, policy_sample=False, state_base=None):
        """`policy_sample`: see `_get_y`.  `state_base` = (window states [B, L, S], t) with
        `nx_states[:, 0] == window[:, t]`.  `aux` (only with siamese / prediction heads): dict(n_indexes, n_pre_actions,
        n_pre_seq_hidden_states, nx_target_states) for the auxiliary losses of reference 1577-1600."""
        if (self._stock_c_only() and aux is None and self.clip_epsilon > 0 and not nx_states.requires_grad
                and self.optimizer_rep is None):
            return self._train_rep_q_stock(n_last_masks, n_padding_masks, nx_obses_list, nx_states, nx_actions,
                                           n_rewards, n_dones, n_mu_probs, priority_is, policy_sample)
        dsum = self.d_action_summed_size
        obs_list = [o[:, 0] for o in nx_obses_list]
        state, action = nx_states[:, 0], nx_actions[:, 0]
        if state_base is not None and not nx_states.requires_grad:
            state = state_base[0][:, state_base[1]]     # (`_step_rep_and_q`: the differentiable pass covered this position only)
        d_action, c_action = action[..., :dsum], action[..., dsum:]

        if (self._fused_q_state_grads and self._fq is not None and self._ftq is not None and not self.d_action_sizes
                and self.c_action_size and self.clip_epsilon > 0 and aux is None and state_base is not None
                and state_base[0].dim() == 3 and state.requires_grad):
            # stock critics behind a trainable representation: target Q of the stored pair -> return target -> ONE
            # launch for [critics forward, clipped double-Q loss, backward] that also returns d loss / d state (the
            # separate critic forward and loss launches of the autograd form disappear); the representation's backward
            # continues from that gradient
            base, t = state_base
            with torch.no_grad():
                x0 = StockMLP._rows(base.detach()[:, t], self.state_size)
                a0 = StockMLP._rows(c_action, self.c_action_size)
                # one launch: target Q of the stored pair (for the clipped loss) beside the policy over the window
                states_y = nx_states.detach()
                job_tq, t_q = self._ftq.job(x0, a0, out=self._tq_buf)
                job_pi, ls_y = self._fpi.job(StockMLP._rows_in_place(states_y, self.state_size), None)
                ls_y = ls_y[0].view(*states_y.shape[:2], 2 * self.c_action_size)
                # ... whose lanes also draw the window's sample (`_return_sample_epilogue`): no elementwise launch behind it
                pre = self._return_sample_epilogue([job_tq, job_pi], 1, ls_y, nx_actions, self._eps_y, policy_sample)
                sampled = pre is not None
                if not sampled:
                    native.mlp_forward_multi([job_tq, job_pi])
                self._defer_return, self._deferred_return = self._fused_q_return, None
                _, c_y = self._get_y(n_last_masks, n_padding_masks, nx_obses_list, states_y, nx_actions,
                                     n_rewards, n_dones, n_mu_probs if self.use_n_step_is else None,
                                     eps_buf=self._eps_y, subset_prefix='y', y_out=self._y_buf,
                                     policy_sample=policy_sample and not sampled, ls=ls_y,
                                     sample=pre[0] if sampled else None, stored_pi=pre[1] if sampled else None)
                self._defer_return = False
                w = priority_is.reshape(-1).contiguous() if priority_is is not None else None
                ret, self._deferred_return = self._deferred_return, None
                # one GPU: the critics' tile reduction is folded into their Adam launch (as without a trainable
                # representation), which then no longer waits for the representation's backward
                opt = self.optimizer_q_list[0]
                fold = self._dist is None and self._fold_rep_q_adam and self._rep_q_adam_alike()
                if ret is not None and self._fq.backward_qloss_return_ok(x0.shape[-2], ret[0]):
                    # (short windows: the return target is formed by the backward's own workgroups, no launch of its own)
                    g0 = self._fq.backward_qloss_return(x0, a0, t_q.view(self.ensemble_q_num, -1), ret[0], w,
                                                        self.clip_epsilon, self._loss_q_e, state_grads=True, defer=fold)
                else:
                    if ret is not None:
                        native.vtrace_return_min(ret[0])
                    g0 = self._fq.backward_qloss(x0, a0, t_q.view(self.ensemble_q_num, -1), c_y.reshape(-1), w,
                                                 self.clip_epsilon, self._loss_q_e, state_grads=True, defer=fold)
                if fold:
                    self._fq.adam_partials(opt, loss_out=self._loss_q_e)
                # d loss / d (window states): zero except at position t — a buffer that stays zero elsewhere, so only the
                # slice is written each step (no memset launch)
                g_base = self._g_state_base
                if g_base is None or g_base.shape != base.shape:
                    g_base = self._g_state_base = torch.zeros_like(base)
                at_position = self._gru_backward_at and fused_gru.is_fused_top(base)
                from_head = not at_position and self._head_sums_members and fused_linear.is_fused_head_output(base)
                if not (at_position or from_head):
                    torch.sum(g0, dim=0, out=g_base[:, t])
            rep_stepped = False
            # (the backward launches' second launches — per-workgroup partials summed into the flat gradient — run as one
            # launch when the walk is done, in front of the optimizer step: fused_mlp.DeferredPartialSums)
            with direct_param_grads(), DeferredPartialSums() as sums_later:
                if at_position:
                    # the window IS a fused GRU's output: its backward sums the members' gradients itself and starts at
                    # position t (the steps behind it only feed detached targets) — no member-sum launch in between;
                    # where that GRU is all the representation has, the launch finishing its gradients steps it too
                    rep_stepped = fused_gru.backward_from_position(base, g0, t, g_base,
                                                                   adam=self._rep_adam_epilogue() if fold else None)
                elif from_head:
                    # the window is a fused Linear + Tanh head's output: its backward launch sums the members
                    fused_linear.backward_from_members(base, g0, t, g_base)
                else:
                    torch.autograd.backward([base], [g_base])
            sums_later.flush()
            if fold:
                if not rep_stepped:
                    self.optimizer_q_list[0].step(*self._params.span('rep'))
                return
            return self._finish_rep_q(None, None)
        q_list = None
        if self.d_action_sizes:
            q_list = [q(state, c_action, obs_list) for q in self.model_q_list]
            c_q = torch.stack([q[1] for q in q_list]).squeeze(-1) if self.c_action_size else None
        else:
            c_q = self._c_q_values(False, state, c_action, obs_list, select=state_base)   # [E, B]
        d_y, c_y = self._get_y(n_last_masks, n_padding_masks, nx_obses_list, nx_states.detach(), nx_actions,
                               n_rewards, n_dones, n_mu_probs if self.use_n_step_is else None,
                               eps_buf=self._eps_y, subset_prefix='y', y_out=self._y_buf,
                               policy_sample=policy_sample)

        losses = None
        if self.d_action_sizes:
            qs = torch.stack([torch.sum(d_action * q[0], dim=-1, keepdim=True) / self.d_action_branch_size
                              for q in q_list])                                   # [E, B, 1]
            losses = functional.mse_loss(qs, d_y.expand_as(qs), reduction='none')
        if self.c_action_size:
            if self.clip_epsilon > 0:
                with torch.no_grad():
                    t_q = self._c_q_values(True, state.detach(), c_action, obs_list)
                if losses is None and (aux is None or self._fused_q_loss_with_aux):
                    # loss value and d(sum_e l_e)/dq from one launch; back-propagation starts at q (with auxiliary heads
                    # too: they only need the graph kept for the prediction models' second pass and the main gradients
                    # in place before they gate theirs)
                    w = priority_is.reshape(-1).contiguous() if priority_is is not None else None
                    native.q_loss_fwd_bwd(c_q.detach().contiguous(), t_q.contiguous(), c_y.reshape(-1), w,
                                          self.clip_epsilon, self._loss_q_e, self._grad_q)
                    with direct_param_grads(), DeferredPartialSums() as sums_later:
                        torch.autograd.backward([c_q], [self._grad_q],
                                                retain_graph=aux is not None and self.use_prediction)
                    sums_later.flush()
                    if aux is None:
                        return self._finish_rep_q(None, None)
                    return self._finish_rep_q(None, None, aux,
                                              dict(n_padding_masks=n_padding_masks, nx_obses_list=nx_obses_list,
                                                   nx_states=nx_states, nx_actions=nx_actions, n_rewards=n_rewards))
                clipped = t_q + torch.clamp(c_q - t_q, -self.clip_epsilon, self.clip_epsilon)
                yv = c_y.reshape(1, -1)
                c_loss = torch.maximum((clipped - yv) ** 2, (c_q - yv) ** 2).unsqueeze(-1)
            else:   # 1556: `+=` of (self + mse) doubles the running loss
                c_loss = functional.mse_loss(c_q, c_y.reshape(1, -1).expand_as(c_q), reduction='none').unsqueeze(-1)
                losses = losses * 2 if losses is not None else None
            losses = c_loss if losses is None else losses + c_loss
        if priority_is is not None:
            losses = losses * priority_is.unsqueeze(0)
        loss_q_list = losses.mean(dim=(1, 2))
        return self._finish_rep_q(loss_q_list.sum(), loss_q_list[0], aux,
                                  dict(n_padding_masks=n_padding_masks, nx_obses_list=nx_obses_list,
                                       nx_states=nx_states, nx_actions=nx_actions, n_rewards=n_rewards))

    def _rep_q_adam_alike(self) -> bool:
        """the representation's and the critics' optimizers have the same hyper-parameters: only then may ONE Adam launch
        step both spans (`fold_rep_q_adam`); a schedule or a user edit that moves one of them un-folds the step"""
        a, b = self.optimizer_rep, self.optimizer_q_list[0]
        return a is None or (a.lr, tuple(a.betas), a.eps) == (b.lr, tuple(b.betas), b.eps)

    def _rep_adam_epilogue(self):
        """-> `native.adam_epilogue` for the representation's parameters when they are exactly one fused GRU layer's
        cell weights (then the launch that finishes their gradients can step them), else None"""
        hp = (self.optimizer_rep.lr, *self.optimizer_rep.betas, self.optimizer_rep.eps) if self.optimizer_rep is not None else None
        if self._rep_epilogue is False or self._rep_epilogue_hp != hp:      # (a changed learning rate rebuilds it)
            self._rep_epilogue, self._rep_epilogue_hp = None, hp
            from .nn_models.layers.seq_layers import GRU
            grus = [m for m in self.model_rep.modules() if isinstance(m, GRU)]
            opt, (s, e) = self.optimizer_q_list[0], self._params.span('rep')
            if (len(grus) == 1 and grus[0]._fusable and self.optimizer_rep is not None and e > s
                    and {id(p) for p in grus[0].parameters()} == {id(p) for p in self.model_rep.parameters()}):
                g = self._params
                self._rep_epilogue = native.adam_epilogue(g.flat[s:e], g.grad[s:e], opt.exp_avg[s:e], opt.exp_avg_sq[s:e],
                                                          self.optimizer_rep.lr, *self.optimizer_rep.betas,
                                                          self.optimizer_rep.eps, opt.steps_done)
        return self._rep_epilogue

    def _finish_rep_q(self, total_loss, loss_q0, aux=None, ctx=None):
        if total_loss is not None:
            # the prediction heads differentiate the representation graph again (sac_aux._train_rpm)
            # the main loss accumulates into every parameter it reaches (representation + Q ensemble): the fused
            # layers add their parameter gradients in place
            with direct_param_grads(), DeferredPartialSums() as sums_later:
                total_loss.backward(retain_graph=aux is not None and self.use_prediction)
            sums_later.flush()
            self._stats['loss_q'].copy_(loss_q0.detach())
        start, stop = self._params.span('rep', f'q_{self.ensemble_q_num - 1}')
        if aux is None:
            if self._dist is not None:
                self._dist.all_reduce_grads(self._params.grad, start, stop)
            # Q optimizers then the representation optimizer (1589-1603): adjacent segments, one launch
            self.optimizer_q_list[0].step(start, stop)
            return
        # with auxiliary heads the reference order matters: siamese -> Q step -> prediction -> rep step
        grads_rep_main = [p.grad.detach() for p in self.model_rep.parameters()]
        grads_q_main = [[p.grad.detach() for p in q.parameters()] for q in self.model_q_list]
        if self.siamese is not None:
            n_obs = [o[:, :-1] for o in ctx['nx_obses_list']]
            self._train_siamese_representation_learning(grads_rep_main, grads_q_main, aux['n_indexes'],
                                                        ctx['n_padding_masks'], n_obs, aux['n_pre_actions'],
                                                        aux['n_pre_seq_hidden_states'])
        # data parallel: every segment is averaged over the ranks AFTER the auxiliary heads have added their (gated)
        # gradients to it and right before its optimizer steps, so the replicas stay identical
        q_start = self._params.span('q_0')[0]
        if self._dist is not None:
            self._dist.all_reduce_grads(self._params.grad, q_start, stop)
        self.optimizer_q_list[0].step(q_start, stop)
        if self.use_prediction:
            self._train_rpm(grads_rep_main, ctx['nx_obses_list'], ctx['nx_states'], aux['nx_target_states'],
                            ctx['nx_actions'][:, :-1], ctx['n_rewards'])
        if self.optimizer_rep is not None:
            if self._dist is not None:
                self._dist.all_reduce_grads(self._params.grad, *self._params.span('rep'))
            self.optimizer_rep.step()

    def _stock_c_only(self) -> bool:
        """Stock Q ensemble and stock policy on a continuous-only action space: the whole policy / alpha
        / write-back chain runs on libasac_hip kernels without autograd."""
        return (self._fpi is not None and self._fq is not None and bool(self.c_action_size)
                and not self.d_action_sizes and not (self.offline_enabled and self.offline_loss))

    @torch.no_grad()
    def _train_policy_stock(self, state, ls=None):
        """Policy step (reference 1841-1911) for the stock networks as an explicit kernel chain:
        [policy forward] -> rsample/tanh/log-prob -> Q ensemble forward -> objective + its gradients
        -> Q backward (input gradients only) -> sampling backward (sums the members' action gradients)
        -> policy backward -> Adam.  `ls` = the policy's [B, 2A] (loc | scale) output for `state` if
        the caller already has it (same parameters, same input)."""
        A, E = self.c_action_size, self.ensemble_q_num
        x = StockMLP._rows(state, self.state_size)
        B = x.shape[0]
        fusable = (self._fused_policy_step and self.ensemble_q_sample == 2 and self._fpi.policy_step_fused_ok(self._fq, B))
        sample_out = None
        if ls is None and not self._pi_sampled and fusable:
            # nothing of the policy's forward is at hand (trainable representation: the state is new): the one-launch
            # policy step samples the action itself
            f32 = dict(dtype=torch.float32, device=self.device)
            self.noise.normal_(self._eps_pi)
            sample_out = (torch.empty((B, A), **f32), torch.empty(B, **f32), torch.empty((B, 2 * A), **f32))
            a_tanh, logp, scale = sample_out[0], sample_out[1], sample_out[2][:, A:]
        else:
            if ls is None:
                ls = self._fpi._launch_forward(x, None)[0]
            loc, scale = ls[..., :A], ls[..., A:]
            if self._pi_sampled:        # drawn by the target computation's launch from the same `ls`
                a_tanh, logp, self._pi_sampled = self._pi_a, self._pi_logp, False
            else:
                self.noise.normal_(self._eps_pi)
                a_tanh = torch.empty((B, A), dtype=torch.float32, device=self.device)
                logp = torch.empty(B, dtype=torch.float32, device=self.device)
                native.squash_sample_fwd(loc, scale, self._eps_pi, a_tanh, logp)
        sub = self._subsets['pi_c']
        self.noise.subset_(sub, E)
        self._pi_stats_src = (logp, scale)
        opt = self.optimizer_policy
        fold = self._dist is None and (opt.start, opt.stop) == (self._fpi._start, self._fpi._start + self._fpi.member_stride)
        if fusable and a_tanh.is_contiguous():
            # two critics sampled (of two or more): critics forward, objective gradient, critics backward to the action,
            # sampling backward and policy backward in ONE launch (bit-identical to the chain below)
            self._fpi.policy_step_fused(self._fq, x, None if sample_out is not None else a_tanh, self._eps_pi,
                                        self.log_c_alpha, q_out=self._pi_q, defer=fold, subset=sub if E != 2 else None,
                                        sample_out=sample_out)
        else:
            c_qs = self._fq._launch_forward(x, a_tanh, out=self._pi_q)                   # [E, B, 1]
            # objective gradients formed on chip: dL/dq inside the Q backward (from the value table), dL/dlogp =
            # alpha / B inside the sampling backward; the logged statistics are computed on demand
            g_a = self._fq.backward_policy_q(x, a_tanh, c_qs.view(E, B), sub if self.ensemble_q_sample != E else None,
                                             self.ensemble_q_sample)                    # [E, B, A]
            # sampling backward (sums the members' action gradients) + policy backward in one launch
            self._fpi.backward_policy_sample(x, self._eps_pi, g_a, self.log_c_alpha, defer=fold)
        if fold:
            self._fpi.adam_partials(opt)
            return
        if self._dist is not None:
            self._dist.all_reduce_grads(self._params.grad, *self._params.span('policy'))
        opt.step()

    def _train_policy(self, obs_list, state, action, mu_d_policy_probs, ls=None):
        if self._stock_c_only():
            return self._train_policy_stock(state, ls)
        dsum, E = self.d_action_summed_size, self.ensemble_q_num
        d_policy, c_policy, loc, scale, plain = self._policy(state, obs_list)
        loss_d = loss_c = None
        with torch.no_grad():
            d_alpha = torch.exp(self.log_d_alpha) if self.d_action_sizes else None
            c_alpha = torch.exp(self.log_c_alpha) if self.c_action_size else None

        if self.d_action_sizes and self.discrete_dqn_like and not self.c_action_size:
            with torch.no_grad():   # nothing to optimise (1905); keep the logged entropy
                self._stats['d_entropy'].copy_(torch.mean(d_policy.entropy().sum(-1) / self.d_action_branch_size))
            return
        if self.d_action_sizes and not self.discrete_dqn_like:
            probs = d_policy.probs
            c_action = action[..., dsum:]
            d_qs = torch.stack([q(state, c_action, obs_list)[0] for q in self.model_q_list])
            sub = self._subsets['pi_d']
            self.noise.subset_(sub, E)
            mean_q = d_qs.index_select(0, sub.long()).mean(0)
            inner = d_alpha * torch.log(probs.clamp(min=1e-8)) - mean_q.detach()
            loss_d = torch.sum(probs * inner, dim=1, keepdim=True) / self.d_action_branch_size
            mu_ent = -torch.sum(mu_d_policy_probs * torch.log(mu_d_policy_probs.clamp(min=1e-8)), dim=-1) \
                / self.d_action_branch_size
            pi_ent = d_policy.entropy().sum(-1) / self.d_action_branch_size
            loss_d = loss_d + self.d_policy_entropy_penalty * (torch.pow(mu_ent - pi_ent, 2.) / 2.).unsqueeze(-1)

        # restricted to the policy: its parameters (and the fused policy's anchor); the Q ensemble only passes the
        # action gradient through (param_grads=False), so the direct accumulation reaches the policy alone
        pi_inputs = list(self.model_policy.parameters()) + ([self._fpi._anchor] if self._fpi is not None else [])
        if self.c_action_size and not self.d_action_sizes and plain and not (self.offline_enabled and self.offline_loss):
            # continuous-only fast path: objective, its gradients and the entropy statistic from one
            # launch; back-propagation starts at (logp, q) with the kernel-produced gradients
            self.noise.normal_(self._eps_pi)
            if self._ls is not None:
                a_tanh, logp = squash_sample_ls(self._ls, self._eps_pi)
            else:
                a_tanh, logp = squash_sample(loc, scale, self._eps_pi)
            c_qs = self._c_q_values(False, state, a_tanh, obs_list, param_grads=False)        # [E, B]
            sub = self._subsets['pi_c']
            self.noise.subset_(sub, E)
            native.policy_loss_fwd_bwd(logp.detach(), c_qs.detach().contiguous(),
                                       sub if self.ensemble_q_sample != E else None, self.ensemble_q_sample,
                                       self.log_c_alpha, scale.detach(), self._stats['loss_policy'],
                                       self._grad_logp, self._grad_q, self._stats['c_entropy'])
            with direct_param_grads(only=pi_inputs):
                torch.autograd.backward([logp, c_qs], [self._grad_logp, self._grad_q], inputs=pi_inputs)
            if self._dist is not None:
                self._dist.all_reduce_grads(self._params.grad, *self._params.span('policy'))
            self.optimizer_policy.step()
            return

        if self.c_action_size:
            self.noise.normal_(self._eps_pi)
            if plain:
                a_tanh, logp = squash_sample(loc, scale, self._eps_pi)
                logp = logp.unsqueeze(-1)
            else:
                sampled = self._rsample(c_policy, self._eps_pi)
                a_tanh = torch.tanh(sampled)
                logp = sum_log_prob(squash_correction_log_prob(c_policy, sampled), keepdim=True)
            if self.d_action_sizes:
                c_qs = torch.stack([q(state, a_tanh, obs_list)[1] for q in self.model_q_list])   # [E, B, 1]
            else:   # gradient flows to the action only (the update is restricted to the policy)
                c_qs = self._c_q_values(False, state, a_tanh, obs_list, param_grads=False).unsqueeze(-1)
            sub = self._subsets['pi_c']
            self.noise.subset_(sub, E)
            if self.ensemble_q_sample != E:
                c_qs = c_qs.index_select(0, sub.long())
            loss_c = c_alpha * logp - c_qs.min(dim=0)[0]
            if self.offline_enabled and self.offline_loss:
                loss_c = loss_c + functional.mse_loss(torch.atanh(a_tanh.clamp(-0.999999, 0.999999)), action[..., dsum:],
                                                      reduction='none').sum(-1, keepdim=True)

        loss = torch.mean(loss_c if loss_d is None else (loss_d if loss_c is None else loss_d + loss_c))
        with direct_param_grads(only=pi_inputs):
            loss.backward(inputs=pi_inputs)
        if self._dist is not None:
            self._dist.all_reduce_grads(self._params.grad, *self._params.span('policy'))
        self.optimizer_policy.step()
        with torch.no_grad():
            self._stats['loss_policy'].copy_(loss.detach())
            if self.d_action_sizes:
                self._stats['d_entropy'].copy_(torch.mean(d_policy.entropy().sum(-1) / self.d_action_branch_size))
            if self.c_action_size:
                if plain:   # Normal entropy = 1/2 + 1/2 log(2 pi) + log(scale)
                    self._stats['c_entropy'].copy_(torch.mean((torch.log(scale) + 1.4189385332046727).sum(-1)))
                else:
                    self._stats['c_entropy'].copy_(torch.mean(sum_entropy(c_policy.entropy())))

    def _train_alpha(self, obs_list, state, ls=None, logp=None):
        """`ls`: the (updated) policy's [B, 2A] (loc | scale) output for `state`, if the caller has it;
        `logp`: log pi of an action already drawn from it with `self._eps_alpha`."""
        with torch.no_grad():
            if ls is not None:
                A = self.c_action_size
                d_policy = c_policy = None
                loc, scale, plain = ls[..., :A], ls[..., A:], True
            else:
                d_policy, c_policy, loc, scale, plain = self._policy(state, obs_list)
        loss_d = loss_c = None
        if self.c_action_size and not self.d_action_sizes and plain:
            # continuous-only fast path: dL/dlog_alpha = mean(-logp) - target straight into its gradient slot
            with torch.no_grad():
                if logp is None:
                    self.noise.normal_(self._eps_alpha)
                    scratch = torch.empty(loc.shape, dtype=torch.float32, device=self.device)
                    logp = torch.empty(loc.shape[:-1], dtype=torch.float32, device=self.device)
                    native.squash_sample_fwd(loc, scale, self._eps_alpha, scratch, logp)
                seg0, seg1 = self._params.segments['alpha']
                target = self.target_c_alpha * -float(self.c_action_size)
                opt = self.optimizer_alpha
                if self._dist is None and (opt.start, opt.stop) == (seg0, seg1):
                    g = self._params                       # gradient + Adam in one launch
                    # the last optimizer launch of the step when no further head trains: it also
                    # advances the shared Adam step counter
                    last = self.curiosity is None and not self.use_rnd
                    native.alpha_adam_step(logp, target, 1, g.flat[seg0:seg1], g.grad[seg0:seg1],
                                           opt.exp_avg[seg0:seg1], opt.exp_avg_sq[seg0:seg1], opt.lr,
                                           opt.betas[0], opt.betas[1], opt.eps, opt.steps_done,
                                           advance_counter=last)
                    self._counter_advanced = last
                    return
                native.alpha_grad(logp, target, self._params.grad[seg0 + 1:seg0 + 2])   # [log_d_alpha, log_c_alpha]
            if self._dist is not None:
                self._dist.all_reduce_grads(self._params.grad, *self._params.span('alpha'))
            self.optimizer_alpha.step()
            return
        if self.d_action_sizes and not self.discrete_dqn_like:
            probs = d_policy.probs
            inner = self.log_d_alpha * (-torch.log(probs.clamp(min=1e-8)) - self.target_d_alpha)
            loss_d = torch.sum(probs * inner, dim=1, keepdim=True) / self.d_action_branch_size
        if self.c_action_size:
            self.noise.normal_(self._eps_alpha)
            with torch.no_grad():
                if plain:
                    loc, scale = loc.contiguous(), scale.contiguous()
                    scratch = torch.empty_like(loc)
                    logp = torch.empty(loc.shape[:-1], dtype=torch.float32, device=self.device)
                    native.squash_sample_fwd(loc, scale, self._eps_alpha, scratch, logp)
                    logp = logp.unsqueeze(-1)
                    valid = float(self.c_action_size)
                else:
                    sampled = self._eps_alpha * c_policy.scale + c_policy.loc
                    if hasattr(c_policy, 'padding_mask'):
                        sampled[..., c_policy.padding_mask] = 0.
                    lp = squash_correction_log_prob(c_policy, sampled)
                    valid = torch.sum(lp != torch.inf, dim=-1, keepdim=True)
                    logp = sum_log_prob(lp, keepdim=True)
            loss_c = self.log_c_alpha * (-logp - self.target_c_alpha * -valid)
        loss = torch.mean(loss_c if loss_d is None else (loss_d if loss_c is None else loss_d + loss_c))
        loss.backward(inputs=[self.log_d_alpha, self.log_c_alpha])
        if self._dist is not None:
            self._dist.all_reduce_grads(self._params.grad, *self._params.span('alpha'))
        self.optimizer_alpha.step()

    def _alpha_sidecar(self, logp):
        """the continuous-only temperature step (`_train_alpha`'s one-launch form) as a sidecar job, or None when
        that form does not apply (an optimizer over more than the two temperatures).  Data parallel: the caller
        averages `logp` over the ranks between the launch that writes it and the launch that hosts the job
        (`_alpha_logp_over_ranks`): the gradient mean_b(-logp_b) - target is linear in it, so the job then applies the
        rank-averaged gradient — one collective on B floats instead of gradient launch + collective + Adam launch"""
        seg0, seg1 = self._params.segments['alpha']
        opt = self.optimizer_alpha
        if (opt.start, opt.stop) != (seg0, seg1) or self.d_action_sizes:
            return None
        g = self._params
        last = self.curiosity is None and not self.use_rnd      # the step's last optimizer launch advances the counter
        self._counter_advanced = last
        return native.sidecar_alpha_adam(logp, self.target_c_alpha * -float(self.c_action_size), 1, g.flat[seg0:seg1],
                                         g.grad[seg0:seg1], opt.exp_avg[seg0:seg1], opt.exp_avg_sq[seg0:seg1], opt.lr,
                                         opt.betas[0], opt.betas[1], opt.eps, opt.steps_done, advance_counter=last)

    def _alpha_logp_over_ranks(self, logp) -> None:
        if self._dist is not None:
            self._dist.all_reduce_grads(logp, 0, logp.numel())

    def _train_curiosity(self, n_padding_masks, nx_states, n_actions):
        # the reference differentiates w.r.t. the model's parameters only (`backward(inputs=parameters)`,
        # sac_base.py:1951-1976): same gradients from detached inputs, which lets the fused stack add its
        # parameter gradients where they live
        nx_states, n_actions = nx_states.detach(), n_actions.detach()
        n_states, next_n_states = nx_states[:, :-1], nx_states[:, 1:]
        if self.curiosity == CURIOSITY.FORWARD:
            pred, target = self.model_forward_dynamic(n_states, n_actions), next_n_states
        else:
            pred, target = self.model_inverse_dynamic(n_states, next_n_states), n_actions
        _masked_mse_backward(pred, target, n_padding_masks, self._stats['loss_curiosity'], fused=self._fused_curiosity)
        if self._dist is not None:
            self._dist.all_reduce_grads(self._params.grad, *self._params.span('curiosity'))
        self.optimizer_curiosity.step()

    @torch.no_grad()
    def _get_td_error(self, n_last_masks, n_padding_masks, nx_obses_list, state, nx_target_states, nx_actions,
                      n_rewards, n_dones, n_mu_probs, ls=None, sample=None, stored_pi=None, c_q=None, q_table=None):
        """mean_e |Q_e(s0, a0) - y(target states)| -> self._td_error [B] (reference 2182-2245).
        `ls`: see `_get_y`."""
        dsum = self.d_action_summed_size
        obs_list = [o[:, 0] for o in nx_obses_list]
        action = nx_actions[:, 0]
        d_action, c_action = action[..., :dsum], action[..., dsum:]
        q_list = None
        if c_q is not None:
            pass                  # [E, B] online Q of (s0, a0), already computed by the caller
        elif self.d_action_sizes:
            q_list = [q(state, c_action, obs_list) for q in self.model_q_list]
            c_q = torch.stack([q[1] for q in q_list]).squeeze(-1).contiguous() if self.c_action_size else None
        else:
            c_q = self._c_q_values(False, state, c_action, obs_list).contiguous()
        fused_td = bool(self.c_action_size) and not self.d_action_sizes
        d_y, c_y = self._get_y(n_last_masks, n_padding_masks, nx_obses_list, nx_target_states, nx_actions,
                               n_rewards, n_dones, n_mu_probs, eps_buf=self._eps_td, subset_prefix='td',
                               y_out=self._y_td_buf, q_online=c_q if fused_td else None, td_out=self._td_error,
                               ls=ls, sample=sample, stored_pi=stored_pi, q_table=q_table)
        if fused_td:
            return self._td_error
        err = torch.zeros((self.ensemble_q_num, state.shape[0], 1), device=self.device)
        if self.d_action_sizes:
            d_q = torch.stack([torch.sum(d_action * q[0], dim=-1, keepdim=True) / self.d_action_branch_size
                               for q in q_list])
            err = err + torch.abs(d_q - d_y)
        if self.c_action_size:
            err = err + torch.abs(c_q.unsqueeze(-1) - c_y)
        self._td_error.copy_(err.mean(dim=0).reshape(-1))
        return self._td_error

    # ==========================================================================================
    # episode ingress (reference sac_base.py:2303-2396)
    # ==========================================================================================
    def put_episode(self, ep_indexes, ep_obses_list, ep_actions, ep_rewards, ep_dones, ep_probs,
                    ep_pre_seq_hidden_states) -> None:
        """Reference sac_base.py:2303-2396.  NumPy arrays (the reference's callers) or tensors already in HBM
        (the device-resident `AgentManager`): the latter go slab -> ring without touching the host."""
        if ep_indexes.shape[1] < self.n_step:
            return
        if isinstance(ep_indexes, torch.Tensor):
            assert ep_indexes.dtype == torch.int32
            last = ep_indexes == -1
            last[:, -1] = True
            obs_dev = [o[0] for o in ep_obses_list]
        else:
            assert ep_indexes.dtype == np.int32
            last = np.zeros_like(ep_indexes, dtype=bool)
            last[:, -1] = True
            last[ep_indexes == -1] = True
            obs_dev = None
        rows = {'index': ep_indexes[0], 'last_mask': last[0],
                **{f'obs_{name}': o[0] for name, o in zip(self.obs_names, ep_obses_list)},
                'action': ep_actions[0], 'reward': ep_rewards[0], 'done': ep_dones[0],
                'mu_prob': ep_probs[0], 'pre_seq_hidden_state': ep_pre_seq_hidden_states[0]}
        if self.use_normalization:
            self._update_normalizer(obs_dev if obs_dev is not None
                                    else [torch.from_numpy(o[0]).to(self.device) for o in ep_obses_list])
        self.replay_buffer.add(rows, ignore_size=1)

    # ==========================================================================================
    # the step
    # ==========================================================================================
    def _device_step(self) -> None:
        # the representation passes of a step see the same window buffers: attention blocks build their index /
        # padding / attention masks once per step (nn_models.layers.seq_layers.step_mask_cache)
        with step_mask_cache() if self.seq_encoder == SEQ_ENCODER.ATTN else contextlib.nullcontext():
            self._device_step_body()

    def _device_step_body(self) -> None:
        """Everything one `train()` does on the device, without a single host synchronisation
        (reference `_sample_from_replay_buffer` 2398-2494, `_train` 2027-2126, write-backs 2558-2605).
        Reads / writes only static buffers, so it can be captured and replayed as a hipGraph.  Phases:
          _step_sample            prologue (Polyak, draws) + PER sample + window gather -> views of the static batch
          _step_rep_and_q         representation passes (online, target), Q step, states under the updated representation
          _step_policy            policy step
          _step_after_policy_*    stock networks: the UPDATED policy over the window -> new behaviour probabilities, the
                                  TD target's sample, the temperature step's sample (one launch, or the chain it replaces)
          _step_temperature_and_aux  temperature step (unless a sidecar took it), curiosity / RND
          _step_write_backs       TD error -> priorities, mu-probability and hidden-state rows"""
        self._counter_advanced = False
        w = self._step_sample()
        self._step_rep_and_q(w)
        self._step_policy(w)
        post = _AfterPolicy()
        self._vtrace_sidecars = self._pending_alpha = None
        self._join_lookahead()       # (everything below may write priorities / replay rows)
        if w.stock and self.use_n_step_is:
            with torch.no_grad():
                if not self._step_after_policy_one_launch(w, post):
                    self._step_after_policy_chain(w, post)
        self._step_temperature_and_aux(w, post)
        self._step_write_backs(w, post)
        if not self._counter_advanced:
            self._opt_steps.add_(1)

    def _join_lookahead(self) -> None:
        if self._la_gather is not None:      # no launch hosted the next batch's gather: on its own, before any write-back
            self._la_gather = None
            self.replay_buffer.gather_next_now()

    def _take_la_gather(self):
        """-> the pending gather of the NEXT batch as a sidecar list for a launch that reads nothing of it, or None"""
        sc, self._la_gather = self._la_gather, None
        return None if sc is None else [sc]

    def _step_sample(self):
        """-> the step's window views (`_Window`): [B, L] tensors are `bnx_*`, their first L - 1 rows `bn_*`"""
        rb, b = self.replay_buffer, self.burn_in_step
        # Polyak of every step rides in the step's first launch, together with every uniform / Gaussian draw and
        # ensemble subset of the step (recorded test noise: a plain Polyak launch, draws injected by the test)
        polyak = None
        if self.update_target_per_step == 1 and self._polyak_len > 0:
            polyak = (self._target_params.flat[:self._polyak_len], self._params.flat[:self._polyak_len], self.tau)
        zero = None if self._grads_overwrite else self._params.grad
        if self._lookahead:
            # the batch this step trains on was drawn during the previous step (`train` swapped the sets); the NEXT one
            # is drawn now, from the tree and the rows as the previous step left them, before this step's first write to
            # the replay
            sampled = False
            if self._use_sidecars:
                # the prologue launch hosts the NEXT batch's tree walk (its sampler workgroup), as in the plain schedule
                rb.swap_sets()
                try:
                    sampled = self.noise.begin_step_with_sample(self._opt_steps, rb, self._eps_all, self._subsets_all,
                                                                self.ensemble_q_num, polyak=polyak, zero=zero)
                finally:
                    rb.swap_sets()
            if not sampled:
                self.noise.begin_step(self._opt_steps, rb.next_uniforms() if rb.uniform_source is self.noise else None,
                                      self._eps_all, self._subsets_all, self.ensemble_q_num, polyak=polyak, zero=zero)
            if sampled and self._la_gather_sidecar:
                # ... and its window gather rides as extra workgroups of the step's first policy / critic launch
                # (`_take_la_gather`; `_join_lookahead` runs it on its own if no launch took it)
                self._la_gather = rb.next_gather_sidecar()
            else:
                rb.sample_next_into_static(sampled=sampled)      # same launches, in line
        else:
            # ... and the sampler of the batch it draws: one launch for K1 + K2 + K5
            sampled = self._use_sidecars and self.noise.begin_step_with_sample(
                self._opt_steps, rb, self._eps_all, self._subsets_all, self.ensemble_q_num, polyak=polyak, zero=zero,
                defer_weights=self._defer_is_weights)
            if not sampled:
                self.noise.begin_step(self._opt_steps, rb._u if rb.uniform_source is self.noise else None, self._eps_all,
                                      self._subsets_all, self.ensemble_q_num, polyak=polyak, zero=zero)
            rb.sample_into_static(sampled=sampled)
        batch = rb._batch
        w = _Window()
        w.ids = rb._ids
        w.priority_is = rb._w.unsqueeze(-1) if self.use_priority else None
        w.bnx_obses_list = [batch[f'obs_{name}'] for name in self.obs_names]
        w.bnx_actions, w.bnx_pad = batch['action'], batch['padding_mask']
        w.bn_indexes, w.bn_last, w.bn_pad = batch['index'][:, :-1], batch['last_mask'][:, :-1], w.bnx_pad[:, :-1]
        w.bn_actions, w.bn_rewards, w.bn_dones = w.bnx_actions[:, :-1], batch['reward'][:, :-1], batch['done'][:, :-1]
        w.bn_mu_probs = batch['mu_prob'][:, :-1]
        w.bnx_hidden = batch['pre_seq_hidden_state']
        w.nx_obs = [o[:, b:] for o in w.bnx_obses_list]
        w.obs_b = [o[:, b] for o in w.bnx_obses_list]
        w.stock = self._stock_c_only()
        w.rep_trainable = self.optimizer_rep is not None
        self.noise.prefill(self._eps_all)          # (torch fallback: one launch for all Gaussian draws)
        if type(self.model_rep) is ModelSimpleRep:
            # the stock concatenation rep ignores index / mask / previous actions: do not build them
            w.rep_in = (None, None, w.bnx_obses_list, None, w.bnx_hidden)
        else:
            if rb.derived is not None and rb.sharded is None:
                # (the gather delivered them as derived keys of its launch: no `asac_window_aux`)
                d = rb.derived
                bnx_indexes, bnx_padding_masks, bnx_pre_actions = d['index_x'], d['padding_mask_x'], d['pre_action']
            else:
                bnx_indexes, bnx_padding_masks, bnx_pre_actions = self.get_bnx_data(
                    w.bn_indexes, w.bn_pad, w.bn_actions, pre_action_out=rb.joint_pre_action)
            w.rep_in = (bnx_indexes, bnx_padding_masks, w.bnx_obses_list, bnx_pre_actions, w.bnx_hidden)
        return w

    def _step_rep_and_q(self, w) -> None:
        """reference `_train` 2066-2103: online and target representation over the window, `_train_rep_q`, the states
        again under the updated representation -> w.bnx_states, w.bnx_target_states, w.next_hidden"""
        b = self.burn_in_step
        cat_mode = self._cat_mode if self._cat_mode is not None else contextlib.nullcontext
        with_aux = self.siamese is not None or self.use_prediction
        # A representation without a sequence encoder maps every step on its own (it is handed single steps when
        # acting), and the Q loss reads the state of ONE window position — everything else of the window only feeds
        # detached targets.  The differentiable pass then covers that position's rows alone (B of B L frames: the
        # backward of a convolution stack shrinks L-fold) beside a no-grad pass over the window.
        # (where the pass in front of the update already covers only a few positions — `from_b` below, n + 1 <= 8 — the
        # extra forward's small-launch floors cost more than the backward over those few positions saves: cfg4, n + 1 = 4,
        # 2 755 with it against 2 787 steps/s without)
        one_position = (self._rep_grad_one_position and self.seq_encoder is None and w.rep_trainable and not with_aux
                        and type(self.model_rep) is not ModelSimpleRep and w.bnx_actions.shape[1] > 1
                        and not (self._rep_from_burn_in and b > 0 and self.n_step + 1 <= 8))
        # ... and the burn-in positions of the window feed nothing before the update (a sequence encoder would carry them
        # forward; here states[:, b:] is all `_train_rep_q`, the return and the TD error read): the two passes in front of
        # the update run on the positions from b on — (n + 1) / L of the frames, read in place (`asac_conv2_forward_windows`)
        from_b = (self._rep_from_burn_in and self.seq_encoder is None and b > 0 and w.rep_trainable
                  and type(self.model_rep) is not ModelSimpleRep)     # (a fixed representation's one pass serves the whole step)
        rep_in, pb = w.rep_in, b
        if from_b:
            tail = lambda x: None if x is None else x[:, b:]  # noqa: E731
            idx, pad, obs, pre, hidden = w.rep_in
            rep_in, pb = (tail(idx), tail(pad), [o[:, b:] for o in obs], tail(pre), tail(hidden)), 0
        with (self._rep_twin if self._rep_twin else contextlib.nullcontext()), cat_mode():
            with torch.no_grad() if one_position else contextlib.nullcontext():
                bnx_states, next_hidden = self.get_l_states(*rep_in, is_target=False)
            with torch.no_grad():
                w.bnx_target_states, _ = self.get_l_states(*rep_in, is_target=True)
        w.nx_target_states = w.bnx_target_states[:, pb:]
        state_base = (bnx_states, pb)
        if one_position:
            at = lambda x: None if x is None else x[:, b:b + 1]  # noqa: E731
            idx, pad, obs, pre, hidden = w.rep_in
            with cat_mode():
                state_b, _ = self.get_l_states(at(idx), at(pad), [o[:, b:b + 1] for o in obs], at(pre), at(hidden),
                                               is_target=False)
            state_base = (state_b, 0)
        aux = None
        if with_aux:
            aux = dict(n_indexes=w.bn_indexes[:, b:],
                       n_pre_actions=w.bn_actions[:, b - 1:-1] if b > 0 else w.bn_actions[:, 0:0],
                       n_pre_seq_hidden_states=w.bnx_hidden[:, b:-1], nx_target_states=w.nx_target_states)
        self._train_rep_q(w.bn_last[:, b:], w.bn_pad[:, b:], w.nx_obs, time_slice(bnx_states, pb), w.bnx_actions[:, b:],
                          w.bn_rewards[:, b:], w.bn_dones[:, b:], w.bn_mu_probs[:, b:], w.priority_is, aux,
                          policy_sample=w.stock and not w.rep_trainable,
                          state_base=state_base)
        if w.rep_trainable:   # states under the updated representation (reference 2097-2103)
            with torch.no_grad(), cat_mode():
                w.bnx_states, w.next_hidden = self.get_l_states(*w.rep_in, is_target=False)
        else:
            w.bnx_states, w.next_hidden = bnx_states.detach(), next_hidden.detach()

    def _step_policy(self, w) -> None:
        b = self.burn_in_step
        w.state_b = w.bnx_states[:, b]
        # the target computation already ran the (still unchanged) policy on this state: reuse its output
        ls_b = self._ls_y[:, 0] if (w.stock and not w.rep_trainable and self._ls_y is not None) else None
        self._train_policy(w.obs_b, w.state_b, w.bn_actions[:, b], w.bn_mu_probs[:, b, :self.d_action_summed_size], ls=ls_b)

    def _auto_alpha(self) -> bool:
        return bool(self.use_auto_alpha and ((self.d_action_sizes and not self.discrete_dqn_like) or self.c_action_size))

    def _same_states(self, w) -> bool:
        """the target representation's states ARE the online ones (parameter-free representation, no burn-in)"""
        return (self.burn_in_step == 0 and w.bnx_target_states.data_ptr() == w.bnx_states.data_ptr()
                and w.bnx_target_states.stride() == w.bnx_states.stride())

    def _step_after_policy_one_launch(self, w, post) -> bool:
        """One forward of the UPDATED stock policy over the whole window serves the new mu-probabilities (rows < L - 1),
        the temperature step (row b) and, where the target representation is the online one, the TD-error target
        (rows >= b): policy -> [pi(stored actions) = the new mu, the TD target's sample, the temperature step's sample] ->
        target critics on the TD sample, with the TD error's online Q of (s_b, a_b) riding along and the mu-probability
        write-back electing beside it — ONE launch (bit-identical to the chain `_step_after_policy_chain` issues).  The
        write-back's second pass rides in the TD error's return launch; the temperature step (it needs every tile's
        sample) rides in the priority update, the step's last launch, and the TD error's return, which must already see
        the new temperature, evaluates the value that step will write (`pending_alpha`).  -> False: not applicable."""
        rb, b, n = self.replay_buffer, self.burn_in_step, self.n_step
        if not (self._fused_td_chain and self.use_priority and self._same_states(w) and self._use_sidecars
                and not self._wide_critics and rb.sharded is None and self.curiosity is None and not self.use_rnd):
            return False
        B_, L_, A = *w.bnx_states.shape[:2], self.c_action_size
        f32 = dict(dtype=torch.float32, device=self.device)
        auto_alpha = self._auto_alpha()
        rows_win = StockMLP._rows(w.bnx_states, self.state_size)
        job_pi, ls_out = self._fpi.job(rows_win, None)
        probs_win = torch.empty((B_, L_, A), **f32)
        td_sample = (torch.empty((B_, L_, A), **f32), torch.empty((B_, L_), **f32))
        job_tq, td_q_table = self._ftq.job(rows_win, td_sample[0].view(-1, A))
        alpha_logp = scratch = None
        if auto_alpha:
            alpha_logp, scratch = torch.empty(B_, **f32), torch.empty((B_, A), **f32)
        job = native.pi_q_job(job_pi, job_tq, self._eps_td, td_sample[0], td_sample[1], L_,
                              action=w.bnx_actions, prob_out=probs_win,
                              eps2=self._eps_alpha if auto_alpha else None, t2=b,
                              a2_out=scratch, logp2_out=alpha_logp)
        if not native.policy_sample_q_forward_ok(job):
            return False
        if auto_alpha:
            self.noise.normal_(self._eps_alpha)
        self.noise.normal_(self._eps_td)
        xb = StockMLP._rows(w.bnx_states[:, b], self.state_size)
        ab = StockMLP._rows(w.bnx_actions[:, b], self.c_action_size)
        job_q, _ = self._fq.job(xb, ab, out=self._cq_td_buf)
        # launches off the step's critical path ride as sidecar workgroups of launches that are on it (csrc/asac_sidecar.h)
        sc_elect, post.sc_write = rb.window_scatter_sidecars(w.ids, -b, b + n, w.bnx_pad, 'mu_prob', probs_win[:, :-1])
        if auto_alpha:
            post.sc_alpha = self._alpha_sidecar(alpha_logp)
        native.policy_sample_q_forward(job, [job_q], sidecars=[sc_elect])
        if post.sc_alpha is not None:
            self._alpha_logp_over_ranks(alpha_logp)
        self._pending_alpha = post.sc_alpha      # (None: a wider optimizer -> `_train_alpha`)
        post.ls_win = ls_out[0].view(B_, L_, 2 * A)
        post.probs_win, post.td_sample, post.alpha_logp = probs_win, td_sample, alpha_logp
        post.td_q_table = td_q_table.view(self.ensemble_q_num, B_, L_)
        post.side_cq = self._cq_td_buf.view(self.ensemble_q_num, -1)
        self._vtrace_sidecars = [post.sc_write]
        post.mu_written = True
        return True

    def _step_after_policy_chain(self, w, post) -> None:
        """The chain form: policy forward over the window (the TD target's policy forward over the TARGET states, where
        they are not the online ones, beside it) -> ONE elementwise launch [temperature sample, pi(stored actions), TD
        target's sample] -> the TD error's online Q of (s_b, a_b) with its target ensemble on the sampled window actions
        (two networks, one launch; hosts the write-back's second pass and the temperature step as sidecars)."""
        rb, b, n = self.replay_buffer, self.burn_in_step, self.n_step
        B_, L_, A = *w.bnx_states.shape[:2], self.c_action_size
        f32 = dict(dtype=torch.float32, device=self.device)
        auto_alpha, same_states = self._auto_alpha(), self._same_states(w)
        rows_win = StockMLP._rows(w.bnx_states, self.state_size)
        td_own = self.use_priority and not same_states
        td_rows = sc_elect = None
        probs_win = torch.empty((B_, L_, A), **f32)
        # (with priorities the TD error's online-Q launch follows and hosts the second pass + the temperature step)
        if self.use_priority and self._use_sidecars and rb.sharded is None:
            sc_elect, post.sc_write = rb.window_scatter_sidecars(w.ids, -b, b + n, w.bnx_pad, 'mu_prob', probs_win[:, :-1])
        # the policy forward(s) WITH the elementwise work on their outputs as the forward launch's epilogue
        # (`native.mlp_forward_multi_sampled`: temperature sample at position b, pi(stored actions), the TD target's sample);
        # the chain form below is what it replaces, bit for bit
        if auto_alpha:
            post.alpha_logp, scratch = torch.empty(B_, **f32), torch.empty((B_, A), **f32)
        if self.use_priority and same_states:
            post.td_sample = (torch.empty((B_, L_, A), **f32), torch.empty((B_, L_), **f32))
        elif td_own:
            post.td_sample = (torch.empty((B_, n + 1, A), **f32), torch.empty((B_, n + 1), **f32))
            post.td_pi = torch.empty((B_, n + 1, A), **f32)
        job_win, ls_out = self._fpi.job(rows_win, None)
        jobs = [job_win]
        epis = None
        if native.SAMPLE_EPILOGUE and 2 * A <= 16 and w.bnx_actions.stride(-1) == 1:
            epis = [native.sample_epilogue(job_win, *((self._eps_td, *post.td_sample) if self.use_priority and same_states
                                                      else (None, None, None)), L_,
                                           action=w.bnx_actions, prob_out=probs_win,
                                           eps2=self._eps_alpha if auto_alpha else None, t2=b,
                                           a2_out=scratch if auto_alpha else None,
                                           logp2_out=post.alpha_logp if auto_alpha else None)]
        if td_own:
            td_rows = StockMLP._rows_in_place(w.nx_target_states, self.state_size)
            job_pi_td, ls_td_out = self._fpi.job(td_rows, None)
            jobs.append(job_pi_td)
            post.ls_td = ls_td_out[0].view(B_, n + 1, 2 * A)
            if epis is not None:
                epis.append(native.sample_epilogue(job_pi_td, self._eps_td, *post.td_sample, n + 1,
                                                   action=w.bnx_actions[:, b:], prob_out=post.td_pi))
        ls_win = ls_out[0].view(B_, L_, 2 * A)
        if epis is not None and not native.mlp_forward_multi_sampled_ok(jobs, epis):
            epis = None
        if auto_alpha:
            self.noise.normal_(self._eps_alpha)
        if self.use_priority:
            self.noise.normal_(self._eps_td)
        if auto_alpha and self.use_priority and self._use_sidecars and rb.sharded is None:
            post.sc_alpha = self._alpha_sidecar(post.alpha_logp)
        if epis is not None:
            native.mlp_forward_multi_sampled(jobs, epis, sidecars=[sc_elect] if sc_elect is not None else None)
        else:
            native.mlp_forward_multi(jobs)
            sq = [native.squash_job(ls_win[..., :A], ls_win[..., A:], action=w.bnx_actions, prob_out=probs_win)]
            if auto_alpha:
                sq.append(native.squash_job(ls_win[:, b, :A], ls_win[:, b, A:], self._eps_alpha, scratch, post.alpha_logp))
            if self.use_priority and same_states:
                sq.append(native.squash_job(ls_win[..., :A], ls_win[..., A:], self._eps_td, *post.td_sample))
            elif td_own:
                sq.append(native.squash_job(post.ls_td[..., :A], post.ls_td[..., A:], self._eps_td, *post.td_sample,
                                            action=w.bnx_actions[:, b:], prob_out=post.td_pi))
            native.squash_multi(sq, sidecars=[sc_elect] if sc_elect is not None else None)
        if post.sc_alpha is not None:
            self._alpha_logp_over_ranks(post.alpha_logp)
        post.ls_win, post.probs_win = ls_win, probs_win
        if self.use_priority:
            xb = StockMLP._rows(w.bnx_states[:, b], self.state_size)
            ab = StockMLP._rows(w.bnx_actions[:, b], self.c_action_size)
            riders = [sc for sc in (post.sc_write, post.sc_alpha) if sc is not None]
            if riders and self._wide_critics:
                # no critic launch to ride in: the write-back's second pass goes with the TD error's return, the temperature
                # step with the priority update (the return evaluates the value that step will write)
                self._vtrace_sidecars = [post.sc_write] if post.sc_write is not None else None
                self._pending_alpha = post.sc_alpha
                riders = []
            if post.td_sample is not None:
                job_q, _ = self._fq.job(xb, ab, out=self._cq_td_buf)
                job_tq, td_q_table = self._ftq.job(
                    td_rows if td_rows is not None else StockMLP._rows(w.bnx_target_states, self.state_size),
                    StockMLP._rows(post.td_sample[0], self.c_action_size))
                native.mlp_forward_multi([job_q, job_tq], sidecars=riders or None)
                post.td_q_table = td_q_table.view(self.ensemble_q_num, *post.td_sample[1].shape)
            elif riders:
                job_q, _ = self._fq.job(xb, ab, out=self._cq_td_buf)
                native.mlp_forward_multi([job_q], sidecars=riders)
            else:
                self._fq._launch_forward(xb, ab, out=self._cq_td_buf)
            post.side_cq = self._cq_td_buf.view(self.ensemble_q_num, -1)
        if post.sc_write is None:
            rb.update_window_transitions(w.ids, -b, b + n, w.bnx_pad, 'mu_prob', probs_win[:, :-1])
        post.mu_written = True

    def _step_temperature_and_aux(self, w, post) -> None:
        b = self.burn_in_step
        if self._auto_alpha() and post.sc_alpha is None:      # (a sidecar: done by the launch that carries it)
            self._train_alpha(w.obs_b, w.state_b, ls=None if post.ls_win is None else post.ls_win[:, b], logp=post.alpha_logp)
        if self.curiosity is not None:
            self._train_curiosity(w.bn_pad[:, b:], w.bnx_states[:, b:], w.bn_actions[:, b:])
        if self.use_rnd:
            self._train_rnd(w.bn_pad[:, b:], w.bnx_states[:, b:-1], w.bn_actions[:, b:])

    def _step_write_backs(self, w, post) -> None:
        """reference 2558-2605: TD error -> priorities; new behaviour probabilities and hidden states -> the ring"""
        rb, b, n, ids = self.replay_buffer, self.burn_in_step, self.n_step, w.ids
        bn_states = w.bnx_states[:, :-1]
        pi_probs = None
        if self.use_n_step_is:
            if post.probs_win is not None:
                pi_probs = post.probs_win[:, :-1]          # the last row's probability is not stored (1159-1189)
            else:
                pi_probs = self.get_l_probs([o[:, :-1] for o in w.bnx_obses_list], bn_states, w.bn_actions)
        # the hidden-state write-back rides along too: its election beside the TD error's return, its write pass beside the
        # priority update (the step's last two launches; its own election scratch: the mu-probability write pass may
        # share the return launch)
        hidden_write = None
        if (self.seq_hidden_state_shape[-1] != 0 and self.use_priority and self._use_sidecars
                and rb.sharded is None and bool(self.c_action_size) and not self.d_action_sizes
                and len(self._vtrace_sidecars or ()) < native.MAX_SIDECARS):
            hidden_rows = w.next_hidden.detach().contiguous()
            h_elect, hidden_write = rb.window_scatter_sidecars(ids, 1 - b, b + n, w.bnx_pad, 'pre_seq_hidden_state',
                                                               hidden_rows, side=True)
            self._vtrace_sidecars = list(self._vtrace_sidecars or ()) + [h_elect]
        if self.use_priority:
            # no write pass waiting for the update's launch: the TD error's return and the priority update are one launch
            # (one workgroup forms every return: it pays while each of its threads has at most one step of one window —
            # measured: B 256 n 4 +2.5 %, B 512 n 3 -0.8 %, B 1024 n 3 -0.3 %)
            merged = (self._fused_td_update and hidden_write is None and self._use_sidecars
                      and bool(self.c_action_size) and not self.d_action_sizes and ids.numel() * n <= 1024
                      and rb.td_update_ok(ids, n))
            self._td_update_with = (rb, ids) if merged else None
            own_td_policy = post.td_pi is not None     # the TD target's policy ran over the target states
            td = self._get_td_error(w.bn_last[:, b:], w.bn_pad[:, b:], w.nx_obs, bn_states[:, b],
                                    w.nx_target_states, w.bnx_actions[:, b:], w.bn_rewards[:, b:],
                                    w.bn_dones[:, b:], pi_probs[:, b:] if self.use_n_step_is else None,
                                    ls=None if post.td_sample is None else (post.ls_td if own_td_policy else post.ls_win),
                                    sample=post.td_sample,
                                    stored_pi=None if post.td_sample is None else (post.td_pi if own_td_policy else post.probs_win),
                                    c_q=post.side_cq,
                                    q_table=post.td_q_table)
            assert not self._vtrace_sidecars, 'the TD error\'s return launch did not take its sidecars'
            if merged and self._td_update_with is None:
                assert self._pending_alpha is None      # the return's launch ran the temperature step and the update
            else:
                self._td_update_with = None
                rb.update(ids, td, sidecars=[sc for sc in (self._pending_alpha, hidden_write) if sc is not None] or None)
                self._pending_alpha = None
        if self.seq_hidden_state_shape[-1] != 0 and hidden_write is None:
            rb.update_window_transitions(ids, 1 - b, b + n, w.bnx_pad, 'pre_seq_hidden_state',
                                         w.next_hidden.detach().contiguous())
        if self.use_n_step_is and not post.mu_written:
            rb.update_window_transitions(ids, -b, b + n, w.bnx_pad, 'mu_prob', pi_probs)

    def _finish_graph(self, graph) -> None:
        """A captured (kept, not yet instantiated) graph -> executable: its memset nodes become kernel nodes first — on this
        ROCm a captured hipMemsetAsync takes effect on the first launch only, and ATen's split reductions (every
        nn.Linear's bias gradient) zero their semaphores with one (csrc/graph_fix.hip)."""
        replaced, kept = native.graph_replace_memset_nodes(int(graph.raw_cuda_graph()))
        if kept:
            # a pitched (2-D) memset has the same replay fault the pass exists to repair and is not rewritten: a step that
            # captured one must not be replayed (no kernel of the library or of ATen's step issues one today)
            raise RuntimeError(f'{kept} 2-D memset node(s) in the captured step: not replayable on this ROCm')
        graph.instantiate()
        if replaced or kept:
            self._logger.info(f'captured graph: {replaced} memset node(s) replaced by fill kernels' +
                              (f', {kept} 2-D memset node(s) left as captured' if kept else ''))
        self._graph_memsets = (replaced, kept)

    @staticmethod
    def _graph_api_ok() -> bool:
        """`CUDAGraph(keep_graph=True)` / `raw_cuda_graph()` / `instantiate()`: the memset-node repair needs the graph before
        instantiation.  A torch build without them cannot replay a captured step correctly here."""
        import inspect
        try:
            return ('keep_graph' in inspect.signature(torch.cuda.CUDAGraph.__new__).parameters or
                    'keep_graph' in (torch.cuda.CUDAGraph.__new__.__doc__ or '') or
                    hasattr(torch.cuda.CUDAGraph, 'raw_cuda_graph')) and hasattr(torch.cuda.CUDAGraph, 'instantiate')
        except (TypeError, ValueError):
            return hasattr(torch.cuda.CUDAGraph, 'raw_cuda_graph') and hasattr(torch.cuda.CUDAGraph, 'instantiate')

    def _try_capture(self) -> None:
        """Warm up on a side stream, then capture `_device_step` into one hipGraph."""
        if not self._graph_api_ok():
            self._graph_failed = True
            self._logger.warning('this torch build has no CUDAGraph(keep_graph=True) / raw_cuda_graph() / instantiate(): captured '
                                 'memset nodes could not be repaired (they take effect on the first replay only on this '
                                 'ROCm), so the train step is NOT captured and runs eagerly — several times slower')
            return
        try:
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream())
            graph = torch.cuda.CUDAGraph(keep_graph=True)
            # thread_local: the RCCL watchdog thread may query events while this thread captures
            with torch.cuda.graph(graph, stream=side, capture_error_mode='thread_local'):
                self._device_step()
            self._finish_graph(graph)
            self._graph = graph
            self._graph_exec, self._graph_exec_checked = None, False
            # (logp, scale) of the policy step live in THIS graph's private pool: whichever graph ran last is the one
            # whose tensors `_refresh_policy_stats` must read (a k-step run has its own, `train_steps`)
            self._graph_stats_src = self._pi_stats_src
            self._logger.info('train step captured into a hipGraph')
        except Exception as e:   # user models with host syncs etc.: stay eager, loudly
            self._graph_failed = True
            self._graph = None
            torch.cuda.synchronize()
            self._logger.warning(f'hipGraph capture of the train step failed, staying eager: {e!r}')

    def _replay_graph(self) -> None:
        """torch's `CUDAGraph.replay()` re-seeds its Philox generator before every launch (two fill
        kernels).  The first replay goes through torch and watches the generator offset: if the
        captured step consumed no torch random numbers (all draws come from `asac_noise_fill`), later
        steps launch the instantiated graph directly."""
        self._pi_stats_src = self._graph_stats_src
        if self._graph_exec is not None:
            native.graph_launch(self._graph_exec)
            return
        if self._graph_exec_checked or not self._direct_graph_launch:
            self._graph.replay()
            return
        gen = torch.cuda.default_generators[self.device.index or 0]
        before = gen.get_offset()
        self._graph.replay()
        self._graph_exec_checked = True
        if gen.get_offset() == before and hasattr(self._graph, 'raw_cuda_graph_exec'):
            try:
                self._graph_exec = int(self._graph.raw_cuda_graph_exec())
                self._logger.info('captured step draws no torch random numbers: launching the graph directly')
            except Exception as e:   # older torch: keep torch's replay
                self._logger.warning(f'raw graph handle unavailable, using CUDAGraph.replay(): {e!r}')

    def _ready_to_train(self) -> bool:
        """Reference `train` 2503-2506: no step until the buffer holds more than a batch.  A data-parallel step holds
        collectives, so the ranks decide TOGETHER (one small all-reduce per call until it turns true, none afterwards):
        throughput mode needs every rank's shard above its own batch; parity mode needs the UNION above the global batch
        (and every rank to know the transition layout: it allocates its batch from it) — a rank with a short shard
        takes part from the first step."""
        rb = self.replay_buffer
        if self._dist is None:
            return rb.is_lg_batch_size
        if self._dist_ready:
            return True
        if rb.sharded is not None:
            all_have_keys, union = self._dist.all_ready(rb._columns is not None and bool(rb._columns), rb.size, self.device)
            self._dist_ready = all_have_keys and union > rb.sharded.B
        else:
            self._dist_ready, _ = self._dist.all_ready(rb.is_lg_batch_size, rb.size, self.device)
        return self._dist_ready

    def _optimizer_hp(self) -> tuple:
        """(lr, betas, eps) of every optimizer of the learner: host floats that a captured step holds as kernel arguments"""
        opts = self._all_optimizers
        if opts is None:      # (collected once: `train()` is the host's hot loop)
            opts = self._all_optimizers = [o for o in vars(self).values() if isinstance(o, FlatAdam)] + \
                [o for o in self.optimizer_q_list if o is not None]
        return tuple((o.lr, o.betas, o.eps) for o in opts)

    def _drop_graphs_if_hp_changed(self) -> None:
        """a learning rate (betas, eps) changed after capture — a schedule, a user edit — would otherwise go unnoticed by
        every replay: the graphs are dropped and the step is captured again with the new values"""
        hp = self._optimizer_hp()
        if hp != self._graph_hp:
            if self._graph_hp is not None and (self._graph is not None or self._la_graphs or self._graph_runs):
                self._logger.info('optimizer hyper-parameters changed: the captured step is dropped and captured again')
                self._graph, self._graph_exec, self._graph_exec_checked = None, None, False
                self._la_graphs = {}
                self._graph_runs.clear()
            self._graph_hp = hp

    @unified_elapsed_timer('train a step', 10)
    def train(self) -> int:
        step = self.get_global_step()
        rb = self.replay_buffer
        if not self._ready_to_train():
            self._profiler('train a step').ignore()
            return step
        self._drop_graphs_if_hp_changed()
        if rb._gather_keys is None:
            rb._build_batch()
            self._graph = None
            self._la_graphs = {}

        if self._lookahead:
            if not rb.lookahead:
                rb.enable_lookahead()
            if not rb.next_valid:
                # the very first batch (the reference's thread has drawn two before the first `sample()` returns)
                with torch.cuda.device(self.device):
                    rb.sample_next_into_static()
            rb.swap_sets()
            # the captured step holds the sets' addresses: one graph per arrangement, taken in turn
            self._graph, self._graph_exec, self._graph_exec_checked, self._graph_stats_src = \
                self._la_graphs.get(rb.parity, (None, None, False, None))
            rb.next_valid = False

        with self._profiler('train', repeat=10):
            if self.update_target_per_step != 1 and step % self.update_target_per_step == 0:
                self._update_target_variables(tau=self.tau)
            graph_ok = (self._use_graph and not self._graph_failed and isinstance(self.noise, DeviceNoise)
                        and (self._dist is None or self._graph_collectives))
            if graph_ok and self._graph is None and self._eager_steps >= self._graph_warmup:
                self._try_capture()
            if graph_ok and self._graph is not None:
                self._replay_graph()
            else:
                self._device_step()
                self._eager_steps += 1
        if self._lookahead:
            self._la_graphs[rb.parity] = (self._graph, self._graph_exec, self._graph_exec_checked, self._graph_stats_src)
            rb.next_valid = True

        # host synchronisation points the reference has too: here the NaN flag of the priority update kernels is
        # read back, so a diverged run raises the reference's 'td_error has nan' (replay_buffer.py:418-420) within
        # `write_summary_per_step` steps instead of training on with frozen priorities
        if step % self.write_summary_per_step == 0:
            rb.check_health()
            if self.summary_writer is not None:
                self._write_train_summaries(step)
        if step % self.save_model_per_step == 0:
            self.save_model()
        return self.increase_global_step()

    def train_steps(self, n_steps: int) -> int:
        """`n_steps` consecutive `train()` calls.  Where the step already replays as a hipGraph and nothing has to
        happen on the host between the steps of this run (target update every step, no summary / health check / checkpoint
        falling due inside it), the run is ONE replay of a graph holding `n_steps` steps: the boundary between two graph
        launches costs ~4.6 us more than a kernel boundary inside one (cfg2: 92.9 us per step as one step per launch,
        89.7 as two, 88.3 as four).  Same launches, same order, same device-side counters and draws: the state after the
        run is that of the `train()` calls bit for bit.  Episodes enter between runs, not between the steps of one."""
        k = int(n_steps)
        step = self.get_global_step()
        rb = self.replay_buffer
        self._drop_graphs_if_hp_changed()
        due = any((step + i) % self.write_summary_per_step == 0 or (step + i) % self.save_model_per_step == 0
                  for i in range(k))
        if (k <= 1 or due or self._lookahead or self._graph is None or self._graph_exec is None or self.update_target_per_step != 1
                or not rb.is_lg_batch_size or rb._gather_keys is None):
            for _ in range(k):
                step = self.train()
            return step
        cached = self._graph_runs.get(k)
        if cached is None or cached[0] is not self._graph:      # (captured per run length; dropped with the step's graph)
            try:
                side = torch.cuda.Stream(device=self.device)
                side.wait_stream(torch.cuda.current_stream())
                graph = torch.cuda.CUDAGraph(keep_graph=True)
                with torch.cuda.graph(graph, stream=side, capture_error_mode='thread_local'):
                    for _ in range(k):
                        self._device_step()
                self._finish_graph(graph)
                torch.cuda.current_stream().wait_stream(side)
                cached = self._graph_runs[k] = (self._graph, graph, int(graph.raw_cuda_graph_exec()), self._pi_stats_src)
            except Exception as e:
                torch.cuda.synchronize()
                self._logger.warning(f'hipGraph capture of a {k}-step run failed, replaying single steps: {e!r}')
                self._graph_runs[k] = cached = (self._graph, None, None, None)
        if cached[2] is None:
            for _ in range(k):
                step = self.train()
            return step
        self._pi_stats_src = cached[3]      # the run's last step left its (logp, scale) in the run graph's pool
        with self._profiler('train', repeat=10):
            native.graph_launch(cached[2])
        self.global_step.add_(k)
        return self.global_step.item()

    @torch.no_grad()
    def _refresh_policy_stats(self, log_c_alpha: torch.Tensor | None = None) -> None:
        """The stock policy step forms its gradients on chip and leaves the logged statistics (policy
        objective, Gaussian entropy) to be computed here, on demand, from the step's buffers.  The objective
        is weighted by the temperature: by default the current one (already moved by the step's temperature
        update, which is what a log line sees); `log_c_alpha` = the value the policy step itself used."""
        if self._pi_stats_src is None:
            return
        logp, scale = self._pi_stats_src
        E, Es = self.ensemble_q_num, self.ensemble_q_sample
        native.policy_loss_fwd_bwd(logp, self._pi_q.view(E, -1), self._subsets['pi_c'] if Es != E else None, Es,
                                   self.log_c_alpha if log_c_alpha is None else log_c_alpha, scale,
                                   self._stats['loss_policy'],
                                   torch.empty_like(self._grad_logp), torch.empty_like(self._grad_q),
                                   self._stats['c_entropy'])

    def _write_train_summaries(self, step: int) -> None:
        self.summary_available = True
        self._refresh_policy_stats()
        w = self.summary_writer
        w.add_scalar('metric/replay_id', self.replay_buffer.get_curr_id(), step)
        w.add_scalar('loss/q', self._stats['loss_q'].item(), step)
        if self.d_action_sizes:
            w.add_scalar('loss/d_entropy', self._stats['d_entropy'].item(), step)
            if self.use_auto_alpha:
                w.add_scalar('loss/d_alpha', torch.exp(self.log_d_alpha).item(), step)
        if self.c_action_size:
            w.add_scalar('loss/c_entropy', self._stats['c_entropy'].item(), step)
            if self.use_auto_alpha:
                w.add_scalar('loss/c_alpha', torch.exp(self.log_c_alpha).item(), step)
        if self.curiosity is not None:
            w.add_scalar('loss/curiosity', self._stats['loss_curiosity'].item(), step)
        w.flush()

    def close(self):
        self._closed = True
        # captured graphs go first (they hold the collectives' kernels of a data-parallel step: a process group must
        # not be torn down under them)
        self._graph = self._graph_exec = None
        self._graph_runs.clear()
        if self.device.type == 'cuda':
            torch.cuda.synchronize(self.device)
        if hasattr(self, 'replay_buffer'):
            self.replay_buffer.close()


def nn_parameter_scalar(value: float, device) -> torch.nn.Parameter:
    return torch.nn.Parameter(torch.tensor(value, dtype=torch.float32, device=device), requires_grad=True)
