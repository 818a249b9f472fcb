"""Optional learner heads of `SAC_Base`: siamese representation learning (ATC / BYOL), the
recurrent prediction models, random network distillation, observation normalisation and the
DQN-like discrete target.  They are auxiliary losses on user modules (reference SURVEY.md §8a row
a21: "extra fwd/bwd; stays PyTorch"), run as eager PyTorch-ROCm ops inside the same (capturable)
device step, on parameters that live in the learner's flat buffers so their optimizers are the same
fused Adam launches.

Restates reference `algorithm/sac_base.py`: `calculate_adaptive_weights` 1607-1631,
`_train_siamese_representation_learning` 1633-1796, `_train_rpm` 1798-1839, `_train_rnd` 1978-2025,
`rnd_sample_*` 792-856, `_update_normalizer` 766-781, `get_dqn_like_d_y` 1193-1242.
"""
from itertools import chain

import torch
from torch import autograd, distributions, nn
from torch.nn import functional

from .utils.enums import SEQ_ENCODER, SIAMESE
from .utils.operators import get_last_false_indexes


class _NormalNllKlFn(autograd.Function):
    """-mean(log N(x; loc, scale)) + w mean(KL(N(loc, scale) || N(0, 1))) and the mean entropy as ONE launch that also
    leaves d loss / d loc and d loss / d scale (`asac_normal_nll_kl`); the backward scales them."""

    @staticmethod
    def forward(ctx, loc, scale, target, w):
        from asac_amd import native
        g_loc, g_scale = torch.empty(loc.shape, device=loc.device), torch.empty(loc.shape, device=loc.device)
        out = torch.empty(2, device=loc.device)
        native.normal_nll_kl(loc.detach(), scale.detach(), target.detach(), w, g_loc, g_scale, out)
        ctx.save_for_backward(g_loc, g_scale)
        entropy = out[1]
        ctx.mark_non_differentiable(entropy)
        return out[0], entropy

    @staticmethod
    def backward(ctx, g_loss, _g_entropy):
        g_loc, g_scale = ctx.saved_tensors
        return g_loss * g_loc, g_loss * g_scale, None, None


class _NormalNllKlLogstdFn(autograd.Function):
    """`_NormalNllKlFn` on the transition model's raw output (mean | logstd) [B, T, 2K]: the head's
    clamp(exp(logstd), lo, hi) and its backward run inside the launch (`asac_normal_nll_kl_logstd`)"""

    @staticmethod
    def forward(ctx, raw, target, w, lo, hi):
        from asac_amd import native
        g_raw = torch.empty(raw.shape, device=raw.device)
        out = torch.empty(2, device=raw.device)
        native.normal_nll_kl_logstd(raw.detach(), lo, hi, target.detach(), w, g_raw, out)
        ctx.save_for_backward(g_raw)
        entropy = out[1]
        ctx.mark_non_differentiable(entropy)
        ctx.set_materialize_grads(False)
        return out[0], entropy

    @staticmethod
    def backward(ctx, g_loss, _g_entropy):
        (g_raw,) = ctx.saved_tensors
        if g_loss is None:
            return None, None, None, None, None
        return (g_raw if _is_unit(g_loss) else g_loss * g_raw), None, None, None, None


_UNIT = {}


def unit_gradient(like: torch.Tensor) -> torch.Tensor:
    """a cached scalar 1 on `like`'s device: the root gradient of `autograd.grad(loss, ...)` without the fill launch that
    building `ones_like(loss)` costs on every call; the loss functions of this module recognise it and skip the scaling"""
    key = (like.device, like.dtype)
    one = _UNIT.get(key)
    if one is None:
        one = torch.ones((), dtype=like.dtype, device=like.device)
        # (made during capture: it belongs to that graph's pool and is filled by a captured node — not cached; `_is_unit`
        # then does not recognise it and the loss functions multiply by it, which is the same value)
        if not (like.is_cuda and torch.cuda.is_current_stream_capturing()):
            _UNIT[key] = one
    return one


_CONST = {}


def const_gradient(like: torch.Tensor, value: float) -> torch.Tensor:
    """a cached scalar `value` on `like`'s device — what `_DivConstFn` hands down for a unit root gradient, recognisable by
    `_const_value` (so that a consumer that prepared for exactly this factor needs no launch at all)"""
    key = (like.device, like.dtype, float(value))
    c = _CONST.get(key)
    if c is None:
        c = torch.full((), float(value), dtype=like.dtype, device=like.device)
        if not (like.is_cuda and torch.cuda.is_current_stream_capturing()):
            _CONST[key] = c
    return c


def _const_value(g: torch.Tensor):
    """the host value of `g` if it is one of the cached constant gradients (or the unit), else None"""
    if g.dim() != 0:
        return None
    if _is_unit(g):
        return 1.0
    for (dev, dt, value), c in _CONST.items():
        if dev == g.device and dt == g.dtype and c.data_ptr() == g.data_ptr():
            return value
    return None


class _DivConstFn(torch.autograd.Function):
    """loss / n for a host constant n; a unit root gradient comes out as the cached constant 1 / n (no launch)"""

    @staticmethod
    def forward(ctx, loss, n):
        ctx.n = float(n)
        return loss / n

    @staticmethod
    def backward(ctx, g):
        if _is_unit(g):
            return const_gradient(g, 1.0 / ctx.n), None
        return g / ctx.n, None


def _is_unit(g: torch.Tensor) -> bool:
    one = _UNIT.get((g.device, g.dtype))
    return one is not None and g.data_ptr() == one.data_ptr() and g.dim() == 0


def _stock_transition(model) -> bool:
    """the model's forward is the library's (plugins usually override `_build_model` only)"""
    from .nn_models.predictions import ModelTransition
    return isinstance(model, ModelTransition) and type(model).forward is ModelTransition.forward \
        and type(model).mean_logstd is ModelTransition.mean_logstd


def _normal_nll_kl_raw_ok(raw, target) -> bool:
    from asac_amd import native
    return (raw.is_cuda and raw.dim() == 3 and raw.dtype == torch.float32 and raw.stride(-1) == 1
            and raw.shape[-1] == 2 * target.shape[-1] and raw.shape[:2] == target.shape[:2] and target.dtype == torch.float32
            and target.stride(-1) == 1 and 0 < target.numel() <= native.MASKED_MSE_MAX and not target.requires_grad)


def _normal_nll_kl_ok(dist, target) -> bool:
    from asac_amd import native
    loc, scale = dist.loc, dist.scale
    return (type(dist) is distributions.Normal and loc.is_cuda and loc.dim() == 3 and loc.dtype == torch.float32
            and scale.shape == loc.shape and target.shape == loc.shape and target.dtype == torch.float32
            and loc.stride(-1) == 1 and scale.stride(-1) == 1 and target.stride(-1) == 1
            and 0 < loc.numel() <= native.MASKED_MSE_MAX and not target.requires_grad)


class AuxHeadsMixin:
    # ------------------------------------------------------------------------------------------
    # construction
    # ------------------------------------------------------------------------------------------
    def _wrap_normalized_rep(self, ModelRep):
        """`use_normalization`: observations are whitened by running statistics before the user rep
        (reference 302-336)."""
        dev = self.device
        self.normalizer_step = torch.tensor(0, dtype=torch.int32, device=dev, requires_grad=False)
        self.running_means = [torch.zeros(s, device=dev) for s in self.obs_shapes]
        self.running_variances = [torch.ones(s, device=dev) for s in self.obs_shapes]
        owner = self

        def whiten(obs_list):
            return [torch.clamp((o - m) / torch.sqrt(v / (owner.normalizer_step + 1)), -5, 5)
                    for o, m, v in zip(obs_list, owner.running_means, owner.running_variances)]

        if self.seq_encoder == SEQ_ENCODER.ATTN:
            class NormalizedRep(ModelRep):
                def forward(self, seq_q_len, index, obs_list, *a, **k):
                    return super().forward(seq_q_len, index, whiten(obs_list), *a, **k)
        else:
            class NormalizedRep(ModelRep):
                def forward(self, obs_list, *a, **k):
                    return super().forward(whiten(obs_list), *a, **k)
        NormalizedRep.__name__ = ModelRep.__name__
        return NormalizedRep

    def _build_aux(self, nn_mod, test_obs_list) -> list:
        """Creates the optional modules; returns their (segment name, parameters) for the flat buffer."""
        dev, named = self.device, []
        A_all = self.d_action_summed_size + self.c_action_size
        if self.siamese in (SIAMESE.ATC, SIAMESE.BYOL):
            with torch.no_grad():
                enc = self.model_rep.get_augmented_encoders(test_obs_list)
            enc = enc if isinstance(enc, tuple) else (enc,)
            if self.siamese == SIAMESE.ATC:
                self.contrastive_weight_list = [
                    nn.Parameter(torch.randn((e.shape[-1], e.shape[-1]), device=dev)) for e in enc]
                named.append(('siamese', list(self.contrastive_weight_list)))
            else:
                self.model_rep_projection_list = [nn_mod.ModelRepProjection(e.shape[-1]).to(dev) for e in enc]
                self.model_target_rep_projection_list = [nn_mod.ModelRepProjection(e.shape[-1]).to(dev) for e in enc]
                with torch.no_grad():
                    proj = [p(e) for p, e in zip(self.model_rep_projection_list, enc)]
                self.model_rep_prediction_list = [nn_mod.ModelRepPrediction(p.shape[-1]).to(dev) for p in proj]
                named.append(('siamese', list(chain(*[m.parameters() for m in self.model_rep_projection_list],
                                                    *[m.parameters() for m in self.model_rep_prediction_list]))))
        if self.use_prediction:
            self.model_transition = nn_mod.ModelTransition(self.state_size, self.d_action_summed_size,
                                                           self.c_action_size, self.use_extra_data).to(dev)
            self.model_reward = nn_mod.ModelReward(self.state_size).to(dev)
            self.model_observation = nn_mod.ModelObservation(self.state_size, self.obs_shapes,
                                                             self.use_extra_data).to(dev)
            # dense stacks inside the (user) prediction models run as fused launches when they fit (fused_mlp.fused_dense
            # falls back to the module path otherwise)
            from .nn_models.layers.linear_layers import LinearLayers
            if self._fuse_prediction_dense:
                for mod in (self.model_transition, self.model_reward, self.model_observation):
                    for sub in mod.modules():
                        if isinstance(sub, LinearLayers):
                            sub.fuse = True
            named.append(('prediction', list(chain(self.model_transition.parameters(), self.model_reward.parameters(),
                                                   self.model_observation.parameters()))))
        if self.use_rnd:
            self.model_rnd = nn_mod.ModelRND(self.state_size, self.d_action_summed_size, self.c_action_size).to(dev)
            self.model_target_rnd = nn_mod.ModelRND(self.state_size, self.d_action_summed_size,
                                                    self.c_action_size).to(dev)
            for p in self.model_target_rnd.parameters():
                p.requires_grad = False
            named.append(('rnd', list(self.model_rnd.parameters())))
        return named

    def _aux_ckpt(self, ck: dict) -> None:
        """reference `_build_ckpt` 498-560 entries for the optional heads"""
        if self.use_normalization:
            ck['normalizer_step'] = self.normalizer_step
            for i, v in enumerate(self.running_means):
                ck[f'running_means_{i}'] = v
            for i, v in enumerate(self.running_variances):
                ck[f'running_variances_{i}'] = v
        if self.siamese == SIAMESE.ATC:
            for i, w in enumerate(self.contrastive_weight_list):
                ck[f'contrastive_weights_{i}'] = w
            ck['optimizer_siamese'] = self.optimizer_siamese
        elif self.siamese == SIAMESE.BYOL:
            for i, m in enumerate(self.model_rep_projection_list):
                ck[f'model_rep_projection_{i}'] = m
            for i, m in enumerate(self.model_target_rep_projection_list):
                ck[f'model_target_rep_projection_{i}'] = m
            for i, m in enumerate(self.model_rep_prediction_list):
                ck[f'model_rep_prediction_{i}'] = m
            ck['optimizer_siamese'] = self.optimizer_siamese
        if self.use_prediction:
            ck['model_transition'], ck['model_reward'] = self.model_transition, self.model_reward
            ck['model_observation'], ck['optimizer_prediction'] = self.model_observation, self.optimizer_prediction
        if self.use_rnd:
            ck['model_rnd'], ck['model_target_rnd'], ck['optimizer_rnd'] = \
                self.model_rnd, self.model_target_rnd, self.optimizer_rnd

    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def _update_normalizer(self, obs_list) -> None:
        """Welford-style running mean / (unnormalised) variance over the episode rows (766-781)."""
        self.normalizer_step.add_(obs_list[0].shape[0])
        for i, obs in enumerate(obs_list):
            to_old = obs - self.running_means[i]
            new_mean = self.running_means[i] + torch.sum(to_old / self.normalizer_step, dim=0)
            new_var = self.running_variances[i] + torch.sum((obs - new_mean) * to_old, dim=0)
            self.running_means[i].copy_(new_mean)
            self.running_variances[i].copy_(new_var)

    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def calculate_adaptive_weights(self, grads_main, loss_list, model) -> None:
        """Adds each auxiliary gradient to `model`'s .grad only when it does not oppose the main
        gradient (cosine sign gating), reference 1607-1631."""
        params = list(model.parameters())
        with torch.enable_grad():
            aux = [autograd.grad(loss, params, allow_unused=True, retain_graph=True) for loss in loss_list]
        self._add_gated_gradients(grads_main, aux, params)

    @torch.no_grad()
    def _add_gated_gradients(self, grads_main, aux, params) -> None:
        """The second half of `calculate_adaptive_weights` (reference 1619-1631): `aux` = one gradient list per loss"""
        aux = [[ga if ga is not None else torch.zeros_like(gm) for gm, ga in zip(grads_main, gs)] for gs in aux]
        flat_main = torch.cat([g.reshape(1, -1) for g in grads_main], dim=1)
        # inside the learner the gradients are consecutive views of ONE buffer: the gated sum is then two launches per
        # loss over that buffer instead of two per parameter tensor (same products, same sums, entry by entry)
        flat_grad = None
        if all(p.grad is not None for p in params) and params[0].is_cuda:
            from .fused_mlp import _flat_alias
            flat_grad = _flat_alias([p.grad for p in params])
        if (flat_grad is not None and self._fused_gating and flat_main.numel() == flat_grad.numel()
                and len(aux) <= 4 and flat_grad.numel() <= (1 << 20)):
            # the K cosines' signs and the gated additions, loss by loss, as ONE launch (csrc/optim.hip k_cosine_gate_add)
            from asac_amd import native
            native.cosine_gate_add(flat_main.view(-1), [torch.cat([g.reshape(-1) for g in gs]) for gs in aux], flat_grad)
            return
        for gs in aux:
            flat_aux = torch.cat([g.reshape(1, -1) for g in gs], dim=1)
            cos = functional.cosine_similarity(flat_main, flat_aux)
            gate = torch.sign(cos).clamp(min=0)
            if flat_grad is not None and flat_aux.numel() == flat_grad.numel():
                flat_grad.add_(flat_aux.view(-1).mul_(gate))
                continue
            for p, g in zip(params, gs):
                p.grad += gate * g

    def _train_siamese_representation_learning(self, grads_rep_main, grads_q_main_list, n_indexes, n_padding_masks,
                                               n_obses_list, n_pre_actions, n_pre_seq_hidden_states):
        """ATC (bilinear InfoNCE-style logits against the target encoder) or BYOL (predict the target
        projection), optionally with a Q-consistency term; reference 1633-1796.
        -> (loss_siamese, loss_siamese_q | None)"""
        if not any(p.requires_grad for p in self.model_rep.parameters()):
            return None, None
        enc = self.model_rep.get_augmented_encoders(n_obses_list)
        t_enc = self.model_target_rep.get_augmented_encoders(n_obses_list)
        if not isinstance(enc, tuple):
            enc, t_enc = (enc,), (t_enc,)
        batch, n = enc[0].shape[:2]
        flat = lambda xs: [x.reshape(batch * n, -1) for x in xs]  # noqa: E731

        if self.siamese == SIAMESE.ATC:
            logits = [torch.mm(torch.mm(e, w), te.t())
                      for e, w, te in zip(flat(enc), self.contrastive_weight_list, flat(t_enc))]
            labels = torch.block_diag(*torch.ones(batch, n, n, device=self.device))
            pad = n_padding_masks.reshape(batch * n, 1)
            loss_siamese_list = [
                (functional.binary_cross_entropy_with_logits(lg, labels, reduction='none') * pad).mean()
                for lg in logits]
        else:
            proj = [p(e) for p, e in zip(self.model_rep_projection_list, flat(enc))]
            pred = [p(x) for p, x in zip(self.model_rep_prediction_list, proj)]
            t_proj = [p(e) for p, e in zip(self.model_target_rep_projection_list, flat(t_enc))]
            pad = n_padding_masks.reshape(batch * n)
            loss_siamese_list = [(functional.cosine_similarity(a, b) * pad).mean() for a, b in zip(pred, t_proj)]

        q_loss_list = []
        if self.siamese_use_q:
            obs_at = [o[:, 0:1] for o in n_obses_list]
            e_at = [e[:, 0:1] for e in enc]
            te_at = [e[:, 0:1] for e in t_enc]
            e_arg = e_at if len(e_at) > 1 else e_at[0]
            pre_a, pad_at = n_pre_actions[:, 0:1], n_padding_masks[:, 0:1]
            if self.seq_encoder == SEQ_ENCODER.ATTN:
                idx_at = n_indexes[:, 0:1]
                state = self.model_rep.get_state_from_encoders(1, e_arg, idx_at, obs_at, pre_a, None,
                                                               padding_mask=pad_at)
                # (the reference feeds the ONLINE encoders to the target rep here, 1730-1736)
                t_state = self.model_target_rep.get_state_from_encoders(1, e_arg, idx_at, obs_at, pre_a, None,
                                                                        padding_mask=pad_at)
            else:
                hid_at = n_pre_seq_hidden_states[:, 0:1]
                te_arg = te_at if len(te_at) > 1 else te_at[0]
                state = self.model_rep.get_state_from_encoders(e_arg, obs_at, pre_a, hid_at, padding_mask=pad_at)
                t_state = self.model_target_rep.get_state_from_encoders(te_arg, obs_at, pre_a, hid_at,
                                                                        padding_mask=pad_at)
            state, t_state = state[:, 0], t_state[:, 0]
            obs0 = [o[:, 0] for o in n_obses_list]
            d_action = n_pre_actions[:, 1, :self.d_action_summed_size]
            c_action = n_pre_actions[:, 1, self.d_action_summed_size:]
            qs = [q(state, c_action, obs0) for q in self.model_q_list]
            t_qs = [q(t_state, c_action, obs0) for q in self.model_target_q_list]
            if self.d_action_sizes:
                pick = lambda out: torch.sum(d_action * out[0], dim=-1) / self.d_action_branch_size  # noqa: E731
                q_loss_list += [functional.mse_loss(pick(a), pick(b)) for a, b in zip(qs, t_qs)]
            if self.c_action_size:
                q_loss_list += [functional.mse_loss(a[1], b[1]) for a, b in zip(qs, t_qs)]
            if self.siamese_use_adaptive:
                for g_main, ql, q in zip(grads_q_main_list, q_loss_list, self.model_q_list):
                    self.calculate_adaptive_weights(g_main, [ql], q)
            else:
                for ql, q in zip(q_loss_list, self.model_q_list):
                    ql.backward(inputs=list(q.parameters()), retain_graph=True)

        loss_list = loss_siamese_list + q_loss_list
        loss = sum(loss_list)
        if self.siamese_use_adaptive:
            self.calculate_adaptive_weights(grads_rep_main, loss_list, self.model_rep)
        else:
            loss.backward(inputs=list(self.model_rep.parameters()), retain_graph=True)

        self.optimizer_siamese.zero_grad()
        own = list(self.contrastive_weight_list) if self.siamese == SIAMESE.ATC else \
            list(chain(*[m.parameters() for m in self.model_rep_projection_list],
                       *[m.parameters() for m in self.model_rep_prediction_list]))
        loss.backward(inputs=own, retain_graph=True)
        if self._dist is not None:
            self._dist.all_reduce_grads(self._params.grad, *self._params.span('siamese'))
        self.optimizer_siamese.step()
        return sum(loss_siamese_list), (sum(q_loss_list) if self.siamese_use_q else None)

    def _train_rpm(self, grads_rep_main, nx_obses_list, nx_states, nx_target_states, n_actions, n_rewards):
        """Transition / reward / observation models on the step's states (reference 1798-1839).
        The reference back-propagates through the representation graph a second time here, which
        PyTorch refuses once the Q loss has freed it; this build keeps that graph alive
        (`retain_graph` on the Q loss) so the head runs."""
        n_obs = [o[:, :-1] for o in nx_obses_list]
        entropy_next = dist_next = None
        target_next = nx_target_states[:, 1:]
        if self._fused_rpm_loss and _stock_transition(self.model_transition):
            # the library's own transition model: its raw (mean | logstd) output goes to the loss launch, which applies the
            # head's exp / clamp and their backward itself
            from .fused import time_slice
            mt = self.model_transition
            raw = mt.mean_logstd(n_obs, time_slice(nx_states, 0, -1), n_actions)
            if _normal_nll_kl_raw_ok(raw, target_next):
                loss_transition, entropy_next = _NormalNllKlLogstdFn.apply(raw, target_next, float(self.transition_kl),
                                                                          float(mt.SCALE_MIN), float(mt.SCALE_MAX))
            else:
                mean, logstd = torch.chunk(raw, 2, dim=-1)
                dist_next = distributions.Normal(mean, torch.clamp(torch.exp(logstd), mt.SCALE_MIN, mt.SCALE_MAX),
                                                 validate_args=False)
        else:
            dist_next = self.model_transition(n_obs, nx_states[:, :-1], n_actions)
        if entropy_next is not None:
            pass
        elif self._fused_rpm_loss and _normal_nll_kl_ok(dist_next, nx_target_states[:, 1:]):
            loss_transition, entropy_next = _NormalNllKlFn.apply(dist_next.loc, dist_next.scale, nx_target_states[:, 1:],
                                                                 float(self.transition_kl))
        else:
            loss_transition = -torch.mean(dist_next.log_prob(nx_target_states[:, 1:]))
            std_normal = distributions.Normal(torch.zeros_like(dist_next.loc), torch.ones_like(dist_next.scale),
                                              validate_args=False)
            loss_transition = loss_transition + self.transition_kl * torch.mean(
                distributions.kl.kl_divergence(dist_next, std_normal))
        if self._fused_rpm_loss:
            from .fused import scaled_mse, time_slice
            loss_reward = scaled_mse(self.model_reward(time_slice(nx_states, 1)), n_rewards.unsqueeze(2), self.n_step)
        else:
            loss_reward = functional.mse_loss(self.model_reward(nx_states[:, 1:]), n_rewards.unsqueeze(2)) / self.n_step
        if self._fused_rpm_loss and nx_states.is_cuda:
            from .fused import fused_mse_loss
            ws = getattr(self, '_mse_big_ws', None)
            if ws is None:       # (first eager step: zeroed exchange words of `asac_mse_mean_grad`, kept by the learner)
                from asac_amd import native
                ws = self._mse_big_ws = torch.zeros(native.mse_mean_grad_workspace(), dtype=torch.float32, device=self.device)
            # (the frame loss prepares its gradient for the division below: no second pass over the frames' gradient)
            with fused_mse_loss(ws, grad_scale=1.0 / self.n_step):
                loss_obs = _DivConstFn.apply(self.model_observation.get_loss(nx_states, list(nx_obses_list)), self.n_step)
        else:
            loss_obs = self.model_observation.get_loss(nx_states, list(nx_obses_list)) / self.n_step
        model_params = [list(mod.parameters()) for mod in (self.model_transition, self.model_reward, self.model_observation)]
        pred_params = list(chain(*model_params))
        losses = [loss_transition, loss_reward, loss_obs]
        if grads_rep_main and self._rpm_single_backward and nx_states.requires_grad and self._rpm_models_disjoint():
            # Every loss reaches the representation through `nx_states` alone and only its own model's parameters, so
            # ONE walk through each model yields both what the reference's two walks do (1827: d loss_i / d rep for the
            # gates; 1831-1834: d (sum of losses) / d model_i = d loss_i / d model_i): the walk stops at `nx_states`,
            # the representation's graph is entered from there once per loss as in `calculate_adaptive_weights`.
            rep_params = list(self.model_rep.parameters())
            aux, gs = [], []
            # the convolution stack is the leaf of each of the three walks through the representation: its backward waits
            # until all three output gradients are known and runs as ONE launch (fused_conv.DeferredConvBackward)
            from .fused_conv import DeferredConvBackward
            conv_later = DeferredConvBackward() if self._rpm_conv_one_launch else None
            # ... and none of the walks' parameter gradients is read before all walks are done: the fixed-order sums of the
            # backward launches' per-workgroup partials (twelve second launches) wait too and run as one launch
            # (fused_mlp.DeferredPartialSums; walk ('m', k): through model k down to `nx_states`, ('r', k): the representation)
            from .fused_mlp import DeferredPartialSums
            with DeferredPartialSums() as sums_later:
                for k, (loss_i, params_i) in enumerate(zip(losses, model_params)):
                    sums_later.walk = ('m', k)
                    got = autograd.grad(loss_i, [nx_states, *params_i], grad_outputs=unit_gradient(loss_i), allow_unused=True,
                                        retain_graph=True)
                    gs += got[1:]
                    sums_later.walk = ('r', k)
                    if got[0] is None:
                        aux.append([None] * len(rep_params))
                    elif conv_later is None:
                        aux.append(list(autograd.grad(nx_states, rep_params, grad_outputs=got[0], allow_unused=True,
                                                      retain_graph=True)))
                    else:
                        conv_later.walk = len(aux)
                        with conv_later:
                            aux.append(list(autograd.grad(nx_states, rep_params, grad_outputs=got[0], allow_unused=True,
                                                          retain_graph=True)))
                conv_late = conv_later.flush() if conv_later is not None else {}     # (its slab sums join the others)

            def put(grads, params, late):       # a deferred launch's gradients where autograd left None (or beside its own)
                for j, p_ in enumerate(params):
                    g_late = late.get(id(p_))
                    if g_late is not None:
                        grads[j] = g_late if grads[j] is None else grads[j] + g_late

            late, off = sums_later.flush(), 0
            for k, params_i in enumerate(model_params):
                put_into = gs[off:off + len(params_i)]
                put(put_into, params_i, late.get(('m', k), {}))
                gs[off:off + len(params_i)] = put_into
                off += len(params_i)
                put(aux[k], rep_params, late.get(('r', k), {}))
            for walk, grads in conv_late.items():
                put(aux[walk], rep_params, grads)
            self._add_gated_gradients(grads_rep_main, aux, rep_params)
        else:
            if grads_rep_main:
                self.calculate_adaptive_weights(grads_rep_main, losses, self.model_rep)
            gs = None
        loss = loss_transition + loss_reward + loss_obs
        flat_grad = None
        if pred_params and pred_params[0].is_cuda and all(p.grad is not None for p in pred_params):
            from .fused_mlp import _flat_alias
            flat_grad = _flat_alias([p.grad for p in pred_params])
        if flat_grad is not None:
            # the models' gradients are consecutive views of one buffer: written there by ONE concatenation instead of a
            # zero fill and an accumulation launch per parameter tensor (0 + g = g)
            if gs is None:
                gs = autograd.grad(loss, pred_params, allow_unused=True)
            torch.cat([(g if g is not None else torch.zeros_like(p)).reshape(-1) for g, p in zip(gs, pred_params)],
                      out=flat_grad)
        elif gs is not None:
            self.optimizer_prediction.zero_grad()
            for p, g in zip(pred_params, gs):
                if g is not None:
                    p.grad = g if p.grad is None else p.grad.copy_(g)
        else:
            self.optimizer_prediction.zero_grad()
            loss.backward(inputs=pred_params)
        if self._dist is not None:
            self._dist.all_reduce_grads(self._params.grad, *self._params.span('prediction'))
        self.optimizer_prediction.step()
        if entropy_next is None:
            entropy_next = torch.mean(dist_next.entropy())
        return entropy_next.detach(), loss_reward.detach(), loss_obs.detach()

    def _rpm_models_disjoint(self) -> bool:
        """The one-walk form of `_train_rpm` assumes that a prediction loss touches only its own model's parameters and
        reaches the representation through `nx_states` alone.  A plugin whose prediction models share a sub-module or a
        parameter with each other, or with the representation (an encoder reused through `extra_obs`), breaks that: the
        shared terms would drop out of the gates and of the models' gradients.  Checked once: parameter sets pairwise
        disjoint and disjoint from the representation's — otherwise the reference's two-walk form
        (`calculate_adaptive_weights` + the summed backward, reference sac_base.py:1827-1834) runs."""
        ok = getattr(self, '_rpm_disjoint', None)
        if ok is None:
            sets = [{id(p) for p in m.parameters()}
                    for m in (self.model_transition, self.model_reward, self.model_observation, self.model_rep)]
            ok = all(not (sets[i] & sets[j]) for i in range(len(sets)) for j in range(i + 1, len(sets)))
            if not ok:
                self._logger.info('prediction models share parameters (with each other or the representation): '
                                  '`_train_rpm` uses the two-walk form')
            self._rpm_disjoint = ok
        return ok

    def _train_rnd(self, n_padding_masks, n_states, n_actions):
        """Distil the frozen random target network on visited (state, action) pairs (1978-2025)."""
        dsum = self.d_action_summed_size
        d_act, c_act = n_actions[..., :dsum], n_actions[..., dsum:]
        keep = ~n_padding_masks.unsqueeze(-1)
        loss = torch.scalar_tensor(0., device=self.device)

        def masked_mse(a, b):
            return torch.mean(functional.mse_loss(a, b, reduction='none') * keep)

        if self.d_action_sizes:
            if self.discrete_dqn_like:
                s = torch.sigmoid(self.model_rnd.cal_s_rnd(n_states))
                with torch.no_grad():
                    t = torch.sigmoid(self.model_target_rnd.cal_s_rnd(n_states))
                loss = loss + masked_mse(s, t)
            else:
                sel = d_act.unsqueeze(-1)
                d = (sel * self.model_rnd.cal_d_rnd(n_states)).sum(-2)
                with torch.no_grad():
                    t = (sel * self.model_target_rnd.cal_d_rnd(n_states)).sum(-2)
                loss = loss + masked_mse(d, t)
        if self.c_action_size:
            c = self.model_rnd.cal_c_rnd(n_states, c_act)
            with torch.no_grad():
                t = self.model_target_rnd.cal_c_rnd(n_states, c_act)
            loss = loss + masked_mse(c, t)
        self.optimizer_rnd.zero_grad()
        loss.backward(inputs=list(self.model_rnd.parameters()))
        if self._dist is not None:
            self._dist.all_reduce_grads(self._params.grad, *self._params.span('rnd'))
        self.optimizer_rnd.step()
        return loss.detach()

    # ------------------------------------------------------------------------------------------
    # RND-guided action sampling (acting path, 792-856)
    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def rnd_sample_d_action(self, state, d_policy):
        batch, k = state.shape[0], self.rnd_n_sample
        acts = d_policy.sample((k,)).transpose(0, 1)                       # [batch, k, D]
        sel = acts.unsqueeze(-1)
        d = (sel * self.model_rnd.cal_d_rnd(state).unsqueeze(1)).sum(-2)       # [batch, k, f]
        t = (sel * self.model_target_rnd.cal_d_rnd(state).unsqueeze(1)).sum(-2)
        best = torch.argmax(torch.sum(torch.pow(d - t, 2), dim=-1), dim=1)
        return acts[torch.arange(batch), best]

    @torch.no_grad()
    def rnd_sample_c_action(self, state, c_policy):
        batch, k = state.shape[0], self.rnd_n_sample
        acts = torch.tanh(c_policy.sample((k,))).transpose(0, 1)           # [batch, k, A]
        states = state.unsqueeze(1).expand(-1, k, -1)
        err = torch.sum(torch.pow(self.model_rnd.cal_c_rnd(states, acts)
                                  - self.model_target_rnd.cal_c_rnd(states, acts), 2), dim=-1)
        return acts[torch.arange(batch), torch.argmax(err, dim=1)]

    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def get_dqn_like_d_y(self, n_last_masks, n_padding_masks, n_rewards, n_dones, stacked_next_n_d_qs,
                         stacked_next_target_n_d_qs):
        """Double-DQN n-step target at the last valid step of each row (reference 1193-1242).
        stacked_*: [E_sample, batch, n, D] -> y [batch, 1]"""
        rows = torch.arange(n_padding_masks.shape[0], device=self.device)
        last = get_last_false_indexes(torch.logical_or(n_last_masks, n_padding_masks), dim=1)
        done = n_dones[rows, last].unsqueeze(-1)
        next_q = stacked_next_n_d_qs[:, rows, last, :]
        next_t = stacked_next_target_n_d_qs[:, rows, last, :]
        greedy = torch.cat([functional.one_hot(torch.argmax(part, dim=-1), size)
                            for part, size in zip(next_q.split(self.d_action_sizes, dim=-1), self.d_action_sizes)],
                           dim=-1)
        picked = torch.sum(next_t * greedy, dim=-1, keepdim=True) / self.d_action_branch_size
        next_v, _ = torch.min(picked, dim=0)
        g = torch.sum(self._gamma_ratio * n_rewards, dim=-1, keepdim=True)
        return g + torch.pow(self.gamma, last.unsqueeze(-1) + 1) * next_v * ~done
