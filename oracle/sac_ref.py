"""ORACLE (test infrastructure, not product code) — CPU restatement of the reference's SAC
learner step, `SAC_Base.train()` (reference `algorithm/sac_base.py:2496-2609`), in plain eager
PyTorch on the host plus the NumPy replay of `oracle/per_ref.py`.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
module; the product path never does.  It doubles as the "port" CPU baseline of bench.py.

Parity pinning: `tests/test_oracle_golden.py` drives this class with the weights, episodes and
random draws recorded from the imported reference (`tests/golden/f6_step_*.npz`, minted by
`tests/golden/make_golden.py`) and requires the same sampled ids, losses, td-errors, tree bytes
and post-step weights.

Each function cites the reference lines it restates.  Every random draw (stratified uniforms,
Gaussian noise, ensemble permutations) comes from `self.noise`, so recorded draws can be replayed.
Scope: vector/any observations through user `ModelRep`, continuous and/or discrete (non
DQN-like) actions, n-step V-trace, ensemble-min, PER, seq_encoder None / RNN / ATTN, optional
FORWARD/INVERSE curiosity, and the recurrent prediction models (`use_prediction`: `_train_rpm` with the cosine-sign
gating of `calculate_adaptive_weights`, pinned by `tests/golden/f11_rpm.npz`).  Siamese / RND heads are compared
against their own recorded reference steps directly (tests/test_sac_aux_gpu.py).
"""
import numpy as np
import torch
from torch import nn, optim

from .per_ref import PrioritizedReplayRef

SQUASH_FLOOR = 1e-2


# ---------------------------------------------------------------------------------------------
# random draws
# ---------------------------------------------------------------------------------------------
class TorchNoise:
    """Default source: NumPy global RNG for the stratified uniforms, torch CPU generator else."""

    def uniforms(self, n):
        return np.random.random_sample(n)

    def standard_normal(self, shape, dtype=torch.float32):
        return torch.randn(shape, dtype=dtype)

    def permutation(self, n):
        return torch.randperm(n)


class RecordedNoise:
    """Replays draws captured from the reference run (tests/golden/ref_shims.DrawRecorder)."""

    def __init__(self, u=(), eps=(), perm=()):
        self.u, self.eps, self.perm = list(u), list(eps), list(perm)

    def uniforms(self, n):
        u = np.asarray(self.u.pop(0), dtype=np.float64)
        assert u.shape == (n,)
        return u

    def standard_normal(self, shape, dtype=torch.float32):
        e = torch.as_tensor(self.eps.pop(0), dtype=dtype)
        assert tuple(e.shape) == tuple(shape), (e.shape, shape)
        return e

    def permutation(self, n):
        p = torch.as_tensor(self.perm.pop(0), dtype=torch.int64)
        assert p.shape == (n,)
        return p


# ---------------------------------------------------------------------------------------------
# elementwise pieces (reference algorithm/utils/operators.py:12-36)
# ---------------------------------------------------------------------------------------------
def _jac(x):
    return torch.maximum(1 - torch.square(torch.tanh(x)), torch.tensor(SQUASH_FLOOR))


def squash_log_prob(dist, x):  # operators.py:12-14 (correction summed, then broadcast per dim)
    return dist.log_prob(x) - torch.sum(torch.log(_jac(x)), dim=-1, keepdim=True)


def squash_prob(dist, x):  # operators.py:17-19
    return torch.exp(dist.log_prob(x)) / torch.prod(_jac(x), dim=-1, keepdim=True)


def masked_sum_log_prob(lp, keepdim=False):  # operators.py:22-24
    lp = torch.where(lp == torch.inf, torch.zeros_like(lp), lp)
    return lp.sum(-1, keepdim=keepdim)


def masked_prod_prob(p):  # operators.py:27-31 — overwrites +-inf entries of its INPUT (the
    # reference's in-place behaviour; the input can alias the pi-prob tensor written back to replay)
    p[torch.isinf(p)] = 1.
    out = p.prod(-1)
    return torch.where(torch.isfinite(out), out, torch.ones_like(out))


def masked_sum_entropy(e):  # operators.py:34-36
    e = torch.where(e == torch.inf, torch.zeros_like(e), e)
    return e.sum(-1)


def pre_actions_keep_last(actions):  # operators.py:39-52 with keep_last_action=True
    if actions.shape[1] == 0:
        return actions.new_zeros((actions.shape[0], 1, *actions.shape[2:]))
    return torch.cat([torch.zeros_like(actions[:, :1]), actions], dim=1)


def v_trace(*, gamma, gamma_ratio, lambda_ratio, v_rho, v_c, use_n_step_is,
            n_last_masks, n_padding_masks, n_rewards, n_dones, n_mu_probs, n_pi_probs,
            n_vs, next_n_vs):
    """sac_base.py:1244-1295.  All [B, n]; returns y [B, 1]."""
    td = n_rewards + gamma * ~n_dones * next_n_vs - n_vs
    td = gamma_ratio * td
    if use_n_step_is:
        td = lambda_ratio * td
        ratio = n_pi_probs / n_mu_probs.clamp(min=1e-8)
        rho = torch.minimum(ratio, v_rho)
        c = torch.minimum(ratio, v_c)
        c = torch.cat([torch.ones((ratio.shape[0], 1)), c[..., :-1]], dim=-1)
        c = torch.cumprod(c, dim=1)
        td = c * rho * td
    td = td * ~(torch.logical_or(n_last_masks, n_padding_masks))
    return n_vs[:, 0:1] + torch.sum(td, dim=1, keepdim=True)


def pad_window(batch: dict, burn_in_step: int, padding_action: torch.Tensor) -> None:
    """Episode-continuity padding of a sampled [B, L] window, in place (sac_base.py:2435-2453)."""
    idx = batch['index']
    L = idx.shape[1]
    rel = torch.arange(L) - burn_in_step
    invalid = (idx - idx[:, burn_in_step].unsqueeze(1)) != rel.unsqueeze(0)
    invalid[:, burn_in_step] = False
    batch['padding_mask'] = torch.zeros_like(batch['last_mask'])
    batch['index'][invalid] = -1
    batch['padding_mask'][invalid] = True
    batch['action'][invalid] = padding_action
    batch['reward'][invalid] = 0.
    batch['done'][invalid] = True
    batch['mu_prob'][invalid] = 1.
    batch['pre_seq_hidden_state'][invalid] = 0.


# ---------------------------------------------------------------------------------------------
class SacRef:
    """Eager CPU SAC learner with the reference's update order.  Construct with the same
    arguments as `SAC_Base` (subset)."""

    def __init__(self, obs_names, obs_shapes, d_action_sizes, c_action_size, nn_module, *,
                 nn_config=None, ensemble_q_num=2, ensemble_q_sample=2, burn_in_step=0, n_step=1,
                 seq_encoder=None, batch_size=256, tau=0.005, update_target_per_step=1,
                 init_log_alpha=-2.3, use_auto_alpha=True, target_d_alpha=0.98, target_c_alpha=1.,
                 d_policy_entropy_penalty=0.5, learning_rate=3e-4, gamma=0.99, v_lambda=1.,
                 v_rho=1., v_c=1., clip_epsilon=0.2, use_n_step_is=True, use_priority=True,
                 curiosity=None, curiosity_strength=1., use_prediction=False, transition_kl=0.8, use_extra_data=True,
                 replay_config=None, noise=None, lookahead=False):
        self.lookahead, self._queued = bool(lookahead), None
        self.obs_names, self.obs_shapes = list(obs_names), list(obs_shapes)
        self.d_action_sizes, self.c_action_size = list(d_action_sizes), c_action_size
        self.d_sum, self.d_branches = sum(d_action_sizes), len(d_action_sizes)
        self.E, self.E_sample = ensemble_q_num, ensemble_q_sample
        self.b, self.n = burn_in_step, n_step
        self.seq_encoder = seq_encoder  # None | 'RNN' | 'ATTN' (enum .name accepted)
        if seq_encoder is not None and not isinstance(seq_encoder, str):
            self.seq_encoder = seq_encoder.name
        self.batch_size, self.tau = batch_size, tau
        self.update_target_per_step = update_target_per_step
        self.use_auto_alpha = use_auto_alpha
        self.target_c_alpha = target_c_alpha
        self.d_policy_entropy_penalty = d_policy_entropy_penalty
        self.gamma, self.clip_epsilon = gamma, clip_epsilon
        self.use_n_step_is, self.use_priority = use_n_step_is, use_priority
        self.curiosity = curiosity if curiosity is None or isinstance(curiosity, str) else curiosity.name
        self.curiosity_strength = curiosity_strength
        self.use_prediction, self.transition_kl = use_prediction, transition_kl
        self.noise = noise or TorchNoise()
        self.global_step = 0

        nn_config = dict(nn_config or {})
        rep_kw, pol_kw = nn_config.get('rep') or {}, nn_config.get('policy') or {}

        # sac_base.py:285-294
        self.gamma_ratio = torch.logspace(0, n_step - 1, n_step, gamma)
        self.lambda_ratio = torch.logspace(0, n_step - 1, n_step, v_lambda)
        self.v_rho, self.v_c = torch.tensor(v_rho), torch.tensor(v_c)
        pad = [np.eye(s, dtype=np.float32)[0] for s in d_action_sizes] + [np.zeros(c_action_size, np.float32)]
        self.padding_action = torch.from_numpy(np.concatenate(pad, axis=-1))

        adam = lambda ps: optim.Adam(ps, lr=learning_rate) if len(ps) else None  # noqa: E731

        # sac_base.py:338-393
        mk_rep = lambda tgt: nn_module.ModelRep(obs_names, obs_shapes, d_action_sizes,  # noqa: E731
                                                c_action_size, tgt, None, **rep_kw)
        self.model_rep, self.model_target_rep = mk_rep(False), mk_rep(True)
        obs = [torch.rand(batch_size, 1, *s) for s in obs_shapes]
        pre_a = torch.rand(batch_size, 1, self.d_sum + c_action_size)
        if self.seq_encoder == 'ATTN':
            st, hs, _ = self.model_rep(1, torch.zeros((batch_size, 1), dtype=torch.int32), obs, pre_a, None)
        else:
            st, hs = self.model_rep(obs, pre_a, None)
        self.state_size, self.seq_hidden_state_shape = st.shape[-1], tuple(hs.shape[2:])
        for p in self.model_target_rep.parameters():
            p.requires_grad = False
        self.optimizer_rep = adam(list(self.model_rep.parameters()))

        # sac_base.py:395-419
        mk_q = lambda tgt: nn_module.ModelQ(self.state_size, d_action_sizes, c_action_size, tgt, None)  # noqa: E731
        self.model_q_list = [mk_q(False) for _ in range(self.E)]
        self.model_target_q_list = [mk_q(True) for _ in range(self.E)]
        for q in self.model_target_q_list:
            for p in q.parameters():
                p.requires_grad = False
        self.optimizer_q_list = [adam(list(q.parameters())) for q in self.model_q_list]
        self.model_policy = nn_module.ModelPolicy(self.state_size, d_action_sizes, c_action_size, None, **pol_kw)
        self.optimizer_policy = adam(list(self.model_policy.parameters()))

        # sac_base.py:459-483
        self.log_d_alpha = torch.tensor(init_log_alpha, dtype=torch.float32, requires_grad=True)
        self.log_c_alpha = torch.tensor(init_log_alpha, dtype=torch.float32, requires_grad=True)
        if d_action_sizes:
            s = torch.tensor(d_action_sizes)
            s = torch.repeat_interleave(s.type(torch.float32), s)
            self.target_d_alpha = target_d_alpha * (-torch.log(1 / s))
        if use_auto_alpha:
            self.optimizer_alpha = adam([self.log_d_alpha, self.log_c_alpha])
        if self.curiosity == 'FORWARD':
            self.model_forward_dynamic = nn_module.ModelForwardDynamic(self.state_size, self.d_sum + c_action_size)
            self.optimizer_curiosity = adam(list(self.model_forward_dynamic.parameters()))
        elif self.curiosity == 'INVERSE':
            self.model_inverse_dynamic = nn_module.ModelInverseDynamic(self.state_size, self.d_sum + c_action_size)
            self.optimizer_curiosity = adam(list(self.model_inverse_dynamic.parameters()))

        if use_prediction:  # sac_base.py:421-441
            self.model_transition = nn_module.ModelTransition(self.state_size, self.d_sum, c_action_size, use_extra_data)
            self.model_reward = nn_module.ModelReward(self.state_size)
            self.model_observation = nn_module.ModelObservation(self.state_size, obs_shapes, use_extra_data)
            self.optimizer_prediction = adam(self.prediction_parameters())

        self.replay_buffer = PrioritizedReplayRef(batch_size=batch_size, sample_prev_n=burn_in_step,
                                                  sample_post_n=n_step, **(replay_config or {}))
        self.update_target(1.)  # sac_base.py:629

    # -- modules whose weights a golden fixture carries -----------------------------------------
    def named_modules(self) -> dict:
        d = {'model_rep': self.model_rep, 'model_target_rep': self.model_target_rep,
             'model_policy': self.model_policy}
        for i in range(self.E):
            d[f'model_q_{i}'] = self.model_q_list[i]
            d[f'model_target_q_{i}'] = self.model_target_q_list[i]
        if self.curiosity == 'FORWARD':
            d['model_forward_dynamic'] = self.model_forward_dynamic
        elif self.curiosity == 'INVERSE':
            d['model_inverse_dynamic'] = self.model_inverse_dynamic
        if self.use_prediction:
            d.update(model_transition=self.model_transition, model_reward=self.model_reward,
                     model_observation=self.model_observation)
        return d

    def prediction_parameters(self) -> list:
        return [*self.model_transition.parameters(), *self.model_reward.parameters(), *self.model_observation.parameters()]

    def named_optimizers(self) -> dict:
        """named like the reference's ckpt_dict (sac_base.py:493-566)"""
        d = {'optimizer_rep': self.optimizer_rep, 'optimizer_policy': self.optimizer_policy}
        for i in range(self.E):
            d[f'optimizer_q_{i}'] = self.optimizer_q_list[i]
        if self.use_auto_alpha:
            d['optimizer_alpha'] = self.optimizer_alpha
        if self.curiosity is not None:
            d['optimizer_curiosity'] = self.optimizer_curiosity
        if self.use_prediction:
            d['optimizer_prediction'] = self.optimizer_prediction
        return {k: v for k, v in d.items() if v is not None}

    # -- sac_base.py:745-764 ---------------------------------------------------------------------
    @torch.no_grad()
    def update_target(self, tau):
        pairs = list(zip(self.model_target_rep.parameters(), self.model_rep.parameters()))
        for tq, q in zip(self.model_target_q_list, self.model_q_list):
            pairs += list(zip(tq.parameters(), q.parameters()))
        for t, s in pairs:
            t.data.copy_(t.data * (1. - tau) + s.data * tau)

    # -- sac_base.py:2303-2396 -------------------------------------------------------------------
    def put_episode(self, ep_indexes, ep_obses_list, ep_actions, ep_rewards, ep_dones, ep_probs,
                    ep_pre_seq_hidden_states):
        if ep_indexes.shape[1] < self.n:
            return
        last = np.zeros_like(ep_indexes, dtype=bool)
        last[:, -1] = True
        last[ep_indexes == -1] = True
        rows = {'index': ep_indexes[0], 'last_mask': last[0],
                **{f'obs_{k}': o[0] for k, o in zip(self.obs_names, ep_obses_list)},
                'action': ep_actions[0], 'reward': ep_rewards[0], 'done': ep_dones[0],
                'mu_prob': ep_probs[0], 'pre_seq_hidden_state': ep_pre_seq_hidden_states[0]}
        self.replay_buffer.add(rows, ignore_size=1)

    # -- sac_base.py:1117-1157 -------------------------------------------------------------------
    def l_states(self, idx, pad, obs_list, pre_actions, hidden, target=False):
        rep = self.model_target_rep if target else self.model_rep
        if self.seq_encoder == 'ATTN':
            st, hs, _ = rep(idx.shape[1], idx, obs_list, pre_actions, hidden[:, :1],
                            is_prev_hidden_state=True, padding_mask=pad)
            return st, hs
        return rep(obs_list, pre_actions, hidden, padding_mask=pad)

    # -- sac_base.py:1297-1466 -------------------------------------------------------------------
    @torch.no_grad()
    def get_y(self, n_last, n_pad, nx_obs, nx_states, n_actions, n_rewards, n_dones, n_mu_probs):
        d_alpha, c_alpha = torch.exp(self.log_d_alpha), torch.exp(self.log_c_alpha)
        n_states, next_n_states = nx_states[:, :-1], nx_states[:, 1:]
        nx_actions = torch.cat([n_actions, torch.zeros_like(n_actions[:, :1])], dim=1)
        d_policy, c_policy = self.model_policy(nx_states, nx_obs)

        if self.curiosity is not None:  # 1333-1343: writes through into the sampled reward window
            if self.curiosity == 'FORWARD':
                approx = self.model_forward_dynamic(n_states, n_actions)
                bonus = torch.sum(torch.pow(approx - next_n_states, 2), dim=-1) * 0.5
            else:
                approx = self.model_inverse_dynamic(n_states, next_n_states)
                bonus = torch.sum(torch.pow(approx - n_actions, 2), dim=-1) * 0.5
            n_rewards += bonus * self.curiosity_strength

        if self.c_action_size:
            eps = self.noise.standard_normal(tuple(c_policy.loc.shape))
            sampled = c_policy.loc + eps * c_policy.scale          # Normal.rsample
        else:
            sampled = torch.zeros(0)
        nx_qs = [q(nx_states, torch.tanh(sampled), nx_obs) for q in self.model_target_q_list]

        vt = dict(gamma=self.gamma, gamma_ratio=self.gamma_ratio, lambda_ratio=self.lambda_ratio,
                  v_rho=self.v_rho, v_c=self.v_c, use_n_step_is=self.use_n_step_is,
                  n_last_masks=n_last, n_padding_masks=n_pad, n_rewards=n_rewards, n_dones=n_dones)
        subset = lambda: self.noise.permutation(self.E)[:self.E_sample]  # noqa: E731
        d_y = c_y = None

        if self.d_action_sizes:  # 1356-1421 (policy-based branch)
            n_d = torch.stack([q[0][:, :-1] for q in nx_qs])
            nxt_d = torch.stack([q[0][:, 1:] for q in nx_qs])[subset()]
            n_d = n_d[subset()]
            mean_n, mean_next = n_d.mean(0), nxt_d.mean(0)
            probs = d_policy.probs
            n_p, next_p = probs[:, :-1], probs[:, 1:]
            v_n = torch.sum(n_p * (mean_n - d_alpha * torch.log(n_p.clamp(min=1e-8))), -1) / self.d_branches
            v_next = torch.sum(next_p * (mean_next - d_alpha * torch.log(next_p.clamp(min=1e-8))), -1) / self.d_branches
            mu = pi = None
            if self.use_n_step_is:
                mu = n_mu_probs[..., :self.d_sum] * n_actions[..., :self.d_sum]
                mu = torch.where(mu == 0., torch.ones_like(mu), mu).prod(-1)
                pi = torch.exp(d_policy.log_prob(nx_actions[..., :self.d_sum]).sum(-1))[:, :-1]
            d_y = v_trace(**vt, n_mu_probs=mu, n_pi_probs=pi, n_vs=v_n, next_n_vs=v_next)

        if self.c_action_size:  # 1423-1464
            n_c = torch.stack([q[1][:, :-1] for q in nx_qs])
            nxt_c = torch.stack([q[1][:, 1:] for q in nx_qs])
            logp = masked_sum_log_prob(squash_log_prob(c_policy, sampled))   # [B, n+1]
            n_c = n_c[subset()]
            nxt_c = nxt_c[subset()]
            min_n = n_c.min(dim=0)[0].squeeze(-1)
            min_next = nxt_c.min(dim=0)[0].squeeze(-1)
            v_n = min_n - c_alpha * logp[:, :-1]
            v_next = min_next - c_alpha * logp[:, 1:]
            mu = pi = None
            if self.use_n_step_is:
                stored = torch.atanh(torch.clamp(nx_actions[..., self.d_sum:], -0.999, 0.999))
                pi = masked_prod_prob(squash_prob(c_policy, stored)[:, :-1])
                mu = masked_prod_prob(n_mu_probs[..., self.d_sum:])
            c_y = v_trace(**vt, n_mu_probs=mu, n_pi_probs=pi, n_vs=v_n, next_n_vs=v_next)
        return d_y, c_y

    # -- sac_base.py:1468-1605 -------------------------------------------------------------------
    def train_rep_q(self, n_last, n_pad, nx_obs, nx_states, n_actions, n_rewards, n_dones,
                    n_mu_probs, priority_is, nx_target_states=None):
        obs0 = [o[:, 0] for o in nx_obs]
        state, action = nx_states[:, 0], n_actions[:, 0]
        d_action, c_action = action[..., :self.d_sum], action[..., self.d_sum:]
        qs = [q(state, c_action, obs0) for q in self.model_q_list]
        d_y, c_y = self.get_y(n_last, n_pad, nx_obs, nx_states, n_actions, n_rewards, n_dones,
                              n_mu_probs if self.use_n_step_is else None)
        losses = [torch.zeros((state.shape[0], 1)) for _ in range(self.E)]
        mse = nn.MSELoss(reduction='none')
        if self.d_action_sizes:
            for i in range(self.E):
                q_single = torch.sum(d_action * qs[i][0], dim=-1, keepdim=True) / self.d_branches
                losses[i] = losses[i] + mse(q_single, d_y)
        if self.c_action_size:
            for i in range(self.E):
                if self.clip_epsilon > 0:
                    tq = self.model_target_q_list[i](state.detach(), c_action, obs0)[1]
                    clipped = tq + torch.clamp(qs[i][1] - tq, -self.clip_epsilon, self.clip_epsilon)
                    losses[i] = losses[i] + torch.maximum(mse(clipped, c_y), mse(qs[i][1], c_y))
                else:  # 1556: `+=` of (self + mse) doubles the running loss
                    losses[i] = losses[i] + (losses[i] + mse(qs[i][1], c_y))
        if priority_is is not None:
            losses = [l * priority_is for l in losses]
        losses = [torch.mean(l) for l in losses]
        if self.optimizer_rep:
            self.optimizer_rep.zero_grad()
        for o in self.optimizer_q_list:
            o.zero_grad()
        # `use_prediction`: the reference's `_train_rpm` differentiates the representation's graph a second time,
        # which its own `loss.backward()` (1570) has freed — it raises.  The product keeps that graph alive so the head
        # runs; the oracle restates THAT choice here, everything else is 1570-1603 in order.
        torch.stack(losses).sum().backward(retain_graph=self.use_prediction)
        grads_rep_main = [p.grad.detach() for p in self.model_rep.parameters()]
        for o in self.optimizer_q_list:
            o.step()
        self.last_rpm = None
        if self.use_prediction:
            self.last_rpm = self.train_rpm(grads_rep_main, nx_obs, nx_states, nx_target_states, n_actions, n_rewards)
        if self.optimizer_rep:
            self.optimizer_rep.step()
        return losses[0].detach()

    # -- sac_base.py:1607-1631 -------------------------------------------------------------------
    @staticmethod
    @torch.no_grad()
    def adaptive_weights(grads_main, loss_list, params):
        """`calculate_adaptive_weights`: every auxiliary loss's gradient w.r.t. `params` is ADDED to their `.grad`
        iff its cosine with the main gradient (all tensors flattened into one row) is positive — gate =
        clamp(sign(cos), min=0), so cos == 0 (and NaN-free zero gradients) gate to 0.  -> the gates"""
        params = list(params)
        with torch.enable_grad():
            aux_list = [torch.autograd.grad(loss, params, allow_unused=True, retain_graph=True) for loss in loss_list]
        aux_list = [[a if a is not None else torch.zeros_like(m) for m, a in zip(grads_main, aux)] for aux in aux_list]
        flat_main = torch.cat([g.reshape(1, -1) for g in grads_main], dim=1)
        gates = [torch.sign(nn.functional.cosine_similarity(flat_main, torch.cat([a.reshape(1, -1) for a in aux], dim=1))).clamp(min=0)
                 for aux in aux_list]
        for aux, gate in zip(aux_list, gates):
            for p, a in zip(params, aux):
                p.grad += gate * a
        return gates

    # -- sac_base.py:1798-1839 -------------------------------------------------------------------
    def train_rpm(self, grads_rep_main, nx_obs, nx_states, nx_target_states, n_actions, n_rewards):
        """transition model: -mean log N(s'_target | s, a) + transition_kl * mean KL(N || N(0, 1)); reward model: MSE / n;
        observation model: its own `get_loss` / n.  The three losses gate into the REPRESENTATION's gradient
        (`adaptive_weights`); their sum trains the three models (one Adam).  -> dict of observables"""
        n_obs = [o[:, :-1] for o in nx_obs]
        dist = self.model_transition(n_obs, nx_states[:, :-1], n_actions)
        loss_transition = -torch.mean(dist.log_prob(nx_target_states[:, 1:]))
        std_normal = torch.distributions.Normal(torch.zeros_like(dist.loc), torch.ones_like(dist.scale), validate_args=False)
        loss_transition = loss_transition + self.transition_kl * torch.mean(torch.distributions.kl.kl_divergence(dist, std_normal))
        loss_reward = nn.functional.mse_loss(self.model_reward(nx_states[:, 1:]), torch.unsqueeze(n_rewards, 2)) / self.n
        loss_obs = self.model_observation.get_loss(nx_states, list(nx_obs)) / self.n
        gates = None
        if grads_rep_main:      # (a parameter-free representation: the reference raises in autograd.grad; nothing to gate)
            gates = self.adaptive_weights(grads_rep_main, [loss_transition, loss_reward, loss_obs], self.model_rep.parameters())
        self.optimizer_prediction.zero_grad()
        (loss_transition + loss_reward + loss_obs).backward(inputs=self.prediction_parameters())
        self.optimizer_prediction.step()
        return dict(losses=torch.stack([loss_transition, loss_reward, loss_obs]).detach(),
                    gates=None if gates is None else torch.cat(gates), entropy=torch.mean(dist.entropy()).detach())

    # -- sac_base.py:1841-1911 -------------------------------------------------------------------
    def train_policy(self, obs_list, state, action, mu_d_policy_probs):
        B = state.shape[0]
        d_policy, c_policy = self.model_policy(state, obs_list)
        loss_d, loss_c = torch.zeros((B, 1)), torch.zeros((B, 1))
        with torch.no_grad():
            d_alpha, c_alpha = torch.exp(self.log_d_alpha), torch.exp(self.log_c_alpha)
        if self.d_action_sizes:
            probs = d_policy.probs
            c_action = action[..., self.d_sum:]
            d_qs = torch.stack([q(state, c_action, obs_list)[0] for q in self.model_q_list])
            mean_q = d_qs[self.noise.permutation(self.E)[:self.E_sample]].mean(0)
            inner = d_alpha * torch.log(probs.clamp(min=1e-8)) - mean_q.detach()
            loss_d = torch.sum(probs * inner, dim=1, keepdim=True) / self.d_branches
            mu_ent = -torch.sum(mu_d_policy_probs * torch.log(mu_d_policy_probs.clamp(min=1e-8)), -1) / self.d_branches
            pi_ent = d_policy.entropy().sum(-1) / self.d_branches
            loss_d = loss_d + self.d_policy_entropy_penalty * (torch.pow(mu_ent - pi_ent, 2.) / 2.).unsqueeze(-1)
        if self.c_action_size:
            eps = self.noise.standard_normal(tuple(c_policy.loc.shape))
            sampled = c_policy.loc + eps * c_policy.scale
            c_qs = torch.stack([q(state, torch.tanh(sampled), obs_list)[1] for q in self.model_q_list])
            c_qs = c_qs[self.noise.permutation(self.E)[:self.E_sample]]
            logp = masked_sum_log_prob(squash_log_prob(c_policy, sampled), keepdim=True)
            loss_c = c_alpha * logp - c_qs.min(dim=0)[0]
        loss = torch.mean(loss_d + loss_c)
        self.optimizer_policy.zero_grad()
        loss.backward(inputs=list(self.model_policy.parameters()))
        self.optimizer_policy.step()
        d_ent = torch.mean(d_policy.entropy().sum(-1) / self.d_branches).detach() if self.d_action_sizes else None
        c_ent = torch.mean(masked_sum_entropy(c_policy.entropy())).detach() if self.c_action_size else None
        self.last_loss_policy = loss.detach()
        return d_ent, c_ent

    # -- sac_base.py:1913-1949 -------------------------------------------------------------------
    def train_alpha(self, obs_list, state):
        B = state.shape[0]
        with torch.no_grad():
            d_policy, c_policy = self.model_policy(state, obs_list)
        loss_d, loss_c = torch.zeros((B, 1)), torch.zeros((B, 1))
        if self.d_action_sizes:
            probs = d_policy.probs
            inner = self.log_d_alpha * (-torch.log(probs.clamp(min=1e-8)) - self.target_d_alpha)
            loss_d = torch.sum(probs * inner, dim=1, keepdim=True) / self.d_branches
        if self.c_action_size:
            eps = self.noise.standard_normal(tuple(c_policy.loc.shape))
            sampled = eps * c_policy.scale + c_policy.loc            # Normal.sample
            lp = squash_log_prob(c_policy, sampled)
            valid = torch.sum(lp != torch.inf, dim=-1, keepdim=True)
            lp = masked_sum_log_prob(lp, keepdim=True)
            loss_c = self.log_c_alpha * (-lp - self.target_c_alpha * -valid)
        loss = torch.mean(loss_d + loss_c)
        self.optimizer_alpha.zero_grad()
        loss.backward(inputs=[self.log_d_alpha, self.log_c_alpha])
        self.optimizer_alpha.step()

    # -- sac_base.py:1951-1976 -------------------------------------------------------------------
    def train_curiosity(self, n_pad, nx_states, n_actions):
        n_states, next_n_states = nx_states[:, :-1], nx_states[:, 1:]
        self.optimizer_curiosity.zero_grad()
        if self.curiosity == 'FORWARD':
            model, pred, tgt = self.model_forward_dynamic, None, next_n_states
            pred = model(n_states, n_actions)
        else:
            model, tgt = self.model_inverse_dynamic, n_actions
            pred = model(n_states, next_n_states)
        loss = nn.functional.mse_loss(pred, tgt, reduction='none') * ~n_pad.unsqueeze(-1)
        loss = torch.mean(loss)
        loss.backward(inputs=list(model.parameters()))
        self.optimizer_curiosity.step()
        return loss.detach()

    # -- sac_base.py:1159-1189 -------------------------------------------------------------------
    @torch.no_grad()
    def l_probs(self, l_obs, l_states, l_actions):
        d_policy, c_policy = self.model_policy(l_states, l_obs)
        probs = torch.ones((*l_states.shape[:2], self.d_sum + self.c_action_size))
        if self.d_action_sizes:
            probs[..., :self.d_sum] = d_policy.probs
        if self.c_action_size:
            stored = torch.atanh(torch.clamp(l_actions[..., self.d_sum:], -0.999, 0.999))
            probs[..., self.d_sum:] = squash_prob(c_policy, stored)
        return probs

    # -- sac_base.py:2182-2245 -------------------------------------------------------------------
    @torch.no_grad()
    def td_error(self, n_last, n_pad, nx_obs, state, nx_target_states, n_actions, n_rewards,
                 n_dones, n_mu_probs):
        obs0 = [o[:, 0] for o in nx_obs]
        action = n_actions[:, 0]
        d_action, c_action = action[..., :self.d_sum], action[..., self.d_sum:]
        qs = [q(state, c_action, obs0) for q in self.model_q_list]
        d_y, c_y = self.get_y(n_last, n_pad, nx_obs, nx_target_states, n_actions, n_rewards,
                              n_dones, n_mu_probs)
        errs = [torch.zeros((state.shape[0], 1)) for _ in range(self.E)]
        for i in range(self.E):
            if self.d_action_sizes:
                errs[i] += torch.abs(torch.sum(d_action * qs[i][0], -1, keepdim=True) / self.d_branches - d_y)
            if self.c_action_size:
                errs[i] += torch.abs(qs[i][1] - c_y)
        return torch.mean(torch.cat(errs, dim=-1), dim=-1, keepdim=True)

    # -- sac_base.py:2398-2609 -------------------------------------------------------------------
    def train(self):
        """One learner step; returns a dict of observables (ids, losses, td_error, ...) or None when
        the buffer holds <= batch_size rows."""
        rb, b, n = self.replay_buffer, self.b, self.n
        if not rb.is_lg_batch_size:
            return None
        if self.lookahead:
            # one batch in flight: the reference's `Queue(maxsize=1)` + prefetch thread (replay_buffer.py:275, 339-396)
            # hold batch k + 1 — drawn and gathered BEFORE step k's priority update and row write-backs — while the
            # learner trains on batch k; before the first step two batches are drawn from the initial tree
            if self._queued is None:
                self._queued = rb.sample(self.noise.uniforms(self.batch_size))
            drawn = rb.sample(self.noise.uniforms(self.batch_size))
            (ids, windows, is_w), self._queued = self._queued, drawn
        else:
            ids, windows, is_w = rb.sample(self.noise.uniforms(self.batch_size))
        batch = {k: torch.as_tensor(v) for k, v in windows.items()}
        priority_is = torch.as_tensor(is_w)
        pad_window(batch, b, self.padding_action)

        bnx_obs = [batch[f'obs_{k}'] for k in self.obs_names]
        for i, o in enumerate(bnx_obs):   # 783-788
            if o.dtype == torch.uint8:
                bnx_obs[i] = o.type(torch.float32) / 255.
            elif o.dtype == torch.bool:
                bnx_obs[i] = o.type(torch.float32)
        bn_idx, bn_last, bn_pad = batch['index'][:, :-1], batch['last_mask'][:, :-1], batch['padding_mask'][:, :-1]
        bn_act, bn_rew, bn_done = batch['action'][:, :-1], batch['reward'][:, :-1], batch['done'][:, :-1]
        bn_mu = batch['mu_prob'][:, :-1]
        bnx_hidden = batch['pre_seq_hidden_state']

        # _train, sac_base.py:2027-2126
        if self.global_step % self.update_target_per_step == 0:
            self.update_target(self.tau)
        bnx_idx = torch.cat([bn_idx, bn_idx[:, -1:] + (bn_idx[:, -1:] != -1)], dim=1)
        bnx_pad = torch.cat([bn_pad, bn_pad[:, -1:]], dim=1)
        bnx_pre_act = pre_actions_keep_last(bn_act)
        rep_in = (bnx_idx, bnx_pad, bnx_obs, bnx_pre_act, bnx_hidden)
        bnx_states, _ = self.l_states(*rep_in)
        bnx_target_states, _ = self.l_states(*rep_in, target=True)

        w = priority_is if self.use_priority else None
        loss_q = self.train_rep_q(bn_last[:, b:], bn_pad[:, b:], [o[:, b:] for o in bnx_obs],
                                  bnx_states[:, b:], bn_act[:, b:], bn_rew[:, b:], bn_done[:, b:],
                                  bn_mu[:, b:], w, nx_target_states=bnx_target_states[:, b:])
        with torch.no_grad():
            bnx_states, next_hidden = self.l_states(*rep_in)
        obs_b = [o[:, b] for o in bnx_obs]
        state_b = bnx_states[:, b]
        d_ent, c_ent = self.train_policy(obs_b, state_b, bn_act[:, b], bn_mu[:, b, :self.d_sum])
        if self.use_auto_alpha:
            self.train_alpha(obs_b, state_b)
        loss_cur = None
        if self.curiosity is not None:
            loss_cur = self.train_curiosity(bn_pad[:, b:], bnx_states[:, b:], bn_act[:, b:])

        # write-backs, sac_base.py:2558-2605
        out = dict(ids=ids, is_weights=is_w, loss_q=loss_q, d_entropy=d_ent, c_entropy=c_ent,
                   loss_policy=self.last_loss_policy, loss_curiosity=loss_cur, rpm=self.last_rpm, padding_mask=batch['padding_mask'].numpy().copy(),
                   index=batch['index'].numpy().copy())
        bn_states = bnx_states[:, :-1]
        pi_probs = None
        if self.use_n_step_is:
            pi_probs = self.l_probs([o[:, :-1] for o in bnx_obs], bn_states, bn_act)
        if self.use_priority:
            td = self.td_error(bn_last[:, b:], bn_pad[:, b:], [o[:, b:] for o in bnx_obs],
                               bn_states[:, b], bnx_target_states[:, b:], bn_act[:, b:],
                               bn_rew[:, b:], bn_done[:, b:],
                               pi_probs[:, b:] if self.use_n_step_is else None).numpy()
            rb.update(ids, td)
            out['td_error'] = td
        keep = ~bn_pad.numpy().reshape(-1)
        if len(self.seq_hidden_state_shape) and self.seq_hidden_state_shape[-1] != 0:
            tgt = np.stack([ids + 1 + i for i in range(-b, n)], axis=1).reshape(-1)
            h = next_hidden[:, :-1].numpy()
            h = h.reshape(-1, *h.shape[2:])
            rb.update_transitions(tgt[keep], 'pre_seq_hidden_state', h[keep])
        if self.use_n_step_is:
            tgt = np.stack([ids + i for i in range(-b, n)], axis=1).reshape(-1)
            pp = pi_probs.numpy()
            pp = pp.reshape(-1, *pp.shape[2:])
            rb.update_transitions(tgt[keep], 'mu_prob', pp[keep])
            out['pi_probs'] = pi_probs.numpy()
        self.global_step += 1
        out['step'] = self.global_step
        return out
