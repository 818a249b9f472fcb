"""Policy plugin base, the stock policy network and its action distributions.

Follows reference `algorithm/nn_models/policy.py:10-174`:
`ModelPolicy(state_size, d_action_sizes, c_action_size, model_abs_dir, **nn_config['policy'])`,
`forward(state, obs_list) -> (JointOneHotCategorical | None, Normal | None)`; the continuous head
is `Normal(5*tanh(mean/5), exp(clamp(logstd, -20, 0.5)))`.
"""
import torch
from torch import nn
from torch.distributions.utils import _standard_normal

from .layers import LinearLayers

__all__ = ['NormalWithPadding', 'JointOneHotCategorical', 'ModelBasePolicy', 'ModelPolicy',
           'ModelTermination']


class NormalWithPadding(torch.distributions.Normal):
    """Normal whose padded action components sample 0 and report +inf log-prob / entropy
    (consumers mask +inf, see `utils.operators.sum_log_prob`)."""

    def __init__(self, loc, scale, padding_mask, validate_args=None):
        super().__init__(loc, scale, validate_args)
        self.padding_mask = padding_mask

    def sample(self, sample_shape=torch.Size()):
        v = super().sample(sample_shape)
        v[..., self.padding_mask] = 0.
        return v

    def rsample(self, sample_shape=torch.Size()):
        eps = _standard_normal(self._extended_shape(sample_shape),
                               dtype=self.loc.dtype, device=self.loc.device)
        keep = ~self.padding_mask
        return self.loc * keep + eps * (self.scale * keep)

    def log_prob(self, value):
        lp = super().log_prob(value)
        lp[self.padding_mask] = torch.inf
        return lp

    def entropy(self):
        ent = super().entropy()
        ent[self.padding_mask] = torch.inf
        return ent


class JointOneHotCategorical(torch.distributions.Distribution):
    """Independent one-hot categoricals, one per discrete action branch, concatenated."""

    def __init__(self, dists):
        self._dists = dists
        self.logits_size_list = [d.logits.shape[-1] for d in dists]

    @property
    def dists(self):
        return self._dists

    @property
    def probs(self):
        return torch.cat([d.probs for d in self._dists], dim=-1)

    @property
    def logits(self):
        return torch.cat([d.logits for d in self._dists], dim=-1)

    def sample(self, sample_shape=torch.Size()):
        return torch.cat([d.sample(sample_shape) for d in self._dists], dim=-1)

    def sample_deter(self):
        parts = self.logits.split(self.logits_size_list, dim=-1)
        return torch.cat([nn.functional.one_hot(p.argmax(dim=-1), n)
                          for p, n in zip(parts, self.logits_size_list)], dim=-1)

    def log_prob(self, value):
        parts = value.split(self.logits_size_list, dim=-1)
        return torch.stack([d.log_prob(v) for d, v in zip(self._dists, parts)], dim=-1)

    def entropy(self):
        return torch.stack([d.entropy() for d in self._dists], dim=-1)


class ModelBasePolicy(nn.Module):
    def __init__(self, state_size, d_action_sizes, c_action_size, model_abs_dir=None, **kwargs):
        super().__init__()
        self.state_size = state_size
        self.d_action_sizes = d_action_sizes
        self.c_action_size = c_action_size
        self.model_abs_dir = model_abs_dir
        self._build_model(**kwargs)

    def _build_model(self, **kwargs):
        pass

    def forward(self, state, obs_list):
        raise NotImplementedError('ModelPolicy not implemented')

    def __call__(self, state, obs_list):
        return nn.Module.__call__(self, state, obs_list)


class ModelPolicy(ModelBasePolicy):
    def _build_model(self, dense_n=64, dense_depth=0,
                     d_dense_n=64, d_dense_depth=3,
                     c_dense_n=64, c_dense_depth=3,
                     mean_n=64, mean_depth=0,
                     logstd_n=64, logstd_depth=0,
                     dropout=0.):
        self.dense = LinearLayers(self.state_size, dense_n, dense_depth, dropout=dropout)
        trunk = self.dense.output_size

        if self.d_action_sizes:
            self.d_dense_list = nn.ModuleList([
                LinearLayers(trunk, d_dense_n, d_dense_depth, size, dropout=dropout)
                for size in self.d_action_sizes])

        if self.c_action_size:
            self.c_dense = LinearLayers(trunk, c_dense_n, c_dense_depth, dropout=dropout)
            width = self.c_dense.output_size
            self.mean_dense = LinearLayers(width, mean_n, mean_depth, self.c_action_size, dropout=dropout)
            self.logstd_dense = LinearLayers(width, logstd_n, logstd_depth, self.c_action_size, dropout=dropout)

    def c_head_raw(self, state):
        """(mean, logstd) before the bounding non-linearities; the fused sampling kernel
        (`policy_sample_logp`) applies them itself."""
        h = self.c_dense(self.dense(state))
        return self.mean_dense(h), self.logstd_dense(h)

    def forward(self, state, obs_list):
        h = self.dense(state)
        d_policy = c_policy = None
        if self.d_action_sizes:
            d_policy = JointOneHotCategorical([
                torch.distributions.OneHotCategorical(logits=head(h), validate_args=False)
                for head in self.d_dense_list])
        if self.c_action_size:
            z = self.c_dense(h)
            mean, logstd = self.mean_dense(z), self.logstd_dense(z)
            c_policy = torch.distributions.Normal(torch.tanh(mean / 5.) * 5.,
                                                  torch.exp(torch.clamp(logstd, -20, 0.5)),
                                                  validate_args=False)
        return d_policy, c_policy


class ModelTermination(nn.Module):
    """Option-termination head (option-critic variant; API surface only)."""

    def __init__(self, state_size):
        super().__init__()
        self.state_size = state_size
        self._build_model()

    def _build_model(self, dense_n=64, dense_depth=2, dropout=0.):
        self.dense = LinearLayers(self.state_size, dense_n, dense_depth, output_size=1, dropout=dropout)

    def forward(self, state, obs_list):
        return torch.sigmoid(torch.clamp(self.dense(state), -3., 3.))
