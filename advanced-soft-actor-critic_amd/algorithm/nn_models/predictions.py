"""Recurrent-prediction-model heads of the plugin surface: transition, reward and observation models
(reference `algorithm/nn_models/predictions.py:7-105`)."""
import torch
from torch import nn

from .layers import LinearLayers

__all__ = ['ModelBaseTransition', 'ModelTransition', 'ModelBaseReward', 'ModelReward', 'ModelBaseObservation']


class ModelBaseTransition(nn.Module):
    def __init__(self, state_size, d_action_size, c_action_size, use_extra_data):
        super().__init__()
        self.state_size = state_size
        self.d_action_size, self.c_action_size = d_action_size, c_action_size
        self.use_extra_data = use_extra_data
        self.action_size = d_action_size + c_action_size
        self._build_model()

    def _build_model(self):
        pass

    def forward(self, obs_list, state, action):
        """(s_t [, extra_obs_t], a_t) -> Normal over s_t+1"""
        raise NotImplementedError('ModelBaseTransition not implemented')

    def extra_obs(self, obs_list):
        raise NotImplementedError('ModelBaseTransition.extra_obs not implemented')


class ModelTransition(ModelBaseTransition):
    def _build_model(self, dense_n=64, dense_depth=0, extra_size=0):
        n_in = self.state_size + self.action_size
        if self.use_extra_data:
            if extra_size == 0:
                raise Exception('use_extra_data is True but extra_size is zero')
            n_in += extra_size
        self.dense = LinearLayers(n_in, dense_n, dense_depth, self.state_size * 2)
        self.dense.fuse = True     # one launch per pass when the stack fits (fused_mlp.describe_dense)

    SCALE_MIN, SCALE_MAX = 0.1, 1.0

    def mean_logstd(self, obs_list, state, action):
        """the dense stack's raw output [..., 2 * state_size] = (mean | logstd); the learner's fused transition loss
        (`asac_normal_nll_kl_logstd`) applies exp / clamp itself"""
        parts = [state, self.extra_obs(obs_list), action] if self.use_extra_data else [state, action]
        return self.dense(torch.cat(parts, dim=-1))       # (one concatenation: same columns as cat(cat(s, e), a))

    def forward(self, obs_list, state, action):
        mean, logstd = torch.chunk(self.mean_logstd(obs_list, state, action), 2, dim=-1)
        return torch.distributions.Normal(mean, torch.clamp(torch.exp(logstd), self.SCALE_MIN, self.SCALE_MAX),
                                          validate_args=False)


class ModelBaseReward(nn.Module):
    def __init__(self, state_size):
        super().__init__()
        self.state_size = state_size
        self._build_model()

    def _build_model(self):
        pass

    def forward(self, state):
        raise NotImplementedError('ModelBaseReward not implemented')


class ModelReward(ModelBaseReward):
    def _build_model(self, dense_n=64, dense_depth=0):
        self.dense = LinearLayers(self.state_size, dense_n, dense_depth, 1)
        self.dense.fuse = True

    def forward(self, state):
        return self.dense(state)


class ModelBaseObservation(nn.Module):
    def __init__(self, state_size, obs_shapes, use_extra_data):
        super().__init__()
        self.state_size, self.obs_shapes, self.use_extra_data = state_size, obs_shapes, use_extra_data
        self._build_model()

    def _build_model(self):
        pass

    def forward(self, state):
        """s_t -> approx o_t (a tensor or a list of tensors)"""
        raise NotImplementedError('ModelBaseObservation not implemented')

    def get_loss(self, state, obs_list):
        raise NotImplementedError('ModelBaseObservation.get_loss not implemented')
