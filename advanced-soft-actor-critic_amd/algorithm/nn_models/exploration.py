"""Exploration heads of the model-plugin surface: random network distillation and forward / inverse
dynamics (reference `algorithm/nn_models/exploration.py:7-138`; same names, ctor args, sub-modules)."""
import torch
from torch import nn

from .layers import LinearLayers

__all__ = ['ModelRND', 'ModelOptionSelectorRND', 'ModelBaseForwardDynamic', 'ModelForwardDynamic',
           'ModelBaseInverseDynamic', 'ModelInverseDynamic']


class ModelRND(nn.Module):
    def __init__(self, state_size, d_action_summed_size, c_action_size):
        super().__init__()
        self.state_size = state_size
        self.d_action_summed_size = d_action_summed_size
        self.c_action_size = c_action_size
        self._build_model()

    def _build_model(self, dense_n=64, dense_depth=2, output_size=None):
        mk = lambda n_in: LinearLayers(n_in, dense_n, dense_depth, output_size)  # noqa: E731
        self.s_dense = mk(self.state_size)
        if self.d_action_summed_size:
            self.d_dense_list = nn.ModuleList([mk(self.state_size) for _ in range(self.d_action_summed_size)])
        if self.c_action_size:
            self.c_dense = mk(self.state_size + self.c_action_size)

    def cal_s_rnd(self, state):
        """-> [*batch, f]"""
        return self.s_dense(state)

    def cal_d_rnd(self, state):
        """-> [*batch, d_action_summed_size, f]"""
        return torch.stack([d(state) for d in self.d_dense_list], dim=-2)

    def cal_c_rnd(self, state, c_action):
        """-> [*batch, f]"""
        return self.c_dense(torch.cat([state, c_action], dim=-1))


class ModelOptionSelectorRND(nn.Module):
    def __init__(self, state_size, num_options):
        super().__init__()
        self.state_size, self.num_options = state_size, num_options
        self._build_model()

    def _build_model(self, dense_n=64, dense_depth=2, output_size=None):
        self.dense_list = nn.ModuleList([LinearLayers(self.state_size, dense_n, dense_depth, output_size)
                                         for _ in range(self.num_options)])

    def cal_rnd(self, state):
        """-> [*batch, num_options, f]"""
        return torch.stack([d(state) for d in self.dense_list], dim=-2)


class ModelBaseForwardDynamic(nn.Module):
    def __init__(self, state_size, action_size):
        super().__init__()
        self.state_size, self.action_size = state_size, action_size
        self._build_model()

    def _build_model(self):
        pass

    def forward(self, state, action):
        raise NotImplementedError('ModelBaseForwardDynamic not implemented')


class ModelForwardDynamic(ModelBaseForwardDynamic):
    """(s_t, a_t) -> approx s_t+1"""

    def _build_model(self, dense_n=64, dense_depth=2):
        self.dense = LinearLayers(self.state_size + self.action_size, dense_n, dense_depth, self.state_size)
        self.dense.fuse = True     # one launch per pass when the stack fits (fused_mlp.describe_dense)

    def forward(self, state, action):
        return self.dense(torch.cat([state, action], dim=-1))


class ModelBaseInverseDynamic(nn.Module):
    def __init__(self, state_size, action_size):
        super().__init__()
        self.state_size, self.action_size = state_size, action_size
        self._build_model()

    def _build_model(self):
        pass

    def forward(self, state_from, state_to):
        raise NotImplementedError('ModelBaseInverseDynamic not implemented')


class ModelInverseDynamic(ModelBaseInverseDynamic):
    """(s_t, s_t+1) -> approx a_t"""

    def _build_model(self, dense_n=64, dense_depth=2):
        self.dense = LinearLayers(self.state_size * 2, dense_n, dense_depth, self.action_size)
        self.dense.fuse = True

    def forward(self, state_from, state_to):
        return self.dense(torch.cat([state_from, state_to], dim=-1))
