"""Q-function plugin base and the stock dueling-free Q network.

Interface and sub-module names follow reference `algorithm/nn_models/q.py:9-91`:
`ModelQ(state_size, d_action_sizes, c_action_size, is_target, model_abs_dir)`,
`forward(state, c_action, obs_list) -> (d_qs | None, c_q | None)`.
"""
import torch
from torch import nn

from .layers import LinearLayers

__all__ = ['ModelBaseQ', 'ModelQ']


class ModelBaseQ(nn.Module):
    def __init__(self, state_size, d_action_sizes, c_action_size, is_target, model_abs_dir=None):
        super().__init__()
        self.state_size = state_size
        self.d_action_sizes = d_action_sizes
        self.c_action_size = c_action_size
        self.is_target = is_target
        self.model_abs_dir = model_abs_dir
        self._build_model()

    def _build_model(self):
        pass

    def forward(self, state, action, obs_list):
        raise NotImplementedError('ModelQ not implemented')

    def __call__(self, state, action, obs_list):
        return nn.Module.__call__(self, state, action, obs_list)


class ModelQ(ModelBaseQ):
    """state -> dense -> {per-branch discrete heads, [c_state ‖ c_action] -> c_dense -> 1}."""

    def _build_model(self, dense_n=64, dense_depth=0,
                     d_dense_n=64, d_dense_depth=3,
                     c_state_n=64, c_state_depth=0,
                     c_action_n=64, c_action_depth=0,
                     c_dense_n=64, c_dense_depth=3,
                     dropout=0.):
        self.dense = LinearLayers(self.state_size, dense_n, dense_depth, dropout=dropout)
        trunk = self.dense.output_size

        if self.d_action_sizes:
            self.d_dense_list = nn.ModuleList([
                LinearLayers(trunk, d_dense_n, d_dense_depth, size, dropout=dropout)
                for size in self.d_action_sizes])

        if self.c_action_size:
            self.c_state_dense = LinearLayers(trunk, c_state_n, c_state_depth, dropout=dropout)
            self.c_action_dense = LinearLayers(self.c_action_size, c_action_n, c_action_depth, dropout=dropout)
            self.c_dense = LinearLayers(self.c_state_dense.output_size + self.c_action_dense.output_size,
                                        c_dense_n, c_dense_depth, 1, dropout=dropout)

    def forward(self, state, c_action, obs_list):
        h = self.dense(state)
        d_qs = c_q = None
        if self.d_action_sizes:
            d_qs = torch.cat([head(h) for head in self.d_dense_list], dim=-1)
        if self.c_action_size:
            joint = torch.cat([self.c_state_dense(h), self.c_action_dense(c_action)], dim=-1)
            c_q = self.c_dense(joint)
        return d_qs, c_q
