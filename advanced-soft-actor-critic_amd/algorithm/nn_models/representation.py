"""Representation-model plugin bases.

Constructor argument order and `forward` contracts follow reference
`algorithm/nn_models/representation.py:9-139` (the BYOL projection / prediction heads at the end: 254-307);
`SAC_Base._build_model` instantiates
`nn.ModelRep(obs_names, obs_shapes, d_action_sizes, c_action_size, is_target, model_abs_dir,
**nn_config['rep'])`.
"""
import torch
from torch import nn

from .layers.linear_layers import LinearLayers

__all__ = ['ModelBaseRep', 'ModelSimpleRep', 'ModelBaseAttentionRep', 'ModelBaseOptionSelectorRep',
           'ModelBaseOptionSelectorAttentionRep', 'ModelVOverOptions',
           'ModelBaseRepProjection', 'ModelRepProjection', 'ModelBaseRepPrediction', 'ModelRepPrediction']


class ModelBaseRep(nn.Module):
    def __init__(self, obs_names, obs_shapes, d_action_sizes, c_action_size, is_target,
                 model_abs_dir=None, **kwargs):
        super().__init__()
        self.obs_names = obs_names
        self.obs_shapes = obs_shapes
        self.d_action_sizes = d_action_sizes
        self.c_action_size = c_action_size
        self.is_target = is_target
        self.model_abs_dir = model_abs_dir
        self._build_model(**kwargs)

    def _build_model(self, **kwargs):
        pass

    def forward(self, obs_list, pre_action, pre_seq_hidden_state, padding_mask=None):
        """obs_list: list([batch, l, *obs_shape_i]); pre_action [batch, l, A];
        pre_seq_hidden_state [batch, l, *hidden] -> (state [batch, l, S], seq_hidden_state)"""
        raise NotImplementedError('ModelRep not implemented')

    def __call__(self, obs_list, pre_action, pre_seq_hidden_state, padding_mask=None):
        return nn.Module.__call__(self, obs_list, pre_action, pre_seq_hidden_state, padding_mask)

    def _get_empty_seq_hidden_state(self, state):
        return state.new_zeros((*state.shape[:-1], 0))

    def get_augmented_encoders(self, obs_list):
        raise NotImplementedError('get_augmented_encoders not implemented')

    def get_state_from_encoders(self, encoders, obs_list, pre_action, pre_seq_hidden_state,
                                padding_mask=None):
        raise NotImplementedError('get_state_from_encoders not implemented')


class ModelSimpleRep(ModelBaseRep):
    """State = concatenation of every rank-1 observation; no sequence state."""

    def forward(self, obs_list, pre_action, pre_seq_hidden_state, padding_mask=None):
        vec = [o for o, shape in zip(obs_list, self.obs_shapes) if len(shape) == 1]
        state = vec[0] if len(vec) == 1 else torch.cat(vec, dim=-1)
        return state, self._get_empty_seq_hidden_state(state)


class ModelBaseAttentionRep(ModelBaseRep):
    def forward(self, seq_q_len, index, obs_list, pre_action, pre_seq_hidden_state,
                is_prev_hidden_state=False, query_only_attend_to_rest_key=False, padding_mask=None):
        raise NotImplementedError('ModelAttentionRep not implemented')

    def __call__(self, seq_q_len, index, obs_list, pre_action, pre_seq_hidden_state,
                 is_prev_hidden_state=False, query_only_attend_to_rest_key=False, padding_mask=None):
        return nn.Module.__call__(self, seq_q_len, index, obs_list, pre_action, pre_seq_hidden_state,
                                  is_prev_hidden_state, query_only_attend_to_rest_key, padding_mask)

    def get_state_from_encoders(self, encoders, seq_q_len, index, obs_list, pre_action,
                                pre_seq_hidden_state, is_prev_hidden_state=False,
                                query_only_attend_to_rest_key=False, padding_mask=None):
        raise NotImplementedError('get_state_from_encoders not implemented')


# ---- option-critic plugin bases ---------------------------------------------------------------------
# The option-critic learner (reference algorithm/oc/*) is outside the MI355X hot path, but user plugin
# files define their option-selector models next to the SAC ones, so the bases they subclass are part of
# the surface (reference representation.py:145-251).

class ModelBaseOptionSelectorRep(ModelBaseRep):
    def __init__(self, obs_names, obs_shapes, d_action_sizes, c_action_size, is_target, use_dilation,
                 model_abs_dir=None, **kwargs):
        self.use_dilation = use_dilation    # read by `_build_model`, which the base constructor calls
        super().__init__(obs_names, obs_shapes, d_action_sizes, c_action_size, is_target, model_abs_dir, **kwargs)

    def forward(self, obs_list, pre_action, pre_seq_hidden_state, pre_termination_mask=None, padding_mask=None):
        """as `ModelBaseRep.forward`, plus pre_termination_mask bool[batch]"""
        raise NotImplementedError('ModelOptionSelectorRep not implemented')

    def __call__(self, obs_list, pre_action, pre_seq_hidden_state, pre_termination_mask=None, padding_mask=None):
        return nn.Module.__call__(self, obs_list, pre_action, pre_seq_hidden_state, pre_termination_mask, padding_mask)


class ModelBaseOptionSelectorAttentionRep(ModelBaseOptionSelectorRep, ModelBaseAttentionRep):
    def forward(self, seq_q_len, index, obs_list, pre_action, pre_seq_hidden_state, pre_termination_mask=None,
                is_prev_hidden_state=False, query_only_attend_to_rest_key=False, padding_mask=None):
        raise NotImplementedError('ModelOptionSelectorAttentionRep not implemented')

    def __call__(self, seq_q_len, index, obs_list, pre_action, pre_seq_hidden_state, pre_termination_mask=None,
                 is_prev_hidden_state=False, query_only_attend_to_rest_key=False, padding_mask=None):
        return nn.Module.__call__(self, seq_q_len, index, obs_list, pre_action, pre_seq_hidden_state,
                                  pre_termination_mask, is_prev_hidden_state, query_only_attend_to_rest_key,
                                  padding_mask)


class ModelVOverOptions(nn.Module):
    """state -> one value per option"""

    def __init__(self, state_size, num_options, is_target):
        super().__init__()
        self.state_size, self.num_options, self.is_target = state_size, num_options, is_target
        self._build_model()

    def _build_model(self, dense_n=64, dense_depth=2):
        self.dense = LinearLayers(self.state_size, dense_n, dense_depth, self.num_options)

    def forward(self, state):
        return self.dense(state)


class ModelBaseRepProjection(nn.Module):
    def __init__(self, encoder_size):
        super().__init__()
        self.encoder_size = encoder_size
        self._build_model()

    def _build_model(self):
        pass

    def forward(self, encoder):
        raise NotImplementedError('ModelBaseRepProjection not implemented')


class ModelRepProjection(ModelBaseRepProjection):
    def _build_model(self, dense_n=None, dense_depth=1, projection_size=None):
        dense_n = self.encoder_size if dense_n is None else dense_n
        projection_size = dense_n - 2 if projection_size is None else projection_size
        self.dense = LinearLayers(self.encoder_size, dense_n, dense_depth, projection_size)

    def forward(self, encoder):
        return self.dense(encoder)


class ModelBaseRepPrediction(nn.Module):
    def __init__(self, encoder_size):
        super().__init__()
        self.encoder_size = encoder_size
        self._build_model()

    def _build_model(self):
        pass

    def forward(self, encoder):
        raise NotImplementedError('ModelBaseRepPrediction not implemented')


class ModelRepPrediction(ModelBaseRepPrediction):
    def _build_model(self, dense_n=None, dense_depth=1):
        dense_n = self.encoder_size if dense_n is None else dense_n
        self.dense = LinearLayers(self.encoder_size, dense_n, dense_depth, self.encoder_size)

    def forward(self, encoder):
        return self.dense(encoder)
