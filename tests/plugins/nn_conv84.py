"""Model plugin for vector(10) + image(3,84,84) (the frame size of the reference's environments: `envs/roller/nn_visual_hard_attn.py`, every `ConvLayers(84, 84, ...)` under `envs/`) observations: 'simple' conv stack -> 8 features,
concatenated with the vector -> Linear+tanh state of size 8 (the composition of the reference's
`tests/nn_conv_vanilla.py:7-56`, which BASELINE configs[3] names)."""
import torch
from torch import nn

import algorithm.nn_models as m


class ModelRep(m.ModelBaseRep):
    def _build_model(self):
        self.conv = m.ConvLayers(84, 84, 3, 'simple', out_dense_depth=2, output_size=8)
        self.dense = nn.Sequential(nn.Linear(self.conv.output_size + self.obs_shapes[0][0], 8), nn.Tanh())

    def forward(self, obs_list, pre_action, pre_seq_hidden_state, padding_mask=None):
        vec, img = obs_list
        state = self.dense(torch.cat([vec, self.conv(img)], dim=-1))
        return state, self._get_empty_seq_hidden_state(state)

    def get_augmented_encoders(self, obs_list):
        return self.conv(obs_list[1])

    def get_state_from_encoders(self, encoders, obs_list, pre_action, pre_seq_hidden_state, padding_mask=None):
        return self.dense(torch.cat([obs_list[0], encoders], dim=-1))


ModelQ = m.ModelQ
ModelPolicy = m.ModelPolicy
ModelForwardDynamic = m.ModelForwardDynamic
ModelRND = m.ModelRND
ModelRepProjection = m.ModelRepProjection
ModelRepPrediction = m.ModelRepPrediction
