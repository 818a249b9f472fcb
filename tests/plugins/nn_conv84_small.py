"""Model plugin for image(3,84,84) + vector(10) observations with a SMALL head behind the 'simple' convolution stack
(2 592 -> 8 -> 8 features), so that a recorded reference step (tests/golden/f6_step_conv84.npz) stays a few MB: the frame
size of the reference's environments (`ConvLayers(84, 84, ...)` under `envs/`) through the product's tiled convolution
kernels against the reference's own Conv2d.  Written against the plugin API only: it loads under the reference."""
import torch
from torch import nn

import algorithm.nn_models as m


class ModelRep(m.ModelBaseRep):
    def _build_model(self):
        self.conv = m.ConvLayers(84, 84, 3, 'simple', out_dense_n=8, out_dense_depth=1, output_size=8)
        self.dense = nn.Sequential(nn.Linear(self.conv.output_size + self.obs_shapes[0][0], 8), nn.Tanh())

    def forward(self, obs_list, pre_action, pre_seq_hidden_state, padding_mask=None):
        vec, img = obs_list
        state = self.dense(torch.cat([vec, self.conv(img)], dim=-1))
        return state, self._get_empty_seq_hidden_state(state)


ModelQ = m.ModelQ
ModelPolicy = m.ModelPolicy
