"""Model plugin for vector(10) + image(3,30,30) observations with an episodic attention representation:
'simple' conv stack -> 8 features -> EpisodeMultiheadAttention -> Linear+tanh state of size 8 (the
composition of the reference's `tests/nn_conv_attn.py:9-96`), plus the forward-dynamics model the
FORWARD curiosity of BASELINE configs[4] needs and the recurrent prediction models of its `use_prediction=true`
(transition / reward / observation heads composed like the reference's `envs/roller/nn_visual_hard_attn.py:51-96`:
no reference plugin file combines ATTN + prediction + curiosity for these observations).
Written against the plugin API only."""
import torch
from torch import nn

import algorithm.nn_models as m


class ModelRep(m.ModelBaseAttentionRep):
    def _build_model(self):
        self.conv = m.ConvLayers(30, 30, 3, 'simple', out_dense_depth=2, output_size=8)
        embed_dim = self.conv.output_size
        self.attn = m.EpisodeMultiheadAttention(embed_dim)
        self.dense = nn.Sequential(nn.Linear(embed_dim, 8), nn.Tanh())

    def forward(self, seq_q_len, index, obs_list, pre_action, pre_seq_hidden_state, is_prev_hidden_state=False,
                query_only_attend_to_rest_key=False, padding_mask=None):
        _, obs_vis = obs_list
        vis = self.conv(obs_vis)
        state, hn, attn_weights_list = self.attn(vis, seq_q_len=seq_q_len, hidden_state=pre_seq_hidden_state,
                                                 is_prev_hidden_state=is_prev_hidden_state,
                                                 query_only_attend_to_rest_key=query_only_attend_to_rest_key,
                                                 key_index=index, key_padding_mask=padding_mask)
        return self.dense(state), hn, attn_weights_list


EXTRA_SIZE = 10     # the vector observation rides along as "extra data"


class ModelTransition(m.ModelTransition):
    def _build_model(self):
        return super()._build_model(dense_depth=2, extra_size=EXTRA_SIZE)

    def extra_obs(self, obs_list):
        return obs_list[0]


class ModelReward(m.ModelReward):
    def _build_model(self):
        return super()._build_model(dense_depth=2)


class ModelObservation(m.ModelBaseObservation):
    """state -> (3, 30, 30) frame through a transposed-convolution decoder (2x2 -> 6x6 -> 28x28 -> 30x30) and -> the
    vector observation through a dense head"""

    def _build_model(self):
        self.conv_transpose = m.ConvTransposeLayers(
            self.state_size, 64, 1, 2, 2, 32,
            conv_transpose=nn.Sequential(nn.ConvTranspose2d(32, 32, 4, 2), nn.LeakyReLU(),
                                         nn.ConvTranspose2d(32, 16, 8, 4), nn.LeakyReLU(),
                                         nn.ConvTranspose2d(16, 3, 3, 1), nn.LeakyReLU()))
        self.vec_dense = m.LinearLayers(self.state_size, dense_depth=2, output_size=EXTRA_SIZE)

    def forward(self, state):
        return self.conv_transpose(state), self.vec_dense(state)

    def get_loss(self, state, obs_list):
        approx_vis, approx_vec = self(state)
        vec, vis = obs_list
        mse = nn.functional.mse_loss
        return mse(approx_vis, vis) + mse(approx_vec, vec) if self.use_extra_data else mse(approx_vis, vis)


ModelQ = m.ModelQ
ModelPolicy = m.ModelPolicy
ModelForwardDynamic = m.ModelForwardDynamic
