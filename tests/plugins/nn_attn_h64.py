"""Model plugin with the episodic attention core of the reference's environments — `EpisodeMultiheadAttention(64, num_layers 2,
num_heads 8)` over a dense embedding of [obs ‖ previous action] and a Linear + tanh state of size 8 (the composition of
`envs/gym/toy_queue/nn_attn.py:28-45` / `envs/square/obstacle/nn_attn.py:16`) — on the vector observation of the TEST
configurations.  Written against the plugin API only."""
import torch
from torch import nn

import algorithm.nn_models as m

EMBED = 64


class ModelRep(m.ModelBaseAttentionRep):
    def _build_model(self):
        in_size = self.obs_shapes[0][0] + sum(self.d_action_sizes) + self.c_action_size
        self.embed = m.LinearLayers(in_size, dense_n=EMBED, dense_depth=1)
        self.attn = m.EpisodeMultiheadAttention(EMBED, num_layers=2, num_heads=8)
        self.dense = nn.Sequential(nn.Linear(EMBED, 8), nn.Tanh())

    def forward(self, seq_q_len, index, obs_list, pre_action, pre_seq_hidden_state, is_prev_hidden_state=False,
                query_only_attend_to_rest_key=False, padding_mask=None):
        x = self.embed(torch.cat([obs_list[0], pre_action], dim=-1))
        state, hn, attn_weights_list = self.attn(x, seq_q_len=seq_q_len, hidden_state=pre_seq_hidden_state,
                                                 is_prev_hidden_state=is_prev_hidden_state,
                                                 query_only_attend_to_rest_key=query_only_attend_to_rest_key,
                                                 key_index=index, key_padding_mask=padding_mask)
        return self.dense(state), hn, attn_weights_list


ModelQ = m.ModelQ
ModelPolicy = m.ModelPolicy
