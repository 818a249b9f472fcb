"""Model plugin with an episodic 2-layer attention representation over [obs ‖ previous action] whose state is BOUNDED: the
attention output goes through a Linear + tanh head of size 8 (the head of the reference's `tests/nn_conv_attn.py:9-96`, on
the vector observation of `envs/test/nn_attn.py:9-35`).  With |state| <= 1 the stock policy stays away from its log-std clamp
and the reference's own f32 gradients are well-conditioned: the step golden of this plugin compares EVERY gradient against
the reference at the tolerances of the other cases.  Written against the plugin API only (loads under the reference too)."""
import torch
from torch import nn

import algorithm.nn_models as m


class ModelRep(m.ModelBaseAttentionRep):
    def _build_model(self):
        embed_dim = self.obs_shapes[0][0] + self.c_action_size + sum(self.d_action_sizes)
        self.attn = m.EpisodeMultiheadAttention(embed_dim)
        self.dense = nn.Sequential(nn.Linear(embed_dim, 8), nn.Tanh())

    def forward(self, seq_q_len, index, obs_list, pre_action, pre_seq_hidden_state, is_prev_hidden_state=False,
                query_only_attend_to_rest_key=False, padding_mask=None):
        x = torch.cat([obs_list[0], pre_action], dim=-1)
        state, hn, attn_weights_list = self.attn(x, seq_q_len=seq_q_len, hidden_state=pre_seq_hidden_state,
                                                 is_prev_hidden_state=is_prev_hidden_state,
                                                 query_only_attend_to_rest_key=query_only_attend_to_rest_key,
                                                 key_index=index, key_padding_mask=padding_mask)
        return self.dense(state), hn, attn_weights_list


ModelQ = m.ModelQ
ModelPolicy = m.ModelPolicy
