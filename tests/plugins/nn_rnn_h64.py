"""Model plugin with a one-layer GRU(64) representation over [obs ‖ previous action] — the recurrent core of the reference's
`envs/square/memory_corridor/nn.py:15-30` (`m.GRU(16 + c_action_size, 64, 1)`), on the vector observation of the TEST
configurations.  Written against the plugin API only."""
import torch

import algorithm.nn_models as m


class ModelRep(m.ModelBaseRep):
    def _build_model(self):
        in_size = self.obs_shapes[0][0] + sum(self.d_action_sizes) + self.c_action_size
        self.rnn = m.GRU(in_size, 64, 1)

    def forward(self, obs_list, pre_action, pre_seq_hidden_state, padding_mask=None):
        h0 = None if pre_seq_hidden_state is None else pre_seq_hidden_state[:, 0]
        return self.rnn(torch.cat([obs_list[0], pre_action], dim=-1), h0)


ModelQ = m.ModelQ
ModelPolicy = m.ModelPolicy
