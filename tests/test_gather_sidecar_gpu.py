"""GPU: the window gather as a sidecar job (ASAC_SIDECAR_WINDOW_GATHER, hosted by `asac_policy_sample_q_forward`) — the batch
the lookahead schedule draws one step ahead — writes exactly what the stand-alone `asac_window_gather_pad` launch writes
(every key: widened bytes and flags, padded rows, the joint layout beside the previous actions, the derived keys), and the
hosting launch's own outputs are untouched by it."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.test_fused_mlp_gpu import _setup  # noqa: E402


@pytest.mark.parametrize('batch,joint', [(32, True), (256, False), (1000, True)])
def test_next_window_gather_rides_in_the_policy_critics_launch(batch, joint):
    import asac_amd  # noqa: F401
    from asac_amd import native
    from algorithm.replay_buffer import PrioritizedReplayBuffer
    gen = np.random.default_rng(0)
    n = 3000
    data = dict(index=(np.arange(n) % 37).astype(np.int32), obs_vec=gen.standard_normal((n, 6)).astype(np.float32),
                obs_img=gen.integers(0, 255, (n, 4, 4, 3)).astype(np.uint8),
                obs_flag=gen.integers(0, 2, (n, 5)).astype(bool),
                action=gen.standard_normal((n, 3)).astype(np.float32), reward=gen.standard_normal(n).astype(np.float32),
                done=np.zeros(n, bool), last_mask=np.zeros(n, bool), mu_prob=np.ones((n, 3), np.float32),
                pre_seq_hidden_state=gen.standard_normal((n, 2)).astype(np.float32))
    rb = PrioritizedReplayBuffer(batch_size=batch, capacity=4096, sample_prev_n=3, sample_post_n=4, device='cuda')
    rb.set_window_padding(torch.zeros(3))
    rb.add(data)
    rb.sample()
    if joint:
        rb.join_vector_obs_with_pre_action(3)
        rb._build_batch()
    rb.enable_lookahead()
    # the batch in flight: ids drawn into the other set, gathered by the stand-alone launch -> the expected bytes
    rb.swap_sets()
    rb.sample_into_static()
    rb.swap_sets()
    alt = rb._alt['_batch']
    want = {k: v.clone() for k, v in alt.items()}
    want_pre = None if rb._alt['joint_pre_action'] is None else rb._alt['joint_pre_action'].clone()
    assert bool(want['padding_mask'].any())
    for v in alt.values():
        v.zero_() if v.dtype != torch.bool else v.fill_(False)
    # the host launch: policy -> sample -> critics over unrelated rows, with the gather as its sidecar
    B, T, S, A, E = 256, 5, 6, 2, 2
    _, _, fq = _setup(E, S, A)
    _, _, fpi = _setup(1, S, A, policy=True)
    xs = torch.randn(B * T, S, device='cuda')
    eps = torch.randn(B * T, A, device='cuda')
    outs = []
    for sidecars in (None, [rb.next_gather_sidecar()]):
        job_pi, _ = fpi.job(xs, None)
        a_g, lp_g = torch.empty(B, T, A, device='cuda'), torch.empty(B, T, device='cuda')
        job_q, q_g = fq.job(xs, a_g.view(B * T, A))
        job = native.pi_q_job(job_pi, job_q, eps, a_g, lp_g, T)
        assert native.policy_sample_q_forward_ok(job)
        native.policy_sample_q_forward(job, sidecars=sidecars)
        outs.append((a_g, lp_g, q_g))
    torch.cuda.synchronize()
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    for k, v in want.items():
        assert torch.equal(alt[k], v), k
    if want_pre is not None:
        assert torch.equal(rb._alt['joint_pre_action'], want_pre)
