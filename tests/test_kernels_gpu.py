"""GPU parity: every libasac_hip.so entry point, called through the C ABI (ctypes), against the
oracle (`oracle/`) and the golden vectors minted from the reference.  Bit-exact for index / tree /
byte work; stated fp32 tolerances for the float chains (device libm vs host libm)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import sac_ref  # noqa: E402
from oracle.per_ref import PrioritizedReplayRef, SumTreeRef  # noqa: E402


@pytest.fixture(scope='module')
def nat():
    from asac_amd import native
    native.load()
    assert torch.cuda.is_available()
    return native


def dev(x, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(x))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


class DevTree:
    def __init__(self, nat, C, extra=0):
        self.nat, self.C = nat, C
        self.tree = torch.zeros(2 * C - 1, dtype=torch.float32, device='cuda')
        self.winner = torch.full((C + extra,), -1, dtype=torch.int32, device='cuda')
        self.nan_flag = torch.zeros(1, dtype=torch.int32, device='cuda')

    def set_priorities(self, idx, p):
        self.nat.sumtree_update(self.tree, self.C, dev(idx, torch.int64), None, dev(p, torch.float32),
                                0.9, 0.01, 1.0, 1, self.winner, self.nan_flag)


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('tag,C', [('c16', 16), ('c1024', 1024), ('c524288', 2 ** 19)])
def test_sumtree_update_sample_bit_exact(nat, golden_dir, tag, C):
    g = np.load(golden_dir / 'f1_sumtree.npz')
    t = DevTree(nat, C, extra=2 * 30000)
    ref = SumTreeRef(C)
    for i in ('1', '2'):
        t.set_priorities(g[f'{tag}_idx{i}'], g[f'{tag}_p{i}'])
        ref.update(g[f'{tag}_idx{i}'], g[f'{tag}_p{i}'])
    tree = t.tree.cpu().numpy()
    assert np.array_equal(bits(tree), bits(ref.tree)), 'tree bytes differ from the oracle'
    if C <= 1024:
        assert np.array_equal(bits(tree), bits(g[f'{tag}_tree']))
    else:
        assert np.bitwise_xor.reduce(bits(tree)) == g[f'{tag}_tree_xor']
    assert (t.winner[:C] == -1).all(), 'winner scratch must be handed back clean'
    chk = torch.zeros(1, dtype=torch.int32, device='cuda')
    nat.sumtree_check(t.tree, C, chk)
    assert chk.item() == 0

    B = int(g[f'{tag}_batch'])
    slot_ids = torch.arange(C, dtype=torch.int64, device='cuda') + 7 * C   # any id map
    for u_key, leaf_key, p_key in [('u', 'leaf', 'p'), ('ub', 'leaf_b', 'p_b')]:
        leaf = torch.zeros(B, dtype=torch.int32, device='cuda')
        p = torch.zeros(B, dtype=torch.float32, device='cuda')
        ids = torch.zeros(B, dtype=torch.int64, device='cuda')
        w = torch.zeros(B, dtype=torch.float32, device='cuda')
        beta = torch.tensor([0.4], dtype=torch.float64, device='cuda')
        minp = torch.zeros(2, dtype=torch.float32, device='cuda')
        nat.sumtree_sample(t.tree, C, B, dev(g[f'{tag}_{u_key}'], torch.float64), slot_ids, beta, 0.001,
                           leaf, p, ids, w, minp)
        assert np.array_equal(leaf.cpu().numpy(), g[f'{tag}_{leaf_key}']), 'PER index selection'
        assert np.array_equal(bits(p.cpu().numpy()), bits(g[f'{tag}_{p_key}']))
        assert np.array_equal(ids.cpu().numpy(), g[f'{tag}_{leaf_key}'].astype(np.int64) - (C - 1) + 7 * C)
        # importance weights: float64 pow on device vs NumPy, allow 1 ulp of float32
        pr = g[f'{tag}_{p_key}']
        ratio = pr / ref.tree[0]
        got_w = w.cpu().numpy()
        if ratio.min() > 0:
            w_ref = np.power(ratio / np.min(ratio), -np.float64(0.401)).astype(np.float32)
            np.testing.assert_allclose(got_w, w_ref, rtol=2e-7, atol=0)
            assert np.isfinite(got_w).all() and got_w.max() == 1.0
        else:
            # a zero-priority leaf was drawn (the `right == 0` rule of the descent): the reference divides by a minimum
            # ratio of 0 (replay_buffer.py:352-354) — weight 0 for every row with p > 0 (inf ** -beta), NaN for the rows
            # with p == 0 (0 / 0).  Asserted as such, not through NaN == NaN; non-degenerate weights are compared in
            # test_sumtree_sample_multiblock_matches_oracle and in every step test
            assert (pr == 0).any()
            assert np.array_equal(np.isnan(got_w), pr == 0) and (got_w[pr > 0] == 0).all()
        assert beta.item() == pytest.approx(0.401, abs=0) and minp[0].item() == pr.min()
    mx = torch.zeros(1, dtype=torch.float32, device='cuda')
    nat.sumtree_leaf_max(t.tree, C, mx)
    assert mx.item() == g[f'{tag}_max']


@pytest.mark.parametrize('B', [3000, 700, 1024])
def test_sumtree_sample_multiblock_matches_oracle(nat, B):
    """batch > 1024 takes the multi-workgroup + two-pass-weights path, 256 < batch <= 1024 the single
    workgroup strides over the batch."""
    C = 4096
    rng = np.random.default_rng(0)
    ref = SumTreeRef(C)
    idx = rng.permutation(C)[:3000]
    p = rng.random(3000).astype(np.float32)
    ref.update(idx, p)
    t = DevTree(nat, C, extra=6000)
    t.set_priorities(idx, p)
    assert np.array_equal(bits(t.tree.cpu().numpy()), bits(ref.tree))
    u = rng.random(B)
    leaf_ref, p_ref = ref.sample(B, u)
    leaf = torch.zeros(B, dtype=torch.int32, device='cuda')
    pp = torch.zeros(B, dtype=torch.float32, device='cuda')
    ids = torch.zeros(B, dtype=torch.int64, device='cuda')
    w = torch.zeros(B, dtype=torch.float32, device='cuda')
    beta = torch.tensor([0.999], dtype=torch.float64, device='cuda')
    minp = torch.zeros(2, dtype=torch.float32, device='cuda')
    nat.sumtree_sample(t.tree, C, B, dev(u), torch.arange(C, dtype=torch.int64, device='cuda'), beta, 0.001,
                       leaf, pp, ids, w, minp)
    assert np.array_equal(leaf.cpu().numpy(), leaf_ref)
    ratio = p_ref / ref.tree[0]
    w_ref = np.power(ratio / np.min(ratio), -np.float64(1.0)).astype(np.float32)
    np.testing.assert_allclose(w.cpu().numpy(), w_ref, rtol=2e-7)
    assert beta.item() == 1.0


def test_sumtree_update_priority_mode_and_stale_and_nan(nat):
    C, k = 256, 100
    rng = np.random.default_rng(1)
    rb = PrioritizedReplayRef(batch_size=8, capacity=C)
    rb.storage.add({'x': np.zeros(C + 50, np.float32)})       # ids 0..C+49, slots wrapped once
    rb.tree.update(np.arange(C), rng.random(C).astype(np.float32))
    t = DevTree(nat, C)
    t.tree.copy_(dev(rb.tree.tree))
    slot_ids = dev(rb.storage.columns['_id'])
    ids = rng.integers(0, C + 50, size=k).astype(np.int64)     # some stale (overwritten), some dup
    td = np.abs(rng.standard_normal(k)).astype(np.float32)
    rb.update(ids, td)
    nat.sumtree_update(t.tree, C, dev(ids), slot_ids, dev(td), 0.9, 0.01, 1.0, 0, t.winner, t.nan_flag)
    got, want = t.tree.cpu().numpy(), rb.tree.tree
    # pow() on device vs NumPy float32 power: leaves within 1 ulp; parents are exact sums of leaves
    np.testing.assert_allclose(got, want, rtol=3e-7, atol=0)
    assert t.nan_flag.item() == 0
    leaves_before = got.copy()
    td[3] = np.nan
    nat.sumtree_update(t.tree, C, dev(ids), slot_ids, dev(td), 0.9, 0.01, 1.0, 0, t.winner, t.nan_flag)
    assert t.nan_flag.item() == 1
    assert np.array_equal(bits(t.tree.cpu().numpy()), bits(leaves_before)), 'NaN batch must not touch the tree'


def test_per_add_matches_oracle(nat):
    C = 64
    rb = PrioritizedReplayRef(batch_size=4, capacity=C)
    t = DevTree(nat, C)
    slot_ids = torch.zeros(C, dtype=torch.int64, device='cuda')
    mx = torch.zeros(1, dtype=torch.float32, device='cuda')
    rng = np.random.default_rng(3)
    next_id, size = 0, 0
    for it in range(40):
        T = int(rng.integers(1, 30)) if it != 20 else 150        # one episode longer than the ring
        rb.add({'x': np.zeros(T, np.float32)}, ignore_size=1)
        if size == 0:
            nat.per_add(t.tree, C, next_id, T, 1, None, 1.0, slot_ids)
        else:
            nat.sumtree_leaf_max(t.tree, C, mx)
            nat.per_add(t.tree, C, next_id, T, 1, mx, 0.0, slot_ids)
        size = min(size + T, C)
        next_id = (next_id + T) % (10 * C)
        assert np.array_equal(bits(t.tree.cpu().numpy()), bits(rb.tree.tree)), f'add #{it}'
        assert np.array_equal(slot_ids.cpu().numpy(), rb.storage.columns['_id'])
        # randomise priorities so the running max is not constant
        idx = rng.integers(0, C, 10)
        pr = rng.random(10).astype(np.float32)
        rb.tree.update(idx, pr)
        t.set_priorities(idx, pr)


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('prev_n,post_n,B', [(0, 4, 64), (3, 2, 33), (40, 40, 16)])
def test_window_gather_pad_matches_oracle(nat, prev_n, post_n, B):
    C = 512
    L = prev_n + 1 + post_n
    rng = np.random.default_rng(prev_n)
    rb = PrioritizedReplayRef(batch_size=B, sample_prev_n=prev_n, sample_post_n=post_n, capacity=C)
    cols = None
    while rb.storage.size < C or rb.storage.next_id < C + 100:      # wrap the ring
        T = int(rng.integers(5, 60))
        ep = {'index': np.arange(T, dtype=np.int32), 'last_mask': np.zeros(T, bool),
              'obs_vec': rng.standard_normal((T, 6)).astype(np.float32),
              'obs_img': rng.integers(0, 255, (T, 3, 4, 4)).astype(np.uint8),
              'obs_flag': rng.integers(0, 2, (T, 5)).astype(bool),
              'obs_wide': rng.standard_normal((T, 36)).astype(np.float32),   # 144 B rows: 16 B units
              # 2 000 B rows (125 units: windows of >= 512 units take the span form, a workgroup per window chunk), one
              # kept as stored, one padded like a hidden state
              'obs_frame': rng.standard_normal((T, 500)).astype(np.float32),
              'hid_frame': rng.standard_normal((T, 2, 250)).astype(np.float32),
              'action': rng.random((T, 3)).astype(np.float32),
              'reward': rng.standard_normal(T).astype(np.float32), 'done': rng.integers(0, 2, T).astype(bool),
              'mu_prob': rng.random((T, 3)).astype(np.float32),
              'pre_seq_hidden_state': rng.standard_normal((T, 2, 4)).astype(np.float32)}
        ep['last_mask'][-1] = True
        rb.add(ep, ignore_size=1)
        cols = list(ep)
    ids = np.concatenate([rng.integers(0, rb.storage.next_id, B - 3),
                          [0, 1, rb.storage.next_id - 1]]).astype(np.int64)   # incl. negative window ids
    win = rb.storage.rows_at((ids[:, None] + np.arange(-prev_n, post_n + 1)[None]).reshape(-1))
    batch = {k: torch.as_tensor(v.reshape(B, L, *v.shape[1:]).copy()) for k, v in win.items()}
    pad_action = torch.tensor([1., 0., 0.])
    sac_ref.pad_window(batch, prev_n, pad_action)
    batch['hid_frame'][batch['padding_mask']] = 0.
    batch['obs_img'] = batch['obs_img'].float() / 255.
    batch['obs_flag'] = batch['obs_flag'].float()

    ring = {k: dev(rb.storage.columns[k]) for k in cols}
    out = {k: torch.zeros_like(v).cuda() for k, v in batch.items()}
    pad_row = pad_action.cuda()
    f32 = lambda x: int(np.float32(x).view(np.uint32))  # noqa: E731
    spec = {'index': (nat.PAD_WORD, 0xffffffff), 'last_mask': (nat.PAD_KEEP, 0), 'obs_vec': (nat.PAD_KEEP, 0),
            'obs_wide': (nat.PAD_KEEP, 0), 'obs_frame': (nat.PAD_KEEP, 0), 'hid_frame': (nat.PAD_WORD, f32(0.)),
            'action': (nat.PAD_ROW, 0), 'reward': (nat.PAD_WORD, f32(0.)),
            'done': (nat.PAD_BYTE, 1), 'mu_prob': (nat.PAD_WORD, f32(1.)),
            'pre_seq_hidden_state': (nat.PAD_WORD, f32(0.))}
    specs = [dict(src=ring[k], dst=out[k], row_bytes=ring[k][0].numel() * ring[k].element_size(),
                  pad_mode=m, pad_word=w, pad_row=pad_row if m == nat.PAD_ROW else None)
             for k, (m, w) in spec.items()]
    specs.append(dict(src=ring['obs_img'], dst=out['obs_img'], row_bytes=48, pad_mode=nat.PAD_KEEP,
                      convert=nat.CVT_U8_TO_F32_UNIT))
    specs.append(dict(src=ring['obs_flag'], dst=out['obs_flag'], row_bytes=5, pad_mode=nat.PAD_KEEP,
                      convert=nat.CVT_BOOL_TO_F32))
    specs.append(dict(src=None, dst=out['padding_mask'], pad_mode=nat.PAD_EMIT_MASK))
    keys = nat.make_gather_keys(specs)
    nat.window_gather_pad(keys, dev(ids), B, prev_n, post_n, C, ring['index'])
    torch.cuda.synchronize()
    for k, want in batch.items():
        got = out[k].cpu()
        assert torch.equal(got.view(torch.uint8) if got.dtype != torch.bool else got,
                           want.view(torch.uint8) if want.dtype != torch.bool else want), k
    assert batch['padding_mask'].any() and not batch['padding_mask'].all()


def test_scatter_rows_matches_oracle(nat):
    C, B, b, n, A = 128, 24, 2, 3, 4
    rng = np.random.default_rng(5)
    rb = PrioritizedReplayRef(batch_size=B, sample_prev_n=b, sample_post_n=n, capacity=C)
    rb.storage.add({'mu_prob': rng.random((C + 40, A)).astype(np.float32)})
    ids = rng.integers(0, C + 40, B).astype(np.int64)
    ids[:6] = ids[6:12] + 1                                       # overlapping windows -> duplicate targets
    pad = rng.random((B, b + n + 1)) < 0.25
    new = rng.random((B, b + n, A)).astype(np.float32)
    ring = dev(rb.storage.columns['mu_prob'])
    slot_ids = dev(rb.storage.columns['_id'])
    winner = torch.full((C,), -1, dtype=torch.int32, device='cuda')
    for first_off in (-b, 1 - b):
        tgt = np.stack([ids + first_off + j for j in range(b + n)], axis=1).reshape(-1)
        keep = ~pad[:, :b + n].reshape(-1)
        rb.update_transitions(tgt[keep], 'mu_prob', new.reshape(-1, A)[keep])
        d_new, d_pad = dev(new), dev(pad)
        nat.scatter_rows_if_id_match(ring, A * 4, C, dev(ids), B, first_off, b + n, slot_ids, d_pad,
                                     b + n + 1, d_new, (b + n) * A * 4, A * 4, winner)
        assert np.array_equal(bits(ring.cpu().numpy()), bits(rb.storage.columns['mu_prob']))
        assert (winner == -1).all()


# ------------------------------------------------------------------------------------------------
def _vtrace_args(nat, *, q, logp, log_alpha, reward, done, last, pad, mu, pi, A, gamma_ratio, lambda_ratio,
                 gamma, rho, c, use_is, y, subset_n=None, subset_next=None, E_sample=None, q_online=None, td=None):
    a = nat.VtraceArgs()
    B, n = reward.shape
    if q is not None:
        a.q = q.data_ptr()
        a.q_stride_e, a.q_stride_b, a.q_stride_t = q.stride(0), q.stride(1), q.stride(2)
        a.logp, a.log_alpha = logp.data_ptr(), log_alpha.data_ptr()
        a.E_sample = E_sample or q.shape[0]
        a.subset_n = subset_n.data_ptr() if subset_n is not None else None
        a.subset_next = subset_next.data_ptr() if subset_next is not None else None
    a.reward, a.reward_stride = reward.data_ptr(), reward.stride(0)
    a.done, a.last_mask, a.padding_mask = done.data_ptr(), last.data_ptr(), pad.data_ptr()
    a.mask_stride = done.stride(0)
    if use_is and mu is not None:
        a.mu_prob, a.mu_stride_b, a.mu_stride_t, a.mu_offset = mu.data_ptr(), mu.stride(0), mu.stride(1), 0
        a.pi_prob, a.pi_stride_b, a.pi_stride_t, a.A = pi.data_ptr(), pi.stride(0), pi.stride(1), A
    a.gamma_ratio, a.lambda_ratio = gamma_ratio.data_ptr(), lambda_ratio.data_ptr()
    a.gamma, a.v_rho, a.v_c, a.use_n_step_is, a.B, a.n = gamma, rho, c, int(use_is), B, n
    if q_online is not None:
        a.q_online, a.E_online, a.td_error_out = q_online.data_ptr(), q_online.shape[0], td.data_ptr()
    a.y_out = y.data_ptr()
    return a


@pytest.mark.parametrize('n', [1, 4, 40])
@pytest.mark.parametrize('use_is', [True, False])
def test_vtrace_direct_vs_golden(nat, golden_dir, n, use_is):
    g = np.load(golden_dir / 'f3_vtrace.npz')
    gamma, lam, rho, c = (float(x) for x in g['params'])
    tag = f'n{n}_is{int(use_is)}'
    t = {k: dev(g[f'{tag}_{k}']) for k in ('n_last_masks', 'n_padding_masks', 'n_rewards', 'n_dones',
                                              'n_mu_probs', 'n_pi_probs', 'n_vs', 'next_n_vs',
                                              'gamma_ratio', 'lambda_ratio')}
    B = t['n_rewards'].shape[0]
    y = torch.zeros(B, device='cuda')
    a = _vtrace_args(nat, q=None, logp=None, log_alpha=None, reward=t['n_rewards'], done=t['n_dones'],
                     last=t['n_last_masks'], pad=t['n_padding_masks'], mu=None, pi=None, A=0,
                     gamma_ratio=t['gamma_ratio'], lambda_ratio=t['lambda_ratio'], gamma=gamma, rho=rho, c=c,
                     use_is=use_is, y=y)
    nat.vtrace_return_direct(a, t['n_vs'], t['next_n_vs'], t['n_pi_probs'], t['n_mu_probs'])
    # sequential f32 sum on device vs torch's vectorised CPU sum: a few ulp on |y| ~ 1..10
    np.testing.assert_allclose(y.cpu().numpy()[:, None], g[f'{tag}_y'], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('tag', ['n4_e2', 'n3_e4s2', 'n40_e2', 'n1_e2_nois'])
def test_get_y_pipeline_vs_golden(nat, golden_dir, tag):
    """rsample + squash log-prob + stored-action probs + ensemble subset/min + V + V-trace,
    against the reference's _get_y driven with tabulated policy / target-Q outputs."""
    g = np.load(golden_dir / 'f4_get_y.npz')
    n, E, Es, A, use_is = (int(x) for x in g[f'{tag}_cfg'])
    loc, scale, eps = dev(g[f'{tag}_loc']), dev(g[f'{tag}_scale']), dev(g[f'{tag}_eps'])
    B = loc.shape[0]
    a_tanh = torch.zeros_like(loc)
    logp = torch.zeros(B, n + 1, device='cuda')
    nat.squash_sample_fwd(loc, scale, eps, a_tanh, logp)
    # cross-check the kernel against the oracle's eager ops
    dist = torch.distributions.Normal(loc.cpu(), scale.cpu(), validate_args=False)
    x = loc.cpu() + eps.cpu() * scale.cpu()
    lp_ref = sac_ref.masked_sum_log_prob(sac_ref.squash_log_prob(dist, x))
    np.testing.assert_allclose(logp.cpu().numpy(), lp_ref.numpy(), rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(a_tanh.cpu().numpy(), torch.tanh(x).numpy(), rtol=1e-6, atol=1e-6)

    actions = torch.zeros(B, n + 1, A, device='cuda')
    actions[:, :n] = dev(g[f'{tag}_n_actions'])
    pi = torch.zeros(B, n + 1, A, device='cuda')
    nat.squash_prob(loc, scale, actions, 0, pi, 0)
    stored = torch.atanh(torch.clamp(actions.cpu(), -0.999, 0.999))
    np.testing.assert_allclose(pi.cpu().numpy(), sac_ref.squash_prob(dist, stored).numpy(), rtol=3e-5, atol=1e-7)
    # the single-launch form (sample + stored-action probabilities) on the (loc | scale) halves of one
    # [B, n+1, 2A] tensor must give the same bits as the two separate launches on dense tensors
    ls = torch.cat([loc, scale], dim=-1).contiguous()
    a2, lp2, pi2 = torch.zeros_like(loc), torch.zeros_like(logp), torch.zeros_like(pi)
    nat.squash_sample_fwd(ls[..., :A], ls[..., A:], eps, a2, lp2, None, actions, 0, pi2, 0)
    assert torch.equal(a2, a_tanh) and torch.equal(lp2, logp) and torch.equal(pi2, pi)

    q = dev(g[f'{tag}_q']).squeeze(-1).contiguous()              # [E, B, n+1]
    perm = g[f'{tag}_perm']
    sub_n, sub_next = dev(perm[0][:Es], torch.int32), dev(perm[1][:Es], torch.int32)
    y = torch.zeros(B, device='cuda')
    gr = torch.logspace(0, n - 1, n, 0.99).cuda()
    lr = torch.logspace(0, n - 1, n, 1.0).cuda()
    a = _vtrace_args(nat, q=q, logp=logp, log_alpha=dev(g[f'{tag}_log_alpha']).reshape(1),
                     reward=dev(g[f'{tag}_n_rewards']), done=dev(g[f'{tag}_n_dones']),
                     last=dev(g[f'{tag}_n_last_masks']), pad=dev(g[f'{tag}_n_padding_masks']),
                     mu=dev(g[f'{tag}_n_mu_probs']), pi=pi, A=A, gamma_ratio=gr, lambda_ratio=lr,
                     gamma=0.99, rho=1.0, c=1.0, use_is=bool(use_is), y=y, subset_n=sub_n,
                     subset_next=sub_next, E_sample=Es)
    nat.vtrace_return_min(a)
    np.testing.assert_allclose(y.cpu().numpy()[:, None], g[f'{tag}_y'], rtol=2e-5, atol=2e-5)


def test_return_with_the_pending_temperature_step(nat, golden_dir):
    """The TD error's return must see the temperature AFTER the step's temperature update.  One-launch steps defer
    that update to the priority update's launch (a sidecar) and give the return launch the pending job instead:
    return(pending) + tree update(sidecar) == temperature step, then return, then tree update — bit for bit."""
    g = np.load(golden_dir / 'f4_get_y.npz')
    tag = 'n4_e2'
    n, E, Es, A, use_is = (int(x) for x in g[f'{tag}_cfg'])
    loc, scale, eps = dev(g[f'{tag}_loc']), dev(g[f'{tag}_scale']), dev(g[f'{tag}_eps'])
    B = loc.shape[0]
    a_tanh, logp, pi = torch.zeros_like(loc), torch.zeros(B, n + 1, device='cuda'), torch.zeros(B, n + 1, A, device='cuda')
    actions = torch.zeros(B, n + 1, A, device='cuda')
    actions[:, :n] = dev(g[f'{tag}_n_actions'])
    nat.squash_sample_fwd(loc, scale, eps, a_tanh, logp, None, actions, 0, pi, 0)
    q = dev(g[f'{tag}_q']).squeeze(-1).contiguous()
    perm = g[f'{tag}_perm']
    sub_n, sub_next = dev(perm[0][:Es], torch.int32), dev(perm[1][:Es], torch.int32)
    gr, lr = torch.logspace(0, n - 1, n, 0.99).cuda(), torch.logspace(0, n - 1, n, 1.0).cuda()
    q_on = torch.randn(E, B, device='cuda')
    C = 1024
    rng = np.random.default_rng(1)
    assert B <= C
    ids = dev(rng.permutation(C)[:B].astype(np.int64))
    la0 = float(np.asarray(g[f'{tag}_log_alpha']).reshape(-1)[0])
    alpha_logp = dev(rng.standard_normal(B).astype(np.float32) * 2)

    def run(deferred, merged=False):
        seg = torch.tensor([0.3, la0], device='cuda')      # [log_d_alpha, log_c_alpha]
        state = [torch.zeros(2, device='cuda'), torch.full((2,), 0.01, device='cuda'), torch.full((2,), 1e-3, device='cuda')]
        steps = torch.tensor(7, dtype=torch.int64, device='cuda')
        tree, slot_ids = torch.zeros(2 * C - 1, device='cuda'), torch.arange(C, dtype=torch.int64, device='cuda')
        winner, nan_flag = torch.full((C + 2 * 4096,), -1, dtype=torch.int32, device='cuda'), torch.zeros(1, dtype=torch.int32, device='cuda')
        y, td = torch.zeros(B, device='cuda'), torch.zeros(B, device='cuda')
        a = _vtrace_args(nat, q=q, logp=logp, log_alpha=seg[1:], reward=dev(g[f'{tag}_n_rewards']),
                         done=dev(g[f'{tag}_n_dones']), last=dev(g[f'{tag}_n_last_masks']),
                         pad=dev(g[f'{tag}_n_padding_masks']), mu=dev(g[f'{tag}_n_mu_probs']), pi=pi, A=A,
                         gamma_ratio=gr, lambda_ratio=lr, gamma=0.99, rho=1.0, c=1.0, use_is=bool(use_is), y=y,
                         subset_n=sub_n, subset_next=sub_next, E_sample=Es)
        a.q_online, a.E_online, a.td_error_out = q_on.data_ptr(), E, td.data_ptr()
        job = nat.sidecar_alpha_adam(alpha_logp, -float(A), 1, seg, *state, 3e-4, 0.9, 0.999, 1e-8, steps, advance_counter=True)
        if merged:      # ... and the return itself inside the priority update's launch, the temperature step in front
            nat.td_update(a, tree, C, ids, slot_ids, 0.9, 0.01, 1.0, winner, nan_flag, alpha_step=job)
        elif deferred:
            nat.vtrace_return_min(a, pending_alpha=job)
            assert float(seg[1]) == float(np.float32(la0)), 'the return launch only previews the step'
            nat.sumtree_update(tree, C, ids, slot_ids, td, 0.9, 0.01, 1.0, 0, winner, nan_flag, sidecars=[job])
        else:
            nat.alpha_adam_step(alpha_logp, -float(A), 1, seg, *state, 3e-4, 0.9, 0.999, 1e-8, steps, advance_counter=True)
            nat.vtrace_return_min(a)
            nat.sumtree_update(tree, C, ids, slot_ids, td, 0.9, 0.01, 1.0, 0, winner, nan_flag)
        return [y, td, seg, *state, steps, tree]

    want = run(False)
    assert float(want[2][1]) != float(np.float32(la0))
    for got in (run(True), run(True, merged=True)):
        for name, w_, g_ in zip(('y', 'td', 'temperatures', 'grad', 'exp_avg', 'exp_avg_sq', 'steps', 'tree'), want, got):
            assert torch.equal(w_, g_), name


@pytest.mark.parametrize('B,n', [(256, 1), (256, 3), (64, 8), (300, 17), (1024, 5), (1000, 14), (255, 4), (7, 2)])
@pytest.mark.parametrize('use_is,ordered', [(True, True), (False, True), (True, False)])
def test_td_error_and_priority_update_in_one_launch(nat, B, n, use_is, ordered):
    """asac_td_update == asac_vtrace_return_min (TD errors) + asac_sumtree_update, bit for bit: returns, TD errors, the
    tree (duplicate ids: the last writer wins; stale ids are skipped) — ids in leaf order (the sampler's) and in any order."""
    torch.manual_seed(B * 31 + n)
    E, A = 3, 2
    C = 2 ** 19 if n in (3, 17, 4) else 2048
    f = dict(device='cuda')
    q = torch.randn(E, B, n + 1, **f)
    logp, log_alpha = torch.randn(B, n + 1, **f), torch.tensor([-1.2], **f)
    reward, done = torch.randn(B, n, **f), torch.rand(B, n, **f) < 0.1
    last, pad = torch.rand(B, n, **f) < 0.05, torch.rand(B, n, **f) < 0.1
    mu, pi = torch.rand(B, n, A, **f) + 0.05, torch.rand(B, n + 1, A, **f) + 0.05
    gr, lr = torch.logspace(0, n - 1, n, 0.99).cuda(), torch.logspace(0, n - 1, n, 0.95).cuda()
    q_on = torch.randn(E, B, **f)
    ids = torch.randint(0, C, (B,), device='cuda')
    ids[B // 2:B // 2 + min(8, B // 2)] = ids[:min(8, B // 2)]                    # duplicates
    if ordered:      # what the stratified sampler hands out: leaf order (duplicates are neighbours, stale rows between them)
        ids = ids.sort().values
    slot0 = torch.arange(C, dtype=torch.int64, device='cuda')
    slot0[ids[3]] += C                                  # overwritten since it was sampled: skipped
    start = DevTree(nat, C, extra=2 * 4096)
    some = np.random.default_rng(0).permutation(C)[:2048]
    start.set_priorities(some, np.random.default_rng(0).random(2048).astype(np.float32))
    base_tree = start.tree
    rows = torch.randn(B, 4, **f)

    def run(merged):
        tree, slot_ids = base_tree.clone(), slot0.clone()
        winner, nan_flag = torch.full((C + 2 * 4096,), -1, dtype=torch.int32, device='cuda'), torch.zeros(1, dtype=torch.int32, device='cuda')
        y, td = torch.zeros(B, **f), torch.zeros(B, **f)
        a = _vtrace_args(nat, q=q, logp=logp, log_alpha=log_alpha, reward=reward, done=done, last=last, pad=pad, mu=mu, pi=pi,
                         A=A, gamma_ratio=gr, lambda_ratio=lr, gamma=0.99, rho=1.0, c=1.0, use_is=use_is, y=y,
                         q_online=q_on, td=td)
        if merged:
            nat.td_update(a, tree, C, ids, slot_ids, 0.9, 0.01, 1.0, winner, nan_flag)
        else:
            nat.vtrace_return_min(a)
            nat.sumtree_update(tree, C, ids, slot_ids, td, 0.9, 0.01, 1.0, 0, winner, nan_flag)
        assert int(nan_flag) == 0 and bool((winner[:C] == -1).all())
        # every parent == left + right after the climb (its barriers are taken by the waves that own rows only: a batch
        # that is not a multiple of the wave size leaves a partly filled last wave at them)
        chk = torch.zeros(1, dtype=torch.int32, device='cuda')
        nat.sumtree_check(tree, C, chk)
        assert int(chk) == 0
        return y, td, tree

    for name, w_, g_ in zip(('y', 'td', 'tree'), run(False), run(True)):
        assert torch.equal(w_, g_), name


def test_squash_sample_backward_vs_autograd(nat):
    torch.manual_seed(0)
    B, A = 257, 3
    loc = torch.randn(B, A, requires_grad=True)
    scale = torch.rand(B, A).add(0.05).requires_grad_()
    eps = torch.randn(B, A) * 1.5
    ga, gl = torch.randn(B, A), torch.randn(B)
    x = loc + eps * scale
    dist = torch.distributions.Normal(loc, scale, validate_args=False)
    logp = sac_ref.masked_sum_log_prob(sac_ref.squash_log_prob(dist, x))
    (torch.tanh(x) * ga).sum().add((logp * gl).sum()).backward()
    g_loc, g_scale = torch.zeros(B, A, device='cuda'), torch.zeros(B, A, device='cuda')
    nat.squash_sample_bwd(loc.detach().cuda(), scale.detach().cuda(), eps.cuda(), ga.cuda(), gl.cuda(), g_loc, g_scale)
    np.testing.assert_allclose(g_loc.cpu().numpy(), loc.grad.numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(g_scale.cpu().numpy(), scale.grad.numpy(), rtol=1e-4, atol=2e-4)


@pytest.mark.parametrize('use_w', [True, False])
def test_q_loss_vs_autograd(nat, use_w):
    torch.manual_seed(1)
    E, B, eps = 3, 256, 0.2
    q = torch.randn(E, B, requires_grad=True)
    tq = q.detach() + torch.randn(E, B) * 0.3
    y, w = torch.randn(B), torch.rand(B) + 0.1
    losses = []
    for e in range(E):
        clipped = tq[e] + torch.clamp(q[e] - tq[e], -eps, eps)
        l = torch.maximum((clipped - y) ** 2, (q[e] - y) ** 2)
        losses.append(torch.mean(l * w if use_w else l))
    torch.stack(losses).sum().backward()
    loss_out = torch.zeros(E, device='cuda')
    grad = torch.zeros(E, B, device='cuda')
    nat.q_loss_fwd_bwd(q.detach().cuda(), tq.cuda(), y.cuda(), w.cuda() if use_w else None, eps, loss_out, grad)
    np.testing.assert_allclose(loss_out.cpu().numpy(), torch.stack(losses).detach().numpy(), rtol=1e-5)
    np.testing.assert_allclose(grad.cpu().numpy(), q.grad.numpy(), rtol=1e-5, atol=1e-8)


def test_polyak_bit_exact_vs_golden(nat, golden_dir):
    g = np.load(golden_dir / 'f5_polyak.npz')
    n = sum(1 for k in g.files if k.startswith('src_'))
    src = np.concatenate([g[f'src_{i}'].reshape(-1) for i in range(n)])
    before = np.concatenate([g[f'before_{i}'].reshape(-1) for i in range(n)])
    after = np.concatenate([g[f'after_{i}'].reshape(-1) for i in range(n)])
    for shift in (0, 1):          # aligned (float4) and misaligned (scalar) paths
        t = torch.zeros(len(src) + 4, device='cuda')[shift:shift + len(src)]
        s = torch.zeros(len(src) + 4, device='cuda')[shift:shift + len(src)]
        t.copy_(dev(before)), s.copy_(dev(src))
        nat.polyak(t, s, float(g['tau']))
        assert np.array_equal(bits(t.cpu().numpy()), bits(after))


def test_adam_vs_torch(nat):
    torch.manual_seed(2)
    n = 10007
    p = torch.randn(n)
    ref = p.clone().requires_grad_()
    opt = torch.optim.Adam([ref], lr=3e-4)
    dp = p.cuda()
    m, v = torch.zeros(n, device='cuda'), torch.zeros(n, device='cuda')
    steps = torch.zeros(1, dtype=torch.int64, device='cuda')
    for _ in range(5):
        g = torch.randn(n)
        ref.grad = g.clone()
        opt.step()
        nat.adam_step(dp, g.cuda(), m, v, 3e-4, 0.9, 0.999, 1e-8, steps)
        steps += 1
    np.testing.assert_allclose(dp.cpu().numpy(), ref.detach().numpy(), rtol=1e-6, atol=1e-7)


def test_kernels_are_graph_capturable(nat):
    """The step is replayed as one hipGraph: launches issued through ctypes on torch's capture
    stream must record and replay."""
    n = 4096
    t, s = torch.zeros(n, device='cuda'), torch.ones(n, device='cuda')
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        nat.polyak(t, s, 0.5)
    torch.cuda.current_stream().wait_stream(side)
    t.zero_()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        nat.polyak(t, s, 0.5)
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    assert torch.allclose(t, torch.full_like(t, 1 - 0.5 ** 3))


def test_noise_fill_distribution_and_stream_position(nat):
    """`asac_noise_fill`: uniforms in [0,1) and N(0,1) draws pass KS tests, a launch is a pure function
    of (seed, step counter), and different steps / seeds give different blocks."""
    from scipy import stats
    native = nat
    step = torch.zeros(1, dtype=torch.int64, device='cuda')
    u = torch.empty(100_003, dtype=torch.float64, device='cuda')
    z = torch.empty(400_001, dtype=torch.float32, device='cuda')
    native.noise_fill(1234, step, u, z)
    u0, z0 = u.cpu().numpy().copy(), z.cpu().numpy().copy()
    assert (u0 >= 0).all() and (u0 < 1).all() and np.isfinite(z0).all()
    assert stats.kstest(u0, 'uniform').pvalue > 1e-3
    assert stats.kstest(z0.astype(np.float64), 'norm').pvalue > 1e-3
    assert abs(z0.mean()) < 0.01 and abs(z0.std() - 1) < 0.01
    assert abs(np.corrcoef(z0[:-1], z0[1:])[0, 1]) < 0.01          # the Box-Muller pairs are independent
    native.noise_fill(1234, step, u, z)                               # same (seed, step) -> same block
    assert np.array_equal(u.cpu().numpy(), u0) and np.array_equal(z.cpu().numpy(), z0)
    step.add_(1)
    native.noise_fill(1234, step, u, z)
    assert not np.array_equal(z.cpu().numpy()[:1000], z0[:1000])
    assert abs(np.corrcoef(z.cpu().numpy(), z0)[0, 1]) < 0.01
    step.zero_()
    native.noise_fill(1235, step, u, z)
    assert not np.array_equal(u.cpu().numpy()[:1000], u0[:1000])
    native.noise_fill(7, step, None, z[:5])                           # either output alone, ragged tails
    native.noise_fill(7, step, u[:3], None)
    # ensemble subsets: distinct members in range, every member equally likely in every position
    subs = torch.full((3000, 3), -1, dtype=torch.int32, device='cuda')
    native.noise_fill(11, step, None, None, subs, 5)
    sb = subs.cpu().numpy()
    assert sb.min() == 0 and sb.max() == 4 and all(len(set(r)) == 3 for r in sb)
    for col in range(3):
        counts = np.bincount(sb[:, col], minlength=5)
        assert stats.chisquare(counts).pvalue > 1e-3
    native.noise_fill(11, step, None, None, subs, 3)                  # E_sample == E: a permutation
    assert all(sorted(r) == [0, 1, 2] for r in subs.cpu().numpy()[:50])


@pytest.mark.parametrize('with_polyak,with_zero', [(True, True), (True, False), (False, True)])
def test_step_prologue_is_polyak_plus_memset_plus_noise(nat, with_polyak, with_zero):
    """`asac_step_prologue`: the same draws as `asac_noise_fill`, the same target as `asac_polyak`, and the
    gradient buffer cleared (odd lengths, a misaligned start), neighbours untouched."""
    step = torch.full((1,), 5, dtype=torch.int64, device='cuda')
    u, z = torch.empty(777, dtype=torch.float64, device='cuda'), torch.empty(4099, device='cuda')
    nat.noise_fill(99, step, u, z)
    u0, z0 = u.clone(), z.clone()
    g = torch.Generator().manual_seed(0)
    target, source = torch.randn(100_003, generator=g).cuda(), torch.randn(100_003, generator=g).cuda()
    want_t = target.clone()
    nat.polyak(want_t, source, 0.005)
    kept = target.clone()
    whole = torch.randn(300_010, generator=g).cuda()
    grad = whole[3:300_004]                                 # 4-byte aligned only, length not a multiple of 4
    u.zero_(); z.zero_()
    nat.step_prologue((target, source, 0.005) if with_polyak else None, grad if with_zero else None, 99, step, u, z)
    assert torch.equal(u, u0) and torch.equal(z, z0)
    assert torch.equal(target, want_t if with_polyak else kept)
    if with_zero:
        assert not grad.any() and whole[:3].all() and whole[300_004:].all()
    else:
        assert whole.all()


@pytest.mark.parametrize('B', [256, 257, 512, 700, 1024])
def test_step_prologue_with_sampler_is_prologue_then_sampler(nat, B):
    """`asac_step_prologue_sample` == `asac_step_prologue` followed by `asac_sumtree_sample` on the uniforms it drew:
    leaves, priorities, ids, IS weights, beta, minimum, and the prologue's own outputs, bit for bit.  With the weights
    deferred (is_weights_out = NULL: sharded replay) the launch leaves beta alone and stores min p and min p / total;
    `asac_per_is_weights` on that ratio then writes the very same weights and advances beta, in one launch."""
    C = 4096
    rng = np.random.default_rng(5)
    t = DevTree(nat, C, extra=2 * 4096)
    idx = rng.permutation(C)[:3000]
    t.set_priorities(idx, (rng.random(3000) + 0.01).astype(np.float32))
    slot_ids = torch.arange(C, dtype=torch.int64, device='cuda') * 3 + 1
    step = torch.full((1,), 11, dtype=torch.int64, device='cuda')
    g = torch.Generator().manual_seed(1)
    source = torch.randn(50_001, generator=g).cuda()
    target0 = torch.randn(50_001, generator=g).cuda()

    def outs():
        return dict(leaf=torch.zeros(B, dtype=torch.int32, device='cuda'), p=torch.zeros(B, device='cuda'),
                    ids=torch.zeros(B, dtype=torch.int64, device='cuda'), w=torch.zeros(B, device='cuda'),
                    beta=torch.tensor([0.4], dtype=torch.float64, device='cuda'), minp=torch.zeros(528, device='cuda'),
                    u=torch.zeros(B, dtype=torch.float64, device='cuda'), z=torch.zeros(1001, device='cuda'),
                    target=target0.clone(), grad=torch.ones(777, device='cuda'))

    a = outs()      # the two launches
    nat.step_prologue((a['target'], source, 0.005), a['grad'], 99, step, a['u'], a['z'])
    nat.sumtree_sample(t.tree, C, B, a['u'], slot_ids, a['beta'], 0.001, a['leaf'], a['p'], a['ids'], a['w'], a['minp'])
    b = outs()      # one launch
    nat.step_prologue_sample((b['target'], source, 0.005), b['grad'], 99, step, b['u'], b['z'], None, 0, t.tree, C, B,
                             slot_ids, b['beta'], 0.001, b['leaf'], b['p'], b['ids'], b['w'], b['minp'])
    for k in ('u', 'z', 'target', 'grad', 'leaf', 'p', 'ids', 'w', 'beta'):
        assert torch.equal(a[k], b[k]), k
    assert float(a['minp'][0]) == float(b['minp'][0])
    c = outs()      # one launch, weights deferred; then the weights
    nat.step_prologue_sample((c['target'], source, 0.005), c['grad'], 99, step, c['u'], c['z'], None, 0, t.tree, C, B,
                             slot_ids, c['beta'], 0.001, c['leaf'], c['p'], c['ids'], None, c['minp'])
    assert float(c['beta']) == 0.4 and not c['w'].any()
    assert float(c['minp'][0]) == float(a['minp'][0])
    assert float(c['minp'][1]) == float(np.float32(a['minp'][0].item()) / np.float32(t.tree[0].item()))
    nat.per_is_weights(c['p'], B, t.tree, c['minp'][1:2], c['beta'], 0.001, c['w'])
    for k in ('leaf', 'p', 'ids', 'w', 'beta'):
        assert torch.equal(a[k], c[k]), k
    # the sampler workgroups' exchange (minp[2..7]: partial minima, arrivals, departures) is left ready for the next launch:
    # a second launch on the same buffers — and replays of it inside a hipGraph — draw the same batch again
    assert not b['minp'][6:8].view(torch.int32).any() and not c['minp'][6:8].view(torch.int32).any()
    first = {k: b[k].clone() for k in ('leaf', 'p', 'ids', 'w')}
    beta1 = float(b['beta'])
    nat.step_prologue_sample(None, b['grad'], 99, step, b['u'], b['z'], None, 0, t.tree, C, B,
                             slot_ids, b['beta'], 0.0, b['leaf'], b['p'], b['ids'], b['w'], b['minp'])
    stream = torch.cuda.Stream()
    stream.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(stream):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            nat.step_prologue_sample(None, b['grad'], 99, step, b['u'], b['z'], None, 0, t.tree, C, B,
                                     slot_ids, b['beta'], 0.0, b['leaf'], b['p'], b['ids'], b['w'], b['minp'])
        for _ in range(5):
            for k in ('leaf', 'p', 'ids', 'w'):
                b[k].zero_()
            g.replay()
            torch.cuda.synchronize()
            assert float(b['beta']) == beta1
            # (weights: beta has advanced once since `first` was drawn, so only the draw itself is compared bit for bit)
            for k in ('leaf', 'p', 'ids'):
                assert torch.equal(first[k], b[k]), k
            assert not b['minp'][6:8].view(torch.int32).any()


@pytest.mark.parametrize('B', [257, 512, 1000, 1024])
def test_partial_sampler_plus_weights_in_the_gather_launch(nat, B):
    """`asac_step_prologue_sample_partial` + `asac_window_gather_pad_w` == `asac_step_prologue_sample` +
    `asac_window_gather_pad`, bit for bit (leaves, priorities, ids, IS weights, beta, min p, gathered rows), launch after
    launch on a changing tree, eagerly and as hipGraph replays."""
    C, prev_n, post_n = 8192, 2, 3
    L = prev_n + 1 + post_n
    rng = np.random.default_rng(B)
    t = DevTree(nat, C, extra=2 * C)
    t.set_priorities(np.arange(C), (rng.random(C) + 0.01).astype(np.float32))
    slot_ids = torch.arange(C, dtype=torch.int64, device='cuda') + 7 * C
    step = torch.full((1,), 3, dtype=torch.int64, device='cuda')
    g = torch.Generator().manual_seed(2)
    wide = torch.randn(C, 64, generator=g).cuda()
    index = (torch.arange(C, dtype=torch.int32) % 37).cuda()
    source, target0 = torch.randn(30_000, generator=g).cuda(), torch.randn(30_000, generator=g).cuda()

    def outs():
        o = dict(leaf=torch.zeros(B, dtype=torch.int32, device='cuda'), p=torch.zeros(B, device='cuda'),
                 ids=torch.zeros(B, dtype=torch.int64, device='cuda'), w=torch.zeros(B, device='cuda'),
                 beta=torch.tensor([0.4], dtype=torch.float64, device='cuda'), minp=torch.zeros(528, device='cuda'),
                 u=torch.zeros(B, dtype=torch.float64, device='cuda'), z=torch.zeros(999, device='cuda'),
                 target=target0.clone(), grad=torch.ones(555, device='cuda'),
                 wide=torch.zeros(B, L, 64, device='cuda'), index=torch.zeros(B, L, dtype=torch.int32, device='cuda'),
                 mask=torch.zeros(B, L, dtype=torch.bool, device='cuda'))
        o['keys'] = nat.make_gather_keys([
            dict(src=wide, dst=o['wide'], row_bytes=256, pad_mode=nat.PAD_WORD, pad_word=0),
            dict(src=index, dst=o['index'], row_bytes=4, pad_mode=nat.PAD_WORD, pad_word=0xffffffff),
            dict(src=None, dst=o['mask'], pad_mode=nat.PAD_EMIT_MASK)])
        return o

    def two(o):
        nat.step_prologue_sample((o['target'], source, 0.005), o['grad'], 99, step, o['u'], o['z'], None, 0, t.tree, C, B,
                                 slot_ids, o['beta'], 0.001, o['leaf'], o['p'], o['ids'], o['w'], o['minp'])
        nat.window_gather_pad(o['keys'], o['ids'], B, prev_n, post_n, C, index)

    def deferred(o):
        nat.step_prologue_sample_partial((o['target'], source, 0.005), o['grad'], 99, step, o['u'], o['z'], None, 0, t.tree, C,
                                         B, slot_ids, o['leaf'], o['p'], o['ids'], o['minp'])
        nat.window_gather_pad_w(o['keys'], o['ids'], B, prev_n, post_n, C, index, o['p'], t.tree, o['beta'], 0.001, o['w'],
                                o['minp'])

    names = ('u', 'z', 'target', 'grad', 'leaf', 'p', 'ids', 'w', 'beta', 'wide', 'index', 'mask')
    a, b = outs(), outs()
    stream = torch.cuda.Stream()
    graph = None
    for it in range(6):
        if it == 3:
            stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(stream):
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=stream):
                    deferred(b)
            torch.cuda.synchronize()
        two(a)
        if graph is None:
            deferred(b)
        else:
            graph.replay()
        torch.cuda.synchronize()
        for k in names:
            assert torch.equal(a[k], b[k]), (it, k)
        assert float(a['minp'][0]) == float(b['minp'][0])
        idx = rng.permutation(C)[:500]
        t.set_priorities(idx, (rng.random(500) * 3).astype(np.float32))
        step += 1


def test_window_aux_matches_get_bnx_data_concatenations(nat):
    """`asac_window_aux` == the three concatenations of SAC_Base.get_bnx_data on window views."""
    import asac_amd  # noqa: F401
    from algorithm.utils.operators import gen_n_pre_actions
    g = torch.Generator().manual_seed(0)
    B, L, A = 37, 9, 3
    index = torch.randint(-1, 50, (B, L), generator=g, dtype=torch.int32).cuda()
    index[3, :] = -1
    pad = (torch.rand(B, L, generator=g) < 0.3).cuda()
    action = torch.randn(B, L, A, generator=g).cuda()
    bn_i, bn_p, bn_a = index[:, :-1], pad[:, :-1], action[:, :-1]
    want_i = torch.concat([bn_i, bn_i[:, -1:] + (bn_i[:, -1:] != -1)], dim=1)
    want_p = torch.concat([bn_p, bn_p[:, -1:]], dim=1)
    want_a = gen_n_pre_actions(bn_a, keep_last_action=True)
    got_i = torch.empty(B, L, dtype=torch.int32, device='cuda')
    got_p = torch.empty(B, L, dtype=torch.bool, device='cuda')
    got_a = torch.empty(B, L, A, device='cuda')
    nat.window_aux(bn_i, bn_p, bn_a, got_i, got_p, got_a)
    assert torch.equal(got_i, want_i.to(torch.int32)) and torch.equal(got_p, want_p) and torch.equal(got_a, want_a)
    # ... and with the previous actions as a column block of a wider tensor (the block beside the vector observations)
    joint = torch.full((B, L, 5 + A), 9.0, device='cuda')
    nat.window_aux(bn_i, bn_p, bn_a, got_i, got_p, joint[..., 5:])
    assert torch.equal(joint[..., 5:], want_a) and bool((joint[..., :5] == 9.0).all())


def test_replay_window_lands_beside_the_previous_actions(nat):
    """A static batch whose vector observations are column blocks of one [B, L, widths + A] tensor
    (`join_vector_obs_with_pre_action`: the gather's `dst_row_pitch`) holds the same values as the dense batch, and the
    concatenation [obs..., pre_action] of its blocks is a view (`adjacent_cat.joined_view`)."""
    import asac_amd  # noqa: F401
    from algorithm.adjacent_cat import AdjacentCat, joined_view
    from algorithm.replay_buffer import PrioritizedReplayBuffer
    gen = np.random.default_rng(0)
    n = 600

    data = dict(index=(np.arange(n) % 37).astype(np.int32), obs_vec=gen.standard_normal((n, 6)).astype(np.float32),
                obs_img=gen.integers(0, 255, (n, 4, 4, 3)).astype(np.uint8),
                obs_flag=gen.integers(0, 2, (n, 5)).astype(bool),
                action=gen.standard_normal((n, 3)).astype(np.float32), reward=gen.standard_normal(n).astype(np.float32),
                done=np.zeros(n, bool), last_mask=np.zeros(n, bool), mu_prob=np.ones((n, 3), np.float32),
                pre_seq_hidden_state=gen.standard_normal((n, 2)).astype(np.float32))
    joined = PrioritizedReplayBuffer(batch_size=32, capacity=1024, sample_prev_n=3, sample_post_n=4, device='cuda')
    joined.set_window_padding(torch.zeros(3))
    joined.add(data)
    _, batch_d, _ = joined.sample()                      # dense tensors per key
    batch_d = {k: v.clone() for k, v in batch_d.items()}
    joined.join_vector_obs_with_pre_action(3)
    joined._build_batch()
    joined.sample_into_static(sampled=True)              # the same ids gathered into the joint layout
    batch_j = joined._batch
    for k in batch_d:
        assert torch.equal(batch_d[k], batch_j[k]), k
    vec, flag, pre = batch_j['obs_vec'], batch_j['obs_flag'], joined.joint_pre_action
    assert pre.shape == (32, 8, 3) and flag.dtype == torch.float32 and not vec.is_contiguous()
    # the derived keys of the same launch == `asac_window_aux` on the gathered window (SAC_Base.get_bnx_data)
    d = joined.derived
    assert d is not None and d['pre_action'].data_ptr() == pre.data_ptr()
    want_i = torch.empty(32, 8, dtype=torch.int32, device='cuda')
    want_p = torch.empty(32, 8, dtype=torch.bool, device='cuda')
    want_a = torch.empty(32, 8, 3, device='cuda')
    nat.window_aux(batch_j['index'][:, :-1], batch_j['padding_mask'][:, :-1], batch_j['action'][:, :-1], want_i, want_p, want_a)
    assert bool(batch_j['padding_mask'].any()) and bool((batch_j['index'][:, -2] == -1).any())   # (padded rows exist)
    assert torch.equal(d['index_x'], want_i) and torch.equal(d['padding_mask_x'], want_p) and torch.equal(pre, want_a)
    pre.copy_(torch.randn(32, 8, 3, device='cuda'))
    want = torch.cat([vec, flag, pre], dim=-1)
    view = joined_view([vec, flag, pre], -1)
    assert view is not None and view.data_ptr() == vec.data_ptr() and torch.equal(view, want)
    with AdjacentCat():
        got = torch.cat([vec, flag, pre], dim=-1)
        other = torch.cat([flag, vec], dim=-1)           # not adjacent in that order: ATen's copy
        part = torch.concat((flag, pre), -1)
    assert got.data_ptr() == vec.data_ptr() and torch.equal(other, torch.cat([flag.clone(), vec.clone()], -1))
    assert part.data_ptr() == flag.data_ptr() and torch.equal(part, want[..., 6:])
    assert joined_view([batch_d['obs_vec'], batch_d['action']], -1) is None


def test_gelu_against_torch(nat):
    """The one function of the library that is an approximation by design: GELU of the fused MLP / convolution kernels
    (Abramowitz-Stegun 7.1.26 erf, `__expf`, `v_rcp_f32`; csrc/asac_gelu.h) against `torch.nn.functional.gelu` and its
    derivative, evaluated in float64, on [-12, 12] (2^20 points + the neighbourhood of 0).  The bound asserted here is
    the one include/asac_hip.h states."""
    from tests import parity_utils as pu
    z = torch.cat([torch.linspace(-12, 12, 2 ** 20), torch.linspace(-1e-3, 1e-3, 4097), torch.randn(2 ** 16) * 3]).cuda()
    value, deriv = torch.empty_like(z), torch.empty_like(z)
    nat.gelu_eval(z, value, deriv)
    zd = z.double().requires_grad_()
    want = torch.nn.functional.gelu(zd)
    want_d, = torch.autograd.grad(want.sum(), zd)
    pu.check('kernels/gelu/value', value, want.detach(), rtol=3e-7, atol=6e-7)
    pu.check('kernels/gelu/derivative', deriv, want_d, rtol=0., atol=6e-7)
    # ... and ATen's own f32 evaluation for scale: how far IT is from the float64 value
    pu.check('kernels/gelu/aten_f32_value_for_scale', torch.nn.functional.gelu(z), want.detach(), rtol=3e-7, atol=6e-7, enforce=False)


def test_curiosity_bonus_and_masked_mse_against_the_aten_chains(nat):
    """`asac_curiosity_bonus` / `asac_masked_mse` == the elementwise chains of SAC_Base._get_y (reference
    sac_base.py:1333-1343) and _train_curiosity (1951-1976) on strided window views; f32 rounding of a K-term sum
    (rtol 1e-6) — the chains' own summation order is ATen's, not the reference's to pin."""
    g = torch.Generator().manual_seed(4)
    B, L, K = 70, 9, 8
    states = torch.randn(B, L, K, generator=g).cuda()
    rewards = torch.randn(B, L, generator=g).cuda()
    approx = torch.randn(B, L - 1, K, generator=g).cuda()
    nxt, r_view = states[:, 1:], rewards[:, :-1]
    want_r = rewards.clone()
    d = approx - nxt
    want_r[:, :-1].add_(torch.sum(d * d, dim=-1).mul_(0.5), alpha=0.37)
    nat.curiosity_bonus(approx, nxt, r_view, 0.37)
    torch.testing.assert_close(rewards, want_r, rtol=1e-6, atol=1e-6)
    assert torch.equal(rewards[:, -1], want_r[:, -1])

    mask = (torch.rand(B, L - 1, generator=g) < 0.3).cuda()
    dm = (approx - nxt) * (~mask).unsqueeze(-1)
    want_loss = (dm * dm).double().sum() / dm.numel()
    grad, loss = torch.empty_like(approx), torch.zeros((), device='cuda')
    nat.masked_mse(approx, nxt, mask, grad, loss)
    torch.testing.assert_close(grad, dm * (2. / dm.numel()), rtol=1e-6, atol=1e-9)
    assert abs(float(loss) - float(want_loss)) <= 2e-6 * float(want_loss)
    nat.masked_mse(approx, nxt, None, grad, loss)
    torch.testing.assert_close(grad, (approx - nxt) * (2. / dm.numel()), rtol=1e-6, atol=1e-9)


def test_normal_nll_kl_against_torch_distributions(nat):
    """`asac_normal_nll_kl` == the transition head's loss of SAC_Base._train_rpm written with torch.distributions
    (reference sac_base.py:1798-1816), its entropy statistic and autograd's gradients w.r.t. loc / scale, on strided
    views (the two halves of a model output, a slice of a window); f32 rounding of N-term sums (rtol 2e-6 on the loss,
    1e-5 on gradient entries)."""
    from torch import distributions as D
    g = torch.Generator().manual_seed(9)
    B, L, S, w = 64, 4, 8, 0.37
    raw = torch.randn(B, L - 1, 2 * S, generator=g).cuda().requires_grad_(True)
    window = torch.randn(B, L, S, generator=g).cuda()
    loc, logstd = torch.chunk(raw, 2, dim=-1)
    scale = torch.clamp(torch.exp(logstd), 0.1, 1.0)
    target = window[:, 1:]
    dist = D.Normal(loc, scale, validate_args=False)
    std = D.Normal(torch.zeros_like(loc), torch.ones_like(scale), validate_args=False)
    want = -torch.mean(dist.log_prob(target)) + w * torch.mean(D.kl.kl_divergence(dist, std))
    want_loc, want_scale = torch.autograd.grad(want, [loc, scale])
    g_loc, g_scale = torch.empty(B, L - 1, S, device='cuda'), torch.empty(B, L - 1, S, device='cuda')
    out = torch.empty(2, device='cuda')
    nat.normal_nll_kl(loc.detach(), scale.detach(), target, w, g_loc, g_scale, out)
    torch.testing.assert_close(out[0], want.detach(), rtol=2e-6, atol=1e-6)
    torch.testing.assert_close(out[1], dist.entropy().mean().detach(), rtol=2e-6, atol=1e-6)
    torch.testing.assert_close(g_loc, want_loc, rtol=1e-5, atol=1e-9)
    torch.testing.assert_close(g_scale, want_scale, rtol=1e-5, atol=1e-9)
    nat.normal_nll_kl(loc.detach(), scale.detach(), target, w, g_loc, g_scale, out)      # the workspace is clean again
    torch.testing.assert_close(out[0], want.detach(), rtol=2e-6, atol=1e-6)


def test_mse_mean_grad_and_the_intercepted_mse_loss(nat):
    """`asac_mse_mean_grad` against torch's mse_loss (value 1e-6, gradient bit-level formula (a - b) * 2 / N), target a
    strided slice of a window batch, run-to-run identical, replayable; `fused.fused_mse_loss` routes exactly the calls it
    describes and leaves the function as it found it."""
    from algorithm.fused import fused_mse_loss
    g = torch.Generator().manual_seed(4)
    B, L, b = 128, 9, 5
    frames = torch.randn(B, L, 3, 30, 30, generator=g).cuda()
    target = frames[:, b:]                                   # [B, 4, 3, 30, 30], batch stride L * 2700
    pred = torch.randn(B, L - b, 3, 30, 30, generator=g).cuda().requires_grad_()
    assert pred.numel() >= 1 << 20
    ws = torch.zeros(nat.mse_mean_grad_workspace(), device='cuda')
    orig = torch.nn.functional.mse_loss
    with fused_mse_loss(ws):
        with nat.LaunchProfiler(repeat=1) as prof:
            loss = torch.nn.functional.mse_loss(pred, target)
        assert prof.summary()['asac_mse_mean_grad']['calls'] == 1
        small = torch.nn.functional.mse_loss(pred[:2], target[:2])                       # below the threshold: ATen
        summed = torch.nn.functional.mse_loss(pred, target, reduction='sum')             # other arguments: ATen
    assert torch.nn.functional.mse_loss is orig
    (g_fused,) = torch.autograd.grad(loss * 3.0, pred)
    want = orig(pred, target)
    (g_want,) = torch.autograd.grad(want * 3.0, pred)
    np.testing.assert_allclose(loss.item(), want.item(), rtol=2e-6)
    np.testing.assert_allclose(g_fused.cpu().numpy(), g_want.cpu().numpy(), rtol=2e-6, atol=0)
    np.testing.assert_allclose(small.item(), orig(pred[:2], target[:2]).item(), rtol=1e-6)
    np.testing.assert_allclose(summed.item(), orig(pred, target, reduction='sum').item(), rtol=1e-5)
    assert not ws[-1:].view(torch.int32).any()
    # the raw entry point: identical results launch after launch, also as hipGraph replays
    p3, t3 = pred.detach().view(B, L - b, -1), target.view(B, L - b, -1)
    grad, out = torch.empty_like(p3), torch.zeros((), device='cuda')
    nat.mse_mean_grad(p3, t3, grad, out, ws)
    first = (out.clone(), grad.clone())
    stream = torch.cuda.Stream()
    stream.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(stream):
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=stream):
            nat.mse_mean_grad(p3, t3, grad, out, ws)
        for _ in range(3):
            grad.zero_()
            out.zero_()
            graph.replay()
            torch.cuda.synchronize()
            assert torch.equal(out, first[0]) and torch.equal(grad, first[1])
    assert float(first[0]) == float(loss)
