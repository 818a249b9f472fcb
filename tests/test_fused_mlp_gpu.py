"""GPU: the fused stock-MLP kernels (`asac_mlp_forward/backward`, `asac_gauss_head_*`) against the
very `nn.Module`s they replace (eager PyTorch-ROCm forward + autograd), f32 tolerance."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(E, state, A, policy=False):
    import asac_amd  # noqa: F401
    import algorithm.nn_models as m
    from algorithm.fused import FlatParamGroup
    from algorithm.fused_mlp import StockMLP, describe_policy, describe_q
    torch.manual_seed(0)
    if policy:
        mods = [m.ModelPolicy(state, [], A).cuda()]
        desc = describe_policy(mods[0])
    else:
        mods = [m.ModelQ(state, [], A, False).cuda() for _ in range(E)]
        desc = describe_q(mods[0])
    assert desc is not None
    with torch.no_grad():      # non-trivial biases
        for mod in mods:
            for p in mod.parameters():
                if p.dim() == 1:
                    p.normal_(0, 0.3)
    group = FlatParamGroup([(f'm{i}', list(mod.parameters())) for i, mod in enumerate(mods)], 'cuda')
    stride = group.segments['m0'][1] - group.segments['m0'][0]
    mlp = StockMLP(desc, group.flat, group.grad, 0, stride, len(mods), torch.device('cuda'))
    return mods, group, mlp


# (S = 64: critics on a 64-wide state + the action — a first layer of 66 inputs, the wide instantiations of the kernels,
#  the reference environments' `m.GRU(..., 64, 1)` states)
@pytest.mark.parametrize('N,S', [(256, 6), (1280, 6), (100, 6), (33, 6), (12345, 6), (256, 64), (1000, 64), (33, 126)])
def test_q_ensemble_forward_backward(N, S):     # 12345 rows: workgroups loop over row tiles
    E, A = 3, 2
    mods, group, mlp = _setup(E, S, A)
    x = torch.randn(N, S, device='cuda', requires_grad=True)
    a = torch.randn(N, A, device='cuda').tanh().requires_grad_()
    gout = torch.randn(E, N, 1, device='cuda')
    # reference: the modules
    ref = torch.stack([q(x, a, None)[1] for q in mods])
    (ref * gout).sum().backward()
    ref_gx, ref_ga = x.grad.clone(), a.grad.clone()
    ref_gp = group.grad.clone()
    x.grad = a.grad = None
    group.grad.zero_()
    out = mlp(x, a)
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().cpu().numpy(), rtol=2e-5, atol=2e-6 if S <= 16 else 6e-6)
    (out * gout).sum().backward()
    np.testing.assert_allclose(x.grad.cpu().numpy(), ref_gx.cpu().numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(a.grad.cpu().numpy(), ref_ga.cpu().numpy(), rtol=1e-4, atol=1e-5)
    # parameter gradients are sums over all N rows: the f32 summation order differs (tiles vs GEMM), so the
    # absolute tolerance scales with the gradient magnitude
    gp_scale = max(1.0, float(ref_gp.abs().max()))
    np.testing.assert_allclose(group.grad.cpu().numpy(), ref_gp.cpu().numpy(), rtol=1e-4, atol=2e-5 * gp_scale)
    # input gradients only (policy update through the critics): parameter grads stay untouched
    group.grad.zero_()
    x.grad = a.grad = None
    out = mlp(x.detach(), a, param_grads=False)
    (out * gout).sum().backward()
    assert torch.count_nonzero(group.grad) == 0
    np.testing.assert_allclose(a.grad.cpu().numpy(), ref_ga.cpu().numpy(), rtol=1e-4, atol=1e-5)
    # per-member inputs [E, N, .] and strided rows
    xs = torch.randn(N, 5, S, device='cuda')[:, 2]
    a3 = torch.randn(E, N, A, device='cuda')
    out = mlp(xs, a3, param_grads=False)
    ref = torch.stack([q(xs, a3[i], None)[1] for i, q in enumerate(mods)])
    np.testing.assert_allclose(out.cpu().numpy(), ref.detach().cpu().numpy(), rtol=2e-5, atol=2e-6 if S <= 16 else 6e-6)


@pytest.mark.parametrize('N,weighted', [(256, True), (100, False), (33, True)])
def test_q_loss_backward_and_adam_from_partials(N, weighted):
    """`asac_mlp_backward_qloss` (+ `asac_adam_step_partials`) against the chain it folds: module forward,
    clipped double-Q loss (autograd), backward, torch.optim.Adam."""
    from asac_amd import native
    from algorithm.fused import FlatAdam
    E, S, A, clip = 3, 6, 2, 0.2
    mods, group, mlp = _setup(E, S, A)
    x, a = torch.randn(N, S, device='cuda'), torch.randn(N, A, device='cuda').tanh()
    tq = torch.randn(E, N, device='cuda') * 0.3
    y = torch.randn(N, device='cuda') * 0.3
    w = torch.rand(N, device='cuda') + 0.5 if weighted else None
    # reference
    q = torch.stack([m(x, a, None)[1] for m in mods]).squeeze(-1)
    clipped = tq + torch.clamp(q - tq, -clip, clip)
    l = torch.maximum((clipped - y) ** 2, (q - y) ** 2)
    if w is not None:
        l = l * w
    loss_e = l.mean(dim=1)
    loss_e.sum().backward()
    ref_gp, ref_loss = group.grad.clone(), loss_e.detach().clone()
    # fused: reduce in the backward launch
    group.grad.zero_()
    loss = torch.zeros(E, device='cuda')
    mlp.backward_qloss(x, a, tq, y, w, clip, loss)
    np.testing.assert_allclose(loss.cpu().numpy(), ref_loss.cpu().numpy(), rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(group.grad.cpu().numpy(), ref_gp.cpu().numpy(), rtol=2e-4, atol=2e-6)
    # deferred reduction folded into Adam == reduce + separate Adam launch
    flat0 = group.flat.clone()
    steps = torch.zeros(1, dtype=torch.int64, device='cuda')
    m, v = torch.zeros_like(group.flat), torch.zeros_like(group.flat)
    opt = FlatAdam(group, [f'm{i}' for i in range(E)], 1e-3, steps, m, v)
    opt.step()
    want, want_m = group.flat.clone(), m.clone()
    group.flat.copy_(flat0)
    m.zero_(); v.zero_(); group.grad.zero_(); loss.zero_()
    mlp.accumulate = False
    mlp.backward_qloss(x, a, tq, y, w, clip, loss, defer=True)
    assert torch.count_nonzero(group.grad) == 0
    mlp.adam_partials(opt, loss_out=loss)
    np.testing.assert_allclose(loss.cpu().numpy(), ref_loss.cpu().numpy(), rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(group.grad.cpu().numpy(), ref_gp.cpu().numpy(), rtol=2e-4, atol=2e-6)
    assert torch.equal(group.flat, want) and torch.equal(m, want_m)


@pytest.mark.parametrize('B,n,E,Es,use_is', [(256, 4, 2, 2, True), (100, 1, 3, 2, True), (33, 7, 2, 2, False),
                                              (1024, 3, 4, 2, True), (256, 16, 2, 2, True),
                                              (256, 40, 2, 2, True),      # BASELINE configs[2]: n_step 40 on 16-wave tiles
                                              (300, 64, 2, 2, False), (6000, 16, 2, 2, True)])
def test_q_loss_backward_forming_its_own_return_is_the_chain_bit_for_bit(B, n, E, Es, use_is):
    """`asac_mlp_backward_qloss_return` == `asac_vtrace_return_min` followed by `asac_mlp_backward_qloss`: the return
    target y, the per-member losses and every parameter gradient, bit for bit (reduced in the launch and deferred)."""
    from asac_amd import native as nat
    S, A, clip = 6, 2, 0.2
    mods, group, mlp = _setup(E, S, A)
    f = dict(device='cuda')
    torch.manual_seed(B + n)
    x, a = torch.randn(B, S, **f), torch.randn(B, A, **f).tanh()
    tq = torch.randn(E, B, **f) * 0.3
    w = torch.rand(B, **f) + 0.5
    q = torch.randn(E, B, n + 1, **f)
    logp, log_alpha = torch.randn(B, n + 1, **f), torch.tensor([-1.2], **f)
    reward, done = torch.randn(B, n, **f), torch.rand(B, n, **f) < 0.1
    last, pad = torch.rand(B, n, **f) < 0.05, torch.rand(B, n, **f) < 0.1
    mu, pi = torch.rand(B, n, A, **f) + 0.05, torch.rand(B, n + 1, A, **f) + 0.05
    gr, lr = torch.logspace(0, n - 1, n, 0.99).cuda(), torch.logspace(0, n - 1, n, 0.95).cuda()
    sub_n = torch.randperm(E, **f)[:Es].to(torch.int32) if Es != E else None
    sub_next = torch.randperm(E, **f)[:Es].to(torch.int32) if Es != E else None

    def ret_args(y):
        r = nat.VtraceArgs()
        r.q = q.data_ptr()
        r.q_stride_e, r.q_stride_b, r.q_stride_t = q.stride(0), q.stride(1), q.stride(2)
        r.logp, r.log_alpha, r.E_sample = logp.data_ptr(), log_alpha.data_ptr(), Es
        r.subset_n = sub_n.data_ptr() if sub_n is not None else None
        r.subset_next = sub_next.data_ptr() if sub_next is not None else None
        r.reward, r.reward_stride = reward.data_ptr(), reward.stride(0)
        r.done, r.last_mask, r.padding_mask, r.mask_stride = done.data_ptr(), last.data_ptr(), pad.data_ptr(), done.stride(0)
        if use_is:
            r.mu_prob, r.mu_stride_b, r.mu_stride_t, r.mu_offset = mu.data_ptr(), mu.stride(0), mu.stride(1), 0
            r.pi_prob, r.pi_stride_b, r.pi_stride_t, r.A = pi.data_ptr(), pi.stride(0), pi.stride(1), A
        r.gamma_ratio, r.lambda_ratio = gr.data_ptr(), lr.data_ptr()
        r.gamma, r.v_rho, r.v_c, r.use_n_step_is, r.B, r.n = 0.99, 1.0, 1.0, int(use_is), B, n
        r.y_out = y.data_ptr()
        return r

    def run(fused, defer):
        group.grad.zero_()
        y, loss = torch.zeros(B, **f), torch.zeros(E, **f)
        r = ret_args(y)
        if fused:
            assert mlp.backward_qloss_return_ok(B, r)
            g0 = mlp.backward_qloss_return(x, a, tq, r, w, clip, loss, defer=defer, state_grads=not defer)
        else:
            nat.vtrace_return_min(r)
            g0 = mlp.backward_qloss(x, a, tq, y, w, clip, loss, defer=defer, state_grads=not defer)
        partials = mlp._workspace_for(B).clone() if defer else None
        return y, loss.clone(), group.grad.clone(), partials, g0

    mlp.accumulate = False
    for defer in (False, True):
        want, got = run(False, defer), run(True, defer)
        for name, w_, g_ in zip(('y', 'loss', 'grads', 'partials', 'state grads'), want, got):
            if w_ is not None:
                assert torch.equal(w_, g_), (name, defer)
    r = ret_args(torch.zeros(B, **f))
    r.n = 65            # (16-row tiles run 16 waves: one thread per (row, step) up to n = 64; 32-row tiles 512 threads: n <= 16)
    assert not mlp.backward_qloss_return_ok(B, r), 'more steps per tile than threads: the two launches'
    if B == 6000:
        r.n = 17
        assert not mlp.backward_qloss_return_ok(B, r)


@pytest.mark.parametrize('N', [256, 45])
def test_policy_step_fused_backwards_match_the_kernel_chain(N):
    """`asac_mlp_backward_policy_q` / `_policy_sample` against the launches they fold (objective kernel +
    Q backward; sampling backward + policy backward), which are themselves checked against autograd."""
    from asac_amd import native
    E, S, A = 3, 6, 2
    _, qgroup, fq = _setup(E, S, A)
    _, pgroup, fpi = _setup(1, S, A, policy=True)
    x = torch.randn(N, S, device='cuda')
    eps = torch.randn(N, A, device='cuda')
    log_alpha = torch.tensor(-0.7, device='cuda')
    ls = fpi._launch_forward(x, None)[0]
    loc, scale = ls[:, :A], ls[:, A:]
    a, logp = torch.empty(N, A, device='cuda'), torch.empty(N, device='cuda')
    native.squash_sample_fwd(loc, scale, eps, a, logp)
    q = fq._launch_forward(x, a)
    for subset, Es in ((None, E), (torch.tensor([2, 0], dtype=torch.int32, device='cuda'), 2)):
        # chain: objective kernel -> Q backward (action grads) -> sampling backward -> policy backward
        g_logp, g_q = torch.empty(N, device='cuda'), torch.empty(E, N, device='cuda')
        loss, ent = torch.zeros((), device='cuda'), torch.zeros((), device='cuda')
        native.policy_loss_fwd_bwd(logp, q.view(E, N), subset, Es, log_alpha, scale, loss, g_logp, g_q, ent)
        _, ga_ref = fq._launch_backward(x, a, g_q.view(E, N, 1), False, True, False, reduce_members=False)
        g_ls = torch.empty(N, 2 * A, device='cuda')
        native.squash_sample_bwd(loc, scale, eps, ga_ref, g_logp, g_ls[:, :A], g_ls[:, A:])
        pgroup.grad.zero_()
        fpi._launch_backward(x, None, g_ls.view(1, N, 2 * A), False, False, True)
        want = pgroup.grad.clone()
        # fused
        ga = fq.backward_policy_q(x, a, q.view(E, N), subset, Es)
        np.testing.assert_allclose(ga.cpu().numpy(), ga_ref.cpu().numpy(), rtol=1e-5, atol=1e-8)
        pgroup.grad.zero_()
        fpi.backward_policy_sample(x, eps, ga, log_alpha)
        np.testing.assert_allclose(pgroup.grad.cpu().numpy(), want.cpu().numpy(), rtol=2e-4, atol=1e-7)


@pytest.mark.parametrize('N,S,A', [(256, 6, 2), (45, 6, 2), (1024, 8, 4), (4096, 8, 4), (16, 64 - 3, 3)])
def test_policy_step_in_one_launch_is_the_chain_bit_for_bit(N, S, A):
    """`asac_policy_step_fused` (two critics) against critics forward + `asac_mlp_backward_policy_q` +
    `asac_mlp_backward_policy_sample`: the value table, the reduced policy gradient and the per-tile partials the
    Adam launch sums must be IDENTICAL (same MFMA chains, same summation orders)."""
    from asac_amd import native
    E = 2
    _, qgroup, fq = _setup(E, S, A)
    _, pgroup, fpi = _setup(1, S, A, policy=True)
    win = torch.randn(N, 3, S, device='cuda')
    x = win[:, 1]                                    # strided rows, like states[:, b]
    eps = torch.randn(N, A, device='cuda')
    log_alpha = torch.tensor(-0.7, device='cuda')
    ls = fpi._launch_forward(x, None)[0]
    a, logp = torch.empty(N, A, device='cuda'), torch.empty(N, device='cuda')
    native.squash_sample_fwd(ls[:, :A], ls[:, A:], eps, a, logp)
    assert fpi.policy_step_fused_ok(fq, N)
    # chain
    q = fq._launch_forward(x, a)
    ga = fq.backward_policy_q(x, a, q.view(E, N), None, E)
    pgroup.grad.zero_()
    fpi.backward_policy_sample(x, eps, ga, log_alpha)
    want = pgroup.grad.clone()
    fpi.backward_policy_sample(x, eps, ga, log_alpha, defer=True)
    tiles = native.mlp_backward_tiles(N, 1)
    used = native.mlp_param_extent(fpi.desc)
    want_partials = fpi._workspace[:tiles * fpi.member_stride].view(tiles, -1)[:, :used].clone()
    # one launch
    q_out = torch.zeros(E, N, 1, device='cuda')
    pgroup.grad.zero_()
    fpi._workspace.fill_(float('nan'))
    fpi.policy_step_fused(fq, x, a, eps, log_alpha, q_out=q_out)
    assert torch.equal(q_out, q), 'value table'
    assert torch.equal(pgroup.grad, want), 'reduced policy gradient'
    fpi._workspace.fill_(float('nan'))
    fpi.policy_step_fused(fq, x, a, eps, log_alpha, q_out=None, defer=True)
    got_partials = fpi._workspace[:tiles * fpi.member_stride].view(tiles, -1)[:, :used]
    assert torch.equal(got_partials, want_partials), 'per-tile partials'
    assert qgroup.grad.count_nonzero() == 0
    # the action sampled inside the launch == policy forward + sampling launch in front of it
    a_g, lp_g, ls_g = torch.zeros(N, A, device='cuda'), torch.zeros(N, device='cuda'), torch.zeros(N, 2 * A, device='cuda')
    pgroup.grad.zero_()
    q_out.zero_()
    fpi.policy_step_fused(fq, x, None, eps, log_alpha, q_out=q_out, sample_out=(a_g, lp_g, ls_g))
    assert torch.equal(a_g, a) and torch.equal(lp_g, logp) and torch.equal(ls_g, ls), 'sampled on chip'
    assert torch.equal(q_out, q) and torch.equal(pgroup.grad, want)
    assert not fpi.policy_step_fused_ok(fq, 4097)      # too many rows for 16-row tiles: the learner keeps the chain
    # an ensemble of four with two sampled: only the subset's members are evaluated, in the subset's order
    _, qgroup4, fq4 = _setup(4, S, A)
    for pair in ((2, 0), (1, 3)):
        sub = torch.tensor(pair, dtype=torch.int32, device='cuda')
        q4 = fq4._launch_forward(x, a)
        ga4 = fq4.backward_policy_q(x, a, q4.view(4, N), sub, 2)
        pgroup.grad.zero_()
        fpi.backward_policy_sample(x, eps, ga4, log_alpha)
        want4 = pgroup.grad.clone()
        q_out4 = torch.zeros(4, N, 1, device='cuda')
        pgroup.grad.zero_()
        fpi.policy_step_fused(fq4, x, a, eps, log_alpha, q_out=q_out4, subset=sub)
        assert torch.equal(pgroup.grad, want4), pair
        for e in pair:
            assert torch.equal(q_out4[e], q4[e])
        assert qgroup4.grad.count_nonzero() == 0


@pytest.mark.parametrize('B,T,S,A,E,window', [(256, 5, 6, 2, 2, False), (256, 5, 6, 2, 2, True), (37, 3, 8, 4, 3, True),
                                               (1024, 9, 8, 4, 2, False), (6000, 2, 6, 2, 2, False)])
def test_policy_sample_critics_forward_in_one_launch_is_the_chain_bit_for_bit(B, T, S, A, E, window):
    """`asac_policy_sample_q_forward` against policy forward -> `asac_squash_multi` -> critics forward (+ the extra
    plain forward job riding along): every output identical."""
    from asac_amd import native
    _, _, fq = _setup(E, S, A)
    _, _, fpi = _setup(1, S, A, policy=True)
    N = B * T
    if window:      # [B, T, S] view of a longer window (states[:, b:]), read in place
        base = torch.randn(B, T + 2, S, device='cuda')
        xs = native.WindowRows(base[:, 2:])
        x_flat = base[:, 2:].reshape(N, S)
    else:
        x_flat = torch.randn(N, S, device='cuda')
        xs = x_flat
    eps, eps2 = torch.randn(N, A, device='cuda'), torch.randn(B, A, device='cuda')
    stored = torch.rand(B, T, A + 1, device='cuda') * 1.9 - 0.95       # stored actions behind one discrete column
    x0, a0 = torch.randn(B, S, device='cuda'), torch.randn(B, A, device='cuda').tanh()
    t2 = T - 1
    # the chain
    ls = fpi._launch_forward(xs, None)[0].view(B, T, 2 * A)
    a_w, lp_w, pr_w = torch.empty(B, T, A, device='cuda'), torch.empty(B, T, device='cuda'), torch.zeros(B, T, A, device='cuda')
    a2_w, lp2_w = torch.empty(B, A, device='cuda'), torch.empty(B, device='cuda')
    native.squash_multi([native.squash_job(ls[..., :A], ls[..., A:], eps, a_w, lp_w, stored, 1, pr_w, 0),
                         native.squash_job(ls[:, t2, :A], ls[:, t2, A:], eps2, a2_w, lp2_w)])
    q_w = fq._launch_forward(xs, a_w.view(N, A))
    x_w = fq._launch_forward(x0, a0)
    # one launch
    job_pi, ls_g = fpi.job(xs, None)
    a_g, lp_g, pr_g = torch.empty(B, T, A, device='cuda'), torch.empty(B, T, device='cuda'), torch.zeros(B, T, A, device='cuda')
    a2_g, lp2_g = torch.empty(B, A, device='cuda'), torch.empty(B, device='cuda')
    job_q, q_g = fq.job(xs, a_g.view(N, A))
    job_x, x_g = fq.job(x0, a0)
    job = native.pi_q_job(job_pi, job_q, eps, a_g, lp_g, T, action=stored, action_offset=1, prob_out=pr_g,
                          eps2=eps2, t2=t2, a2_out=a2_g, logp2_out=lp2_g)
    assert native.policy_sample_q_forward_ok(job)
    native.policy_sample_q_forward(job, [job_x])
    for name, got, want in (('loc|scale', ls_g[0].view(B, T, 2 * A), ls), ('action', a_g, a_w), ('logp', lp_g, lp_w),
                            ('stored-action probabilities', pr_g, pr_w), ('second sample', a2_g, a2_w),
                            ('second logp', lp2_g, lp2_w), ('critics', q_g, q_w), ('extra job', x_g, x_w)):
        assert torch.equal(got, want), name
    # no stored actions / second sample; and a job that does not qualify
    job = native.pi_q_job(job_pi, job_q, eps, a_g.zero_(), lp_g.zero_(), T)
    q_g.zero_()
    native.policy_sample_q_forward(job)
    assert torch.equal(a_g, a_w) and torch.equal(lp_g, lp_w) and torch.equal(q_g, q_w)
    other = fq.job(torch.randn(N, S, device='cuda'), a_g.view(N, A))[0]
    assert not native.policy_sample_q_forward_ok(native.pi_q_job(job_pi, other, eps, a_g, lp_g, T))


@pytest.mark.parametrize('B,T,S,A,E,window', [(256, 5, 6, 2, 2, False), (256, 5, 6, 2, 2, True), (37, 3, 8, 4, 3, True),
                                               (1024, 9, 8, 4, 2, False), (6000, 2, 6, 2, 2, False), (256, 41, 8, 2, 2, True),
                                               (19, 4, 6, 8, 1, False)])
def test_policy_forward_with_sampling_epilogue_is_the_chain_bit_for_bit(B, T, S, A, E, window):
    """`asac_mlp_forward_multi_sampled` against `asac_mlp_forward_multi` -> `asac_squash_multi`: the policy's (loc | scale),
    the main sample and its log-probabilities, the stored actions' probabilities, the second sample at a window position —
    and a critic job riding beside it — identical bit for bit; every subset of the epilogue's three parts; 16- and 32-row
    tiles; window addressing; two policy jobs with an epilogue each."""
    from asac_amd import native
    _, _, fq = _setup(E, S, A)
    _, _, fpi = _setup(1, S, A, policy=True)
    N = B * T
    f = dict(device='cuda')
    torch.manual_seed(B * T + A)
    if window:
        base = torch.randn(B, T + 2, S, **f)
        xs = native.WindowRows(base[:, 2:])
    else:
        xs = torch.randn(N, S, **f)
    eps, eps2 = torch.randn(N, A, **f), torch.randn(B, A, **f)
    stored = torch.rand(B, T, A + 1, **f) * 1.9 - 0.95       # stored actions behind one discrete column
    x0, a0 = torch.randn(B, S, **f), torch.randn(B, A, **f).tanh()
    t2 = T - 1

    def outs():
        return dict(a=torch.zeros(B, T, A, **f), lp=torch.zeros(B, T, **f), pr=torch.zeros(B, T, A + 2, **f),
                    a2=torch.zeros(B, A, **f), lp2=torch.zeros(B, **f))

    for main, prob, second in ((True, True, True), (True, False, False), (False, True, False), (False, False, True),
                               (True, False, True), (False, True, True)):
        # the chain
        w = outs()
        job_x, x_w = fq.job(x0, a0)
        job_pi, ls_w = fpi.job(xs, None)
        native.mlp_forward_multi([job_x, job_pi])
        ls = ls_w[0].view(B, T, 2 * A)
        jobs = []
        if main or prob:
            jobs.append(native.squash_job(ls[..., :A], ls[..., A:], eps if main else None, w['a'] if main else None,
                                          w['lp'] if main else None, stored if prob else None, 1, w['pr'][..., 1:] if prob else None, 0))
        if second:
            jobs.append(native.squash_job(ls[:, t2, :A], ls[:, t2, A:], eps2, w['a2'], w['lp2']))
        native.squash_multi(jobs)
        # one launch
        g = outs()
        job_x2, x_g = fq.job(x0, a0)
        job_pi2, ls_g = fpi.job(xs, None)
        epi = native.sample_epilogue(job_pi2, eps if main else None, g['a'] if main else None, g['lp'] if main else None, T,
                                     action=stored if prob else None, action_offset=1, prob_out=g['pr'][..., 1:] if prob else None,
                                     eps2=eps2 if second else None, t2=t2, a2_out=g['a2'] if second else None,
                                     logp2_out=g['lp2'] if second else None)
        both = [job_x2, job_pi2], [native.sample_epilogue(), epi]
        assert native.mlp_forward_multi_sampled_ok(*both)
        with native.LaunchProfiler(repeat=1) as prof:
            native.mlp_forward_multi_sampled(*both)
        assert list(prof.summary()) == ['asac_mlp_forward_multi_sampled']
        assert torch.equal(ls_g[0], ls_w[0]) and torch.equal(x_g, x_w), (main, prob, second)
        for k in w:
            assert torch.equal(g[k], w[k]), (k, main, prob, second)
        if main:
            assert g['lp'].abs().sum() > 0 and torch.isfinite(g['lp']).all()
    # two policy jobs, an epilogue each (the TD target's own policy pass beside the window's)
    xs_b = torch.randn(B * 2, S, **f)
    eps_b = torch.randn(B * 2, A, **f)
    job_a, ls_a = fpi.job(xs, None)
    job_b, ls_b = fpi.job(xs_b, None)
    native.mlp_forward_multi([job_a, job_b])
    w, wb = outs(), (torch.zeros(B * 2, A, **f), torch.zeros(B * 2, **f))
    la, lb = ls_a[0].view(B, T, 2 * A), ls_b[0].view(B * 2, 2 * A)
    native.squash_multi([native.squash_job(la[..., :A], la[..., A:], action=stored, action_offset=1, prob_out=w['pr'][..., 1:]),
                         native.squash_job(la[:, t2, :A], la[:, t2, A:], eps2, w['a2'], w['lp2']),
                         native.squash_job(lb[..., :A], lb[..., A:], eps_b, *wb)])
    g, gb = outs(), (torch.zeros(B * 2, A, **f), torch.zeros(B * 2, **f))
    job_a, ls_a2 = fpi.job(xs, None)
    job_b, ls_b2 = fpi.job(xs_b, None)
    native.mlp_forward_multi_sampled(
        [job_a, job_b],
        [native.sample_epilogue(job_a, action=stored, action_offset=1, prob_out=g['pr'][..., 1:], eps2=eps2, t2=t2,
                                a2_out=g['a2'], logp2_out=g['lp2']),
         native.sample_epilogue(job_b, eps_b, *gb)])
    assert torch.equal(ls_a2[0], ls_a[0]) and torch.equal(ls_b2[0], ls_b[0])
    for k in w:
        assert torch.equal(g[k], w[k]), k
    assert torch.equal(gb[0], wb[0]) and torch.equal(gb[1], wb[1])
    # what does not qualify: no epilogue at all, an epilogue on a critic job
    assert not native.mlp_forward_multi_sampled_ok([job_a], [native.sample_epilogue()])
    bad = native.sample_epilogue(job_a, eps, g['a'], g['lp'], T)
    assert not native.mlp_forward_multi_sampled_ok([fq.job(x0, a0)[0]], [bad])
    # ... and windows whose tiles outnumber the launch's workgroups (the epilogue would run several times per workgroup)
    big = fpi.job(torch.randn(20000, S, **f), None)[0]
    eb = native.sample_epilogue(big, torch.randn(20000, A, **f), torch.zeros(20000, A, **f), torch.zeros(20000, **f), 1)
    assert not native.mlp_forward_multi_sampled_ok([big], [eb])


@pytest.mark.parametrize('N', [256, 1280, 7])
def test_policy_forward_backward_and_gauss_head(N):
    from algorithm.fused_mlp import gauss_head
    import asac_amd  # noqa: F401
    from asac_amd import native
    S, A = 6, 2
    mods, group, mlp = _setup(1, S, A, policy=True)
    pi = mods[0]
    x = torch.randn(N, S, device='cuda') * 2
    g_loc, g_scale = torch.randn(N, A, device='cuda'), torch.randn(N, A, device='cuda')
    _, dist = pi(x, None)
    (dist.loc * g_loc + dist.scale * g_scale).sum().backward()
    ref_gp = group.grad.clone()
    group.grad.zero_()
    ls = mlp(x)[0]                      # the fused policy emits (loc | scale) directly
    loc, scale = ls[..., :A], ls[..., A:]
    # stand-alone Gaussian-head kernel on raw (mean | logstd) values
    raw = torch.randn(N, 2 * A, device='cuda') * 3
    l2, s2 = gauss_head(raw, A)
    np.testing.assert_allclose(l2.cpu().numpy(), (torch.tanh(raw[:, :A] / 5) * 5).cpu().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(s2.cpu().numpy(), torch.exp(raw[:, A:].clamp(-20, 0.5)).cpu().numpy(), rtol=1e-5)
    np.testing.assert_allclose(loc.detach().cpu().numpy(), dist.loc.detach().cpu().numpy(), rtol=2e-5, atol=1e-5)
    np.testing.assert_allclose(scale.detach().cpu().numpy(), dist.scale.detach().cpu().numpy(), rtol=2e-5, atol=1e-5)
    (loc * g_loc + scale * g_scale).sum().backward()
    np.testing.assert_allclose(group.grad.cpu().numpy(), ref_gp.cpu().numpy(), rtol=1e-4, atol=2e-5)


def test_non_stock_models_are_not_fused():
    import asac_amd  # noqa: F401
    import algorithm.nn_models as m
    from algorithm.fused_mlp import describe_policy, describe_q
    assert describe_q(m.ModelQ(6, [3], 2, False)) is None          # discrete head present
    assert describe_policy(m.ModelPolicy(6, [3], 2)) is None

    class WideQ(m.ModelQ):
        def _build_model(self):
            super()._build_model(c_dense_n=128)
    assert describe_q(WideQ(6, [], 2, False)) is None


def _dense_in_flat_buffers(in_size, widths, out, seed=0):
    """a `LinearLayers` stack whose parameters / gradients are views of flat buffers (as inside SAC_Base) and a
    free-standing copy of it"""
    import copy
    import asac_amd  # noqa: F401
    import algorithm.nn_models as m
    from algorithm.fused import FlatParamGroup
    torch.manual_seed(seed)
    ref = m.LinearLayers(in_size, widths, len(widths), out).cuda()
    with torch.no_grad():
        for p in ref.parameters():
            if p.dim() == 1:
                p.normal_(0, 0.3)
    dev = copy.deepcopy(ref)
    group = FlatParamGroup([('dense', list(dev.parameters()))], 'cuda')
    dev.fuse = True
    return ref, dev, group


@pytest.mark.parametrize('N,in_size,widths,out', [
    (4608, 128, [64, 64], 8),        # the cfg4 encoder head: ResBlock(128->64), ResBlock(64->64)+res, Linear(64->8)
    (37, 128, [64, 64, 64], 16),
    (300, 100, [48], 3),             # ragged second half (36 columns), single block
    (70, 65, [64, 32], 1),           # one column in the second half
    (129, 40, [64, 64], 8),          # narrow input through the same entry (not wide)
])
def test_fused_dense_stack_matches_modules(N, in_size, widths, out):
    """`LinearLayers` with `fuse` (inputs up to 128 wide: two K halves of the first layer): forward, input
    gradient and parameter gradients (added into the flat `.grad` views) against the module path."""
    from asac_amd import native
    ref, dev, group = _dense_in_flat_buffers(in_size, widths, out)
    gen = torch.Generator().manual_seed(1)
    x = torch.randn(N, in_size, generator=gen).cuda()
    gy = torch.randn(N, out, generator=gen).cuda()
    xr = x.clone().requires_grad_(True)
    want = ref(xr)
    (want * gy).sum().backward()
    xd = x.clone().requires_grad_(True)
    group.grad.zero_()
    with native.LaunchProfiler() as prof:
        got = dev(xd)
        (got * gy).sum().backward()
    launches = prof.summary()
    assert launches['asac_mlp_forward']['calls'] == 1 and launches['asac_mlp_backward']['calls'] == 1
    np.testing.assert_allclose(got.detach().cpu().numpy(), want.detach().cpu().numpy(), rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(xd.grad.cpu().numpy(), xr.grad.cpu().numpy(), rtol=2e-4, atol=2e-6)
    for pr, pd in zip(ref.parameters(), dev.parameters()):
        scale = max(float(pr.grad.abs().max()), 1.0)
        np.testing.assert_allclose(pd.grad.cpu().numpy(), pr.grad.cpu().numpy(), rtol=3e-4, atol=3e-6 * scale)
    # accumulates into the existing gradient; inference takes the same launch without autograd
    before = group.grad.clone()
    (dev(xd) * gy).sum().backward()
    np.testing.assert_allclose(group.grad.cpu().numpy(), 2 * before.cpu().numpy(), rtol=1e-5, atol=1e-6)
    with torch.no_grad():
        assert torch.equal(dev(x), got.detach())
    # leading batch dims, and the module path when the stack does not fit / is free-standing
    assert dev(x[:24].reshape(2, 3, 4, in_size)).shape == (2, 3, 4, out)
    ref.fuse = True
    with native.LaunchProfiler() as prof:
        ref(x)          # parameters are separate allocations: no flat alias
    assert 'asac_mlp_forward' not in prof.summary()


def test_wide_job_beside_a_narrow_one_goes_launch_by_launch():
    """`mlp_forward_multi` with a critic whose first layer is wider than 64 inputs: the jobs are issued one launch each
    (the narrow one keeps its window addressing), results those of the single-network forward"""
    from asac_amd import native
    _, _, fq = _setup(2, 64, 2)
    _, _, fpi = _setup(1, 8, 2, policy=True)
    assert fq.wide and not fpi.wide
    B, T = 40, 5
    xs, a = torch.randn(B * T, 64, device='cuda'), torch.randn(B * T, 2, device='cuda').tanh()
    base = torch.randn(B, T + 2, 8, device='cuda')
    job_q, q_out = fq.job(xs, a)
    job_pi, pi_out = fpi.job(native.WindowRows(base[:, 2:]), None)
    native.mlp_forward_multi([job_q, job_pi])
    assert torch.equal(q_out, fq._launch_forward(xs, a))
    assert torch.equal(pi_out, fpi._launch_forward(base[:, 2:].reshape(B * T, 8), None))
