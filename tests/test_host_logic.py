"""CPU: host-side logic of the package — plugin surface compatibility with the reference's
checkpoints (state_dict keys / shapes taken from the golden fixtures), operators against the oracle,
the flat-parameter re-homing used by the fused optimizer kernels, config enums."""
import numpy as np
import pytest
import torch

import algorithm.nn_models as m
from algorithm.fused import FlatParamGroup
from algorithm.utils import enums, operators
from oracle import sac_ref
from tests.plugins import nn_rnn, nn_vec


@pytest.mark.parametrize('case,nn_mod,d_sizes', [('cfg2', nn_vec, []), ('cfg3', nn_rnn, []), ('hybrid', nn_vec, [3, 2])])
def test_state_dict_keys_match_reference_checkpoints(golden_dir, case, nn_mod, d_sizes):
    g = np.load(golden_dir / f'f6_step_{case}.npz')
    state = 8 if case == 'cfg3' else 6
    mods = {'model_q_0': nn_mod.ModelQ(state, d_sizes, 2, False),
            'model_policy': nn_mod.ModelPolicy(state, d_sizes, 2),
            'model_rep': nn_mod.ModelRep(['vector'], [(6,)], d_sizes, 2, False)}
    for name, mod in mods.items():
        want = {k[len(f'w0/{name}/'):]: g[k].shape for k in g.files if k.startswith(f'w0/{name}/')}
        have = {k: tuple(v.shape) for k, v in mod.state_dict().items()}
        assert have == want, name


def test_linear_layers_contract():
    ll = m.LinearLayers(10, dense_n=[32, 16], output_size=3)
    assert ll.output_size == 3 and ll(torch.randn(5, 10)).shape == (5, 3)
    assert m.LinearLayers(7).output_size == 7 and len(list(m.LinearLayers(7).parameters())) == 0
    rb = m.ResBlock(8, 8)
    x = torch.randn(4, 8)
    torch.testing.assert_close(rb(x), torch.nn.functional.gelu(rb.linear(x)) + x)
    assert not m.ResBlock(8, 4).residual
    assert torch.count_nonzero(ll.dense[0].linear.bias) == 0


def test_gru_masks_like_a_packed_sequence():
    torch.manual_seed(0)
    gru = m.GRU(5, 8, 2)
    x, h0 = torch.randn(3, 6, 5), torch.randn(3, 2, 8)
    pad = torch.tensor([[1, 1, 0, 0, 0, 0], [0, 0, 0, 0, 0, 0], [0, 0, 0, 0, 1, 1]], dtype=torch.bool)
    out, hn = gru(x, h0, pad)
    assert out.shape == (3, 6, 8) and hn.shape == (3, 6, 2, 8)
    assert torch.all(out[pad] == 0) and torch.all(hn[pad] == 0)
    # row 0: the valid block starts at t=2 from h0 -> equals running the GRU on x[0, 2:]
    ref, _ = gru(x[0:1, 2:], h0[0:1])
    torch.testing.assert_close(out[0, 2:], ref[0])
    ref1, _ = gru(x[1:2], h0[1:2])
    torch.testing.assert_close(out[1], ref1[0])


def test_operators_match_oracle():
    torch.manual_seed(1)
    loc, scale = torch.randn(7, 3), torch.rand(7, 3) + 0.1
    dist = torch.distributions.Normal(loc, scale, validate_args=False)
    x = torch.randn(7, 3)
    torch.testing.assert_close(operators.squash_correction_log_prob(dist, x), sac_ref.squash_log_prob(dist, x))
    torch.testing.assert_close(operators.squash_correction_prob(dist, x), sac_ref.squash_prob(dist, x))
    p = torch.tensor([[0.5, float('inf'), 2.0], [float('inf'), float('inf'), 1.0]])
    assert operators.prod_prob(p.clone()).tolist() == [1.0, 1.0]
    lp = torch.tensor([[1.0, float('inf')]])
    assert operators.sum_log_prob(lp).item() == 1.0 and lp[0, 1].item() == 0.0   # input mutated, like the reference
    a = torch.arange(12.).reshape(1, 4, 3)
    pre = operators.gen_n_pre_actions(a, keep_last_action=True)
    assert pre.shape == (1, 5, 3) and torch.all(pre[:, 0] == 0) and torch.equal(pre[:, 1:], a)
    assert operators.gen_n_pre_actions(a.numpy()).shape == (1, 4, 3)
    mask = torch.tensor([[False, True, False, True], [False, False, False, False]])
    assert operators.get_last_false_indexes(mask, dim=1).tolist() == [2, 3]


def test_flat_param_group_rehomes_params_and_grads():
    torch.manual_seed(2)
    q = nn_vec.ModelQ(6, [], 2, False)
    pol = nn_vec.ModelPolicy(6, [], 2)
    before = {k: v.clone() for k, v in q.state_dict().items()}
    g = FlatParamGroup([('q', list(q.parameters())), ('policy', list(pol.parameters()))], 'cpu')
    assert g.segments['q'][0] == 0 and g.segments['policy'][0] % 4 == 0
    for k, v in q.state_dict().items():
        torch.testing.assert_close(v, before[k])
    q(torch.randn(5, 6), torch.randn(5, 2), None)[1].sum().backward()
    s, e = g.span('q')
    assert torch.count_nonzero(g.grad[s:e]) > 0 and torch.count_nonzero(g.grad[e:]) == 0
    first = next(q.parameters())
    assert first.grad.data_ptr() == g.grad.data_ptr() and first.data.data_ptr() == g.flat.data_ptr()
    g.grad.zero_()
    assert all(torch.count_nonzero(p.grad) == 0 for p in q.parameters())


def test_config_enums_roundtrip():
    cfg = {'seq_encoder': 'RNN', 'siamese': None, 'curiosity': 'FORWARD'}
    enums.convert_config_to_enum(cfg)
    assert cfg['seq_encoder'] is enums.SEQ_ENCODER.RNN and cfg['curiosity'] is enums.CURIOSITY.FORWARD
    enums.convert_config_to_string(cfg)
    assert cfg == {'seq_encoder': 'RNN', 'siamese': None, 'curiosity': 'FORWARD'}


def test_adjacent_cat_returns_views_only_for_side_by_side_blocks():
    """`adjacent_cat.joined_view` / `AdjacentCat`: a last-dim concatenation of adjacent column blocks of one tensor is
    handed out as a view of it; anything else is ATen's concatenation."""
    import torch
    import asac_amd  # noqa: F401
    from algorithm.adjacent_cat import AdjacentCat, joined_view
    base = torch.arange(4 * 5 * 9, dtype=torch.float32).reshape(4, 5, 9)
    a, b, c = base[..., 0:4], base[..., 4:6], base[..., 6:9]
    v = joined_view([a, b], -1)
    assert v is not None and v.data_ptr() == a.data_ptr() and torch.equal(v, torch.cat([a, b], -1))
    assert torch.equal(joined_view((a, b, c), 2), base)
    assert joined_view([a, c], -1) is None                         # a gap between the blocks
    assert joined_view([b, a], -1) is None                         # wrong order
    assert joined_view([a, b], 0) is None and joined_view([a], -1) is None
    assert joined_view([a, b.clone()], -1) is None                 # another storage
    assert joined_view([a, b.double()], -1) is None
    assert joined_view([a[:, :3], b[:, 1:4]], -1) is None          # different rows
    ag = a.clone().requires_grad_()
    assert joined_view([ag, b], -1) is None
    flat = torch.arange(12.)
    assert joined_view([flat[:6], flat[6:]], 0) is None            # one-dimensional: no row pitch to stay inside
    tail = base[:, :, 5:9]
    assert joined_view([tail, base[:, :, 0:4].roll(0)], -1) is None
    with AdjacentCat():
        assert torch.cat([a, b], dim=-1).data_ptr() == a.data_ptr()
        assert torch.cat((a, b), -1).data_ptr() == a.data_ptr()
        assert torch.concat([b, c], dim=2).data_ptr() == b.data_ptr()
        fresh = torch.cat([a, c], dim=-1)
        assert fresh.data_ptr() != a.data_ptr() and torch.equal(fresh, torch.cat([a.clone(), c.clone()], -1))
        out = torch.empty(4, 5, 6)
        torch.cat([a, b], dim=-1, out=out)
        assert torch.equal(out, base[..., :6])
        assert torch.equal(torch.cat([a, b], dim=1), torch.cat([a.clone(), b.clone()], 1)) if a.shape[-1] == b.shape[-1] else True
        assert torch.equal(torch.stack([a, a]).sum(0), 2 * a)


def test_deferred_cat_behaves_like_the_concatenation_for_every_other_consumer():
    """`adjacent_cat.DeferredCat`: whatever touches the object — a torch function, a method, an operator, an attribute,
    `nn.Linear` — sees `torch.cat(parts, -1)`, formed once; gradients reach the parts."""
    import torch
    from torch import nn
    import asac_amd  # noqa: F401
    from algorithm.adjacent_cat import DeferredCat, deferrable
    a = torch.randn(5, 3, 4)
    b = torch.randn(5, 3, 2, requires_grad=True)
    want = torch.cat([a, b], dim=-1)
    d = DeferredCat([a, b])
    assert d.width == 6 and d._value is None
    assert torch.equal(torch.relu(d), torch.relu(want)) and d._value is not None
    first = d._value
    assert torch.equal(d + 1, want + 1) and torch.equal(2 * d, 2 * want) and torch.equal(-d, -want)
    assert d.shape == want.shape and d.dim() == 3 and len(d) == 5 and torch.equal(d[1], want[1])
    assert torch.equal(d.reshape(15, 6), want.reshape(15, 6)) and torch.equal(torch.stack([d, d]), torch.stack([want, want]))
    assert d._value is first                                   # materialised once
    lin = nn.Linear(6, 2)
    out = lin(DeferredCat([a, b]))
    out.sum().backward()
    assert b.grad is not None and torch.equal(out, lin(want))
    assert not deferrable([a, b], -1, 64)                      # host tensors are never deferred
    assert not deferrable([a], -1, 64) and not deferrable([a, b, a], -1, 64)


def test_train_is_the_method_timed_as_train_a_step():
    """Reference `sac_base.py:2496`: `@unified_elapsed_timer('train a step', 10)` sits on `train` — the log hook the
    reference's users read; helpers next to it must not take the decorator."""
    from algorithm.sac_base import SAC_Base
    assert getattr(SAC_Base.train, 'elapsed_log', None) == 'train a step'
    assert SAC_Base.train.__wrapped__.__name__ == 'train'
    for helper in ('_optimizer_hp', '_drop_graphs_if_hp_changed', '_ready_to_train'):
        assert not hasattr(getattr(SAC_Base, helper), 'elapsed_log'), helper


def test_fused_row_routes_are_refused_for_unaligned_parameters():
    """Parameters are views packed back to back in the learner's flat buffer; only segment starts are 16-byte aligned.  A
    model with an earlier parameter of numel % 4 != 0 shifts the Linears behind it off the 16-byte grid the MFMA row
    kernels read on: the gating predicates must send such layers to nn.Linear instead of into a `bad_arg`."""
    from algorithm.nn_models.layers import seq_layers as sl
    lin = torch.nn.Linear(64, 64)
    stack = torch.nn.Module()
    stack.dense = torch.nn.Sequential(lin)
    assert sl._plain_linear(stack) is lin
    flat = torch.zeros(64 * 64 + 64 + 8)
    for off, ok in ((4, True), (1, False), (2, False)):        # element offsets into an aligned buffer
        base = (-flat.data_ptr() // 4) % 4                     # first element on a 16-byte boundary
        w = flat[base + off:base + off + 64 * 64].view(64, 64)
        b = flat[base + off + 64 * 64:base + off + 64 * 64 + 64]
        lin.weight.data, lin.bias.data = w, b
        assert sl._params_aligned16(lin) == ok
        assert (sl._plain_linear(stack) is lin) == ok


def test_mse_intercept_only_reroutes_the_thread_that_installed_it():
    """`fused.fused_mse_loss` patches the process-global `torch.nn.functional.mse_loss` while the learner's `get_loss`
    runs: a call from any OTHER thread in that window must reach the original function (never the learner's workspace)."""
    import threading
    import torch.nn.functional as F
    from algorithm import fused
    seen = {}
    orig = F.mse_loss
    with fused.fused_mse_loss(torch.zeros(8), grad_scale=0.25):
        patched = F.mse_loss
        assert patched is not orig

        def other():
            a = torch.ones(2, 2, 3, requires_grad=True)
            seen['value'] = float(F.mse_loss(a, torch.zeros(2, 2, 3)))
        t = threading.Thread(target=other)
        t.start()
        t.join()
        # (small CPU tensors take the original path on the owner thread too: the route itself is GPU-tested)
        assert float(F.mse_loss(torch.ones(2, 2, 3, requires_grad=True), torch.zeros(2, 2, 3))) == 1.0
    assert F.mse_loss is orig and seen['value'] == 1.0
