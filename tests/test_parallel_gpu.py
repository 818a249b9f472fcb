"""GPU: the multi-rank pieces on the device.
* `asac_sumtree_descend` against the oracle's descent (explicit values, zero-priority leaves, boundary values);
* "parity" sharded sampling (`algorithm/parallel.py`) with ONE rank and the product buffer as the shard:
  it must reproduce the plain sampler bit for bit (ids, IS weights, windows) and route the write-backs;
* two product learners on two GPUs over RCCL (skipped on a single-GPU box): throughput mode keeps the replicas'
  weights identical step after step; parity mode draws the same global batch on both ranks."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle.per_ref import SumTreeRef  # noqa: E402
from tests import parity_utils as pu  # noqa: E402


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


@pytest.mark.parametrize('C,n', [(16, 5), (1024, 700), (2 ** 19, 4096)])
def test_sumtree_descend_matches_oracle(C, n):
    import asac_amd  # noqa: F401
    from asac_amd import native
    native.load()
    rng = np.random.default_rng(C)
    leaves = (rng.random(C) * (rng.random(C) < 0.8)).astype(np.float32)
    t = SumTreeRef(C)
    t.update(np.arange(C), leaves)
    v = rng.random(n) * float(t.total)
    v[0], v[-1] = 0.0, float(t.total)
    leaf_ref, p_ref = t.descend(v)
    tree = torch.from_numpy(t.tree).cuda()
    slot_ids = (torch.arange(C, dtype=torch.int64) * 3 + 1).cuda()
    leaf, p, ids = (torch.empty(n, dtype=torch.int32, device='cuda'), torch.empty(n, device='cuda'),
                    torch.empty(n, dtype=torch.int64, device='cuda'))
    native.sumtree_descend(tree, C, torch.from_numpy(v).cuda(), slot_ids, leaf, p, ids)
    assert np.array_equal(leaf.cpu().numpy(), leaf_ref)
    assert np.array_equal(p.cpu().numpy().view(np.uint32), p_ref.view(np.uint32))
    assert np.array_equal(ids.cpu().numpy(), (leaf_ref - (C - 1)) * 3 + 1)


@pytest.mark.parametrize('G', [1, 2, 4, 8])
def test_sharded_plan_kernels_equal_single_tree(G):
    """asac_sumtree_plan_top + asac_sumtree_descend_owned + asac_per_is_weights_slice, run once per shard the way G ranks
    would, against ONE reference tree over the union of the shards: owners, leaves, priorities bit-exact; the IS weights
    of every rank's slice equal the single buffer's (replay_buffer.py:185-205, 352-354) bit for bit; and against the
    host statement of the plan (`parallel.plan_global_sample`)."""
    import asac_amd  # noqa: F401
    from asac_amd import native
    from algorithm.parallel import plan_global_sample
    native.load()
    rng = np.random.default_rng(17 + G)
    Cs, B = 1024, 96 * G if G < 8 else 1024
    leaves = (rng.random(G * Cs) * (rng.random(G * Cs) < 0.7)).astype(np.float32)
    # the boundary values u = 0 and u -> 1 must land on leaves WITH priority, so that the weights below are real numbers
    # for every G (a zero-priority first leaf made min(ratio) 0 and the comparison vacuous at G = 1); the degenerate
    # draw is asserted on its own at the end
    leaves[0], leaves[-1] = max(leaves[0], 0.25), max(leaves[-1], 0.5)
    union = SumTreeRef(G * Cs)
    union.update(np.arange(G * Cs), leaves)
    shards = []
    for g in range(G):
        t = SumTreeRef(Cs)
        t.update(np.arange(Cs), leaves[g * Cs:(g + 1) * Cs])
        shards.append(t)
    u = rng.random(B)
    u[0], u[-1] = 0.0, np.nextafter(1.0, 0.0)
    leaf_ref, p_ref = union.sample(B, u)
    owner_ref, v_ref, total_ref = plan_global_sample(np.array([t.total for t in shards], np.float32), B, u)
    f = dict(device='cuda')
    roots = torch.tensor([t.total for t in shards], dtype=torch.float32, **f)
    u_d = torch.from_numpy(u).cuda()
    p_sum = torch.zeros(B, **f)
    per = B // G
    for g in range(G):
        owner, v, total = torch.zeros(B, dtype=torch.int32, **f), torch.zeros(B, dtype=torch.float64, **f), torch.zeros(1, **f)
        native.sumtree_plan_top(roots, B, u_d, owner, v, total)
        assert np.array_equal(owner.cpu().numpy(), owner_ref) and float(total) == float(total_ref) == float(union.total)
        assert np.array_equal(v.cpu().numpy(), v_ref), 'residual values (f64, bit-exact)'
        assert np.array_equal(owner.cpu().numpy(), (leaf_ref - (G * Cs - 1)) // Cs)
        tree = torch.from_numpy(shards[g].tree).cuda()
        slot_ids = (torch.arange(Cs, dtype=torch.int64) * 5 + 2).cuda()
        leaf, p, ids = torch.zeros(B, dtype=torch.int32, **f), torch.zeros(B, **f), torch.zeros(B, dtype=torch.int64, **f)
        native.sumtree_descend_owned(tree, Cs, v, owner, g, slot_ids, leaf, p, ids)
        mine = owner_ref == g
        leaf, p, ids = leaf.cpu().numpy(), p.cpu().numpy(), ids.cpu().numpy()
        assert np.array_equal(leaf[mine] - (Cs - 1) + g * Cs, leaf_ref[mine] - (G * Cs - 1))
        assert np.array_equal(p[mine].view(np.uint32), p_ref[mine].view(np.uint32))
        assert np.array_equal(ids[mine], (leaf[mine] - (Cs - 1)) * 5 + 2)
        assert (leaf[~mine] == -1).all() and (p[~mine] == 0).all() and (ids[~mine] == -1).all()
        p_sum += torch.from_numpy(p).cuda()
    assert np.array_equal(p_sum.cpu().numpy().view(np.uint32), p_ref.view(np.uint32)), 'the all-reduce\'s sum'
    assert (p_ref > 0).all(), 'every drawn leaf carries priority: the weights below are finite and non-trivial'
    ratio = p_ref / union.total
    w_ref = np.power(ratio / np.min(ratio), -np.float64(0.401)).astype(np.float32)
    assert np.isfinite(w_ref).all() and w_ref.max() == 1.0 and np.unique(w_ref).size > B // 2
    total_d = torch.tensor([float(union.total)], **f)
    for g in range(G):
        beta = torch.tensor([0.4], dtype=torch.float64, **f)
        w = torch.zeros(per, **f)
        native.per_is_weights_slice(p_sum, g * per, per, total_d, beta, 0.001, w)
        assert float(beta) == 0.401
        np.testing.assert_allclose(w.cpu().numpy(), w_ref[g * per:(g + 1) * per], rtol=2e-6, atol=0)
    # a zero-priority leaf among the draws (the reference divides by a minimum ratio of 0, replay_buffer.py:352-354):
    # weight 0 where p > 0 (inf ** -beta), NaN where p == 0 (0 / 0) — asserted as that pattern, not through NaN == NaN
    p_deg = p_ref.copy()
    p_deg[1] = 0.0
    p_deg_d = torch.from_numpy(p_deg).cuda()
    for g in range(G):
        beta = torch.tensor([0.4], dtype=torch.float64, **f)
        w = torch.zeros(per, **f)
        native.per_is_weights_slice(p_deg_d, g * per, per, total_d, beta, 0.001, w)
        got, mine = w.cpu().numpy(), p_deg[g * per:(g + 1) * per]
        assert np.array_equal(np.isnan(got), mine == 0) and (got[mine > 0] == 0).all()


def _agent(dist_ctx=None, sampling='throughput', device='cuda:0', graph=False, seed=3):
    import asac_amd  # noqa: F401
    from algorithm.sac_base import SAC_Base
    from algorithm.utils.enums import SEQ_ENCODER
    torch.manual_seed(seed)
    return SAC_Base(['vector'], [(6,)], [], 2, None, pu.plugin('nn_rnn'), device=device, batch_size=32, n_step=3,
                    burn_in_step=2, seq_encoder=SEQ_ENCODER.RNN, replay_config={'capacity': 512},
                    hip_config={'use_graph': graph, 'dist': dist_ctx, 'dist_sampling': sampling})


def test_parity_sampling_single_rank_equals_plain_sampler():
    """world size 1: the sharded protocol (asac_sumtree_plan_top, asac_sumtree_descend_owned, asac_per_is_weights_slice,
    windows by gather_windows, write-backs through the plan — all on the device) must be the plain step — same ids
    drawn, same weights, same batch, same priorities / mu-probabilities / hidden states written back, same weights
    after the steps."""
    import torch.distributed as dist
    from algorithm.fused import RecordedNoise
    from algorithm.parallel import DataParallelContext
    dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{_free_port()}', rank=0, world_size=1,
                            device_id=torch.device('cuda:0'))
    try:
        plain = _agent()
        sharded = _agent(DataParallelContext(always=False), 'parity')
        assert sharded.replay_buffer.sharded is not None
        sharded._params.flat.copy_(plain._params.flat)
        sharded._target_params.flat.copy_(plain._target_params.flat)
        rng = np.random.default_rng(0)
        for T in (60, 45, 70, 80, 33):
            ep = pu.synthetic_episode(rng, [(6,)], [], 2, (2, 8), T)
            plain.put_episode(**ep)
            sharded.put_episode(**ep)
        for step in range(4):
            u, eps, perm = pu.host_draws(rng, 32, 3, 2, 2)
            for ag in (plain, sharded):
                ag.noise = RecordedNoise(list(u), [e.copy() for e in eps], list(perm))
                ag.replay_buffer.uniform_source = ag.noise
                ag.train()
            a, b = plain.replay_buffer, sharded.replay_buffer
            assert torch.equal(a._w, b._w), f'step {step}: IS weights'
            for k in a._batch:
                assert torch.equal(a._batch[k], b._batch[k]), f'step {step}: window {k}'
            assert torch.equal(a._tree, b._tree), f'step {step}: priorities written back'
            assert torch.equal(a._columns['mu_prob'], b._columns['mu_prob'])
            assert torch.equal(a._columns['pre_seq_hidden_state'], b._columns['pre_seq_hidden_state'])
            assert torch.equal(plain._params.flat, sharded._params.flat), f'step {step}: weights'
        plain.close()
        sharded.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('plugin,kw', [('nn_vec', dict(n_step=4, burn_in_step=0)),
                                       ('nn_rnn', dict(n_step=3, burn_in_step=2, seq_encoder='RNN'))])
@pytest.mark.parametrize('graph', [False, True])
@pytest.mark.parametrize('sampling', ['throughput', 'parity'])
def test_data_parallel_single_rank_equals_plain_step(plugin, kw, graph, sampling):
    """world size 1 with the collectives forced on (`always=True`: every RCCL call of the data-parallel step is
    issued, each an identity): the data-parallel step — throughput mode: sampler with deferred IS weights + MIN
    all-reduce; parity mode: roots all-gather, plan / owned descent / weight-slice launches, priority all-reduce,
    one all-to-all per key, all-gathers for the write-backs; both: gradient all-reduces in front of every Adam, the
    temperature step on the rank-averaged log-probabilities — must leave bit-identical weights, priorities and
    write-backs to the plain step's, eagerly AND as a captured graph (no host synchronisation in either mode)."""
    import torch.distributed as dist
    import asac_amd  # noqa: F401
    from algorithm.parallel import DataParallelContext
    from algorithm.sac_base import SAC_Base
    from algorithm.utils.enums import SEQ_ENCODER
    dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{_free_port()}', rank=0, world_size=1,
                            device_id=torch.device('cuda:0'))
    try:
        kw = dict(kw)
        hidden = (2, 8) if 'seq_encoder' in kw else (0,)
        if 'seq_encoder' in kw:
            kw['seq_encoder'] = SEQ_ENCODER.RNN

        def make(ctx):
            torch.manual_seed(5)
            return SAC_Base(['vector'], [(6,)], [], 2, None, pu.plugin(plugin), device='cuda:0', batch_size=32, seed=11,
                            replay_config={'capacity': 512},
                            hip_config={'use_graph': graph, 'graph_warmup': 2, 'dist': ctx, 'dist_sampling': sampling}, **kw)
        plain, dp = make(None), make(DataParallelContext(always=True))
        assert (dp.replay_buffer.min_ratio_reducer is not None) == (sampling == 'throughput')
        assert (dp.replay_buffer.sharded is not None) == (sampling == 'parity')
        dp._params.flat.copy_(plain._params.flat)
        dp._target_params.flat.copy_(plain._target_params.flat)
        rng = np.random.default_rng(0)
        for T in (60, 45, 70, 80, 33):
            ep = pu.synthetic_episode(rng, [(6,)], [], 2, hidden, T)
            plain.put_episode(**ep)
            dp.put_episode(**ep)
        for step in range(6):
            plain.train()
            dp.train()
            a, b = plain.replay_buffer, dp.replay_buffer
            if sampling == 'throughput':       # (parity mode: the ids live in the owners' exchange plan)
                assert torch.equal(a._ids, b._ids), f'step {step}: ids'
            else:
                assert torch.equal(a._ids, b.sharded._plan[1]), f'step {step}: ids'
            assert torch.equal(a._w, b._w), f'step {step}: IS weights'
            for k in a._batch:
                assert torch.equal(a._batch[k], b._batch[k]), f'step {step}: window {k}'
            assert float(a._beta) == float(b._beta)
            assert torch.equal(a._tree, b._tree), f'step {step}: priorities written back'
            assert torch.equal(a._columns['mu_prob'], b._columns['mu_prob'])
            assert torch.equal(plain._params.flat, dp._params.flat), f'step {step}: weights'
        if graph:
            assert plain._graph is not None and dp._graph is not None
        plain.close()
        dp.close()
    finally:
        dist.destroy_process_group()


def _two_rank_worker(rank, port, out_dir):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.cuda.set_device(rank)
    dist.init_process_group('nccl', rank=rank, world_size=2, device_id=torch.device('cuda', rank))
    from algorithm.parallel import DataParallelContext, shard_of_episode
    for sampling in ('throughput', 'parity'):
        ctx = DataParallelContext()
        agent = _agent(ctx, sampling, device=f'cuda:{rank}', graph=(sampling == 'throughput'), seed=3 + rank)
        rng = np.random.default_rng(1)            # same episode stream everywhere, dealt round-robin to the shards
        for ep_i, T in enumerate((60, 45, 70, 80, 33, 52, 41, 66)):
            ep = pu.synthetic_episode(rng, [(6,)], [], 2, (2, 8), T)
            if shard_of_episode(ep_i, 2) == rank:
                agent.put_episode(**ep)
        for _ in range(8):
            agent.train()
        torch.cuda.synchronize()
        agent.replay_buffer.check_health()
        flat = agent._params.flat.clone()
        other = [torch.zeros_like(flat) for _ in range(2)]
        dist.all_gather(other, flat)
        assert torch.equal(other[0], other[1]), f'{sampling}: the replicas diverged'
        assert torch.isfinite(flat).all()
        if sampling == 'throughput':
            assert agent._graph is not None, 'the step with its RCCL collectives is captured'
        agent.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs two GPUs')
def test_two_product_ranks_over_rccl(tmp_path):
    import torch.multiprocessing as mp
    mp.spawn(_two_rank_worker, args=(_free_port(), str(tmp_path)), nprocs=2, join=True)
