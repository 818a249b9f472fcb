"""Mint golden input/output vectors from the *reference* implementation.

Run in the BUILD CONTAINER ONLY (it imports `/root/reference`, which does not exist on the GPU
box):      python tests/golden/make_golden.py
It writes the small `.npz` fixtures committed next to this file.  Fixtures are data only (inputs,
recorded random draws, expected outputs); no reference source is copied.  The reference's own
tests hold no vectors for this path, so these are what pins the oracle
(`oracle/`), and through it the HIP path.

Fixtures:
  f1_sumtree.npz   SumTree.update / sample: priorities + idx -> tree bytes; u -> leaf idx, p
  f2_per.npz       PER front-end: adds with ignore_size, ring wrap, stale ids, beta schedule
  f3_vtrace.npz    SAC_Base._v_trace on random [B, n] inputs (n in {1, 4, 40}, IS on/off)
  f4_get_y.npz     SAC_Base._get_y with table-driven policy / target-Q stubs (ensemble min + V)
  f5_polyak.npz    SAC_Base._update_target_variables
  f6_step_<case>.npz  full train() steps: weights before/after, episodes, draws, observables
  f11_rpm.npz      SAC_Base._train_rpm + calculate_adaptive_weights on a fresh graph (use_prediction)

    python tests/golden/make_golden.py [fixture function names...]      (default: all)
"""
import importlib.util
import random
import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
import ref_shims  # noqa: E402

ref_shims.install()

from algorithm.replay_buffer import PrioritizedReplayBuffer, SumTree  # noqa: E402
from algorithm.sac_base import SAC_Base  # noqa: E402
from algorithm.utils.enums import SEQ_ENCODER  # noqa: E402
from agent_script import AGENT_CASES, agent_script  # noqa: E402


def load_ref_nn(rel):
    """a model-plugin file of the reference tree, or (absolute path) one of this repo's test plugins,
    which are written against the plugin API only and therefore load under the reference package too"""
    path = rel if str(rel).startswith('/') else f'{ref_shims.REFERENCE_ROOT}/{rel}'
    spec = importlib.util.spec_from_file_location('ref_nn_' + Path(rel).stem, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def seed_all(s):
    np.random.seed(s)
    random.seed(s)
    torch.manual_seed(s)


# ------------------------------------------------------------------------------------------------
# synchronous drive of the reference buffer (its prefetch thread makes runs irreproducible)
# ------------------------------------------------------------------------------------------------
_orig_loop = PrioritizedReplayBuffer._prefetch_loop


def _sync_sample(self):
    if not self.is_lg_batch_size:
        return None
    box = []
    buf = self

    class OneShot:
        def put(self, item):
            box.append(item)
            buf._closed = True   # leave the reference loop after exactly one body

        def empty(self):
            return True

    real_q, self._queue = self._queue, OneShot()
    self._closed = False
    _orig_loop(self)
    self._queue, self._closed = real_q, False
    ids, transitions, w = box[0]
    return ids, transitions, w.unsqueeze(-1)


PrioritizedReplayBuffer._prefetch_loop = lambda self: None
PrioritizedReplayBuffer.sample = _sync_sample


# ------------------------------------------------------------------------------------------------
def f1_sumtree():
    out = {}
    rng = np.random.default_rng(1)
    for tag, C, k, B in [('c16', 16, 9, 4), ('c1024', 1024, 700, 64), ('c524288', 2 ** 19, 30000, 256)]:
        t = SumTree(C)
        idx1 = rng.permutation(C)[:k].astype(np.int64)
        p1 = np.abs(rng.standard_normal(k)).astype(np.float32)
        p1[::7] = 0.                                  # zero-priority leaves
        t.update(idx1, p1)
        idx2 = rng.integers(0, C, size=min(k, 300)).astype(np.int64)   # duplicates: last wins
        idx2[:4] = idx2[4:8]
        p2 = rng.random(len(idx2)).astype(np.float32)
        t.update(idx2, p2)
        tree_after = t._tree.copy()
        with ref_shims.DrawRecorder() as rec:
            np.random.seed(7)
            leaf, p = t.sample(B)
        u = rec.u[0]
        # boundary uniforms: 0 and the largest double below 1
        ub = u.copy()
        ub[0], ub[-1] = 0.0, np.nextafter(1.0, 0.0)
        with ref_shims.DrawRecorder():
            orig = np.random.random_sample
            np.random.random_sample = lambda size=None: ub.copy()
            try:
                leaf_b, p_b = t.sample(B)
            finally:
                np.random.random_sample = orig
        out.update({f'{tag}_idx1': idx1, f'{tag}_p1': p1, f'{tag}_idx2': idx2, f'{tag}_p2': p2,
                    f'{tag}_u': u, f'{tag}_leaf': leaf, f'{tag}_p': p, f'{tag}_max': t.max,
                    f'{tag}_ub': ub, f'{tag}_leaf_b': leaf_b, f'{tag}_p_b': p_b,
                    f'{tag}_batch': B})
        if C <= 1024:
            out[f'{tag}_tree'] = tree_after
        else:   # 4 MiB of tree is too large to commit: keep the top 4095 nodes + checksums
            out[f'{tag}_tree_top'] = tree_after[:4095]
            out[f'{tag}_tree_sum64'] = np.float64(tree_after.astype(np.float64).sum())
            out[f'{tag}_tree_xor'] = np.bitwise_xor.reduce(tree_after.view(np.uint32))
    np.savez_compressed(HERE / 'f1_sumtree.npz', **out)


def f2_per():
    """A scripted session over a 64-slot ring: adds (wrapping twice), samples, priority updates
    (some stale), transition write-backs.  Every observable after every op is recorded."""
    seed_all(2)
    rng = np.random.default_rng(2)
    B, prev_n, post_n, C = 8, 2, 3, 64
    rb = PrioritizedReplayBuffer(B, prev_n, post_n, torch.device('cpu'), capacity=C)
    script, out = [], {}
    step = 0

    def episode(T):
        return {'index': np.arange(T, dtype=np.int32),
                'obs_vec': rng.standard_normal((T, 3)).astype(np.float32),
                'reward': rng.standard_normal(T).astype(np.float32),
                'done': rng.integers(0, 2, T).astype(bool),
                'mu_prob': rng.random((T, 2)).astype(np.float32)}

    for it in range(14):
        T = int(rng.integers(4, 23))
        ep = episode(T)
        rb.add(ep, ignore_size=1)
        for k, v in ep.items():
            out[f's{step}_add_{k}'] = v
        out[f's{step}_tree'] = rb._sum_tree._tree.copy()
        out[f's{step}_ids'] = rb._trans_storage._buffer['_id'].copy()
        script.append(f'add:{T}')
        step += 1
        if it >= 1:
            with ref_shims.DrawRecorder() as rec:
                sampled = rb.sample()
            if sampled is None:
                script.append('sample:none')
                step += 1
                continue
            ids, trans, w = sampled
            out[f's{step}_u'] = rec.u[0]
            out[f's{step}_sample_ids'] = ids
            out[f's{step}_w'] = w.numpy()
            out[f's{step}_beta'] = np.float64(rb.beta)
            for k, v in trans.items():
                out[f's{step}_win_{k}'] = v.numpy()
            script.append('sample')
            step += 1
            if it % 3 == 2:   # interleave an add so some sampled ids go stale before the update
                ep2 = episode(int(rng.integers(30, 50)))
                rb.add(ep2, ignore_size=1)
                for k, v in ep2.items():
                    out[f's{step}_add_{k}'] = v
                out[f's{step}_tree'] = rb._sum_tree._tree.copy()
                out[f's{step}_ids'] = rb._trans_storage._buffer['_id'].copy()
                script.append('add:stale')
                step += 1
            td = np.abs(rng.standard_normal((B, 1))).astype(np.float32) * 0.7
            rb.update(ids, td)
            out[f's{step}_td'] = td
            out[f's{step}_upd_ids'] = ids
            out[f's{step}_tree'] = rb._sum_tree._tree.copy()
            script.append('update')
            step += 1
            tgt = (ids[:, None] + np.arange(-prev_n, post_n)[None, :]).reshape(-1)
            new_mu = rng.random((len(tgt), 2)).astype(np.float32)
            rb.update_transitions(tgt, 'mu_prob', new_mu)
            out[f's{step}_ut_ids'] = tgt
            out[f's{step}_ut_data'] = new_mu
            out[f's{step}_mu_prob'] = rb._trans_storage._buffer['mu_prob'].copy()
            script.append('update_transitions')
            step += 1
    out['script'] = np.array(script)
    out['config'] = np.array([B, prev_n, post_n, C])
    rb.close()
    np.savez_compressed(HERE / 'f2_per.npz', **out)


def _tiny_sac(nn_mod, **kw):
    base = dict(obs_names=['vector'], obs_shapes=[(6,)], d_action_sizes=[], c_action_size=2,
                model_abs_dir=None, nn=nn_mod, device='cpu', batch_size=16,
                replay_config={'capacity': 64})
    base.update(kw)
    return SAC_Base(**base)


def f3_vtrace():
    nn_vec = load_ref_nn('envs/test/nn.py')
    out = {}
    rng = np.random.default_rng(3)
    for n in (1, 4, 40):
        for use_is in (True, False):
            sac = _tiny_sac(nn_vec, n_step=n, use_n_step_is=use_is, gamma=0.99, v_lambda=0.95,
                            v_rho=1.0, v_c=0.9)
            B = 37
            args = dict(
                n_last_masks=rng.random((B, n)) < 0.15, n_padding_masks=rng.random((B, n)) < 0.2,
                n_rewards=rng.standard_normal((B, n)).astype(np.float32),
                n_dones=rng.random((B, n)) < 0.3,
                n_mu_probs=(rng.random((B, n)) * 2).astype(np.float32),
                n_pi_probs=(rng.random((B, n)) * 2).astype(np.float32),
                n_vs=rng.standard_normal((B, n)).astype(np.float32),
                next_n_vs=rng.standard_normal((B, n)).astype(np.float32))
            args['n_mu_probs'][0, 0] = 0.0   # exercises clamp(min=1e-8)
            y = sac._v_trace(**{k: torch.from_numpy(v) for k, v in args.items()})
            tag = f'n{n}_is{int(use_is)}'
            for k, v in args.items():
                out[f'{tag}_{k}'] = v
            out[f'{tag}_y'] = y.numpy()
            out[f'{tag}_gamma_ratio'] = sac._gamma_ratio.numpy()
            out[f'{tag}_lambda_ratio'] = sac._lambda_ratio.numpy()
            sac.close()
    out['params'] = np.array([0.99, 0.95, 1.0, 0.9])   # gamma, lambda, rho_bar, c_bar
    np.savez_compressed(HERE / 'f3_vtrace.npz', **out)


def f4_get_y():
    """_get_y (continuous branch) with the policy and target-Q networks replaced by tables, so the
    fixture pins exactly the non-GEMM arithmetic the fused kernels implement: rsample, tanh squash
    log-prob, ensemble subset + min, V = minQ - alpha*logpi, pi/mu ratios, V-trace."""
    nn_vec = load_ref_nn('envs/test/nn.py')
    out = {}
    rng = np.random.default_rng(4)
    for tag, n, E, Es, A, use_is in [('n4_e2', 4, 2, 2, 2, True), ('n3_e4s2', 3, 4, 2, 4, True),
                                     ('n40_e2', 40, 2, 2, 2, True), ('n1_e2_nois', 1, 2, 2, 2, False)]:
        sac = _tiny_sac(nn_vec, n_step=n, c_action_size=A, ensemble_q_num=E, ensemble_q_sample=Es,
                        use_n_step_is=use_is)
        B = 21
        loc = torch.from_numpy(rng.standard_normal((B, n + 1, A)).astype(np.float32))
        scale = torch.from_numpy(np.exp(rng.uniform(-3, 0.5, (B, n + 1, A))).astype(np.float32))
        qtab = [torch.from_numpy(rng.standard_normal((B, n + 1, 1)).astype(np.float32)) for _ in range(E)]
        sac.model_policy = lambda states, obs: (None, torch.distributions.Normal(loc, scale, validate_args=False))
        sac.model_target_q_list = [(lambda s, a, o, t=t: (None, t)) for t in qtab]
        with torch.no_grad():
            sac.log_c_alpha.fill_(float(rng.uniform(-3, 0)))
        args = dict(
            n_last_masks=rng.random((B, n)) < 0.15, n_padding_masks=rng.random((B, n)) < 0.2,
            n_actions=np.clip(rng.uniform(-1.2, 1.2, (B, n, A)), -1, 1).astype(np.float32),
            n_rewards=rng.standard_normal((B, n)).astype(np.float32),
            n_dones=rng.random((B, n)) < 0.3,
            n_mu_probs=(rng.random((B, n, A)) * 2).astype(np.float32))
        nx_states = torch.zeros((B, n + 1, 6))
        seed_all(40 + n)
        with ref_shims.DrawRecorder() as rec:
            _, c_y = sac._get_y(nx_obses_list=[nx_states], nx_states=nx_states,
                                **{k: torch.from_numpy(v.copy()) for k, v in args.items()})
        for k, v in args.items():
            out[f'{tag}_{k}'] = v
        out[f'{tag}_loc'], out[f'{tag}_scale'] = loc.numpy(), scale.numpy()
        out[f'{tag}_q'] = torch.stack(qtab).numpy()
        out[f'{tag}_eps'] = rec.eps[0].numpy()
        out[f'{tag}_perm'] = torch.stack(rec.perm).numpy()
        out[f'{tag}_log_alpha'] = sac.log_c_alpha.detach().numpy()
        out[f'{tag}_y'] = c_y.numpy()
        out[f'{tag}_cfg'] = np.array([n, E, Es, A, int(use_is)])
        sac.close()
    np.savez_compressed(HERE / 'f4_get_y.npz', **out)


def f5_polyak():
    nn_vec = load_ref_nn('envs/test/nn.py')
    seed_all(5)
    sac = _tiny_sac(nn_vec)
    with torch.no_grad():
        for q in sac.model_q_list:
            for p in q.parameters():
                p.add_(torch.randn_like(p) * 0.1)
    out = {}
    src = [p.detach().clone().numpy() for q in sac.model_q_list for p in q.parameters()]
    before = [p.detach().clone().numpy() for q in sac.model_target_q_list for p in q.parameters()]
    sac._update_target_variables(tau=0.005)
    after = [p.detach().clone().numpy() for q in sac.model_target_q_list for p in q.parameters()]
    for i, (s, b, a) in enumerate(zip(src, before, after)):
        out[f'src_{i}'], out[f'before_{i}'], out[f'after_{i}'] = s, b, a
    out['tau'] = np.float64(0.005)
    sac.close()
    np.savez_compressed(HERE / 'f5_polyak.npz', **out)


# ------------------------------------------------------------------------------------------------
def gen_episode(rng, obs_shapes, d_action_sizes, c_action_size, hidden_shape, T):
    """Synthetic episode with the layout of reference tests/get_synthesis_data.py:114-126."""
    parts = []
    for s in d_action_sizes:
        parts.append(np.eye(s, dtype=np.float32)[rng.integers(0, s, T)])
    if c_action_size:
        parts.append(rng.random((T, c_action_size)).astype(np.float32))
    def obs(s):   # images are k/255 (8-bit pixels widened to f32), so fixtures can keep them as uint8
        if len(s) == 3:
            return rng.integers(0, 256, (1, T, *s)).astype(np.uint8).astype(np.float32) / np.float32(255.)
        return rng.standard_normal((1, T, *s)).astype(np.float32)

    return dict(
        ep_indexes=np.arange(T, dtype=np.int32)[None],
        ep_obses_list=[obs(s) for s in obs_shapes],
        ep_actions=np.concatenate(parts, -1)[None],
        ep_rewards=rng.standard_normal((1, T)).astype(np.float32),
        ep_dones=(rng.random((1, T)) < 0.5),
        ep_probs=rng.random((1, T, sum(d_action_sizes) + c_action_size)).astype(np.float32),
        ep_pre_seq_hidden_states=rng.standard_normal((1, T, *hidden_shape)).astype(np.float32))


def f6_step(case, nn_rel, sac_kw, ep_lens, n_steps, obs_shapes=((6,),), obs_names=('vector',),
            d_action_sizes=(), c_action_size=2, seed=6):
    nn_mod = load_ref_nn(nn_rel)
    seed_all(seed)
    rng = np.random.default_rng(seed)
    sac = SAC_Base(obs_names=list(obs_names), obs_shapes=list(obs_shapes),
                   d_action_sizes=list(d_action_sizes), c_action_size=c_action_size,
                   model_abs_dir=None, nn=nn_mod, device='cpu', **sac_kw)
    out = {}
    mods = {k: v for k, v in sac.ckpt_dict.items() if isinstance(v, torch.nn.Module)}
    for name, m in mods.items():
        for k, v in m.state_dict().items():
            out[f'w0/{name}/{k}'] = v.numpy().copy()
    tensors = {k: v for k, v in sac.ckpt_dict.items()
               if isinstance(v, torch.Tensor) and k not in ('global_step', 'log_d_alpha', 'log_c_alpha')}
    for name, t in tensors.items():      # contrastive weights, normaliser statistics
        out[f'w0/t/{name}'] = t.detach().numpy().copy()
    out['w0/log_d_alpha'] = sac.log_d_alpha.detach().numpy().copy()
    out['w0/log_c_alpha'] = sac.log_c_alpha.detach().numpy().copy()

    for i, T in enumerate(ep_lens):
        ep = gen_episode(rng, obs_shapes, d_action_sizes, c_action_size, tuple(sac.seq_hidden_state_shape), T)
        sac.put_episode(**ep)
        for k, v in ep.items():
            if k == 'ep_obses_list':
                for j, o in enumerate(v):
                    if o.ndim == 5:      # image k/255: stored as the 8-bit k (tests/parity_utils.golden_episodes)
                        u8 = np.rint(o * 255.).astype(np.uint8)
                        assert np.array_equal(u8.astype(np.float32) / np.float32(255.), o)
                        out[f'ep{i}/obs_{j}_u8'] = u8
                    else:
                        out[f'ep{i}/obs_{j}'] = o
            else:
                out[f'ep{i}/{k}'] = v
    out['n_episodes'] = np.int64(len(ep_lens))

    # observe _train_rep_q's returned loss and policy entropies without touching behaviour
    seen = {}
    orig_rq, orig_pol = sac._train_rep_q, sac._train_policy

    rep_trains = any(p.requires_grad for p in sac.model_rep.parameters())

    def rq(*a, **k):
        r = orig_rq(*a, **k)
        seen['loss_q'] = r[0].detach().numpy().copy()
        if rep_trains:
            # representation and critics right after their Adam updates (sac_base.py:1589-1603): everything the
            # step computes afterwards starts from these weights
            for name, m in mods.items():
                if name == 'model_rep' or name.startswith('model_q_'):
                    for kk, v in m.state_dict().items():
                        seen[f'w_rq/{name}/{kk}'] = v.numpy().copy()
        return r

    def pol(*a, **k):
        # the policy objective is a local of the reference's `_train_policy` (sac_base.py:1903): observe the
        # tensor its `.backward(inputs=...)` is called on
        orig_backward = torch.Tensor.backward

        def spy(t, *ba, **bk):
            seen.setdefault('loss_policy', t.detach().numpy().copy())
            return orig_backward(t, *ba, **bk)

        torch.Tensor.backward = spy
        try:
            r = orig_pol(*a, **k)
        finally:
            torch.Tensor.backward = orig_backward
        seen['d_ent'] = None if r[0] is None else r[0].detach().numpy().copy()
        seen['c_ent'] = None if r[1] is None else r[1].detach().numpy().copy()
        return r

    sac._train_rep_q, sac._train_policy = rq, pol
    if sac.curiosity is not None:
        orig_cur = sac._train_curiosity

        def cur(*a, **k):
            r = orig_cur(*a, **k)
            seen['loss_curiosity'] = r.detach().numpy().copy()
            return r

        sac._train_curiosity = cur
    orig_update = sac.replay_buffer.update

    def upd(ids, td):
        seen['ids'], seen['td'] = np.array(ids).copy(), np.array(td).copy()
        return orig_update(ids, td)

    sac.replay_buffer.update = upd
    orig_sample = sac.replay_buffer.sample

    def smp():
        r = orig_sample()
        if r is not None:
            seen['sample_ids'], seen['w'] = np.array(r[0]).copy(), r[2].numpy().copy()
        return r

    sac.replay_buffer.sample = smp

    for s in range(n_steps):
        seen.clear()
        with ref_shims.DrawRecorder() as rec:
            step = sac.train()
        assert step == s + 1, (step, s)
        out[f'step{s}/u'] = rec.u[0]
        for j, e in enumerate(rec.eps):
            out[f'step{s}/eps{j}'] = e.numpy()
        out[f'step{s}/n_eps'] = np.int64(len(rec.eps))
        out[f'step{s}/perm'] = (torch.stack(rec.perm).numpy() if rec.perm else np.zeros((0, 0), np.int64))
        out[f'step{s}/sample_ids'] = seen['sample_ids']
        out[f'step{s}/is_weights'] = seen['w']
        out[f'step{s}/loss_q'] = seen['loss_q']
        if 'loss_policy' in seen:
            out[f'step{s}/loss_policy'] = seen['loss_policy']
        if s == 0:
            # Adam's first moment after the first step is (1 - beta1) * gradient: the step's gradients, per
            # optimizer and parameter (in `parameters()` order), without touching the step
            for oname, opt in sac.ckpt_dict.items():
                if oname.startswith('optimizer') and opt is not None:
                    for j, p in enumerate(opt.param_groups[0]['params']):
                        st = opt.state.get(p)
                        if st:
                            out[f'g0/{oname}/{j}'] = st['exp_avg'].detach().numpy().copy()
        if seen.get('c_ent') is not None:
            out[f'step{s}/c_entropy'] = seen['c_ent']
        if seen.get('d_ent') is not None:
            out[f'step{s}/d_entropy'] = seen['d_ent']
        if 'td' in seen:
            out[f'step{s}/td_error'] = seen['td']
        out[f'step{s}/tree'] = sac.replay_buffer._sum_tree._tree.copy()
        out[f'step{s}/mu_prob'] = sac.replay_buffer._trans_storage._buffer['mu_prob'].copy()
        out[f'step{s}/hidden'] = sac.replay_buffer._trans_storage._buffer['pre_seq_hidden_state'].copy()
        out[f'step{s}/log_c_alpha'] = sac.log_c_alpha.detach().numpy().copy()
        out[f'step{s}/log_d_alpha'] = sac.log_d_alpha.detach().numpy().copy()
        if 'loss_curiosity' in seen:
            out[f'step{s}/loss_curiosity'] = seen['loss_curiosity']
        for kk, v in seen.items():
            if kk.startswith('w_rq/'):
                out[f'step{s}/{kk}'] = v
    for name, m in mods.items():
        for k, v in m.state_dict().items():
            out[f'w1/{name}/{k}'] = v.numpy().copy()
    for name, t in tensors.items():
        out[f'w1/t/{name}'] = sac.ckpt_dict[name].detach().numpy().copy()
    out['n_steps'] = np.int64(n_steps)
    out['torch_version'] = np.array(torch.__version__)
    out['numpy_version'] = np.array(np.__version__)
    sac.close()
    np.savez_compressed(HERE / f'f6_step_{case}.npz', **out)


def f7_attention():
    """Reference attention layers: weights + inputs -> outputs for the three hidden-state modes and a
    spread of positional encodings / gates / head counts / masks."""
    from algorithm.nn_models.layers.seq_layers import (GATE, POSITIONAL_ENCODING, EpisodeMultiheadAttention,
                                                     MultiheadAttention)
    out = {}
    rng = np.random.default_rng(7)
    cases = {
        'plain': dict(embed_dim=8),
        'rope_res_ln': dict(embed_dim=8, num_layers=3, num_heads=2, pe=POSITIONAL_ENCODING.ROPE,
                            gate=GATE.RESIDUAL, use_layer_norm=True),
        'rope2_out': dict(embed_dim=8, num_layers=2, num_heads=[1, 4], pe=POSITIONAL_ENCODING.ROPE2, gate=GATE.OUTPUT),
        'abs_rec': dict(embed_dim=6, num_layers=2, num_heads=2, pe=POSITIONAL_ENCODING.ABSOLUTE, gate=GATE.RECURRENT,
                        qkv_dense_depth=1),
        'abscat_cat': dict(embed_dim=4, num_layers=2, pe=POSITIONAL_ENCODING.ABSOLUTE_CAT, gate=GATE.CAT),
        'single': dict(embed_dim=8, num_layers=1, num_heads=2, pe=POSITIONAL_ENCODING.ROPE),
    }
    B, K, Q = 5, 7, 3
    for tag, kw in cases.items():
        seed_all(70)
        attn = EpisodeMultiheadAttention(**kw)
        E = kw['embed_dim']
        key = torch.from_numpy(rng.standard_normal((B, K, E)).astype(np.float32))
        index = torch.from_numpy(np.stack([np.arange(s, s + K) for s in rng.integers(0, 20, B)]).astype(np.int32))
        pad = torch.zeros(B, K, dtype=torch.bool)
        pad[0, :2] = True
        pad[1, -2:] = True
        pad[2, :] = True           # fully padded row: must give zeros, not NaN
        for k_, v in attn.named_parameters():      # positional-encoding tables are buffers: deterministic, not stored
            out[f'{tag}/w/{k_}'] = v.detach().numpy().copy()
        out[f'{tag}/key'], out[f'{tag}/index'], out[f'{tag}/pad'] = key.numpy(), index.numpy(), pad.numpy()
        with torch.no_grad():
            y, h, w = attn(key, seq_q_len=Q, key_index=index, key_padding_mask=pad)
            out[f'{tag}/A/y'], out[f'{tag}/A/h'] = y.numpy(), h.numpy()
            for i, wi in enumerate(w):
                out[f'{tag}/A/w{i}'] = wi.numpy()
            y, h, w = attn(key, seq_q_len=K, cut_query=True, key_index=index, key_padding_mask=pad)
            out[f'{tag}/A_full/y'], out[f'{tag}/A_full/h'] = y.numpy(), h.numpy()
            hd = attn.output_hidden_state_dim
            # training mode: one stored state per window (reference sac_base.py:1149-1155)
            hs1 = torch.from_numpy(rng.standard_normal((B, 1, hd)).astype(np.float32))
            y, h, w = attn(key, seq_q_len=K, hidden_state=hs1, is_prev_hidden_state=True, key_index=index,
                           key_padding_mask=pad)
            out[f'{tag}/C/hs'], out[f'{tag}/C/y'], out[f'{tag}/C/h'] = hs1.numpy(), y.numpy(), h.numpy()
            # acting mode: a history of states for the positions before the query (1064-1070)
            hsK = torch.from_numpy(rng.standard_normal((B, K, hd)).astype(np.float32))
            y, h, w = attn(key, seq_q_len=1, hidden_state=hsK, is_prev_hidden_state=False, key_index=index,
                           key_padding_mask=pad)
            out[f'{tag}/B/hs'], out[f'{tag}/B/y'], out[f'{tag}/B/h'] = hsK.numpy(), y.numpy(), h.numpy()
            y, h, w = attn(key, seq_q_len=Q, query_only_attend_to_rest_key=True, key_index=index)
            out[f'{tag}/R/y'], out[f'{tag}/R/h'] = y.numpy(), h.numpy()
    seed_all(71)
    mha = MultiheadAttention(8, num_heads=2, pe=POSITIONAL_ENCODING.ROPE2, out_dense_depth=1, out_size=5)
    q = torch.from_numpy(rng.standard_normal((2, 3, 4, 8)).astype(np.float32))
    k = torch.from_numpy(rng.standard_normal((2, 3, 6, 8)).astype(np.float32))
    kpm = torch.from_numpy(rng.random((2, 3, 6)) < 0.3)
    for k_, v in mha.named_parameters():
        out[f'mha/w/{k_}'] = v.detach().numpy().copy()
    with torch.no_grad():
        y, w = mha(q, k, k, key_padding_mask=kpm)
    out['mha/q'], out['mha/k'], out['mha/kpm'], out['mha/y'], out['mha/wts'] = q.numpy(), k.numpy(), kpm.numpy(), y.numpy(), w.numpy()
    np.savez_compressed(HERE / 'f7_attention.npz', **out)


AUX_PLUGIN = str(HERE.parent / 'plugins' / 'nn_vec_full.py')


def aux_cases():
    """optional learner heads on the all-heads test plugin"""
    from algorithm.utils.enums import CURIOSITY, SIAMESE
    tiny = dict(batch_size=16, replay_config={'capacity': 256}, n_step=3)
    eps = [40, 30, 50]
    f6_step('aux_curiosity', AUX_PLUGIN, dict(curiosity=CURIOSITY.FORWARD, **tiny), eps, 2)
    f6_step('aux_rnd', AUX_PLUGIN, dict(use_rnd=True, **tiny), eps, 2, d_action_sizes=(3,), c_action_size=2)
    f6_step('aux_norm', AUX_PLUGIN, dict(use_normalization=True, **tiny), eps, 2)
    f6_step('aux_dqn', AUX_PLUGIN, dict(discrete_dqn_like=True, **tiny), eps, 2, d_action_sizes=(3, 2), c_action_size=0)
    f6_step('aux_atc', AUX_PLUGIN, dict(siamese=SIAMESE.ATC, siamese_use_q=True, burn_in_step=2, **tiny), eps, 2)
    f6_step('aux_byol', AUX_PLUGIN, dict(siamese=SIAMESE.BYOL, siamese_use_q=True, siamese_use_adaptive=True,
                                          burn_in_step=2, **tiny), eps, 2)


IMG_OBS = dict(obs_names=('vector', 'image'), obs_shapes=((10,), (3, 30, 30)), c_action_size=4)


def conv_cases():
    """BASELINE configs[3] / configs[4] compositions, scaled down: the reference's own conv plugins
    (`tests/nn_conv_vanilla.py`, `tests/nn_conv_attn.py`), burn_in_step 5 / n_step 3 as in its
    `tests/test_sac_params.py:75-76`.  Two of four critics sampled: the two randperm subsets of `_get_y`
    (sac_base.py:1434-1442) are live; FORWARD curiosity: the bonus lands twice in the aliased reward window
    (1333-1343 through 2223)."""
    from algorithm.utils.enums import CURIOSITY
    tiny = dict(batch_size=16, replay_config={'capacity': 256}, burn_in_step=5, n_step=3)
    f6_step('conv', 'tests/nn_conv_vanilla.py', dict(ensemble_q_num=4, ensemble_q_sample=2, **tiny),
            [40, 30, 50], 2, **IMG_OBS)
    f6_step('conv_attn_cur', 'tests/nn_conv_attn.py',
            dict(seq_encoder=SEQ_ENCODER.ATTN, curiosity=CURIOSITY.FORWARD, **tiny), [40, 30, 50, 9], 2, **IMG_OBS)


def conv84_case():
    """The frame size of the reference's environments (`ConvLayers(84, 84, C, 'simple')`: every 84 x 84 plugin under
    `envs/uav`, `envs/ugv`, `envs/roller`) in a recorded train step sequence: the reference's Conv2d stack against the
    product's tiled convolution kernels.  The plugin is this repository's tests/plugins/nn_conv84_small.py (plugin API only,
    so it loads under the reference; a small head keeps the fixture at a few MB — `envs/uav/uav_search/nn.py` itself records
    10 MB of weights)."""
    tiny = dict(batch_size=8, replay_config={'capacity': 128}, burn_in_step=2, n_step=3)
    f6_step('conv84', str(HERE.parent / 'plugins' / 'nn_conv84_small.py'), tiny, [24, 20], 2,
            obs_names=('vector', 'image'), obs_shapes=((10,), (3, 84, 84)), c_action_size=4)


def wide_cases():
    """The recurrent and attention widths of the reference's environments (`m.GRU(.., 64, 1)`:
    envs/square/memory_corridor/nn.py:19; `EpisodeMultiheadAttention(64, 2 layers, 8 heads)`:
    envs/gym/toy_queue/nn_attn.py:28-45) in recorded train steps: the product's wide-GRU and multi-head attention
    kernels against the reference's own modules.  Plugins: this repository's tests/plugins/nn_rnn_h64.py /
    nn_attn_h64.py (plugin API only: they load under the reference)."""
    small = dict(batch_size=16, replay_config={'capacity': 256})
    f6_step('rnn_h64', str(HERE.parent / 'plugins' / 'nn_rnn_h64.py'),
            dict(n_step=3, burn_in_step=3, seq_encoder=SEQ_ENCODER.RNN, **small), [40, 30, 50, 12], 2)
    f6_step('attn_h64', str(HERE.parent / 'plugins' / 'nn_attn_h64.py'),
            dict(n_step=3, burn_in_step=4, seq_encoder=SEQ_ENCODER.ATTN, **small), [40, 30, 50, 12], 2)


# ------------------------------------------------------------------------------------------------
def f8_interop():
    """Files the reference writes (`<step>.pth`, `<step>-rb_tree.npy`, `<step>-rb_storage.npz`;
    sac_base.py:654-668, replay_buffer.py:96-111, 220-227, 436-440) after two train steps, and what the
    reference does when it restores them and trains one more step — the product has to load the same files
    and continue the same way."""
    import shutil
    import tempfile
    nn_mod = load_ref_nn('envs/test/nn_rnn.py')
    kw = dict(obs_names=['vector'], obs_shapes=[(6,)], d_action_sizes=[], c_action_size=2, nn=nn_mod, device='cpu',
              batch_size=16, n_step=3, burn_in_step=2, seq_encoder=SEQ_ENCODER.RNN, replay_config={'capacity': 128})
    seed_all(8)
    rng = np.random.default_rng(8)
    tmp = Path(tempfile.mkdtemp())
    sac = SAC_Base(model_abs_dir=tmp, **kw)
    for T in (40, 30, 50, 45):       # 165 rows on a 128-slot ring: the id map has wrapped
        sac.put_episode(**gen_episode(rng, [(6,)], [], 2, tuple(sac.seq_hidden_state_shape), T))
    for _ in range(2):
        sac.train()
    sac.save_model(save_replay_buffer=True)
    sac.close()
    dst = HERE / 'interop'
    dst.mkdir(exist_ok=True)
    for name in ('2.pth', '2-rb_tree.npy', '2-rb_storage.npz'):
        shutil.copy(tmp / 'model' / name, dst / name)

    seed_all(9)
    sac = SAC_Base(model_abs_dir=tmp, **kw)          # restores step 2 and the replay files
    assert sac.get_global_step() == 2
    out, seen = {}, {}
    orig_update = sac.replay_buffer.update
    sac.replay_buffer.update = lambda ids, td: (seen.update(ids=np.array(ids).copy(), td=np.array(td).copy()),
                                                orig_update(ids, td))[1]
    orig_sample = sac.replay_buffer.sample

    def smp():
        r = orig_sample()
        seen['sample_ids'], seen['w'] = np.array(r[0]).copy(), r[2].numpy().copy()
        return r

    sac.replay_buffer.sample = smp
    orig_rq = sac._train_rep_q
    sac._train_rep_q = lambda *a, **k: (lambda r: (seen.update(loss_q=r[0].detach().numpy().copy()), r)[1])(orig_rq(*a, **k))
    with ref_shims.DrawRecorder() as rec:
        assert sac.train() == 3
    out['u'] = rec.u[0]
    for j, e in enumerate(rec.eps):
        out[f'eps{j}'] = e.numpy()
    out['n_eps'] = np.int64(len(rec.eps))
    out['perm'] = torch.stack(rec.perm).numpy()
    out['sample_ids'], out['is_weights'], out['td_error'], out['loss_q'] = seen['sample_ids'], seen['w'], seen['td'], seen['loss_q']
    out['tree'] = sac.replay_buffer._sum_tree._tree.copy()
    out['mu_prob'] = sac.replay_buffer._trans_storage._buffer['mu_prob'].copy()
    out['hidden'] = sac.replay_buffer._trans_storage._buffer['pre_seq_hidden_state'].copy()
    out['log_c_alpha'] = sac.log_c_alpha.detach().numpy().copy()
    for name, m in sac.ckpt_dict.items():
        if isinstance(m, torch.nn.Module):
            for k, v in m.state_dict().items():
                out[f'w1/{name}/{k}'] = v.numpy().copy()
    sac.close()
    shutil.rmtree(tmp)
    np.savez_compressed(HERE / 'f8_interop.npz', **out)


# ------------------------------------------------------------------------------------------------
def f9_acting():
    """`choose_action` / `choose_attn_action` (sac_base.py:968-1086): inputs, weights, the Gaussian draw of
    `c_policy.sample()` -> action, probability, next hidden state.  Vector, RNN and attention agents;
    sampled and deterministic."""
    out = {}
    rng = np.random.default_rng(9)
    n_env = 5
    cases = {'vec': ('envs/test/nn.py', dict()),
             'rnn': ('envs/test/nn_rnn.py', dict(seq_encoder=SEQ_ENCODER.RNN, burn_in_step=3)),
             'attn': ('envs/test/nn_attn.py', dict(seq_encoder=SEQ_ENCODER.ATTN, burn_in_step=4))}
    for tag, (nn_rel, kw) in cases.items():
        seed_all(90)
        sac = _tiny_sac(load_ref_nn(nn_rel), n_step=3, **kw)
        for name, m in sac.ckpt_dict.items():
            if isinstance(m, torch.nn.Module):
                for k, v in m.state_dict().items():
                    out[f'{tag}/w0/{name}/{k}'] = v.numpy().copy()
        out[f'{tag}/w0/log_c_alpha'] = sac.log_c_alpha.detach().numpy().copy()
        out[f'{tag}/w0/log_d_alpha'] = sac.log_d_alpha.detach().numpy().copy()
        hs = tuple(sac.seq_hidden_state_shape)
        if tag == 'attn':
            T = 7                      # episode so far; the window keeps the last burn_in_step positions
            args = dict(ep_indexes=np.tile(np.arange(T, dtype=np.int32), (n_env, 1)),
                        ep_padding_masks=np.zeros((n_env, T), dtype=bool),
                        ep_obses_list=[rng.standard_normal((n_env, T, 6)).astype(np.float32)],
                        ep_pre_actions=rng.random((n_env, T, 2)).astype(np.float32),
                        ep_pre_attn_states=rng.standard_normal((n_env, T, *hs)).astype(np.float32))
            args['ep_padding_masks'][0, :5] = True
            args['ep_indexes'][0, :5] = -1
            fn = sac.choose_attn_action
        else:
            args = dict(obs_list=[rng.standard_normal((n_env, 6)).astype(np.float32)],
                        pre_action=rng.random((n_env, 2)).astype(np.float32),
                        pre_seq_hidden_state=rng.standard_normal((n_env, *hs)).astype(np.float32))
            fn = sac.choose_action
        for k, v in args.items():
            if isinstance(v, list):
                out[f'{tag}/in/{k}'] = v[0]
            else:
                out[f'{tag}/in/{k}'] = v
        for mode, extra in (('sample', {}), ('deter', dict(disable_sample=True))):
            with ref_shims.DrawRecorder() as rec:
                action, prob, hidden = fn(**{k: (list(v) if isinstance(v, list) else v.copy()) for k, v in args.items()},
                                          **extra)
            out[f'{tag}/{mode}/n_eps'] = np.int64(len(rec.eps))
            for j, e in enumerate(rec.eps):
                out[f'{tag}/{mode}/eps{j}'] = e.numpy()
            out[f'{tag}/{mode}/action'], out[f'{tag}/{mode}/prob'], out[f'{tag}/{mode}/hidden'] = action, prob, hidden
        sac.close()
    np.savez_compressed(HERE / 'f9_acting.npz', **out)


def f10_agent():
    """Agent-side episode assembly (agent.py:21-690): the reference's `AgentManager` driven by `agent_script`
    through get_action / end_episode, the Gaussian draw of every step recorded -> actions per step, every episode
    it hands to `put_episode` (all seven arrays), the agents' statistics at the end."""
    from algorithm.agent import AgentManager
    out = {}
    for tag, (nn_rel, _, kw, max_len) in AGENT_CASES.items():
        kw = dict(kw)
        if 'seq_encoder' in kw:
            kw['seq_encoder'] = SEQ_ENCODER[kw['seq_encoder']]
        seed_all(100)
        sac = _tiny_sac(load_ref_nn(nn_rel), n_step=3, **kw)
        for name, m in sac.ckpt_dict.items():
            if isinstance(m, torch.nn.Module):
                for k, v in m.state_dict().items():
                    out[f'{tag}/w0/{name}/{k}'] = v.numpy().copy()
        out[f'{tag}/w0/log_c_alpha'] = sac.log_c_alpha.detach().numpy().copy()
        out[f'{tag}/w0/log_d_alpha'] = sac.log_d_alpha.detach().numpy().copy()
        mgr = AgentManager('test', ['vector'], [(6,)], [np.float32], [], 2, max_episode_length=max_len, hit_reward=1)
        mgr.set_rl(sac)
        n_ep = 0

        def drain(t):
            nonlocal n_ep
            for ep in mgr.get_tmp_episode_trans_list():
                out[f'{tag}/ep{n_ep}/step'] = np.int64(t)
                for k, v in ep.items():
                    out[f'{tag}/ep{n_ep}/{k}'] = (v[0] if isinstance(v, list) else v).copy()
                n_ep += 1
            mgr.clear_tmp_episode_trans_list()

        script = agent_script(tag)
        for t, st in enumerate(script):
            with ref_shims.DrawRecorder() as rec:
                d_action, c_action = mgr.get_action(st['agent_ids'], [st['obs'].copy()], st['last_reward'].copy())
            assert len(rec.eps) == 1
            out[f'{tag}/t{t}/eps'] = rec.eps[0].numpy()
            out[f'{tag}/t{t}/c_action'] = c_action.copy()
            m = st['term']
            mgr.end_episode(st['agent_ids'][m], [st['term_obs'][m]], st['term_reward'][m], st['term_max'][m])
            drain(t)
        last = script[-1]
        n = len(last['agent_ids'])
        mgr.end_episode(last['agent_ids'], [last['term_obs']], last['term_reward'], np.ones(n, dtype=bool),
                        force_terminated=True)
        mgr.force_end_all_episodes()
        drain(len(script))
        out[f'{tag}/n_episodes'] = np.int64(n_ep)
        ids = sorted(mgr.agents_dict)
        out[f'{tag}/final/agent_ids'] = np.asarray(ids)
        for f in ('steps', 'reward', 'done', 'max_reached', 'force_terminated', 'hit', 'current_step'):
            out[f'{tag}/final/{f}'] = np.asarray([getattr(mgr.agents_dict[i], f) for i in ids], dtype=np.float64)
        out[f'{tag}/final/liveness'] = np.asarray([mgr.agents_liveness[i] for i in ids])
        sac.close()
    np.savez_compressed(HERE / 'f10_agent.npz', **out)


def f11_rpm():
    """`_train_rpm` (sac_base.py:1798-1839) with `calculate_adaptive_weights` (1607-1631) driven DIRECTLY on freshly
    computed states.  The reference's whole step cannot be recorded with `use_prediction`: `_train_rep_q` has freed
    the representation's graph when `_train_rpm` differentiates it again ("backward through the graph a second
    time"; with the parameter-free `ModelSimpleRep`, "grad requires non-empty inputs").  Called on its own graph —
    the way f4 drives `_get_y` — the function runs.  Two variants per learner: main gradient g and -g, so every
    auxiliary loss is seen once with gate 1 and once with gate 0."""
    from torch import autograd
    from torch.nn import functional
    nn_mod = load_ref_nn(AUX_PLUGIN)
    out = {}
    B, n = 16, 3
    for tag, flip, kw in [('a', 1., {}), ('b', -1., {}), ('c', 1., dict(transition_kl=0.3, use_extra_data=False))]:
        seed_all(11)
        rng = np.random.default_rng(11)
        sac = SAC_Base(obs_names=['vector'], obs_shapes=[(6,)], d_action_sizes=[], c_action_size=2, model_abs_dir=None,
                       nn=nn_mod, device='cpu', batch_size=B, n_step=n, use_prediction=True,
                       replay_config={'capacity': 256}, **kw)
        heads = {k: getattr(sac, k) for k in ('model_rep', 'model_target_rep', 'model_transition', 'model_reward',
                                               'model_observation')}
        for name, m in heads.items():
            for k, v in m.state_dict().items():
                out[f'{tag}/w0/{name}/{k}'] = v.numpy().copy()
        obs = rng.standard_normal((B, n + 1, 6)).astype(np.float32)
        actions = rng.random((B, n, 2)).astype(np.float32)
        rewards = rng.standard_normal((B, n)).astype(np.float32)
        coef = rng.standard_normal((8,)).astype(np.float32)
        nx_obses_list = [torch.from_numpy(obs.copy())]
        nx_states, _ = sac.model_rep(nx_obses_list, None, None)
        with torch.no_grad():
            nx_target_states, _ = sac.model_target_rep(nx_obses_list, None, None)
        # the "main" loss whose gradient the gates compare against (the Q loss in a real step)
        main = flip * torch.mean(torch.square(torch.sum(nx_states * torch.from_numpy(coef), dim=-1)))
        sac.optimizer_rep.zero_grad()
        main.backward(retain_graph=True)
        grads_rep_main = [p.grad.detach() for p in sac.model_rep.parameters()]      # (aliases .grad, as in 1577)
        for j, g_ in enumerate(grads_rep_main):
            out[f'{tag}/g_main/{j}'] = g_.numpy().copy()
        seen = {}
        orig = sac.calculate_adaptive_weights

        def spy(grads_main, loss_list, model):
            seen['losses'] = np.array([float(l_) for l_ in loss_list], dtype=np.float32)
            flat_main = torch.cat([g_.reshape(1, -1) for g_ in grads_main], dim=1)
            cos = []
            for l_ in loss_list:
                ga = autograd.grad(l_, list(model.parameters()), allow_unused=True, retain_graph=True)
                ga = [a if a is not None else torch.zeros_like(m_) for m_, a in zip(grads_main, ga)]
                cos.append(float(functional.cosine_similarity(flat_main, torch.cat([a.reshape(1, -1) for a in ga], dim=1))))
            seen['cos'] = np.array(cos, dtype=np.float32)
            return orig(grads_main, loss_list, model)

        sac.calculate_adaptive_weights = spy
        ret = sac._train_rpm(grads_rep_main, nx_obses_list, nx_states, nx_target_states, torch.from_numpy(actions.copy()),
                             torch.from_numpy(rewards.copy()))
        out[f'{tag}/obs'], out[f'{tag}/actions'], out[f'{tag}/rewards'], out[f'{tag}/coef'] = obs, actions, rewards, coef
        out[f'{tag}/flip'] = np.float32(flip)
        out[f'{tag}/nx_states'] = nx_states.detach().numpy().copy()
        out[f'{tag}/nx_target_states'] = nx_target_states.numpy().copy()
        out[f'{tag}/losses'] = seen['losses']            # transition (incl. KL), reward / n_step, observation / n_step
        out[f'{tag}/cos'] = seen['cos']
        out[f'{tag}/gate'] = (np.sign(seen['cos']).clip(min=0)).astype(np.float32)
        out[f'{tag}/ret'] = np.array([float(r_) for r_ in ret], dtype=np.float32)   # mean entropy, loss_reward, loss_obs
        for j, p_ in enumerate(sac.model_rep.parameters()):
            out[f'{tag}/g_rep_after/{j}'] = p_.grad.numpy().copy()
        pred = list(sac.model_transition.parameters()) + list(sac.model_reward.parameters()) + list(sac.model_observation.parameters())
        for j, p_ in enumerate(pred):
            out[f'{tag}/g_pred/{j}'] = p_.grad.numpy().copy()
        for name in ('model_transition', 'model_reward', 'model_observation'):
            for k, v in heads[name].state_dict().items():
                out[f'{tag}/w1/{name}/{k}'] = v.numpy().copy()
        out[f'{tag}/cfg'] = np.array([B, n, kw.get('transition_kl', 0.8), float(kw.get('use_extra_data', True))])
        sac.close()
    assert set(out['a/gate'].tolist()) | set(out['b/gate'].tolist()) == {0., 1.}
    np.savez_compressed(HERE / 'f11_rpm.npz', **out)


def attn_tanh_case():
    small = dict(batch_size=32, replay_config={'capacity': 512})
    f6_step('attn_tanh', str(HERE.parent / 'plugins' / 'nn_attn_tanh.py'),
            dict(n_step=3, burn_in_step=4, seq_encoder=SEQ_ENCODER.ATTN, **small), [60, 45, 70, 12], 3)


def main():
    torch.set_num_threads(1)
    if len(sys.argv) > 1:
        for name in sys.argv[1:]:
            globals()[name]()
        return
    f1_sumtree()
    f2_per()
    f3_vtrace()
    f4_get_y()
    f5_polyak()
    f7_attention()
    small = dict(batch_size=32, replay_config={'capacity': 512})
    # cfg1: n_step 1, use_priority false (BASELINE.json configs[0], scaled down)
    f6_step('cfg1', 'envs/test/nn.py', dict(n_step=1, use_priority=False, **small), [60, 45, 70], 3)
    # cfg2: PER + n_step 4 V-trace (configs[1], scaled down; ring wraps: 9 episodes > 512 rows)
    f6_step('cfg2', 'envs/test/nn.py', dict(n_step=4, **small), [60, 45, 70, 80, 33, 90, 64, 77, 58], 4)
    # cfg3: RNN burn-in (configs[2], scaled down)
    f6_step('cfg3', 'envs/test/nn_rnn.py', dict(n_step=3, burn_in_step=3, seq_encoder=SEQ_ENCODER.RNN, **small),
            [60, 45, 70, 12], 3)
    # ATTN representation (configs[4]'s sequence encoder, scaled down)
    f6_step('attn', 'envs/test/nn_attn.py', dict(n_step=3, burn_in_step=4, seq_encoder=SEQ_ENCODER.ATTN, **small),
            [60, 45, 70, 12], 3)
    # ... and with a bounded state (Linear + tanh head behind the attention, this repository's tests/plugins/nn_attn_tanh.py:
    # plugin API only, so it loads under the reference): the well-conditioned ATTN case
    f6_step('attn_tanh', str(HERE.parent / 'plugins' / 'nn_attn_tanh.py'),
            dict(n_step=3, burn_in_step=4, seq_encoder=SEQ_ENCODER.ATTN, **small), [60, 45, 70, 12], 3)
    # discrete + continuous actions, ensemble 3 of 2 sampled
    f6_step('hybrid', 'envs/test/nn.py', dict(n_step=3, ensemble_q_num=3, ensemble_q_sample=2, **small),
            [60, 45, 70], 3, d_action_sizes=(3, 2), c_action_size=2)
    aux_cases()
    conv_cases()
    conv84_case()
    wide_cases()
    f8_interop()
    f9_acting()
    f10_agent()
    f11_rpm()
    print('golden fixtures written to', HERE)


if __name__ == '__main__':
    main()
