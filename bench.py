"""bench.py — SAC train steps/sec (PER sample + grad step, batch 256) on MI355X.

Workload = BASELINE.json configs[1]: env_type TEST-like synthetic transitions (vector obs 6,
continuous action 2, stock MLP Q / policy), PER capacity 524288, n_step 4 V-trace, batch 256,
buffer pre-filled with 2^18 transitions (episode length 100) and priorities randomised by one
update pass (SURVEY.md §8d "Synthetic inputs").  One "step" = one `SAC_Base.train()`:
PER sample of 256 windows + every gradient/optimizer step + Polyak + priority / mu-prob write-back.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--scaling strong|weak]
N > 1: one rank per GPU over RCCL — under torch.distributed.run, or spawned by this script itself when it is started
plainly with --gpus N.  Every rank owns a replay shard (capacity 524288/N); gradients are mean all-reduced.
  strong (default) the GLOBAL batch stays the BASELINE's 256, 256/N rows per rank: `value` = global-batch
         train steps per second — the BASELINE metric at every N
  weak   every rank samples its own batch of 256 (global batch 256 N): `value` = batch-256 steps of all ranks per second
`value` is always the rate of plain `SAC_Base.train()` calls (one call, one hipGraph launch, one step — the reference's
loop); `value_runs_of_4` beside it is the same work issued as `SAC_Base.train_steps(4)` (one graph launch per four steps).

Output (rank 0).  The LAST line of stdout is ONE compact JSON line under 4 KB — what the driver parses: the contract fields,
`config`, ONE `roofline` (the dominant kernel), ONE `roofline_hbm` (K1-K4), `cpu_baseline`, the side configurations as bare
rates (`compact_line`, tests/test_bench_line.py).  The full record goes to bench_details.json (repo root and gpurun_out/):
  roofline      the dominant KERNEL of the step (launches grouped by kernel name): algorithmic flops or bytes / time vs the
                gfx950 peak, PMC traffic.  `frac` / `achieved` / `avg_launch_us` are the IN-SITU figures — the kernel's mean
                duration inside the replayed hipGraph step, from the committed rocprofv3 --kernel-trace --stats summary of this
                command (profiles/r05_<config>_kernel_stats.json, named in `frac_source`); the live HIP-event figures (each
                launch re-issued 20x back to back on the launch stream: warm caches, optimistic) sit beside them as
                `*_hip_events`
  roofline_hbm  the north-star's "sample + return" kernels K1-K4 (sample + IS weights, window gather, return / min / V)
                as one group, same convention
  sweep         the same kernels at saturating sizes (tools/kernel_sweep.py), where the HBM fraction is meaningful
  configs       the other BASELINE configurations and the side configurations (cfg3 / cfg4 / cfg4_84 / cfg5 / ..., the
                headline with one batch in flight, the headline on one rank with every collective forced), each in its own
                process with ITS rooflines and CPU baseline
  kernels       the per-entry-point accounting for every libasac_hip launch of the step
  cpu_baseline  the CPU oracle (`oracle/sac_ref.py`, a port of the reference's step) timed on this
                host's cores on the same workload (bounded sample)
`--emit full` prints the full record as the one line instead (what the side runs and tools/ read).
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s achievable
MFMA_F32_PEAK_TFLOPS = 157.3   # dense f32-input MFMA peak (v_mfma_f32_16x16x4_f32), same guide

CONFIGS = {
    # BASELINE.json configs[0] at the metric's batch size — the reference's CPU-runnable plumbing case: n_step 1, no
    # priorities (IS weights dropped, the tree sampled but never updated: sac_base.py:2494, 2571-2584)
    'cfg1': dict(obs_names=['vector'], obs_shapes=[(6,)], d_action_sizes=[], c_action_size=2, plugin='nn_vec',
                 n_step=1, burn_in_step=0, batch_size=256, ensemble_q_num=2, ensemble_q_sample=2, capacity=524288,
                 fill=2 ** 18, episode_len=100, hidden=(0,), seq_encoder=None, use_priority=False,
                 desc='cfg1: TEST vector obs(6) c_action(2) stock MLP, n_step=1, use_priority=false'),
    # BASELINE.json configs[1] — the metric's configuration
    'cfg2': dict(obs_names=['vector'], obs_shapes=[(6,)], d_action_sizes=[], c_action_size=2, plugin='nn_vec',
                 n_step=4, burn_in_step=0, batch_size=256, ensemble_q_num=2, ensemble_q_sample=2, capacity=524288,
                 fill=2 ** 18, episode_len=100, hidden=(0,), seq_encoder=None,
                 desc='cfg2: TEST vector obs(6) c_action(2) stock MLP, PER capacity 524288, n_step=4 V-trace'),
    # configs[2] — R2D2-style RNN: burn-in 40 + 40 train steps (window 81)
    'cfg3': dict(obs_names=['vector'], obs_shapes=[(6,)], d_action_sizes=[], c_action_size=2, plugin='nn_rnn',
                 n_step=40, burn_in_step=40, batch_size=256, ensemble_q_num=2, ensemble_q_sample=2, capacity=524288,
                 fill=2 ** 18, episode_len=200, hidden=(2, 8), seq_encoder='RNN',
                 desc='cfg3: GRU(8)x2 rep, burn_in_step=40 n_step=40 (window 81), PER capacity 524288'),
    # configs[3] — image + vector obs, conv rep, 4 critics (2 sampled), batch 512, burn-in 5 / n-step 3
    'cfg4': dict(obs_names=['vector', 'image'], obs_shapes=[(10,), (3, 30, 30)], d_action_sizes=[], c_action_size=4,
                 plugin='nn_conv', n_step=3, burn_in_step=5, batch_size=512, ensemble_q_num=4, ensemble_q_sample=2,
                 capacity=65536, fill=2 ** 15, episode_len=100, hidden=(0,), seq_encoder=None,
                 desc='cfg4: vector(10)+image(3,30,30) conv rep, ensemble 4 (2 sampled), b=5 n=3, PER capacity 65536'),
    # configs[4] — conv + episodic attention rep, FORWARD curiosity, recurrent prediction models, batch 1024.  (The
    # reference itself raises with `use_prediction` and a trainable representation — its `_train_rpm` differentiates a
    # graph its Q loss has freed; here the graph is kept and the head runs; `_train_rpm` itself is
    # pinned against the reference's, tests/golden/f11_rpm.npz)
    'cfg5': dict(obs_names=['vector', 'image'], obs_shapes=[(10,), (3, 30, 30)], d_action_sizes=[], c_action_size=4,
                 plugin='nn_conv_attn', n_step=3, burn_in_step=5, batch_size=1024, ensemble_q_num=2,
                 ensemble_q_sample=2, capacity=65536, fill=2 ** 15, episode_len=100, hidden=(8,), seq_encoder='ATTN',
                 curiosity='FORWARD', use_prediction=True,
                 desc='cfg5: vector(10)+image(3,30,30) conv + attention rep, FORWARD curiosity, use_prediction '
                      '(transition / reward / observation models), b=5 n=3, PER capacity 65536'),
}
# ... and the same without the prediction heads (the round-1 / round-2 workload of this name: their decoder and losses are
# the plugin's own PyTorch modules — MIOpen transposed convolutions and ~300 elementwise launches per step — and
# dominate the step; kept beside `cfg5` so that the representation / attention / curiosity path can be followed
# across rounds)
CONFIGS['cfg5_without_prediction'] = dict(
    CONFIGS['cfg5'], use_prediction=False,
    desc='cfg5 without use_prediction: vector(10)+image(3,30,30) conv + attention rep, FORWARD curiosity, b=5 n=3, '
         'PER capacity 65536')
# cfg4 on the frame size of the reference's environments (ConvLayers(84, 84, 3, 'simple'): 20 x 20 -> 9 x 9 positions, the
# tiled form of the fused convolution stack); 84.7 KB per frame: capacity and batch sized so that ring and batch stay
# within a few GB
CONFIGS['cfg4_84'] = dict(
    CONFIGS['cfg4'], obs_shapes=[(10,), (3, 84, 84)], plugin='nn_conv84', batch_size=256, capacity=16384, fill=2 ** 13,
    desc='cfg4_84: vector(10)+image(3,84,84) conv rep (the reference environments\' frame size), ensemble 4 (2 sampled), '
         'b=5 n=3, batch 256, PER capacity 16384')
# cfg3 with the recurrent core of the reference's environments (one GRU layer of 64 units: envs/square/memory_corridor/nn.py:19)
CONFIGS['cfg3_h64'] = dict(
    CONFIGS['cfg3'], plugin='nn_rnn_h64', hidden=(1, 64),
    desc='cfg3_h64: GRU(64)x1 rep (the reference environments\' width), burn_in_step=40 n_step=40 (window 81), PER capacity 524288')
# the attention representation at the reference environments' width (64 channels, 8 heads, 2 layers:
# envs/gym/toy_queue/nn_attn.py:28-45) on the TEST observations, windows of 9 as configs[4]
CONFIGS['cfg_attn_h64'] = dict(
    obs_names=['vector'], obs_shapes=[(6,)], d_action_sizes=[], c_action_size=2, plugin='nn_attn_h64', n_step=3,
    burn_in_step=5, batch_size=1024, ensemble_q_num=2, ensemble_q_sample=2, capacity=65536, fill=2 ** 15, episode_len=100,
    hidden=(64,), seq_encoder='ATTN',
    desc='cfg_attn_h64: EpisodeMultiheadAttention(64, 2 layers, 8 heads) rep (the reference environments\' width), b=5 n=3, '
         'batch 1024, PER capacity 65536')
CFG = dict(CONFIGS['cfg2'])


def synthetic_episode(rng, T):
    A = CFG['c_action_size']
    return dict(ep_indexes=np.arange(T, dtype=np.int32)[None],
                ep_obses_list=[rng.standard_normal((1, T, *s)).astype(np.float32) for s in CFG['obs_shapes']],
                ep_actions=rng.random((1, T, A)).astype(np.float32),
                ep_rewards=rng.standard_normal((1, T)).astype(np.float32),
                ep_dones=(rng.random((1, T)) < 0.5),
                ep_probs=rng.random((1, T, A)).astype(np.float32),
                ep_pre_seq_hidden_states=rng.standard_normal((1, T, *CFG['hidden'])).astype(np.float32))


# entry point -> the kernel that does its work (rocprofv3 names; template arguments dropped)
_KERNEL_OF = {'asac_mlp_forward': 'asac::k_mlp_fwd', 'asac_mlp_forward_multi': 'asac::k_mlp_fwd_multi',
              'asac_mlp_backward': 'asac::k_mlp_bwd', 'asac_mlp_backward_qloss': 'asac::k_mlp_bwd',
              'asac_mlp_backward_qloss_return': 'asac::k_mlp_bwd',
              'asac_mlp_backward_policy_q': 'asac::k_mlp_bwd', 'asac_mlp_backward_policy_sample': 'asac::k_mlp_bwd',
              'asac_window_gather_pad': 'asac::k_window_gather_pad', 'asac_vtrace_return_min': 'asac::k_vtrace_return_min',
              'asac_sumtree_sample': 'asac::k_sumtree_sample', 'asac_step_prologue_sample': 'asac::k_prologue_sample',
              'asac_step_prologue_sample_partial': 'asac::k_prologue_sample', 'asac_window_gather_pad_w': 'asac::k_window_gather_pad_w',
              'asac_sumtree_update': 'asac::k_sumtree_update', 'asac_squash_multi': 'asac::k_squash_multi',
              'asac_squash_sample_fwd': 'asac::k_squash_sample_fwd', 'asac_gru_forward': 'asac::k_gru_fwd',
              'asac_gru_backward': 'asac::k_gru_bwd', 'asac_scatter_rows_if_id_match': 'asac::k_scatter_write',
              'asac_conv2_forward': 'asac::k_conv2_fwd', 'asac_conv2_backward': 'asac::k_conv2_bwd',
              'asac_conv2_forward_windows': 'asac::k_conv2_fwd', 'asac_conv2_backward_windows': 'asac::k_conv2_bwd',
              'asac_attention_proj_forward': 'asac::k_attn_proj_fwd', 'asac_attention_proj_backward': 'asac::k_attn_proj_bwd',
              'asac_linear_tanh_forward': 'asac::k_linear_tanh_fwd', 'asac_linear_tanh_backward': 'asac::k_linear_tanh_bwd',
              'asac_linear_tanh_forward2': 'asac::k_linear_tanh_fwd', 'asac_linear_tanh_backward2': 'asac::k_linear_tanh_bwd',
              'asac_gru_backward_at': 'asac::k_gru_bwd',
              'asac_adam_step_partials': 'asac::k_adam_partials', 'asac_adam_step': 'asac::k_adam',
              'asac_step_prologue': 'asac::k_noise_fill', 'asac_policy_sample_q_forward': 'asac::k_pi_sample_q',
              'asac_policy_step_fused': 'asac::k_policy_step', 'asac_rows_move': 'asac::k_rows_move',
              'asac_td_update': 'asac::k_td_update',
              'asac_obs_decoder_forward': 'asac::dec::k_dec_fwd12+asac::dec::k_dec_fwd3',
              'asac_obs_decoder_backward': 'asac::dec::k_dec_bwd3+asac::dec::k_dec_bwd2dx+asac::dec::k_dec_bwd2dw+asac::dec::k_dec_reduce',
              'asac_cosine_gate_add': 'asac::k_cosine_gate_add',
              'asac_attention_mh_forward': 'asac::amh::k_attn_mh', 'asac_attention_mh_backward': 'asac::amh::k_attn_mh',
              'asac_attention_mh_proj_forward': 'asac::amh::k_attn_mh', 'asac_attention_mh_block_backward': 'asac::amh::k_attn_mh',
              'asac_rows_proj_forward': 'asac::rowsp::k_rows_proj_fwd', 'asac_rows_proj_backward': 'asac::rowsp::k_rows_proj_bwd',
              'asac_rows_resblock_forward': 'asac::rowsp::k_rows_res_fwd', 'asac_rows_resblock_backward': 'asac::rowsp::k_rows_res_bwd',
              'asac_rows_affine_forward': 'asac::rowsp::k_rows_affine', 'asac_rows_affine_gelu_forward': 'asac::rowsp::k_rows_affine',
              'asac_xty': 'asac::xty::k_xty', 'asac_xty_multi': 'asac::xty::k_xty_multi',
              'asac_gru_wide_forward_twin': 'asac::gruw::k_gruw_fwd',
              'asac_gru_wide_forward': 'asac::gruw::k_gruw_fwd', 'asac_gru_wide_backward': 'asac::gruw::k_gruw_bwd',
              'asac_normal_nll_kl': 'asac::k_normal_nll_kl', 'asac_normal_nll_kl_logstd': 'asac::k_normal_nll_kl',
              'asac_masked_mse': 'asac::k_masked_mse'}
SAMPLE_RETURN = ('asac_step_prologue_sample', 'asac_step_prologue_sample_partial',
                 'asac_sumtree_sample', 'asac_window_gather_pad', 'asac_window_gather_pad_w', 'asac_vtrace_return_min',
                 'asac_td_update')      # (the TD error's return, formed inside the priority update's launch: K4 + K6)
ROUND = 'r06'


def library_hash() -> str:
    """content hash of the library this process runs (sources + headers + flags: csrc/build.py `source_hash`)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location('asac_build', ROOT / 'advanced-soft-actor-critic_amd' / 'csrc' / 'build.py')
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.source_hash()


def _committed(config: str, kind: str, profiles_dir=None, lib_hash=None):
    """(document, 'profiles/<file>') of this round's committed summary `profiles/<ROUND>_<config>_<kind>.json` — ONLY when
    it was recorded with the very library that is running (its `_meta.lib_hash` stamp, tools/summarize_rocprof.py /
    summarize_pmc.py, equals `library_hash()`); (None, reason) otherwise.  No fallback to older rounds: a kernel change
    without a profile refresh must not print a stale duration beside live work counts."""
    path = Path(profiles_dir or ROOT / 'profiles') / f'{ROUND}_{config}_{kind}.json'
    if not path.exists():
        return None, 'no committed profile'
    d = json.loads(path.read_text())
    stamp = (d.get('_meta') or {}).get('lib_hash')
    if stamp != (lib_hash or library_hash()):
        return None, 'profile stale'
    return d, f'profiles/{path.name}'


def pmc_traffic(config: str, kernel: str, profiles_dir=None, lib_hash=None):
    """(HBM bytes per launch, source) of a kernel (all template instances, launch-weighted), or (None, reason): the PMC
    passes run under rocprofv3, not inside this process, so the committed summary of the same command is read
    (profiles/<round>_<config>_pmc_traffic.json, tools/summarize_pmc.py: FETCH_SIZE x2 (gfx950) + WRITE_SIZE)."""
    d, src = _committed(config, 'pmc_traffic', profiles_dir, lib_hash)
    if d is None:
        return None, src
    recs = [v for k, v in d.items()
            if (k == kernel or k.startswith(kernel + '<') or k.startswith(kernel + '_sc<'))     # (_sc: with sidecars)
            and isinstance(v, dict) and v.get('fetch_bytes_corrected') is not None]
    if not recs:
        return None, 'kernel not in the committed profile'
    calls = sum(r.get('launches', 1) for r in recs) or 1
    tot = sum((r['fetch_bytes_corrected'] + (r.get('write_bytes_raw') or 0.0)) * r.get('launches', 1) for r in recs)
    return round(tot / calls), src


def in_situ(config: str, kernel: str, profiles_dir=None, lib_hash=None):
    """(mean launch duration in us, launches per step, source) of a kernel INSIDE the replayed hipGraph step, from the
    committed rocprofv3 --kernel-trace --stats summary of `bench.py --config <config>` (profiles/<round>_<config>_
    kernel_stats.json, tools/summarize_rocprof.py); `a+b`: a launch group, durations added.  (None, None, reason) when
    there is none for the running library."""
    d, src = _committed(config, 'kernel_stats', profiles_dir, lib_hash)
    if d is None:
        return None, None, src
    us, per_step = 0.0, None
    for k in kernel.split('+'):
        if k not in d:
            return None, None, 'kernel not in the committed profile'
        us += d[k]['avg_us']
        per_step = d[k]['launches_per_step'] if per_step is None else per_step
    return us, per_step, src


def roofline_with_profile(roofline: dict, config: str, kernel: str, profiles_dir=None, lib_hash=None) -> dict:
    """`roofline` as measured LIVE in this run (HIP events on the launch stream) -> the record that is printed: the live
    figures stay as `*_hip_events`; when this round's committed rocprofv3 summary was recorded with the running library
    its in-situ duration (inside the replayed step) becomes `frac` / `achieved` / `avg_launch_us` and is named in
    `frac_source`, otherwise the live figures are THE figures and `frac_source` says why ("live (profile stale)")."""
    r = dict(roofline)
    r['traffic'], r['traffic_source'] = pmc_traffic(config, kernel, profiles_dir, lib_hash)
    r.update({'achieved_hip_events': r['achieved'], 'frac_hip_events': r['frac'], 'avg_launch_us_hip_events': r['avg_launch_us']})
    us_situ, _, src = in_situ(config, kernel, profiles_dir, lib_hash)
    if us_situ is None:
        r['frac_source'] = f'live ({src})'
        return r
    work = r.get('alg_flops_per_launch') or r.get('alg_bytes_per_launch')
    ach = work / (us_situ * 1e-6) / (1e12 if r['bound'] == 'mfma' else 1e9)
    r.update({'achieved': round(ach, 4), 'frac': round(ach / r['peak'], 6), 'avg_launch_us': round(us_situ, 3), 'frac_source': src})
    return r


def _gru_bytes(B, L):
    """Fused GRU stack (cfg3 plugin: input = obs + action, 2 layers x 8): x in, per-layer outputs + saved
    gate activations out (forward); x, saved activations and the output gradient in, dx out (backward)."""
    if CFG['seq_encoder'] != 'RNN':
        return {}
    layers, H = CFG['hidden']
    I = CFG['obs_shapes'][0][0] + CFG['c_action_size']
    return {'asac_gru_forward': B * L * (4 * I + layers * 24 * H + 4 * H),
            'asac_gru_backward': B * L * (8 * I + layers * 20 * H + 4 * H)}


def _conv_bytes(B, L):
    """Fused `simple` convolution stack over the B*L frames of the sampled windows (cfg4 / cfg5 plugins:
    3x30x30 frames -> 16x6x6 -> 32x2x2).  Forward (averaged over the step's passes, one of three saves the
    pre-activations): frames in, 128 features out; backward: frames, saved pre-activations and the output
    gradient in."""
    shapes = [sh for sh in CFG['obs_shapes'] if len(sh) == 3]
    if not shapes:
        return {}
    frame = 4 * int(np.prod(shapes[0]))
    h1 = (shapes[0][1] - 8) // 4 + 1
    h2 = (h1 - 4) // 2 + 1
    z1, z2 = 4 * 16 * h1 * h1, 4 * 32 * h2 * h2
    return {'asac_conv2_forward': B * L * (frame + z2) + B * L * (z1 + z2) // 3,
            'asac_conv2_backward': B * L * (frame + z1 + 2 * z2)}


def algorithmic_bytes(P_polyak, P_seg):
    """Per-launch algorithmic bytes of each hot-path kernel at this workload (SURVEY.md §8d;
    f32 = 4 B).  B batch, L window, T bytes per stored transition, D tree depth."""
    B, n, b, A, E = CFG['batch_size'], CFG['n_step'], CFG['burn_in_step'], CFG['c_action_size'], CFG['ensemble_q_num']
    L, D = b + n + 1, int(np.log2(CFG['capacity']))
    obs_bytes = sum(4 * int(np.prod(sh)) for sh in CFG['obs_shapes'])
    T = 4 + 1 + obs_bytes + 4 * A + 4 + 1 + 4 * A + 4 * int(np.prod(CFG['hidden']))   # index, last_mask, obs, action, reward, done, mu_prob, hidden
    return {
        'asac_sumtree_sample': B * (8 + 8 * D + 8) + 8 * B,              # K1 + K2
        'asac_window_gather_pad': 8 * B + 2 * B * L * T,                 # K3
        'asac_vtrace_return_min': B * (n * (4 + 1 + 1 + 1 + 4 + 4) + E * (n + 1) * 4 + (n + 1) * 4 + 4),   # K4
        'asac_squash_sample_fwd': B * (n + 1) * A * 4 * 4 + B * (n + 1) * 4,   # loc, scale, eps in; a out; logp out
        'asac_squash_prob': B * (n + 1) * A * 4 * 4,
        'asac_polyak': 12 * P_polyak,                                    # K5
        'asac_sumtree_update': B * (4 + 8 + 8 + 4) + 12 * B * D,         # K6
        'asac_td_update': B * (n * (4 + 1 + 1 + 1 + 4 + 4) + E * (n + 1) * 4 + (n + 1) * 4 + 4) + B * (4 * E + 4)   # K4 (TD error)
        + B * (8 + 8 + 4) + 12 * B * D,                                  # + K6 (the TD errors stay in registers)
        'asac_scatter_rows_if_id_match': B * (b + n) * (8 + 4 * A),      # K7 (mu_prob)
        'asac_q_loss_fwd_bwd': E * B * 4 * 3 + B * 8,
        'asac_adam_step': 28 * P_seg,
        **_gru_bytes(B, L),
        **_conv_bytes(B, L),
    }


def build_agent(device, dist_ctx, capacity, seed):
    import importlib
    import asac_amd  # noqa: F401
    from algorithm.sac_base import SAC_Base
    from algorithm.utils.enums import CURIOSITY, SEQ_ENCODER
    plugin = importlib.import_module(f'tests.plugins.{CFG["plugin"]}')
    torch.manual_seed(seed)
    return SAC_Base(CFG['obs_names'], CFG['obs_shapes'], CFG['d_action_sizes'], CFG['c_action_size'], None, plugin,
                    device=device, n_step=CFG['n_step'], burn_in_step=CFG['burn_in_step'],
                    batch_size=CFG['batch_size'], ensemble_q_num=CFG['ensemble_q_num'],
                    ensemble_q_sample=CFG['ensemble_q_sample'],
                    seq_encoder=SEQ_ENCODER[CFG['seq_encoder']] if CFG['seq_encoder'] else None,
                    curiosity=CURIOSITY[CFG['curiosity']] if CFG.get('curiosity') else None,
                    use_priority=CFG.get('use_priority', True), use_prediction=CFG.get('use_prediction', False),
                    replay_config={'capacity': capacity},
                    hip_config={'dist': dist_ctx, **json.loads(os.environ.get('ASAC_BENCH_HIP_CONFIG', '{}'))})


def fill_buffer(agent, rng, n_transitions):
    T = CFG['episode_len']
    for _ in range(n_transitions // T):
        agent.put_episode(**synthetic_episode(rng, T))
    # one randomising priority pass: |N(0,1)| td-errors on every resident row
    rb = agent.replay_buffer
    ids = torch.arange(rb.size, device=rb.device, dtype=torch.int64)
    td = torch.from_numpy(np.abs(rng.standard_normal(rb.size)).astype(np.float32)).to(rb.device)
    for s in range(0, rb.size, 4096):
        rb.update(ids[s:s + 4096], td[s:s + 4096])
    # keep the "last row of each episode / ring tail is never sampled" invariant of add()
    last = ids[T - 1::T]
    rb._update_ids(last, torch.zeros(last.numel(), device=rb.device), stale_check=False, mode=1)
    torch.cuda.synchronize()


def cpu_baseline(budget_s=24.0, fill=None):
    """The oracle port on this host: same workload, bounded sample.  The step is ~40 tiny eager ops
    on [256, <=64] tensors, so more intra-op threads only add synchronisation cost: a short sweep
    picks the best thread count and that one is reported (`cores` = threads actually used)."""
    import importlib
    from oracle import sac_ref
    import asac_amd  # noqa: F401
    plugin = importlib.import_module(f'tests.plugins.{CFG["plugin"]}')
    torch.manual_seed(0)
    np.random.seed(0)
    rng = np.random.default_rng(0)
    agent = sac_ref.SacRef(CFG['obs_names'], CFG['obs_shapes'], [], CFG['c_action_size'], plugin,
                           n_step=CFG['n_step'], burn_in_step=CFG['burn_in_step'], batch_size=CFG['batch_size'],
                           ensemble_q_num=CFG['ensemble_q_num'], ensemble_q_sample=CFG['ensemble_q_sample'],
                           seq_encoder=CFG['seq_encoder'], curiosity=CFG.get('curiosity'),
                           use_priority=CFG.get('use_priority', True), use_prediction=CFG.get('use_prediction', False),
                           replay_config={'capacity': CFG['capacity']})
    fill = CFG['fill'] if fill is None else fill      # the rows the GPU run holds resident
    for _ in range(fill // CFG['episode_len']):
        agent.put_episode(**synthetic_episode(rng, CFG['episode_len']))
    default_threads = torch.get_num_threads()
    candidates = sorted({1, 4, 8, 16, min(32, default_threads)}) if budget_s >= 20 else [1, 8]
    best, sweep = None, {}
    for th in candidates:
        torch.set_num_threads(th)
        for _ in range(3):
            agent.train()
        t0, k = time.perf_counter(), 0
        while time.perf_counter() - t0 < budget_s / (len(candidates) + 2) and k < 500:
            agent.train()
            k += 1
        sweep[th] = k / (time.perf_counter() - t0)
        if best is None or sweep[th] > sweep[best]:
            best = th
    torch.set_num_threads(best)
    t0, k = time.perf_counter(), 0
    while time.perf_counter() - t0 < 2 * budget_s / (len(candidates) + 2) and k < 2000:
        agent.train()
        k += 1
    dt = time.perf_counter() - t0
    torch.set_num_threads(default_threads)
    return {'value': round(k / dt, 3), 'unit': 'train_steps/s', 'cores': best, 'kind': 'port',
            'threads': best, 'host_cores': os.cpu_count(), 'rows_resident': int(fill), 'sample_steps': k,
            'sample_seconds': round(dt, 1), 'thread_sweep_steps_per_s': {str(t): round(v, 1) for t, v in sweep.items()},
            'sample': f'{k} steps of the same workload in {dt:.1f}s, oracle/sac_ref.SacRef, best of a thread sweep'}


# ---- the line the driver parses ---------------------------------------------------------------------------------------------
PROSE_LIMIT = 100      # free-text fields of the line (sizes and counts travel as numbers, never inside prose)
LINE_LIMIT = 4096      # the driver keeps a tail of stdout: the LAST line must fit in it whole (round 4's 21 KB line did not)
_ROOF_KEYS = ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'kernel', 'kernels', 'avg_launch_us', 'us_per_step',
              'launches_per_step', 'alg_flops_per_launch', 'alg_bytes_per_launch', 'alg_bytes_per_step',
              'traffic_over_algorithmic', 'achieved_hip_events', 'frac_hip_events', 'frac_source')


def _clip(text, n):
    text = str(text)
    return text if len(text) <= n else text[:n - 1] + '…'


def compact_line(full: dict, details_path: str = 'bench_details.json') -> str:
    """The ONE JSON line the driver reads (last line of stdout, < LINE_LIMIT bytes): the contract fields, `config`, ONE
    `roofline` (dominant kernel), ONE `roofline_hbm` (K1-K4), `cpu_baseline`, the side configurations as bare rates.
    Everything else (`kernels`, `sweep`, the side configurations' own rooflines) is in `details_path`."""
    keep = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
            'vs_baseline', 'dtype', 'data')
    out = {k: full.get(k) for k in keep}
    for k, v in full.items():
        if k.startswith('value_') and v is not None:
            out[k] = v
    cfg = dict(full.get('config') or {})
    cfg['workload'] = _clip(cfg.get('workload', ''), PROSE_LIMIT)
    out['config'] = {k: cfg[k] for k in ('workload', 'per_gpu_batch', 'global_batch', 'rows_resident', 'replay_capacity',
                                         'replay_shard_capacity', 'n_step', 'burn_in_step', 'parallelism',
                                         'ranks', 'collectives', 'hipgraph') if k in cfg}
    for name in ('roofline', 'roofline_hbm'):
        r = full.get(name)
        out[name] = None if r is None else {k: (_clip(r[k], 80) if isinstance(r[k], str) else r[k])
                                            for k in _ROOF_KEYS if k in r and r[k] is not None or k == 'traffic' and k in r}
    c = full.get('cpu_baseline')
    out['cpu_baseline'] = None if c is None else {**{k: c.get(k) for k in ('value', 'unit', 'cores', 'kind', 'threads', 'host_cores',
                                                                            'rows_resident', 'sample_steps', 'sample_seconds')
                                                     if c.get(k) is not None},
                                                  'sample': _clip(c.get('sample', ''), PROSE_LIMIT)}
    side = {}
    for name, d in (full.get('configs') or {}).items():
        if 'error' in d:
            side[name] = None
            continue
        e = {'steps_per_s': d.get('value')}
        h = d.get('roofline_hbm') or {}
        if h.get('frac') is not None:
            e['k1_k4_hbm_frac'] = h['frac']
        side[name] = e
    if side:
        out['side_configs'] = side
    out['details'] = details_path
    line = json.dumps(out, ensure_ascii=True, separators=(',', ':'))
    if len(line) >= LINE_LIMIT:        # never expected; shed the optional parts rather than lose the headline again
        for k in ('side_configs', 'roofline_hbm'):
            out.pop(k, None)
            line = json.dumps(out, ensure_ascii=True, separators=(',', ':'))
            if len(line) < LINE_LIMIT:
                break
    assert len(line) < LINE_LIMIT, len(line)
    return line


def emit(full: dict, mode: str) -> None:
    """mode 'full': the whole record on one line (what `other_configs` and tools/ read from a child process);
    'compact' (default): the record goes to bench_details.json (repo root and, where present, gpurun_out/), the compact
    line is the last line of stdout."""
    import ctypes
    # libraries (RCCL's version banner) hold text in the C stdio buffer until exit: push it out first so
    # the JSON record is the last line on stdout
    ctypes.CDLL(None).fflush(None)
    sys.stdout.flush()
    if mode == 'full':
        print(json.dumps(full), flush=True)
        return
    text = json.dumps(full, indent=1)
    written = None
    for d in (ROOT, ROOT / 'gpurun_out'):
        try:
            if d.is_dir():
                (d / 'bench_details.json').write_text(text)
                written = written or str((d / 'bench_details.json').relative_to(ROOT))
        except OSError:
            pass
    print(compact_line(full, written or 'not written'), flush=True)


def _free_port() -> int:
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]


def other_configs(names=('cfg2_lookahead', 'cfg2_force_dist', 'cfg3', 'cfg3_h64', 'cfg4', 'cfg4_84', 'cfg5', 'cfg5_without_prediction', 'cfg_attn_h64'), steps=300, warmup=40) -> dict:
    """train steps/s of the other BASELINE configurations, each in its own process (its own replay buffers and
    hipGraph), same timing contract, fewer steps"""
    import subprocess
    out = {}
    for name in names:
        env = dict(os.environ)
        if name.endswith('_lookahead'):       # the headline workload with one sampled batch in flight (hip_config['lookahead'])
            cmd = [sys.executable, str(Path(__file__).resolve()), '--config', name[:-len('_lookahead')], '--steps', '2000',
                   '--warmup', '100', '--no-cpu-baseline', '--profile-steps', '0', '--no-extras', '--run-length', '0', '--emit', 'full']
            env['ASAC_BENCH_HIP_CONFIG'] = json.dumps({**json.loads(env.get('ASAC_BENCH_HIP_CONFIG', '{}')), 'lookahead': 1})
        elif name.endswith('_force_dist'):    # one rank with every collective of the data-parallel step issued (RCCL, captured)
            cmd = [sys.executable, str(Path(__file__).resolve()), '--config', name[:-len('_force_dist')], '--steps', '2000',
                   '--warmup', '100', '--no-cpu-baseline', '--profile-steps', '0', '--no-extras', '--run-length', '0', '--emit', 'full',
                   '--force-dist']
            env.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
        else:
            cmd = [sys.executable, str(Path(__file__).resolve()), '--config', name, '--steps', str(steps), '--warmup',
                   str(warmup), '--cpu-budget', '8', '--profile-steps', '12', '--no-extras', '--run-length', '0', '--emit', 'full']
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1]
            d = json.loads(line)
            out[name] = {'value': d['value'], 'unit': d['unit'], 'ms_per_step': d['ms_per_step'], 'steps': d['steps'],
                         'warmup': d['warmup'], 'workload': d['config']['workload'], 'hipgraph': d['config']['hipgraph'],
                         'ranks': d['config'].get('ranks'), 'collectives': d['config'].get('collectives'),
                         'roofline': d.get('roofline'), 'roofline_hbm': d.get('roofline_hbm'),
                         'cpu_baseline': d.get('cpu_baseline')}
        except Exception as e:   # a failed side run must not lose the main line
            out[name] = {'error': repr(e)[:200]}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2000)
    ap.add_argument('--warmup', type=int, default=100)
    ap.add_argument('--scaling', choices=('weak', 'strong'), default='strong',
                    help='strong (default): global batch 256, 256/N rows per GPU (SURVEY 8d, the BASELINE metric); '
                         'weak: batch 256 per GPU')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-budget', type=float, default=24.0, help='seconds of host time for the CPU port leg')
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--run-length', type=int, default=4,
                    help='the secondary figure `value_runs_of_<k>`: the same steps issued as SAC_Base.train_steps(k) '
                         '(one hipGraph replay holds k steps); 0 or 1 = skip it.  `value` is always plain train() calls')
    ap.add_argument('--no-extras', action='store_true', help='skip the saturating-size sweep and the cfg3-5 side runs')
    ap.add_argument('--profile-steps', type=int, default=50)
    ap.add_argument('--fill', type=int, default=None, help='transitions resident before timing')
    ap.add_argument('--force-dist', action='store_true',
                    help='take the multi-rank code path (process group, RCCL collectives in the step) even with one rank')
    ap.add_argument('--emit', choices=('compact', 'full'), default='compact',
                    help='compact: the < 4 KB line the driver parses + bench_details.json; full: the whole record on one line')
    ap.add_argument('--config', choices=sorted(CONFIGS), default='cfg2',
                    help='cfg2 = the BASELINE metric configuration; the others are informational')
    args = ap.parse_args()
    CFG.clear()
    CFG.update(CONFIGS[args.config])
    if args.fill is None:
        args.fill = CFG['fill']

    # --gpus N started plainly (no launcher): spawn the N ranks ourselves, one per GPU, and hand their line through
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        import subprocess
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
               '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), str(Path(__file__).resolve()), *sys.argv[1:]]
        raise SystemExit(subprocess.call(cmd))

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but the launcher started {world} ranks')
    # ASAC_BENCH_ONE_DEVICE=1 (tests only: tests/test_parallel_world2_gpu.py): every rank on cuda:0, gloo instead of RCCL, no
    # graph capture — the N > 1 HOST path of this script (launcher, shard fill, max-over-ranks timing, rank 0's line) on a
    # one-GPU box.  The line says so (`config.collectives`); it is not a measurement.
    one_device = os.environ.get('ASAC_BENCH_ONE_DEVICE') == '1'
    device = torch.device('cuda', 0 if one_device else local_rank)
    torch.cuda.set_device(device)
    if args.scaling == 'strong':
        if CFG['batch_size'] % world:
            raise SystemExit(f'strong scaling needs the batch ({CFG["batch_size"]}) divisible by the ranks ({world})')
        CFG['batch_size'] //= world

    dist_ctx = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        if one_device:
            dist.init_process_group('gloo')
            args.no_graph = True
        else:
            dist.init_process_group('nccl', device_id=device)
        import asac_amd  # noqa: F401
        from algorithm.parallel import DataParallelContext
        dist_ctx = DataParallelContext(always=args.force_dist)

    from asac_amd import native
    native.load()
    agent = build_agent(device, dist_ctx, CFG['capacity'] // world, seed=0)
    if args.no_graph:
        agent._use_graph = False
    rng = np.random.default_rng(1234 + rank)
    fill_buffer(agent, rng, args.fill // world)

    def sync_all():
        torch.cuda.synchronize()
        if dist_ctx is not None:
            dist_ctx.barrier()
            torch.cuda.synchronize()

    # the hipGraph is captured after `graph_warmup` eager steps: make sure that happens before the W warm-up steps
    # even when the caller asks for very few of them (untimed either way)
    for _ in range(max(0, agent._graph_warmup + 3 - args.warmup)):
        agent.train()
    for _ in range(args.warmup):
        agent.train()
    # EXACTLY `steps` plain train() calls: one call = one step = one graph launch (the reference's loop, the metric)
    def max_over_ranks(seconds):
        if dist_ctx is None:
            return seconds
        t = torch.tensor([seconds], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        return float(t.item())

    # The interpreter's full (generation 2) collections take 45-60 ms over the ~2e5 objects a filled learner holds, and where
    # one falls depends on allocation counts (it fell inside the 0.1 s timed region of every process but the first on a box —
    # byte-code caches change the counts — and read as 2 300 instead of 3 170 steps/s at cfg4, round 6): everything alive
    # after the warm-up is long-lived, so it is collected once here and frozen; the collector stays ON for what the timed
    # steps allocate.
    import gc
    gc.collect()
    gc.freeze()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        agent.train()
    sync_all()
    dt = max_over_ranks(time.perf_counter() - t0)
    # ... and, reported beside it, the same number of steps as runs of k (SAC_Base.train_steps: one graph replay per run;
    # the first run of that length captures its graph, so one is done before the clock starts)
    spl, dt_runs = max(1, args.run_length), None
    if spl > 1:
        agent.train_steps(spl)
        sync_all()
        t1 = time.perf_counter()
        for _ in range(args.steps // spl):
            agent.train_steps(spl)
        for _ in range(args.steps % spl):
            agent.train()
        sync_all()
        dt_runs = max_over_ranks(time.perf_counter() - t1)
    graph_used = agent._graph is not None
    graph_memsets = getattr(agent, '_graph_memsets', None)
    agent.replay_buffer.check_health()

    # ---- per-kernel HIP-event timing of the same step, eager, on the launch stream -------------
    kernels, by_kernel, roofline, roofline_hbm = {}, {}, None, None
    summ = None
    if args.profile_steps > 0:
        # EVERY rank runs these eager steps (their gradient all-reduces are collectives); only rank 0 times
        # its launches (the library re-issues each of ITS kernels 20x; the collectives stay one per step)
        agent._graph, agent._use_graph = None, False
        for _ in range(10):
            agent.train()
        if rank == 0:
            with native.LaunchProfiler(repeat=20) as prof:
                for _ in range(args.profile_steps):
                    agent.train()
            summ = prof.summary()
        else:
            for _ in range(args.profile_steps):
                agent.train()
        sync_all()
    if summ is not None:
        P_polyak = agent._polyak_len
        P_rq = agent._params.span('rep', f'q_{agent.ensemble_q_num - 1}')
        alg = algorithmic_bytes(P_polyak, P_rq[1] - P_rq[0])
        B = CFG['batch_size']
        # the fused first launch: sampler (K1 + K2) + Polyak (K5) + the step's draws
        alg['asac_step_prologue_sample'] = alg['asac_sumtree_sample'] + alg['asac_polyak'] + 8 * B + 4 * agent._eps_all.numel()
        # batches of 257 .. 1 024: K2 (IS weights, 8 B bytes) is formed by an extra workgroup of the gather's launch
        alg['asac_step_prologue_sample_partial'] = alg['asac_step_prologue_sample'] - 8 * B
        alg['asac_window_gather_pad_w'] = alg['asac_window_gather_pad'] + 8 * B
        for name, st in sorted(summ.items(), key=lambda kv: -kv[1]['avg_us'] * kv[1]['calls']):
            by = alg.get(name)
            calls_per_step = st['calls'] / args.profile_steps
            kernels[name] = {'avg_us': round(st['avg_us'], 3), 'min_us': round(st['min_us'], 3),
                             'med_us': round(st['med_us'], 3),
                             'launches_per_step': round(calls_per_step, 2),
                             'alg_bytes_per_launch': by,
                             'achieved_GBs': None if by is None else round(by / (st['avg_us'] * 1e-6) / 1e9, 3)}
            if 'tflops' in st:
                kernels[name]['alg_flops_per_launch'] = round(st['flops_per_launch'])
                kernels[name]['achieved_TFLOPs'] = round(st['tflops'], 3)
            # the same launches grouped by the KERNEL that runs them (the backward has three entry points)
            g = by_kernel.setdefault(_KERNEL_OF.get(name, name), {'entry_points': [], 'launches_per_step': 0.0, 'us_per_step': 0.0,
                                                                  'flops_per_step': 0.0, 'bytes_per_step': 0.0, 'sized': True})
            g['entry_points'].append(name)
            g['launches_per_step'] += calls_per_step
            g['us_per_step'] += st['avg_us'] * calls_per_step
            if 'tflops' in st:
                g['flops_per_step'] += st['flops_per_launch'] * calls_per_step
            elif by is not None:
                g['bytes_per_step'] += by * calls_per_step
            else:
                g['sized'] = False
        small = (f'batch {CFG["batch_size"]} gives a launch of the Q / policy networks tens of MFLOP at most: latency-bound by '
                 'construction (SURVEY.md §8d); `sweep` holds the saturating-size runs of the HBM-bound kernels')
        # dominant = the kernel with the largest device time per step; mean launch time throughout
        dom, d = max(by_kernel.items(), key=lambda kv: kv[1]['us_per_step'])
        mean_us = d['us_per_step'] / d['launches_per_step']
        common = {'kernel': dom, 'entry_points': d['entry_points'], 'avg_launch_us': round(mean_us, 3),
                  'launches_per_step': round(d['launches_per_step'], 2), 'us_per_step': round(d['us_per_step'], 2),
                  'timing': 'mean over launches, HIP events on the launch stream', 'note': small}
        if d['flops_per_step'] > 0:
            ach = d['flops_per_step'] / (d['us_per_step'] * 1e-6) / 1e12
            roofline = {'bound': 'mfma', 'achieved': round(ach, 4), 'peak': MFMA_F32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                        'frac': round(ach / MFMA_F32_PEAK_TFLOPS, 6), 'traffic': None,
                        'alg_flops_per_launch': round(d['flops_per_step'] / d['launches_per_step']), **common}
        elif d['sized']:
            ach = d['bytes_per_step'] / (d['us_per_step'] * 1e-6) / 1e9
            roofline = {'bound': 'hbm', 'achieved': round(ach, 3), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                        'frac': round(ach / HBM_PEAK_GBS, 6), 'traffic': None,
                        'alg_bytes_per_launch': round(d['bytes_per_step'] / d['launches_per_step']), **common}
        if roofline is not None:
            roofline = roofline_with_profile(roofline, args.config, dom)

        # the north-star's "sample + return kernels" (K1-K4) as ONE group at this batch size
        members = [n_ for n_ in SAMPLE_RETURN if n_ in kernels]
        if members:
            by_step = sum(kernels[m]['alg_bytes_per_launch'] * kernels[m]['launches_per_step'] for m in members)
            us_step = sum(kernels[m]['avg_us'] * kernels[m]['launches_per_step'] for m in members)
            traffic, srcs = 0, set()
            for m in members:
                tr, src = pmc_traffic(args.config, _KERNEL_OF[m])
                if tr is None:
                    traffic = None
                    break
                traffic += tr * kernels[m]['launches_per_step']
                srcs.add(src)
            ach = by_step / (us_step * 1e-6) / 1e9
            situ = [in_situ(args.config, _KERNEL_OF[m]) for m in members]
            us_situ = None if any(x[0] is None for x in situ) else sum(x[0] * kernels[m]['launches_per_step'] for x, m in zip(situ, members))
            ach_situ = None if us_situ is None else by_step / (us_situ * 1e-6) / 1e9
            roofline_hbm = {'kernels': [_KERNEL_OF[m] for m in members], 'bound': 'hbm',
                            'achieved': round(ach if ach_situ is None else ach_situ, 3),
                            'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                            'frac': round((ach if ach_situ is None else ach_situ) / HBM_PEAK_GBS, 6),
                            'alg_bytes_per_step': int(by_step),
                            'us_per_step': round(us_step if us_situ is None else us_situ, 2),
                            'traffic': None if traffic is None else int(traffic),
                            'traffic_over_algorithmic': None if traffic is None else round(traffic / by_step, 2),
                            'traffic_source': sorted(srcs) if traffic is not None else None,
                            'achieved_hip_events': round(ach, 3), 'frac_hip_events': round(ach / HBM_PEAK_GBS, 6),
                            'us_per_step_hip_events': round(us_step, 2),
                            'frac_source': situ[0][2] if us_situ is not None else f'live ({next(x[2] for x in situ if x[0] is None)})',
                            'note': 'K1+K2 sample / IS weights (with the step prologue), K3 window gather, K4 return + min; '
                                    'in situ = inside the replayed hipGraph step'}

    sweep = configs = None
    if rank == 0 and world == 1 and not args.no_extras:
        agent._graph = None
        torch.cuda.empty_cache()
        from tools import kernel_sweep
        sweep = kernel_sweep.saturating_sweep()
        configs = other_configs()

    cpu = None
    if rank == 0 and not args.no_cpu_baseline and world == 1:
        cpu = cpu_baseline(args.cpu_budget, args.fill)

    if rank == 0:
        value = (world if args.scaling == 'weak' else 1) * args.steps / dt
        out = {
            'metric': f'SAC train steps/sec (PER sample + grad step), batch {CFG["batch_size"] * world}',
            'value': round(value, 2), 'unit': 'train_steps/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(1e3 * dt / args.steps, 4),
            f'value_runs_of_{spl}': None if dt_runs is None else round((world if args.scaling == 'weak' else 1) * args.steps / dt_runs, 2),
            'value_lookahead': (configs or {}).get('cfg2_lookahead', {}).get('value') if args.config == 'cfg2' else None,
            'higher_is_better': True, 'scaling': args.scaling, 'vs_baseline': None, 'dtype': 'f32',
            'data': 'synthetic',
            'config': {'workload': CFG.get('short') or CFG['desc'], 'workload_long': CFG['desc'],
                       'per_gpu_batch': CFG['batch_size'], 'rows_resident': int(args.fill), 'replay_capacity': CFG['capacity'],
                       'n_step': CFG['n_step'], 'burn_in_step': CFG['burn_in_step'],
                       'global_batch': CFG['batch_size'] * world,
                       'replay_shard_capacity': CFG['capacity'] // world,
                       'parallelism': f'dp{world}' if world > 1 else 'single',
                       'ranks': torch.distributed.get_world_size() if dist_ctx is not None else 1, 'collectives': (('gloo, every rank on ONE device (test mode, not a measurement)' if one_device else 'RCCL (nccl backend)')
                                       if dist_ctx is not None else None),
                       'hipgraph': bool(graph_used),
                       'hipgraph_memset_nodes_replaced': None if graph_memsets is None else graph_memsets[0],
                       'steps_per_graph_launch': 1},
            'roofline': roofline, 'roofline_hbm': roofline_hbm, 'sweep': sweep, 'configs': configs,
            'kernels': kernels, 'cpu_baseline': cpu,
        }
        emit(out, args.emit)
    agent.close()
    if dist_ctx is not None:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
