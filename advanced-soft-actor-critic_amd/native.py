"""ctypes binding of `libasac_hip.so` (C ABI: `include/asac_hip.h`).

There is NO fallback: if the library is missing or a launch fails this module raises.  torch must
be imported before the library is loaded so that both bind to the same `libamdhip64.so.7` (the HIP
runtime bundled with PyTorch-ROCm), which is what lets these kernels run on torch's streams, on
torch-allocated HBM, and inside torch-captured hipGraphs.
"""
import ctypes as C
from pathlib import Path

import torch  # noqa: F401  (must precede the dlopen below, see module docstring)

PKG_DIR = Path(__file__).resolve().parent
import os as _os

LIB_PATH = Path(_os.environ.get('ASAC_HIP_LIB', PKG_DIR / 'lib' / 'libasac_hip.so'))   # env override: debugging builds
ABI_VERSION = 81

MAX_GATHER_KEYS = 16
PAD_KEEP, PAD_WORD, PAD_BYTE, PAD_ROW, PAD_EMIT_MASK = 0, 1, 2, 3, 4
CVT_NONE, CVT_U8_TO_F32_UNIT, CVT_BOOL_TO_F32 = 0, 1, 2


class AsacNativeError(RuntimeError):
    pass


class GatherKey(C.Structure):
    _fields_ = [('src', C.c_void_p), ('dst', C.c_void_p), ('pad_row', C.c_void_p),
                ('row_bytes', C.c_int32), ('pad_mode', C.c_int32), ('pad_word', C.c_uint32),
                ('convert', C.c_int32), ('dst_row_pitch', C.c_int32), ('derive', C.c_int32)]


class AdamEpilogue(C.Structure):
    _fields_ = [('param_base', C.c_void_p), ('grad_base', C.c_void_p), ('exp_avg_base', C.c_void_p),
                ('exp_avg_sq_base', C.c_void_p), ('lr', C.c_float), ('beta1', C.c_float), ('beta2', C.c_float),
                ('eps', C.c_float), ('steps_done', C.c_void_p)]


def adam_epilogue(param_flat, grad_flat, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, steps_done) -> AdamEpilogue:
    """The optimizer step a gradient-finishing launch takes itself (`asac_adam_epilogue_t`): flat buffers of one
    layout; the gradients the launch writes must be views of `grad_flat`."""
    for t in (param_flat, grad_flat, exp_avg, exp_avg_sq):
        assert t.is_cuda and t.is_contiguous() and t.dtype == torch.float32 and t.numel() == param_flat.numel()
    assert steps_done.dtype == torch.int64 and steps_done.is_cuda
    ep = AdamEpilogue(param_flat.data_ptr(), grad_flat.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(),
                      float(lr), float(beta1), float(beta2), float(eps), steps_done.data_ptr())
    ep._keep = (param_flat, grad_flat, exp_avg, exp_avg_sq, steps_done)
    ep._span = (grad_flat.data_ptr(), grad_flat.data_ptr() + 4 * grad_flat.numel())
    return ep


ROW_ITEM, ROW_SLOT, ROW_SLOT_ROW, ROW_BROADCAST = 0, 1, 2, 3
DERIVE_NONE, DERIVE_PREVIOUS, DERIVE_HOLD_LAST, DERIVE_HOLD_LAST_NEXT = 0, 1, 2, 3


class RowMove(C.Structure):
    _fields_ = [('src', C.c_void_p), ('dst', C.c_void_p),
                ('src_stride0', C.c_int64), ('src_stride1', C.c_int64),
                ('dst_stride0', C.c_int64), ('dst_stride1', C.c_int64),
                ('row_bytes', C.c_int32), ('src_mode', C.c_int32), ('dst_mode', C.c_int32),
                ('src_row_offset', C.c_int32), ('pad_word', C.c_uint32), ('reserved_', C.c_int32)]


class VtraceArgs(C.Structure):
    _fields_ = [
        ('q', C.c_void_p), ('q_stride_e', C.c_int64), ('q_stride_b', C.c_int64), ('q_stride_t', C.c_int64),
        ('subset_n', C.c_void_p), ('subset_next', C.c_void_p), ('E_sample', C.c_int32),
        ('logp', C.c_void_p), ('log_alpha', C.c_void_p),
        ('reward', C.c_void_p), ('reward_stride', C.c_int64),
        ('done', C.c_void_p), ('last_mask', C.c_void_p), ('padding_mask', C.c_void_p), ('mask_stride', C.c_int64),
        ('mu_prob', C.c_void_p), ('mu_stride_b', C.c_int64), ('mu_stride_t', C.c_int64), ('mu_offset', C.c_int32),
        ('pi_prob', C.c_void_p), ('pi_stride_b', C.c_int64), ('pi_stride_t', C.c_int64), ('A', C.c_int32),
        ('gamma_ratio', C.c_void_p), ('lambda_ratio', C.c_void_p),
        ('gamma', C.c_float), ('v_rho', C.c_float), ('v_c', C.c_float),
        ('use_n_step_is', C.c_int32), ('B', C.c_int32), ('n', C.c_int32),
        ('q_online', C.c_void_p), ('E_online', C.c_int32),
        ('td_error_out', C.c_void_p), ('y_out', C.c_void_p)]


class MlpDesc(C.Structure):
    _fields_ = [('in0', C.c_int32), ('in1', C.c_int32), ('n_blocks', C.c_int32),
                ('width', C.c_int32 * 4), ('residual', C.c_int32 * 4), ('head_cols', C.c_int32 * 2),
                ('w_off', C.c_int64 * 4), ('b_off', C.c_int64 * 4),
                ('head_w_off', C.c_int64 * 2), ('head_b_off', C.c_int64 * 2),
                ('head_transform', C.c_int32), ('reserved_', C.c_int32)]


class MlpJob(C.Structure):
    _fields_ = [('desc', C.POINTER(MlpDesc)), ('params', C.c_void_p), ('member_stride', C.c_int64),
                ('x0', C.c_void_p), ('x0_row_stride', C.c_int64), ('x0_member_stride', C.c_int64),
                ('x1', C.c_void_p), ('x1_row_stride', C.c_int64), ('x1_member_stride', C.c_int64),
                ('N', C.c_int64), ('out', C.c_void_p), ('E', C.c_int32), ('x0_window_T', C.c_int32),
                ('x0_sample_stride', C.c_int64)]


MLP_MAX_JOBS = 2
# the policy forward's sampling epilogue (`mlp_forward_multi_sampled`) instead of a `squash_multi` launch behind it;
# ASAC_SAMPLE_EPILOGUE=0: the chain (A/B runs)
SAMPLE_EPILOGUE = _os.environ.get('ASAC_SAMPLE_EPILOGUE', '1') != '0'


class Sidecar(C.Structure):
    """asac_sidecar_t: a small off-critical-path job that rides as extra workgroups of a hosting launch"""
    _fields_ = [('kind', C.c_int32),
                ('logp', C.c_void_p), ('B', C.c_int32), ('target', C.c_float), ('slot', C.c_int32),
                ('param', C.c_void_p), ('grad', C.c_void_p), ('exp_avg', C.c_void_p), ('exp_avg_sq', C.c_void_p),
                ('n', C.c_int32), ('lr', C.c_float), ('beta1', C.c_float), ('beta2', C.c_float), ('eps', C.c_float),
                ('steps_done', C.c_void_p), ('advance_counter', C.c_int32),
                ('ring', C.c_void_p), ('row_bytes', C.c_int32), ('capacity', C.c_int32), ('ids', C.c_void_p),
                ('batch', C.c_int32), ('first_off', C.c_int32), ('count', C.c_int32), ('slot_ids', C.c_void_p),
                ('padding_mask', C.c_void_p), ('mask_sample_stride', C.c_int32), ('rows', C.c_void_p),
                ('rows_sample_stride_bytes', C.c_int64), ('rows_row_stride_bytes', C.c_int64), ('winner', C.c_void_p),
                ('gather_plan', C.c_void_p), ('gather_blocks', C.c_int32)]


SIDECAR_ALPHA_ADAM, SIDECAR_SCATTER_ELECT, SIDECAR_SCATTER_WRITE, SIDECAR_WINDOW_GATHER, MAX_SIDECARS = 1, 2, 3, 4, 4


class SquashJob(C.Structure):
    _fields_ = [('loc', C.c_void_p), ('scale', C.c_void_p), ('ls_row_stride', C.c_int64), ('eps', C.c_void_p),
                ('rows', C.c_int64), ('A', C.c_int32), ('T', C.c_int32), ('a_tanh_out', C.c_void_p),
                ('logp_out', C.c_void_p), ('x_out', C.c_void_p), ('action', C.c_void_p),
                ('action_stride_b', C.c_int64), ('action_stride_t', C.c_int64), ('prob_out', C.c_void_p),
                ('prob_stride_b', C.c_int64), ('prob_stride_t', C.c_int64), ('action_offset', C.c_int32),
                ('prob_offset', C.c_int32)]


SQUASH_MAX_JOBS = 4


class PiQJob(C.Structure):
    """asac_pi_q_job_t: policy forward -> sampling -> critics forward over the same rows"""
    _fields_ = [('pi', MlpJob), ('sample', SquashJob), ('eps2', C.c_void_p), ('t2', C.c_int32), ('reserved_', C.c_int32),
                ('a2_out', C.c_void_p), ('logp2_out', C.c_void_p), ('q', MlpJob)]


class SampleEpilogue(C.Structure):
    """asac_mlp_sample_epilogue_t: asac_squash_multi's jobs as the epilogue of a policy job of `mlp_forward_multi_sampled`"""
    _fields_ = [('sample', SquashJob), ('eps2', C.c_void_p), ('t2', C.c_int32), ('reserved_', C.c_int32),
                ('a2_out', C.c_void_p), ('logp2_out', C.c_void_p)]


class PartialSum(C.Structure):
    """asac_partial_sum_t: one fixed-order sum of per-workgroup partials (`sum_partials_multi`)"""
    _fields_ = [('partial', C.c_void_p), ('out', C.c_void_p), ('slab_stride', C.c_int64), ('n', C.c_int64),
                ('slabs', C.c_int32), ('slices', C.c_int32), ('accumulate', C.c_int32), ('pad_', C.c_int32)]


class GruDesc(C.Structure):
    _fields_ = [('input', C.c_int32), ('hidden', C.c_int32), ('hidden_pow2', C.c_int32), ('layers', C.c_int32)]


class Conv2Desc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ('channels', 'height', 'width', 'out1', 'kernel1', 'stride1',
                                         'out2', 'kernel2', 'stride2')]


GRU_MAX_LAYERS, GRU_MAX_DIM = 2, 16
_PtrArray = C.c_void_p * GRU_MAX_LAYERS

class ObsDecoderParams(C.Structure):
    """`asac_obs_decoder_params_t`: the ten parameter tensors of the observation decoder (or their gradients)"""
    _fields_ = [(n, C.c_void_p) for n in ('dense1_w', 'dense1_b', 'dense2_w', 'dense2_b', 'ct1_w', 'ct1_b', 'ct2_w',
                                          'ct2_b', 'ct3_w', 'ct3_b')]


_SIGNATURES = {
    'asac_version': (C.c_int, []),
    'asac_last_error': (C.c_char_p, []),
    'asac_set_launch_repeat': (C.c_int, [C.c_int]),
    'asac_sumtree_sample': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p]),
    'asac_per_is_weights': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double,
                                      C.c_void_p, C.c_void_p]),
    'asac_sumtree_update': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_float, C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_void_p,
                                      C.c_void_p]),
    'asac_per_add': (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_float,
                               C.c_void_p, C.c_void_p]),
    'asac_sumtree_leaf_max': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    'asac_sumtree_check': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    'asac_gelu_eval': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    'asac_sumtree_plan_top': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'asac_sumtree_descend_owned': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'asac_per_is_weights_slice': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_double,
                                            C.c_void_p, C.c_void_p]),
    'asac_window_gather_pad': (C.c_int, [C.POINTER(GatherKey), C.c_int, C.c_void_p, C.c_int, C.c_int,
                                         C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    'asac_window_gather_plan_bytes': (C.c_int64, []),
    'asac_window_gather_plan': (C.c_int, [C.POINTER(GatherKey), C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]),
    'asac_sumtree_descend': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p]),
    'asac_gather_rows': (C.c_int, [C.POINTER(GatherKey), C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    'asac_rows_move': (C.c_int, [C.POINTER(RowMove), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                 C.c_void_p]),
    'asac_window_aux': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64,
                                  C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    'asac_scatter_rows_if_id_match': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                                C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                                C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
    'asac_squash_sample_fwd': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int,
                                         C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_int,
                                         C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_void_p]),
    'asac_squash_sample_bwd': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int,
                                         C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p,
                                         C.c_void_p, C.c_int64, C.c_void_p]),
    'asac_squash_multi': (C.c_int, [C.POINTER(SquashJob), C.c_int, C.POINTER(Sidecar), C.c_int, C.c_void_p]),
    'asac_squash_prob': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_int64, C.c_int64,
                                   C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_int,
                                   C.c_void_p]),
    'asac_vtrace_return_min': (C.c_int, [C.POINTER(VtraceArgs), C.c_void_p]),
    'asac_vtrace_return_min_sc': (C.c_int, [C.POINTER(VtraceArgs), C.POINTER(Sidecar), C.c_int, C.POINTER(Sidecar),
                                            C.c_void_p]),
    'asac_td_update': (C.c_int, [C.POINTER(VtraceArgs), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_float,
                                 C.c_float, C.c_void_p, C.c_void_p, C.POINTER(Sidecar), C.c_int, C.POINTER(Sidecar),
                                 C.c_void_p]),
    'asac_sumtree_update_sc': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float,
                                         C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(Sidecar), C.c_int,
                                         C.c_void_p]),
    'asac_vtrace_return_direct': (C.c_int, [C.POINTER(VtraceArgs), C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_void_p]),
    'asac_q_loss_fwd_bwd': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                      C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    'asac_mlp_forward': (C.c_int, [C.POINTER(MlpDesc), C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int64,
                                   C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
    'asac_mlp_forward_multi': (C.c_int, [C.POINTER(MlpJob), C.c_int, C.POINTER(Sidecar), C.c_int, C.c_void_p]),
    'asac_mlp_forward_multi_sampled_ok': (C.c_int, [C.POINTER(MlpJob), C.c_int, C.POINTER(SampleEpilogue)]),
    'asac_mlp_forward_multi_sampled': (C.c_int, [C.POINTER(MlpJob), C.c_int, C.POINTER(SampleEpilogue), C.POINTER(Sidecar),
                                                 C.c_int, C.c_void_p]),
    'asac_mlp_backward_workspace': (C.c_int64, [C.c_int64, C.c_int, C.c_int64]),
    'asac_mlp_backward_tiles': (C.c_int64, [C.c_int64, C.c_int]),
    'asac_mlp_backward': (C.c_int, [C.POINTER(MlpDesc), C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int64,
                                    C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    'asac_mlp_backward_qloss': (C.c_int, [C.POINTER(MlpDesc), C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int64,
                                          C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_int, C.c_void_p]),
    'asac_mlp_backward_qloss_gx': (C.c_int, [C.POINTER(MlpDesc), C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int64,
                                             C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_int, C.c_void_p]),
    'asac_mlp_backward_qloss_return_ok': (C.c_int, [C.POINTER(MlpDesc), C.c_void_p, C.c_int64, C.c_int, C.c_int64,
                                                    C.POINTER(VtraceArgs)]),
    'asac_mlp_backward_qloss_return': (C.c_int, [C.POINTER(MlpDesc), C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int64,
                                                 C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p,
                                                 C.POINTER(VtraceArgs), C.c_void_p, C.c_float, C.c_void_p, C.c_void_p,
                                                 C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    'asac_mlp_backward_policy_q': (C.c_int, [C.POINTER(MlpDesc), C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int64,
                                             C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p,
                                             C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    'asac_mlp_backward_policy_sample': (C.c_int, [C.POINTER(MlpDesc), C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                                  C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                                  C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    'asac_struct_size': (C.c_int64, [C.c_char_p]),
    'asac_policy_sample_q_forward_ok': (C.c_int, [C.POINTER(PiQJob)]),
    'asac_policy_sample_q_forward': (C.c_int, [C.POINTER(PiQJob), C.POINTER(MlpJob), C.c_int, C.POINTER(Sidecar), C.c_int,
                                               C.c_void_p]),
    'asac_policy_step_fused_ok': (C.c_int, [C.POINTER(MlpDesc), C.c_void_p, C.c_int64, C.POINTER(MlpDesc), C.c_void_p,
                                            C.c_int64, C.c_int64]),
    'asac_policy_step_fused': (C.c_int, [C.POINTER(MlpDesc), C.c_void_p, C.c_int64, C.POINTER(MlpDesc), C.c_void_p,
                                         C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    'asac_mlp_param_extent': (C.c_int64, [C.POINTER(MlpDesc)]),
    'asac_adam_step_partials': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float,
                                          C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_int64, C.c_int,
                                          C.c_int64, C.c_int64, C.c_int, C.c_void_p, C.c_int64, C.c_void_p]),
    'asac_gauss_head_fwd': (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    'asac_gauss_head_bwd': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p,
                                      C.c_void_p]),
    'asac_attention_supported': (C.c_int, [C.c_int, C.c_int, C.c_int]),
    'asac_attention_forward': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64,
                                         C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p]),
    'asac_attention_backward': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                          C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'asac_attention_proj_workspace': (C.c_int64, [C.c_int, C.c_int, C.c_int, C.c_int]),
    'asac_attention_proj_forward': (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int64,
                                              C.c_void_p * 8, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int,
                                              C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]),
    'asac_attention_proj_backward': (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int64,
                                               C.c_void_p * 8, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                               C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64,
                                               C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.c_int, C.c_void_p, C.c_void_p]),
    'asac_linear_tanh_workspace': (C.c_int64, [C.c_int64, C.c_int, C.c_int]),
    'asac_linear_tanh_forward': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int,
                                           C.c_void_p, C.c_void_p]),
    'asac_linear_tanh_backward': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int,
                                            C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    'asac_linear_tanh_forward2': (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_void_p,
                                            C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]),
    'asac_linear_tanh_forward2w': (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_int64,
                                             C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]),
    'asac_linear_tanh_backward2w': (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_int64,
                                              C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                              C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                              C.c_void_p]),
    'asac_linear_tanh_backward2': (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int,
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    'asac_curiosity_bonus': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int,
                                       C.c_int, C.c_int, C.c_float, C.c_void_p]),
    'asac_masked_mse': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_int,
                                  C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'asac_mse_mean_grad_workspace': (C.c_int64, []),
    'asac_mse_mean_grad': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_float,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'asac_masked_mse_workspace': (C.c_int64, [C.c_int64]),
    'asac_normal_nll_kl_workspace': (C.c_int64, [C.c_int64]),
    'asac_normal_nll_kl': (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p,
                                     C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p]),
    'asac_normal_nll_kl_logstd': (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_float, C.c_float, C.c_void_p, C.c_int64,
                                            C.c_int64, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_void_p]),
    'asac_sum_partials_multi': (C.c_int, [C.c_int, C.POINTER(PartialSum), C.c_void_p]),
    'asac_rows_wide_supported': (C.c_int, [C.c_int64, C.c_int, C.c_int]),
    'asac_rows_wide_workspace': (C.c_int64, [C.c_int64, C.c_int, C.c_int]),
    'asac_rows_wide_forward': (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'asac_rows_wide_backward_input': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_int,
                                                C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    'asac_rows_wide_backward_params': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_void_p,
                                                 C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    'asac_xty_supported': (C.c_int, [C.c_int64, C.c_int, C.c_int]),
    'asac_xty_workspace': (C.c_int64, [C.c_int64, C.c_int, C.c_int]),
    'asac_xty': (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_int64, C.c_void_p, C.c_void_p,
                           C.c_int, C.c_void_p, C.c_void_p]),
    'asac_xty_multi_workspace': (C.c_int64, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    'asac_xty_multi': (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    'asac_conv2_tiles': (C.c_int, [C.POINTER(Conv2Desc)]),
    'asac_conv2_z1_floats': (C.c_int64, [C.POINTER(Conv2Desc), C.c_int64]),
    'asac_conv2_supported': (C.c_int, [C.POINTER(Conv2Desc)]),
    'asac_conv2_group_frames': (C.c_int, [C.POINTER(Conv2Desc)]),
    'asac_conv2_backward_windows': (C.c_int, [C.POINTER(Conv2Desc), C.c_void_p, C.c_int64, C.c_int, C.c_int64, C.c_void_p,
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                              C.c_void_p]),
    'asac_conv2_backward_multi_max': (C.c_int, [C.POINTER(Conv2Desc)]),
    'asac_conv2_backward_slabs': (C.c_int, [C.POINTER(Conv2Desc), C.c_int64, C.c_int]),
    'asac_conv2_backward_multi': (C.c_int, [C.POINTER(Conv2Desc), C.c_void_p, C.c_int64, C.c_int, C.c_int64, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_int,
                                            C.c_void_p, C.c_void_p]),
    'asac_conv2_forward_windows': (C.c_int, [C.POINTER(Conv2Desc), C.c_void_p, C.c_int64, C.c_int, C.c_int64, C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_void_p]),
    'asac_conv2_param_count': (C.c_int64, [C.POINTER(Conv2Desc)]),
    'asac_conv2_backward_workspace': (C.c_int64, [C.POINTER(Conv2Desc), C.c_int64]),
    'asac_conv2_forward': (C.c_int, [C.POINTER(Conv2Desc), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'asac_conv2_backward': (C.c_int, [C.POINTER(Conv2Desc), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    'asac_gru_param_count': (C.c_int64, [C.POINTER(GruDesc)]),
    'asac_gru_backward_workspace': (C.c_int64, [C.POINTER(GruDesc), C.c_int]),
    'asac_gru_forward': (C.c_int, [C.POINTER(GruDesc), _PtrArray, _PtrArray, _PtrArray, _PtrArray, C.c_void_p,
                                   C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_int,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'asac_gru_forward_twin': (C.c_int, [C.POINTER(GruDesc), _PtrArray, _PtrArray, _PtrArray, _PtrArray, _PtrArray,
                                        _PtrArray, _PtrArray, _PtrArray, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p,
                                        C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'asac_gru_backward': (C.c_int, [C.POINTER(GruDesc), _PtrArray, _PtrArray, _PtrArray, _PtrArray, C.c_void_p,
                                    C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_int,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    'asac_gru_backward_at': (C.c_int, [C.POINTER(GruDesc), _PtrArray, _PtrArray, _PtrArray, _PtrArray, C.c_void_p,
                                       C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_int,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    'asac_policy_loss_fwd_bwd': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                           C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_void_p]),
    'asac_alpha_grad': (C.c_int, [C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p]),
    'asac_noise_fill': (C.c_int, [C.c_uint64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p,
                                  C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'asac_step_prologue': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_void_p, C.c_int64, C.c_uint64,
                                     C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int,
                                     C.c_int, C.c_int, C.c_void_p]),
    'asac_step_prologue_sample': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_void_p, C.c_int64, C.c_uint64,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_int,
                                            C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_double,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'asac_step_prologue_sample_partial': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_void_p, C.c_int64, C.c_uint64,
                                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_int,
                                                    C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                                    C.c_void_p, C.c_void_p, C.c_void_p]),
    'asac_window_gather_pad_w': (C.c_int, [C.POINTER(GatherKey), C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p,
                                           C.c_void_p]),
    'asac_graph_launch': (C.c_int, [C.c_void_p, C.c_void_p]),
    'asac_alpha_adam_step': (C.c_int, [C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float,
                                       C.c_void_p, C.c_int, C.c_void_p]),
    'asac_polyak': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_void_p]),
    'asac_adam_step': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float,
                                 C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p]),
    'asac_graph_replace_memset_nodes': (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    'asac_attention_mh_supported': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    'asac_attention_mh_forward': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64,
                                            C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'asac_attention_mh_proj_supported': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    'asac_attention_mh_proj_forward': (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                                 C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                 C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'asac_attention_mh_block_backward': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64,
                                                   C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int64,
                                                   C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'asac_attention_mh_backward': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64,
                                             C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'asac_rows_proj_supported': (C.c_int, [C.c_int]),
    'asac_rows_proj_forward': (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'asac_rows_proj_backward': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                          C.c_void_p]),
    'asac_rows_resblock_forward': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int,
                                             C.c_void_p, C.c_void_p, C.c_void_p]),
    'asac_rows_resblock_backward': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p,
                                              C.c_void_p, C.c_void_p]),
    'asac_rows_affine_supported': (C.c_int, [C.c_int, C.c_int]),
    'asac_rows_affine_forward': (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p,
                                           C.c_void_p]),
    'asac_rows_affine_gelu_forward': (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int,
                                                C.c_void_p, C.c_void_p, C.c_void_p]),
    'asac_gru_wide_supported': (C.c_int, [C.c_int]),
    'asac_gru_wide_forward_twin': (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                             C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    'asac_gru_wide_forward': (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                        C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_int64,
                                        C.c_void_p, C.c_void_p, C.c_void_p]),
    'asac_gru_wide_backward': (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p]),
    'asac_cosine_gate_add': (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_int64, C.c_void_p, C.c_void_p,
                                       C.c_void_p]),
    'asac_obs_decoder_packed_floats': (C.c_int64, []),
    'asac_obs_decoder_saved_floats': (C.c_int64, [C.c_int64]),
    'asac_obs_decoder_workspace_floats': (C.c_int64, [C.c_int64]),
    'asac_obs_decoder_forward': (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.POINTER(ObsDecoderParams),
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'asac_obs_decoder_backward': (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.POINTER(ObsDecoderParams), C.c_int, C.c_void_p,
                                            C.c_void_p]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None


def load() -> C.CDLL:
    """dlopen the library (once) and type every entry point; raises if absent or ABI-mismatched."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise AsacNativeError(
            f'{LIB_PATH} is missing: build it with `python __graft_entry__.py` (hipcc, gfx950). '
            'There is no CPU or eager fallback for the SAC hot path.')
    lib = C.CDLL(str(LIB_PATH))
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if the symbol is not exported
        fn.restype, fn.argtypes = res, args
    if lib.asac_version() != ABI_VERSION:
        raise AsacNativeError(f'libasac_hip.so ABI {lib.asac_version()} != binding {ABI_VERSION}; rebuild')
    _lib = lib
    return lib


def _check(rc: int, what: str):
    if rc != 0:
        raise AsacNativeError(f'{what} failed (hipError {rc}): {load().asac_last_error().decode()}')


def _p(t):
    """device pointer of a tensor (or None)"""
    if t is None:
        return None
    assert t.is_cuda, 'libasac_hip kernels take device memory only'
    return C.c_void_p(t.data_ptr())


def _stream():
    # (the raw handle of torch's current stream on the current device; `torch.cuda.current_stream().cuda_stream` builds
    # a Stream object per call: 13 us of host time in front of every eager launch)
    return C.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))


class LaunchProfiler:
    """Per-entry-point device time from HIP events recorded on the launch stream (eager mode only:
    nothing is recorded while a graph is being captured).  A single launch here lasts 3-10 us, which
    is below both what one event pair resolves and the host cost of issuing it through ctypes, so
    while the profiler is active the library re-issues every launch `repeat` times back-to-back
    (`asac_set_launch_repeat`) and the elapsed time is divided by `repeat`.  Used by bench.py's
    profile pass AFTER the timed region, where repeating an optimizer / Polyak launch is harmless.
    `summary()` synchronises."""

    def __init__(self, repeat: int = 20):
        self.repeat = max(1, int(repeat))
        self.records: dict[str, list] = {}

    def __enter__(self):
        load().asac_set_launch_repeat(self.repeat)
        set_profiler(self)
        return self

    def __exit__(self, *exc):
        set_profiler(None)
        load().asac_set_launch_repeat(1)
        return False

    def bracket(self, name, fn, *a, **k):
        global _last_work
        if torch.cuda.is_current_stream_capturing():
            return fn(*a, **k)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        _last_work = None
        e0.record()
        out = fn(*a, **k)          # the library issues each launch `repeat` times (asac_set_launch_repeat)
        e1.record()
        self.records.setdefault(name, []).append((e0, e1, _last_work))
        return out

    def summary(self) -> dict:
        torch.cuda.synchronize()
        out = {}
        for name, evs in self.records.items():
            us = [1e3 * a.elapsed_time(b) / self.repeat for a, b, _ in evs]
            # a bracket also spans the host's time to issue the first launch when the device had run dry
            # (allocator hiccups): the median over calls is the robust per-launch figure, the mean is kept
            med = sorted(us)[len(us) // 2]
            kept = [t for t in us if t <= 3.0 * med]       # (host-stall outliers: one 4 ms bracket in 50 doubles a mean)
            out[name] = {'calls': len(us), 'avg_us': sum(kept) / len(kept), 'avg_us_raw': sum(us) / len(us),
                         'min_us': min(us), 'med_us': med}
            pairs = [(t, ev[2]) for t, ev in zip(us, evs) if ev[2] is not None and t <= 3.0 * med]
            if pairs:   # wrappers that know their per-launch work (flops) report it: achieved = sum / sum over
                        # the calls that are not host-stall outliers (launch sizes differ, so no plain median)
                out[name]['flops_per_launch'] = sum(w for _, w in pairs) / len(pairs)
                out[name]['tflops'] = sum(w for _, w in pairs) / (sum(t for t, _ in pairs) * 1e-6) / 1e12
        return out


_profiler: LaunchProfiler | None = None
_last_work = None    # per-launch algorithmic work reported by a wrapper to the active profiler


def set_profiler(p: 'LaunchProfiler | None') -> None:
    global _profiler
    _profiler = p


def _profiled(fn):
    name = 'asac_' + fn.__name__

    def wrapper(*a, **k):
        if _profiler is None:
            return fn(*a, **k)
        return _profiler.bracket(name, fn, *a, **k)
    wrapper.__name__, wrapper.__doc__ = fn.__name__, fn.__doc__
    return wrapper


# ------------------------------------------------------------------------------------------------
# thin typed wrappers (tensor in, launch on torch's current stream)
# ------------------------------------------------------------------------------------------------
@_profiled
def sumtree_sample(tree, capacity, batch, u, slot_ids, beta_state, beta_increment, leaf_out, p_out,
                   ids_out, is_weights_out, min_p_out):
    assert tree.dtype == torch.float32 and u.dtype == torch.float64 and slot_ids.dtype == torch.int64
    assert leaf_out.dtype == torch.int32 and ids_out.dtype == torch.int64 and min_p_out.numel() >= 2
    _check(load().asac_sumtree_sample(_p(tree), capacity, batch, _p(u), _p(slot_ids), _p(beta_state),
                                      float(beta_increment), _p(leaf_out), _p(p_out), _p(ids_out),
                                      _p(is_weights_out), _p(min_p_out), _stream()), 'asac_sumtree_sample')


@_profiled
def sumtree_descend(tree, capacity, values, slot_ids, leaf_out, p_out, ids_out):
    """leaf / priority / id for explicit f64 `values` (the descent of `sumtree_sample` alone)"""
    assert values.dtype == torch.float64 and leaf_out.dtype == torch.int32 and ids_out.dtype == torch.int64
    _check(load().asac_sumtree_descend(_p(tree), capacity, values.numel(), _p(values), _p(slot_ids), _p(leaf_out),
                                       _p(p_out), _p(ids_out), _stream()), 'asac_sumtree_descend')


@_profiled
def per_is_weights(p, batch, total, min_ratio, beta_state, beta_increment, w_out):
    _check(load().asac_per_is_weights(_p(p), batch, _p(total), _p(min_ratio), _p(beta_state),
                                      float(beta_increment), _p(w_out), _stream()), 'asac_per_is_weights')


@_profiled
def sumtree_update(tree, capacity, ids, slot_ids, td_error, alpha, td_min, td_max, mode, winner, nan_flag, sidecars=None):
    k = ids.numel()
    assert ids.dtype == torch.int64 and td_error.dtype == torch.float32 and td_error.numel() == k
    assert winner.dtype == torch.int32 and winner.numel() >= capacity + (2 * k if k > 1024 else 0)
    sc, n_sc = _sidecar_array(sidecars)
    _check(load().asac_sumtree_update_sc(_p(tree), capacity, k, _p(ids), _p(slot_ids), _p(td_error),
                                         alpha, td_min, td_max, mode, _p(winner), _p(nan_flag), sc, n_sc, _stream()),
           'asac_sumtree_update')


@_profiled
def per_add(tree, capacity, first_id, count, ignore_size, max_p_dev, max_p_host, slot_ids):
    _check(load().asac_per_add(_p(tree), capacity, int(first_id), int(count), int(ignore_size),
                               _p(max_p_dev), float(max_p_host), _p(slot_ids), _stream()), 'asac_per_add')


@_profiled
def sumtree_leaf_max(tree, capacity, out):
    _check(load().asac_sumtree_leaf_max(_p(tree), capacity, _p(out), _stream()), 'asac_sumtree_leaf_max')


def sumtree_check(tree, capacity, out):
    _check(load().asac_sumtree_check(_p(tree), capacity, _p(out), _stream()), 'asac_sumtree_check')


def sumtree_plan_top(shard_roots, batch, u, owner_out, value_out, total_out):
    assert shard_roots.dtype == torch.float32 and u.dtype == torch.float64 and owner_out.dtype == torch.int32
    assert value_out.dtype == torch.float64 and u.numel() >= batch
    _check(load().asac_sumtree_plan_top(_p(shard_roots), shard_roots.numel(), batch, _p(u), _p(owner_out), _p(value_out),
                                        _p(total_out), _stream()), 'asac_sumtree_plan_top')


def sumtree_descend_owned(tree, capacity, values, owner, rank, slot_ids, leaf_out, p_out, ids_out):
    assert values.dtype == torch.float64 and owner.dtype == torch.int32 and ids_out.dtype == torch.int64
    _check(load().asac_sumtree_descend_owned(_p(tree), capacity, values.numel(), _p(values), _p(owner), int(rank),
                                             _p(slot_ids), _p(leaf_out), _p(p_out), _p(ids_out), _stream()),
           'asac_sumtree_descend_owned')


def per_is_weights_slice(p_all, first, count, total, beta_state, beta_increment, w_out):
    _check(load().asac_per_is_weights_slice(_p(p_all), p_all.numel(), first, count, _p(total), _p(beta_state),
                                            float(beta_increment), _p(w_out), _stream()), 'asac_per_is_weights_slice')


def gelu_eval(z, value, deriv):
    _check(load().asac_gelu_eval(_p(z), _p(value), _p(deriv), z.numel(), _stream()), 'asac_gelu_eval')


@_profiled
def window_gather_pad(keys, ids, batch, prev_n, post_n, capacity, index_ring):
    """keys: ctypes array of GatherKey (build once with `make_gather_keys`)."""
    _check(load().asac_window_gather_pad(keys, len(keys), _p(ids), batch, prev_n, post_n, capacity,
                                         _p(index_ring), _stream()), 'asac_window_gather_pad')


def sidecar_window_gather(keys, ids, batch, prev_n, post_n, capacity, index_ring) -> Sidecar:
    """`window_gather_pad` as a sidecar job (hosted by `policy_sample_q_forward`): the launch description goes to device
    memory once (a blocking copy: call at build time, not inside a capture); the job keeps it alive through `_keep`."""
    import torch
    plan = torch.empty(int(load().asac_window_gather_plan_bytes()), dtype=torch.uint8, device=ids.device)
    blocks = C.c_int(0)
    _check(load().asac_window_gather_plan(keys, len(keys), _p(ids), batch, prev_n, post_n, capacity, _p(index_ring), _p(plan),
                                          C.byref(blocks)), 'asac_window_gather_plan')
    sc = Sidecar(kind=SIDECAR_WINDOW_GATHER, gather_plan=_p(plan), gather_blocks=blocks.value)
    sc._keep = (plan, keys, ids, index_ring)
    return sc


@_profiled
def gather_rows(keys, ids, capacity):
    """dst_key[r] = ring_key[ids[r] % capacity] for every key of `keys` (all PAD_KEEP) in one launch"""
    _check(load().asac_gather_rows(keys, len(keys), _p(ids), ids.numel(), capacity, _stream()), 'asac_gather_rows')


def make_row_moves(specs):
    """specs: list of dicts(src, dst, row_bytes, src_mode, dst_mode, src_stride0/1, dst_stride0/1 (bytes),
    src_row_offset=0, pad_word=0); src / dst are tensors (their data_ptr is taken; keep them alive)."""
    assert 0 < len(specs) <= MAX_GATHER_KEYS
    arr = (RowMove * len(specs))()
    for k, s in zip(arr, specs):
        k.src, k.dst = s['src'].data_ptr(), s['dst'].data_ptr()
        k.row_bytes = int(s['row_bytes'])
        k.src_mode, k.dst_mode = int(s['src_mode']), int(s['dst_mode'])
        k.src_stride0, k.src_stride1 = int(s.get('src_stride0', 0)), int(s.get('src_stride1', 0))
        k.dst_stride0, k.dst_stride1 = int(s.get('dst_stride0', 0)), int(s.get('dst_stride1', 0))
        k.src_row_offset = int(s.get('src_row_offset', 0))
        k.pad_word = int(s.get('pad_word', 0)) & 0xffffffff
    return arr


@_profiled
def rows_move(keys, slot, src_row, dst_row, n_items):
    """one launch: every key of `keys` (ctypes array from `make_row_moves`) moves `n_items` rows"""
    _check(load().asac_rows_move(keys, len(keys), _p(slot), _p(src_row), _p(dst_row), int(n_items), _stream()),
           'asac_rows_move')


def make_gather_keys(specs):
    """specs: list of dicts(src, dst, row_bytes, pad_mode, pad_word=0, pad_row=None, convert=0)."""
    assert 0 < len(specs) <= MAX_GATHER_KEYS
    arr = (GatherKey * len(specs))()
    for k, s in zip(arr, specs):
        k.src = s['src'].data_ptr() if s.get('src') is not None else None
        k.dst = s['dst'].data_ptr()
        k.pad_row = s['pad_row'].data_ptr() if s.get('pad_row') is not None else None
        k.row_bytes = int(s.get('row_bytes', 1))
        k.pad_mode = int(s['pad_mode'])
        k.pad_word = int(s.get('pad_word', 0)) & 0xffffffff
        k.convert = int(s.get('convert', 0))
        k.dst_row_pitch = int(s.get('dst_row_pitch', 0))
        k.derive = int(s.get('derive', 0))
    return arr


@_profiled
def scatter_rows_if_id_match(ring, row_bytes, capacity, ids, batch, first_off, count, slot_ids,
                             padding_mask, mask_sample_stride, rows, rows_sample_stride_bytes,
                             rows_row_stride_bytes, winner):
    _check(load().asac_scatter_rows_if_id_match(
        _p(ring), row_bytes, capacity, _p(ids), batch, first_off, count, _p(slot_ids), _p(padding_mask),
        mask_sample_stride, _p(rows), rows_sample_stride_bytes, rows_row_stride_bytes, _p(winner),
        _stream()), 'asac_scatter_rows_if_id_match')


def _ls_rows(loc, scale):
    """loc / scale [..., A] with a dense inner dim and one uniform row stride (dense [rows, A] tensors
    or the two column halves of a dense [rows, 2A] tensor) -> (rows, A, row_stride)."""
    A = loc.shape[-1]
    assert loc.shape == scale.shape and loc.stride() == scale.stride()
    assert loc.stride(-1) == 1 or A == 1
    rs = loc.stride(-2) if loc.dim() >= 2 else A
    for d in range(loc.dim() - 2):            # leading dims must collapse onto the row stride
        assert loc.stride(d) == loc.stride(d + 1) * loc.shape[d + 1], 'loc/scale rows are not uniformly strided'
    return loc.numel() // A, A, rs


@_profiled
def window_aux(bn_indexes, bn_padding_masks, bn_actions, index_x, pad_x, pre_action):
    """bn_* = the L-1 leading rows of the sampled window ([B, L-1(, A)] views) -> index_x i32 [B, L],
    pad_x bool [B, L] (dense), pre_action f32 [B, L, A] (dense, or a column block of a wider [B, L, *] tensor)."""
    B, Lm1 = bn_indexes.shape
    A = bn_actions.shape[-1]
    assert bn_indexes.dtype == torch.int32 and bn_indexes.stride(1) == 1 and bn_padding_masks.stride(1) == 1
    assert bn_padding_masks.element_size() == 1 and bn_actions.stride(2) == 1 and bn_actions.dtype == torch.float32
    assert index_x.is_contiguous() and pad_x.is_contiguous()
    assert pre_action.shape == (B, Lm1 + 1, A) and pre_action.stride(2) == 1 and pre_action.dtype == torch.float32
    assert pre_action.stride(0) == (Lm1 + 1) * pre_action.stride(1)
    _check(load().asac_window_aux(_p(bn_indexes), bn_indexes.stride(0), _p(bn_padding_masks), bn_padding_masks.stride(0),
                                  _p(bn_actions), bn_actions.stride(0), bn_actions.stride(1), B, Lm1 + 1, A,
                                  _p(index_x), _p(pad_x), _p(pre_action), pre_action.stride(1), _stream()),
           'asac_window_aux')


@_profiled
def squash_sample_fwd(loc, scale, eps, a_out, logp_out, x_out=None, action=None, action_offset=0,
                      prob_out=None, prob_offset=0):
    """eps / a_out dense [rows, A]; with `action` ([S, T, >=off+A] view) also writes the stored-action
    probabilities into prob_out ([S, T, >=off+A] view) in the same launch."""
    rows, A, ls = _ls_rows(loc, scale)
    assert eps.is_contiguous() and a_out.is_contiguous()
    T = asb = ast = psb = pst = 0
    if action is not None:
        assert action.dim() == 3 and prob_out.dim() == 3 and action.stride(-1) == 1 and prob_out.stride(-1) == 1
        assert action.shape[0] * action.shape[1] == rows
        T, asb, ast, psb, pst = action.shape[1], action.stride(0), action.stride(1), prob_out.stride(0), prob_out.stride(1)
    _check(load().asac_squash_sample_fwd(_p(loc), _p(scale), ls, _p(eps), rows, A, _p(a_out), _p(logp_out), _p(x_out),
                                         _p(action), T, asb, ast, action_offset, _p(prob_out), psb, pst, prob_offset,
                                         _stream()), 'asac_squash_sample_fwd')


@_profiled
def squash_sample_bwd(loc, scale, eps, grad_a, grad_logp, grad_loc, grad_scale, log_alpha=None):
    """grad_a: dense [rows, A] or [members, rows, A] (one gradient per ensemble member that consumed the
    action: summed in member order by the kernel).  grad_logp None + log_alpha: dL/dlogp = exp(log_alpha)/rows."""
    rows, A, ls = _ls_rows(loc, scale)
    _, _, gs = _ls_rows(grad_loc, grad_scale)
    members, mstride = 1, 0
    if grad_a is not None:
        assert grad_a.is_contiguous() and grad_a.numel() % (rows * A) == 0
        members, mstride = grad_a.numel() // (rows * A), rows * A
    _check(load().asac_squash_sample_bwd(_p(loc), _p(scale), ls, _p(eps), _p(grad_a), members, mstride,
                                         _p(grad_logp), _p(log_alpha), rows, A, _p(grad_loc), _p(grad_scale), gs,
                                         _stream()), 'asac_squash_sample_bwd')


def squash_job(loc, scale, eps=None, a_out=None, logp_out=None, action=None, action_offset=0, prob_out=None,
               prob_offset=0) -> SquashJob:
    """One job of `squash_multi`: a sampling job (eps, a_out, logp_out; optionally also the stored-action
    probabilities) or, without eps, a probability-only job.  Tensor conventions as `squash_sample_fwd` /
    `squash_prob`.  The job holds raw pointers: keep the tensors alive until the launch."""
    rows, A, ls = _ls_rows(loc, scale)
    j = SquashJob()
    j.loc, j.scale, j.ls_row_stride, j.rows, j.A = loc.data_ptr(), scale.data_ptr(), ls, rows, A
    if eps is not None:
        assert eps.is_contiguous() and a_out.is_contiguous() and eps.numel() == rows * A
        j.eps, j.a_tanh_out, j.logp_out = eps.data_ptr(), a_out.data_ptr(), logp_out.data_ptr()
    if action is not None:
        assert action.dim() == 3 and prob_out.dim() == 3 and action.stride(-1) == 1 and prob_out.stride(-1) == 1
        assert action.shape[0] * action.shape[1] == rows and prob_out.shape[:2] == action.shape[:2]
        j.action, j.T = action.data_ptr(), action.shape[1]
        j.action_stride_b, j.action_stride_t, j.action_offset = action.stride(0), action.stride(1), action_offset
        j.prob_out, j.prob_stride_b, j.prob_stride_t, j.prob_offset = \
            prob_out.data_ptr(), prob_out.stride(0), prob_out.stride(1), prob_offset
    else:
        assert eps is not None, 'a job needs eps (sampling) or action (stored-action probabilities)'
    return j


def _sidecar_array(sidecars):
    sidecars = list(sidecars or ())
    assert len(sidecars) <= MAX_SIDECARS
    return ((Sidecar * len(sidecars))(*sidecars) if sidecars else None), len(sidecars)


def sidecar_alpha_adam(logp, target, slot, param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, steps_done,
                       advance_counter=True) -> Sidecar:
    """`alpha_adam_step` as a sidecar of a hosting launch (keeps the tensors alive through `_keep`)"""
    sc = Sidecar(kind=SIDECAR_ALPHA_ADAM, logp=_p(logp), B=logp.numel(), target=float(target), slot=int(slot),
                 param=_p(param), grad=_p(grad), exp_avg=_p(exp_avg), exp_avg_sq=_p(exp_avg_sq), n=param.numel(),
                 lr=float(lr), beta1=float(beta1), beta2=float(beta2), eps=float(eps), steps_done=_p(steps_done),
                 advance_counter=int(bool(advance_counter)))
    sc._keep = (logp, param, grad, exp_avg, exp_avg_sq, steps_done)
    return sc


def sidecar_scatter(kind, ring, row_bytes, capacity, ids, batch, first_off, count, slot_ids, padding_mask,
                    mask_sample_stride, rows, rows_sample_stride_bytes, rows_row_stride_bytes, winner) -> Sidecar:
    """one pass (`SIDECAR_SCATTER_ELECT` / `SIDECAR_SCATTER_WRITE`) of `scatter_rows_if_id_match` as a sidecar"""
    sc = Sidecar(kind=kind, ring=_p(ring), row_bytes=row_bytes, capacity=capacity, ids=_p(ids), batch=batch,
                 first_off=first_off, count=count, slot_ids=_p(slot_ids), padding_mask=_p(padding_mask),
                 mask_sample_stride=mask_sample_stride, rows=_p(rows), rows_sample_stride_bytes=rows_sample_stride_bytes,
                 rows_row_stride_bytes=rows_row_stride_bytes, winner=_p(winner))
    sc._keep = (ring, ids, slot_ids, padding_mask, rows, winner)
    return sc


@_profiled
def squash_multi(jobs, sidecars=None):
    """Run 1..SQUASH_MAX_JOBS `squash_job`s in one launch (+ sidecar jobs as extra workgroups)."""
    assert 1 <= len(jobs) <= SQUASH_MAX_JOBS
    arr = (SquashJob * len(jobs))(*jobs)
    sc, n_sc = _sidecar_array(sidecars)
    _check(load().asac_squash_multi(arr, len(jobs), sc, n_sc, _stream()), 'asac_squash_multi')


@_profiled
def squash_prob(loc, scale, action, action_offset, prob_out, prob_offset):
    """loc/scale: [S, T, A] (uniform row stride); action / prob_out: [S, T, >=offset+A] views."""
    rows, A, ls = _ls_rows(loc, scale)
    S, T = action.shape[0], action.shape[1]
    assert S * T == rows and action.stride(-1) == 1 and prob_out.stride(-1) == 1 and prob_out.shape[:2] == (S, T)
    _check(load().asac_squash_prob(_p(loc), _p(scale), ls, _p(action), T, action.stride(0), action.stride(1),
                                   action_offset, rows, A, _p(prob_out), prob_out.stride(0),
                                   prob_out.stride(1), prob_offset, _stream()), 'asac_squash_prob')


@_profiled
def vtrace_return_min(args: VtraceArgs, sidecars=None, pending_alpha: 'Sidecar | None' = None):
    """`pending_alpha`: the temperature step's sidecar job when it has not run yet (it rides in a LATER launch): the
    return is evaluated with the temperature that job will write"""
    sc, n_sc = _sidecar_array(sidecars)
    _check(load().asac_vtrace_return_min_sc(C.byref(args), sc, n_sc,
                                            C.byref(pending_alpha) if pending_alpha is not None else None, _stream()),
           'asac_vtrace_return_min')


def td_update_ok(B: int, n: int) -> bool:
    """can `td_update` take a batch of B windows of n steps (one workgroup, its LDS)?"""
    return 0 < B <= 1024 and (((2 * B * ((n + 1) | 1) + 3) & ~3) + 2 * ((B + 63) & ~63) + 4) * 4 <= 128 * 1024


@_profiled
def td_update(args: VtraceArgs, tree, capacity, ids, slot_ids, alpha, td_min, td_max, winner, nan_flag, sidecars=None,
              alpha_step: 'Sidecar | None' = None):
    """the TD errors' return (args.q_online / td_error_out set) + the priority update of `ids` with them: one launch"""
    assert ids.dtype == torch.int64 and ids.numel() == args.B and winner.dtype == torch.int32
    sc, n_sc = _sidecar_array(sidecars)
    _check(load().asac_td_update(C.byref(args), _p(tree), capacity, _p(ids), _p(slot_ids), alpha, td_min, td_max,
                                 _p(winner), _p(nan_flag), sc, n_sc,
                                 C.byref(alpha_step) if alpha_step is not None else None, _stream()), 'asac_td_update')


@_profiled
def vtrace_return_direct(args: VtraceArgs, v_n, v_next, pi_prod, mu_prod):
    _check(load().asac_vtrace_return_direct(C.byref(args), _p(v_n), _p(v_next), _p(pi_prod), _p(mu_prod),
                                            _stream()), 'asac_vtrace_return_direct')


@_profiled
def q_loss_fwd_bwd(q, tq, y, w, clip_eps, loss_out, grad_q_out):
    E, B = q.shape[0], q.shape[1]
    _check(load().asac_q_loss_fwd_bwd(_p(q), _p(tq), _p(y), _p(w), E, B, float(clip_eps), _p(loss_out),
                                      _p(grad_q_out), _stream()), 'asac_q_loss_fwd_bwd')


def _rows_view(x):
    """(ptr, row_stride, member_stride) of an input given as [N, K] (shared by every ensemble
    member) or [E, N, K]; the inner dimension must be dense."""
    if x is None:
        return None, 0, 0
    assert x.stride(-1) == 1 or x.shape[-1] == 1
    if x.dim() == 2:
        return _p(x), x.stride(0), 0
    assert x.dim() == 3
    return _p(x), x.stride(1), x.stride(0)


def mlp_flops(desc, E, N, backward=False, param_grads=True) -> float:
    """Algorithmic FLOPs of one fused-MLP pass: 2*rows*in*out per Linear (x2 for dX, x2 for dW)."""
    k, mac = desc.in0 + desc.in1, 0
    for l in range(desc.n_blocks):
        mac += k * desc.width[l]
        k = desc.width[l]
    mac += k * (desc.head_cols[0] + desc.head_cols[1])
    passes = (1 + 1 + (1 if param_grads else 0)) if backward else 1    # recompute + dX (+ dW)
    return 2.0 * mac * N * E * passes


@_profiled
def mlp_forward(desc, params, member_stride, E, x0, x1, N, out):
    global _last_work
    _last_work = mlp_flops(desc, E, N)
    p0, rs0, ms0 = _rows_view(x0)
    p1, rs1, ms1 = _rows_view(x1)
    _check(load().asac_mlp_forward(C.byref(desc), _p(params), member_stride, E, p0, rs0, ms0, p1, rs1, ms1,
                                   N, _p(out), _stream()), 'asac_mlp_forward')


class WindowRows:
    """Marks a non-collapsible [samples, T, K] view (dense last dim) as the row source of a forward job:
    the kernel addresses rows as (sample, t) instead of the caller staging a contiguous copy."""

    def __init__(self, t: torch.Tensor):
        self.t = t
        self.shape = (t.shape[0] * t.shape[1], t.shape[2])


def mlp_job(desc, params, member_stride, E, x0, x1, N, out) -> MlpJob:
    """One forward pass of `mlp_forward_multi` (arguments as `mlp_forward`; raw pointers: keep the tensors
    and the descriptor alive until the launch)."""
    j = MlpJob()
    j.desc, j.params, j.member_stride, j.E, j.N = C.pointer(desc), params.data_ptr(), member_stride, E, N
    if isinstance(x0, WindowRows):      # [samples, T, K] view shared by every member, read in place
        t = x0.t
        assert t.dim() == 3 and t.stride(2) == 1 and t.shape[0] * t.shape[1] == N
        p0, j.x0_row_stride, j.x0_member_stride = _p(t), t.stride(1), 0
        j.x0_window_T, j.x0_sample_stride = t.shape[1], t.stride(0)
    else:
        p0, j.x0_row_stride, j.x0_member_stride = _rows_view(x0)
    p1, j.x1_row_stride, j.x1_member_stride = _rows_view(x1)
    j.x0, j.x1 = p0.value if p0 is not None else None, p1.value if p1 is not None else None
    assert out.is_cuda and out.is_contiguous()
    j.out = out.data_ptr()
    return j


@_profiled
def mlp_forward_multi(jobs, sidecars=None):
    global _last_work
    _last_work = sum(mlp_flops(j.desc.contents, j.E, j.N) for j in jobs)
    assert 1 <= len(jobs) <= MLP_MAX_JOBS
    arr = (MlpJob * len(jobs))(*jobs)
    sc, n_sc = _sidecar_array(sidecars)
    _check(load().asac_mlp_forward_multi(arr, len(jobs), sc, n_sc, _stream()), 'asac_mlp_forward_multi')


def sample_epilogue(job: MlpJob | None = None, eps=None, a_out=None, logp_out=None, T=0, action=None, action_offset=0,
                    prob_out=None, prob_offset=0, eps2=None, t2=0, a2_out=None, logp2_out=None) -> SampleEpilogue:
    """The epilogue of a policy job of `mlp_forward_multi_sampled` (`job` from `mlp_job`; None / no arguments: the job has
    none): the main sample over every row (eps -> a_out, logp_out), the stored actions' probabilities ([samples, T, >= A]
    `action` -> `prob_out`), a second sample at window position `t2` of every sample.  Raw pointers: keep the tensors alive."""
    e = SampleEpilogue()
    if job is None:
        return e
    A, N = job.desc.contents.head_cols[0], job.N
    s = e.sample
    s.rows, s.A, s.T = N, A, T
    if eps is not None:
        assert eps.is_contiguous() and a_out.is_contiguous() and eps.numel() == N * A and logp_out.numel() == N
        assert logp_out.is_contiguous()
        s.eps, s.a_tanh_out, s.logp_out = eps.data_ptr(), a_out.data_ptr(), logp_out.data_ptr()
    if action is not None:
        assert action.dim() == 3 and prob_out.dim() == 3 and action.stride(-1) == 1 and prob_out.stride(-1) == 1
        assert action.shape[0] * action.shape[1] == N and prob_out.shape[:2] == action.shape[:2]
        s.T = action.shape[1]
        assert T in (0, s.T)
        s.action, s.action_stride_b, s.action_stride_t, s.action_offset = \
            action.data_ptr(), action.stride(0), action.stride(1), action_offset
        s.prob_out, s.prob_stride_b, s.prob_stride_t, s.prob_offset = \
            prob_out.data_ptr(), prob_out.stride(0), prob_out.stride(1), prob_offset
    if eps2 is not None:
        assert s.T > 0 and eps2.is_contiguous() and a2_out.is_contiguous() and eps2.numel() == (N // s.T) * A
        assert logp2_out.is_contiguous() and logp2_out.numel() == N // s.T
        e.eps2, e.t2, e.a2_out, e.logp2_out = eps2.data_ptr(), t2, a2_out.data_ptr(), logp2_out.data_ptr()
    return e


def mlp_forward_multi_sampled_ok(jobs, epilogues) -> bool:
    assert len(jobs) == len(epilogues) and 1 <= len(jobs) <= MLP_MAX_JOBS
    return bool(load().asac_mlp_forward_multi_sampled_ok((MlpJob * len(jobs))(*jobs), len(jobs),
                                                         (SampleEpilogue * len(jobs))(*epilogues)))


@_profiled
def mlp_forward_multi_sampled(jobs, epilogues, sidecars=None):
    """`mlp_forward_multi` whose policy jobs sample from / score stored actions under the rows they form (`sample_epilogue`):
    the policy forward and `squash_multi` as one launch"""
    global _last_work
    _last_work = sum(mlp_flops(j.desc.contents, j.E, j.N) for j in jobs)
    assert len(jobs) == len(epilogues) and 1 <= len(jobs) <= MLP_MAX_JOBS
    arr = (MlpJob * len(jobs))(*jobs)
    epi = (SampleEpilogue * len(jobs))(*epilogues)
    sc, n_sc = _sidecar_array(sidecars)
    _check(load().asac_mlp_forward_multi_sampled(arr, len(jobs), epi, sc, n_sc, _stream()), 'asac_mlp_forward_multi_sampled')


def pi_q_job(pi_job: MlpJob, q_job: MlpJob, eps, a_out, logp_out, T, action=None, action_offset=0, prob_out=None,
             prob_offset=0, eps2=None, t2=0, a2_out=None, logp2_out=None) -> PiQJob:
    """The fused policy -> sample -> critics forward (`policy_sample_q_forward`): `pi_job` / `q_job` from `mlp_job`
    on the same rows ([samples, T] flattened); the main sample (eps -> a_out, logp_out), optionally the stored
    actions' probabilities and a second sample at window position `t2`.  Raw pointers: keep the tensors alive."""
    j = PiQJob()
    j.pi, j.q = pi_job, q_job
    A = pi_job.desc.contents.head_cols[0]
    N = pi_job.N
    assert eps.is_contiguous() and a_out.is_contiguous() and eps.numel() == N * A and logp_out.numel() == N
    s = j.sample
    s.eps, s.a_tanh_out, s.logp_out, s.rows, s.A, s.T = eps.data_ptr(), a_out.data_ptr(), logp_out.data_ptr(), N, A, T
    if action is not None:
        assert action.dim() == 3 and prob_out.dim() == 3 and action.stride(-1) == 1 and prob_out.stride(-1) == 1
        assert action.shape[0] * action.shape[1] == N and action.shape[1] == T and prob_out.shape[:2] == action.shape[:2]
        s.action, s.action_stride_b, s.action_stride_t, s.action_offset = \
            action.data_ptr(), action.stride(0), action.stride(1), action_offset
        s.prob_out, s.prob_stride_b, s.prob_stride_t, s.prob_offset = \
            prob_out.data_ptr(), prob_out.stride(0), prob_out.stride(1), prob_offset
    if eps2 is not None:
        assert eps2.is_contiguous() and a2_out.is_contiguous() and eps2.numel() == (N // T) * A
        j.eps2, j.t2, j.a2_out, j.logp2_out = eps2.data_ptr(), t2, a2_out.data_ptr(), logp2_out.data_ptr()
    return j


def policy_sample_q_forward_ok(job: PiQJob) -> bool:
    return bool(load().asac_policy_sample_q_forward_ok(C.byref(job)))


@_profiled
def policy_sample_q_forward(job: PiQJob, extra_jobs=(), sidecars=None):
    global _last_work
    # algorithmic work: the policy once (the E workgroups of a tile repeat it side by side), every critic once
    _last_work = (mlp_flops(job.pi.desc.contents, 1, job.pi.N) + mlp_flops(job.q.desc.contents, job.q.E, job.q.N)
                  + sum(mlp_flops(j.desc.contents, j.E, j.N) for j in extra_jobs))
    extra_jobs = list(extra_jobs)
    assert len(extra_jobs) <= MLP_MAX_JOBS
    arr = (MlpJob * len(extra_jobs))(*extra_jobs) if extra_jobs else None
    sc, n_sc = _sidecar_array(sidecars)
    _check(load().asac_policy_sample_q_forward(C.byref(job), arr, len(extra_jobs), sc, n_sc, _stream()),
           'asac_policy_sample_q_forward')


def mlp_backward_workspace(member_stride, E, N) -> int:
    return int(load().asac_mlp_backward_workspace(member_stride, E, N))


def mlp_backward_tiles(N, E) -> int:
    """row tiles (= per-tile partial slabs) the backward of an [E][N] pass uses"""
    return int(load().asac_mlp_backward_tiles(N, E))


MLP_REDUCE_OVERWRITE, MLP_REDUCE_ACCUMULATE, MLP_REDUCE_DEFER = 0, 1, 2


def mlp_param_extent(desc) -> int:
    return int(load().asac_mlp_param_extent(C.byref(desc)))


@_profiled
def mlp_backward_qloss(desc, params, member_stride, E, x0, x1, N, target_q, y, weights, clip_eps, loss_out,
                       grad_params, workspace, reduce_mode, grad_x0=None):
    """Q loss + backward of the stock Q ensemble in one launch (parameter gradients; with `grad_x0` [E, N, in0] also
    the members' gradients w.r.t. the state input)."""
    global _last_work
    _last_work = mlp_flops(desc, E, N, backward=True, param_grads=True)
    p0, rs0, ms0 = _rows_view(x0)
    p1, rs1, ms1 = _rows_view(x1)
    assert target_q.is_contiguous() and target_q.numel() == E * N and y.is_contiguous() and y.numel() == N
    assert grad_x0 is None or (grad_x0.is_contiguous() and grad_x0.numel() == E * N * desc.in0)
    _check(load().asac_mlp_backward_qloss_gx(C.byref(desc), _p(params), member_stride, E, p0, rs0, ms0, p1, rs1, ms1, N,
                                             _p(target_q), _p(y), _p(weights), float(clip_eps), _p(loss_out), _p(grad_x0),
                                             _p(grad_params), _p(workspace), int(reduce_mode), _stream()),
           'asac_mlp_backward_qloss')


def mlp_backward_qloss_return_ok(desc, params, member_stride, E, N, ret: VtraceArgs) -> bool:
    return bool(load().asac_mlp_backward_qloss_return_ok(C.byref(desc), _p(params), member_stride, E, N, C.byref(ret)))


@_profiled
def mlp_backward_qloss_return(desc, params, member_stride, E, x0, x1, N, target_q, ret: VtraceArgs, weights, clip_eps,
                              loss_out, grad_params, workspace, reduce_mode, grad_x0=None):
    """`mlp_backward_qloss` whose workgroups form the return target `ret` describes themselves (no return launch)"""
    global _last_work
    _last_work = mlp_flops(desc, E, N, backward=True, param_grads=True)
    p0, rs0, ms0 = _rows_view(x0)
    p1, rs1, ms1 = _rows_view(x1)
    assert target_q.is_contiguous() and target_q.numel() == E * N
    assert grad_x0 is None or (grad_x0.is_contiguous() and grad_x0.numel() == E * N * desc.in0)
    _check(load().asac_mlp_backward_qloss_return(C.byref(desc), _p(params), member_stride, E, p0, rs0, ms0, p1, rs1, ms1,
                                                 N, _p(target_q), C.byref(ret), _p(weights), float(clip_eps),
                                                 _p(loss_out), _p(grad_x0), _p(grad_params), _p(workspace), int(reduce_mode),
                                                 _stream()), 'asac_mlp_backward_qloss_return')


@_profiled
def mlp_backward_policy_q(desc, params, member_stride, E, x0, x1, N, q_table, subset, E_sample, grad_x1):
    """Policy step: action gradients of mean_b(-min_{e in subset} q_e) through the stock Q ensemble."""
    global _last_work
    _last_work = mlp_flops(desc, E, N, backward=True, param_grads=False)
    p0, rs0, ms0 = _rows_view(x0)
    p1, rs1, ms1 = _rows_view(x1)
    assert q_table.is_contiguous() and q_table.numel() == E * N and grad_x1.is_contiguous()
    _check(load().asac_mlp_backward_policy_q(C.byref(desc), _p(params), member_stride, E, p0, rs0, ms0, p1, rs1, ms1,
                                             N, _p(q_table), _p(subset), E_sample, _p(grad_x1), _stream()),
           'asac_mlp_backward_policy_q')


@_profiled
def mlp_backward_policy_sample(desc, params, member_stride, x0, N, eps, grad_a, log_alpha, grad_params, workspace,
                               reduce_mode):
    """Policy step: sampling backward + policy backward of the stock Gaussian-head policy in one launch."""
    global _last_work
    _last_work = mlp_flops(desc, 1, N, backward=True, param_grads=True)
    p0, rs0, _ = _rows_view(x0)
    A = desc.head_cols[0]
    assert eps.is_contiguous() and eps.numel() == N * A and grad_a.is_contiguous() and grad_a.numel() % (N * A) == 0
    _check(load().asac_mlp_backward_policy_sample(C.byref(desc), _p(params), member_stride, p0, rs0, N, _p(eps),
                                                  _p(grad_a), grad_a.numel() // (N * A), _p(log_alpha),
                                                  _p(grad_params), _p(workspace), int(reduce_mode), _stream()),
           'asac_mlp_backward_policy_sample')


def policy_step_fused_ok(q_desc, q_params, q_member_stride, pi_desc, pi_params, pi_member_stride, N) -> bool:
    return bool(load().asac_policy_step_fused_ok(C.byref(q_desc), _p(q_params), q_member_stride, C.byref(pi_desc),
                                                 _p(pi_params), pi_member_stride, N))


@_profiled
def policy_step_fused(q_desc, q_params, q_member_stride, pi_desc, pi_params, pi_member_stride, x, N, action, eps,
                      log_alpha, q_out, pi_grad_params, workspace, reduce_mode, subset=None, a_out=None, logp_out=None,
                      ls_out=None):
    """The stock networks' whole policy step (two critics) in one launch: critics forward on (x, action), objective
    gradient, critics backward to the action, sampling backward, policy backward (per-tile partials / reduced)."""
    global _last_work
    _last_work = (mlp_flops(q_desc, 2, N, backward=True, param_grads=False)
                  + mlp_flops(pi_desc, 1, N, backward=True, param_grads=True))
    p0, rs0, _ = _rows_view(x)
    A = pi_desc.head_cols[0]
    assert eps.is_contiguous() and eps.numel() == N * A
    if action is not None:
        assert action.is_contiguous() and action.numel() == N * A
    else:       # sampled on chip
        assert a_out.is_contiguous() and a_out.numel() == N * A and logp_out.is_contiguous() and logp_out.numel() == N
        assert ls_out is None or (ls_out.is_contiguous() and ls_out.numel() == 2 * N * A)
    assert q_out is None or (q_out.is_contiguous() and q_out.numel() % N == 0 and q_out.numel() >= 2 * N)
    assert subset is None or (subset.dtype == torch.int32 and subset.numel() == 2)
    _check(load().asac_policy_step_fused(C.byref(q_desc), _p(q_params), q_member_stride, C.byref(pi_desc), _p(pi_params),
                                         pi_member_stride, p0, rs0, N, _p(action), _p(eps), _p(log_alpha), _p(subset),
                                         _p(q_out), _p(a_out), _p(logp_out), _p(ls_out), _p(pi_grad_params), _p(workspace),
                                         int(reduce_mode), _stream()),
           'asac_policy_step_fused')


@_profiled
def adam_step_partials(param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, steps_done, workspace, tiles, E,
                       member_stride, used, accumulate, loss_out=None, loss_rows=0):
    """Adam over E member blocks whose gradients are still per-tile partial sums in `workspace`."""
    assert param.numel() >= E * member_stride and steps_done.dtype == torch.int64
    _check(load().asac_adam_step_partials(_p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), lr, beta1, beta2, eps,
                                          _p(steps_done), _p(workspace), tiles, E, member_stride, used,
                                          int(bool(accumulate)), _p(loss_out), loss_rows, _stream()),
           'asac_adam_step_partials')


@_profiled
def mlp_backward(desc, params, member_stride, E, x0, x1, N, grad_out, grad_x0, grad_x1, grad_params, workspace,
                 reduce_mode=MLP_REDUCE_ACCUMULATE):
    global _last_work
    _last_work = mlp_flops(desc, E, N, backward=True, param_grads=grad_params is not None)
    p0, rs0, ms0 = _rows_view(x0)
    p1, rs1, ms1 = _rows_view(x1)
    _check(load().asac_mlp_backward(C.byref(desc), _p(params), member_stride, E, p0, rs0, ms0, p1, rs1, ms1, N,
                                    _p(grad_out), _p(grad_x0), _p(grad_x1), _p(grad_params), _p(workspace),
                                    int(reduce_mode), _stream()), 'asac_mlp_backward')


def attention_supported(Lq: int, Lk: int, D: int) -> bool:
    return bool(load().asac_attention_supported(int(Lq), int(Lk), int(D)))


@_profiled
def attention_forward(q, k, v, mask, out, weights, keep):
    """q [B, Lq, D], k / v [B, Lk, D] dense f32; mask bool / u8 broadcastable view of [B, Lq, Lk] (True = blocked)
    or None -> out [B, Lq, D], weights [B, Lq, Lk] (x keep), keep [B, Lq]."""
    _dense_f32(q, k, v, out, weights, keep)
    B, Lq, D = q.shape
    sb = si = sj = 0
    if mask is not None:
        assert mask.element_size() == 1 and mask.dim() == 3
        sb, si, sj = (0 if mask.shape[d] == 1 else mask.stride(d) for d in range(3))
    _check(load().asac_attention_forward(_p(q), _p(k), _p(v), _p(mask), sb, si, sj, B, Lq, k.shape[1], D, _p(out),
                                         _p(weights), _p(keep), _stream()), 'asac_attention_forward')


@_profiled
def attention_backward(q, k, v, weights, grad_out, grad_weights, grad_q, grad_k, grad_v):
    _dense_f32(q, k, v, weights, grad_out, grad_weights, grad_q, grad_k, grad_v)
    B, Lq, D = q.shape
    _check(load().asac_attention_backward(_p(q), _p(k), _p(v), _p(weights), _p(grad_out), _p(grad_weights), B, Lq,
                                          k.shape[1], D, _p(grad_q), _p(grad_k), _p(grad_v), _stream()),
           'asac_attention_backward')


def _proj_ptrs(params):
    """6 tensors (Wq, bq, Wk, bk, Wv, bv) or 8 (+ Wo, bo: the output ResBlock)"""
    assert len(params) in (6, 8)
    arr = (C.c_void_p * 8)()
    for i, t in enumerate(params):
        assert t.is_cuda and t.is_contiguous() and t.dtype == torch.float32
        arr[i] = t.data_ptr()
    return arr


def _rows3(x):
    """[B, L, E] view with a dense last dim -> (ptr, batch stride, row stride)"""
    assert x.dim() == 3 and x.stride(2) == 1 and x.dtype == torch.float32 and x.is_cuda
    return _p(x), x.stride(0), x.stride(1)


def attention_proj_workspace(B, Lq, Lk, E) -> int:
    return int(load().asac_attention_proj_workspace(B, Lq, Lk, E))


def _row_mask(row_zero):
    if row_zero is None:
        return None, 0, 0
    assert row_zero.dim() == 2 and row_zero.element_size() == 1 and row_zero.is_cuda
    return _p(row_zero), row_zero.stride(0), row_zero.stride(1)


@_profiled
def attention_proj_forward(xq, xk, params, mask, out, weights, keep, attn_out=None, row_zero=None):
    """q / k / v projections (params = Wq, bq, Wk, bk, Wv, bv [, Wo, bo]) + attention core [+ output ResBlock and
    the dead-row rule] in one launch; xq [B, Lq, E] and xk [B, Lk, E] may be strided views (dense last dim);
    attn_out [B, Lq, E] receives the attention output when the output block is on chip."""
    _dense_f32(out, weights, keep, attn_out)
    pq, qsb, qsr = _rows3(xq)
    pk, ksb, ksr = _rows3(xk)
    sb = si = sj = 0
    if mask is not None:
        assert mask.element_size() == 1 and mask.dim() == 3
        sb, si, sj = (0 if mask.shape[d] == 1 else mask.stride(d) for d in range(3))
    _check(load().asac_attention_proj_forward(pq, qsb, qsr, pk, ksb, ksr, _proj_ptrs(params), _p(mask), sb, si, sj,
                                              xq.shape[0], xq.shape[1], xk.shape[1], xq.shape[2], _p(out), _p(weights),
                                              _p(keep), _p(attn_out), *_row_mask(row_zero), _stream()),
           'asac_attention_proj_forward')


@_profiled
def attention_proj_backward(xq, xk, params, weights, grad_out, grad_weights, grad_xq, grad_xk, grad_params, accumulate,
                            workspace, keep=None, attn_out=None, row_zero=None):
    """grad_out: [B, Lq, E] view with a dense last dim (read in place); grad_xq None: the queries are the last Lq key
    rows and their gradient is added into those rows of grad_xk by the launch"""
    _dense_f32(weights, grad_weights, grad_xq, grad_xk, grad_params, workspace, keep, attn_out)
    pq, qsb, qsr = _rows3(xq)
    pk, ksb, ksr = _rows3(xk)
    pg, gsb, gsr = _rows3(grad_out)
    _check(load().asac_attention_proj_backward(pq, qsb, qsr, pk, ksb, ksr, _proj_ptrs(params), _p(weights), _p(keep),
                                               _p(attn_out), pg, gsb, gsr, _p(grad_weights), *_row_mask(row_zero),
                                               xq.shape[0], xq.shape[1], xk.shape[1], xq.shape[2],
                                               _p(grad_xq), _p(grad_xk), _p(grad_params),
                                               _sum_mode(accumulate),
                                               _p(workspace), _stream()), 'asac_attention_proj_backward')


ATTN_SUM_DEFER = CONV_SUM_DEFER = SUM_DEFER = 2      # `accumulate` of attention_proj_backward / conv2_backward*: the partials
                                                     # stay in the workspace (for `sum_partials_multi`)
SUM_PARTIALS_MAX_JOBS = 16


def attention_proj_partials(workspace, E: int, output_block: bool):
    """-> (slabs, n) of the partials an `attention_proj_backward(..., accumulate=ATTN_SUM_DEFER)` left in `workspace`"""
    return workspace.numel() // (4 * (E * E + E)), (4 if output_block else 3) * (E * E + E)


@_profiled
def sum_partials_multi(jobs):
    """jobs: [(partial, slabs, slices, slab_stride, n, out, accumulate), ...] (at most SUM_PARTIALS_MAX_JOBS) — out[:n]
    (+)= the slabs of `partial` in the fixed order of the launch each job stands for (slices 16: the sliced reductions of
    `mlp_backward` with >= 64 tiles and of `attention_proj_backward`; 1: slab order), as ONE launch"""
    assert 1 <= len(jobs) <= SUM_PARTIALS_MAX_JOBS
    arr = (PartialSum * len(jobs))()
    for k, (partial, slabs, slices, slab_stride, n, out, accumulate) in enumerate(jobs):
        assert partial.dtype == out.dtype == torch.float32 and partial.is_cuda and out.is_cuda
        assert partial.numel() >= (slabs - 1) * slab_stride + n and out.numel() >= n
        arr[k] = PartialSum(partial.data_ptr(), out.data_ptr(), int(slab_stride), int(n), int(slabs), int(slices),
                            int(bool(accumulate)), 0)
    _check(load().asac_sum_partials_multi(len(jobs), arr, _stream()), 'asac_sum_partials_multi')


LINEAR_TANH_MAX_IN, LINEAR_TANH_MAX_OUT = 64, 16


def _rows2(x):
    """[N, K] view with a dense last dim -> (ptr, row stride)"""
    assert x.dim() == 2 and x.stride(1) == 1 and x.dtype == torch.float32 and x.is_cuda
    return _p(x), x.stride(0)


def linear_tanh_workspace(N, K, O) -> int:
    return int(load().asac_linear_tanh_workspace(N, K, O))


@_profiled
def linear_tanh_forward(x, weight, bias, y):
    """y[N, O] = tanh(x[N, K] weight[O, K]^T + bias) in one launch; x may have a row stride"""
    _dense_f32(weight, bias, y)
    px, sx = _rows2(x)
    _check(load().asac_linear_tanh_forward(px, sx, _p(weight), _p(bias), x.shape[0], x.shape[1], weight.shape[0], _p(y),
                                           _stream()), 'asac_linear_tanh_forward')


@_profiled
def linear_tanh_backward(x, weight, y, grad_y, grad_x, grad_params, accumulate, workspace):
    """grad_x [N, K] (or None) and the packed (weight | bias) gradient from grad_y [N, O]; `workspace` holds
    linear_tanh_workspace floats and must be zero before its first use"""
    _dense_f32(weight, y, grad_y, grad_x, grad_params, workspace)
    px, sx = _rows2(x)
    _check(load().asac_linear_tanh_backward(px, sx, _p(weight), _p(y), _p(grad_y), x.shape[0], x.shape[1],
                                            weight.shape[0], _p(grad_x), _p(grad_params), int(bool(accumulate)),
                                            _p(workspace), _stream()), 'asac_linear_tanh_backward')


MASKED_MSE_MAX = 1 << 20
_MSE_WS = {}


def _window3(t):
    """[B, T, K] f32 view with a dense last dim -> (pointer, batch stride, step stride)"""
    assert t.dim() == 3 and t.dtype == torch.float32 and t.is_cuda and (t.stride(2) == 1 or t.shape[2] == 1)
    return _p(t), t.stride(0), t.stride(1)


@_profiled
def curiosity_bonus(approx, actual, reward, strength):
    """reward[B, T] += strength * 0.5 * sum_k (approx - actual)^2 in place (approx dense [B, T, K]; actual, reward views)"""
    B, T, K = approx.shape
    assert approx.is_contiguous() and actual.shape == approx.shape and reward.shape == (B, T)
    assert reward.dtype == torch.float32 and (reward.stride(1) == 1 or T == 1)
    pa, sb, st = _window3(actual)
    _check(load().asac_curiosity_bonus(_p(approx), pa, sb, st, _p(reward), reward.stride(0), B, T, K, float(strength),
                                       _stream()), 'asac_curiosity_bonus')


@_profiled
def masked_mse(pred, target, padding_mask, grad_out, loss_out):
    """loss_out <- mean over all elements of ((pred - target) * ~mask)^2, grad_out <- its gradient w.r.t. pred"""
    B, T, K = pred.shape
    assert pred.is_contiguous() and grad_out.is_contiguous() and grad_out.shape == pred.shape and target.shape == pred.shape
    pt, sb, st = _window3(target)
    pm, ms = None, 0
    if padding_mask is not None:
        assert padding_mask.shape == (B, T) and padding_mask.element_size() == 1 and (padding_mask.stride(1) == 1 or T == 1)
        pm, ms = _p(padding_mask), padding_mask.stride(0)
    key = (pred.numel(), pred.device)
    if key not in _MSE_WS:      # zero before first use, left zero by every launch
        _MSE_WS[key] = torch.zeros(int(load().asac_masked_mse_workspace(pred.numel())), dtype=torch.float32, device=pred.device)
    _check(load().asac_masked_mse(_p(pred), pt, sb, st, pm, ms, B, T, K, _p(grad_out), _p(loss_out), _p(_MSE_WS[key]),
                                  _stream()), 'asac_masked_mse')


@_profiled
def mse_mean_grad(pred, target, grad_out, loss_out, workspace, grad_scale: float = 1.0):
    """loss_out <- mean((pred - target)^2), grad_out <- its gradient w.r.t. pred; pred [B, T, K] dense, target a [B, T, K]
    view with a dense last dimension (see `mse_mean_grad_ok`); `workspace`: zeros(mse_mean_grad_workspace()), kept by the caller"""
    B, T, K = pred.shape
    pt, sb, st = _window3(target)
    _check(load().asac_mse_mean_grad(_p(pred), pt, sb, st, B, T, K, float(grad_scale), _p(grad_out), _p(loss_out),
                                     _p(workspace), _stream()), 'asac_mse_mean_grad')


def mse_mean_grad_workspace() -> int:
    return int(load().asac_mse_mean_grad_workspace())


def mse_mean_grad_ok(pred, target) -> bool:
    return (pred.is_cuda and pred.dtype == torch.float32 and target.dtype == torch.float32 and pred.dim() == 3
            and pred.is_contiguous() and target.shape == pred.shape and target.stride(2) == 1 and pred.shape[2] % 4 == 0
            and target.stride(0) % 4 == 0 and target.stride(1) % 4 == 0 and pred.data_ptr() % 16 == 0
            and target.data_ptr() % 16 == 0 and 0 < pred.numel() < 2 ** 33)


_NLL_WS = {}


@_profiled
def normal_nll_kl(loc, scale, target, kl_weight, grad_loc, grad_scale, out):
    """out[0] <- -mean(log N(target; loc, scale)) + kl_weight * mean(KL(N(loc, scale) || N(0, 1))), out[1] <- mean
    entropy, grad_loc / grad_scale <- d out[0] / d loc, / d scale ([B, T, K] views in, dense gradients out)"""
    B, T, K = loc.shape
    assert scale.shape == loc.shape and target.shape == loc.shape and grad_loc.is_contiguous() and grad_scale.is_contiguous()
    assert grad_loc.numel() == loc.numel() == grad_scale.numel() and out.numel() == 2 and out.is_contiguous()
    pl, lb, lt = _window3(loc)
    ps, sb, st = _window3(scale)
    pt, tb, tt = _window3(target)
    key = (loc.numel(), loc.device)
    if key not in _NLL_WS:
        _NLL_WS[key] = torch.zeros(int(load().asac_normal_nll_kl_workspace(loc.numel())), dtype=torch.float32, device=loc.device)
    _check(load().asac_normal_nll_kl(pl, lb, lt, ps, sb, st, pt, tb, tt, B, T, K, float(kl_weight), _p(grad_loc),
                                     _p(grad_scale), _p(out), _p(_NLL_WS[key]), _stream()), 'asac_normal_nll_kl')


@_profiled
def normal_nll_kl_logstd(raw, scale_min, scale_max, target, kl_weight, grad_raw, out):
    """`normal_nll_kl` of N(mean, clamp(exp(logstd), scale_min, scale_max)) with (mean | logstd) = the halves of raw
    [B, T, 2K]; grad_raw [B, T, 2K] dense <- d out[0] / d raw"""
    B, T, K2 = raw.shape
    K = K2 // 2
    assert K2 == 2 * K and raw.stride(2) == 1 and target.shape == (B, T, K) and target.stride(2) == 1
    assert grad_raw.is_contiguous() and grad_raw.shape == raw.shape and out.numel() == 2 and out.is_contiguous()
    _dense_f32(grad_raw, out)
    key = (B * T * K, raw.device)
    if key not in _NLL_WS:
        _NLL_WS[key] = torch.zeros(int(load().asac_normal_nll_kl_workspace(B * T * K)), dtype=torch.float32, device=raw.device)
    _check(load().asac_normal_nll_kl_logstd(_p(raw), raw.stride(0), raw.stride(1), float(scale_min), float(scale_max),
                                            _p(target), target.stride(0), target.stride(1), B, T, K, float(kl_weight),
                                            _p(grad_raw), _p(out), _p(_NLL_WS[key]), _stream()), 'asac_normal_nll_kl_logstd')


@_profiled
def linear_tanh_forward2(x0, x1, weight, bias, y):
    """y[N, O] = tanh([x0 | x1][N, K0 + K1] weight^T + bias): the concatenation read in place (x1 may be None)"""
    _dense_f32(weight, bias, y)
    if isinstance(x0, WindowRows):      # [samples, T, K0] slice of the sampled windows, read in place
        t = x0.t
        assert t.dim() == 3 and t.stride(2) == 1 and t.dtype == torch.float32
        p0, s0, T, sb, N, K0 = _p(t), t.stride(1), t.shape[1], t.stride(0), t.shape[0] * t.shape[1], t.shape[2]
    else:
        (p0, s0), T, sb, N, K0 = _rows2(x0), 0, 0, x0.shape[0], x0.shape[1]
    p1, s1 = _rows2(x1) if x1 is not None else (None, 0)
    assert x1 is None or x1.shape[0] == N
    _check(load().asac_linear_tanh_forward2w(p0, s0, T, sb, K0, p1, s1, 0 if x1 is None else x1.shape[1], _p(weight),
                                             _p(bias), N, weight.shape[0], _p(y), _stream()),
           'asac_linear_tanh_forward2')


@_profiled
def linear_tanh_backward2(x0, x1, weight, y, grad_y, grad_x0, grad_x1, grad_params, accumulate, workspace,
                          members=1, window=1, position=0):
    """`linear_tanh_backward` over the two-part input; grad_y [members, N / window, O]: row r's output gradient is
    sum_e grad_y[e, r // window] when r % window == position, zero otherwise (1, 1, 0: dense [N, O])."""
    _dense_f32(weight, y, grad_y, grad_x0, grad_x1, grad_params, workspace)
    if isinstance(x0, WindowRows):      # [samples, T, K0] slice of the sampled windows, read in place
        t = x0.t
        assert t.dim() == 3 and t.stride(2) == 1 and t.dtype == torch.float32
        p0, s0, T, sb, N, K0 = _p(t), t.stride(1), t.shape[1], t.stride(0), t.shape[0] * t.shape[1], t.shape[2]
    else:
        (p0, s0), T, sb, N, K0 = _rows2(x0), 0, 0, x0.shape[0], x0.shape[1]
    p1, s1 = _rows2(x1) if x1 is not None else (None, 0)
    O = weight.shape[0]
    assert grad_y.numel() == members * (N // window) * O and N % window == 0
    _check(load().asac_linear_tanh_backward2w(p0, s0, T, sb, K0, p1, s1, 0 if x1 is None else x1.shape[1], _p(weight),
                                              _p(y), _p(grad_y), members, window, position, N, O, _p(grad_x0),
                                              _p(grad_x1), _p(grad_params), _sum_mode(accumulate), _p(workspace),
                                              _stream()), 'asac_linear_tanh_backward2')


def conv2_desc(channels, height, width, out1, kernel1, stride1, out2, kernel2, stride2) -> Conv2Desc:
    return Conv2Desc(channels, height, width, out1, kernel1, stride1, out2, kernel2, stride2)


def conv2_supported(desc) -> bool:
    return bool(load().asac_conv2_supported(C.byref(desc)))


def conv2_param_count(desc) -> int:
    return int(load().asac_conv2_param_count(C.byref(desc)))


def _sum_mode(accumulate) -> int:
    """False / True / SUM_DEFER -> the entry points' `accumulate` argument"""
    return SUM_DEFER if (accumulate is not True and accumulate is not False and accumulate == SUM_DEFER) else int(bool(accumulate))


def conv2_backward_slabs(desc, N, n_cot=1) -> int:
    """partial slabs a backward launch over N frames with n_cot cotangents leaves in its workspace (`SUM_DEFER`)"""
    return int(load().asac_conv2_backward_slabs(C.byref(desc), N, n_cot))


def conv2_backward_workspace(desc, N) -> int:
    return int(load().asac_conv2_backward_workspace(C.byref(desc), N))


def _dense_f32(*ts):
    for t in ts:
        assert t is None or (t.is_cuda and t.is_contiguous() and t.dtype == torch.float32)


def conv2_flops(desc, N, backward=False) -> float:
    """multiply-adds x 2 of the two implicit GEMMs (backward: both weight-gradient GEMMs + the layer-1
    activation gradient)"""
    h1 = (desc.height - desc.kernel1) // desc.stride1 + 1
    w1 = (desc.width - desc.kernel1) // desc.stride1 + 1
    h2, w2 = (h1 - desc.kernel2) // desc.stride2 + 1, (w1 - desc.kernel2) // desc.stride2 + 1
    g1 = h1 * w1 * desc.channels * desc.kernel1 ** 2 * desc.out1
    g2 = h2 * w2 * desc.out1 * desc.kernel2 ** 2 * desc.out2
    return 2.0 * N * ((g1 + 2 * g2) if backward else (g1 + g2))


@_profiled
def conv2_forward(desc, x, w1, b1, w2, b2, y, z1_out=None, z2_out=None):
    """x [N, C, H, W] -> y [N, out2*H2*W2] (Conv2d GELU Conv2d GELU, flattened channel-major); z1_out
    [N, H1*W1, out1] / z2_out [N, out2*H2*W2]: pre-activations for the backward (both or neither)."""
    global _last_work
    _last_work = conv2_flops(desc, x.shape[0])
    _dense_f32(x, w1, b1, w2, b2, y, z1_out, z2_out)
    _check(load().asac_conv2_forward(C.byref(desc), _p(x), x.shape[0], _p(w1), _p(b1), _p(w2), _p(b2), _p(y),
                                     _p(z1_out), _p(z2_out), _stream()), 'asac_conv2_forward')


def conv2_group_frames(desc) -> int:
    return int(load().asac_conv2_group_frames(C.byref(desc)))


def conv2_tiles(desc) -> int:
    """blocks per frame (1: the whole frame in one piece; > 1: frames whose second-layer map exceeds 16 positions)"""
    return int(load().asac_conv2_tiles(C.byref(desc)))


def conv2_z1_floats(desc, N) -> int:
    """size of the `z1_out` buffer of a training forward over N frames"""
    return int(load().asac_conv2_z1_floats(C.byref(desc), int(N)))


@_profiled
def conv2_forward_windows(desc, x, w1, b1, w2, b2, y, z1_out=None, z2_out=None):
    """`conv2_forward` over x [B, T, C, H, W] = a slice of the sampled windows (dense frames, samples x.stride(0)
    floats apart, T a multiple of `conv2_group_frames`), read in place -> y [B * T, out]"""
    global _last_work
    B, T = x.shape[:2]
    _last_work = conv2_flops(desc, B * T)
    _dense_f32(w1, b1, w2, b2, y, z1_out, z2_out)
    assert x.dtype == torch.float32 and x.is_cuda and x[0].is_contiguous()
    _check(load().asac_conv2_forward_windows(C.byref(desc), _p(x), B * T, T, x.stride(0), _p(w1), _p(b1), _p(w2), _p(b2),
                                             _p(y), _p(z1_out), _p(z2_out), _stream()), 'asac_conv2_forward_windows')


@_profiled
def conv2_backward_windows(desc, x, w2, z1, z2, grad_y, grad_params, workspace, accumulate=False):
    """`conv2_backward` with x [B, T, C, H, W] a slice of the sampled windows read in place (see `conv2_forward_windows`)"""
    global _last_work
    B, T = x.shape[:2]
    _last_work = conv2_flops(desc, B * T, backward=True)
    _dense_f32(w2, z1, z2, grad_y, grad_params, workspace)
    assert x.dtype == torch.float32 and x.is_cuda and x[0].is_contiguous()
    _check(load().asac_conv2_backward_windows(C.byref(desc), _p(x), B * T, T, x.stride(0), _p(w2), _p(z1), _p(z2),
                                              _p(grad_y), _p(grad_params), _sum_mode(accumulate), _p(workspace),
                                              _stream()), 'asac_conv2_backward_windows')


CONV2_MAX_COTANGENTS = 4


def conv2_backward_multi_max(desc) -> int:
    """cotangents ONE launch of `conv2_backward_multi` takes for these frames (more: launches of at most this many)"""
    return int(load().asac_conv2_backward_multi_max(C.byref(desc)))


@_profiled
def conv2_backward_multi(desc, x, w2, z1, z2, grad_ys, grads_out, workspace, accumulate=False):
    """Several backward walks of ONE forward pass as one launch: `grad_ys` = 1..4 output gradients, `grads_out`
    [len(grad_ys), param_count] <- the packed gradients per cotangent, bit-identical to one `conv2_backward(_windows)` each.
    `x` [N, C, H, W] dense, or [B, T, C, H, W] a slice of the sampled windows read in place; workspace: len(grad_ys) x
    `conv2_backward_workspace` floats."""
    global _last_work
    nc = len(grad_ys)
    assert 1 <= nc <= CONV2_MAX_COTANGENTS and grads_out.shape[0] == nc
    windows = x.dim() == 5
    n_frames = x.shape[0] * x.shape[1] if windows else x.shape[0]
    _last_work = nc * conv2_flops(desc, n_frames, backward=True)
    _dense_f32(w2, z1, z2, grads_out, workspace, *grad_ys)
    if windows:
        assert x.dtype == torch.float32 and x.is_cuda and x[0].is_contiguous()
        fps, stride = x.shape[1], x.stride(0)
    else:
        _dense_f32(x)
        fps, stride = 0, 0
    ptrs = (C.c_void_p * nc)(*[g.data_ptr() for g in grad_ys])
    _check(load().asac_conv2_backward_multi(C.byref(desc), _p(x), n_frames, fps, stride, _p(w2), _p(z1), _p(z2), ptrs, nc,
                                            _p(grads_out), _sum_mode(accumulate), _p(workspace), _stream()),
           'asac_conv2_backward_multi')


@_profiled
def conv2_backward(desc, x, w2, z1, z2, grad_y, grad_params, workspace, accumulate=False):
    """-> grad_params (packed w1 | b1 | w2 | b2: written, or added with `accumulate`)."""
    global _last_work
    _last_work = conv2_flops(desc, x.shape[0], backward=True)
    _dense_f32(x, w2, z1, z2, grad_y, grad_params, workspace)
    _check(load().asac_conv2_backward(C.byref(desc), _p(x), x.shape[0], _p(w2), _p(z1), _p(z2), _p(grad_y),
                                      _p(grad_params), _sum_mode(accumulate), _p(workspace), _stream()),
           'asac_conv2_backward')


def gru_desc(input_size: int, hidden: int, layers: int) -> GruDesc:
    hp = 1
    while hp < hidden:
        hp <<= 1
    return GruDesc(input_size, hidden, hp, layers)


def gru_supported(input_size: int, hidden: int, layers: int) -> bool:
    return 1 <= input_size <= GRU_MAX_DIM and 1 <= hidden <= GRU_MAX_DIM and 1 <= layers <= GRU_MAX_LAYERS


def gru_param_count(desc) -> int:
    return int(load().asac_gru_param_count(C.byref(desc)))


def gru_backward_workspace(desc, B) -> int:
    return int(load().asac_gru_backward_workspace(C.byref(desc), B))


def _gru_ptrs(weights, desc):
    """weights: per layer (w_ih, w_hh, b_ih, b_hh) contiguous f32 device tensors."""
    arrs = [_PtrArray() for _ in range(4)]
    for l in range(desc.layers):
        for k in range(4):
            t = weights[l][k]
            assert t.is_cuda and t.is_contiguous() and t.dtype == torch.float32
            arrs[k][l] = t.data_ptr()
    return arrs


def _gru_x(x):
    assert x.dim() == 3 and x.stride(2) == 1 and x.dtype == torch.float32
    return _p(x), x.stride(0), x.stride(1)


def _gru_mask(mask):
    if mask is None:
        return None, 0
    assert mask.dim() == 2 and mask.stride(1) == 1 and mask.element_size() == 1
    return _p(mask), mask.stride(0)


def _gru_h0(h0, desc):
    if h0 is None:
        return None, 0
    assert h0.dim() == 3 and h0.shape[1:] == (desc.layers, desc.hidden) and h0.stride(2) == 1
    assert h0.stride(1) == desc.hidden, 'h0: layers x hidden must be dense per row'
    return _p(h0), h0.stride(0)


@_profiled
def gru_forward(desc, weights, x, h0, padding_mask, hn_out, out_top, gates_out):
    """x [B, L, I]; h0 [B, layers, H] (any batch stride) | None; padding_mask bool/u8 [B, L] | None ->
    hn_out [B, L, layers, H], out_top [B, L, H] | None, gates_out [B, L, layers, 5H] | None."""
    wi, wh, bi, bh = _gru_ptrs(weights, desc)
    px, sb, st = _gru_x(x)
    pm, ms = _gru_mask(padding_mask)
    ph, hs = _gru_h0(h0, desc)
    _check(load().asac_gru_forward(C.byref(desc), wi, wh, bi, bh, px, sb, st, ph, hs, pm, ms, x.shape[0],
                                   x.shape[1], _p(hn_out), _p(out_top), _p(gates_out), _stream()), 'asac_gru_forward')


@_profiled
def gru_forward_twin(desc, weights, twin_weights, x, h0, padding_mask, hn_out, out_top, gates_out, twin_hn_out,
                     twin_out_top):
    """`gru_forward` plus a second parameter set (`twin_weights`, inference only) over the same window in
    the same launch -> twin_hn_out [B, L, layers, H], twin_out_top [B, L, H] | None."""
    wi, wh, bi, bh = _gru_ptrs(weights, desc)
    ti, th, tbi, tbh = _gru_ptrs(twin_weights, desc)
    px, sb, st = _gru_x(x)
    pm, ms = _gru_mask(padding_mask)
    ph, hs = _gru_h0(h0, desc)
    _check(load().asac_gru_forward_twin(C.byref(desc), wi, wh, bi, bh, ti, th, tbi, tbh, px, sb, st, ph, hs, pm, ms,
                                        x.shape[0], x.shape[1], _p(hn_out), _p(out_top), _p(gates_out),
                                        _p(twin_hn_out), _p(twin_out_top), _stream()), 'asac_gru_forward_twin')


def _gru_grad_ptrs(grad_tensors, desc):
    if grad_tensors is None:
        return None
    gt = (C.c_void_p * (4 * GRU_MAX_LAYERS))()
    for l in range(desc.layers):
        for k in range(4):
            t = grad_tensors[l][k]
            assert t.is_cuda and t.is_contiguous() and t.dtype == torch.float32
            gt[4 * l + k] = t.data_ptr()
    return gt


@_profiled
def gru_backward(desc, weights, x, h0, padding_mask, hn, gates, grad_hn, grad_top, grad_x, grad_h0, grad_params,
                 grad_tensors, accumulate, workspace):
    """grad_params: packed f32 buffer (written) | None; grad_tensors: per layer (w_ih, w_hh, b_ih, b_hh) gradient
    tensors written / added in place | None — exactly one of the two."""
    wi, wh, bi, bh = _gru_ptrs(weights, desc)
    px, sb, st = _gru_x(x)
    pm, ms = _gru_mask(padding_mask)
    ph, hs = _gru_h0(h0, desc)
    gt = _gru_grad_ptrs(grad_tensors, desc)
    _check(load().asac_gru_backward(C.byref(desc), wi, wh, bi, bh, px, sb, st, ph, hs, pm, ms, x.shape[0],
                                    x.shape[1], _p(hn), _p(gates), _p(grad_hn), _p(grad_top), _p(grad_x), _p(grad_h0),
                                    _p(grad_params), gt, int(bool(accumulate)), _p(workspace), _stream()),
           'asac_gru_backward')


@_profiled
def gru_backward_at(desc, weights, x, h0, padding_mask, hn, gates, grad_top_members, position, grad_x, grad_h0,
                    grad_params, grad_tensors, accumulate, workspace, adam=None):
    """`gru_backward` for an output gradient that lives at ONE window position: grad_top_members [E, B, H] are the
    ensemble members' gradients of out_top[:, position] (summed inside the launch); the recursion starts there.
    `adam` (`adam_epilogue`, with `grad_tensors`): the launch that finishes the gradients also steps those parameters."""
    wi, wh, bi, bh = _gru_ptrs(weights, desc)
    px, sb, st = _gru_x(x)
    pm, ms = _gru_mask(padding_mask)
    ph, hs = _gru_h0(h0, desc)
    m = grad_top_members
    assert m.dim() == 3 and m.is_contiguous() and m.dtype == torch.float32 and m.shape[1:] == (x.shape[0], desc.hidden)
    gt = _gru_grad_ptrs(grad_tensors, desc)
    if adam is not None:
        assert grad_tensors is not None
        for l in range(desc.layers):
            for t in grad_tensors[l]:
                assert adam._span[0] <= t.data_ptr() and t.data_ptr() + 4 * t.numel() <= adam._span[1], \
                    'adam epilogue: a gradient tensor outside the flat gradient buffer'
    _check(load().asac_gru_backward_at(C.byref(desc), wi, wh, bi, bh, px, sb, st, ph, hs, pm, ms, x.shape[0],
                                       x.shape[1], _p(hn), _p(gates), _p(m), m.shape[0], int(position), _p(grad_x),
                                       _p(grad_h0), _p(grad_params), gt, int(bool(accumulate)),
                                       C.byref(adam) if adam is not None else None, _p(workspace),
                                       _stream()), 'asac_gru_backward_at')


@_profiled
def gauss_head_fwd(raw, A, loc, scale):
    _check(load().asac_gauss_head_fwd(_p(raw), raw.numel() // (2 * A), A, _p(loc), _p(scale), _stream()),
           'asac_gauss_head_fwd')


@_profiled
def gauss_head_bwd(raw, grad_loc, grad_scale, A, grad_raw):
    _check(load().asac_gauss_head_bwd(_p(raw), _p(grad_loc), _p(grad_scale), raw.numel() // (2 * A), A,
                                      _p(grad_raw), _stream()), 'asac_gauss_head_bwd')


@_profiled
def policy_loss_fwd_bwd(logp, q, subset, E_sample, log_alpha, scale, loss_out, grad_logp, grad_q, entropy_out):
    E, B = q.shape
    A = scale.shape[-1] if scale is not None else 0
    rs = scale.stride(-2) if scale is not None else 0
    _check(load().asac_policy_loss_fwd_bwd(_p(logp), _p(q), _p(subset), E, E_sample, B, _p(log_alpha), _p(scale),
                                           rs, A, _p(loss_out), _p(grad_logp), _p(grad_q), _p(entropy_out), _stream()),
           'asac_policy_loss_fwd_bwd')


@_profiled
def alpha_grad(logp, target, grad_slot):
    _check(load().asac_alpha_grad(_p(logp), logp.numel(), float(target), _p(grad_slot), _stream()),
           'asac_alpha_grad')


@_profiled
def noise_fill(seed, step_counter, uniform_out, normal_out, subsets_out=None, ensemble=0):
    """uniform_out: f64 tensor | None; normal_out: f32 tensor | None (both dense); `step_counter` i64[1] on
    the device selects the block of the Philox stream.  subsets_out: i32 [k, E_sample] | None — k random
    E_sample-subsets of range(ensemble)."""
    assert step_counter.dtype == torch.int64
    nu = 0 if uniform_out is None else uniform_out.numel()
    nn_ = 0 if normal_out is None else normal_out.numel()
    assert (uniform_out is None or (uniform_out.dtype == torch.float64 and uniform_out.is_contiguous()))
    assert (normal_out is None or (normal_out.dtype == torch.float32 and normal_out.is_contiguous()))
    ns = es = 0
    if subsets_out is not None:
        assert subsets_out.dtype == torch.int32 and subsets_out.is_contiguous() and subsets_out.dim() == 2
        ns, es = subsets_out.shape
    _check(load().asac_noise_fill(C.c_uint64(int(seed) & (2 ** 64 - 1)), _p(step_counter), _p(uniform_out), nu,
                                  _p(normal_out), nn_, _p(subsets_out), ns, es, int(ensemble), _stream()),
           'asac_noise_fill')


@_profiled
def step_prologue(polyak, zero, seed, step_counter, uniform_out, normal_out, subsets_out=None, ensemble=0):
    """`polyak` (= (target_flat, source_flat, tau) | None) + a memset of `zero` (flat f32 | None) + `noise_fill`
    in one launch."""
    target_flat, source_flat, tau = polyak if polyak is not None else (None, None, 0.0)
    if polyak is not None:
        assert target_flat.is_contiguous() and source_flat.is_contiguous() \
            and target_flat.numel() == source_flat.numel()
    assert zero is None or (zero.is_contiguous() and zero.dtype == torch.float32)
    nu = 0 if uniform_out is None else uniform_out.numel()
    nn_ = 0 if normal_out is None else normal_out.numel()
    ns = es = 0
    if subsets_out is not None:
        assert subsets_out.dtype == torch.int32 and subsets_out.is_contiguous() and subsets_out.dim() == 2
        ns, es = subsets_out.shape
    _check(load().asac_step_prologue(_p(target_flat), _p(source_flat), 0 if polyak is None else target_flat.numel(),
                                     float(tau), _p(zero), 0 if zero is None else zero.numel(),
                                     C.c_uint64(int(seed) & (2 ** 64 - 1)), _p(step_counter), _p(uniform_out), nu,
                                     _p(normal_out), nn_, _p(subsets_out), ns, es, int(ensemble), _stream()),
           'asac_step_prologue')


PROLOGUE_SAMPLE_MAX_BATCH = 1024


@_profiled
def step_prologue_sample(polyak, zero, seed, step_counter, uniform_out, normal_out, subsets_out, ensemble, tree,
                         capacity, batch, slot_ids, beta_state, beta_increment, leaf_out, p_out, ids_out, is_weights_out,
                         min_p_out):
    """`step_prologue` + the single-workgroup `sumtree_sample` (batch <= 1024, IS weights fused) in one launch; the
    sampler draws its `batch` uniforms itself and stores them in `uniform_out`."""
    target_flat, source_flat, tau = polyak if polyak is not None else (None, None, 0.0)
    assert uniform_out.numel() == batch and uniform_out.dtype == torch.float64 and batch <= PROLOGUE_SAMPLE_MAX_BATCH
    assert min_p_out.numel() >= 528 and min_p_out.dtype == torch.float32      # [2..9]: the workgroups' exchange
    assert zero is None or (zero.is_contiguous() and zero.dtype == torch.float32)
    nn_ = 0 if normal_out is None else normal_out.numel()
    ns = es = 0
    if subsets_out is not None:
        assert subsets_out.dtype == torch.int32 and subsets_out.is_contiguous() and subsets_out.dim() == 2
        ns, es = subsets_out.shape
    _check(load().asac_step_prologue_sample(
        _p(target_flat), _p(source_flat), 0 if polyak is None else target_flat.numel(), float(tau), _p(zero),
        0 if zero is None else zero.numel(), C.c_uint64(int(seed) & (2 ** 64 - 1)), _p(step_counter), _p(uniform_out),
        _p(normal_out), nn_, _p(subsets_out), ns, es, int(ensemble), _p(tree), capacity, batch, _p(slot_ids),
        _p(beta_state), float(beta_increment), _p(leaf_out), _p(p_out), _p(ids_out), _p(is_weights_out), _p(min_p_out),
        _stream()), 'asac_step_prologue_sample')


@_profiled
def step_prologue_sample_partial(polyak, zero, seed, step_counter, uniform_out, normal_out, subsets_out, ensemble, tree,
                                 capacity, batch, slot_ids, leaf_out, p_out, ids_out, min_p_out):
    """`step_prologue_sample` for 256 < batch <= 1024 without the weights: the sampler workgroups leave their minima in
    `min_p_out[2:]`; `window_gather_pad_w` (the next launch) forms the weights and advances beta"""
    target_flat, source_flat, tau = polyak if polyak is not None else (None, None, 0.0)
    assert uniform_out.numel() == batch and uniform_out.dtype == torch.float64 and 256 < batch <= PROLOGUE_SAMPLE_MAX_BATCH
    assert min_p_out.numel() >= 528 and min_p_out.dtype == torch.float32
    assert zero is None or (zero.is_contiguous() and zero.dtype == torch.float32)
    nn_ = 0 if normal_out is None else normal_out.numel()
    ns = es = 0
    if subsets_out is not None:
        assert subsets_out.dtype == torch.int32 and subsets_out.is_contiguous() and subsets_out.dim() == 2
        ns, es = subsets_out.shape
    _check(load().asac_step_prologue_sample_partial(
        _p(target_flat), _p(source_flat), 0 if polyak is None else target_flat.numel(), float(tau), _p(zero),
        0 if zero is None else zero.numel(), C.c_uint64(int(seed) & (2 ** 64 - 1)), _p(step_counter), _p(uniform_out),
        _p(normal_out), nn_, _p(subsets_out), ns, es, int(ensemble), _p(tree), capacity, batch, _p(slot_ids),
        _p(leaf_out), _p(p_out), _p(ids_out), _p(min_p_out), _stream()), 'asac_step_prologue_sample_partial')


@_profiled
def window_gather_pad_w(keys, ids, batch, prev_n, post_n, capacity, index_ring, p, tree, beta_state, beta_increment,
                        is_weights_out, min_p_out):
    """`window_gather_pad` + the IS weights of the batch `step_prologue_sample_partial` drew, one launch"""
    _check(load().asac_window_gather_pad_w(keys, len(keys), _p(ids), batch, prev_n, post_n, capacity, _p(index_ring), _p(p),
                                           _p(tree), _p(beta_state), float(beta_increment), _p(is_weights_out),
                                           _p(min_p_out), _stream()), 'asac_window_gather_pad_w')


def graph_launch(graph_exec: int):
    """Replay an instantiated hipGraphExec_t (raw pointer value) on torch's current stream."""
    _check(load().asac_graph_launch(C.c_void_p(int(graph_exec)), _stream()), 'asac_graph_launch')


def graph_replace_memset_nodes(graph: int):
    """Fix-up pass over a captured hipGraph_t (raw pointer value, not yet instantiated): 1-D memset nodes -> kernel nodes
    (csrc/graph_fix.hip: captured memsets take effect on the first launch only on this ROCm).  -> (replaced, kept)"""
    rep, kept = C.c_int(0), C.c_int(0)
    _check(load().asac_graph_replace_memset_nodes(C.c_void_p(int(graph)), C.byref(rep), C.byref(kept)),
           'asac_graph_replace_memset_nodes')
    return rep.value, kept.value


@_profiled
def alpha_adam_step(logp, target, slot, param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, steps_done,
                    advance_counter=False):
    """param / grad / exp_avg / exp_avg_sq: the temperature segment [n]; grad[slot] <- mean(-logp) - target,
    then Adam on the segment (single launch); optionally advances the shared step counter afterwards."""
    assert steps_done.dtype == torch.int64 and param.numel() == grad.numel()
    _check(load().asac_alpha_adam_step(_p(logp), logp.numel(), float(target), int(slot), _p(param), _p(grad),
                                       _p(exp_avg), _p(exp_avg_sq), param.numel(), lr, beta1, beta2, eps,
                                       _p(steps_done), int(bool(advance_counter)), _stream()), 'asac_alpha_adam_step')


@_profiled
def polyak(target_flat, source_flat, tau):
    assert target_flat.numel() == source_flat.numel() and target_flat.dtype == torch.float32
    _check(load().asac_polyak(_p(target_flat), _p(source_flat), target_flat.numel(), float(tau), _stream()),
           'asac_polyak')


@_profiled
def adam_step(param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, steps_done):
    assert steps_done.dtype == torch.int64
    _check(load().asac_adam_step(_p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), param.numel(), lr,
                                 beta1, beta2, eps, _p(steps_done), _stream()), 'asac_adam_step')


# ------------------------------------------------------------------------------------------------
# observation decoder of the prediction models (csrc/decoder.hip)
# ------------------------------------------------------------------------------------------------
OBS_DECODER_SHAPES = ((64, None), (64,), (128, 64), (128,), (32, 32, 4, 4), (32,), (32, 16, 8, 8), (16,), (16, 3, 3, 3), (3,))
# multiply-adds of one state's pass through the three transposed convolutions and the dense head
OBS_DECODER_MACS = 64 * 16 + 128 * 64 + 4 * 16 * 32 * 32 + 36 * 64 * 32 * 16 + 784 * 9 * 16 * 3


def _obs_decoder_params(tensors, state_size) -> ObsDecoderParams:
    assert len(tensors) == 10
    for t, shape in zip(tensors, OBS_DECODER_SHAPES):
        want = tuple(state_size if d is None else d for d in shape)
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and tuple(t.shape) == want, (t.shape, want)
    ps = ObsDecoderParams(*[t.data_ptr() for t in tensors])
    ps._keep = tuple(tensors)
    return ps


def obs_decoder_packed_floats() -> int:
    return int(load().asac_obs_decoder_packed_floats())


def obs_decoder_saved_floats(N) -> int:
    return int(load().asac_obs_decoder_saved_floats(N))


def obs_decoder_workspace_floats(N) -> int:
    return int(load().asac_obs_decoder_workspace_floats(N))


@_profiled
def obs_decoder_forward(state, params, packed, saved, frames):
    """state [N, S] -> frames [N, 3, 30, 30]; `packed` / `saved`: see include/asac_hip.h"""
    global _last_work
    N, S = state.shape
    _last_work = 2.0 * N * OBS_DECODER_MACS
    _dense_f32(packed, saved, frames)
    assert state.is_cuda and state.dtype == torch.float32 and state.stride(1) == 1 and frames.shape == (N, 3, 30, 30)
    ps = _obs_decoder_params(params, S)
    _check(load().asac_obs_decoder_forward(_p(state), state.stride(0), N, S, C.byref(ps), _p(packed), _p(saved),
                                           _p(frames), _stream()), 'asac_obs_decoder_forward')


@_profiled
def obs_decoder_backward(state, packed, saved, frames, grad_frames, grad_state, grad_params, workspace, accumulate=False):
    """grad_frames [N, 3, 30, 30] -> grad_state [N, S] (or None) and the ten parameter gradients"""
    global _last_work
    N, S = state.shape
    _last_work = 4.0 * N * OBS_DECODER_MACS
    _dense_f32(packed, saved, frames, grad_frames, grad_state, workspace)
    assert state.is_cuda and state.dtype == torch.float32 and state.stride(1) == 1
    gs = _obs_decoder_params(grad_params, S)
    _check(load().asac_obs_decoder_backward(_p(state), state.stride(0), N, S, _p(packed), _p(saved), _p(frames),
                                            _p(grad_frames), _p(grad_state), C.byref(gs), int(bool(accumulate)),
                                            _p(workspace), _stream()), 'asac_obs_decoder_backward')


GATE_MAX_LOSSES, GATE_MAX_N = 4, 1 << 20


@_profiled
def cosine_gate_add(main, aux_list, grad, gates_out=None):
    """grad += sum_k clamp(sign(cos(main, aux_k)), min=0) * aux_k, k in order (flat f32 tensors of one length)"""
    n = main.numel()
    assert 0 < len(aux_list) <= GATE_MAX_LOSSES and 0 < n <= GATE_MAX_N
    _dense_f32(main, grad, gates_out, *aux_list)
    assert grad.numel() == n and all(t.numel() == n for t in aux_list)
    ptrs = (C.c_void_p * len(aux_list))(*[t.data_ptr() for t in aux_list])
    _check(load().asac_cosine_gate_add(_p(main), ptrs, len(aux_list), n, _p(grad), _p(gates_out), _stream()),
           'asac_cosine_gate_add')


# ------------------------------------------------------------------------------------------------
# GRU recurrence for hidden 32 / 64 / 128 (csrc/gru_wide.hip)
# ------------------------------------------------------------------------------------------------
def gru_wide_supported(hidden: int) -> bool:
    return bool(load().asac_gru_wide_supported(int(hidden)))


def _mask_ptr(mask):
    if mask is None:
        return None, 0
    assert mask.is_cuda and mask.element_size() == 1 and mask.dim() == 2 and mask.stride(1) == 1
    return _p(mask), mask.stride(0)


@_profiled
def gru_wide_forward(gi, w_hh, b_hh, h0, mask, out, h_raw, gates):
    """one layer's recurrence: gi [B, L, 3H] -> out [B, L, H] (a strided view is fine); see include/asac_hip.h"""
    global _last_work
    B, L, H3 = gi.shape
    H = H3 // 3
    _last_work = 2.0 * B * L * 3 * H * H
    assert gi.stride(2) == 1 and out.stride(2) == 1 and out.shape == (B, L, H)
    _dense_f32(w_hh, b_hh, h_raw, gates)
    pm, ms = _mask_ptr(mask)
    _check(load().asac_gru_wide_forward(_p(gi), gi.stride(0), gi.stride(1), _p(w_hh), _p(b_hh), _p(h0),
                                        0 if h0 is None else h0.stride(0), pm, ms, B, L, H, _p(out), out.stride(0),
                                        out.stride(1), _p(h_raw), _p(gates), _stream()), 'asac_gru_wide_forward')


@_profiled
def gru_wide_forward_twin(gi2, w_hh, b_hh, w_hh_twin, b_hh_twin, h0, mask, out2, h_raw, gates):
    """the recurrence of ONE layer of two networks over the same B windows: gi2 / out2 [2B, L, .] (network 1 then its twin),
    h0 / mask [B, .], h_raw / gates [B, L, .] (network 1) or None"""
    global _last_work
    B2, L, H3 = gi2.shape
    B, H = B2 // 2, H3 // 3
    _last_work = 2.0 * B2 * L * 3 * H * H
    assert gi2.stride(2) == 1 and out2.stride(2) == 1 and out2.shape == (B2, L, H) and B % 16 == 0
    _dense_f32(w_hh, b_hh, w_hh_twin, b_hh_twin, h_raw, gates)
    pm, ms = _mask_ptr(mask)
    _check(load().asac_gru_wide_forward_twin(_p(gi2), gi2.stride(0), gi2.stride(1), _p(w_hh), _p(b_hh), _p(w_hh_twin),
                                             _p(b_hh_twin), _p(h0), 0 if h0 is None else h0.stride(0), pm, ms, B, L, H,
                                             _p(out2), out2.stride(0), out2.stride(1), _p(h_raw), _p(gates), _stream()),
           'asac_gru_wide_forward_twin')


@_profiled
def gru_wide_backward(grad_out, w_hh_t, gates, h_raw, h0, mask, grad_gi, grad_gh, grad_h0):
    global _last_work
    B, L, H = grad_out.shape
    _last_work = 2.0 * B * L * 3 * H * H
    assert grad_out.stride(2) == 1
    _dense_f32(w_hh_t, gates, h_raw, grad_gi, grad_gh, grad_h0)
    pm, ms = _mask_ptr(mask)
    _check(load().asac_gru_wide_backward(_p(grad_out), grad_out.stride(0), grad_out.stride(1), _p(w_hh_t), _p(gates),
                                         _p(h_raw), _p(h0), 0 if h0 is None else h0.stride(0), pm, ms, B, L, H,
                                         _p(grad_gi), _p(grad_gh), _p(grad_h0), _stream()), 'asac_gru_wide_backward')


# ------------------------------------------------------------------------------------------------
# multi-head attention core (csrc/attn_mh.hip)
# ------------------------------------------------------------------------------------------------
def attention_mh_supported(Lq, Lk, heads, head_dim) -> bool:
    return bool(load().asac_attention_mh_supported(int(Lq), int(Lk), int(heads), int(head_dim)))


def _mask3(mask, B):
    """[1 | B, 1 | Lq, Lk] byte mask -> (pointer, stride_b, stride_q, stride_k) with broadcast strides of 0"""
    if mask is None:
        return None, 0, 0, 0
    assert mask.is_cuda and mask.element_size() == 1 and mask.dim() == 3
    return (_p(mask), 0 if mask.shape[0] == 1 and B > 1 else mask.stride(0), 0 if mask.shape[1] == 1 else mask.stride(1),
            mask.stride(2))


@_profiled
def attention_mh_forward(q, k, v, mask, heads, out, weights, keep, p_heads, row_zero=None, keep_rows=None):
    global _last_work
    B, Lq, E = q.shape
    Lk = k.shape[1]
    _last_work = 4.0 * B * Lq * Lk * E
    _dense_f32(q, k, v, out, weights, keep, p_heads)
    pm, sb, si, sj = _mask3(mask, B)
    if row_zero is not None:
        assert row_zero.shape == (B, Lq) and row_zero.is_contiguous() and row_zero.element_size() == 1 and keep_rows is not None
    _check(load().asac_attention_mh_forward(_p(q), _p(k), _p(v), pm, sb, si, sj, B, Lq, Lk, heads, E // heads, _p(out),
                                            _p(weights), _p(keep), _p(p_heads), _p(row_zero), _p(keep_rows), _stream()),
           'asac_attention_mh_forward')


def attention_mh_proj_supported(Lq, Lk, heads, head_dim) -> bool:
    return bool(load().asac_attention_mh_proj_supported(int(Lq), int(Lk), int(heads), int(head_dim)))


@_profiled
def attention_mh_proj_forward(x, weights, biases, mask, heads, q, k, v, out, attn_weights, keep, p_heads, row_zero=None,
                              keep_rows=None, out_weight=None, out_bias=None, y=None, pre=None):
    """`attention_mh_forward` of q / k / v = the three projections of x [B, Lk, E] (q: its last Lq positions), formed in the
    same launch and written to q / k / v for the backward"""
    global _last_work
    B, Lk, E = x.shape
    Lq = q.shape[1]
    assert x.stride(2) == 1 and len(weights) == len(biases) == 3
    _last_work = 4.0 * B * Lq * Lk * E + 2.0 * B * (Lq + 2 * Lk) * E * E
    _dense_f32(*weights, *biases, q, k, v, out, attn_weights, keep, p_heads, out_weight, out_bias, y, pre)
    if out_weight is not None:
        _last_work += 2.0 * B * Lq * E * E
    pm, sb, si, sj = _mask3(mask, B)
    if row_zero is not None:      # (a slice of a wider mask is read in place: batch stride)
        assert row_zero.shape == (B, Lq) and row_zero.stride(1) == 1 and row_zero.element_size() == 1 and keep_rows is not None
    _check(load().asac_attention_mh_proj_forward(_p(x), x.stride(0), x.stride(1), _ptr_array(weights), _ptr_array(biases), pm, sb,
                                                 si, sj, B, Lq, Lk, heads, E // heads, _p(q), _p(k), _p(v), _p(out),
                                                 _p(attn_weights), _p(keep), _p(p_heads), _p(row_zero),
                                                 0 if row_zero is None else row_zero.stride(0), _p(keep_rows),
                                                 _p(out_weight), _p(out_bias), _p(y), _p(pre), _stream()),
           'asac_attention_mh_proj_forward')


@_profiled
def attention_mh_block_backward(q, k, v, mask, heads, p_heads, grad_y, pre, row_scale, out_weight, grad_weights, proj_weights,
                                grad_q, grad_k, grad_v, grad_pre, grad_x):
    """the backward of `attention_mh_proj_forward` with its output block: ResBlock backward, core backward and the projections'
    input gradient as one launch"""
    global _last_work
    B, Lq, E = q.shape
    Lk = k.shape[1]
    _last_work = 10.0 * B * Lq * Lk * E + 2.0 * B * (2 * Lq + 2 * Lk) * E * E
    _dense_f32(q, k, v, p_heads, pre, row_scale, out_weight, grad_weights, *proj_weights, grad_q, grad_k, grad_v, grad_pre, grad_x)
    assert grad_y.shape == q.shape and grad_y.stride(2) == 1 and grad_y.dtype == torch.float32 and grad_y.is_cuda
    pm, sb, si, sj = _mask3(mask, B)
    _check(load().asac_attention_mh_block_backward(_p(q), _p(k), _p(v), pm, sb, si, sj, B, Lq, Lk, heads, E // heads, _p(p_heads),
                                                   _p(grad_y), grad_y.stride(0), grad_y.stride(1), _p(pre), _p(row_scale),
                                                   _p(out_weight), _p(grad_weights),
                                                   _ptr_array(proj_weights), _p(grad_q), _p(grad_k), _p(grad_v), _p(grad_pre),
                                                   _p(grad_x), _stream()), 'asac_attention_mh_block_backward')


@_profiled
def attention_mh_backward(q, k, v, mask, heads, p_heads, grad_out, grad_weights, grad_q, grad_k, grad_v):
    global _last_work
    B, Lq, E = q.shape
    Lk = k.shape[1]
    _last_work = 10.0 * B * Lq * Lk * E
    _dense_f32(q, k, v, p_heads, grad_out, grad_weights, grad_q, grad_k, grad_v)
    pm, sb, si, sj = _mask3(mask, B)
    _check(load().asac_attention_mh_backward(_p(q), _p(k), _p(v), pm, sb, si, sj, B, Lq, Lk, heads, E // heads, _p(p_heads),
                                             _p(grad_out), _p(grad_weights), _p(grad_q), _p(grad_k), _p(grad_v), _stream()),
           'asac_attention_mh_backward')


# ------------------------------------------------------------------------------------------------
# the Linear layers around the multi-head attention core (csrc/rows_proj.hip)
# ------------------------------------------------------------------------------------------------
def rows_proj_supported(width) -> bool:
    return bool(load().asac_rows_proj_supported(int(width)))


def _ptr_array(tensors):
    return (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


@_profiled
def rows_proj_forward(x, weights, biases, tails, outs):
    """outs[j] [B][tails[j]][E] = x[:, L - tails[j]:] weights[j]^T + biases[j] for the <= 3 jobs, one launch; x [B][L][E] with
    feature stride 1"""
    global _last_work
    B, L, E = x.shape
    assert x.stride(2) == 1 and len(weights) == len(biases) == len(tails) == len(outs) <= 3
    _dense_f32(*weights, *biases, *outs)
    for o, n in zip(outs, tails):
        assert o.shape == (B, n, E)
    _last_work = 2.0 * B * sum(tails) * E * E
    _check(load().asac_rows_proj_forward(_p(x), x.stride(0), x.stride(1), B, L, E, len(outs), _ptr_array(weights),
                                         _ptr_array(biases), (C.c_int * len(tails))(*tails), _ptr_array(outs), _stream()),
           'asac_rows_proj_forward')


@_profiled
def rows_proj_backward(grads, tails, weights, grad_x):
    """grad_x [B][L][E] = sum_j grads[j] weights[j] (job j reaching the newest tails[j] positions), one launch"""
    global _last_work
    B, L, E = grad_x.shape
    _dense_f32(*grads, *weights, grad_x)
    for g, n in zip(grads, tails):
        assert g.shape == (B, n, E)
    _last_work = 2.0 * B * sum(tails) * E * E
    _check(load().asac_rows_proj_backward(_ptr_array(grads), (C.c_int * len(tails))(*tails), len(grads), _ptr_array(weights), B,
                                          L, E, _p(grad_x), _stream()), 'asac_rows_proj_backward')


def rows_affine_supported(K, N) -> bool:
    return bool(load().asac_rows_affine_supported(int(K), int(N)))


@_profiled
def rows_affine_forward(x, weight, bias, y):
    """y [rows, N] = x [rows, K] weight^T + bias (K <= 64, N a multiple of 16): one launch; x may have a row stride"""
    global _last_work
    rows, K = x.shape
    N = weight.shape[0]
    assert x.stride(1) == 1 and x.dtype == torch.float32 and x.is_cuda and y.shape == (rows, N)
    _dense_f32(weight, bias, y)
    _last_work = 2.0 * rows * K * N
    _check(load().asac_rows_affine_forward(_p(x), x.stride(0), K, _p(weight), _p(bias), rows, N, _p(y), _stream()),
           'asac_rows_affine_forward')


@_profiled
def rows_affine_gelu_forward(x, weight, bias, y, pre):
    """y = gelu(x weight^T + bias), pre = x weight^T + bias; as `rows_affine_forward`"""
    global _last_work
    rows, K = x.shape
    N = weight.shape[0]
    assert x.stride(1) == 1 and x.dtype == torch.float32 and x.is_cuda and y.shape == (rows, N) and pre.shape == (rows, N)
    _dense_f32(weight, bias, y, pre)
    _last_work = 2.0 * rows * K * N
    _check(load().asac_rows_affine_gelu_forward(_p(x), x.stride(0), K, _p(weight), _p(bias), rows, N, _p(y), _p(pre), _stream()),
           'asac_rows_affine_gelu_forward')


@_profiled
def rows_resblock_forward(x, weight, bias, row_scale, y, pre):
    """y = (x + gelu(x W^T + b)) * row_scale[:, None], pre = x W^T + b; x [rows][E] with feature stride 1"""
    global _last_work
    rows, E = x.shape
    assert x.stride(1) == 1
    _dense_f32(weight, bias, y, pre, *([] if row_scale is None else [row_scale]))
    _last_work = 2.0 * rows * E * E
    _check(load().asac_rows_resblock_forward(_p(x), x.stride(0), _p(weight), _p(bias), _p(row_scale), rows, E, _p(y), _p(pre),
                                             _stream()), 'asac_rows_resblock_forward')


@_profiled
def rows_resblock_backward(grad_y, pre, weight, row_scale, grad_x, grad_pre):
    global _last_work
    rows, E = grad_y.shape
    _dense_f32(grad_y, pre, weight, grad_x, grad_pre, *([] if row_scale is None else [row_scale]))
    _last_work = 2.0 * rows * E * E
    _check(load().asac_rows_resblock_backward(_p(grad_y), _p(pre), _p(weight), _p(row_scale), rows, E, _p(grad_x), _p(grad_pre),
                                              _stream()), 'asac_rows_resblock_backward')


# ------------------------------------------------------------------------------------------------
# products over the rows of two tall matrices (csrc/xty.hip)
# ------------------------------------------------------------------------------------------------
def xty_supported(rows, M, N) -> bool:
    return bool(load().asac_xty_supported(int(rows), int(M), int(N)))


def rows_wide_supported(R: int, K: int, N: int) -> bool:
    return bool(load().asac_rows_wide_supported(R, K, N))


def rows_wide_workspace(x, N: int) -> torch.Tensor:
    R, K = x.shape
    return torch.empty(int(load().asac_rows_wide_workspace(R, K, N)), dtype=torch.float32, device=x.device)


@_profiled
def rows_wide_forward(x, w, b, y, pre=None, act=True, workspace=None):
    """y [R, N] = act(x W^T + b) for a wide x [R, K] (row stride allowed), W [N, K]; `pre`: the pre-activations (training)"""
    global _last_work
    R, K = x.shape
    N = w.shape[0]
    assert x.stride(1) == 1 and w.is_contiguous() and b.is_contiguous() and y.is_contiguous() and y.shape == (R, N)
    assert pre is None or (pre.is_contiguous() and pre.shape == (R, N))
    _last_work = 2.0 * R * K * N
    ws = rows_wide_workspace(x, N) if workspace is None else workspace
    _check(load().asac_rows_wide_forward(_p(x), x.stride(0), R, K, _p(w), _p(b), N, int(bool(act)), _p(y), _p(pre), _p(ws),
                                         _stream()), 'asac_rows_wide_forward')


@_profiled
def rows_wide_backward_input(grad_y, pre, w, dpre, dx=None, act=True):
    """dpre [R, N] = grad_y * act'(pre); dx [R, K] = dpre W (dx None: dpre only)"""
    global _last_work
    R, N = grad_y.shape
    K = w.shape[1]
    assert grad_y.is_contiguous() and dpre.is_contiguous() and w.is_contiguous() and (pre is None or pre.is_contiguous())
    assert dx is None or (dx.stride(1) == 1 and dx.shape == (R, K))
    _last_work = 2.0 * R * K * N if dx is not None else 0.0
    _check(load().asac_rows_wide_backward_input(_p(grad_y), _p(pre), int(bool(act)), R, K, _p(w), N, _p(dpre), _p(dx),
                                                0 if dx is None else dx.stride(0), _stream()), 'asac_rows_wide_backward_input')


@_profiled
def rows_wide_backward_params(dpre, x, dw, db=None, accumulate=False, workspace=None):
    """dw [N, K] (+)= dpre^T x, db [N] (+)= column sums of dpre"""
    global _last_work
    R, K = x.shape
    N = dpre.shape[1]
    assert dpre.is_contiguous() and x.stride(1) == 1 and dw.is_contiguous() and dw.shape == (N, K)
    assert db is None or (db.is_contiguous() and db.numel() == N)
    _last_work = 2.0 * R * K * N
    ws = rows_wide_workspace(x, N) if workspace is None else workspace
    _check(load().asac_rows_wide_backward_params(_p(dpre), _p(x), x.stride(0), R, K, N, _p(dw), _p(db), int(bool(accumulate)),
                                                 _p(ws), _stream()), 'asac_rows_wide_backward_params')


@_profiled
def xty(x, y, out, colsum_x=None, accumulate=False):
    """out [M, N] (+)= x^T y over the rows of x [R, M] and y [R, N] (row strides allowed, dense last dim);
    colsum_x [M] (+)= the column sums of x"""
    global _last_work
    R, M = x.shape
    N = y.shape[1]
    assert y.shape[0] == R and x.stride(1) == 1 and y.stride(1) == 1 and out.shape == (M, N) and out.is_contiguous()
    assert x.dtype == y.dtype == out.dtype == torch.float32 and x.is_cuda
    _last_work = 2.0 * R * M * N
    ws = torch.empty(int(load().asac_xty_workspace(R, M, N)), dtype=torch.float32, device=x.device)
    if colsum_x is not None:
        assert colsum_x.numel() == M and colsum_x.is_contiguous()
    _check(load().asac_xty(_p(x), x.stride(0), M, _p(y), y.stride(0), N, R, _p(out), _p(colsum_x), int(bool(accumulate)), _p(ws),
                           _stream()), 'asac_xty')


@_profiled
def xty_multi(jobs, accumulate=False):
    """`xty(x, y, out, colsum_x)` for up to 4 jobs [(x, y, out, colsum_x | None), ...] as one launch pair (same values)"""
    global _last_work
    n = len(jobs)
    assert 1 <= n <= 4
    work = 0.0
    for x, y, out, cs in jobs:
        R, M = x.shape
        N = y.shape[1]
        assert y.shape[0] == R and x.stride(1) == 1 and y.stride(1) == 1 and out.shape == (M, N) and out.is_contiguous()
        assert x.dtype == y.dtype == out.dtype == torch.float32 and x.is_cuda
        assert cs is None or (cs.numel() == M and cs.is_contiguous())
        work += 2.0 * R * M * N
    _last_work = work
    i64, i32, ptr = C.c_int64 * n, C.c_int * n, C.c_void_p * n
    rows = i64(*[j[0].shape[0] for j in jobs])
    Ms, Ns = i32(*[j[0].shape[1] for j in jobs]), i32(*[j[1].shape[1] for j in jobs])
    ws = torch.empty(int(load().asac_xty_multi_workspace(n, rows, Ms, Ns)), dtype=torch.float32, device=jobs[0][0].device)
    _check(load().asac_xty_multi(n, ptr(*[j[0].data_ptr() for j in jobs]), i64(*[j[0].stride(0) for j in jobs]), Ms,
                                 ptr(*[j[1].data_ptr() for j in jobs]), i64(*[j[1].stride(0) for j in jobs]), Ns, rows,
                                 ptr(*[j[2].data_ptr() for j in jobs]),
                                 ptr(*[None if j[3] is None else j[3].data_ptr() for j in jobs]), int(bool(accumulate)),
                                 _p(ws), _stream()), 'asac_xty_multi')
